"""The per-stream container logic of the product on the CPU tier.

divans_amd/csrc/host_stream.cpp + ffi.cpp (ring-buffer command emission, CMD coder, Mux, CRC, the call-by-call semantics of
divans_encode / divans_encode_flush / divans_decode, the parser of untrusted containers) are built here against
tests/c/hostsim_device_stub.cpp, which answers the nine GPU entry points they use with the CPU oracle -- test infrastructure, see
tests/hostsim.py.  What these tests pin is the HOST code: the same source files the product library is built from, compared with
the oracle's container (itself checked against the independent restatement of tests/ref_container.py), and run under
AddressSanitizer / UBSan against damaged input.  The kernels are not involved; their parity tests are the "-m gpu" tier."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import ffi_harness as fh
import hostsim
import pyoracle as po


_SANITIZER_CANNOT_START = ("ThreadSanitizer: unexpected memory mapping", "ThreadSanitizer: failed to", "Shadow memory range interleaves",
                           "ReserveShadowMemoryRange failed", "LeakSanitizer has encountered a fatal error", "LeakSanitizer does not work under ptrace")


def _run(cmd, **kw):
    """subprocess.run for the sanitizer-instrumented programs: a sanitizer RUNTIME that cannot start in this environment (address-space
    layout it does not know, ptrace restrictions) is a skip, not a finding"""
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0 and any(m in r.stderr for m in _SANITIZER_CANNOT_START):
        pytest.skip("sanitizer runtime cannot start here: " + r.stderr.strip().splitlines()[0][:200])
    return r


@pytest.fixture(scope="module")
def lib():
    return fh.bind(ctypes.CDLL(hostsim.build_library()))


@pytest.fixture(scope="module")
def fuzzer():
    return hostsim.build_fuzzer()


@pytest.mark.parametrize("which", range(len(fh.OPTION_SETS)))
@pytest.mark.parametrize("buf", [65536, 4096, 17])
def test_container_is_the_oracles_for_every_option_set_and_buffer(lib, which, buf, corpus):
    ffi_opts, orc_opts = fh.OPTION_SETS[which]
    data = corpus[:152089] if which == 0 and buf == 65536 else corpus[5000:5000 + 70001]
    coded = fh.ffi_compress(lib, data, ffi_opts, buf_size=buf)
    ref = po.stream_compress_raw(data, po.stream_options(call_buffer_size=buf, **orc_opts))
    assert coded.size == ref.size and (coded == ref).all()
    assert (fh.ffi_decompress(lib, coded, data.size, buf_size=buf, feed=max(buf // 3, 1)) == data).all()


def test_call_patterns_past_the_window(lib, corpus):
    """the ring laps inside the encode calls (raw_to_cmd/mod.rs:55-104): whole-input, 4 KiB and odd pieces at window 16 and 10"""
    import workload
    data = workload.make_blocks(corpus, 7, 24).reshape(-1)                 # 1.5 MiB
    for feed in (None, 4096, 70001):
        coded = fh.ffi_compress(lib, data, [(5, 0), (2, 16)], buf_size=4096, feed=feed)
        calls = [data.size] if feed is None else [feed] * (data.size // feed) + ([data.size % feed] if data.size % feed else [])
        ref = po.stream_compress_raw(data, po.stream_options(window_size=16, call_buffer_size=4096), call_inputs=calls)
        assert coded.size == ref.size and (coded == ref).all(), feed
    assert (fh.ffi_decompress(lib, coded, data.size, buf_size=4096, feed=4096) == data).all()
    for n in (0, 1, 1023, 1024, 1025, 2047, 2048, 3070, 3071, 3072, 5000):
        d = corpus[:n]
        coded = fh.ffi_compress(lib, d, [(5, 0), (2, 10)], buf_size=777)
        assert (coded == po.stream_compress_raw(d, po.stream_options(window_size=10, call_buffer_size=777), call_inputs=[n] if n else None)).all(), n
        assert (fh.ffi_decompress(lib, coded, n, buf_size=100, feed=13) == d).all()


def test_decoder_hands_out_bytes_while_the_input_arrives(lib, corpus):
    """the decoder is incremental: fed 4 KiB at a time, output appears long before the last input byte (a literal-only stream's CMD
    bytes only come with the flush, so nothing can appear before the container's tail has been seen -- but no later than that)"""
    import workload
    data = workload.make_blocks(corpus, 3, 16).reshape(-1)                 # 1 MiB, window 16: 16 laps
    coded = fh.ffi_compress(lib, data, [(5, 0), (2, 16)])
    st = lib.divans_new_decompressor()
    buf = np.empty(65536, np.uint8); out = bytearray(); off = 0
    calls_after_last_input = 0
    while True:
        ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
        n = min(4096, coded.size - off)
        r = lib.divans_decode(st, coded.ctypes.data + off, n, ctypes.byref(ro), buf.ctypes.data, buf.size, ctypes.byref(wo))
        assert r != 3
        off += ro.value; out += buf[:wo.value].tobytes()
        if off == coded.size:
            calls_after_last_input += 1
        if r == 0:
            break
    lib.divans_free_decompressor(st)
    assert bytes(out) == data.tobytes()
    assert calls_after_last_input <= data.size // buf.size + 2           # what is left once the input is in: one buffer per call


def test_damaged_containers_are_refused(lib, corpus):
    data = corpus[:40000]
    coded = fh.ffi_compress(lib, data, [(5, 0)]).copy()

    def decode_result(buf, skip_crc):
        lib.divans_new_decompressor_with_custom_alloc.restype = ctypes.c_void_p
        lib.divans_new_decompressor_with_custom_alloc.argtypes = [fh.CAllocator, ctypes.c_uint8, ctypes.c_uint8]
        st = lib.divans_new_decompressor_with_custom_alloc(fh.CAllocator(None, None, None), skip_crc, 0)
        out = np.empty(1 << 20, np.uint8); ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
        r = lib.divans_decode(st, buf.ctypes.data, buf.size, ctypes.byref(ro), out.ctypes.data, out.size, ctypes.byref(wo))
        lib.divans_free_decompressor(st)
        return r, out[:wo.value]

    r, out = decode_result(coded, 0)
    assert r == 0 and (out == data).all()
    bad = coded.copy(); bad[coded.size // 2] ^= 0x40
    assert decode_result(bad, 0)[0] == 3 and decode_result(bad, 1)[0] == 3
    assert decode_result(coded[:coded.size - 500].copy(), 0)[0] in (1, 3)
    for magic_byte in range(4):                                             # header magic, divans_compressor.rs:126-131
        bad = coded.copy(); bad[magic_byte] ^= 1
        assert decode_result(bad, 0)[0] == 3
    # a stream that decodes to more than the caller's bound is refused, not followed (divans_decompressor_set_max_output_size)
    lib.divans_decompressor_set_max_output_size.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    for bound, expect in ((data.size, 0), (data.size - 1, 3)):
        st = lib.divans_new_decompressor()
        lib.divans_decompressor_set_max_output_size(st, bound)
        out = np.empty(1 << 20, np.uint8); ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
        assert lib.divans_decode(st, coded.ctypes.data, coded.size, ctypes.byref(ro), out.ctypes.data, out.size, ctypes.byref(wo)) == expect
        lib.divans_free_decompressor(st)


@pytest.mark.parametrize("opts,n,iters", [(["5=0"], 3000, 1500), (["5=0", "4=2", "2=10"], 9000, 1500), (["5=0", "7=0", "9=1", "4=0"], 70001, 150)])
def test_fuzzed_containers_under_sanitizers(fuzzer, opts, n, iters, tmp_path, corpus):
    """tests/c/hostsim_fuzz.cpp: random piece / buffer sizes both ways, then damaged containers through divans_decode -- no sanitizer
    report, no spinning, no success with wrong bytes while the CRC is checked"""
    src = tmp_path / "in.bin"
    corpus[1234:1234 + n].tofile(src)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = _run([fuzzer, str(src), str(n), str(iters)] + opts, timeout=900, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "damaged" in r.stdout


@pytest.mark.parametrize("name,iters", [("alice29", 150), ("alice29-priors", 150), ("ends_with_truncated_dictionary", 1500)])
def test_fuzzed_ir_text_under_sanitizers(fuzzer, name, iters, tmp_path):
    """the textual command IR (divans_amd/csrc/ir.cpp, grammar of the reference's bin/divans.rs:191-483) with damaged text: refused or
    accepted, within bounds, no sanitizer report"""
    import lzma
    src = tmp_path / (name + ".ir")
    with lzma.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ir_" + name + ".ir.xz")) as f:
        src.write_bytes(f.read())
    r = _run([fuzzer, "ir", str(src), "7", str(iters)], timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "still parse" in r.stdout


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")


def test_c_harness_and_cpp_adaptors_under_sanitizers(tmp_path, corpus):
    """tests/c/ffi_roundtrip.c (c/example.c's calling pattern with a counting allocator; the GPU tier links it against the product
    library) and examples/io_adaptors.cpp (examples/divans_io.hpp: the reference's writer.rs / reader.rs adaptors as a usage example of
    the drop-in ABI -- this is the one tier that runs it), here against the harness objects: same exit codes, container == the oracle's, nothing for ASan / UBSan / LSan to report"""
    src = tmp_path / "in.bin"
    data = corpus[:200000]
    data.tofile(src)
    exe = hostsim.build_program("ffi_roundtrip", os.path.join(ROOT, "tests", "c", "ffi_roundtrip.c"), lang="c")
    dv = tmp_path / "a.divans"
    r = _run([exe, str(src), str(dv), "5=0", "4=2", "9=0"], env=SAN_ENV, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-3000:])
    assert (po.stream_decompress(np.fromfile(dv, dtype=np.uint8), data.size) == data).all()
    exe = hostsim.build_program("io_adaptors", os.path.join(ROOT, "examples", "io_adaptors.cpp"))
    dv = tmp_path / "b.divans"
    r = _run([exe, str(src), str(dv)], env=SAN_ENV, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-3000:])
    coded = np.fromfile(dv, dtype=np.uint8)
    calls = [65536] * (data.size // 65536) + [data.size % 65536]
    ref = po.stream_compress_raw(data, po.stream_options(window_size=16, dynamic_context_mixing=2, use_context_map=1, call_buffer_size=4096),
                                 call_inputs=calls)
    assert coded.size == ref.size and (coded == ref).all()


REF_C = "/root/reference/c"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_C, "example.c")), reason="the reference tree is not mounted here")
@pytest.mark.parametrize("args,env", [([], {}), (["-l"], {}), (["-l", "-cm", "-m2", "-s1"], {}), (["-l", "-w12"], {"NO_MALLOC": "1"}), ([], {"RUST_MALLOC": "1"})])
def test_reference_example_c_on_the_host_logic(args, env, tmp_path, corpus):
    """the reference's own C harness (c/example.c, compiled verbatim from where it lies) over the harness objects: compress ->
    decompress -> memcmp with its three allocators, under the sanitizers.  The GPU tier runs the same program against the product."""
    exe = hostsim.build_program("ref_example", os.path.join(REF_C, "example.c"), lang="c", sanitize_main=False)
    e = dict(SAN_ENV, ASAN_OPTIONS="detect_leaks=0"); e.update(env)       # the harness does not free its own buffers
    r = _run([exe] + args, env=e, timeout=300)
    assert r.returncode == 0 and "reduced to" in r.stdout, (r.returncode, r.stdout, r.stderr[-3000:])
    src = tmp_path / "in.bin"
    corpus[:100000].tofile(src)
    r = _run([exe] + args + [str(src)], env=e, timeout=300)
    assert r.returncode == 0 and "File length 100000 reduced to" in r.stdout, (r.returncode, r.stdout, r.stderr[-3000:])


def test_stub_restates_the_librarys_host_rules(lib):
    """the stub carries copies of two host-only rules of capi.cpp (which cannot be linked without HIP): the speed rule and the output
    bound; they must not drift from the library's"""
    import divans_amd as da
    real = da.load_library()
    for L in (lib, real):
        L.divans_gpu_speed_supported.argtypes = [ctypes.c_int32, ctypes.c_int32]; L.divans_gpu_speed_supported.restype = ctypes.c_int
        L.divans_gpu_lit_encode_bound.argtypes = [ctypes.c_size_t]; L.divans_gpu_lit_encode_bound.restype = ctypes.c_size_t
    rng = np.random.default_rng(2)
    cases = [(int(a) >> int(s), (int(b) >> int(t)) - 3) for a, b, s, t in zip(rng.integers(-2, 0x8003, 4000), rng.integers(-2, 0x8003, 4000),
                                                                              rng.integers(0, 8, 4000), rng.integers(0, 8, 4000))]
    for inc, lim in cases:
        assert lib.divans_gpu_speed_supported(inc, lim) == real.divans_gpu_speed_supported(inc, lim), (inc, lim)
    for n in list(range(0, 70)) + [32767, 32768, 32769, 65535, 65536, 65537, 1 << 20, (1 << 24) + 5]:
        assert lib.divans_gpu_lit_encode_bound(n) == real.divans_gpu_lit_encode_bound(n), n


def test_independent_states_on_concurrent_threads_under_tsan(tmp_path, corpus):
    """distinct compressor / decompressor states are independent (src/ffi/interface.rs:49-50): four threads of round trips and refused
    damaged streams over the shared codec cache and the per-thread error string, under ThreadSanitizer"""
    exe = hostsim.build_thread_test()
    src = tmp_path / "in.bin"
    corpus[:100000].tofile(src)
    r = _run([exe, str(src), "4", "6"], timeout=600)
    assert r.returncode == 0 and "0 failures" in r.stdout, (r.returncode, r.stdout, r.stderr[-4000:])


def test_decoder_on_random_command_streams_from_the_oracle(lib):
    """containers whose CMD stream is not the internal compressor's: a random PredictionMode (prediction mode, context map for 1-4
    (sometimes up to 39) literal block types coded through the LRU / mnemonic scheme of context_map.rs:264-331, mixing values 0..8, speeds as f8 pairs,
    mixing parameter, prior depth), a BlockSwitchLiteral, then Literal commands of random lengths -- built by the oracle's encoder
    from the command list, decoded by the product's host code (CommandModel, lit_config_from_prediction_mode) call by call"""
    speeds = [(0, 1024), (2, 1024), (1, 128), (1, 16384), (2, 2048), (4, 1024), (8, 8192), (16, 48), (16, 8192), (32, 4096), (64, 16384),
              (128, 256), (128, 16384), (512, 16384), (1664, 16384)]                           # probability/interface.rs:303-320
    L = po.lib()
    L.orc_speed_to_u8.argtypes = [ctypes.c_int16]; L.orc_speed_to_u8.restype = ctypes.c_uint8
    rng = np.random.default_rng(77)
    done = 0
    for case in range(240):
        n_bt = int(rng.integers(1, 5)) if case % 8 else int(rng.integers(14, 40))    # block types past 12 are coded as two nibbles (block_type.rs)
        style = int(rng.integers(0, 3))
        if style == 0:
            cm = (np.arange(64 * n_bt) & 63).astype(np.uint8)
        elif style == 1:
            cm = rng.integers(0, int(rng.integers(1, 64)), 64 * n_bt).astype(np.uint8)        # few clusters: long mnemonic runs
        else:
            cm = np.repeat(rng.integers(0, 256, 8 * n_bt), 8).astype(np.uint8)                # any byte value, runs of eight
        dm = rng.integers(0, 4, 4 * int(rng.integers(1, 5))).astype(np.uint8)
        mix = (rng.integers(0, 9, 8192) if case % 3 else np.full(8192, int(rng.integers(0, 9)))).astype(np.uint8)
        pm = po.PredictionMode()
        pm.prediction_mode = int(rng.integers(0, 4))
        pm.literal_context_map = cm.ctypes.data; pm.n_literal_context_map = cm.size
        pm.distance_context_map = dm.ctypes.data; pm.n_distance_context_map = dm.size
        pm.mixing_values = mix.ctypes.data
        pm.has_context_speeds = int(rng.integers(0, 2))
        for arr in (pm.context_map_speed_f8, pm.stride_speed_f8, pm.combined_stride_speed_f8):
            for i in range(2):
                inc, lim = speeds[int(rng.integers(0, len(speeds)))]
                arr[i][0] = L.orc_speed_to_u8(inc); arr[i][1] = L.orc_speed_to_u8(lim)
        cmds = []
        c = po.StreamCommand(); c.kind = 7; c.pm = pm; cmds.append(c)
        c = po.StreamCommand(); c.kind = 4; c.btype = int(rng.integers(0, n_bt)) if case % 8 else n_bt - 1; c.stride = int(rng.integers(0, 5)); cmds.append(c)
        pieces = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.integers(1, 1 << int(rng.integers(1, 17))))
            kind = int(rng.integers(0, 3))
            d = (rng.integers(0, 256, n) if kind == 0 else np.resize(rng.integers(32, 127, int(rng.integers(1, 40))), n) if kind == 1
                 else rng.integers(97, 123, n)).astype(np.uint8)
            pieces.append(d)
            c = po.StreamCommand(); c.kind = 3; c.data = d.ctypes.data; c.len = d.size; cmds.append(c)
        data = np.concatenate(pieces)
        o = po.stream_options(window_size=int(rng.integers(10, 23)), dynamic_context_mixing=int(rng.integers(0, 3)),
                              prior_depth=int(rng.integers(0, 3)), use_context_map=int(rng.integers(0, 2)), force_stride=int(rng.integers(0, 3)),
                              call_buffer_size=int(rng.integers(1, 70000)))
        try:
            coded = po.stream_compress_commands(cmds, o, keepalive=(pieces, cm, dm, mix))
        except RuntimeError:
            continue                                             # a combination the oracle's encoder refuses (e.g. a speed pair it cannot run)
        assert (po.stream_decompress(coded, data.size) == data).all(), case
        back = fh.ffi_decompress(lib, coded, data.size, buf_size=int(rng.integers(1, 70000)), feed=int(rng.integers(1, 9000)))
        assert (back == data).all(), case
        done += 1
    assert done >= 180


def test_planned_containers_of_the_batch_interface_equal_the_oracles(fuzzer, tmp_path, corpus):
    """divans_batch_compress builds a container in two phases -- plan_stream (CMD coder, order of events; needs no literal data, runs on
    host threads under the GPU work) and assemble_container (Mux replay once the literal chunks exist).  `hostsim_fuzz plan` runs both
    with random lengths, options, call buffers and call patterns, takes the literal bytes from the oracle's literal coder under the
    plan's configuration, and compares the container with the oracle's; parse_container_host then has to recover the plan's
    configuration and the length."""
    src = tmp_path / "in.bin"
    corpus[3000:3000 + 150000].tofile(src)
    r = _run([fuzzer, "plan", str(src), "5", "150"], timeout=900, env=SAN_ENV)
    assert r.returncode == 0 and "all equal to the oracle's" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_option_stage_and_state_helpers(lib, corpus):
    """divans_set_option (src/ffi/compressor.rs:63-166): accepted, ignored and refused selectors and values, the OptionStage rule, the
    brotli knobs changing nothing about the container, the serial decompressor alias and the state allocators (src/ffi/mod.rs:111-145)"""
    vp = ctypes.c_void_p
    data = corpus[2000:2000 + 20000]
    plain = fh.ffi_compress(lib, data, [(5, 0)])
    # front-end knobs are stored and never read by the literal-only compressor
    assert (fh.ffi_compress(lib, data, [(5, 0), (1, 11), (3, 18), (10, 2), (15, 340), (16, 1), (17, 1), (18, 1), (19, 2), (20, 1)]) == plain).all()
    # brotli command selection 1 / 2 and the brotli bitstream setting fall back to the internal selection: same bytes
    assert (fh.ffi_compress(lib, data, [(5, 1)]) == plain).all() and (fh.ffi_compress(lib, data, [(6, 1)]) == plain).all()
    st = lib.divans_new_compressor()
    for sel, val in [(0, 0), (21, 0), (99, 1), (5, 3), (6, 0), (6, 2), (7, 2), (9, 9), (8, 15), (12, 15), (13, 99), (14, 15)]:
        assert lib.divans_set_option(st, sel, val) == 3, (sel, val)
    for sel, val in [(2, 9), (2, 30), (4, 0), (4, 2), (7, 0), (7, 1), (9, 0), (9, 8), (11, 0), (11, 2), (8, 0), (12, 14), (13, 7), (14, 3)]:
        assert lib.divans_set_option(st, sel, val) == 0, (sel, val)
    q = lib.divans_compressor_uses_internal_command_selection_instead_of_brotli
    q.argtypes = [vp]; q.restype = ctypes.c_uint8
    assert q(st) == 1 and lib.divans_set_option(st, 5, 0) == 0 and q(st) == 0 and q(None) == 0
    for fn, rt in (("divans_compressor_malloc_u8", vp), ("divans_compressor_malloc_usize", vp)):
        getattr(lib, fn).argtypes = [vp, ctypes.c_size_t]; getattr(lib, fn).restype = rt
    lib.divans_compressor_free_u8.argtypes = [vp, vp, ctypes.c_size_t]; lib.divans_compressor_free_usize.argtypes = [vp, vp, ctypes.c_size_t]
    p = lib.divans_compressor_malloc_u8(st, 1000); u = lib.divans_compressor_malloc_usize(st, 100)
    assert p and u
    ctypes.memset(p, 0xAB, 1000); ctypes.memset(u, 0xCD, 100 * ctypes.sizeof(ctypes.c_size_t))
    lib.divans_compressor_free_u8(st, p, 1000); lib.divans_compressor_free_usize(st, u, 100)
    lib.divans_free_compressor(st)
    lib.divans_free_compressor(None)
    # window sizes outside [10, 24] are clamped, not refused (divans_compressor.rs:89)
    assert (fh.ffi_decompress(lib, fh.ffi_compress(lib, data, [(5, 0), (2, 3)]), data.size) == data).all()
    assert (fh.ffi_decompress(lib, fh.ffi_compress(lib, data, [(5, 0), (2, 31)]), data.size) == data).all()
    lib.divans_new_serial_decompressor.restype = vp
    ds = lib.divans_new_serial_decompressor()
    for fn in ("divans_decompressor_malloc_u8", "divans_decompressor_malloc_usize"):
        getattr(lib, fn).argtypes = [vp, ctypes.c_size_t]; getattr(lib, fn).restype = vp
    lib.divans_decompressor_free_u8.argtypes = [vp, vp, ctypes.c_size_t]; lib.divans_decompressor_free_usize.argtypes = [vp, vp, ctypes.c_size_t]
    p = lib.divans_decompressor_malloc_u8(ds, 64); u = lib.divans_decompressor_malloc_usize(ds, 8)
    assert p and u
    lib.divans_decompressor_free_u8(ds, p, 64); lib.divans_decompressor_free_usize(ds, u, 8)
    out = np.empty(data.size, np.uint8); ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
    assert lib.divans_decode(ds, plain.ctypes.data, plain.size, ctypes.byref(ro), out.ctypes.data, out.size, ctypes.byref(wo)) == 0
    assert wo.value == data.size and (out == data).all()
    lib.divans_free_decompressor(ds)
    lib.divans_free_decompressor(None)
    # null arguments are failures, not crashes (src/ffi/mod.rs:70-108,236-262)
    assert lib.divans_encode(None, None, 0, None, None, 0, None) == 3 and lib.divans_encode_flush(None, None, 0, None) == 3
    assert lib.divans_decode(None, None, 0, None, None, 0, None) == 3


@pytest.mark.parametrize("sanitizer,rounds,largest,devices", [("address,undefined", 3, 250, 1), ("thread", 2, 16, 2), ("thread", 1, 24, -4),
                                                              ("thread", 1, 40, -8), ("address,undefined", 1, 120, -3)])
def test_batch_interface_on_the_host_logic(sanitizer, rounds, largest, devices, tmp_path, corpus):
    """include/divans_batch.h without a GPU: divans_amd/csrc/batch.cpp itself (length classes, slices on lanes, persistent thread pool,
    plans and parsing under the "GPU work", container assembly, error paths), compiled by g++ against a stand-in for the 16 HIP runtime
    calls it makes (tests/c/fakehip) and the oracle-backed device stub.  tests/c/hostsim_batch.cpp: batches of mixed lengths and options,
    containers == the oracle's, two configurations interleaved through one decompress call, short buffers, a damaged container that
    has to be named -- under AddressSanitizer + UBSan, and again under ThreadSanitizer.  devices = 2: two host threads, each on its own
    stand-in device, at once -- batch.cpp keeps one set of lanes per device (one process drives all of a node's GPUs; the reference's states
    are independent, src/ffi/interface.rs:49-50), a call on one device does not wait for the other's, and divans_batch_release /
    _release_device from one thread wait for the other thread's running call.  devices = -D: D stand-in devices behind ONE call
    (divans_batch_options::device = DIVANS_BATCH_ALL_DEVICES): contiguous ranges, one driving thread per device inside the
    call, each with its budget of the host thread pool; the containers must equal the oracle's AND those of D one-device calls on the D
    ranges, while another thread keeps the last device busy with calls of its own."""
    exe = hostsim.build_batch_test(sanitizer)
    src = tmp_path / "in.bin"
    corpus.tofile(src)
    r = _run([exe, str(src), "9", str(rounds), str(largest), str(devices)], timeout=1500, env=SAN_ENV)
    assert r.returncode == 0 and "all equal to the oracle's and back" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])

