"""CPU-side checks of the C ABI: the in-tree library loads and exports every symbol include/divans_gpu.h declares,
and refuses to compute without a GPU (no silent CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(divans_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import divans_amd as da
    da.build_extension()
    L = da.load_library()
    declared = _declared("divans_gpu.h")
    assert set(declared) == set(da.exported_symbols())
    for sym in declared:
        assert hasattr(L, sym), sym
    # the reference's per-stream ABI (c/divans/ffi.h) and the IR host layer
    for header, expect in (("divans_ffi.h", None), ("divans_ir.h", set(da.exported_ir_symbols())), ("divans_batch.h", set(da.exported_batch_symbols()))):
        decl = _declared(header)
        if expect is not None:
            assert set(decl) == expect
        for sym in decl:
            assert hasattr(L, sym), (header, sym)


def test_config_helpers_match_oracle_configs():
    import divans_amd as da
    import pyoracle as po
    for g, o in ((da.config_simple(), po.config_simple()), (da.config_context_mixing(), po.config_context_mixing())):
        assert bytes(g) == bytes(o)      # same struct layout and contents as the oracle's orc_lit_config


def test_no_cpu_fallback():
    import torch
    import divans_amd as da
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(da.DivansGpuError):
        da.LiteralCodec(da.config_simple(), 4096)
    # and the raw ABI says so too
    L = da.load_library()
    h = ctypes.c_void_p()
    cfg = da.config_simple()
    rc = L.divans_gpu_codec_create(ctypes.byref(h), ctypes.byref(cfg), 0, None, 4096)
    assert rc != 0 and b"no HIP device" in L.divans_gpu_last_error()


def test_encode_bound_is_monotone_and_sufficient_for_worst_case_model():
    import divans_amd as da
    prev = 0
    for n in (0, 1, 2, 100, 32768, 32769, 65536, 1 << 20):
        b = da.encode_bound(n)
        assert b % 16 == 0 and b >= prev
        nsym = 2 * n
        chunks = (nsym + 65535) // 65536
        assert b >= 16 * chunks + 4 * ((15 * nsym + 31) // 32)
        prev = b


def test_host_crc32c_paths_agree():
    """the SSE4.2 path and the table walk of divans_amd/csrc/host_stream.cpp against the known answers of src/codec/crc32.rs"""
    import ctypes
    import numpy as np
    import divans_amd as da
    L = da.load_library()
    L.divans_host_selftest_crc32c.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.divans_host_selftest_crc32c.restype = None
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 4097, 100003):
        for shift in (0, 1, 3):
            buf = rng.integers(0, 256, size=n + 8, dtype=np.uint8)
            a = ctypes.c_uint32(0); b = ctypes.c_uint32(0)
            L.divans_host_selftest_crc32c(buf.ctypes.data + shift, n, ctypes.byref(a), ctypes.byref(b))
            assert a.value == b.value, (n, shift)
    kat = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    a = ctypes.c_uint32(0); b = ctypes.c_uint32(0)
    L.divans_host_selftest_crc32c(kat.ctypes.data, kat.size, ctypes.byref(a), ctypes.byref(b))
    assert a.value == b.value == 0xE3069283          # CRC-32C check value


def test_speed_supported_means_no_i16_wrap():
    """divans_gpu_speed_supported (host-only) against the Python restatement's FrequentistCDF16::blend with its i16 wrapping:
    a supported speed never takes a count of a row out of i16 -- the kernels' 32-bit arithmetic and the reference's wrapping
    arithmetic are then the same thing -- and the others are the ones under which the reference wraps (they run on the wrap-checked
    streaming kernels, divans_gpu_speed_accepted; only a negative increment is refused)."""
    import numpy as np
    import divans_amd as da
    import ref_restatement as rr

    def wraps(inc, lim, steps=400):
        rng = np.random.default_rng(inc * 65537 + lim)
        a = rr.Cdf(); plain = list(a.cdf)                    # plain: the same update in unbounded integers
        for sym in rng.integers(0, 16, steps):
            a.blend(int(sym), (inc, lim))
            for i in range(int(sym), 16):
                plain[i] += inc
            if plain[15] >= lim:
                plain = [(c + i + 1) - ((c + i + 1) >> 2) for i, c in enumerate(plain)]
            if a.cdf != plain:
                return True
        return False

    palette = [(0, 1024), (2, 1024), (1, 128), (1, 16384), (2, 2048), (4, 1024), (1, 0x4000), (4, 0xa00), (0x10, 0x2000), (0x20, 0x1000),
               (0x30, 0x4000), (0x60, 0x4000), (0x80, 0x4000), (0x180, 0x4000)]       # probability/interface.rs:303-328
    for inc, lim in palette:
        assert da.speed_supported(inc, lim) and not wraps(inc, lim), (inc, lim)
    for inc, lim in [(0x4000, 0x4000), (0x3000, 0x1000), (0x2000, 0x100), (0x4000, 1), (8180, 64), (0x3ff0, 0x4000)]:
        assert not da.speed_supported(inc, lim) and wraps(inc, lim), (inc, lim)
    # the reference's debug-only bounds (inc, lim <= 0x4000, probability/interface.rs:341-365) are no part of the rule: only the trajectory is
    for inc, lim in [(1, 0), (3, -5), (1, 0x4001), (16, 0x6000), (0x30, 0x7000)]:
        assert da.speed_supported(inc, lim) and da.speed_accepted(inc, lim) and not wraps(inc, lim, steps=40000 if lim > 0x4000 else 400), (inc, lim)
    for inc, lim in [(0x4001, 100), (1, 0x7fff), (0x7800, 0x7800), (0x5000, 0x7fff)]:
        assert not da.speed_supported(inc, lim) and da.speed_accepted(inc, lim), (inc, lim)
    for inc, lim in [(-1, 100), (-0x8000, 0x2000), (0x8000, 5), (5, 0x8000)]:
        assert not da.speed_supported(inc, lim) and not da.speed_accepted(inc, lim)
    rng = np.random.default_rng(11)
    checked = 0
    for _ in range(300):
        inc = int(rng.integers(0, 0x8000)) >> int(rng.integers(0, 8)); lim = (int(rng.integers(0, 0x8000)) >> int(rng.integers(0, 6))) - int(rng.integers(0, 3)) * 40
        assert da.speed_accepted(inc, lim)
        if da.speed_supported(inc, lim):
            assert not wraps(inc, lim), (inc, lim)
            checked += 1
    assert checked > 50


def test_ffi_reports_the_command_selection_downgrade():
    """the per-stream ABI replaces the brotli front end (the reference's default) by the internal command selection;
    a caller can ask.  Host-only: no stream is coded."""
    import divans_amd as da
    L = da.load_library()
    L.divans_new_compressor.restype = ctypes.c_void_p
    L.divans_set_option.argtypes = [ctypes.c_void_p, ctypes.c_uint8, ctypes.c_uint32]; L.divans_set_option.restype = ctypes.c_uint8
    q = L.divans_compressor_uses_internal_command_selection_instead_of_brotli
    q.argtypes = [ctypes.c_void_p]; q.restype = ctypes.c_uint8
    L.divans_free_compressor.argtypes = [ctypes.c_void_p]
    st = L.divans_new_compressor()
    assert q(st) == 1                                   # default = UseBrotliCommandSelection (src/ffi/compressor.rs:168-178)
    assert L.divans_set_option(st, 5, 0) == 0 and q(st) == 0
    assert L.divans_set_option(st, 5, 2) == 0 and q(st) == 1
    L.divans_free_compressor(st)


def test_public_headers_compile_as_c99_and_cpp11(tmp_path):
    """every header under include/ on its own, pedantic, as C99 and as C++11: a binding generator or a C caller can include any of them first"""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for h in ("divans_gpu.h", "divans_ffi.h", "divans_ir.h", "divans_batch.h"):
        src = tmp_path / "one.c"
        src.write_text('#include "%s"\n' % h)
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + inc, str(src)], check=True)
        subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + inc, "-x", "c++", str(src)], check=True)
