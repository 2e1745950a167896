"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact on both ANS byte streams."""
import numpy as np
import pytest

import pyoracle as po
import workload

pytestmark = pytest.mark.gpu


def _codec(cfg_name, max_len):
    import divans_amd as da
    cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
    return da, da.LiteralCodec(cfg, max_len)


def _oracle_cfg(cfg_name):
    return po.config_simple() if cfg_name == "simple" else po.config_context_mixing()


def _compare(cfg_name, blocks, cache_rows=None, blocks_grid=0, split=None, encode_path=None):
    n, L = blocks.shape
    da, codec = _codec(cfg_name, max(L, 1))
    if encode_path is None and (cache_rows is not None or split is not None):
        encode_path = 1          # the row caches belong to the streaming kernels: keep their encoder in the comparison
    if encode_path is not None:
        codec.set_encode_path(encode_path)
    if cache_rows is not None or blocks_grid:
        codec.set_geometry(blocks=blocks_grid, cache_rows=cache_rows)
    if split is not None:
        codec.set_split_cache(*split)
    packed, offs, sizes = codec.encode_host(blocks, L)
    ocfg = _oracle_cfg(cfg_name)
    for i in range(n):
        ref = po.lit_encode(ocfg, blocks[i])
        got = packed[int(offs[i]):int(offs[i]) + int(sizes[i])]
        assert got.size == ref.size, (cfg_name, i, got.size, ref.size)
        assert (got == ref).all(), (cfg_name, i, int(np.argmax(got != ref)))
    back = codec.decode_host(packed, offs, sizes, L)
    assert (back == blocks).all()
    codec.close()


def test_division_selftest():
    da, codec = _codec("simple", 64)
    assert codec.selftest_division() == 0
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("length", [1, 2, 15, 16, 17, 255, 4096, 32768, 32769, 65536])
def test_lengths_bit_exact(cfg_name, length, corpus):
    blocks = workload.make_blocks(corpus, 3, 5, block_len=length, perturb_per_block=max(length // 100, 0))
    _compare(cfg_name, blocks)


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_many_streams_bit_exact(cfg_name, corpus):
    # more streams than resident groups so every group recycles its table
    blocks = workload.make_blocks(corpus, 0, 9000, block_len=1024)
    _compare(cfg_name, blocks[:9000:7].copy())
    n = blocks.shape[0]
    da, codec = _codec(cfg_name, 1024)
    packed, offs, sizes = codec.encode_host(blocks, 1024)
    back = codec.decode_host(packed, offs, sizes, 1024)
    assert (back == blocks).all()
    # spot-check a spread of streams against the oracle
    ocfg = _oracle_cfg(cfg_name)
    for i in range(0, n, 257):
        ref = po.lit_encode(ocfg, blocks[i])
        assert (packed[int(offs[i]):int(offs[i]) + int(sizes[i])] == ref).all()
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_adversarial_inputs(cfg_name, shuffle384):
    rng = np.random.default_rng(5)
    L = 40000
    rows = [np.zeros(L, np.uint8), np.full(L, 255, np.uint8), np.resize(shuffle384, L),
            rng.integers(0, 256, L, dtype=np.uint8), np.resize(np.arange(256, dtype=np.uint8), L),
            np.resize(np.frombuffer(b"@" * 7 + b"X", dtype=np.uint8), L)]
    _compare(cfg_name, np.stack(rows))


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("cache_rows", [0, 32, 64, 128, 256])
def test_row_cache_sizes_bit_exact(cfg_name, cache_rows, corpus, shuffle384):
    # the LDS row cache (any size, or none) must not change a single coded byte; few resident rows of lanes
    # so that every one of them recycles its table and cache across several streams
    blocks = workload.make_blocks(corpus, 100, 70, block_len=20000)
    blocks[3] = np.resize(shuffle384, 20000)
    blocks[4] = np.random.default_rng(1).integers(0, 256, 20000, dtype=np.uint8)
    _compare(cfg_name, blocks, cache_rows=cache_rows, blocks_grid=2)


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("split", [(16, 0), (32, 0), (32, 64), (128, 16), (0, 0)])
def test_split_row_caches_bit_exact(cfg_name, split, corpus, shuffle384):
    blocks = workload.make_blocks(corpus, 300, 70, block_len=20000)
    blocks[5] = np.resize(shuffle384, 20000)
    blocks[6] = np.random.default_rng(2).integers(0, 256, 20000, dtype=np.uint8)
    _compare(cfg_name, blocks, blocks_grid=2, split=split)


def test_decode_only_random_then_unicode(random_then_unicode):
    # BASELINE.json configs[3]: random_then_unicode pre-encoded under the config-3 options as ceil(291949/65536) = 5
    # blocks (the last one ragged), replicated in device memory, decoded on the GPU, every copy checked bit-exactly
    import torch
    import divans_amd as da
    data = random_then_unicode
    assert data.size == 291949
    L, copies = 65536, 96
    ocfg = po.config_context_mixing()
    pieces = [data[i:i + L] for i in range(0, data.size, L)]
    coded = [po.lit_encode(ocfg, p) for p in pieces]
    dev = torch.device("cuda", 0)
    offs, sizes, blob, pos = [], [], [], 0
    for _ in range(copies):
        for c in coded:
            offs.append(pos); sizes.append(c.size); blob.append(c); pos += c.size      # sizes are multiples of 4
    d_coded = torch.from_numpy(np.concatenate(blob + [np.zeros(64, np.uint8)])).to(dev)
    n = len(offs)
    out_sizes = [p.size for p in pieces] * copies
    out_offs = np.concatenate([[0], np.cumsum(out_sizes)[:-1]])
    d_out = torch.zeros(int(sum(out_sizes)) + 64, dtype=torch.uint8, device=dev)
    codec = da.LiteralCodec(da.config_context_mixing(), L)
    codec.decode_batch(d_coded, torch.tensor(offs, dtype=torch.int64, device=dev), torch.tensor(sizes, dtype=torch.int32, device=dev), n, L,
                       d_out, torch.tensor(out_offs, dtype=torch.int64, device=dev), torch.tensor(out_sizes, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()[:sum(out_sizes)].reshape(copies, data.size)
    assert (got == data[None, :]).all()
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_ragged_batch_device_api(cfg_name, corpus):
    # ragged streams through the device-pointer entry points (offsets/sizes arrays), incl. an empty stream
    import torch
    import divans_amd as da
    dev = torch.device("cuda", 0)
    lens = [0, 1, 777, 65536, 40000, 3, 32768, 32770, 12345]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = corpus[1000:1000 + sum(lens)].copy()
    d_in = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to(dev)
    codec = da.LiteralCodec(da.config_simple() if cfg_name == "simple" else da.config_context_mixing(), max(lens))
    outs = codec.alloc_encode_outputs(len(lens))
    d_off = torch.tensor(starts, dtype=torch.int64, device=dev); d_sz = torch.tensor(lens, dtype=torch.int32, device=dev)
    codec.encode_batch(d_in, len(lens), max(lens), outs, in_offsets=d_off, in_sizes=d_sz)
    d_back = torch.zeros(sum(lens) + 64, dtype=torch.uint8, device=dev)
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], len(lens), max(lens), d_back, out_offsets=d_off, out_sizes=d_sz)
    torch.cuda.synchronize()
    assert (d_back.cpu().numpy()[:sum(lens)] == flat).all()
    offs = outs["offsets"].cpu().numpy(); szs = outs["sizes"].cpu().numpy(); blob = outs["out"].cpu().numpy()
    ocfg = _oracle_cfg(cfg_name)
    for i, (s0, ln) in enumerate(zip(starts, lens)):
        ref = po.lit_encode(ocfg, flat[s0:s0 + ln])
        assert szs[i] == ref.size and (blob[offs[i]:offs[i] + szs[i]] == ref).all(), i
    codec.close()


def _random_config(rng, da, mixing, modes, mm_values, speeds):
    """A configuration a brotli-driven stream could carry: arbitrary context map, per-context mixing values
    (stride 1/2/3/4/8, half-byte, no-prior, context-only), lossy palette speeds, any prediction mode."""
    import ctypes
    g = da.LitConfig()
    o = po.LitConfig()
    cmap = rng.integers(0, 48, size=256 * 64, dtype=np.uint8)
    mix = rng.choice(np.array(mm_values, dtype=np.uint8), size=8192)
    for cfg in (g, o):
        ctypes.memmove(cfg.literal_context_map, cmap.ctypes.data, cmap.size)
        ctypes.memmove(cfg.mixing_mask, mix.ctypes.data, mix.size)
        cfg.prediction_mode = int(modes)
        cfg.btype = int(rng.integers(0, 4))
        cfg.context_mixing = mixing
        for i in range(4):
            cfg.literal_adaptation[i].inc = speeds[i][0]
            cfg.literal_adaptation[i].lim = speeds[i][1]
        rng_state = rng.bit_generator.state   # keep g and o identical: same btype
        o.btype = g.btype
    return g, o


@pytest.mark.parametrize("mixing", [0, 2])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_generic_mixing_mask_and_context_maps(mixing, mode, corpus, random_then_unicode):
    # the MM = -1 / table-driven path: every branch of code_nibble's index math (mm_opts 0,1,2,3,4,5,6,7,8 and beyond)
    import divans_amd as da
    rng = np.random.default_rng(100 * mixing + mode)
    speeds = [(16, 8192), (64, 16384), (2, 1024), (128, 16384)]
    value_sets = [[0, 1, 2, 3, 4, 5, 6, 7, 8], [4, 5, 8, 12], [0, 3], [1, 2, 4]]
    for vs in value_sets:
        g, o = _random_config(rng, da, mixing, mode, vs, speeds)
        L = 6000
        blocks = np.stack([corpus[3000:3000 + L], random_then_unicode[100000:100000 + L], random_then_unicode[200000:200000 + L],
                           np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), L)])
        codec = da.LiteralCodec(g, L)
        packed, offs, sizes = codec.encode_host(blocks, L)
        for i in range(blocks.shape[0]):
            ref = po.lit_encode(o, blocks[i])
            got = packed[int(offs[i]):int(offs[i]) + int(sizes[i])]
            assert got.size == ref.size and (got == ref).all(), (vs, i)
        assert (codec.decode_host(packed, offs, sizes, L) == blocks).all()
        if da.experimental_decoders():
            try:
                codec.set_decoder(4, (8, 4, 0, 2), (3, 5, 5, 1))      # one lane per stream (lit_decode_t.hip), table-driven instances
            except da.DivansGpuError as e:
                assert "32767 rows" in str(e)                         # (its 15-bit cache tags; the value sets with three planes exceed them)
            else:
                assert (codec.decode_host(packed, offs, sizes, L) == blocks).all(), "generation 4"
        codec.close()


def test_uniform_mm0_specialisation(corpus):
    # mixing value 0 everywhere (reference TestAdapt, benchmark.rs:182-193): the MM = 0 kernel instances
    import ctypes
    import divans_amd as da
    g = da.config_context_mixing(); o = po.config_context_mixing()
    for cfg in (g, o):
        ctypes.memset(cfg.mixing_mask, 0, 8192)
        cfg.context_mixing = 0
        cfg.prediction_mode = 0
    blocks = workload.make_blocks(corpus, 9, 6, block_len=9000)
    codec = da.LiteralCodec(g, 9000)
    packed, offs, sizes = codec.encode_host(blocks, 9000)
    for i in range(6):
        assert (packed[int(offs[i]):int(offs[i]) + int(sizes[i])] == po.lit_encode(o, blocks[i])).all()
    assert (codec.decode_host(packed, offs, sizes, 9000) == blocks).all()
    if da.experimental_decoders():
        codec.set_decoder(4)
        assert (codec.decode_host(packed, offs, sizes, 9000) == blocks).all(), "generation 4"
    codec.close()


@pytest.mark.parametrize("btype", [0, 1])
@pytest.mark.parametrize("context_mixing", [0, 2])
def test_brotli_derived_prediction_mode(btype, context_mixing, corpus):
    # configuration a real brotli -q11 run produced (reference testdata/alice29-priors.ir): clustered context map,
    # per-context mixing values 0/1/2/3, both literal block types; with and without dynamic mixing
    import divans_amd as da
    g = workload.brotli_derived_config(da.LitConfig(), btype, context_mixing)
    o = workload.brotli_derived_config(po.LitConfig(), btype, context_mixing)
    blocks = np.stack([corpus[k * 30011:k * 30011 + 30000] for k in range(8)])
    codec = da.LiteralCodec(g, 30000)
    packed, offs, sizes = codec.encode_host(blocks, 30000)
    for i in range(blocks.shape[0]):
        assert (packed[int(offs[i]):int(offs[i]) + int(sizes[i])] == po.lit_encode(o, blocks[i])).all(), i
    assert (codec.decode_host(packed, offs, sizes, 30000) == blocks).all()
    codec.close()


def _encode_ragged(da, codec, blocks, lens):
    """device entry point with per-stream sizes (no status check of its own) -> (coded bytes, offsets, sizes) on the host"""
    import torch
    dev = torch.device("cuda")
    n, L = blocks.shape
    d_in = torch.from_numpy(np.concatenate([blocks.reshape(-1), np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * L
    d_sz = torch.from_numpy(lens.astype(np.int32)).to(dev)
    outs = codec.alloc_encode_outputs(n, L)
    codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz)
    torch.cuda.synchronize()
    return outs["out"].cpu().numpy(), outs["offsets"].cpu().numpy(), outs["sizes"].cpu().numpy()


def _decode_ragged(da, codec, packed, offs, sizes, lens, L):
    import torch
    dev = torch.device("cuda")
    n = len(lens)
    d_coded = torch.from_numpy(np.concatenate([np.asarray(packed, np.uint8), np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(np.asarray(offs).astype(np.int64)).to(dev); d_sz = torch.from_numpy(np.asarray(sizes).astype(np.int32)).to(dev)
    out = torch.zeros(n * L + 64, dtype=torch.uint8, device=dev)
    o_off = torch.arange(n, dtype=torch.int64, device=dev) * L
    o_sz = torch.from_numpy(np.asarray(lens).astype(np.int32)).to(dev)
    codec.decode_batch(d_coded, d_off, d_sz, n, L, out, out_offsets=o_off, out_sizes=o_sz)
    torch.cuda.synchronize()
    return out[:n * L].cpu().numpy().reshape(n, L)


def test_a_negative_increment_is_rejected():
    import divans_amd as da
    g = da.config_simple()
    g.literal_adaptation[0].inc = -1; g.literal_adaptation[0].lim = 100
    with pytest.raises(da.DivansGpuError):
        da.LiteralCodec(g, 1024)


@pytest.mark.parametrize("mixing", [0, 2])
def test_speeds_under_which_the_references_row_totals_wrap(mixing, corpus):
    """every speed the wire format can carry is taken.  Where FrequentistCDF16::blend's i16 total can wrap
    (divans_gpu_speed_supported is false) the reference codes correctly until a wrapped row is coded with again -- from there its own
    encoder writes what its decoder cannot read (the oracle, which wraps like it, reports the invalid (start, freq)).  The GPU coder agrees
    on both sides of that line: streams the oracle can code come out bit for bit and decode back with a clean status; streams it cannot
    raise BAD_MODEL when encoded, and bytes that are no valid stream under such a speed raise BAD_STREAM when decoded."""
    import divans_amd as da
    rng = np.random.default_rng(5 + mixing)
    for speed in [(0x3000, 0x1000), (0x4000, 0x4000), (0x7800, 0x7800), (8180, 64), (0x2800, 0x6000)]:
        assert not da.speed_supported(*speed) and da.speed_accepted(*speed)
        speeds = [speed, (0x10, 0x2000), speed, speed] if mixing else [speed] * 4
        g, o = _random_config(rng, da, mixing, 2, [4], speeds)
        L = 2048
        cands = [corpus[7000 + 97 * i:7000 + 97 * i + n].copy() for i, n in enumerate([1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 40, 64, 200, 700, 2048])]
        cands += [np.arange(n, dtype=np.uint8) * 37 % 251 for n in (5, 9, 30)] + [np.full(n, 0x61, np.uint8) for n in (1, 2, 3, 4, 5, 7)]
        codec = da.LiteralCodec(g, L)
        n_clean = n_flagged = 0
        for i, c in enumerate(cands):
            try:
                ref = po.lit_encode(o, c)
                oracle_ok = bool((po.lit_decode(o, ref, c.size) == c).all())     # with mixing a wrapped row can yield pairs that LOOK valid and still not decode
            except RuntimeError:
                ref, oracle_ok = None, False
            block = np.zeros((1, L), np.uint8); block[0, :c.size] = c
            lens = np.array([c.size], np.uint32)
            packed, offs, sizes = _encode_ragged(da, codec, block, lens)
            st = codec.status()
            assert st in (0, 1), (speed, i, st)
            if st == 0:             # no wrapped row was coded with: the reference's arithmetic, bit for bit
                n_clean += 1
                assert oracle_ok, (speed, i, c.size, "the GPU coder saw no wrapped row in use, the oracle cannot code the stream")
                got = packed[int(offs[0]):int(offs[0]) + int(sizes[0])]
                assert got.size == ref.size and (got == ref).all(), (speed, i, c.size)
                back = _decode_ragged(da, codec, packed, offs, sizes, lens, L)
                assert codec.status() == 0 and (back[0, :c.size] == c).all(), (speed, i)
            else:                   # BAD_MODEL: a row whose i16 total had wrapped was coded with again
                n_flagged += 1
                if not mixing:
                    assert ref is None, (speed, i, c.size, "flagged, but the oracle codes the stream")
        assert n_clean >= 8 and n_flagged >= 3, (speed, n_clean, n_flagged)      # both sides of the line are exercised
        # ... and what no encoder wrote: a long stream coded under the default speed, read under this one
        plain, po_plain = _random_config(rng, da, mixing, 2, [4], [(0x10, 0x2000)] * 4)
        long_in = corpus[20000:20000 + L].copy()[None, :]
        pc = da.LiteralCodec(plain, L)
        pp, poff, psz = pc.encode_host(long_in, L)
        pc.close()
        _decode_ragged(da, codec, pp, poff, psz, np.array([L], np.uint32), L)
        assert codec.status() & 2, (speed, "BAD_STREAM expected")
        codec.close()


@pytest.mark.parametrize("mixing", [0, 2])
@pytest.mark.parametrize("encode_path", [1, 2])
def test_speeds_whose_total_stays_above_lim(mixing, encode_path, corpus, random_then_unicode):
    # large increments: the row total settles near 4 * inc, far above lim, and every update renormalises -- accepted as long as it
    # stays inside i16 (the oracle's arithmetic wraps like the reference's, so agreement here means nothing wrapped)
    import divans_amd as da
    rng = np.random.default_rng(77 + mixing)
    for speeds in ([(8000, 64), (8100, 16384), (4096, 16384), (8160, 1)], [(8100, 16384), (8000, 64), (8160, 1), (4096, 16384)]):
        for sp in speeds:
            assert da.speed_supported(*sp)
        g, o = _random_config(rng, da, mixing, 0, [4], speeds)     # LSB6: the bucketed passes apply
        L = 9000
        blocks = np.stack([corpus[3000:3000 + L], random_then_unicode[100000:100000 + L], np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), L),
                           rng.integers(0, 256, L, dtype=np.uint8)])
        codec = da.LiteralCodec(g, L)
        codec.set_encode_path(encode_path)
        packed, offs, sizes = codec.encode_host(blocks, L)
        for i in range(blocks.shape[0]):
            ref = po.lit_encode(o, blocks[i])
            got = packed[int(offs[i]):int(offs[i]) + int(sizes[i])]
            assert got.size == ref.size and (got == ref).all(), (speeds, i)
        for gen in (g for g in (1, 3, 4) if g in da.decoder_generations()):
            codec.set_decoder(gen)
            assert (codec.decode_host(packed, offs, sizes, L) == blocks).all(), gen
        codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("encode_path", [1, 2])
@pytest.mark.parametrize("length", [1, 2, 63, 64, 65, 8191, 8192, 8193, 16385, 40000, 65535, 65536])
def test_encode_paths_bit_exact(cfg_name, encode_path, length, corpus, shuffle384):
    # streaming model kernel (1) and bucketed model pass (2, what "automatic" picks for these configurations)
    # must both reproduce the oracle's bytes; lengths straddle the 8 KiB sort pieces and the 32 KiB ANS chunks
    blocks = workload.make_blocks(corpus, 40, 70, block_len=length, perturb_per_block=length // 100)
    if length >= 64:
        blocks[3] = np.resize(shuffle384, length)
        blocks[4] = np.random.default_rng(length).integers(0, 256, length, dtype=np.uint8)
        blocks[5] = 0
        blocks[6] = np.resize(np.frombuffer(b"ab", dtype=np.uint8), length)   # two buckets own every position
    _compare(cfg_name, blocks, encode_path=encode_path)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_bucketed_two_model_pass_over_prediction_modes(mode, corpus, random_then_unicode):
    # the two-model bucketed pass keys its high-nibble rows by (prev, class of prev_prev -> context): every prediction
    # mode (LSB6 / MSB6 / UTF8 / SIGN luts: 1, 1, 4 and 8 classes), random context maps, palette speeds, any block type
    import divans_amd as da
    rng = np.random.default_rng(700 + mode)
    speeds = [(16, 8192), (64, 16384), (2, 1024), (128, 16384)]
    L = 20000
    blocks = np.stack([corpus[3000:3000 + L], random_then_unicode[100000:100000 + L], random_then_unicode[200000:200000 + L],
                       np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), L), rng.integers(0, 256, L, dtype=np.uint8)])
    for _ in range(2):
        g, o = _random_config(rng, da, 2, mode, [4], speeds)
        codec = da.LiteralCodec(g, L)
        codec.set_encode_path(2)
        packed, offs, sizes = codec.encode_host(blocks, L)
        for i in range(blocks.shape[0]):
            ref = po.lit_encode(o, blocks[i])
            got = packed[int(offs[i]):int(offs[i]) + int(sizes[i])]
            assert got.size == ref.size and (got == ref).all(), (mode, i)
        codec.set_encode_path(1)
        packed1, offs1, sizes1 = codec.encode_host(blocks, L)
        assert (sizes == sizes1).all() and (packed == packed1).all()
        assert (codec.decode_host(packed, offs, sizes, L) == blocks).all()
        codec.close()


def test_bucketed_two_model_pass_random_stress(corpus, random_then_unicode):
    # random palette speeds, context maps, block types, prediction modes and ragged lengths (0 .. 64 KiB) through the
    # device-pointer entry point: bucketed pass == streaming kernel on every stream, oracle on a sample
    import torch
    import divans_amd as da
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4242)
    palette = [(1, 16384), (1, 1024), (2, 1024), (4, 2048), (8, 4096), (16, 8192), (32, 4096), (64, 16384), (128, 16384), (256, 16384), (1024, 16384)]
    src = np.tile(np.concatenate([corpus, random_then_unicode]), 8)
    for trial in range(6):
        speeds = [palette[int(k)] for k in rng.integers(0, len(palette), 4)]
        g, o = _random_config(rng, da, 2, int(rng.integers(0, 4)), [4], speeds)
        n = 48
        lens = rng.integers(0, 65537, n).astype(np.int32)
        lens[:5] = [0, 1, 2, 65536, 8193]
        starts = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
        base = int(rng.integers(0, src.size - int(lens.sum()) - 64))
        flat = src[base:base + int(lens.sum())].copy()
        if trial % 2:
            flat[::97] ^= 0x5a
        d_in = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to(dev)
        d_off = torch.from_numpy(starts).to(dev); d_sz = torch.from_numpy(lens).to(dev)
        codec = da.LiteralCodec(g, 65536)
        got = []
        for path in (2, 1):
            codec.set_encode_path(path)
            outs = codec.alloc_encode_outputs(n)
            codec.encode_batch(d_in, n, 65536, outs, in_offsets=d_off, in_sizes=d_sz)
            torch.cuda.synchronize()
            assert codec.status() == 0
            got.append((outs["offsets"].cpu().numpy(), outs["sizes"].cpu().numpy(), outs["out"].cpu().numpy()))
        (o2, s2, b2), (o1, s1, b1) = got
        assert (s2 == s1).all(), trial
        for i in range(n):
            assert (b2[o2[i]:o2[i] + s2[i]] == b1[o1[i]:o1[i] + s1[i]]).all(), (trial, i)
        for i in (0, 1, 2, 4, 7, n - 1):
            ref = po.lit_encode(o, flat[starts[i]:starts[i] + lens[i]])
            assert s2[i] == ref.size and (b2[o2[i]:o2[i] + s2[i]] == ref).all(), (trial, i)
        codec.close()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_bucketed_pass_when_the_context_follows_from_prev(mode, corpus, random_then_unicode):
    # no mixing, mixing value 4, arbitrary context map: LSB6 / MSB6 contexts are a function of the previous byte, so the
    # order-1 bucketed pass applies (the literal-only compressor's configuration); UTF8 / SIGN contexts need prev_prev
    import divans_amd as da
    rng = np.random.default_rng(900 + mode)
    speeds = [(16, 8192), (64, 16384), (2, 1024), (128, 16384)]
    L = 20000
    blocks = np.stack([corpus[3000:3000 + L], random_then_unicode[100000:100000 + L], np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), L),
                       rng.integers(0, 256, L, dtype=np.uint8)])
    g, o = _random_config(rng, da, 0, mode, [4], speeds)
    codec = da.LiteralCodec(g, L)
    if mode >= 2:
        with pytest.raises(da.DivansGpuError):
            codec.set_encode_path(2)
        codec.close()
        return
    codec.set_encode_path(2)
    packed, offs, sizes = codec.encode_host(blocks, L)
    for i in range(blocks.shape[0]):
        ref = po.lit_encode(o, blocks[i])
        got = packed[int(offs[i]):int(offs[i]) + int(sizes[i])]
        assert got.size == ref.size and (got == ref).all(), (mode, i)
    codec.set_encode_path(1)
    packed1, offs1, sizes1 = codec.encode_host(blocks, L)
    assert (sizes == sizes1).all() and (packed == packed1).all()
    assert (codec.decode_host(packed, offs, sizes, L) == blocks).all()
    codec.close()


@pytest.mark.parametrize("cfg_name,encode_path", [("simple", 1), ("simple", 2), ("mixing", 1), ("mixing", 2)])
def test_model_pass_matches_oracle_trace(cfg_name, encode_path, corpus, shuffle384):
    # the (start, freq) pair of every nibble, straight out of the model pass on a fresh codec (nothing stale to hide
    # behind), against the oracle's put_start_freq trace
    import torch
    L = 20011
    blocks = workload.make_blocks(corpus, 77, 40, block_len=L, perturb_per_block=200)
    blocks[1] = np.resize(shuffle384, L)
    blocks[2] = 0
    da, codec = _codec(cfg_name, L)
    codec.set_encode_path(encode_path)
    pairs = codec.model_batch(torch.from_numpy(blocks).to("cuda:0"), blocks.shape[0], L)
    torch.cuda.synchronize()
    pairs = pairs.cpu().numpy().view(np.uint32)[:, :2 * L]
    ocfg = _oracle_cfg(cfg_name)
    for i in range(blocks.shape[0]):
        _, tr = po.lit_encode(ocfg, blocks[i], trace=True)
        want = tr[:, 1].astype(np.uint32) | (tr[:, 2].astype(np.uint32) << 16)
        assert (pairs[i] == want).all(), (i, int(np.argmax(pairs[i] != want)))
    codec.close()


def test_bucketed_encoder_many_streams(corpus):
    # enough buckets that every wave of the chain kernel recycles its lanes many times and reserves several task windows
    blocks = workload.make_blocks(corpus, 9, 3000, block_len=9000)
    da, codec = _codec("simple", 9000)
    codec.set_encode_path(2)
    packed, offs, sizes = codec.encode_host(blocks, 9000)
    codec.set_encode_path(1)
    packed1, offs1, sizes1 = codec.encode_host(blocks, 9000)
    assert (sizes == sizes1).all() and (offs == offs1).all() and (packed == packed1).all()
    ocfg = _oracle_cfg("simple")
    for i in range(0, 3000, 101):
        ref = po.lit_encode(ocfg, blocks[i])
        assert (packed[int(offs[i]):int(offs[i]) + int(sizes[i])] == ref).all(), i
    assert (codec.decode_host(packed, offs, sizes, 9000) == blocks).all()
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_bucketed_encoder_many_streams_both_paths_agree(cfg_name, corpus):
    # ragged lengths, more streams than one launch sequence of the two-model pass takes at once (set to 8192 here)
    import torch
    import divans_amd as da
    dev = torch.device("cuda", 0)
    n, L = 20000, 600
    blocks = workload.make_blocks(corpus, 11, n, block_len=L)
    lens = (np.arange(n) * 37 % (L + 1)).astype(np.int32)
    lens[:4] = [0, 1, L, L - 1]
    starts = np.arange(n, dtype=np.int64) * L
    d_in = torch.from_numpy(np.concatenate([blocks.reshape(-1), np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(starts).to(dev); d_sz = torch.from_numpy(lens).to(dev)
    codec = da.LiteralCodec(da.config_simple() if cfg_name == "simple" else da.config_context_mixing(), L)
    codec.set_bucket_batch(8192)
    got = []
    for path in (2, 1):
        codec.set_encode_path(path)
        outs = codec.alloc_encode_outputs(n)
        codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz)
        torch.cuda.synchronize()
        got.append((outs["offsets"].cpu().numpy(), outs["sizes"].cpu().numpy(), outs["out"].cpu().numpy()))
    (o2, s2, b2), (o1, s1, b1) = got
    assert (s2 == s1).all()
    for i in range(n):
        assert (b2[o2[i]:o2[i] + s2[i]] == b1[o1[i]:o1[i] + s1[i]]).all(), i
    ocfg = _oracle_cfg(cfg_name)
    for i in list(range(0, n, 499)) + [8191, 8192, 16383, 16384, n - 1]:
        ref = po.lit_encode(ocfg, blocks[i][:lens[i]])
        assert s2[i] == ref.size and (b2[o2[i]:o2[i] + s2[i]] == ref).all(), i
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_encoder_sub_batches_of_two_chunk_streams(cfg_name, corpus):
    # 64 KiB slots: the chunk-parallel rANS pass keeps its scratch in the model pass's dead work arrays and reads the pairs
    # where the in-place unsort left them; the two-model pass runs 70 streams as sub-batches of 24, 24, 22 and the rANS pass
    # of each sub-batch is enqueued before the next one reuses the arrays.  Ragged: one- and two-chunk streams mixed.
    import torch
    import divans_amd as da
    dev = torch.device("cuda", 0)
    n, L = 70, 65536
    blocks = workload.make_blocks(corpus, 5, n, block_len=L)
    lens = (20000 + np.arange(n) * 7919 % (L - 20000 + 1)).astype(np.int32)
    lens[:6] = [L, 32768, 32769, 1, 0, L - 1]
    starts = np.arange(n, dtype=np.int64) * L
    d_in = torch.from_numpy(np.concatenate([blocks.reshape(-1), np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(starts).to(dev); d_sz = torch.from_numpy(lens).to(dev)
    codec = da.LiteralCodec(da.config_simple() if cfg_name == "simple" else da.config_context_mixing(), L)
    codec.set_bucket_batch(24)
    got = []
    for path in (2, 1):
        codec.set_encode_path(path)
        outs = codec.alloc_encode_outputs(n)
        chunks = torch.full((n, 2), -1, dtype=torch.int32, device=dev)
        codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz, chunk_bytes=chunks)
        pairs = codec.model_batch(d_in, n, L, in_offsets=d_off, in_sizes=d_sz)
        torch.cuda.synchronize()
        assert codec.status() == 0
        got.append((outs["offsets"].cpu().numpy(), outs["sizes"].cpu().numpy(), outs["out"].cpu().numpy(), chunks.cpu().numpy(), pairs.cpu().numpy()))
    (o2, s2, b2, c2, p2), (o1, s1, b1, c1, p1) = got
    assert (s2 == s1).all() and (o2 == o1).all()
    for i in range(n):
        assert (b2[o2[i]:o2[i] + s2[i]] == b1[o1[i]:o1[i] + s1[i]]).all(), i
        nch = (2 * int(lens[i]) + 65535) // 65536
        assert (c2[i, :nch] == c1[i, :nch]).all() and int(c2[i, :nch].sum()) == int(s2[i]), i
        assert (p2[i, :2 * lens[i]] == p1[i, :2 * lens[i]]).all(), i
    ocfg = _oracle_cfg(cfg_name)
    for i in (0, 1, 2, 3, 5, 23, 24, 47, 48, 69):
        ref = po.lit_encode(ocfg, blocks[i][:lens[i]])
        assert s2[i] == ref.size and (b2[o2[i]:o2[i] + s2[i]] == ref).all(), i
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_stream_coded_piece_by_piece(cfg_name, corpus, random_then_unicode):
    # divans_gpu_lit_stream_*: the model pass of a piece resumes the previous one (tables, Weights, history), chunks are coded as
    # they complete.  Pieces of 1 byte, around the 32 768-byte chunk, several chunks at once; the bytes must be the oracle's
    # stream of the concatenation, the chunk sizes must add up.
    data = np.concatenate([corpus[:70000], random_then_unicode[200000:260000]])
    cuts = [0, 1, 2, 17, 4096, 32767, 32768, 32769, 65536, 65537, 100000, data.size]
    da, codec = _codec(cfg_name, 65536)
    coded, sizes = codec.stream_encode_pieces([data[a:b] for a, b in zip(cuts[:-1], cuts[1:])])
    ref = po.lit_encode(_oracle_cfg(cfg_name), data)
    assert coded.size == ref.size and (coded == ref).all()
    assert sum(sizes) == coded.size and len(sizes) == (2 * data.size + 65535) // 65536
    # and again on the same codec: a new stream starts from fresh tables
    coded2, _ = codec.stream_encode_pieces([data[:5000], data[5000:5001], data[5001:40000]])
    ref2 = po.lit_encode(_oracle_cfg(cfg_name), data[:40000])
    assert coded2.size == ref2.size and (coded2 == ref2).all()
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_stream_decoded_chunk_by_chunk(cfg_name, corpus, random_then_unicode):
    # divans_gpu_lit_stream_decode: whole chunks per call, the model resumed (tables, Weights, history); each call is shown only
    # the bound of the chunks it asks for, never the whole stream
    data = np.concatenate([corpus[:70000], random_then_unicode[200000:260001]])
    ref = po.lit_encode(_oracle_cfg(cfg_name), data)
    da, codec = _codec(cfg_name, 65536)
    for per_call in (1, 2):
        back, used = codec.stream_decode_chunks(ref, data.size, chunks_per_call=per_call)
        assert used == ref.size and (back == data).all(), per_call
    with pytest.raises(da.DivansGpuError):                     # a call that is shown too little of its chunk says so
        codec.stream_decode_chunks(ref, data.size, slack=1000)
    bad = ref.copy(); bad[40000] ^= 4
    with pytest.raises(da.DivansGpuError):
        codec.stream_decode_chunks(bad, data.size)
    codec.close()


def test_bucketed_encoder_only_where_it_applies():
    import ctypes
    import divans_amd as da
    g = da.config_context_mixing()
    ctypes.memset(g.mixing_mask, 5, 8192)                   # stride 2: rows are no longer a function of (prev, ctx, high nibble)
    codec = da.LiteralCodec(g, 4096)
    with pytest.raises(da.DivansGpuError):
        codec.set_encode_path(2)
    codec.close()
    for cfg in (da.config_simple(), da.config_context_mixing()):
        codec = da.LiteralCodec(cfg, 100000)                # streams longer than 8 pieces: streaming kernels only
        with pytest.raises(da.DivansGpuError):
            codec.set_encode_path(2)
        codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("slice_streams", [0, 1, 7, 64])
def test_pipelined_host_wrappers_match_the_serial_ones(cfg_name, slice_streams, corpus):
    # copy-in / code / copy-out pipeline over slices (pageable and page-locked buffers): same bytes, offsets, sizes
    import divans_amd as da
    L, n = 3000, 50
    blocks = workload.make_blocks(corpus, 21, n, block_len=L)
    da_, codec = _codec(cfg_name, L)
    packed, offs, sizes = codec.encode_host(blocks, L)
    pin_in = da.PinnedBuffer(n * L); pin_in.array[:] = blocks.reshape(-1)
    pin_out = da.PinnedBuffer(da.encode_bound(L) * n + 64)
    for data, out in ((blocks, None), (pin_in.array, pin_out.array)):
        p2, o2, s2 = codec.encode_host_pipelined(data, L, slice_streams=slice_streams, out=out)
        assert (s2 == sizes).all() and (o2 == offs).all() and p2.size == packed.size and (p2 == packed).all()
    back = codec.decode_host_pipelined(packed, offs, sizes, L, slice_streams=slice_streams)
    assert (back == blocks).all()
    pin_back = da.PinnedBuffer(n * L)
    back = codec.decode_host_pipelined(pin_out.array[:packed.size], offs, sizes, L, slice_streams=slice_streams, out=pin_back.array)
    assert (back == blocks).all()
    # out-of-order offsets fall back to the serial wrapper; damaged streams are still reported
    perm = np.arange(n)[::-1].copy()
    assert (codec.decode_host_pipelined(packed, offs[perm], sizes[perm], L, slice_streams=slice_streams) == blocks[perm]).all()
    bad = packed.copy(); bad[int(offs[n // 2]) + 20] ^= 0x40
    with pytest.raises(da.DivansGpuError):
        codec.decode_host_pipelined(bad, offs, sizes, L, slice_streams=slice_streams)
    # an output buffer that cannot hold the packed streams is refused, and the codec stays usable
    with pytest.raises(da.DivansGpuError):
        codec.encode_host_pipelined(blocks, L, slice_streams=slice_streams, out=np.empty(1000, np.uint8))
    p3, o3, s3 = codec.encode_host_pipelined(blocks, L, slice_streams=slice_streams)
    assert (s3 == sizes).all() and (p3 == packed).all()
    for b in (pin_in, pin_out, pin_back):
        b.close()
    codec.close()


# ---- second-generation decoder (lit_decode2.hip): direct-mapped caches of every size, with and without low-row caches ----
_DM_GEOMETRIES = [
    ((32, 0, 0, 0), (31, 5, 5, 5)),      # the non-mixing default: high stride rows, indexed by the previous byte
    ((16, 16, 0, 0), (5, 5, 5, 5)),      # the mixing default
    ((0, 0, 0, 0), (5, 5, 5, 5)),        # every row in HBM / L2
    ((4, 4, 4, 4), (0, 1, 2, 3)),        # tiny caches: evictions on almost every access, four different hashes
    ((64, 8, 32, 64), (8, 31, 4, 5)),
    ((128, 0, 64, 0), (5, 5, 7, 5)),
    ((0, 32, 0, 16), (5, 5, 5, 5)),
]


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("generation", [2, 3])          # direct-mapped / 2-way caches
@pytest.mark.parametrize("geom", range(len(_DM_GEOMETRIES)))
def test_decoder2_cache_geometries(cfg_name, generation, geom, corpus, shuffle384, random_then_unicode):
    rows, shifts = _DM_GEOMETRIES[geom]
    L = 20000
    blocks = np.stack([corpus[5000:5000 + L], corpus[90000:90000 + L], np.resize(shuffle384, L), random_then_unicode[99000:99000 + L],
                       random_then_unicode[250000:250000 + L], np.resize(np.frombuffer(b"abracadabra ", dtype=np.uint8), L)])
    import torch
    lens = [L, 17, L - 3, 4097, L, 9000]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = np.concatenate([blocks[i, :lens[i]] for i in range(6)])
    dev = torch.device("cuda", 0)
    da, codec = _codec(cfg_name, L)
    codec.set_decoder(generation, rows, shifts, blocks=2)  # two workgroups: the 6 streams share waves with idle rows of lanes
    d_in = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to(dev)
    outs = codec.alloc_encode_outputs(6)
    d_off = torch.tensor(starts, dtype=torch.int64, device=dev); d_sz = torch.tensor(lens, dtype=torch.int32, device=dev)
    codec.encode_batch(d_in, 6, L, outs, in_offsets=d_off, in_sizes=d_sz)
    d_back = torch.zeros(sum(lens) + 64, dtype=torch.uint8, device=dev)
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], 6, L, d_back, out_offsets=d_off, out_sizes=d_sz)
    torch.cuda.synchronize()
    assert (d_back.cpu().numpy()[:sum(lens)] == flat).all(), (cfg_name, rows)
    assert codec.status() == 0
    offs = outs["offsets"].cpu().numpy(); szs = outs["sizes"].cpu().numpy(); blob = outs["out"].cpu().numpy()
    ocfg = _oracle_cfg(cfg_name)
    for i, (s0, ln) in enumerate(zip(starts, lens)):
        ref = po.lit_encode(ocfg, flat[s0:s0 + ln])
        assert szs[i] == ref.size and (blob[offs[i]:offs[i] + szs[i]] == ref).all(), i
    codec.close()


# ---- generation 4 (lit_decode_t.hip): one lane per stream, direct-mapped caches laid out by lane ----
_T_GEOMETRIES = [
    ((0, 0, 0, 0), (5, 5, 5, 5)),        # no caches: one staging slot per table
    ((4, 4, 4, 4), (0, 1, 2, 3)),        # evictions on almost every access, four different hashes
    ((16, 16, 0, 0), (5, 5, 5, 5)),
    ((8, 2, 1, 16), (8, 31, 4, 5)),
    ((32, 0, 16, 0), (31, 5, 7, 5)),
]


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("geom", range(len(_T_GEOMETRIES)))
def test_lane_per_stream_decoder_geometries(cfg_name, geom, corpus, shuffle384, random_then_unicode):
    """ragged streams (a wave's lanes finish at different times, three rounds of a one-workgroup grid with live and idle lanes), lengths
    across the 32 KiB chunk boundary, unaligned output offsets"""
    import torch
    import divans_amd
    if not divans_amd.experimental_decoders():
        pytest.skip("generation 4 is an experiment build: DIVANS_WITH_EXPERIMENTAL_DECODERS=1 python divans_amd/build.py --force")
    rows, shifts = _T_GEOMETRIES[geom]
    L = 40000
    n = 150
    rng = np.random.default_rng(geom)
    lens = [int(x) for x in rng.integers(1, 3000, n)]
    lens[0] = L; lens[1] = 32768; lens[2] = 32769; lens[3] = 17; lens[70] = 33000; lens[149] = 1
    srcs = [corpus, random_then_unicode, np.resize(shuffle384, 200000), np.resize(np.frombuffer(b"abracadabra ", dtype=np.uint8), 200000)]
    parts = [srcs[i % 4][1000 * i:1000 * i + lens[i]] for i in range(n)]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = np.concatenate(parts)
    dev = torch.device("cuda", 0)
    da, codec = _codec(cfg_name, L)
    codec.set_decoder(4, rows, shifts, blocks=1)
    d_in = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to(dev)
    outs = codec.alloc_encode_outputs(n)
    d_off = torch.tensor(starts, dtype=torch.int64, device=dev); d_sz = torch.tensor(lens, dtype=torch.int32, device=dev)
    codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz)
    d_back = torch.zeros(sum(lens) + 64, dtype=torch.uint8, device=dev)
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, L, d_back, out_offsets=d_off, out_sizes=d_sz)
    torch.cuda.synchronize()
    assert "lit_decode_t_kernel" in codec.last_decode_kernel()
    back = d_back.cpu().numpy()
    assert (back[:sum(lens)] == flat).all(), (cfg_name, rows, int(np.argmax(back[:sum(lens)] != flat)))
    assert (back[sum(lens):] == 0).all()
    assert codec.status() == 0
    offs = outs["offsets"].cpu().numpy(); szs = outs["sizes"].cpu().numpy(); blob = outs["out"].cpu().numpy()
    ocfg = _oracle_cfg(cfg_name)
    for i in (0, 2, 3, 70, 149):
        ref = po.lit_encode(ocfg, flat[starts[i]:starts[i] + lens[i]])
        assert szs[i] == ref.size and (blob[offs[i]:offs[i] + szs[i]] == ref).all(), i
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_table_placement_tuning_changes_no_byte(cfg_name, corpus):
    """divans_gpu_codec_tune_tables: the decode that tunes (three copies of the tables, the launch repeated on each) and the ones after it
    return what an untuned codec returns; a batch too small to time tunes nothing; bad arguments are refused"""
    import torch
    da, codec = _codec(cfg_name, 2048)
    resident = codec.info().resident_groups
    n = resident // 2 + 16
    blocks = workload.make_blocks(corpus, 3, n, block_len=2048)
    d_in = torch.from_numpy(blocks).cuda()
    outs = codec.alloc_encode_outputs(n, 2048)
    codec.encode_batch(d_in, n, 2048, outs)
    plain = torch.zeros((n, 2048), dtype=torch.uint8, device="cuda")
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, plain)
    assert torch.equal(plain, d_in) and codec.status() == 0
    with pytest.raises(da.DivansGpuError):
        codec.tune_tables(17)
    codec.tune_tables(0)                            # the library's policy (the default): tables of 2 GiB and more are tuned, smaller ones are not
    pl = codec.table_placement()
    assert pl["policy_candidates"] == (12 if codec.info().table_bytes >= (2 << 30) else 1), pl
    codec.tune_tables(3)
    assert codec.table_placement()["policy_candidates"] == 3
    few = torch.zeros((40, 2048), dtype=torch.uint8, device="cuda")
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], 40, 2048, few)          # below half the grid: not the batch to time
    assert torch.equal(few, d_in[:40])
    for _ in range(3):                                                                      # the first of these tunes
        back = torch.zeros((n, 2048), dtype=torch.uint8, device="cuda")
        codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
        assert torch.equal(back, d_in) and codec.status() == 0
        assert codec.info().last_decode_ms > 0
    pl = codec.table_placement()                                                            # three placements were decoded on, the fastest stayed
    assert pl["tried"] == 3 and 0 < pl["best_ms"] <= pl["first_ms"] <= pl["worst_ms"], pl
    # a damaged stream is still reported by a tuning decode
    codec.tune_tables(2)
    coded = outs["out"].clone(); coded[int(outs["offsets"][5]) + 30] ^= 0x40
    codec.decode_batch(coded, outs["offsets"], outs["sizes"], n, 2048, back)
    assert codec.status() & 2
    codec.close()
    # chunk-mapped tables of a closed codec wait (mapped) for the next one; divans_gpu_trim gives their memory back.  The address ranges the
    # placements used are counted and stay below the cap
    torch.cuda.synchronize()
    tm = da.table_memory()
    assert tm["va_reserved_bytes"] <= tm["va_cap_bytes"] and tm["idle_ranges"] <= 2, tm
    before = torch.cuda.mem_get_info()[0]
    da.trim()
    assert torch.cuda.mem_get_info()[0] >= before + tm["idle_bytes"] // 2
    assert da.table_memory()["idle_ranges"] == 0 and da.table_memory()["va_reserved_bytes"] == tm["va_reserved_bytes"]     # memory back, addresses not
    da.trim()                                       # nothing left: a no-op
    # past the cap on reserved-and-never-returned address space, big tables are plain hipMalloc blocks (which hipFree does return)
    L = da.load_library()
    L.divans_gpu_set_table_va_cap(0)
    try:
        da3, codec3 = _codec(cfg_name, 2048)
        codec3.tune_tables(2)
        codec3.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
        assert torch.equal(back, d_in) and codec3.status() == 0
        assert codec3.table_placement()["kept"] == "one block" and da.table_memory()["va_reserved_bytes"] == tm["va_reserved_bytes"]
        codec3.close()
    finally:
        L.divans_gpu_set_table_va_cap(tm["va_cap_bytes"])
    da2, codec2 = _codec(cfg_name, 2048)            # and a codec made afterwards maps a new range
    codec2.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
    assert torch.equal(back, d_in) and codec2.status() == 0
    codec2.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
@pytest.mark.parametrize("split", [1, 2])
def test_rans_pass_with_one_or_two_lanes_per_chunk(cfg_name, split, corpus, shuffle384, random_then_unicode):
    """ans.rs:302-378: the chunk-parallel rANS pass with one lane per 65 536-symbol chunk and with two lanes per chunk, one per rANS state
    (the states alternate symbol by symbol; the two lanes merge their pushed words in step order) -- the oracle's bytes either way, for
    ragged batches around the chunk seam, streams of one and of two chunks, one symbol pair, and bytes that make a state push at every step"""
    import torch
    L = 65536
    lens = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 4095, 32766, 32767, 32768, 32769, 32770, 40001, 65535, 65536, 12345, 50000]
    srcs = [corpus, random_then_unicode, np.resize(shuffle384, 200000), np.random.default_rng(9).integers(0, 256, 200000, dtype=np.uint8),
            np.zeros(200000, np.uint8)]
    parts = [srcs[i % 5][777 * i:777 * i + n] for i, n in enumerate(lens)]
    n = len(lens)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = np.concatenate(parts)
    dev = torch.device("cuda", 0)
    da, codec = _codec(cfg_name, L)
    codec.set_rans_split(split)
    d_in = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).to(dev)
    outs = codec.alloc_encode_outputs(n)
    d_off = torch.tensor(starts, dtype=torch.int64, device=dev); d_sz = torch.tensor(lens, dtype=torch.int32, device=dev)
    codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz)
    torch.cuda.synchronize()
    assert codec.status() == 0
    offs = outs["offsets"].cpu().numpy(); szs = outs["sizes"].cpu().numpy(); blob = outs["out"].cpu().numpy()
    ocfg = _oracle_cfg(cfg_name)
    for i in range(n):
        ref = po.lit_encode(ocfg, parts[i])
        assert szs[i] == ref.size and (blob[offs[i]:offs[i] + szs[i]] == ref).all(), (split, i, lens[i], int(szs[i]), ref.size)
    d_back = torch.zeros(sum(lens) + 64, dtype=torch.uint8, device=dev)
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, L, d_back, out_offsets=d_off, out_sizes=d_sz)
    torch.cuda.synchronize()
    assert (d_back.cpu().numpy()[:sum(lens)] == flat).all() and codec.status() == 0
    with pytest.raises(da.DivansGpuError):
        codec.set_rans_split(3)
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_byte_order_of_the_tables_changes_no_byte(cfg_name, corpus, random_then_unicode):
    """divans_gpu_codec_set_byte_order: the order in which the stride-1 decoder lays out a table's rows (text-frequency rank or numeric) is
    private to a launch -- the same bytes come back either way, for text and for bytes the rank knows nothing about"""
    L = 5000
    blocks = np.stack([corpus[100:100 + L], random_then_unicode[:L], random_then_unicode[250000:250000 + L],
                       np.random.default_rng(5).integers(0, 256, L, dtype=np.uint8)] * 12)
    da, codec = _codec(cfg_name, L)
    codec.set_decoder(3, (8, 8, 0, 0), (5, 5, 5, 5), blocks=1)       # tiny caches: most rows go to the table in memory
    packed, offs, sizes = codec.encode_host(blocks, L)
    ocfg = _oracle_cfg(cfg_name)
    for i in range(4):
        assert (packed[int(offs[i]):int(offs[i]) + int(sizes[i])] == po.lit_encode(ocfg, blocks[i])).all()
    for order in (1, 0, 2, 1, 0):
        codec.set_byte_order(order)
        assert (codec.decode_host(packed, offs, sizes, L) == blocks).all(), order
        assert (codec.decode_host(packed, offs, sizes, L) == blocks).all(), order       # (order 0: the second call uses what the first learned)
    with pytest.raises(da.DivansGpuError):
        codec.set_byte_order(3)
    codec.close()


@pytest.mark.gpu
def test_byte_order_is_learned_from_the_data_the_codec_sees(corpus, random_then_unicode):
    """no constant decides the default.  Order 0 ranks the byte values by their frequency in the first batch the codec
    sees -- an encode call's input, or a decode-only codec's first output -- on the device; the rank is a permutation, puts the data's own
    frequent bytes first, and changes no decoded byte"""
    import torch
    L = 4096
    text = np.stack([corpus[i * 1000:i * 1000 + L] for i in range(96)])
    da, codec = _codec("simple", L)
    assert codec.byte_order() == {"mode": 0, "ready": False}
    packed, offs, sizes = codec.encode_host(text, L)            # the encoder's input teaches it
    bo = codec.byte_order(with_rank=True)
    assert bo["ready"] and sorted(bo["rank"]) == list(range(256))
    top = sorted(range(256), key=lambda b: bo["rank"][b])[:8]
    counts = np.bincount(np.concatenate([text[(j * 96) // 64, :2048] for j in range(64)]), minlength=256)       # the sample the kernel takes
    expect = sorted(range(256), key=lambda b: (-int(counts[b]), b))[:8]
    assert top == expect and ord(" ") in top and ord("e") in top, (top, expect)
    assert (codec.decode_host(packed, offs, sizes, L) == text).all()
    codec.close()
    # a codec that only ever decodes: numeric for its first call, learned from that call's output for the next
    da, dec = _codec("simple", L)
    assert (dec.decode_host(packed, offs, sizes, L) == text).all()
    bo2 = dec.byte_order(with_rank=True)
    assert bo2["ready"] and bo2["rank"] == bo["rank"]
    assert (dec.decode_host(packed, offs, sizes, L) == text).all()
    # other data, other order: asking for 0 again re-learns
    binary = np.stack([random_then_unicode[200000 + i * 500:200000 + i * 500 + L] for i in range(96)])
    p2, o2, s2 = dec.encode_host(binary, L)
    assert dec.byte_order(with_rank=True)["rank"] == bo["rank"]            # learned once ...
    dec.set_byte_order(0)
    assert dec.byte_order()["ready"] is False
    p2, o2, s2 = dec.encode_host(binary, L)                                    # ... until asked again
    assert dec.byte_order(with_rank=True)["rank"] != bo["rank"]
    assert (dec.decode_host(p2, o2, s2, L) == binary).all() and (dec.decode_host(packed, offs, sizes, L) == text).all()
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_placement_search_one_candidate_per_call_changes_no_byte(cfg_name, corpus):
    """the library's own placement policy never decodes a batch twice.  Each qualifying call runs on one placement and the
    next reads its time: k candidates are settled after k + 1 calls, every call returns the right bytes, calls of another shape are served
    in between without being compared, and the first call is not synchronised"""
    import torch
    da, codec = _codec(cfg_name, 2048)
    resident = codec.info().resident_groups
    n = resident // 2 + 16
    blocks = workload.make_blocks(corpus, 5, n, block_len=2048)
    d_in = torch.from_numpy(blocks).cuda()
    outs = codec.alloc_encode_outputs(n, 2048)
    codec.encode_batch(d_in, n, 2048, outs)
    codec.search_tables(4)
    pl = codec.table_placement()
    assert pl["policy_candidates"] == 4 and pl["tried"] == 0 and not pl["searching"]
    back = torch.zeros((n, 2048), dtype=torch.uint8, device="cuda")
    seen = []
    for call in range(7):
        back.zero_()
        codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
        if call == 2:      # another shape in between: decoded on the placement in use, not part of the comparison
            few = torch.zeros((48, 2048), dtype=torch.uint8, device="cuda")
            codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], 48, 2048, few)
            assert torch.equal(few, d_in[:48])
        assert torch.equal(back, d_in) and codec.status() == 0, call
        pl = codec.table_placement()
        seen.append((pl["tried"], pl["searching"]))
    # call i (0-based) has read the times of i placements; the fifth call reads the fourth time and ends the search
    assert seen == [(0, True), (1, True), (2, True), (3, True), (4, False), (4, False), (4, False)], seen
    assert 0 < pl["best_ms"] <= pl["first_ms"] <= pl["worst_ms"], pl
    # the search starts again when asked to, and a damaged stream is reported from a searching call like from any other
    codec.search_tables(2)
    coded = outs["out"].clone(); coded[int(outs["offsets"][7]) + 30] ^= 0x40
    codec.decode_batch(coded, outs["offsets"], outs["sizes"], n, 2048, back)
    assert codec.status() & 2
    for _ in range(3):
        codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
    assert torch.equal(back, d_in) and codec.status() == 0 and not codec.table_placement()["searching"]
    # a new policy in the middle of a search ends it on the best placement seen so far (the candidate under test goes back); the eager form
    # then runs from there, and nothing decodes wrongly on the way
    codec.search_tables(6)
    for _ in range(3):
        codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
    assert codec.table_placement()["searching"] and torch.equal(back, d_in)
    codec.tune_tables(3)
    assert not codec.table_placement()["searching"]
    codec.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, 2048, back)
    pl = codec.table_placement()
    assert pl["tried"] == 3 and not pl["searching"] and torch.equal(back, d_in) and codec.status() == 0
    codec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_row_replay_runs_the_decoders_row_traffic_and_leaves_the_codec_usable(cfg_name, corpus):
    """divans_gpu_codec_row_replay (bench.py's roofline.request_ceiling): a measurement aid -- it returns a time, refuses what it has no
    instance for, and a decode after it is unharmed"""
    import torch
    da, codec = _codec(cfg_name, 4096)
    n = 512
    blocks = workload.make_blocks(corpus, 9, n, block_len=4096)
    d_in = torch.from_numpy(blocks).cuda()
    packed, offs, sizes = codec.encode_host(blocks, 4096)
    ms = codec.row_replay(d_in, n, 4096)
    assert 0 < ms < 1000
    lens = torch.full((n,), 4096, dtype=torch.int32, device="cuda"); lens[::3] = 1000
    offsets = (torch.arange(n, dtype=torch.int64, device="cuda") * 4096)
    assert codec.row_replay(d_in, n, 4096, offsets=offsets, sizes=lens) > 0          # ragged batches too
    assert (codec.decode_host(packed, offs, sizes, 4096) == blocks).all() and codec.status() == 0
    codec.set_decoder(3, (8, 8, 8, 0) if cfg_name == "simple" else (8, 8, 0, 8), (5, 5, 5, 5))      # a cache organisation (a low-row cache) the replay has no instance for
    with pytest.raises(da.DivansGpuError):
        codec.row_replay(d_in, n, 4096)
    codec.close()


def test_default_build_does_not_offer_the_decoders_that_lost():
    """generation 4 (one lane per stream, 30-45 % slower) and a user-selected generation 1 exist only in a library built
    with DIVANS_WITH_EXPERIMENTAL_DECODERS=1; the default one answers EINVAL and keeps decoding with what it had"""
    da, codec = _codec("simple", 256)
    if da.experimental_decoders():
        pytest.skip("experiment build")
    for gen in (1, 4):
        with pytest.raises(da.DivansGpuError, match="experiment builds"):
            codec.set_decoder(gen)
    blocks = np.arange(8 * 256, dtype=np.uint32).astype(np.uint8).reshape(8, 256)
    packed, offs, sizes = codec.encode_host(blocks, 256)
    assert (codec.decode_host(packed, offs, sizes, 256) == blocks).all()
    assert "lit_decode2_kernel" in codec.last_decode_kernel()
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_both_decoder_generations_agree(cfg_name, corpus):
    blocks = workload.make_blocks(corpus, 40, 300, block_len=3000)
    da, codec = _codec(cfg_name, 3000)
    packed, offs, sizes = codec.encode_host(blocks, 3000)
    for gen in da.decoder_generations():
        codec.set_decoder(gen)
        assert (codec.decode_host(packed, offs, sizes, 3000) == blocks).all(), gen
    # a damaged stream fails the integrity check of either generation
    bad = packed.copy(); bad[int(offs[7]) + 40] ^= 0x10
    for gen in da.decoder_generations():
        codec.set_decoder(gen)
        with pytest.raises(da.DivansGpuError):
            codec.decode_host(bad, offs, sizes, 3000)
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_decoder_survives_thousands_of_damaged_streams(cfg_name, corpus):
    """The coded bytes are untrusted input to the decode kernels.  2048 streams in one launch, three in four damaged in a seeded way
    (bit flips, overwritten runs, random bytes, sizes cut to 0 / 4 / 12 / 16 / odd lengths, a neighbour's stream, sizes grown into the
    neighbour's slot), every fourth left intact: each generation of the decoder must return, flag EXACTLY the streams whose bytes no
    longer decode to a valid stream -- an intact stream next to a damaged one decodes exactly -- and never flag an intact one.  (A
    damaged stream that still passes the integrity rule would be caught by the container's CRC; here it only has to be harmless.)"""
    import ctypes
    import torch
    n, L = 2048, 2500
    blocks = workload.make_blocks(corpus, 11, n, block_len=L)
    da, codec = _codec(cfg_name, L)
    d_in = torch.from_numpy(blocks).cuda()
    outs = codec.alloc_encode_outputs(n, L)
    codec.encode_batch(d_in, n, L, outs)
    assert codec.status() == 0
    coded = outs["out"].cpu().numpy().copy()
    offs = outs["offsets"].cpu().numpy().copy(); sizes = outs["sizes"].cpu().numpy().copy()
    rng = np.random.default_rng(5)
    changed = np.zeros(n, bool); kinds = np.full(n, -1)
    for i in range(n):
        if i % 4 == 0:
            continue
        o, sz = int(offs[i]), int(sizes[i])
        kind = int(rng.integers(0, 8)); kinds[i] = kind
        before = coded[o:o + sz].copy(); before_sz = sz
        if kind == 0:
            coded[o + int(rng.integers(0, sz))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            a = int(rng.integers(0, sz)); m = min(sz - a, int(rng.integers(1, 64)))
            coded[o + a:o + a + m] = rng.integers(0, 256, m, dtype=np.uint8)
        elif kind == 2:
            coded[o:o + sz] = rng.integers(0, 256, sz, dtype=np.uint8)
        elif kind == 3:
            sizes[i] = [0, 4, 12, 16, 20, sz - 4, sz - 3, sz // 2][int(rng.integers(0, 8))]
        elif kind == 4:
            j = (i + 1) % n
            m = min(sz, int(sizes[j])); coded[o:o + m] = coded[int(offs[j]):int(offs[j]) + m]
        elif kind == 5:
            sizes[i] = min(sz + 4 * int(rng.integers(1, 40)), coded.size - o)   # reads on into the next stream's slot
        elif kind == 6:
            coded[o:o + 16] = 0xff                                           # both rANS states at their maximum
        else:
            coded[o:o + 16] = 0                                              # and at zero
        changed[i] = int(sizes[i]) != before_sz or not np.array_equal(coded[o:o + before_sz], before)
    d_coded = torch.from_numpy(coded).cuda(); d_offs = torch.from_numpy(offs).cuda(); d_sizes = torch.from_numpy(sizes).cuda()
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert codec._lib.divans_gpu_codec_set_stream_flags(codec._h, ctypes.c_void_p(flags.data_ptr())) == 0
    results = []
    for gen in da.decoder_generations():
        codec.set_decoder(gen)
        flags.zero_()
        d_back = torch.full((n, L), 0xEE, dtype=torch.uint8, device="cuda")
        codec.decode_batch(d_coded, d_offs, d_sizes, n, L, d_back)
        st = codec.status()
        f = flags.cpu().numpy().astype(bool); back = d_back.cpu().numpy()
        good = (back == blocks).all(axis=1)
        assert not f[~changed].any(), gen                      # nothing intact is flagged ...
        assert good[~changed].all(), gen                       # ... and everything intact decodes exactly, whatever its neighbours hold
        assert (st & 2) == 2 and f[changed].mean() > 0.9, (gen, f[changed].mean())
        # a damaged stream either is flagged or decoded to valid bytes after all
        # (kind 4 with a neighbour of the same coded size IS a valid stream: the neighbour's)
        good = good | ((kinds == 4) & (back == np.roll(blocks, -1, axis=0)).all(axis=1))
        bad = np.flatnonzero(~(f | good) & changed)
        assert bad.size == 0, (gen, bad[:10], kinds[bad[:10]], sizes[bad[:10]], outs['sizes'].cpu().numpy()[bad[:10]])
        results.append((f.copy(), back.copy()))
    for f, back in results[1:]:
        assert (f == results[0][0]).all()                          # the generations agree on what is acceptable
    assert codec._lib.divans_gpu_codec_set_stream_flags(codec._h, None) == 0
    codec.close()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_encode_packed_equals_encode_batch_plus_pack(cfg_name, corpus):
    """divans_gpu_lit_encode_packed (sub-batches through codec-owned slots, packed at running offsets) leaves exactly the bytes
    divans_gpu_lit_encode_batch + divans_gpu_pack_streams leave, for sub-batch sizes that do and do not divide the batch, ragged streams,
    and says so when the caller's buffer is too small"""
    import torch
    import divans_amd as da
    dev = torch.device("cuda")
    cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
    n, L = 77, 5000
    rng = np.random.default_rng(3)
    lens = rng.integers(0, L + 1, n).astype(np.int32); lens[0] = L; lens[5] = 0
    blocks = np.stack([np.resize(corpus[1000 * i:1000 * i + L], L) for i in range(n)])
    d_in = torch.from_numpy(np.concatenate([blocks.reshape(-1), np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.arange(n, dtype=torch.int64, device=dev) * L
    d_sz = torch.from_numpy(lens).to(dev)
    codec = da.LiteralCodec(cfg, L)
    outs = codec.alloc_encode_outputs(n, L)
    codec.encode_batch(d_in, n, L, outs, in_offsets=d_off, in_sizes=d_sz)
    ref_packed, ref_off, ref_total = codec.pack(outs, n)
    ref_total = int(ref_total.item())
    ref_sizes = outs["sizes"].clone()
    for sub in (0, 16, 77, 10, 1):
        packed = torch.zeros(ref_total + 256, dtype=torch.uint8, device=dev)
        poff = torch.zeros(n, dtype=torch.int64, device=dev); sizes = torch.zeros(n, dtype=torch.int32, device=dev); total = torch.zeros(1, dtype=torch.int64, device=dev)
        codec.encode_packed(d_in, n, L, packed, poff, sizes, total, in_offsets=d_off, in_sizes=d_sz, sub_batch=sub)
        assert codec.status() == 0
        assert int(total.item()) == ref_total and torch.equal(sizes, ref_sizes) and torch.equal(poff, ref_off), sub
        assert torch.equal(packed[:ref_total], ref_packed[:ref_total]), sub
        back = torch.zeros(n * L + 64, dtype=torch.uint8, device=dev)
        codec.decode_batch(packed, poff, sizes, n, L, back, out_offsets=d_off, out_sizes=d_sz)
        assert codec.status() == 0
        got = back[:n * L].cpu().numpy().reshape(n, L)
        for i in range(n):
            assert (got[i, :lens[i]] == blocks[i, :lens[i]]).all(), (sub, i)
        assert codec.info().last_pack_ms >= 0
    # a buffer that ends inside the batch: the streams past it stay out, the status word says so, the sizes still tell what is needed
    small = torch.zeros(ref_total // 2, dtype=torch.uint8, device=dev)
    poff = torch.zeros(n, dtype=torch.int64, device=dev); sizes = torch.zeros(n, dtype=torch.int32, device=dev); total = torch.zeros(1, dtype=torch.int64, device=dev)
    codec.encode_packed(d_in, n, L, small, poff, sizes, total, in_offsets=d_off, in_sizes=d_sz, sub_batch=16)
    assert codec.status() & 8
    assert int(total.item()) == ref_total and torch.equal(sizes, ref_sizes)
    fit = int((ref_off + ((ref_sizes.to(torch.int64) + 3) & ~3) <= small.numel()).sum().item())
    assert 0 < fit < n and torch.equal(small[:int(ref_off[fit].item())], ref_packed[:int(ref_off[fit].item())])
    codec.close()
