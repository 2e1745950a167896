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


def _compare(cfg_name, blocks, cache_rows=None, blocks_grid=0):
    n, L = blocks.shape
    da, codec = _codec(cfg_name, max(L, 1))
    if cache_rows is not None or blocks_grid:
        codec.set_geometry(blocks=blocks_grid, cache_rows=cache_rows)
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
