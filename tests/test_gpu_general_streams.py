"""f3 on the GPU: the literals of general streams (Copy / Dict commands between the Literal commands, literal block-type
switches, one PredictionMode) coded by the segment entry points, bit-exact against the oracle driven by the same IR, for
every testdata IR; BASELINE configs[0](ii) = alice29-q11.ir with use_context_map + dynamic_context_mixing = 2."""
import numpy as np
import pytest

import irtext
import pyoracle as po

pytestmark = pytest.mark.gpu
NAMES = ["alice29", "alice29-q11", "alice29-priors", "asyoulik", "random_then_unicode", "ends_with_truncated_dictionary"]


def _segment_tensors(torch, streams):
    """streams: list of (lit uint8[n], segs structured array) -> device tensors for the segment entry points"""
    dev = torch.device("cuda")
    lit = np.concatenate([s[0] for s in streams]) if streams else np.zeros(0, np.uint8)
    sizes = np.array([s[0].size for s in streams], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1].astype(np.int64))]).astype(np.int64)
    segs = np.concatenate([s[1] for s in streams])
    seg_begin = np.concatenate([[0], np.cumsum([s[1].size for s in streams])]).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(np.concatenate([lit, np.zeros(64, np.uint8)])), t(offs), t(sizes), t(seg_begin), t(segs.view(np.uint8)), int(sizes.max()))


def _roundtrip(da, torch, cfg, n_btypes, streams, oracle_cfg):
    d_lit, d_off, d_sz, d_sb, d_segs, longest = _segment_tensors(torch, streams)
    n = len(streams)
    codec = da.LiteralCodec(cfg, max(longest, 16))
    codec.set_block_types(n_btypes)
    outs = codec.alloc_encode_outputs(n, max(longest, 16))
    codec.encode_segments_batch(d_lit, d_off, d_sz, n, longest, d_sb, d_segs, outs)
    offs = outs["offsets"].cpu().numpy(); sz = outs["sizes"].cpu().numpy()
    assert codec.status() == 0
    for i, (lit, segs) in enumerate(streams):
        got = outs["out"][int(offs[i]):int(offs[i]) + int(sz[i])].cpu().numpy()
        ref = po.lit_segments_encode(oracle_cfg, lit, segs["len"], segs["btype"], segs["last8"])
        assert got.size == ref.size and (got == ref).all(), f"stream {i}: LIT bytes differ from the oracle"
        assert (po.lit_segments_decode(oracle_cfg, ref, lit.size, segs["len"], segs["btype"], segs["last8"]) == lit).all()
    back = torch.zeros_like(d_lit)
    codec.decode_segments_batch(outs["out"], outs["offsets"], outs["sizes"], n, longest, d_sb, d_segs, back, d_off, d_sz)
    assert codec.status() == 0
    total = int(d_sz.sum().item())
    assert torch.equal(back[:total], d_lit[:total])
    codec.close()


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mixing", [0, 2])
def test_ir_literals_bit_exact_vs_oracle(name, mixing):
    import torch
    import divans_amd as da
    ir = da.CommandIR(irtext.load_ir_text(name))
    lit, segs = ir.literal_segments()
    opts = dict(dynamic_context_mixing=mixing, use_context_map=1)
    cfg = ir.lit_config(**opts)
    ocfg = po.LitConfig.from_buffer_copy(bytes(cfg))      # same struct layout (tests/test_abi_cpu.py)
    _roundtrip(da, torch, cfg, ir.num_block_types, [(lit, segs)], ocfg)
    ir.close()


def test_many_general_streams_in_one_batch():
    """a batch of streams with different segment lists under one configuration: every stream cut from alice29-priors' command list"""
    import torch
    import divans_amd as da
    ir = da.CommandIR(irtext.load_ir_text("alice29-priors"))
    lit, segs = ir.literal_segments()
    cfg = ir.lit_config(dynamic_context_mixing=2)
    ocfg = po.LitConfig.from_buffer_copy(bytes(cfg))
    ends = np.cumsum(segs["len"].astype(np.int64))
    streams = []
    for k in range(40):                       # stream k = commands [a, b): ragged, some tiny, block types 0 and 1 mixed
        a = (k * 149) % (segs.size - 400); b = a + 1 + (k * 37) % 390
        lo = int(ends[a - 1]) if a else 0
        streams.append((lit[lo:int(ends[b - 1])].copy(), segs[a:b].copy()))
    _roundtrip(da, torch, cfg, ir.num_block_types, streams, ocfg)
    ir.close()


def test_segments_without_context_change_equal_plain_streams(corpus):
    """splitting a stream into segments that carry the natural last 8 bytes and one block type changes nothing"""
    import torch
    import divans_amd as da
    import workload
    blocks = workload.make_blocks(corpus, 3, 6, block_len=5000)
    for cfg, ocfg in ((da.config_simple(), po.config_simple()), (da.config_context_mixing(), po.config_context_mixing())):
        streams = []
        for b in blocks:
            cuts = [0, 1, 9, 10, 700, 4096, 5000]
            segs = np.zeros(len(cuts) - 1, dtype=np.dtype([("len", "<u4"), ("btype", "<u4"), ("last8", "<u8")]))
            for j in range(len(cuts) - 1):
                tail = bytes(b[max(0, cuts[j] - 8):cuts[j]]).rjust(8, b"\0")
                segs[j] = (cuts[j + 1] - cuts[j], cfg.btype, int.from_bytes(tail, "little"))
            streams.append((b.copy(), segs))
        d_lit, d_off, d_sz, d_sb, d_segs, longest = _segment_tensors(torch, streams)
        codec = da.LiteralCodec(cfg, 5000)
        codec.set_block_types(cfg.btype + 1)
        outs = codec.alloc_encode_outputs(6, 5000)
        codec.encode_segments_batch(d_lit, d_off, d_sz, 6, longest, d_sb, d_segs, outs)
        offs = outs["offsets"].cpu().numpy(); sz = outs["sizes"].cpu().numpy()
        for i, b in enumerate(blocks):
            got = outs["out"][int(offs[i]):int(offs[i]) + int(sz[i])].cpu().numpy()
            assert (got == po.lit_encode(ocfg, b)).all()
        codec.close()


def test_segment_block_type_outside_the_tables_is_reported(corpus):
    """a segment naming a literal block type the codec holds no context table for raises the BAD_SEGMENT status bit in both
    directions and for both decoder generations (it used to be clamped silently)"""
    import torch
    import divans_amd as da
    cfg = da.config_context_mixing()
    b = corpus[1000:5000].copy()
    segs = np.zeros(2, dtype=np.dtype([("len", "<u4"), ("btype", "<u4"), ("last8", "<u8")]))
    segs[0] = (1000, 0, 0); segs[1] = (3000, 5, 0)          # tables exist for block types 0 and 1 only
    d_lit, d_off, d_sz, d_sb, d_segs, longest = _segment_tensors(torch, [(b, segs)])
    codec = da.LiteralCodec(cfg, 4000)
    codec.set_block_types(2)
    outs = codec.alloc_encode_outputs(1, 4000)
    codec.encode_segments_batch(d_lit, d_off, d_sz, 1, longest, d_sb, d_segs, outs)
    assert codec.status() & 4
    back = torch.zeros(4000 + 64, dtype=torch.uint8, device=d_lit.device)
    for gen in (g for g in (1, 2, 3) if g in da.decoder_generations()):
        codec.set_decoder(gen)
        codec.decode_segments_batch(outs["out"], outs["offsets"], outs["sizes"], 1, longest, d_sb, d_segs, back, d_off, d_sz)
        assert codec.status() & 4, gen
    codec.close()
