"""the C oracle against a second restatement of the same Rust (tests/ref_restatement.py, pure Python,
written from the .rs files alone), nibble by nibble -- (symbol, start, freq) of every coded nibble, the LIT-coder bytes, and
the decode direction -- on the golden corpus and on random configurations that reach every mixing value, prediction mode,
block type, speed and the mixing / non-mixing paths."""
import ctypes

import numpy as np
import pytest

import pyoracle as po
import ref_restatement as rr

PALETTE = [(0, 1024), (2, 1024), (1, 128), (1, 16384), (2, 2048), (4, 1024), (8, 8192), (16, 48), (16, 8192), (32, 4096), (64, 16384),
           (128, 256), (128, 16384), (512, 16384), (1664, 16384), (0x10, 0x2000)]      # probability/interface.rs:303-328


def to_rr(cfg):
    return dict(context_map=bytes(cfg.literal_context_map), mixing_mask=bytes(cfg.mixing_mask), prediction_mode=cfg.prediction_mode,
                btype=cfg.btype, mixing_param=cfg.context_mixing, speeds=[(s.inc, s.lim) for s in cfg.literal_adaptation])


def check(cfg, data):
    coded, trace = po.lit_encode(cfg, data, trace=True)
    got, gtrace = rr.encode_stream(to_rr(cfg), bytes(data))
    tr = [tuple(int(v) for v in row) for row in trace]
    assert gtrace == tr, next((i, a, b) for i, (a, b) in enumerate(zip(gtrace, tr)) if a != b)
    assert got == coded.tobytes()
    assert rr.decode_stream(to_rr(cfg), got, len(data)) == bytes(data)
    assert (po.lit_decode(cfg, np.frombuffer(got, dtype=np.uint8), len(data)) == data).all()


def test_benchmark_configs_on_corpus(corpus):
    for cfg in (po.config_simple(), po.config_context_mixing()):
        check(cfg, corpus[1000:1000 + 6000])
        check(cfg, corpus[200000:200000 + 3000])


def test_chunk_boundary(corpus):
    """more than 65 536 symbols: the LIFO chunk flush, the 16-byte state reload and the a/b swap at the seam"""
    data = np.concatenate([corpus[:30000], corpus[100000:103000]])      # 66 000 nibbles
    check(po.config_simple(), data)


def test_brotli_derived_configuration(corpus):
    import workload
    for btype in (0, 1):
        for mixing in (0, 2):
            cfg = workload.brotli_derived_config(po.LitConfig(), btype, mixing)
            check(cfg, corpus[5000:5000 + 2500])


@pytest.mark.parametrize("seed", range(6))
def test_random_configurations(seed, corpus):
    rng = np.random.default_rng(1234 + seed)
    for trial in range(40):
        cfg = po.LitConfig()
        nctx = int(rng.choice([1, 4, 64, 256]))
        cmap = rng.integers(0, nctx, size=po.MAX_CMAP, dtype=np.uint8) if nctx > 1 else np.zeros(po.MAX_CMAP, np.uint8)
        ctypes.memmove(cfg.literal_context_map, cmap.ctypes.data, cmap.size)
        if rng.random() < 0.3:
            mix = np.full(po.NUM_MIXING, int(rng.integers(0, 9)), dtype=np.uint8)
        else:
            mix = rng.integers(0, 9, size=po.NUM_MIXING, dtype=np.uint8)
        ctypes.memmove(cfg.mixing_mask, mix.ctypes.data, mix.size)
        cfg.prediction_mode = int(rng.integers(0, 4))
        cfg.btype = int(rng.integers(0, 4))
        cfg.context_mixing = int(rng.choice([0, 1, 2, 2, 5, 14]))
        for i in range(4):
            inc, lim = PALETTE[int(rng.integers(0, len(PALETTE)))]
            cfg.literal_adaptation[i].inc = inc; cfg.literal_adaptation[i].lim = lim
        n = int(rng.integers(1, 260))
        kind = trial % 4
        if kind == 0:
            data = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1:
            data = np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
        else:
            o = int(rng.integers(0, corpus.size - n))
            data = corpus[o:o + n].copy()
        check(cfg, data)


def test_second_restatement_reads_the_reference_trees_own_compressed_vector():
    """wasm/wasm.html:98-107 (tests/golden/ref_wasm_example.divans) through the Python restatement alone: every command of the CMD
    stream (tests/ref_cmd_walk.py) down to the encoder's start states, then the three Literal commands out of the LIT stream down to
    ITS start states, and the literals re-encoded give the example's LIT bytes back"""
    import os
    import ref_cmd_walk as rw
    c = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wasm_example.divans"), dtype=np.uint8)
    cmd, lit, _ = po.mux_demux(c[16:])
    w = rw.CmdWalk(cmd, example_build=True)
    assert w.run() == [(7, None), (4, (0, 0)), (3, 15), (1, (8, 4)), (3, 11), (2, (9, 648, 0)), (2, (7, 352, 0)), (3, 2), (1, (48, 288)), (15, None)]
    assert (w.nibbles, w.d.pos, w.d.state_a, w.d.state_b) == (8253, 44, 1 << 31, 1 << 31)
    assert w.pm["mode"] == 2 and w.pm["mixing_math"] == 0 and w.pm["speeds"] == [(16, 8192)] * 4
    assert set(w.pm["mixing_values"]) == {4} and w.pm["literal_context_map"] == [] and w.pm["distance_context_map"] == []
    with pytest.raises(Exception):
        bad = rw.CmdWalk(cmd, example_build=False)          # HEAD's two rows: noise
        bad.run()
        assert (bad.d.pos, bad.d.state_a, bad.d.state_b) == (44, 1 << 31, 1 << 31)
    cfg = dict(context_map=bytes(16384), mixing_mask=bytes([4]) * 8192, prediction_mode=w.pm["mode"], btype=0, mixing_param=0, speeds=w.pm["speeds"])
    history = lambda b: int.from_bytes((b"\0" * 8 + b)[-8:], "little")
    dec = rr.AnsDecoder(bytes(lit) + b"\0" * 16)
    lc = rr.LiteralCoder(**cfg)
    out = [lc.code_bytes(None, 15, dec=dec)]
    lc.last_8 = history(out[0] + b"ed, ")                   # Copy(distance 8, 4 bytes)
    out.append(lc.code_bytes(None, 11, dec=dec))
    lc.last_8 = history(b"g")                               # after two dictionary words: any byte whose row is still untouched
    out.append(lc.code_bytes(None, 2, dec=dec))
    assert out == [b"It snowed, rain", b"and hailed ", b".\n"]
    assert (dec.pos, dec.state_a, dec.state_b) == (36, 1 << 31, 1 << 31)
    enc = rr.AnsEncoder()
    lc = rr.LiteralCoder(**cfg)
    lc.code_bytes(out[0], 15, enc=enc)
    lc.last_8 = history(out[0] + b"ed, ")
    lc.code_bytes(out[1], 11, enc=enc)
    lc.last_8 = history(b"g")
    lc.code_bytes(out[2], 2, enc=enc)
    enc.flush_chunk()
    assert bytes(enc.out) == bytes(lit)
