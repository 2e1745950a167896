"""the reference's own unit tests, run against the GPU device primitives (not only the oracle).

* probability/common_tests.rs:152-185 `operation_test_helper` -- two CDF implementations must agree exactly after every blend
  and at the mixing rates 0, 1/4, 1/2, 3/4, 1: implementation A = the device functions the kernels are made of (blend_row,
  blend_row_known_max, average_rows, scaled_div start/freq, the ballot search) through divans_gpu_selftest_cdf_ops,
  implementation B = the CPU restatements (C oracle and the independent Python one).
* common_tests.rs:3-126 `declare_common_tests!` -- start/freq chain identity, every 15-bit offset decodes to a monotone symbol
  inside its range, the LCG sample run (seed 1, common_tests.rs:44-48), repeated symbol 15 keeps all pdf > 0.
* codec/weights.rs through Weights::update sequences (row a19 had no isolated GPU test).
* test_ans.rs:177-260 `encode_test_nibble_helper` for its five TestSelection variants: the model runs on the host (independent
  restatement), its (start, freq) pairs go through the GPU rANS pass alone, bytes compared with ANSEncoder's.
"""
import ctypes
import os

import numpy as np
import pytest

import pyoracle as po
import ref_restatement as rr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MED, SLOW, MUD = (0x30, 0x4000), (0x20, 0x1000), (0x10, 0x2000)
BUF0 = [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 5, 5, 5, 5, 5, 6, 7, 8, 8, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 11, 12, 12, 12, 13, 13, 13, 14, 15,
        15, 15, 15, 15, 15, 15]
BUF1 = [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 5, 5, 5, 5, 5]


@pytest.fixture(scope="module")
def codec():
    import divans_amd as da
    c = da.LiteralCodec(da.config_simple(), 4096)
    yield c
    c.close()


class Mirror:
    """the same script on the CPU restatements; returns the expected records"""
    def __init__(self):
        self.reset()

    def reset(self):
        self.c = [rr.Cdf(), rr.Cdf()]
        self.o = [po.Cdf16(), po.Cdf16()]
        for x in self.o:
            po.lib().orc_cdf_default(ctypes.byref(x))
        self.w = rr.Weights(); self.w.mixing_param = 2
        self.ow = po.Weights(); po.lib().orc_weights_init(ctypes.byref(self.ow)); self.ow.mixing_param = 2

    def run(self, ops):
        L = po.lib()
        out = np.zeros((len(ops), 16), dtype=np.int64)
        for k, (kind, a, b, c) in enumerate(ops):
            if kind in (0, 1, 7):
                i = 1 if kind == 1 else 0
                self.c[i].blend(a, (b, c)); L.orc_cdf_blend(ctypes.byref(self.o[i]), a, po.Speed(b, c))
                assert list(self.o[i].cdf) == self.c[i].cdf
                out[k] = self.c[i].cdf
            elif kind == 2:
                m = self.c[0].average(self.c[1], a)
                om = po.Cdf16(); L.orc_cdf_average(ctypes.byref(self.o[0]), ctypes.byref(self.o[1]), a, ctypes.byref(om))
                assert list(om.cdf) == m.cdf
                out[k] = m.cdf
            elif kind == 3:
                s, f = self.c[0].sym_to_start_and_freq(a)
                out[k, :3] = (s, f, a)
            elif kind == 4:
                sym, s, f = self.c[0].cdf_offset_to_sym_start_and_freq(a)
                sf = po.SymStartFreq(); L.orc_cdf_offset_to_sym_start_and_freq(ctypes.byref(self.o[0]), a, ctypes.byref(sf))
                assert (sf.sym, sf.start, sf.freq) == (sym, s, f)
                out[k, :3] = (s, f, sym)
            elif kind == 5:
                self.w.update([a, b], c)
                probs = (ctypes.c_int16 * 2)(a, b); L.orc_weights_update(ctypes.byref(self.ow), probs, c)
                assert list(self.ow.model_weights) == self.w.model_weights and self.ow.normalized_weight == self.w.normalized_weight
                out[k, :3] = (self.w.model_weights[0], self.w.model_weights[1], self.w.normalized_weight & 0xFFFF)
            elif kind == 6:
                self.reset(); out[k] = self.c[0].cdf
        return out


def run_both(codec, ops):
    got = codec.selftest_cdf_ops(np.array(ops, dtype=np.uint32)).astype(np.int64)
    exp = Mirror().run(ops)
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, (int(bad[0]), ops[int(bad[0])], got[bad[0]].tolist(), exp[bad[0]].tolist())
    return got


def test_operation_test_helper_on_the_device_primitives(codec):
    Q = 1 << 15
    ops = []
    for s in [3, 3, 9, 14, 0, 15, 7, 7, 7, 2] * 3:       # give row 1 a shape of its own (the reference leaves it at the default)
        ops.append((1, s, *SLOW))
    for s in BUF0:
        ops.append((0, s, *MED))
    ops.append((2, Q >> 2, 0, 0))
    for s in BUF1:
        ops.append((7, s, *MED))                          # the pipelined kernels' blend variant must agree too
    for rate in (Q >> 2, Q >> 1, (Q >> 1) + (Q >> 2), 0, Q):
        ops.append((2, rate, 0, 0))
    got = run_both(codec, ops)
    # assert_cdf_similar(average(.., all), cdf0) / (average(.., 0), cdf1): within max0*max1/160 after cross-scaling (common_tests.rs:128-150)
    r0, r1 = got[len(ops) - 6], got[29]
    for rec, ref in ((got[-1], r0), (got[-2], r1)):
        m0, m1 = int(rec[15]), int(ref[15])
        assert all(abs(int(rec[i]) * m1 - int(ref[i]) * m0) < m0 * m1 // 160 for i in range(16))


def test_declare_common_tests_invariants(codec):
    ops = [(0, (i * 7 + 3) & 15, *MED) for i in range(100)]
    ops += [(3, s, 0, 0) for s in range(16)]
    n_blend = len(ops)
    ops += [(4, off, 0, 0) for off in range(0, 1 << 15, 1)]
    got = run_both(codec, ops)
    sf = got[100:116]
    for s in range(1, 16):                                 # common_tests.rs:14-17
        assert sf[s, 0] == 1 + sf[s - 1, 0] + sf[s - 1, 1]
    dec = got[n_blend:]
    assert (np.diff(dec[:, 2]) >= 0).all() and dec[0, 2] == 0 and dec[-1, 2] == 15     # monotone symbols, :29-39
    offs = np.arange(1 << 15)
    assert ((offs >= dec[:, 0] - 1) & (offs <= dec[:, 0] + dec[:, 1])).all()


def test_lcg_sample_run_and_symbol_15_stress(codec):
    # simple_rand, common_tests.rs:44-48 (seed 1): x = x * 1103515245 + 12345; symbols drawn from a fixed pdf through a prefix table
    x = 1
    pdf = [0.1, 0.01, 0.03, 0.2, 0.02, 0.05, 0.04, 0.15, 0.01, 0.06, 0.03, 0.07, 0.02, 0.1, 0.08, 0.03]
    edges = np.cumsum(pdf)
    ops = []
    for _ in range(60000):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        u = ((x >> 16) & 0x7FFF) / 32768.0
        ops.append((0, int(np.searchsorted(edges, u, side="right").clip(0, 15)), *MED))
    ops += [(0, 15, *MED)] * 30000                         # common_tests.rs:94-103
    got = run_both(codec, ops)
    final = got[-1]
    assert (np.diff(np.concatenate([[0], final])) > 0).all()      # every pdf entry still positive


def test_weights_update_sequences(codec):
    rng = np.random.default_rng(7)
    ops = []
    for k in range(30000):
        if k % 5000 == 0:
            ops.append((6, 0, 0, 0))
        mode = (k // 5000) % 3
        if mode == 0:
            p0, p1, pm = (int(v) for v in rng.integers(1, 32767, size=3))
        elif mode == 1:                                    # one model much better than the other: drives the weights to the 2^24 normalisation
            p0, p1 = int(rng.integers(20000, 32767)), int(rng.integers(1, 300)); pm = int(rng.integers(1000, 30000))
        else:
            p0 = p1 = pm = int(rng.integers(1, 32767))
        ops.append((5, p0, p1, pm))
    run_both(codec, ops)


# ---------------------------------------------------------------- test_ans.rs nibble helpers
def init_src(n):
    seed = np.fromfile(os.path.join(GOLDEN, "init_src_seed.bin"), dtype=np.uint8)
    return np.resize(seed, n)


VARIANTS = {  # adapt_probability, adaptive_context_mixing, independent_hilo, two_models   (test_ans.rs:86-176)
    "TestContextMixing": (True, True, False, True),
    "TestContextMixingPureAverage": (True, False, False, True),
    "TestAdapt": (True, False, False, False),
    "TestNoAdapt": (False, False, False, False),
    "TestSimple": (False, False, True, False),
}


def nibble_helper_pairs(src, adapt, acm, indep, two):
    """encode_test_nibble_helper, test_ans.rs:177-260, on the independent restatement; returns the (start, freq) pairs and ANSEncoder's bytes"""
    enc = rr.AnsEncoder()
    pairs = []
    weights = [rr.Weights(), rr.Weights()]
    cdf_high = rr.Cdf(); cdf_low = [rr.Cdf() for _ in range(16)]
    cdf_low_adv = [rr.Cdf() for _ in range(16)]; cdf_high_adv = [rr.Cdf() for _ in range(16)]
    last = 0
    Q2 = 1 << 13
    def put(sym, cdf):
        s, f = cdf.sym_to_start_and_freq(sym)
        pairs.append((s & 0xFFFF) | ((f & 0xFFFF) << 16)); enc.put_start_freq(s, f)
        return f
    for v in src:
        v = int(v)
        b0 = weights[0].norm_weight_as_u16_as_i32() if acm else Q2
        b1 = weights[1].norm_weight_as_u16_as_i32() if acm else Q2
        hi = v >> 4
        fr = put(hi, cdf_high.average(cdf_high_adv[last], b0) if two else cdf_high_adv[last])
        if acm:
            weights[0].update([cdf_high.sym_to_start_and_freq(hi)[1], cdf_high_adv[last].sym_to_start_and_freq(hi)[1]], fr)
        if adapt:
            if two:
                cdf_high.blend(hi, SLOW)
            cdf_high_adv[last].blend(hi, MED)
        cdfl = cdf_low[0 if indep else hi]
        lo = v & 0xF
        fr = put(lo, cdfl.average(cdf_low_adv[last], b1) if two else cdf_low_adv[last])
        if acm:
            weights[1].update([cdfl.sym_to_start_and_freq(lo)[1], cdf_low_adv[last].sym_to_start_and_freq(lo)[1]], fr)
        if adapt:
            if two:
                cdfl.blend(lo, SLOW)
            cdf_low_adv[last].blend(lo, SLOW)
        last = v & 0xF
    enc.flush_chunk()
    return np.array(pairs, dtype=np.uint32), bytes(enc.out)


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("source", ["init_src", "shuffle384"])
def test_ans_nibble_helper_streams_through_the_gpu_rans_pass(codec, variant, source, shuffle384):
    n = 4097 if source == "init_src" else 40000           # entropy16_* sizes are 16 / 4092 / 4097; 40 000 bytes crosses the 65 536-symbol chunk
    src = init_src(n) if source == "init_src" else np.resize(shuffle384, n)
    pairs, ref = nibble_helper_pairs(src, *VARIANTS[variant])
    got = codec.selftest_rans_pairs(pairs)
    assert got.tobytes() == ref
    # and the decoder side of the restatement reads those bytes back with the mirrored model (decode_test_nibble_helper, :262-368)
    dec = rr.AnsDecoder(ref)
    adapt, acm, indep, two = VARIANTS[variant]
    if not two and not acm:
        cdf_low_adv = [rr.Cdf() for _ in range(16)]; cdf_high_adv = [rr.Cdf() for _ in range(16)]
        last = 0
        for v in src[:2000]:
            hi, _, _ = dec.get_nibble(cdf_high_adv[last])
            if adapt:
                cdf_high_adv[last].blend(hi, MED)
            lo, _, _ = dec.get_nibble(cdf_low_adv[last])
            if adapt:
                cdf_low_adv[last].blend(lo, SLOW)
            assert ((hi << 4) | lo) == int(v)
            last = lo
