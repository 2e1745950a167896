"""Pins the oracle's CDF / division / speed / weight arithmetic to the reference's own unit tests.

Each test names the reference test it restates (paths relative to /root/reference)."""
import ctypes
import lzma
import os

import numpy as np
import pytest

import pyoracle as po
from conftest import GOLDEN

L = po.lib()
MED = po.Speed(0x30, 0x4000)   # probability/interface.rs:325


def new_cdf():
    c = po.Cdf16()
    L.orc_cdf_default(ctypes.byref(c))
    return c


def ssf(c, sym):
    o = po.SymStartFreq()
    assert L.orc_cdf_sym_to_start_and_freq(ctypes.byref(c), sym, ctypes.byref(o)) == 0
    return o


def test_default_cdf():
    # frequentist_cdf.rs:17-23
    assert list(new_cdf().cdf) == [4 * (i + 1) for i in range(16)]


def test_sym_to_start_and_freq():
    # common_tests.rs:3-22 test_sym_to_start_and_freq
    c = new_cdf()
    for i in range(100):
        L.orc_cdf_blend(ctypes.byref(c), i & 0xf, MED)
        last_start, last_freq = 0, 0
        for sym in range(16):
            r = ssf(c, sym)
            assert r.sym == sym
            expected = 1 + (0 if sym == 0 else last_start + last_freq)
            assert r.start == expected
            last_start, last_freq = r.start, r.freq


def test_cdf_offset_to_sym_start_and_freq():
    # common_tests.rs:24-42, every 15-bit offset for 100 successive blends
    c = new_cdf()
    o = po.SymStartFreq()
    for i in range(100):
        L.orc_cdf_blend(ctypes.byref(c), i & 0xf, MED)
        prev_sym = 0
        cdf = np.array(list(c.cdf), dtype=np.int64)
        # vectorised restatement of the search, cross-checked against the C oracle on a stride
        vals = np.arange(1 << 15, dtype=np.int64)
        resc = (vals * cdf[15]) >> 15
        sym = (resc[:, None] >= cdf[None, :15]).sum(axis=1)
        assert (np.diff(sym) >= 0).all() and sym[-1] == 15
        d = (cdf << 15) // cdf[15]
        start = np.where(sym == 0, 0, d[np.maximum(sym - 1, 0)]) + 1
        freq = d[sym] - (start - 1) - 1
        assert (start <= vals + 1).all() and (vals <= start + freq).all()
        for val in range(0, 1 << 15, 37):
            assert L.orc_cdf_offset_to_sym_start_and_freq(ctypes.byref(c), val, ctypes.byref(o)) == 0
            assert (o.sym, o.start, o.freq) == (sym[val], start[val], freq[val])
            assert prev_sym <= o.sym
            prev_sym = o.sym


def simple_rand(state):
    # common_tests.rs:44-48
    state = (state * 1103515245 + 12345) & 0xFFFFFFFFFFFFFFFF
    return state, (state // 65536) % 32768


def test_stationary_probability():
    # common_tests.rs:50-91 (1M LCG samples, seed 1) -- run through a tiny C loop via ctypes calls
    gt = [(0, 1), (0, 1), (1, 16), (0, 1), (1, 32), (1, 32), (0, 1), (0, 1),
          (1, 8), (0, 1), (0, 1), (0, 1), (1, 5), (1, 5), (1, 5), (3, 20)]
    cut = []
    s = np.float32(0.0)
    for a, b in gt:
        s = np.float32(s + np.float32(a) / np.float32(b))
        cut.append(int(np.round(np.float32(32768.0) * s)))
    assert cut[15] == 32768
    # generate the symbol sequence in numpy (the LCG is sequential but cheap enough at 1M)
    state = 1
    c = new_cdf()
    blend = L.orc_cdf_blend
    ref = ctypes.byref(c)
    cut_arr = np.array(cut)
    for _ in range(1000000):
        state, r = simple_rand(state)
        j = int(np.searchsorted(cut_arr, r, side="right"))
        blend(ref, j, MED)
    cdf = list(c.cdf)
    assert all(cdf[i] > cdf[i - 1] for i in range(1, 15)) and cdf[0] > 0   # valid()
    for i in range(16):
        pdf = cdf[i] - (cdf[i - 1] if i else 0)
        actual = pdf / cdf[15]
        expected = gt[i][0] / gt[i][1]
        abs_delta = abs(expected - actual)
        assert abs_delta < 0.014 or (expected > 0 and abs_delta / expected < 0.15)


def test_nonzero_pdf():
    # common_tests.rs:93-103 regression: 1M blends of symbol 15 keep every pdf > 0
    c = new_cdf()
    ref = ctypes.byref(c)
    for _ in range(1000000):
        L.orc_cdf_blend(ref, 15, MED)
    cdf = list(c.cdf)
    assert cdf[0] > 0 and all(cdf[i] - cdf[i - 1] > 0 for i in range(1, 15))


def np_blend(cdf, sym, inc, lim):
    """independent numpy restatement with explicit i16 wrapping (frequentist_cdf.rs:74-85)"""
    cdf = cdf.astype(np.int16).copy()
    cdf[sym:] = (cdf[sym:].astype(np.int32) + inc).astype(np.int16)
    if cdf[15] >= lim:
        t = (cdf.astype(np.int32) + np.arange(1, 17)).astype(np.int16)
        cdf = (t.astype(np.int32) - (t >> 2).astype(np.int32)).astype(np.int16)
    return cdf


def np_average(a, b, mix):
    a = a.astype(np.int64); b = b.astype(np.int64)
    prod = int(a[15] * b[15]) & 0xFFFFFFFF
    lz = 32 - prod.bit_length()
    sh = 17 - min(lz, 17)
    inv = (1 << 15) - mix
    ra = (a * b[15]) >> sh
    rb = (b * a[15]) >> sh
    return ((ra * mix + rb * inv + 1) >> 15).astype(np.int16)


def test_operation_helper_against_independent_restatement():
    # common_tests.rs:152-185 operation_test_helper: exact equality after every blend and for
    # average at 0, 1/4, 1/2, 3/4, 1 -- here C oracle vs the numpy restatement above
    buf0 = [0] * 5 + [1, 2, 3, 4] + [5] * 6 + [6, 7, 8, 8, 9, 9] + [10] * 10 + [11, 12, 12, 12, 13, 13, 13, 14] + [15] * 7
    buf1 = [0] * 5 + [1, 2, 3, 4] + [5] * 6
    assert len(buf0) == 46 and len(buf1) == 15
    c0, c1 = new_cdf(), new_cdf()
    n0 = np.array(list(c0.cdf), dtype=np.int16)
    n1 = n0.copy()
    for seq in (buf0, buf1):
        for s in seq:
            L.orc_cdf_blend(ctypes.byref(c0), s, MED)
            n0 = np_blend(n0, s, MED.inc, MED.lim)
            assert list(c0.cdf) == n0.tolist()
    out = po.Cdf16()
    for mix in (0, 1 << 13, 1 << 14, (1 << 14) + (1 << 13), 1 << 15):
        L.orc_cdf_average(ctypes.byref(c0), ctypes.byref(c1), mix, ctypes.byref(out))
        assert list(out.cdf) == np_average(n0, n1, mix).tolist()
    # average(.., 0) ~ other and average(.., all) ~ self (assert_cdf_similar :141-150)
    for mix, tgt in ((0, n1), (1 << 15, n0)):
        L.orc_cdf_average(ctypes.byref(c0), ctypes.byref(c1), mix, ctypes.byref(out))
        o = np.array(list(out.cdf), dtype=np.int64)
        t = tgt.astype(np.int64)
        assert (np.abs(o * t[15] - t * o[15]) < t[15] * o[15] // 160).all()


def test_blend_random_speeds_vs_numpy():
    rng = np.random.default_rng(7)
    for k in range(15):
        sp = L.orc_speed_palette(k)
        c = new_cdf()
        n = np.array(list(c.cdf), dtype=np.int16)
        for s in rng.integers(0, 16, size=3000):
            L.orc_cdf_blend(ctypes.byref(c), int(s), sp)
            n = np_blend(n, int(s), sp.inc, sp.lim)
        assert list(c.cdf) == n.tolist()


def test_divide_kat():
    # numeric.rs:73-85 test_divide
    nums = [3032127, 5049117, 16427165, 23282359, 35903174, 132971515, 163159927, 343856773, 935221996, 1829347323]
    dens = [115, 248, 267, 764, 1337, 4005, 4965, 9846, 24693, 31604]
    for n in nums:
        for d in dens:
            assert L.orc_fast_divide_30bit_by_16bit(n, d) == n // d


def test_reciprocal_tables_match_reference_lut():
    # div_lut.rs tables (generated by make_div_lut.rs) == the oracle's closed forms
    r8 = np.fromfile(os.path.join(GOLDEN, "reciprocal8.i64"), dtype=np.int64)
    assert r8[0] == 0
    assert (r8[1:] == 1 + (1 << 24) // np.arange(1, 256)).all()
    with lzma.open(os.path.join(GOLDEN, "reciprocal16.i64x2.xz")) as f:
        rec = np.frombuffer(f.read(), dtype=np.int64).reshape(65536, 2)
    d = np.arange(1, 32768, dtype=np.int64)
    bit_len = np.floor(np.log2(d)).astype(np.int64) + 1
    inv = ((((1 << bit_len) - d) << 31) // d) + 1
    assert (rec[1:32768, 0] == inv).all() and (rec[1:32768, 1] == bit_len - 1).all()
    # exactness of the reciprocal division for every (cdf<<15)/max the coder can form (make_div_lut.rs:37-39)
    rng = np.random.default_rng(3)
    for dd in rng.integers(1, 32768, size=200):
        n = (np.arange(0, int(dd) + 1, dtype=np.int64) << 15)
        m = int(inv[dd - 1]) * n
        q = ((m >> 31) + ((n - (m >> 31)) >> 1)) >> int(bit_len[dd - 1] - 1)
        assert (q == n // dd).all()
        for nn in (0, int(dd) << 15, (int(dd) // 2) << 15):
            assert L.orc_fast_divide_30bit_by_16bit(nn, int(dd)) == nn // int(dd)


def test_divide_16_by_8_exact():
    # make_div_lut.rs:11-23 asserts exactness for every u16 numerator and u8 divisor
    for d in (1, 2, 3, 7, 127, 128, 129, 200, 255):
        for n in range(0, 65536, 251):
            assert L.orc_fast_divide_16bit_by_8bit(n, d) & 0xFFFF == n // d


def test_speed_f8_roundtrip():
    # probability/interface.rs:590-616 test_u8_to_speed
    for x in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 96, 768, 1280, 1536, 1664):
        assert L.orc_u8_to_speed(L.orc_speed_to_u8(x)) == x


def test_weights_default_and_update_properties():
    # weights.rs:15-21 defaults; update keeps weights >= 1 and the normalised weight in [0, 32768]
    w = po.Weights()
    L.orc_weights_init(ctypes.byref(w))
    assert list(w.model_weights) == [1, 1] and w.mixing_param == 1 and w.normalized_weight == 1 << 14
    rng = np.random.default_rng(11)
    for _ in range(20000):
        p0, p1 = int(rng.integers(1, 32000)), int(rng.integers(1, 32000))
        lo, hi = min(p0, p1), max(p0, p1)
        pm = int(rng.integers(lo, hi + 1))
        probs = (ctypes.c_int16 * 2)(p0, p1)
        L.orc_weights_update(ctypes.byref(w), ctypes.byref(probs), pm)
        assert w.model_weights[0] >= 1 and w.model_weights[1] >= 1
        assert 0 <= (w.normalized_weight & 0xFFFF) <= 32768
    # a model that always assigns the higher probability must end up dominating
    L.orc_weights_init(ctypes.byref(w))
    for _ in range(2000):
        probs = (ctypes.c_int16 * 2)(20000, 2000)
        L.orc_weights_update(ctypes.byref(w), ctypes.byref(probs), 11000)
    assert (w.normalized_weight & 0xFFFF) > 30000


def test_context_luts_match_reference_constants():
    # constants.rs tables (RFC 7932 context lookups) vs the oracle's generated ones
    g = np.fromfile(os.path.join(GOLDEN, "context_luts.bin"), dtype=np.uint8)
    utf8, signed = g[:512], g[512:]
    buf = (ctypes.c_uint8 * 256)()
    L.orc_get_lut0(2, buf); assert list(buf) == utf8[:256].tolist()
    L.orc_get_lut1(2, buf); assert list(buf) == utf8[256:].tolist()
    L.orc_get_lut0(3, buf); assert list(buf) == (signed << 3).tolist()
    L.orc_get_lut1(3, buf); assert list(buf) == signed.tolist()
    L.orc_get_lut0(1, buf); assert list(buf) == [i >> 2 for i in range(256)]
    L.orc_get_lut0(0, buf); assert list(buf) == [i & 0x3f for i in range(256)]
    L.orc_get_lut1(0, buf); assert list(buf) == [0] * 256


def test_crc32c_kats():
    # codec/crc32.rs:95-116
    def crc(b, c=0):
        return L.orc_crc32c_update(c, b, len(b))
    assert crc(b"") == 0
    assert crc(b"123456789") == 0xE3069283
    assert crc(b"6789", crc(b"12345")) == 0xE3069283
    q = b"The quick brown fox jumps over the lazy dog"
    assert crc(q) == 0x22620404
    assert crc(q[18:], crc(q[:18])) == 0x22620404
