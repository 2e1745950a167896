"""A SECOND restatement of the reference's literal-coding path, in pure Python, written from the Rust sources alone
(src/probability/frequentist_cdf.rs, src/probability/interface.rs, src/probability/numeric.rs, src/ans.rs,
src/codec/weights.rs, src/codec/literal.rs, src/codec/interface.rs) -- not from oracle/*.c -- so that a misreading
would have to be made twice, independently, to go unnoticed.  Test infrastructure only; slow
(tens of microseconds per nibble), used on kilobyte-sized streams.

Integer types follow the Rust declarations literally: Prob = i16 with wrapping adds, i32 products, i64 wrapping
arithmetic in Weights, u64 rANS states.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOG2_SCALE = 15
BLEND_FIXED_POINT_PRECISION = 15
M64 = (1 << 64) - 1


def i16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def i64(v):
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def lz32(v):        # i32::leading_zeros on the two's complement pattern
    return 32 - (v & 0xFFFFFFFF).bit_length()


def lz64(v):
    return 64 - (v & M64).bit_length()


# ---------------------------------------------------------------- FrequentistCDF16 (frequentist_cdf.rs)
class Cdf:
    __slots__ = ("cdf",)

    def __init__(self, cdf=None):
        self.cdf = list(cdf) if cdf is not None else [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64]

    def max(self):
        return self.cdf[15]

    def average(self, other, mix_rate):                      # frequentist_cdf.rs:58-72
        ourmax = self.max(); othermax = other.max()
        prod = i32(ourmax * othermax)
        lzc = min(lz32(prod), 17)
        shift = 17 - lzc
        inv = (1 << BLEND_FIXED_POINT_PRECISION) - mix_rate
        out = []
        for s, o in zip(self.cdf, other.cdf):
            rs = i32(s * othermax) >> shift
            ro = i32(o * ourmax) >> shift
            out.append(i16(i32(i32(rs * mix_rate) + i32(ro * inv) + 1) >> BLEND_FIXED_POINT_PRECISION))
        return Cdf(out)

    def blend(self, symbol, speed):                          # frequentist_cdf.rs:74-85
        inc, lim = speed
        c = self.cdf
        for i in range(symbol, 16):
            c[i] = i16(c[i] + inc)
        if c[15] >= lim:
            for i in range(16):
                t = i16(c[i] + (i + 1))
                c[i] = i16(t - (t >> 2))

    def sym_to_start_and_freq(self, sym):                    # probability/interface.rs:97-108 (div_by_max = '/', truncating)
        mx = self.max()
        def div(v):
            q = abs(v) // abs(mx)
            return q if (v >= 0) == (mx > 0) else -q
        cdf_sym = div(self.cdf[sym] << LOG2_SCALE)
        cdf_prev = div(self.cdf[sym - 1] << LOG2_SCALE) if sym != 0 else 0
        freq = cdf_sym - cdf_prev
        return i16(i16(cdf_prev) + 1), i16(i16(freq) - 1)

    def cdf_offset_to_sym_start_and_freq(self, off):         # probability/interface.rs:136-198
        r = i16(i32(off * self.max()) >> LOG2_SCALE)
        sym = 15
        for i in range(15):
            if r < self.cdf[i]:
                sym = i
                break
        start, freq = self.sym_to_start_and_freq(sym)
        return sym, start, freq


# ---------------------------------------------------------------- Weights (codec/weights.rs)
_RECIPROCAL8 = None


def reciprocal8(d):
    """div_lut.rs RECIPROCAL8, the table itself (tests/golden/reciprocal8.i64 <- src/probability/div_lut.rs)"""
    global _RECIPROCAL8
    if _RECIPROCAL8 is None:
        _RECIPROCAL8 = [int(x) for x in np.fromfile(os.path.join(GOLDEN, "reciprocal8.i64"), dtype=np.int64)]
    return _RECIPROCAL8[d]


class Weights:
    def __init__(self):
        self.model_weights = [1, 1]
        self.mixing_param = 1
        self.normalized_weight = 1 << (BLEND_FIXED_POINT_PRECISION - 1)

    def norm_weight_as_u16_as_i32(self):                     # literal.rs:230
        return self.normalized_weight & 0xFFFF

    def update(self, model_probs, weighted_prob):            # weights.rs:23-38
        w = self.model_weights
        if ((w[0] | w[1]) & 0x7f000000) != 0:                # normalize_weights / fix_weights :64-80
            ilog = 32 - min(lz32(w[0]), lz32(w[1]))
            if ilog >= 24:
                w[0] >>= ilog - 24; w[1] >>= ilog - 24
        new = [self._new_weight(model_probs, weighted_prob, w, idx) for idx in (0, 1)]
        self.model_weights = new
        total = new[0] + new[1]                              # i64
        shift = max(56 - lz64(total), 0)
        total_8bit = (total >> shift) & 0xFF                 # `as u8`
        num = ((new[0] >> shift) & 0xFFFF)                   # `as u16`
        num = (num << 8) & 0xFFFF                            # u16 << 8 (release build: wraps)
        q = i16((reciprocal8(total_8bit) * num) >> 24)       # fast_divide_16bit_by_8bit, numeric.rs:62-64
        self.normalized_weight = i16(q << (BLEND_FIXED_POINT_PRECISION - 8))

    @staticmethod
    def _new_weight(probs, weighted_prob, weights, index):   # weights.rs:110-133 (integer variant), all i64 wrapping
        p1 = weighted_prob
        total = 1 << LOG2_SCALE
        p0 = i64(total - p1)
        n1i = probs[index]
        ni = 1 << LOG2_SCALE
        error = i64(total - p1)
        wi = weights[index]
        efficacy = i64(i64(total * n1i) - i64(p1 * ni))
        lg = 64 - lz64(i64(p1 * p0))
        adj = i64(error * efficacy) >> lg if lg < 64 else (0 if i64(error * efficacy) >= 0 else -1)
        return max(1, i32(i64(wi + adj)))


# ---------------------------------------------------------------- ANS (ans.rs)
NORMALIZATION_INTERVAL = 1 << 31
SCALE_MASK = (1 << LOG2_SCALE) - 1
NUM_SYMBOLS_BEFORE_FLUSH = (256 * 1024) >> 2


class AnsEncoder:
    def __init__(self):
        self.pairs = []          # start_freq ByteStack, oldest first
        self.out = bytearray()   # what the Mux would be handed, in order

    def put_start_freq(self, start, freq):                   # ans.rs:287-301
        assert freq != 0
        self.pairs.append((start, freq))
        if len(self.pairs) == NUM_SYMBOLS_BEFORE_FLUSH:
            self.flush_chunk()

    def flush_chunk(self):                                   # ans.rs:331-378: bytes() of the stack = newest symbol first
        if not self.pairs:
            return
        a = b = NORMALIZATION_INTERVAL
        q = []                                               # the output ByteStack, as a list we later read newest-first
        for start, freq in reversed(self.pairs):
            f = freq & M64 if freq >= 0 else (freq + (1 << 64))       # `freq as u64` sign-extends
            lim = ((((NORMALIZATION_INTERVAL >> LOG2_SCALE) << 32) * f) & M64)
            state = a
            if state >= lim:
                q.append(bytes([state & 0xff, (state >> 8) & 0xff, (state >> 16) & 0xff, (state >> 24) & 0xff]))
                state >>= 32
            x = ((((state // f) << LOG2_SCALE) & M64) + state % f + (start & M64 if start >= 0 else start + (1 << 64))) & M64
            a, b = b, x
        a, b = b, a                                          # unconditional swap, :354-356
        q.append(a.to_bytes(8, "little") + b.to_bytes(8, "little"))
        for item in reversed(q):                             # a stack: the last thing pushed is read first
            self.out += item
        self.pairs = []


class AnsDecoder:
    def __init__(self, data):
        self.data = bytes(data); self.pos = 0
        self.state_a = 0; self.state_b = 0; self.sym_count = 0
        self.need_a = 8; self.need_b = 0                     # Default: buffer_a_bytes_required 8 "this will load both buffers"

    def _fill(self):                                         # push_data :428-442 + helper_push_data_rare_cases :173-191 with whole words at hand
        if self.need_a == 0:
            return
        if self.need_a == 1:
            self.state_a = ((self.state_a << 32) & M64) | int.from_bytes(self.data[self.pos:self.pos + 4], "little")
            self.pos += 4; self.need_a = 0
            return
        assert 4 < self.need_a < 16
        self.sym_count = 0
        self.state_a = int.from_bytes(self.data[self.pos:self.pos + 8], "little")
        self.state_b = int.from_bytes(self.data[self.pos + 8:self.pos + 16], "little")
        self.pos += 16; self.need_a = 0

    def get_nibble(self, cdf):                               # get_nibble_internal :246-252 + helper_advance_sym :230-244
        self._fill()
        off = i16(self.state_a & SCALE_MASK)
        sym, start, freq = cdf.cdf_offset_to_sym_start_and_freq(off)
        self.need_a = self.need_b
        self.need_a |= (1 if self.sym_count == NUM_SYMBOLS_BEFORE_FLUSH - 1 else 0) << 3
        x = ((freq & M64) * (self.state_a >> LOG2_SCALE) + (self.state_a & SCALE_MASK) - (start & M64)) & M64
        self.sym_count = (self.sym_count + 1) & 0xFFFF
        self.need_b = 1 if x < NORMALIZATION_INTERVAL else 0
        self.state_a = self.state_b
        self.state_b = x
        return sym, start, freq


# ---------------------------------------------------------------- literal coder (codec/literal.rs, codec/interface.rs)
def context_luts(mode):
    """get_lut0 / get_lut1, codec/interface.rs:199-238; tables = src/constants.rs (tests/golden/context_luts.bin)"""
    raw = np.fromfile(os.path.join(GOLDEN, "context_luts.bin"), dtype=np.uint8)
    utf8, signed = raw[:512], raw[512:768]
    if mode == 3:
        return [(int(x) << 3) & 0xFF for x in signed], [int(x) for x in signed]
    if mode == 2:
        return [int(x) for x in utf8[:256]], [int(x) for x in utf8[256:]]
    if mode == 1:
        return [i >> 2 for i in range(256)], [0] * 256
    return [i & 0x3F for i in range(256)], [0] * 256


class LiteralCoder:
    """LiteralState::code_nibble_array + code_nibble with the LiteralBookKeeping fields they read."""

    def __init__(self, context_map, mixing_mask, prediction_mode, btype, mixing_param, speeds):
        self.cmap = [int(x) for x in context_map]
        self.mixing_mask = [int(x) for x in mixing_mask]
        self.lut0, self.lut1 = context_luts(prediction_mode)
        self.btype = btype
        self.speeds = [(int(a), int(b)) for a, b in speeds]
        self.weights = [Weights(), Weights()]
        for w in self.weights:
            w.mixing_param = mixing_param
        self.mixing = mixing_param > 1                       # should_mix -> MixingTrait
        self.last_8 = 0
        self.high, self.low, self.cm = {}, {}, {}            # default-initialised prior tables, materialised on first touch
        self.trace = []

    @staticmethod
    def _row(table, key):
        r = table.get(key)
        if r is None:
            r = table[key] = Cdf()
        return r

    def _code_nibble(self, is_high, nibble, ctx, prev_byte, stride_bytes, cur_byte_prior, enc, dec):   # literal.rs:154-259
        mmi = ctx
        if not is_high:
            mmi |= (cur_byte_prior & 0xF) << 8
            mmi |= 4096
        else:
            mmi |= (prev_byte >> 4) << 8
        mm_opts = self.mixing_mask[mmi]
        fast_cm = 0xFF if mm_opts != 3 else 0
        mm = 0xFF if (mm_opts != 0 and mm_opts != 3) else 0
        opt1 = 0xF if mm_opts == 1 else 0
        stride_offset = 0 if mm_opts < 4 else min(7, mm_opts ^ 4) << 3
        sb = (stride_bytes >> (0x38 - stride_offset)) & 0xFF
        if is_high:
            index_b = sb & mm & (~opt1 & 0xFF)
            index_c = ctx
        else:
            index_b = (mm & sb) | ((~mm & 0xFF) & ctx)
            index_c = (cur_byte_prior & fast_cm) | ((ctx & opt1) << 4)
        row = self._row(self.high if is_high else self.low, ((mm >> 7) ^ (opt1 >> 2), index_b, index_c))
        coder_prior = Cdf() if mm_opts == 2 else row
        if self.mixing:
            cm = self._row(self.cm, ("first", ctx) if is_high else ("second", cur_byte_prior, ctx))
            w = self.weights[1 if is_high else 0]
            prob = cm.average(row, w.norm_weight_as_u16_as_i32())
            nibble, start, freq = self._get_or_put(nibble, prob, enc, dec)
            probs = [cm.sym_to_start_and_freq(nibble)[1], row.sym_to_start_and_freq(nibble)[1]]
            w.update(probs, freq)
            cm.blend(nibble, self.speeds[2 | (1 if is_high else 0)])
        else:
            nibble, start, freq = self._get_or_put(nibble, coder_prior, enc, dec)
        self.trace.append((nibble, start, freq))
        return nibble, (None if mm_opts == 2 else row)

    @staticmethod
    def _get_or_put(nibble, cdf, enc, dec):
        if enc is not None:
            start, freq = cdf.sym_to_start_and_freq(nibble)
            enc.put_start_freq(start, freq)
            return nibble, start, freq
        return dec.get_nibble(cdf)

    def code_bytes(self, data, n, enc=None, dec=None):       # code_nibble_array, literal.rs:261-394
        out = bytearray()
        for k in range(n):
            byte = data[k] if enc is not None else 0
            prev = (self.last_8 >> 0x38) & 0xFF
            pp = (self.last_8 >> 0x30) & 0xFF
            sel = self.lut0[prev] | self.lut1[pp]
            ctx = self.cmap[sel + (self.btype << 6)]
            stride_bytes = self.last_8
            h, prob = self._code_nibble(True, byte >> 4, ctx, prev, stride_bytes, 0, enc, dec)
            if prob is not None:
                prob.blend(h, self.speeds[0])
            lo, prob = self._code_nibble(False, byte & 0xF, ctx, prev, stride_bytes, h, enc, dec)
            cur = lo | (h << 4)
            self.last_8 = (self.last_8 >> 8) | (cur << 0x38)   # push_literal_byte, codec/interface.rs:280-284
            if prob is not None:
                prob.blend(lo, self.speeds[0])
            out.append(cur)
        return bytes(out)


def encode_stream(cfg, data):
    """cfg: dict(context_map, mixing_mask, prediction_mode, btype, mixing_param, speeds) -> (LIT bytes, trace)"""
    lc = LiteralCoder(**cfg)
    enc = AnsEncoder()
    lc.code_bytes(bytes(data), len(data), enc=enc)
    enc.flush_chunk()
    return bytes(enc.out), lc.trace


def decode_stream(cfg, coded, n):
    lc = LiteralCoder(**cfg)
    return lc.code_bytes(None, n, dec=AnsDecoder(coded))
