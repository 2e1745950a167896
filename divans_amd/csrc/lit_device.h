// lit_device.h -- device helpers shared by the literal-coder kernels (lit_kernels.hip, lit_decode2.hip).
#ifndef DIVANS_LIT_DEVICE_H_
#define DIVANS_LIT_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lit_kernels.h"

namespace divans_hip {

#define DPP_ROW_SHR1 0x111
#define DPP_ROW_BCAST(n) (0x150 + (n))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// value of lane `n` of this 16-lane row, broadcast to the whole row (v_mov_b32_dpp row_newbcast)
template <int N>
__device__ __forceinline__ int row_bcast(int v) {
    // every lane of the row receives lane N, so the "old" operand is never used: leave it undefined (no zero-fill move)
    return __builtin_amdgcn_mov_dpp(v, DPP_ROW_BCAST(N), 0xf, 0xf, false);
}
// value of the previous lane in the row, 0 for the first lane
__device__ __forceinline__ int row_prev_or_zero(int v) {
    return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, true);
}
// value of the next lane in the row, 0 for the last lane
__device__ __forceinline__ int row_next_or_zero(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
}
// value of lane (row_base + idx) -- idx is row-uniform but not compile-time
__device__ __forceinline__ int row_gather(int v, int row_base_lane, int idx) {
    return __builtin_amdgcn_ds_bpermute((row_base_lane + idx) << 2, v);
}

// floor(n / d) for 0 <= n < 2^31, 1 <= d < 2^15, given rcp ~= 1/d (v_rcp_f32, 1 ulp).
// The float estimate is within 2^-6 of the true quotient, so one step of correction makes it exact.
__device__ __forceinline__ uint32_t exact_div(uint32_t n, uint32_t d, float rcp) {
    uint32_t q = (uint32_t)((float)n * rcp);
    int32_t r = (int32_t)(n - q * d);
    q = r < 0 ? q - 1u : q;
    q = r >= (int32_t)d ? q + 1u : q;
    return q;
}

// floor((c << 15) / d) for 0 <= c <= d < 2^15 with a reciprocal biased low: `rcp_lo` = rcp(d) * 2^15 * (1 - 2^-20).
// (float)c is exact, the product's relative error is < 2^-22, so the truncated estimate is q or q-1 (never above)
// and a single compare finishes it.  Checked exhaustively on the GPU (selftest_division_kernel).
__device__ __forceinline__ float biased_rcp15(int d) {
    return __builtin_amdgcn_rcpf((float)d) * (32768.0f * (1.0f - 9.5367431640625e-07f));
}
__device__ __forceinline__ uint32_t scaled_div(int c, int d, float rcp_lo) {
    uint32_t q = (uint32_t)((float)c * rcp_lo);
    int32_t r = (c << 15) - __mul24((int)q, d);      // q < 2^16, d < 2^15: 24-bit multiply is exact
    return r >= d ? q + 1u : q;
}

// Which 128-byte line of a stream's table a row shares with three others is free to choose (the table is private to the launch,
// any bijection of the row index decodes the same bytes), and it decides how many of the stream's row accesses the L2 can serve:
// with 28 672 resident streams a stream owns about 1 KB of it, eight lines.  The stride-1 tables are indexed by the previous
// byte; its four neighbours in a line are the next byte values in THIS order instead of numeric order -- lower-case letters by
// their usual English frequency first, then the separators, capitals, digits, everything else numerically -- so that the bytes
// text keeps coming back to sit together (an LRU model of 1 KB per stream misses 51 % instead of 67 % of the low-row accesses
// of the benchmark text; for data without such a skew the order is as good as any other).
struct BytePerm {
    uint8_t rank[256];
    constexpr BytePerm() : rank{} {
        const char order[] = " etaoinshrdlcumwfgypbvkjxqz\n,.;'\"-!?:()TAISOWHBCMNEPDLFRGYUVKJQXZ0123456789";
        bool used[256] = {};
        uint32_t n = 0;
        for (uint32_t i = 0; i + 1u < sizeof(order); ++i) { const uint8_t b = (uint8_t)order[i]; if (!used[b]) { used[b] = true; rank[b] = (uint8_t)n++; } }
        for (uint32_t b = 0; b < 256u; ++b) if (!used[b]) rank[b] = (uint8_t)n++;
    }
};

// frequentist_cdf.rs:58-72: self = cm row, other = stride row.  Every factor is below 2^15 (and the shifted products
// stay below 2^15, see the bound on `sh`), so the full-rate 24-bit multiplies give the same 32-bit results as `*`.
__device__ __forceinline__ int average_rows(int cm, int st, int cmax, int smax, int mix_rate) {
    const uint32_t prod = __umul24((uint32_t)cmax, (uint32_t)smax);
    int lz = __clz((int)prod);
    lz = lz > 17 ? 17 : lz;
    const int sh = 17 - lz;
    const uint32_t inv = (uint32_t)((1 << 15) - mix_rate);
    const uint32_t rs = __umul24((uint32_t)cm, (uint32_t)smax) >> sh;
    const uint32_t ro = __umul24((uint32_t)st, (uint32_t)cmax) >> sh;
    return (int)(__umul24(rs, (uint32_t)mix_rate) + __umul24(ro, inv) + 1u) >> 15;
}

struct Weights { int w0, w1; int norm; };  // weights.rs:4-8 (norm = normalized_weight as u16)

// weights.rs:110-133 in 32 bits.  error = 2^15 - pmix and prob_i - pmix are below 2^16 in magnitude, so the reference's
// i64 product error * efficacy is (error * (prob_i - pmix)) << 15 with a factor that fits an int; its arithmetic shift by
// lg = bit length of pmix * (2^15 - pmix) (< 2^28) and the wrapping i32 add are then one left or right shift of that int.
__device__ __forceinline__ int new_weight(int prob_i, int pmix, int error, int lg, int wi) {
    const int prod = __mul24(error, prob_i - pmix);      // |error| <= 2^15, |prob_i - pmix| < 2^16: the full-rate 24-bit multiply is exact
    const int l = 15 - lg;
    const int adj = l >= 0 ? (int)((uint32_t)prod << (l & 31)) : (prod >> ((-l) & 31));
    const int nw = (int)((uint32_t)wi + (uint32_t)adj);
    return nw > 1 ? nw : 1;
}

// weights.rs:23-38 + normalize_weights :64-80 + compute_normalized_weight :54-62
__device__ __forceinline__ void weights_update(Weights& w, int p_cm, int p_stride, int pmix) {
    if (((w.w0 | w.w1) & 0x7f000000) != 0) {
        int lz0 = __clz(w.w0), lz1 = __clz(w.w1);
        int ilog = 32 - (lz0 < lz1 ? lz0 : lz1);
        if (ilog >= 24) { w.w0 >>= (ilog - 24); w.w1 >>= (ilog - 24); }
    }
    const int error = (1 << 15) - pmix;
    const uint32_t geo = (uint32_t)__mul24(pmix, error);           // full_model_sum_p1 * full_model_sum_p0, below 2^28 (both factors below 2^16)
    const int lg = geo ? 32 - __clz((int)geo) : 0;
    const int n0 = new_weight(p_cm, pmix, error, lg, w.w0);
    const int n1 = new_weight(p_stride, pmix, error, lg, w.w1);
    w.w0 = n0; w.w1 = n1;
    const uint32_t total = (uint32_t)n0 + (uint32_t)n1;            // both in [1, 2^31): the i64 sum fits 32 bits
    int shift = 24 - __clz((int)total);                            // 56 - leading_zeros of the i64
    shift = shift < 0 ? 0 : shift;
    const uint32_t t8 = (total >> shift) & 0xffu;
    const uint32_t num = ((uint32_t)(n0 >> shift) << 8) & 0xffffu;
    // fast_divide_16bit_by_8bit == exact '/' (make_div_lut.rs:11-23); RECIPROCAL8[0] == 0
    uint32_t q = 0u;
    if (t8) {       // exact_div with the 24-bit multiply: num < 2^16, t8 < 2^8
        q = (uint32_t)((float)num * __builtin_amdgcn_rcpf((float)t8));
        const int32_t r = (int32_t)(num - __umul24(q, t8));
        q = r < 0 ? q - 1u : q;
        q = r >= (int32_t)t8 ? q + 1u : q;
    }
    w.norm = (int)((q << 7) & 0xffffu);
}

// The Weights update (weights.rs:23-38) is per-stream scalar work.  Both nibbles of a byte own a Weights object
// (model_weights[1] high, [0] low, literal.rs:230) that is only read again one byte later, so the mixing paths hand the three
// probabilities of each nibble back (`wfreqs` = cm freq | stride freq << 16, the mixed freq is in the returned pair) and the byte
// loop runs ONE update for both: lanes 0..7 of the row carry the high nibble's Weights, lanes 8..15 the low nibble's.
struct WeightsPair {
    Weights w;   // lane-varying: high nibble's object in lanes 0..7, low nibble's in lanes 8..15
    __device__ __forceinline__ void init() { w.w0 = 1; w.w1 = 1; w.norm = 1 << 14; }
    __device__ __forceinline__ int norm_high() const { return row_bcast<0>(w.norm); }
    __device__ __forceinline__ int norm_low() const { return row_bcast<8>(w.norm); }
    __device__ __forceinline__ void update(int li, uint32_t freqs_h, uint32_t pmix_h, uint32_t freqs_l, uint32_t pmix_l) {
        const bool hi = li < 8;
        const uint32_t fr = hi ? freqs_h : freqs_l;
        const uint32_t pm = hi ? pmix_h : pmix_l;
        weights_update(w, (int)(short)(fr & 0xffffu), (int)(short)(fr >> 16), (int)(short)pm);
    }
};

struct RowSel {
    uint32_t stride_row;   // row index inside the stream's table
    uint32_t cm_row;       // context-map row (mixing only)
    bool is_default;       // mm_opts == 2: code with a fresh default CDF / never blend the stride row
};

// codec/literal.rs:176-208.  All inputs are row-uniform.
// `ctxk` = context | row slot of (prev, class of prev_prev) << 8, as context_of() returns it
template <bool HIGH, int MM>
__device__ __forceinline__ RowSel select_rows(const LitGeometry& g, const uint8_t* lds_mix, uint32_t ctxk, uint64_t last8, uint32_t hi_nib) {
    const uint32_t ctx = ctxk & 0xffu;
    const uint32_t prev_byte = (uint32_t)(last8 >> 56);
    uint32_t mm_opts;
    if (MM >= 0) mm_opts = (uint32_t)MM;
    else mm_opts = lds_mix[ctx | (HIGH ? ((prev_byte >> 4) << 8) : ((hi_nib << 8) | 4096u))];
    const uint32_t fast_cm = (mm_opts != 3) ? 0xffu : 0u;
    const uint32_t mm = (mm_opts != 0 && mm_opts != 3) ? 0xffu : 0u;
    const uint32_t opt1 = (mm_opts == 1) ? 0xfu : 0u;
    uint32_t stride_offset = 0;
    if (mm_opts >= 4) { uint32_t x = mm_opts ^ 4u; stride_offset = (x < 7u ? x : 7u) << 3; }
    const uint32_t sb = (uint32_t)(last8 >> (56 - stride_offset)) & 0xffu;
    uint32_t b, c, width;
    if (HIGH) {
        b = sb & mm & ~opt1 & 0xffu; c = ctx; width = g.nctx;
        // LitGeometry::hs_classes: only the rows (prev, class of prev_prev) can reach exist, [slot of the class][prev] instead of [ctx][prev]
        if (MM == 4 && g.hs_classes) { c = ctxk >> 8; width = g.hs_classes; }
    }
    else { b = ((mm & sb) | (~mm & ctx)) & 0xffu; c = (hi_nib & fast_cm) | ((ctx & opt1) << 4); width = g.low_width; }
    const uint32_t t = (mm >> 7) ^ (opt1 >> 2);
    const uint32_t plane = t == 0 ? g.plane0 : (t == 1 ? g.plane1 : g.plane2);
    RowSel r;
    // rows are ordered [plane][c][b]: the byte-valued index is innermost, so the four rows of a 128-byte L2 line
    // belong to neighbouring byte values with the same nibble / context (neighbouring letters are hot together
    // in text), and the address needs no per-lane multiply
    r.stride_row = (HIGH ? 0u : g.low_base) + ((plane * width + c) << 8) + b;
    r.cm_row = g.cm_base + (HIGH ? ctx : g.nctx + hi_nib + 16u * ctx);
    r.is_default = mm_opts == 2;
    return r;
}

// LDS layout of a workgroup: [16 x row cache (data, then tags)] [context tables unless CTXC] [mixing_mask if MM < 0]
struct LdsView { uint8_t* base; const uint8_t* ctx; const uint8_t* mix; };

template <int MM, bool CTXC>
__device__ __forceinline__ LdsView load_config_to_lds(uint8_t* lds, const LitBatch& b) {
    LdsView v;
    v.base = lds;
    uint8_t* p = lds + b.cache_bytes_per_wg;
    v.ctx = p;
    if (!CTXC) {
        const uint32_t ctx_bytes = LIT_BLOB_CTXF + LIT_CTXF_BYTES * b.geom.n_btypes;
        const uint32_t* src = (const uint32_t*)(b.blob + LIT_BLOB_LUT1CLASS);
        for (uint32_t i = threadIdx.x; i < ctx_bytes / 4; i += blockDim.x) ((uint32_t*)p)[i] = src[i];
        p += ctx_bytes;
    }
    v.mix = p;
    if (MM < 0) {
        const uint32_t* src = (const uint32_t*)(b.blob + b.geom.mix_off);
        for (uint32_t i = threadIdx.x; i < 8192 / 4; i += blockDim.x) ((uint32_t*)p)[i] = src[i];
    }
    __syncthreads();
    return v;
}

// Context of the next byte: literal.rs:87-117 with lut0 / lut1 / context map fused on the host into
// LIT_BLOB_CTXF[block type][prev][lut1 class of prev_prev]; `ctab` = byte offset of the current block type's table,
// `k1` (that class) is carried over from the previous byte.  Where LitGeometry::hs_classes lays the high-nibble stride table out by
// class, the row slot of (prev, class) comes along in bits 8.. (the host put it into the table's free half; select_rows takes the pair apart).
template <bool CTXC>
__device__ __forceinline__ uint32_t context_of(const LitGeometry& g, const uint8_t* lds_ctx, uint32_t ctab, uint32_t prev, uint32_t k1) {
    if (CTXC) return (uint32_t)g.ctx_const;
    const uint32_t at = ctab + (prev << 3) + k1;
    uint32_t v = lds_ctx[at];
    if (g.hs_classes) v |= (uint32_t)lds_ctx[at + 4u] << 8;
    return v;
}

// Walks a stream's segment list (general streams: one segment per Literal command).  `left` = bytes of the current
// segment still to code; advance() is called when it reaches zero and installs the next segment's context.
struct SegCursor {
    const LitSegment* segs; uint32_t idx, end, left; uint32_t* status;
    __device__ __forceinline__ void advance(const LitGeometry& g, uint64_t& last8, uint32_t& ctab) {
        while (left == 0u && idx < end) {
            const u32x4 sg = *(const u32x4*)(segs + idx);
            ++idx;
            left = sg.x;
            last8 = ((uint64_t)sg.w << 32) | sg.z;
            uint32_t t = sg.y - g.bt_first;
            if (t >= g.n_btypes) {       // a block type the codec holds no context table for (divans_gpu_codec_set_block_types): stay
                t = g.n_btypes - 1u;     // inside the tables, but say so -- the stream would be coded under the wrong context map
                if (status) atomicOr(status, LIT_STATUS_BAD_SEGMENT);
            }
            ctab = LIT_BLOB_CTXF + t * LIT_CTXF_BYTES;
        }
    }
    __device__ __forceinline__ void start(const LitBatch& b, uint32_t s, uint64_t& last8, uint32_t& ctab) {
        segs = b.segs; idx = b.seg_begin[s]; end = b.seg_begin[s + 1]; left = 0u; status = b.status;
        advance(b.geom, last8, ctab);
    }
};

}  // namespace divans_hip
#endif
