/*
 * cdf.c -- ORACLE (test infrastructure): 16-symbol adaptive CDF arithmetic.
 * Restates /root/reference/src/probability/{interface,frequentist_cdf,opt_frequentist_cdf,numeric}.rs
 * and src/codec/weights.rs.  Parity: pinned by the reference's CDF/division/speed
 * unit tests (see divans_oracle.h header); compressed bytes unpinned.
 */
#include "divans_oracle.h"
#include <stdlib.h>
#include <string.h>

/* Rust `wrapping_add` / `wrapping_sub` on i16. */
static inline int16_t wadd16(int16_t a, int16_t b) { return (int16_t)(uint16_t)((uint16_t)a + (uint16_t)b); }
static inline int16_t wsub16(int16_t a, int16_t b) { return (int16_t)(uint16_t)((uint16_t)a - (uint16_t)b); }
static inline int32_t wmul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t wadd32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline uint32_t clz32(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 32u; }
static inline uint32_t clz64(uint64_t x) { return x ? (uint32_t)__builtin_clzll(x) : 64u; }

/* frequentist_cdf.rs:17-23 */
void orc_cdf_default(orc_cdf16 *c) {
    for (int i = 0; i < 16; ++i) c->cdf[i] = (orc_prob)(4 * (i + 1));
}

/* frequentist_cdf.rs:74-85: count the symbol, then renormalise by 3/4 (with a +i+1 bias
 * that keeps every bin non-empty) once the running total reaches the speed's limit. */
void orc_cdf_blend(orc_cdf16 *c, uint8_t sym, orc_speed sp) {
    for (int i = sym; i < 16; ++i) c->cdf[i] = wadd16(c->cdf[i], sp.inc);
    if (c->cdf[15] >= sp.lim) {
        for (int i = 0; i < 16; ++i) {
            int16_t t = wadd16(c->cdf[i], (int16_t)(i + 1));
            c->cdf[i] = wsub16(t, (int16_t)(t >> 2)); /* arithmetic shift on i16 */
        }
    }
}

/* frequentist_cdf.rs:58-72 */
void orc_cdf_average(const orc_cdf16 *self, const orc_cdf16 *other, int32_t mix_rate, orc_cdf16 *out) {
    int32_t ourmax = self->cdf[15];
    int32_t othermax = other->cdf[15];
    int32_t prod = wmul32(ourmax, othermax);
    uint32_t lz = clz32((uint32_t)prod);
    if (lz > 17) lz = 17;
    uint32_t desired_shift = 17 - lz;
    int32_t inv_mix_rate = (1 << ORC_BLEND_FIXED_POINT_PRECISION) - mix_rate;
    for (int i = 0; i < 16; ++i) {
        int32_t rescaled_self = wmul32(self->cdf[i], othermax) >> desired_shift;
        int32_t rescaled_other = wmul32(other->cdf[i], ourmax) >> desired_shift;
        int32_t v = wadd32(wadd32(wmul32(rescaled_self, mix_rate), wmul32(rescaled_other, inv_mix_rate)), 1);
        out->cdf[i] = (orc_prob)(v >> ORC_BLEND_FIXED_POINT_PRECISION); /* `as Prob` truncation */
    }
}

/* numeric.rs:16-19 compute_divisor: what div_lut.rs::RECIPROCAL tabulates (make_div_lut.rs:28-40) */
static void compute_divisor(int16_t d, int64_t *inv, uint8_t *bitlen_m1) {
    uint16_t du = (uint16_t)d;
    uint8_t bit_len = (uint8_t)(16 - (du ? (uint32_t)__builtin_clz((uint32_t)du) - 16u : 16u));
    *inv = ((((int64_t)1 << bit_len) - (int64_t)d) << 31) / (int64_t)d + 1;
    *bitlen_m1 = (uint8_t)(bit_len - 1);
}

/* numeric.rs:25-31 */
int32_t orc_fast_divide_30bit_by_16bit(int32_t num, int16_t denom) {
    int64_t inv; uint8_t sh;
    compute_divisor(denom, &inv, &sh);
    int64_t idiv_mul_num = inv * (int64_t)num;
    int32_t hi = (int32_t)(idiv_mul_num >> 31);
    return (hi + (((int32_t)((int64_t)num - (idiv_mul_num >> 31))) >> 1)) >> sh;
}

/* numeric.rs:50-62 (RECIPROCAL8[d] == compute_divisor8(d) for d>=1, RECIPROCAL8[0] == 0) */
int16_t orc_fast_divide_16bit_by_8bit(uint16_t num, uint8_t denom) {
    int32_t inv = denom ? 1 + (1 << 24) / (int32_t)denom : 0;
    return (int16_t)(((int64_t)inv * (int64_t)num) >> 24);
}

/* BaseCDF::div_by_max.  The default build's OptFrequentistCDF16 uses the reciprocal LUT
 * (opt_frequentist_cdf.rs:99-102) which its generator proves equal to integer division
 * (make_div_lut.rs:37-39); plain FrequentistCDF16 uses `/` (frequentist_cdf.rs:36-38). */
static inline int32_t div_by_max(int32_t num, int16_t max) { return num / (int32_t)max; }

/* probability/interface.rs:97-108 */
int orc_cdf_sym_to_start_and_freq(const orc_cdf16 *c, uint8_t sym, orc_sym_start_freq *out) {
    int16_t max = c->cdf[15];
    if (max == 0) return -1;
    int32_t cdf_sym = div_by_max((int32_t)c->cdf[sym & 0xf] << ORC_LOG2_SCALE, max);
    int32_t cdf_prev = sym ? div_by_max((int32_t)c->cdf[(sym - 1) & 0xf] << ORC_LOG2_SCALE, max) : 0;
    int32_t freq = cdf_sym - cdf_prev;
    out->start = wadd16((int16_t)cdf_prev, 1);  /* "major hax" */
    out->freq = wsub16((int16_t)freq, 1);
    out->sym = sym;
    return 0;
}

/* probability/interface.rs:136-198: the first i<15 with rescaled < cdf[i], else 15 */
int orc_cdf_offset_to_sym_start_and_freq(const orc_cdf16 *c, orc_prob cdf_offset, orc_sym_start_freq *out) {
    int16_t max = c->cdf[15];
    int16_t rescaled = (int16_t)(((int32_t)cdf_offset * (int32_t)max) >> ORC_LOG2_SCALE);
    uint8_t sym = 15;
    for (int i = 0; i < 15; ++i) {
        if (rescaled < c->cdf[i]) { sym = (uint8_t)i; break; }
    }
    return orc_cdf_sym_to_start_and_freq(c, sym, out);
}

/* probability/interface.rs:566-575 */
uint8_t orc_speed_to_u8(int16_t data) {
    uint16_t du = (uint16_t)data;
    uint8_t length = (uint8_t)(16 - (du ? (uint32_t)__builtin_clz((uint32_t)du) - 16u : 16u));
    uint8_t mantissa = 0;
    if (data != 0) {
        int16_t rem = (int16_t)(data - (int16_t)(1 << (length - 1)));
        mantissa = (uint8_t)((int16_t)(rem << 3) >> (length - 1));
    }
    return (uint8_t)((length << 3) | mantissa);
}

/* probability/interface.rs:577-585 */
int16_t orc_u8_to_speed(uint8_t data) {
    if (data < 8) return 0;
    unsigned log_val = ((unsigned)(data >> 3) - 1u) & 15u; /* > 15 only in damaged streams: release-build i16 shifts take the amount mod 16 */
    int16_t rem = (int16_t)(uint16_t)(((unsigned)data & 0x7u) << log_val);
    return (int16_t)((int16_t)(uint16_t)(1u << log_val) | (rem >> 3));
}

/* probability/interface.rs:303-320 */
orc_speed orc_speed_palette(int index) {
    static const orc_speed pal[15] = {
        {0, 1024}, {2, 1024}, {1, 128}, {1, 16384}, {2, 2048}, {4, 1024}, {8, 8192}, {16, 48},
        {16, 8192}, {32, 4096}, {64, 16384}, {128, 256}, {128, 16384}, {512, 16384}, {1664, 16384}};
    if (index < 0) index = 0;
    if (index > 14) index = 14;
    return pal[index];
}

/* ---------------- Weights, src/codec/weights.rs ---------------- */
void orc_weights_init(orc_weights *w) {  /* weights.rs:15-21 */
    w->model_weights[0] = w->model_weights[1] = 1;
    w->mixing_param = 1;
    w->normalized_weight = (int16_t)(1 << (ORC_BLEND_FIXED_POINT_PRECISION - 1));
}

/* weights.rs:54-62 */
static int16_t compute_normalized_weight(const int32_t mw[2]) {
    int64_t total = (int64_t)mw[0] + (int64_t)mw[1];
    int16_t lz = (int16_t)clz64((uint64_t)total);
    int16_t shift = (int16_t)(56 - lz);
    if (shift < 0) shift = 0;
    int64_t total_8bit = total >> shift;
    uint16_t num = (uint16_t)((uint16_t)(mw[0] >> shift) << 8);
    int16_t q = orc_fast_divide_16bit_by_8bit(num, (uint8_t)total_8bit);
    return (int16_t)(uint16_t)((uint16_t)q << (ORC_BLEND_FIXED_POINT_PRECISION - 8)); /* i16 `<<` drops high bits */
}

/* weights.rs:64-80 */
static void normalize_weights(int32_t w[2]) {
    if (((w[0] | w[1]) & 0x7f000000) != 0) {
        uint32_t lz0 = clz32((uint32_t)w[0]), lz1 = clz32((uint32_t)w[1]);
        uint32_t ilog = 32 - (lz0 < lz1 ? lz0 : lz1);
        const uint32_t max_log = 24;
        if (ilog >= max_log) {
            w[0] >>= (ilog - max_log);
            w[1] >>= (ilog - max_log);
        }
    }
}

/* weights.rs:110-133 (integer variant; all i64 wrapping) */
static int32_t compute_new_weight(const orc_prob probs[2], orc_prob weighted_prob, const int32_t weights[2], int index) {
    uint64_t p1 = (uint64_t)(int64_t)weighted_prob;
    uint64_t total = (uint64_t)1 << ORC_LOG2_SCALE;
    uint64_t p0 = total - p1;
    uint64_t n1i = (uint64_t)(int64_t)probs[index];
    uint64_t ni = (uint64_t)1 << ORC_LOG2_SCALE;
    uint64_t error = total - p1;
    int64_t wi = (int64_t)weights[index];
    uint64_t efficacy = total * n1i - p1 * ni;
    uint32_t lg = 64 - clz64(p1 * p0);
    int64_t adj = (int64_t)(error * efficacy) >> (lg & 63); /* release-mode shift masks the amount */
    int32_t nw = (int32_t)(uint32_t)((uint64_t)wi + (uint64_t)adj);
    return nw > 1 ? nw : 1;
}

/* weights.rs:23-38 */
void orc_weights_update(orc_weights *w, const orc_prob model_probs[2], orc_prob weighted_prob) {
    normalize_weights(w->model_weights);
    int32_t w0 = compute_new_weight(model_probs, weighted_prob, w->model_weights, 0);
    int32_t w1 = compute_new_weight(model_probs, weighted_prob, w->model_weights, 1);
    w->model_weights[0] = w0;
    w->model_weights[1] = w1;
    w->normalized_weight = compute_normalized_weight(w->model_weights);
}
