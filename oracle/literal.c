/*
 * literal.c -- ORACLE (test infrastructure): the per-byte literal coder.
 * Restates /root/reference/src/codec/literal.rs:87-394, codec/interface.rs:125-340,
 * codec/priors.rs:35-47, priors.rs:76-259 and constants.rs.
 * Compressed bytes unpinned (see divans_oracle.h).
 */
#define _POSIX_C_SOURCE 200809L
#include "divans_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* LiteralNibblePriors = (CombinedNibble, 3, 256, NUM_BLOCK_TYPES=256), codec/priors.rs:35-37.
 * linearize_index!(priors.rs:211-224): i0 + d0*(i1 + d1*i2). */
#define NIBBLE_PRIORS (3 * 256 * 256)
static inline uint32_t nibble_prior_index(uint32_t t, uint32_t b, uint32_t c) { return t + 3u * (b + 256u * c); }
/* LiteralCommandPriorsCM = FirstNibble(1,256) then SecondNibble(1,16,256), codec/priors.rs:45-47 */
#define CM_PRIORS (256 + 16 * 256)
static inline uint32_t cm_first_index(uint32_t ctx) { return ctx; }
static inline uint32_t cm_second_index(uint32_t hi, uint32_t ctx) { return 256u + hi + 16u * ctx; }

typedef struct {
    orc_cdf16 *rows;
    uint32_t n_rows;
    uint32_t *touched;   /* rows that left the default state (cheap re-init between streams) */
    uint8_t *is_touched;
    uint32_t n_touched;
} prior_table;

static void table_init(prior_table *t, uint32_t n) {
    t->rows = (orc_cdf16 *)malloc(sizeof(orc_cdf16) * n);
    t->touched = (uint32_t *)malloc(sizeof(uint32_t) * n);
    t->is_touched = (uint8_t *)calloc(n, 1);
    t->n_rows = n; t->n_touched = 0;
    for (uint32_t i = 0; i < n; ++i) orc_cdf_default(&t->rows[i]); /* ffi/alloc_util.rs:77-79 default-init */
}
static void table_free(prior_table *t) { free(t->rows); free(t->touched); free(t->is_touched); }
static void table_reset(prior_table *t) {
    for (uint32_t k = 0; k < t->n_touched; ++k) {
        uint32_t i = t->touched[k];
        orc_cdf_default(&t->rows[i]);
        t->is_touched[i] = 0;
    }
    t->n_touched = 0;
}
static inline orc_cdf16 *table_row_mut(prior_table *t, uint32_t i) {
    if (!t->is_touched[i]) { t->is_touched[i] = 1; t->touched[t->n_touched++] = i; }
    return &t->rows[i];
}

struct orc_lit_state {
    /* LiteralBookKeeping, codec/interface.rs:125-140 */
    uint64_t last_8_literals;
    uint8_t literal_context_map[ORC_MAX_LITERAL_CONTEXT_MAP_SIZE];
    uint8_t btype_last;
    orc_speed literal_adaptation[4];
    uint8_t literal_lut0[256], literal_lut1[256];
    uint8_t mixing_mask[ORC_NUM_MIXING_VALUES];
    orc_weights model_weights[2];
    int mixing_priors;              /* CodecTraits::MIXING_PRIORS, specializations.rs:27-47 */
    prior_table lit_high_priors, lit_low_priors, lit_cm_priors;
    int16_t *trace;                 /* optional (sym,start,freq) dump */
    size_t trace_pos;
};

/* constants.rs UTF8_CONTEXT_LOOKUP / SIGNED_3_BIT_CONTEXT_LOOKUP are the RFC 7932 section 7.1
 * context tables; generated here from their structure and checked against the reference's
 * arrays by tests/golden/context_luts.bin. */
static uint8_t utf8_lut0(int b) {
    if (b < 32) return (b == 9 || b == 10 || b == 13) ? 4 : 0;
    if (b < 64) {
        static const uint8_t punct[32] = {8, 12, 16, 12, 12, 20, 12, 16, 24, 28, 12, 12, 32, 12, 36, 12,
                                          44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 32, 32, 24, 40, 28, 12};
        return punct[b - 32];
    }
    if (b < 128) {
        int lower = b >= 96;
        int c = b & 31; /* position in the alphabet row */
        if (c == 0) return 12;                        /* '@' '`' */
        if (c <= 26) {
            int vowel = (c == 1 || c == 5 || c == 9 || c == 15 || c == 21);
            return (uint8_t)((lower ? 56 : 48) + (vowel ? 0 : 4));
        }
        if (c == 27) return 24;                       /* '[' '{' */
        if (c == 29) return 28;                       /* ']' '}' */
        if (c == 31) return lower ? 0 : 12;           /* DEL vs '_' */
        return 12;                                    /* '\\' '|' '^' '~' */
    }
    if (b < 192) return (uint8_t)(b & 1);
    return (uint8_t)(2 + (b & 1));
}
static uint8_t utf8_lut1(int b) {
    if (b <= 32 || b == 127) return 0;
    if (b < 128) {
        if (b >= '0' && b <= '9') return 2;
        if (b >= 'A' && b <= 'Z') return 2;
        if (b >= 'a' && b <= 'z') return 3;
        return 1;
    }
    return b >= 224 ? 2 : 0;
}
static uint8_t signed_3bit(int b) {
    if (b == 0) return 0;
    if (b < 16) return 1;
    if (b < 64) return 2;
    if (b < 128) return 3;
    if (b < 192) return 4;
    if (b < 240) return 5;
    if (b < 255) return 6;
    return 7;
}

/* codec/interface.rs:199-222 */
void orc_get_lut0(uint8_t mode, uint8_t out[256]) {
    for (int i = 0; i < 256; ++i) {
        switch (mode) {
        case 3: out[i] = (uint8_t)(signed_3bit(i) << 3); break;
        case 2: out[i] = utf8_lut0(i); break;
        case 1: out[i] = (uint8_t)(i >> 2); break;
        default: out[i] = (uint8_t)(i & 0x3f); break;
        }
    }
}
/* codec/interface.rs:223-238 */
void orc_get_lut1(uint8_t mode, uint8_t out[256]) {
    for (int i = 0; i < 256; ++i) {
        switch (mode) {
        case 3: out[i] = signed_3bit(i); break;
        case 2: out[i] = utf8_lut1(i); break;
        default: out[i] = 0; break;
        }
    }
}

static const orc_speed SPEED_MUD = {0x10, 0x2000}; /* probability/interface.rs:323, default_literal_speed */

/* TestSimple, bin/benchmark.rs:195-206: use_context_map=false => no context-map entry is ever
 * coded (context_map.rs:269-276) and every mixing value is 4 (context_map.rs:386-387); lsb6;
 * no literal-adaptation override => MUD. */
void orc_lit_config_simple(orc_lit_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    memset(cfg->mixing_mask, 4, sizeof(cfg->mixing_mask));
    cfg->prediction_mode = 0;
    cfg->btype = 0;
    cfg->context_mixing = 0;
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = SPEED_MUD;
}

/* TestContextMixing through bench_no_ir, bin/benchmark.rs:156-167,305-343: 256-entry context map
 * cm[i]=i&63, utf8, BlockSwitchLiteral(1,2), mixing values 4, dynamic_context_mixing=2. */
void orc_lit_config_context_mixing(orc_lit_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    for (int i = 0; i < 256; ++i) cfg->literal_context_map[i] = (uint8_t)(i & 63);
    memset(cfg->mixing_mask, 4, sizeof(cfg->mixing_mask));
    cfg->prediction_mode = 2;
    cfg->btype = 1;
    cfg->context_mixing = 2;
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = SPEED_MUD;
}

/* obs_prediction_mode_context_map + obs_literal_block_switch, codec/interface.rs:285-327 */
void orc_lit_state_reconfigure(orc_lit_state *s, const orc_lit_config *cfg) {
    memcpy(s->literal_context_map, cfg->literal_context_map, sizeof(s->literal_context_map));
    memcpy(s->mixing_mask, cfg->mixing_mask, sizeof(s->mixing_mask));
    orc_get_lut0(cfg->prediction_mode, s->literal_lut0);
    orc_get_lut1(cfg->prediction_mode, s->literal_lut1);
    for (int i = 0; i < 4; ++i) s->literal_adaptation[i] = cfg->literal_adaptation[i];
    s->btype_last = cfg->btype;
    s->model_weights[0].mixing_param = cfg->context_mixing; /* obs_dynamic_context_mixing :320-327 */
    s->model_weights[1].mixing_param = cfg->context_mixing;
    s->mixing_priors = cfg->context_mixing > 1;             /* should_mix, weights.rs:44-46 */
}

static void lit_state_reset(orc_lit_state *s, const orc_lit_config *cfg) {
    s->last_8_literals = 0;
    orc_weights_init(&s->model_weights[0]);
    orc_weights_init(&s->model_weights[1]);
    table_reset(&s->lit_high_priors);
    table_reset(&s->lit_low_priors);
    table_reset(&s->lit_cm_priors);
    orc_lit_state_reconfigure(s, cfg);
    s->trace = NULL; s->trace_pos = 0;
}

orc_lit_state *orc_lit_state_new(const orc_lit_config *cfg) {
    orc_lit_state *s = (orc_lit_state *)calloc(1, sizeof(*s));
    table_init(&s->lit_high_priors, NIBBLE_PRIORS);
    table_init(&s->lit_low_priors, NIBBLE_PRIORS);
    table_init(&s->lit_cm_priors, CM_PRIORS);
    lit_state_reset(s, cfg);
    return s;
}

void orc_lit_state_free(orc_lit_state *s) {
    if (!s) return;
    table_free(&s->lit_high_priors); table_free(&s->lit_low_priors); table_free(&s->lit_cm_priors);
    free(s);
}

void orc_lit_set_last8(orc_lit_state *s, uint64_t last8) { s->last_8_literals = last8; }
uint64_t orc_lit_get_last8(const orc_lit_state *s) { return s->last_8_literals; }

typedef struct { uint64_t stride_bytes; uint8_t actual_context, prev_byte; } byte_context;

/* literal.rs:87-117 */
static inline byte_context get_prev_word_context(const orc_lit_state *s) {
    uint8_t prev_byte = (uint8_t)(s->last_8_literals >> 0x38);
    uint8_t prev_prev_byte = (uint8_t)(s->last_8_literals >> 0x30);
    uint8_t selected_context = s->literal_lut0[prev_byte] | s->literal_lut1[prev_prev_byte];
    uint32_t cmap_index = (uint32_t)selected_context + ((uint32_t)s->btype_last << 6);
    byte_context bc;
    bc.actual_context = s->literal_context_map[cmap_index];
    bc.stride_bytes = s->last_8_literals;
    bc.prev_byte = prev_byte;
    return bc;
}

/* literal.rs:154-259.  `enc`/`dec`: exactly one is non-NULL (get_or_put_nibble).
 * Returns the coded nibble; *blendable receives the stride row to blend later (or NULL). */
static inline uint8_t code_nibble(orc_lit_state *s, int is_high, uint8_t cur_nibble, byte_context bc,
                                  uint8_t cur_byte_prior, orc_ans_encoder *enc, orc_ans_decoder *dec,
                                  orc_cdf16 **blendable) {
    uint32_t mixing_mask_index = bc.actual_context;
    if (!is_high) {
        mixing_mask_index |= (uint32_t)(cur_byte_prior & 0xf) << 8;
        mixing_mask_index |= 4096;
    } else {
        mixing_mask_index |= ((uint32_t)bc.prev_byte >> 4) << 8;
    }
    uint8_t mm_opts = s->mixing_mask[mixing_mask_index];
    uint8_t fast_cm_prior_mask = (mm_opts != 3) ? 0xff : 0;
    uint8_t mm = (mm_opts != 0 && mm_opts != 3) ? 0xff : 0;
    uint8_t opt_1_f_mask = (mm_opts == 1) ? 0xf : 0;
    uint32_t stride_offset = 0;
    if (mm_opts >= 4) {
        uint32_t x = (uint32_t)mm_opts ^ 4u;
        stride_offset = (x < 7 ? x : 7) << 3;
    }
    uint8_t stride_selected_byte = (uint8_t)(bc.stride_bytes >> (0x38 - stride_offset));
    uint32_t index_b, index_c;
    if (is_high) {
        index_b = (uint8_t)(stride_selected_byte & mm & (uint8_t)~opt_1_f_mask);
        index_c = bc.actual_context;
    } else {
        index_b = (uint8_t)((mm & stride_selected_byte) | ((uint8_t)~mm & bc.actual_context));
        index_c = (uint8_t)((cur_byte_prior & fast_cm_prior_mask) | ((bc.actual_context & opt_1_f_mask) << 4));
    }
    uint32_t index_a = (uint32_t)((mm >> 7) ^ (opt_1_f_mask >> 2));
    prior_table *tbl = is_high ? &s->lit_high_priors : &s->lit_low_priors;
    orc_cdf16 *nibble_prob = table_row_mut(tbl, nibble_prior_index(index_a, index_b, index_c));
    orc_sym_start_freq coded;
    if (s->mixing_priors) {
        uint32_t cmi = is_high ? cm_first_index(bc.actual_context) : cm_second_index(cur_byte_prior, bc.actual_context);
        orc_cdf16 *cm_prob = table_row_mut(&s->lit_cm_priors, cmi);
        orc_cdf16 prob;
        int32_t mix = (int32_t)(uint16_t)s->model_weights[is_high].normalized_weight; /* `as u16 as i32` */
        orc_cdf_average(cm_prob, nibble_prob, mix, &prob);
        if (enc) orc_ans_put_nibble(enc, cur_nibble, &prob, &coded);
        else cur_nibble = orc_ans_get_nibble(dec, &prob, &coded);
        orc_sym_start_freq a, b;
        orc_prob model_probs[2] = {1, 1};
        if (orc_cdf_sym_to_start_and_freq(cm_prob, cur_nibble, &a) == 0) model_probs[0] = a.freq;
        if (orc_cdf_sym_to_start_and_freq(nibble_prob, cur_nibble, &b) == 0) model_probs[1] = b.freq;
        orc_weights_update(&s->model_weights[is_high], model_probs, coded.freq);
        orc_cdf_blend(cm_prob, cur_nibble, s->literal_adaptation[2 | is_high]);
    } else {
        orc_cdf16 immutable_prior;
        const orc_cdf16 *coder_prior = nibble_prob;
        if (mm_opts == 2) { orc_cdf_default(&immutable_prior); coder_prior = &immutable_prior; }
        if (enc) orc_ans_put_nibble(enc, cur_nibble, coder_prior, &coded);
        else cur_nibble = orc_ans_get_nibble(dec, coder_prior, &coded);
    }
    if (s->trace) {
        s->trace[s->trace_pos++] = cur_nibble;
        s->trace[s->trace_pos++] = coded.start;
        s->trace[s->trace_pos++] = coded.freq;
    }
    *blendable = (mm_opts == 2) ? NULL : nibble_prob;
    return cur_nibble;
}

/* literal.rs:261-394 (the three NibbleArrayCallSite variants differ only in resumability) */
static void code_nibble_array(orc_lit_state *s, orc_ans_encoder *enc, orc_ans_decoder *dec,
                              const uint8_t *in, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint8_t byte_to_encode_val = enc ? in[i] : 0;
        byte_context bc = get_prev_word_context(s);
        orc_cdf16 *prob = NULL;
        uint8_t h_nibble = code_nibble(s, 1, (uint8_t)(byte_to_encode_val >> 4), bc, 0, enc, dec, &prob);
        if (prob) orc_cdf_blend(prob, h_nibble, s->literal_adaptation[0]);     /* :318-322 */
        uint8_t l_nibble = code_nibble(s, 0, (uint8_t)(byte_to_encode_val & 0xf), bc, h_nibble, enc, dec, &prob);
        uint8_t cur_byte = (uint8_t)(l_nibble | (h_nibble << 4));
        s->last_8_literals >>= 8;                                              /* push_literal_byte :280-284 */
        s->last_8_literals |= (uint64_t)cur_byte << 0x38;
        if (out) out[i] = cur_byte;
        if (prob) orc_cdf_blend(prob, l_nibble, s->literal_adaptation[0]);     /* :352-356, index 0 again */
    }
}

void orc_lit_encode_bytes(orc_lit_state *s, orc_ans_encoder *enc, const uint8_t *in, size_t n) {
    code_nibble_array(s, enc, NULL, in, NULL, n);
}
void orc_lit_decode_bytes(orc_lit_state *s, orc_ans_decoder *dec, uint8_t *out, size_t n) {
    code_nibble_array(s, NULL, dec, NULL, out, n);
}

static size_t stream_encode_with(orc_lit_state *s, const orc_lit_config *cfg, const uint8_t *in, size_t n,
                                 uint8_t *out, size_t cap, int16_t *trace) {
    lit_state_reset(s, cfg);
    s->trace = trace;
    orc_ans_encoder enc;
    orc_ans_encoder_init(&enc);
    orc_lit_encode_bytes(s, &enc, in, n);
    orc_ans_flush_chunk(&enc); /* close(), arithmetic_coder.rs:251-254 */
    size_t ret = enc.out.len;
    if (enc.failed || ret > cap) ret = (size_t)-1;
    else memcpy(out, enc.out.data, ret);
    orc_ans_encoder_free(&enc);
    s->trace = NULL;
    return ret;
}

static int stream_decode_with(orc_lit_state *s, const orc_lit_config *cfg, const uint8_t *in, size_t in_len,
                              uint8_t *out, size_t n) {
    lit_state_reset(s, cfg);
    orc_ans_decoder dec;
    orc_ans_decoder_init(&dec, in, in_len);
    orc_lit_decode_bytes(s, &dec, out, n);
    return dec.starved ? -1 : 0;
}

size_t orc_lit_stream_encode(const orc_lit_config *cfg, const uint8_t *in, size_t n, uint8_t *out, size_t cap) {
    orc_lit_state *s = orc_lit_state_new(cfg);
    size_t r = stream_encode_with(s, cfg, in, n, out, cap, NULL);
    orc_lit_state_free(s);
    return r;
}

size_t orc_lit_stream_encode_trace(const orc_lit_config *cfg, const uint8_t *in, size_t n,
                                   uint8_t *out, size_t cap, int16_t *trace) {
    orc_lit_state *s = orc_lit_state_new(cfg);
    size_t r = stream_encode_with(s, cfg, in, n, out, cap, trace);
    orc_lit_state_free(s);
    return r;
}

int orc_lit_stream_decode(const orc_lit_config *cfg, const uint8_t *in, size_t in_len, uint8_t *out, size_t n) {
    orc_lit_state *s = orc_lit_state_new(cfg);
    int r = stream_decode_with(s, cfg, in, in_len, out, n);
    orc_lit_state_free(s);
    return r;
}

/* ---- CPU baseline: every stream independent, one worker thread per core ---- */
typedef struct {
    const orc_lit_config *cfg;
    const uint8_t *in;
    size_t n_streams, stream_len;
    int tid, nthreads;
    double enc_s, dec_s;
    uint64_t coded;
    int bad;
} worker_arg;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *p) {
    worker_arg *a = (worker_arg *)p;
    orc_lit_state *s = orc_lit_state_new(a->cfg);
    size_t cap = a->stream_len * 2 + 64;
    uint8_t *coded = (uint8_t *)malloc(cap);
    uint8_t *back = (uint8_t *)malloc(a->stream_len ? a->stream_len : 1);
    for (size_t i = (size_t)a->tid; i < a->n_streams; i += (size_t)a->nthreads) {
        const uint8_t *src = a->in + i * a->stream_len;
        double t0 = now_s();
        size_t c = stream_encode_with(s, a->cfg, src, a->stream_len, coded, cap, NULL);
        double t1 = now_s();
        if (c == (size_t)-1) { a->bad = 1; continue; }
        int r = stream_decode_with(s, a->cfg, coded, c, back, a->stream_len);
        double t2 = now_s();
        a->enc_s += t1 - t0; a->dec_s += t2 - t1; a->coded += c;
        if (r != 0 || memcmp(back, src, a->stream_len) != 0) a->bad = 1;
    }
    free(coded); free(back);
    orc_lit_state_free(s);
    return NULL;
}

int orc_lit_batch_roundtrip(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len,
                            int nthreads, double *enc_seconds, double *dec_seconds, uint64_t *coded_bytes) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    worker_arg *args = (worker_arg *)calloc((size_t)nthreads, sizeof(worker_arg));
    for (int t = 0; t < nthreads; ++t) {
        args[t].cfg = cfg; args[t].in = in; args[t].n_streams = n_streams; args[t].stream_len = stream_len;
        args[t].tid = t; args[t].nthreads = nthreads;
        pthread_create(&th[t], NULL, worker, &args[t]);
    }
    double enc = 0, dec = 0; uint64_t coded = 0; int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        /* wall time of the slowest worker approximates elapsed time per direction */
        if (args[t].enc_s > enc) enc = args[t].enc_s;
        if (args[t].dec_s > dec) dec = args[t].dec_s;
        coded += args[t].coded; bad |= args[t].bad;
    }
    if (enc_seconds) *enc_seconds = enc;
    if (dec_seconds) *dec_seconds = dec;
    if (coded_bytes) *coded_bytes = coded;
    free(th); free(args);
    return bad ? -1 : 0;
}


/* ---- CPU baseline with nothing but coding inside the timed regions ---- */
typedef struct {
    const orc_lit_config *cfg;
    const uint8_t *in;
    size_t n_streams, stream_len;
    int tid, nthreads;
    pthread_barrier_t *bar;
    uint8_t *coded; size_t slot; size_t *coded_len;   /* shared output slots, one per stream */
    double enc_begin, enc_end, dec_begin, dec_end;
    uint64_t coded_total;
    int bad;
} bench_arg;

static void *bench_worker(void *p) {
    bench_arg *a = (bench_arg *)p;
    orc_lit_state *s = orc_lit_state_new(a->cfg);
    orc_ans_encoder enc;
    orc_ans_encoder_init(&enc);
    enc.words_scratch = (uint32_t *)malloc(sizeof(uint32_t) * ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH);
    enc.out.data = (uint8_t *)malloc(a->slot); enc.out.cap = a->slot;
    uint8_t *back = (uint8_t *)malloc(a->stream_len ? a->stream_len : 1);
    memset(enc.out.data, 0, a->slot); memset(back, 0, a->stream_len);      /* fault the pages in */
    pthread_barrier_wait(a->bar);
    a->enc_begin = now_s();
    for (size_t i = (size_t)a->tid; i < a->n_streams; i += (size_t)a->nthreads) {
        lit_state_reset(s, a->cfg);
        orc_ans_encoder_reset(&enc);
        orc_lit_encode_bytes(s, &enc, a->in + i * a->stream_len, a->stream_len);
        orc_ans_flush_chunk(&enc);
        if (enc.failed || enc.out.len > a->slot) { a->bad = 1; a->coded_len[i] = 0; continue; }
        memcpy(a->coded + i * a->slot, enc.out.data, enc.out.len);
        a->coded_len[i] = enc.out.len; a->coded_total += enc.out.len;
    }
    a->enc_end = now_s();
    pthread_barrier_wait(a->bar);
    a->dec_begin = now_s();
    for (size_t i = (size_t)a->tid; i < a->n_streams; i += (size_t)a->nthreads) {
        lit_state_reset(s, a->cfg);
        orc_ans_decoder dec;
        orc_ans_decoder_init(&dec, a->coded + i * a->slot, a->coded_len[i]);
        orc_lit_decode_bytes(s, &dec, back, a->stream_len);
        if (dec.starved || memcmp(back, a->in + i * a->stream_len, a->stream_len) != 0) a->bad = 1;
    }
    a->dec_end = now_s();
    orc_ans_encoder_free(&enc); free(back);
    orc_lit_state_free(s);
    return NULL;
}

int orc_lit_batch_bench(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len,
                        int nthreads, double *enc_wall, double *dec_wall, uint64_t *coded_bytes) {
    if (nthreads < 1) nthreads = 1;
    const size_t slot = stream_len * 2 + 64;
    uint8_t *coded = (uint8_t *)malloc(slot * (n_streams ? n_streams : 1));
    size_t *coded_len = (size_t *)calloc(n_streams ? n_streams : 1, sizeof(size_t));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    bench_arg *args = (bench_arg *)calloc((size_t)nthreads, sizeof(bench_arg));
    pthread_barrier_t bar;
    if (!coded || !coded_len || !th || !args || pthread_barrier_init(&bar, NULL, (unsigned)nthreads) != 0) return -1;
    memset(coded, 0, slot * (n_streams ? n_streams : 1));
    for (int t = 0; t < nthreads; ++t) {
        args[t].cfg = cfg; args[t].in = in; args[t].n_streams = n_streams; args[t].stream_len = stream_len;
        args[t].tid = t; args[t].nthreads = nthreads; args[t].bar = &bar;
        args[t].coded = coded; args[t].slot = slot; args[t].coded_len = coded_len;
        pthread_create(&th[t], NULL, bench_worker, &args[t]);
    }
    double eb = 1e300, ee = 0, db = 1e300, de = 0; uint64_t total = 0; int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (args[t].enc_begin < eb) eb = args[t].enc_begin;
        if (args[t].enc_end > ee) ee = args[t].enc_end;
        if (args[t].dec_begin < db) db = args[t].dec_begin;
        if (args[t].dec_end > de) de = args[t].dec_end;
        total += args[t].coded_total; bad |= args[t].bad;
    }
    if (enc_wall) *enc_wall = ee - eb;
    if (dec_wall) *dec_wall = de - db;
    if (coded_bytes) *coded_bytes = total;
    pthread_barrier_destroy(&bar);
    free(coded); free(coded_len); free(th); free(args);
    return bad ? -1 : 0;
}


/* ---- bench / test checker: encode many independent streams on nthreads workers and compare each with the coded bytes
 * another implementation produced (coded[off[i] .. off[i] + size[i])).  Returns the number of streams that differ, -1 on
 * allocation failure.  first_bad receives the index of the first differing stream (or n_streams). */
typedef struct {
    const orc_lit_config *cfg; const uint8_t *in; size_t n_streams, stream_len; int tid, nthreads;
    const uint8_t *coded; const uint64_t *off; const uint32_t *size;
    size_t bad, first_bad;
} check_arg;

static void *check_worker(void *p) {
    check_arg *a = (check_arg *)p;
    orc_lit_state *s = orc_lit_state_new(a->cfg);
    orc_ans_encoder enc;
    orc_ans_encoder_init(&enc);
    a->first_bad = a->n_streams;
    for (size_t i = (size_t)a->tid; i < a->n_streams; i += (size_t)a->nthreads) {
        lit_state_reset(s, a->cfg);
        orc_ans_encoder_reset(&enc);
        orc_lit_encode_bytes(s, &enc, a->in + i * a->stream_len, a->stream_len);
        orc_ans_flush_chunk(&enc);
        if (enc.failed || enc.out.len != a->size[i] || memcmp(enc.out.data, a->coded + a->off[i], enc.out.len) != 0) {
            a->bad += 1;
            if (i < a->first_bad) a->first_bad = i;
        }
    }
    orc_ans_encoder_free(&enc);
    orc_lit_state_free(s);
    return NULL;
}

long orc_lit_batch_check(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len, int nthreads,
                         const uint8_t *coded, const uint64_t *off, const uint32_t *size, size_t *first_bad) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    check_arg *args = (check_arg *)calloc((size_t)nthreads, sizeof(check_arg));
    if (!th || !args) return -1;
    for (int t = 0; t < nthreads; ++t) {
        args[t].cfg = cfg; args[t].in = in; args[t].n_streams = n_streams; args[t].stream_len = stream_len;
        args[t].tid = t; args[t].nthreads = nthreads; args[t].coded = coded; args[t].off = off; args[t].size = size;
        pthread_create(&th[t], NULL, check_worker, &args[t]);
    }
    long bad = 0; size_t fb = n_streams;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        bad += (long)args[t].bad;
        if (args[t].first_bad < fb) fb = args[t].first_bad;
    }
    if (first_bad) *first_bad = fb;
    free(th); free(args);
    return bad;
}


/* ---- general streams: the literals of a stream with Copy / Dict commands in between (codec/mod.rs:711-792).
 * One segment per Literal command: the block type in force (obs_literal_block_switch, codec/interface.rs:289-292) and
 * last_8_literals as reloaded from the ring buffer after the previous command (codec/mod.rs:771-783).  Priors, Weights
 * and the LIT coder persist across segments. */
size_t orc_lit_segments_encode(const orc_lit_config *cfg, const uint8_t *lit, size_t n, const uint32_t *seg_len,
                               const uint32_t *seg_btype, const uint64_t *seg_last8, size_t nseg, uint8_t *out, size_t cap) {
    orc_lit_config *c = (orc_lit_config *)malloc(sizeof(*c));
    memcpy(c, cfg, sizeof(*c));
    orc_lit_state *s = orc_lit_state_new(c);
    orc_ans_encoder enc;
    orc_ans_encoder_init(&enc);
    size_t pos = 0;
    for (size_t k = 0; k < nseg && pos + seg_len[k] <= n; ++k) {
        if (c->btype != (uint8_t)seg_btype[k]) { c->btype = (uint8_t)seg_btype[k]; orc_lit_state_reconfigure(s, c); }
        orc_lit_set_last8(s, seg_last8[k]);
        orc_lit_encode_bytes(s, &enc, lit + pos, seg_len[k]);
        pos += seg_len[k];
    }
    orc_ans_flush_chunk(&enc);
    size_t ret = enc.out.len;
    if (enc.failed || ret > cap || pos != n) ret = (size_t)-1;
    else memcpy(out, enc.out.data, ret);
    orc_ans_encoder_free(&enc);
    orc_lit_state_free(s); free(c);
    return ret;
}

int orc_lit_segments_decode(const orc_lit_config *cfg, const uint8_t *in, size_t in_len, const uint32_t *seg_len,
                            const uint32_t *seg_btype, const uint64_t *seg_last8, size_t nseg, uint8_t *out, size_t n) {
    orc_lit_config *c = (orc_lit_config *)malloc(sizeof(*c));
    memcpy(c, cfg, sizeof(*c));
    orc_lit_state *s = orc_lit_state_new(c);
    orc_ans_decoder dec;
    orc_ans_decoder_init(&dec, in, in_len);
    size_t pos = 0;
    for (size_t k = 0; k < nseg && pos + seg_len[k] <= n; ++k) {
        if (c->btype != (uint8_t)seg_btype[k]) { c->btype = (uint8_t)seg_btype[k]; orc_lit_state_reconfigure(s, c); }
        orc_lit_set_last8(s, seg_last8[k]);
        orc_lit_decode_bytes(s, &dec, out + pos, seg_len[k]);
        pos += seg_len[k];
    }
    int rc = (dec.starved || pos != n) ? -1 : 0;
    orc_lit_state_free(s); free(c);
    return rc;
}
