/*
 * ans.c -- ORACLE (test infrastructure): 2-way interleaved 64-bit rANS with 32-bit
 * renormalisation, LIFO-encoded in chunks of 65 536 symbols.
 * Restates /root/reference/src/ans.rs.  Compressed bytes unpinned (see divans_oracle.h).
 */
#include "divans_oracle.h"
#include <stdlib.h>
#include <string.h>

#define NORMALIZATION_INTERVAL ((uint64_t)1 << 31)       /* ans.rs:134 */
#define ENC_START_STATE NORMALIZATION_INTERVAL           /* ans.rs:135 */
#define SCALE_MASK (((uint64_t)1 << ORC_LOG2_SCALE) - 1) /* ans.rs:139 */

void orc_bytes_free(orc_bytes *b) { free(b->data); b->data = NULL; b->len = b->cap = 0; }

static void bytes_reserve(orc_bytes *b, size_t extra) {
    if (b->len + extra <= b->cap) return;
    size_t ncap = b->cap ? b->cap * 2 : 4096;
    while (ncap < b->len + extra) ncap *= 2;
    b->data = (uint8_t *)realloc(b->data, ncap);
    b->cap = ncap;
}

void orc_ans_encoder_init(orc_ans_encoder *e) {
    memset(e, 0, sizeof(*e));
    e->start = (uint16_t *)malloc(sizeof(uint16_t) * ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH);
    e->freq = (uint16_t *)malloc(sizeof(uint16_t) * ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH);
}

void orc_ans_encoder_free(orc_ans_encoder *e) {
    free(e->start); free(e->freq); free(e->words_scratch); orc_bytes_free(&e->out);
    memset(e, 0, sizeof(*e));
}

/* start another stream with the buffers of the previous one (the CPU-baseline workers allocate once, outside the timed region) */
void orc_ans_encoder_reset(orc_ans_encoder *e) {
    e->n_pending = 0; e->out.len = 0; e->failed = 0;
}

/* ans.rs:331-378 flush_chunk + ans.rs:302-329 reverse_put_sym.
 * The reference pushes bytes onto a downward-growing stack; what the decoder reads
 * forward is [state_a LE][state_b LE][renorm words, last-emitted first].  We build the
 * word list in emission order and then write it reversed. */
void orc_ans_flush_chunk(orc_ans_encoder *e) {
    uint32_t len = e->n_pending;
    if (len == 0) return;
    if (!e->words_scratch) e->words_scratch = (uint32_t *)malloc(sizeof(uint32_t) * ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH);
    uint32_t *words = e->words_scratch;  /* at most one word per symbol */
    size_t nwords = 0;
    uint64_t state_a = ENC_START_STATE, state_b = ENC_START_STATE;
    for (uint32_t k = 0; k < len; ++k) {        /* newest symbol first */
        uint32_t idx = len - 1 - k;
        int16_t start = (int16_t)e->start[idx];
        int16_t freq = (int16_t)e->freq[idx];
        if (freq <= 0 || start < 0) { e->failed = 1; freq = freq ? freq : 1; }
        /* `freq as u64` sign-extends in Rust */
        uint64_t f = (uint64_t)(int64_t)freq;
        uint64_t rescale_lim = ((NORMALIZATION_INTERVAL >> ORC_LOG2_SCALE) << 32) * f;
        uint64_t state = state_a;
        if (state >= rescale_lim) {
            words[nwords++] = (uint32_t)state;  /* low 32 bits, read back little-endian */
            state >>= 32;
        }
        uint64_t x = ((state / f) << ORC_LOG2_SCALE) + (state % f) + (uint64_t)(int64_t)start;
        state_a = state_b;
        state_b = x;
    }
    { uint64_t t = state_a; state_a = state_b; state_b = t; } /* unconditional swap, ans.rs:354-356 */
    bytes_reserve(&e->out, 16 + 4 * nwords);
    uint8_t *p = e->out.data + e->out.len;
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(state_a >> (8 * i));
    for (int i = 0; i < 8; ++i) p[8 + i] = (uint8_t)(state_b >> (8 * i));
    p += 16;
    for (size_t w = nwords; w-- > 0;) {
        uint32_t v = words[w];
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        p += 4;
    }
    e->out.len += 16 + 4 * nwords;
    e->n_pending = 0;
}

/* ans.rs:287-301 */
void orc_ans_put_start_freq(orc_ans_encoder *e, orc_prob start, orc_prob freq) {
    e->start[e->n_pending] = (uint16_t)start;
    e->freq[e->n_pending] = (uint16_t)freq;
    e->n_pending++;
    if (e->n_pending == ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH) orc_ans_flush_chunk(e);
}

/* ans.rs:279-286 */
void orc_ans_put_nibble(orc_ans_encoder *e, uint8_t sym, const orc_cdf16 *cdf, orc_sym_start_freq *coded) {
    orc_sym_start_freq sf;
    if (orc_cdf_sym_to_start_and_freq(cdf, sym, &sf) != 0) { e->failed = 1; sf.start = 1; sf.freq = 1; sf.sym = sym; }
    orc_ans_put_start_freq(e, sf.start, sf.freq);
    if (coded) *coded = sf;
}

/* ans.rs:152-162 */
void orc_ans_decoder_init(orc_ans_decoder *d, const uint8_t *in, size_t in_len) {
    memset(d, 0, sizeof(*d));
    d->buffer_a_bytes_required = 8; /* "this will load both buffers" */
    d->in = in; d->in_len = in_len;
}

static uint64_t le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* The byte-at-a-time resumable paths of push_data (ans.rs:173-223) exist only to survive
 * arbitrary caller buffer splits; with the whole stream in memory the decoder always takes
 * either the 16-byte state load (:177-186) or the 4-byte refill (:432-440). */
static void decoder_fill(orc_ans_decoder *d) {
    uint8_t req = d->buffer_a_bytes_required;
    if (req == 0) return;
    if (req == 1) {
        if (d->in_len - d->in_pos < 4) { d->starved = 1; return; }
        const uint8_t *p = d->in + d->in_pos;
        d->state_a <<= 32;
        d->state_a |= (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
        d->in_pos += 4;
        d->buffer_a_bytes_required = 0;
        return;
    }
    /* 4 < req < 16: start of a chunk */
    if (d->in_len - d->in_pos < 16) { d->starved = 1; return; }
    d->sym_count = 0;
    d->state_a = le64(d->in + d->in_pos);
    d->state_b = le64(d->in + d->in_pos + 8);
    d->in_pos += 16;
    d->buffer_a_bytes_required = 0;
}

/* get_nibble_internal ans.rs:246-252 + helper_advance_sym :230-244 */
uint8_t orc_ans_get_nibble(orc_ans_decoder *d, const orc_cdf16 *cdf, orc_sym_start_freq *coded) {
    decoder_fill(d);
    orc_prob cdf_offset = (orc_prob)(d->state_a & SCALE_MASK);
    orc_sym_start_freq sf;
    if (orc_cdf_offset_to_sym_start_and_freq(cdf, cdf_offset, &sf) != 0) { d->starved = 1; sf.start = 1; sf.freq = 1; sf.sym = 0; }
    d->buffer_a_bytes_required = d->buffer_b_bytes_required;
    d->buffer_a_bytes_required |= (uint8_t)((d->sym_count == (uint16_t)(ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH - 1)) << 3);
    uint64_t x = (uint64_t)(int64_t)sf.freq * (d->state_a >> ORC_LOG2_SCALE) + (d->state_a & SCALE_MASK) - (uint64_t)(int64_t)sf.start;
    d->sym_count = (uint16_t)(d->sym_count + 1);
    d->buffer_b_bytes_required = (uint8_t)(x < NORMALIZATION_INTERVAL);
    d->state_a = d->state_b;
    d->state_b = x;
    if (coded) *coded = sf;
    return sf.sym;
}
