/*
 * stream.c -- ORACLE (test infrastructure): a complete literal-only .divans stream.
 * Restates, for streams made of PredictionMode / BlockSwitchLiteral / Literal commands:
 *   src/codec/mod.rs:143-158,409-560,652-792 (command type nibble, flush, checksum trailer)
 *   src/codec/context_map.rs:105-428        (PredictionMode wire format)
 *   src/codec/block_type.rs:31-194          (BlockSwitchLiteral)
 *   src/codec/literal.rs:496-661            (literal length)
 *   src/mux.rs                              (two-stream interleaving)
 *   src/divans_compressor.rs:126-174,276-426 + src/raw_to_cmd/mod.rs:105-181 (internal compressor)
 *   src/divans_decompressor.rs:38-52        (header)
 * Compressed bytes unpinned (see divans_oracle.h); pinned: mux KAT, CRC KATs, round trips and the
 * reference's compressed-size bounds.
 */
#include "divans_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Mux, src/mux.rs */
#define MUX_MAX_HEADER_SIZE 3
#define MUX_MAX_FLUSH_VARIANCE 131073

typedef struct { uint8_t *buf; size_t cap, start, end; } mux_stream; /* AllocatedMemoryRange */
typedef struct {
    mux_stream s[2];
    uint32_t cur_stream_bytes_avail;
    uint8_t cur_stream;
    size_t last_flush[2];
    size_t bytes_flushed;
    int eof;                 /* 0 Running, 1 EofStart, 2 EofMid, 3 EofDone */
} orc_mux;

static void mux_free(orc_mux *m) { free(m->s[0].buf); free(m->s[1].buf); memset(m, 0, sizeof(*m)); }

/* mux.rs:285-329 prep_push_for_n_bytes */
static void mux_prep_push(orc_mux *m, int id, size_t data_len) {
    mux_stream *b = &m->s[id];
    if (b->cap - b->end >= data_len) return;
    size_t have = b->end - b->start;
    if (b->cap >= have + data_len + MUX_MAX_HEADER_SIZE &&
        (b->start == b->end || (b->start >= 16384 && b->start > have + MUX_MAX_HEADER_SIZE))) {
        memmove(b->buf + MUX_MAX_HEADER_SIZE, b->buf + b->start, have);
        b->end = MUX_MAX_HEADER_SIZE + have;
        b->start = MUX_MAX_HEADER_SIZE;
        return;
    }
    uint64_t desired = (uint64_t)(MUX_MAX_HEADER_SIZE + data_len + have);
    uint32_t log_desired = (uint32_t)(64 - __builtin_clzll(desired)) + 1;
    if (log_desired < 9) log_desired = 9;
    size_t ncap = (size_t)1 << log_desired;
    uint8_t *nb = (uint8_t *)calloc(ncap, 1);
    if (have) memcpy(nb + MUX_MAX_HEADER_SIZE, b->buf + b->start, have);
    free(b->buf);
    b->buf = nb; b->cap = ncap;
    b->end = MUX_MAX_HEADER_SIZE + have;
    b->start = MUX_MAX_HEADER_SIZE;
}

static void mux_push_data(orc_mux *m, int id, const uint8_t *data, size_t n) { /* mux.rs:278-281 */
    mux_prep_push(m, id, n);
    memcpy(m->s[id].buf + m->s[id].end, data, n);
    m->s[id].end += n;
}

static size_t mux_chunk_size(size_t last_flushed, int lagging) { /* mux.rs:37-48 */
    if (lagging) return 16;
    if (last_flushed <= 1024) return 4096;
    if (last_flushed <= 65536) return 16384;
    return 65536;
}

/* mux.rs:55-78 get_code: returns header length, fills hdr, *count = payload bytes */
static int mux_get_code(int id, size_t bytes_to_write, int lagging, uint8_t hdr[3], size_t *count) {
    if (!lagging || bytes_to_write == 4096 || bytes_to_write == 16384 || bytes_to_write >= 65536) {
        if (bytes_to_write < 4096) return mux_get_code(id, bytes_to_write, 1, hdr, count);
        if (bytes_to_write < 16384) { hdr[0] = (uint8_t)(id | (1 << 4)); *count = 4096; return 1; }
        if (bytes_to_write < 65536) { hdr[0] = (uint8_t)(id | (2 << 4)); *count = 16384; return 1; }
        hdr[0] = (uint8_t)(id | (3 << 4)); *count = 65536; return 1;
    }
    hdr[0] = (uint8_t)id;
    hdr[1] = (uint8_t)((bytes_to_write - 1) & 0xff);
    hdr[2] = (uint8_t)(((bytes_to_write - 1) >> 8) & 0xff);
    *count = bytes_to_write;
    return 3;
}

static size_t mux_serialize_leftover(orc_mux *m, uint8_t *out, size_t out_len) { /* mux.rs:331-338 */
    size_t n = m->cur_stream_bytes_avail < out_len ? m->cur_stream_bytes_avail : out_len;
    mux_stream *b = &m->s[m->cur_stream];
    memcpy(out, b->buf + b->start, n);
    b->start += n;
    m->cur_stream_bytes_avail -= (uint32_t)n;
    return n;
}

static void mux_serialize_stream_id(orc_mux *m, int id, uint8_t *out, size_t out_len, size_t *off, int lagging) { /* mux.rs:339-382 */
    mux_stream *b = &m->s[id];
    uint8_t hdr[3]; size_t count;
    int hl = mux_get_code(id, b->end - b->start, lagging, hdr, &count);
    m->bytes_flushed += count;
    size_t should_write = count + (size_t)hl;
    b->start -= (size_t)hl;
    memcpy(b->buf + b->start, hdr, (size_t)hl);
    m->last_flush[id] = m->bytes_flushed;
    size_t room = out_len - *off;
    size_t to_write = should_write < room ? should_write : room;
    memcpy(out + *off, b->buf + b->start, to_write);
    b->start += to_write;
    if (b->start == b->end) { b->start = MUX_MAX_HEADER_SIZE; b->end = b->start; }
    *off += to_write;
    if (to_write != should_write) {
        m->cur_stream_bytes_avail = (uint32_t)(should_write - to_write);
        m->cur_stream = (uint8_t)id;
    }
}

static size_t mux_serialize(orc_mux *m, uint8_t *out, size_t out_len) { /* mux.rs:445-476 */
    size_t off = 0;
    if (m->cur_stream_bytes_avail != 0) off += mux_serialize_leftover(m, out, out_len);
    while (off < out_len) {
        int flushed_any = 0;
        size_t lo = m->last_flush[0], hi = m->last_flush[0];
        if (m->last_flush[1] < lo) lo = m->last_flush[1];
        if (m->last_flush[1] > hi) hi = m->last_flush[1];
        for (int i = 0; i < 2; ++i) {
            int lagging = hi > MUX_MAX_FLUSH_VARIANCE + m->last_flush[i];
            if (m->s[i].end - m->s[i].start >= mux_chunk_size(m->last_flush[i], lagging) &&
                m->last_flush[i] <= lo + MUX_MAX_FLUSH_VARIANCE) {
                flushed_any = 1;
                mux_serialize_stream_id(m, i, out, out_len, &off, lagging);
                if (m->cur_stream_bytes_avail != 0) break;
            }
        }
        if (!flushed_any) break;
    }
    return off;
}

static size_t mux_flush_internal(orc_mux *m, uint8_t *out, size_t out_len) { /* mux.rs:519-561 */
    size_t off = 0;
    if (m->cur_stream_bytes_avail != 0) off += mux_serialize_leftover(m, out, out_len);
    while (off < out_len) {
        int flushed_any = 0, have_lf = 0;
        size_t last_flush = 0;
        for (int i = 0; i < 2; ++i) {
            int nonempty = m->s[i].start != m->s[i].end;
            if (!have_lf ? nonempty : (m->last_flush[i] < last_flush && nonempty)) { last_flush = m->last_flush[i]; have_lf = 1; }
        }
        for (int i = 0; i < 2; ++i) {
            if (!have_lf || m->last_flush[i] <= last_flush + MUX_MAX_FLUSH_VARIANCE) {
                size_t written = off;
                if (m->s[i].start != m->s[i].end) mux_serialize_stream_id(m, i, out, out_len, &written, 1);
                if (written != off) flushed_any = 1;
                off = written;
                if (m->cur_stream_bytes_avail != 0) break;
            }
        }
        if (!flushed_any) break;
    }
    return off;
}

static const uint8_t MUX_EOF_MARKER[3] = {0xff, 0xfe, 0xff}; /* mux.rs:54 */

static size_t mux_serialize_close(orc_mux *m, uint8_t *out, size_t out_len) { /* mux.rs:477-518 */
    if (m->eof == 3) return 0;
    size_t ret = mux_flush_internal(m, out, out_len);
    for (int st = 0; st < 3; ++st) {
        if (ret == out_len) return ret;
        if (m->eof == st) { out[ret++] = MUX_EOF_MARKER[st]; m->eof = st + 1; }
    }
    return ret;
}

/* mux.rs:384-444 deserialize (iterative; the reference recurses on the tail) */
typedef struct { int pending_kind; /* 0 none, 1 some, 2 header0, 3 header1 */ int stream; uint32_t count; uint8_t lsb; } mux_parse;
static size_t mux_deserialize(orc_mux *m, mux_parse *p, const uint8_t *in, size_t n) {
    size_t pos = 0;
    while (pos < n && m->eof != 3) {
        uint8_t c = in[pos];
        switch (p->pending_kind) {
        case 2: p->lsb = c; p->pending_kind = 3; pos++; break;
        case 3: p->count = ((uint32_t)p->lsb | ((uint32_t)c << 8)) + 1; p->pending_kind = 1; pos++; break;
        case 1: {
            size_t take = p->count < n - pos ? p->count : n - pos;
            mux_push_data(m, p->stream, in + pos, take);
            pos += take; p->count -= (uint32_t)take;
            if (p->count == 0) p->pending_kind = 0;
            break;
        }
        default:
            if (c == 0xff || c == 0xfe) {
                if (c == 0xff || m->eof != 0) {
                    /* deserialize_eof :384-  consumes marker bytes in order, then returns */
                    int progressed = 0;
                    while (pos < n && m->eof < 3 && in[pos] == MUX_EOF_MARKER[m->eof]) { m->eof++; pos++; progressed = 1; }
                    if (!progressed) return pos;
                    if (m->eof == 3) return pos;
                    if (pos < n) return pos; /* malformed marker: reference stops consuming */
                    break;
                }
            }
            p->stream = c & 1; /* STREAM_ID_MASK for NUM_STREAMS = 2 */
            if (c < 16) { p->pending_kind = 2; pos++; }
            else { p->count = (uint32_t)1024 << ((c >> 4) << 1); p->pending_kind = 1; pos++; }
            break;
        }
    }
    return pos;
}

/* exported for the mux KAT (src/test_mux.rs:1192-1207) */
int orc_mux_demux(const uint8_t *in, size_t n, uint8_t *s0, size_t *n0, uint8_t *s1, size_t *n1, size_t *consumed) {
    orc_mux m; mux_parse p;
    memset(&m, 0, sizeof(m)); memset(&p, 0, sizeof(p));
    size_t used = mux_deserialize(&m, &p, in, n);
    size_t a = m.s[0].end - m.s[0].start, b = m.s[1].end - m.s[1].start;
    if (a > *n0 || b > *n1) { mux_free(&m); return -1; }
    if (a) memcpy(s0, m.s[0].buf + m.s[0].start, a);
    if (b) memcpy(s1, m.s[1].buf + m.s[1].start, b);
    *n0 = a; *n1 = b; if (consumed) *consumed = used;
    int done = m.eof == 3;
    mux_free(&m);
    return done ? 0 : 1;
}

/* ------------------------------------------------------------------ output sink = the caller's buffers */
typedef struct {
    uint8_t *data; size_t len, cap;    /* everything emitted so far */
    size_t call_buf;                   /* size of the output buffer the caller passes per call */
    size_t call_used;                  /* bytes of the current call's buffer already filled */
} sink;

static void sink_reserve(sink *s, size_t extra) {
    if (s->len + extra <= s->cap) return;
    size_t nc = s->cap ? s->cap * 2 : 65536;
    while (nc < s->len + extra) nc *= 2;
    s->data = (uint8_t *)realloc(s->data, nc); s->cap = nc;
}
/* remaining room in the current call's buffer.  A full buffer does NOT by itself end the call: the codec only goes back to its
 * caller where a drain cannot finish or the Mux / header / trailer need room (sink_new_call at exactly those points); until then
 * it keeps coding and the Mux keeps the bytes (drain_or_fill_static_buffer, codec/interface.rs:868-896). */
static size_t sink_room(sink *s) { return s->call_buf - s->call_used; }
static void sink_new_call(sink *s) { s->call_used = 0; }
static uint8_t *sink_ptr(sink *s, size_t want) { sink_reserve(s, want); return s->data + s->len; }
static void sink_commit(sink *s, size_t n) { s->len += n; s->call_used += n; }

/* ------------------------------------------------------------------ CMD-coder side state */
enum { SPEED_MUD_I = 0x10, SPEED_MUD_L = 0x2000 };
static const orc_speed SP_MED = {0x30, 0x4000}, SP_FAST = {0x60, 0x4000}, SP_PLANE = {0x80, 0x4000},
                       SP_ROCKET = {0x180, 0x4000}, SP_SLOW = {0x20, 0x1000}, SP_MUD = {0x10, 0x2000};

#define CONTEXT_MAP_CACHE_SIZE 13
typedef struct {
    /* CrossCommandBookKeeping, codec/interface.rs:142-167, :369-404 */
    orc_cdf16 cc_priors[16];           /* CrossCommandPriors FullSelection(16,1) (EndIndicator unused here) */
    orc_cdf16 lit_len_priors[4 * 256 + 256 * 15]; /* CountSmall(256,16) SizeBegNib SizeLastNib SizeMantissaNib */
    orc_cdf16 prediction_priors[31];   /* codec/priors.rs:125-133 */
    orc_cdf16 btype_priors[10];        /* Mnemonic(3) FirstNibble(3) SecondNibble(3) StrideNibble(1) */
    uint8_t cmap_lru[CONTEXT_MAP_CACHE_SIZE];
    uint8_t distance_context_map[4 * 256];
    uint8_t btype_lru[3][2], btype_max_seen[3];
    uint8_t last_4_states;
    int wire;                          /* ORC_WIRE_HEAD / ORC_WIRE_WASM_EXAMPLE (divans_oracle.h) */
    /* options */
    uint8_t desired_prior_depth, desired_context_mixing, desired_force_stride;
    int desired_do_context_map, has_desired_adaptation;
    orc_speed desired_literal_adaptation[4];
    /* PredictionModeState scratch that persists across PredictionMode commands, context_map.rs:84-94 */
    uint8_t pm_literal_context_map[ORC_MAX_LITERAL_CONTEXT_MAP_SIZE];
    uint8_t pm_mixing[ORC_NUM_MIXING_VALUES];
    uint8_t pm_distance_map[4 * 256];
} cmd_state;

/* prior offsets, codec/priors.rs (define_prior_struct! linearisation, priors.rs:211-259) */
#define LL_COUNT_SMALL(ctype) (ctype)
#define LL_SIZE_BEG(ctype) (4096 + (ctype))
#define LL_SIZE_LAST(ctype) (4096 + 256 + (ctype))
#define LL_SIZE_MANT(ctype) (4096 + 512 + (ctype))
#define PM_ONLY 0
#define PM_FIRST(t) (2 + (t))
#define PM_SECOND(t) (4 + (t))
#define PM_MNEMONIC(t) (6 + (t))
#define PM_MIXING(p) (10 + (p))
#define PM_SPEED(p) (27 + (p))
#define PM_ALIAS_LAST 27 /* DynamicContextMixingSpeed / PriorDepth are not in the struct: fall through to the last entry */
#define BT_MNEMONIC(i) (i)
#define BT_FIRST(i) (3 + (i))
#define BT_SECOND(i) (6 + (i))
#define BT_STRIDE 9

static void cmd_state_init(cmd_state *c, const orc_stream_options *o) {
    memset(c, 0, sizeof(*c));
    for (size_t i = 0; i < sizeof(c->cc_priors) / sizeof(orc_cdf16); ++i) orc_cdf_default(&c->cc_priors[i]);
    for (size_t i = 0; i < sizeof(c->lit_len_priors) / sizeof(orc_cdf16); ++i) orc_cdf_default(&c->lit_len_priors[i]);
    for (size_t i = 0; i < 31; ++i) orc_cdf_default(&c->prediction_priors[i]);
    for (size_t i = 0; i < 10; ++i) orc_cdf_default(&c->btype_priors[i]);
    for (int i = 0; i < 3; ++i) { c->btype_lru[i][0] = 0; c->btype_lru[i][1] = 1; }
    c->last_4_states = 3 << 4;
    uint8_t mixing = o->dynamic_context_mixing;
    if (o->force_stride != 0 && mixing == 0 && o->use_context_map) mixing = 1; /* codec/interface.rs:360-365 */
    c->desired_context_mixing = mixing;
    c->desired_prior_depth = o->prior_depth;
    c->desired_force_stride = o->force_stride;
    c->desired_do_context_map = o->use_context_map;
    c->has_desired_adaptation = o->has_literal_adaptation;
    for (int i = 0; i < 4; ++i) c->desired_literal_adaptation[i] = o->literal_adaptation[i];
}

/* one "get_or_put_nibble + blend" on the CMD coder */
typedef struct { orc_ans_encoder *enc; orc_ans_decoder *dec; uint32_t nibbles; } cmd_coder;
static uint8_t cmd_nibble(cmd_coder *cc, uint8_t nib, orc_cdf16 *prior, orc_speed sp) {
    if (cc->enc) orc_ans_put_nibble(cc->enc, nib, prior, NULL);
    else nib = orc_ans_get_nibble(cc->dec, prior, NULL);
    ++cc->nibbles;
    orc_cdf_blend(prior, nib, sp);
    return nib;
}

/* codec/interface.rs:421-455 obs_context_map_for_lru */
static int obs_context_map_for_lru(cmd_state *c, int type, uint32_t index, uint8_t val) {
    int found = -1;
    for (int i = 0; i < CONTEXT_MAP_CACHE_SIZE; ++i) if (c->cmap_lru[i] == val) { found = i; break; }
    if (found > 0) memmove(c->cmap_lru + 1, c->cmap_lru, (size_t)found);
    else if (found < 0) memmove(c->cmap_lru + 1, c->cmap_lru, CONTEXT_MAP_CACHE_SIZE - 1);
    c->cmap_lru[0] = val;
    if (type == 1) {
        if (index >= sizeof(c->distance_context_map)) return -1;
        c->distance_context_map[index] = val;
    }
    return 0;
}

static uint8_t lru_max_plus_one(const cmd_state *c) {
    uint8_t m = 0;
    for (int i = 0; i < CONTEXT_MAP_CACHE_SIZE; ++i) if (c->cmap_lru[i] > m) m = c->cmap_lru[i];
    return (uint8_t)(m + 1);
}

/* PredictionModeState::encode_or_decode, codec/context_map.rs:105-428.
 * `pm` is the encoder's input command (NULL when decoding); on return `out` holds what the codec's own
 * PredictionModeContextMap contains, i.e. what obs_prediction_mode_context_map will copy. */
static int code_prediction_mode(cmd_state *c, cmd_coder *cc, void (*drain)(void *), void *drain_ctx,
                                const orc_prediction_mode *pm, orc_prediction_mode_result *out) {
    orc_speed desired[4] = {SP_MUD, SP_MUD, SP_MUD, SP_MUD};
    if (pm && pm->has_context_speeds) {
        /* context_map.rs:124-145 */
        for (int i = 0; i < 2; ++i) {
            if (pm->context_map_speed_f8[i][0] || pm->context_map_speed_f8[i][1]) {
                desired[2 + i].inc = orc_u8_to_speed(pm->context_map_speed_f8[i][0]);
                desired[2 + i].lim = orc_u8_to_speed(pm->context_map_speed_f8[i][1]);
            }
            const uint8_t *st = c->desired_context_mixing != 0 ? pm->combined_stride_speed_f8[i] : pm->stride_speed_f8[i];
            if (st[0] || st[1]) { desired[i].inc = orc_u8_to_speed(st[0]); desired[i].lim = orc_u8_to_speed(st[1]); }
        }
    }
    if (c->has_desired_adaptation) for (int i = 0; i < 4; ++i) desired[i] = c->desired_literal_adaptation[i];

    /* Begin :159-178 */
    drain(drain_ctx);
    for (int i = 0; i < CONTEXT_MAP_CACHE_SIZE; ++i) c->cmap_lru[i] = (uint8_t)i;
    for (size_t i = 0; i < sizeof(c->distance_context_map); ++i) c->distance_context_map[i] = (uint8_t)(i & 3);
    uint8_t mode = cmd_nibble(cc, pm ? pm->prediction_mode : 0, &c->prediction_priors[PM_ONLY], SP_MED);
    if (mode > 3) return -1;
    out->prediction_mode = mode;
    /* DynamicContextMixing :179-198 */
    drain(drain_ctx);
    uint8_t is_adv = pm ? pm->is_adv_context_map : 0;
    if (is_adv >> 1) return -1;
    uint8_t mixnib = cmd_nibble(cc, (uint8_t)(c->desired_context_mixing | (is_adv << 3)), &c->prediction_priors[PM_ALIAS_LAST], SP_MED);
    out->mixing_math = mixnib & 3;
    int combine = mixnib != 0;
    /* PriorDepth :199-211 */
    drain(drain_ctx);
    (void)cmd_nibble(cc, c->desired_prior_depth, &c->prediction_priors[PM_ALIAS_LAST], SP_FAST);
    /* AdaptationSpeed :212-252 */
    uint8_t out_f8[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (uint32_t index = 0; index < 16; ++index) {
        drain(drain_ctx);
        uint32_t si = index >> 2, pt = index & 3;
        uint8_t f0 = orc_speed_to_u8(desired[si].inc), f1 = orc_speed_to_u8(desired[si].lim);
        uint8_t nib = pt == 0 ? (uint8_t)((f0 & 0x7f) >> 3) : pt == 1 ? (uint8_t)(f0 & 7) : pt == 2 ? (uint8_t)((f1 & 0x7f) >> 3) : (uint8_t)(f1 & 7);
        nib = cmd_nibble(cc, nib, &c->prediction_priors[PM_SPEED(pt)], SP_FAST);
        if (pt == 0) out_f8[si][0] |= (uint8_t)(nib << 3);
        if (pt == 1) out_f8[si][0] |= nib;
        if (pt == 2) out_f8[si][1] |= (uint8_t)(nib << 3);
        if (pt == 3) out_f8[si][1] |= nib;
    }
    /* set_stride_context_speed / set_context_map_speed store speed_to_u8(u8_to_speed(f8)); readers apply u8_to_speed */
    for (int i = 0; i < 4; ++i) {
        out->literal_adaptation[i].inc = orc_u8_to_speed(orc_speed_to_u8(orc_u8_to_speed(out_f8[i][0])));
        out->literal_adaptation[i].lim = orc_u8_to_speed(orc_speed_to_u8(orc_u8_to_speed(out_f8[i][1])));
    }
    /* context maps :253-384 */
    for (int type = 0; type < 2; ++type) {
        const uint8_t *cur = NULL; size_t cur_len = 0;
        if (pm) {
            if (type == 0) { cur = pm->literal_context_map; cur_len = pm->n_literal_context_map; }
            else if (pm->has_context_speeds) { cur = pm->distance_context_map; cur_len = pm->n_distance_context_map; }
        }
        if (!c->desired_do_context_map) cur_len = 0;
        uint8_t *outmap = type == 0 ? c->pm_literal_context_map : c->pm_distance_map;
        size_t outmap_len = type == 0 ? sizeof(c->pm_literal_context_map) : sizeof(c->pm_distance_map);
        for (uint32_t index = 0;; ++index) {
            drain(drain_ctx);
            uint8_t mn = 14;
            if (pm && index < cur_len) {
                uint8_t target = cur[index];
                mn = 15;
                for (int i = 0; i < CONTEXT_MAP_CACHE_SIZE; ++i) if (c->cmap_lru[i] == target) mn = (uint8_t)i;
                if (target == lru_max_plus_one(c)) mn = 13;
            }
            /* HEAD lists Mnemonic in PredictionModePriors (codec/priors.rs:125-133) -> offset 6 + type.  The build that wrote
             * wasm/wasm.html's example coded the mnemonics of BOTH maps under row 27, the row DynamicContextMixingSpeed / PriorDepth
             * alias (priors.rs:226-237): its two mnemonic nibbles decode to "end of map" only there (slot 32507 is symbol 14 with
             * freq 220 under that row, symbol 15 under a fresh one), and only then do 8192 mixing values of 4 follow. */
            mn = cmd_nibble(cc, mn, &c->prediction_priors[c->wire == ORC_WIRE_WASM_EXAMPLE ? PM_ALIAS_LAST : PM_MNEMONIC(type)], SP_MED);
            if (mn == 14) {
                if (type == 0) for (int i = 0; i < CONTEXT_MAP_CACHE_SIZE; ++i) c->cmap_lru[i] = (uint8_t)i; /* :303-306 */
                break;
            }
            uint8_t val;
            if (mn == 15) {
                drain(drain_ctx);
                uint8_t msn = cmd_nibble(cc, (pm && index < cur_len) ? (uint8_t)(cur[index] >> 4) : 0, &c->prediction_priors[PM_FIRST(type)], SP_MED);
                drain(drain_ctx);
                uint8_t lsn = cmd_nibble(cc, (pm && index < cur_len) ? (uint8_t)(cur[index] & 0xf) : 0, &c->prediction_priors[PM_SECOND(type)], SP_MED);
                val = (uint8_t)((msn << 4) | lsn);
                if (index >= outmap_len) return -1;
                outmap[index] = val;
                if (obs_context_map_for_lru(c, type, index, val)) return -1;
            } else {
                val = mn == 13 ? lru_max_plus_one(c) : c->cmap_lru[mn];
                if (obs_context_map_for_lru(c, type, index, val)) return -1;
                if (index >= outmap_len) return -1;
                outmap[index] = val;
            }
        }
    }
    /* MixingValues :385-422 */
    for (uint32_t index = 0; index < ORC_NUM_MIXING_VALUES; ++index) {
        drain(drain_ctx);
        uint8_t nib = !c->desired_do_context_map ? 4 : (!combine ? 0 : ((pm && pm->has_context_speeds && pm->mixing_values) ? pm->mixing_values[index] : 0));
        uint32_t prior = index >= 256 ? (uint32_t)(c->pm_mixing[index - 256] & 0xf) : 16; /* the codec's own pm always has_context_speeds */
        if (c->wire == ORC_WIRE_WASM_EXAMPLE) prior = 16; /* that build coded all 8192 values under PriorMixingValue[16]: with :396-400's row
                                                             `value 256 places back` the example's nibble 256 would be 6, with [16] it is 4 x 8192 */
        nib = cmd_nibble(cc, nib, &c->prediction_priors[PM_MIXING(prior)], SP_PLANE);
        c->pm_mixing[index] = nib;
    }
    out->literal_context_map = c->pm_literal_context_map;
    out->mixing_values = c->pm_mixing;
    return 0;
}

/* BlockTypeState + LiteralBlockTypeState, codec/block_type.rs:31-194 (switch index 0 = literal) */
static int code_block_switch_literal(cmd_state *c, cmd_coder *cc, void (*drain)(void *), void *drain_ctx,
                                     uint8_t in_btype, uint8_t in_stride, uint8_t *out_btype, uint8_t *out_stride) {
    const int idx = 0;
    uint8_t varint = in_btype == c->btype_lru[idx][1] ? 0 : in_btype == (uint8_t)(c->btype_max_seen[idx] + 1) ? 1 : in_btype <= 12 ? (uint8_t)(in_btype + 2) : 15;
    drain(drain_ctx);
    varint = cmd_nibble(cc, varint, &c->btype_priors[BT_MNEMONIC(idx)], SP_SLOW);
    uint8_t btype;
    if (varint == 0) btype = c->btype_lru[idx][1];
    else if (varint == 1) btype = (uint8_t)(c->btype_max_seen[idx] + 1);
    else if (varint == 15) {
        drain(drain_ctx);
        uint8_t first = cmd_nibble(cc, in_btype & 0xf, &c->btype_priors[BT_FIRST(idx)], SP_SLOW);
        drain(drain_ctx);
        uint8_t second = cmd_nibble(cc, in_btype >> 4, &c->btype_priors[BT_SECOND(idx)], SP_SLOW);
        btype = (uint8_t)((second << 4) | first);
    } else btype = (uint8_t)(varint - 2);
    drain(drain_ctx);
    uint8_t stride = c->desired_force_stride == 9 ? in_stride : c->desired_force_stride; /* UseBrotliRec = 9 */
    stride = cmd_nibble(cc, stride, &c->btype_priors[BT_STRIDE], SP_SLOW);
    /* obs_btypel, codec/interface.rs:527-537 */
    c->last_4_states >>= 2;
    c->btype_lru[idx][1] = c->btype_lru[idx][0]; c->btype_lru[idx][0] = btype;
    if (btype > c->btype_max_seen[idx]) c->btype_max_seen[idx] = btype;
    *out_btype = btype; *out_stride = stride;
    return 0;
}

static uint8_t round_up_mod_4(uint8_t v) { return (uint8_t)((((uint8_t)(v - 1)) | 3) + 1); } /* codec/interface.rs:180-182 */

/* literal length, codec/literal.rs:565-661.  Encoding passes len; decoding returns it. */
static int code_literal_length(cmd_state *c, cmd_coder *cc, void (*drain)(void *), void *drain_ctx, uint32_t len_in, uint32_t *len_out) {
    const uint32_t ctype = c->btype_lru[1][0];
    const uint32_t MN = 14; /* NUM_LITERAL_LENGTH_MNEMONIC */
    uint32_t serialized = len_in - (MN + 1);
    uint8_t lllen = (uint8_t)(serialized ? 32 - __builtin_clz(serialized) : 0);
    drain(drain_ctx);
    uint32_t lm1 = len_in - 1;
    uint8_t shortcut = cmd_nibble(cc, (uint8_t)(lm1 < MN ? lm1 : MN), &c->lit_len_priors[LL_COUNT_SMALL(ctype)], SP_MED);
    if (shortcut == MN + 1) return -1; /* high-entropy literals are not produced by the literal-only path */
    if (shortcut != MN) { *len_out = (uint32_t)shortcut + 1; return 0; }
    drain(drain_ctx);
    uint8_t beg = cmd_nibble(cc, lllen < 15 ? lllen : 15, &c->lit_len_priors[LL_SIZE_BEG(ctype)], SP_MUD);
    uint8_t len_remaining; uint32_t decoded;
    if (beg == 15) {
        drain(drain_ctx);
        uint8_t last = cmd_nibble(cc, (uint8_t)(lllen - 15), &c->lit_len_priors[LL_SIZE_LAST(ctype)], SP_MUD);
        len_remaining = round_up_mod_4((uint8_t)(last + 14));
        decoded = (uint32_t)1 << (last + 14);
    } else if (beg <= 1) {
        *len_out = MN + 1 + beg; return 0;
    } else {
        len_remaining = round_up_mod_4((uint8_t)(beg - 1));
        decoded = (uint32_t)1 << (beg - 1);
    }
    while (1) {
        drain(drain_ctx);
        uint8_t next_rem = (uint8_t)(len_remaining - 4);
        uint8_t nib = cmd_nibble(cc, (uint8_t)((serialized ^ decoded) >> next_rem), &c->lit_len_priors[LL_SIZE_MANT(ctype)], SP_MUD);
        decoded |= (uint32_t)nib << next_rem;
        if (next_rem == 0) { *len_out = decoded + MN + 1; return 0; }
        len_remaining = next_rem;
    }
}

/* command type nibble, codec/mod.rs:662-688 */
static uint8_t code_command_type(cmd_state *c, cmd_coder *cc, void (*drain)(void *), void *drain_ctx, uint8_t code) {
    drain(drain_ctx);
    code = cmd_nibble(cc, code, &c->cc_priors[c->last_4_states >> 4], SP_ROCKET);
    if (code == 3) { c->last_4_states >>= 2; c->last_4_states |= 128; } /* obs_literal_state */
    return code;
}

/* ------------------------------------------------------------------ encoder driver */
typedef struct {
    orc_mux mux; sink out; uint32_t crc;
    orc_ans_encoder cmd, lit;
    size_t cmd_drained, lit_drained;   /* bytes of enc.out already handed to the mux */
    int input_done, next_call_started; /* see call_returns */
} enc_ctx;

/* drain_or_fill_static_buffer for an encoder, codec/interface.rs:868-895 */
/* The call returns NEEDS_MORE_OUTPUT and the application comes back with an empty buffer; what it comes back WITH decides whether
 * a call boundary follows later: once an encode call has taken all of its input (input_done), the application's next call is the
 * next piece's divans_encode or divans_encode_flush (c/example.c:31-46), which first finishes the frozen commands
 * (divans_compressor.rs:189-207) -- the ORC_CMD_NEW_CALL marker that follows has then already happened. */
static void call_returns(enc_ctx *e) {
    sink_new_call(&e->out);
    if (e->input_done) { e->input_done = 0; e->next_call_started = 1; }
}
/* drain_or_fill_static_buffer (codec/interface.rs:868-896) for an encoder: linearize what the Mux will give, make room in the
 * stream's buffer, pop; with the coder still holding bytes and the caller's buffer full it reports NeedsMoreOutput.  `retry`: the
 * caller of the drain hands that to the application and the re-entered call repeats the drain; 0 = the one place where the
 * status is dropped (after the LAST byte of a Literal, literal.rs:376-390: the command completes, the LIT coder keeps its bytes
 * until the next LIT drain).  Returns 1 when the coder is empty. */
static int drain_coder(enc_ctx *e, int id, int retry) {
    orc_ans_encoder *enc = id == 0 ? &e->cmd : &e->lit;
    size_t *drained = id == 0 ? &e->cmd_drained : &e->lit_drained;
    while (*drained < enc->out.len) {
        size_t room = sink_room(&e->out);
        uint8_t *p = sink_ptr(&e->out, room);
        size_t n = mux_serialize(&e->mux, p, room);
        sink_commit(&e->out, n);
        mux_prep_push(&e->mux, 0, 16); mux_prep_push(&e->mux, 1, 16);   /* write_buffer, mux.rs:184-204 */
        mux_stream *b = &e->mux.s[id];
        size_t space = b->cap - b->end, avail = enc->out.len - *drained;
        size_t take = avail < space ? avail : space;
        memcpy(b->buf + b->end, enc->out.data + *drained, take);
        b->end += take; *drained += take;
        if (*drained < enc->out.len && sink_room(&e->out) == 0) {
            if (!retry) return 0;
            call_returns(e);
        }
    }
    return 1;
}
static void drain_cmd_cb(void *p) { (void)drain_coder((enc_ctx *)p, 0, 1); }

static void lit_config_from_pm(orc_lit_config *cfg, const orc_prediction_mode_result *r, uint8_t btype, uint8_t mixing) {
    memcpy(cfg->literal_context_map, r->literal_context_map, sizeof(cfg->literal_context_map));
    memcpy(cfg->mixing_mask, r->mixing_values, sizeof(cfg->mixing_mask));
    cfg->prediction_mode = r->prediction_mode;
    cfg->btype = btype;
    cfg->context_mixing = mixing;
    cfg->reserved = 0;
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = r->literal_adaptation[i];
}

void orc_stream_options_default(orc_stream_options *o) { /* src/interface.rs:463-483 */
    memset(o, 0, sizeof(*o));
    o->window_size = 22; o->dynamic_context_mixing = 1; o->use_context_map = 1; o->force_stride = 9;
    o->call_buffer_size = 65536;
}

static size_t stream_compress_impl(const orc_stream_options *o, const orc_stream_command *cmds, size_t n_cmds, uint8_t *out, size_t cap, int header_in_own_call) {
    enc_ctx e;
    memset(&e, 0, sizeof(e));
    orc_ans_encoder_init(&e.cmd); orc_ans_encoder_init(&e.lit);
    e.out.call_buf = o->call_buffer_size ? o->call_buffer_size : 65536;
    cmd_state *c = (cmd_state *)malloc(sizeof(cmd_state));
    cmd_state_init(c, o);
    cmd_coder cc = {&e.cmd, NULL, 0};
    orc_lit_config *cfg = (orc_lit_config *)calloc(1, sizeof(orc_lit_config));
    /* LiteralBookKeeping::new, codec/interface.rs:244-262 (+ reset on construction = zeroed map) */
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = SP_MUD;
    orc_lit_state *ls = orc_lit_state_new(cfg);
    int bad = 0;
    /* header, divans_compressor.rs:126-131,150-174 */
    {
        int w = o->window_size < 10 ? 10 : (o->window_size > 24 ? 24 : o->window_size);
        uint8_t hdr[16] = {0xff, 0xe5, 0x8c, 0x9f, 0, (uint8_t)w, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        size_t written = 0;
        while (written < 16) {
            if (sink_room(&e.out) == 0) call_returns(&e);
            size_t room = sink_room(&e.out), n = 16 - written < room ? 16 - written : room;
            memcpy(sink_ptr(&e.out, n), hdr + written, n);
            sink_commit(&e.out, n); written += n;
        }
        /* divans_encode(): the header goes out in the encode() call, which returns NeedsMoreInput while the ring
         * buffer is not full (raw_to_cmd/mod.rs:55-104); everything else is produced by the flush() calls */
        if (header_in_own_call) e.out.call_used = 0;
    }
    uint8_t btype = 0; uint8_t mixing = c->desired_context_mixing;
    for (size_t k = 0; k < n_cmds && !bad; ++k) {
        const orc_stream_command *cm = &cmds[k];
        if (cm->kind == ORC_CMD_PREDICTION_MODE) {
            code_command_type(c, &cc, drain_cmd_cb, &e, 7);
            orc_prediction_mode_result r;
            if (code_prediction_mode(c, &cc, drain_cmd_cb, &e, &cm->pm, &r)) { bad = 1; break; }
            mixing = r.mixing_math;
            lit_config_from_pm(cfg, &r, btype, mixing);
            orc_lit_state_reconfigure(ls, cfg);
        } else if (cm->kind == ORC_CMD_BLOCK_SWITCH_LITERAL) {
            code_command_type(c, &cc, drain_cmd_cb, &e, 4);
            uint8_t stride;
            code_block_switch_literal(c, &cc, drain_cmd_cb, &e, cm->btype, cm->stride, &btype, &stride);
            cfg->btype = btype;
            orc_lit_state_reconfigure(ls, cfg);
        } else if (cm->kind == ORC_CMD_NEW_CALL) {
            if (e.next_call_started) e.next_call_started = 0;     /* it began when the previous call ran out of output, see call_returns */
            else sink_new_call(&e.out);
            e.input_done = 0;
        } else if (cm->kind == ORC_CMD_INPUT_DONE) {
            e.input_done = 1;
        } else if (cm->kind == ORC_CMD_LITERAL) {
            if (cm->len == 0) { bad = 1; break; }
            code_command_type(c, &cc, drain_cmd_cb, &e, 3);
            uint32_t len_out;
            if (code_literal_length(c, &cc, drain_cmd_cb, &e, (uint32_t)cm->len, &len_out)) { bad = 1; break; }
            (void)drain_coder(&e, 0, 1); (void)drain_coder(&e, 1, 1);   /* literal.rs:496-520 (CMD), :426-433 (LIT) before the content */
            for (size_t i = 0; i < cm->len; ++i) {     /* code_nibble_array drains the LIT coder after every nibble; a chunk only */
                orc_lit_encode_bytes(ls, &e.lit, cm->data + i, 1);        /* ever completes on a low nibble (65 536 symbols, 2 per byte) */
                (void)drain_coder(&e, 1, i + 1 != cm->len);
            }
        } else bad = 1;
    }
    if (!bad) {
        /* DivansCodec::flush, codec/mod.rs:424-554 */
        code_command_type(c, &cc, drain_cmd_cb, &e, 0xf);
        (void)drain_coder(&e, 0, 1); (void)drain_coder(&e, 1, 1);   /* EncodedShutdownNode */
        orc_ans_flush_chunk(&e.cmd); orc_ans_flush_chunk(&e.lit);   /* ShutdownCoder(0), (1) */
        (void)drain_coder(&e, 0, 1); (void)drain_coder(&e, 1, 1);   /* CoderBufferDrain */
        while (e.mux.eof != 3) {                         /* MuxDrain */
            if (sink_room(&e.out) == 0) call_returns(&e);
            size_t room = sink_room(&e.out);
            size_t n = mux_serialize_close(&e.mux, sink_ptr(&e.out, room), room);
            sink_commit(&e.out, n);
        }
        uint32_t crc = orc_crc32c_update(0, e.out.data, e.out.len);   /* every emitted byte, header included */
        uint8_t tr[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), 'a', 'n', 's', '~'};
        memcpy(sink_ptr(&e.out, 8), tr, 8); e.out.len += 8;
        if (e.cmd.failed || e.lit.failed) bad = 1;
    }
    size_t ret = (size_t)-1;
    if (!bad && e.out.len <= cap) { memcpy(out, e.out.data, e.out.len); ret = e.out.len; }
    free(e.out.data); mux_free(&e.mux);
    orc_ans_encoder_free(&e.cmd); orc_ans_encoder_free(&e.lit);
    orc_lit_state_free(ls); free(cfg); free(c);
    return ret;
}

size_t orc_stream_compress(const orc_stream_options *o, const orc_stream_command *cmds, size_t n_cmds, uint8_t *out, size_t cap) {
    return stream_compress_impl(o, cmds, n_cmds, out, cap, 0);   /* encode_commands(): header and commands share the call */
}

/* The literal-only internal compressor (use_brotli = UseInternalCommandSelection).
 * RawToCmdState (raw_to_cmd/mod.rs:55-181) copies the caller's bytes into a 2^window ring and emits commands only when the
 * ring is full (stream, :83) or at flush: first [PredictionMode], then Literal commands over the ring spans it has not
 * handed out yet.  The spans follow the ring's index dance, restated literally below: the first lap is 2^w bytes, the
 * second 2^w - 1, lap k >= 3 is two commands of k-2 and 2^w-k+1 bytes.  One deviation: when the input ends inside the
 * few bytes a lap first writes at the END of the ring, the reference resets its write index anyway (:70-72) and later
 * emits the whole tail span, stale bytes included (its round trip then fails); only the fresh bytes are emitted here. */
typedef struct { size_t ring, dec, outi, tail_fresh, emitted; int has_header; orc_stream_command *cmds; size_t n; const uint8_t *in;
                 orc_prediction_mode pm; } raw_sim;

static void raw_sim_literal(raw_sim *r, size_t len) {
    if (!len) return;
    orc_stream_command *c = &r->cmds[r->n++];
    memset(c, 0, sizeof(*c));
    c->kind = ORC_CMD_LITERAL; c->data = r->in + r->emitted; c->len = len;
    r->emitted += len;
}
static void raw_sim_flush(raw_sim *r) {             /* RawToCmdState::flush, :105-181 */
    if (!r->has_header) {
        r->has_header = 1;
        orc_stream_command *c = &r->cmds[r->n++];
        memset(c, 0, sizeof(*c));
        c->kind = ORC_CMD_PREDICTION_MODE; c->pm = r->pm;
    }
    if (r->dec < r->outi) {
        raw_sim_literal(r, r->tail_fresh);         /* reference: ring.len() - output_index bytes */
        r->tail_fresh = 0;
        if (r->dec == r->ring) r->dec = 0;
        r->outi = 0;
    }
    if (r->dec != r->outi) { raw_sim_literal(r, r->dec - r->outi); r->outi = r->dec; }
}
static void raw_sim_stream(raw_sim *r, size_t *pos, size_t call_end) {   /* DivansCompressor::encode loop over RawToCmdState::stream */
    for (;;) {
        if (r->dec >= r->outi) {
            size_t mc = r->ring - r->dec < call_end - *pos ? r->ring - r->dec : call_end - *pos;
            *pos += mc; r->dec += mc;
            if (r->outi != 0) { r->tail_fresh = r->dec - r->outi; r->dec = 0; }
        }
        if (r->dec < r->outi) {
            size_t mc = r->outi - 1 - r->dec < call_end - *pos ? r->outi - 1 - r->dec : call_end - *pos;
            *pos += mc; r->dec += mc;
        }
        if (r->dec == r->ring || r->dec + 1 == r->outi) {     /* ring_buffer_full */
            if (*pos == call_end) r->cmds[r->n++].kind = ORC_CMD_INPUT_DONE;   /* what follows is coded with the call's input all taken */
            raw_sim_flush(r);
            if (*pos != call_end) continue;
        }
        break;
    }
}

size_t orc_stream_compress_raw(const orc_stream_options *o, const uint8_t *in, size_t n, uint8_t *out, size_t cap) {
    int w = o->window_size < 10 ? 10 : (o->window_size > 24 ? 24 : o->window_size);
    raw_sim r;
    memset(&r, 0, sizeof(r));
    r.ring = (size_t)1 << w; r.in = in;
    size_t ncalls = o->call_inputs ? o->n_call_inputs : 1;
    r.cmds = (orc_stream_command *)calloc(2 * (n / (r.ring - 1) + 2) + 2 * ncalls + 8, sizeof(orc_stream_command));
    uint8_t cmap[64], dmap[4], mixing[ORC_NUM_MIXING_VALUES];      /* (locals: the function is called from concurrent test threads) */
    for (int i = 0; i < 64; ++i) cmap[i] = (uint8_t)(i & 0x3f);
    for (int i = 0; i < 4; ++i) dmap[i] = (uint8_t)(i & 3);
    memset(mixing, 4, sizeof(mixing));
    r.pm.prediction_mode = 0;
    r.pm.literal_context_map = cmap; r.pm.n_literal_context_map = 64;
    r.pm.distance_context_map = dmap; r.pm.n_distance_context_map = 4;
    r.pm.mixing_values = mixing; r.pm.has_context_speeds = 1;
    size_t pos = 0;
    for (size_t k = 0; k < ncalls; ++k) {            /* every divans_encode call starts with a fresh output buffer; the first also carries the header */
        size_t m = o->call_inputs ? o->call_inputs[k] : n;
        if (m > n - pos) m = n - pos;
        if (k) r.cmds[r.n++].kind = ORC_CMD_NEW_CALL;
        raw_sim_stream(&r, &pos, pos + m);
    }
    /* divans_encode_flush: an empty input still flushes a PredictionMode command (has_produced_header, :114-143) */
    r.cmds[r.n++].kind = ORC_CMD_NEW_CALL;
    raw_sim_flush(&r);
    size_t ret = (pos == n && r.emitted == n) ? stream_compress_impl(o, r.cmds, r.n, out, cap, 0) : (size_t)-1;
    free(r.cmds);
    return ret;
}

/* ------------------------------------------------------------------ decoder */
static void no_drain(void *p) { (void)p; }

/* What LiteralBookKeeping holds after the encoder has coded `pm` under options `o` (obs_prediction_mode_context_map,
 * codec/interface.rs:293-319); pm == NULL: the constructor defaults (codec/interface.rs:244-262). */
int orc_lit_config_from_prediction_mode(const orc_stream_options *o, const orc_prediction_mode *pm, uint8_t btype, orc_lit_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->btype = btype;
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = SP_MUD;
    if (!pm) return 0;
    cmd_state *c = (cmd_state *)malloc(sizeof(cmd_state));
    cmd_state_init(c, o);
    orc_ans_encoder enc;
    orc_ans_encoder_init(&enc);
    cmd_coder cc = {&enc, NULL, 0};
    code_command_type(c, &cc, no_drain, NULL, 7);
    orc_prediction_mode_result r;
    int rc = code_prediction_mode(c, &cc, no_drain, NULL, pm, &r);
    if (!rc) lit_config_from_pm(cfg, &r, btype, r.mixing_math);
    orc_ans_encoder_free(&enc); free(c);
    return rc;
}

int orc_stream_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (n < 16 + 3 + 8) return -1;
    if (in[0] != 0xff || in[1] != 0xe5 || in[2] != 0x8c || in[3] != 0x9f) return -2;  /* divans_decompressor.rs:38-52 */
    if (in[5] < 10 || in[5] >= 25) return -2;
    orc_mux m; mux_parse p;
    memset(&m, 0, sizeof(m)); memset(&p, 0, sizeof(p));
    size_t used = mux_deserialize(&m, &p, in + 16, n - 16);
    if (m.eof != 3 || n - 16 - used < 8) { mux_free(&m); return -3; }
    const uint8_t *tr = in + 16 + used;
    uint32_t crc = orc_crc32c_update(0, in, 16 + used);
    uint8_t want[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), 'a', 'n', 's', '~'};
    if (memcmp(tr, want, 8) != 0) { mux_free(&m); return -4; }    /* codec/mod.rs:949-1017 */
    orc_stream_options o;
    orc_stream_options_default(&o);
    cmd_state *c = (cmd_state *)malloc(sizeof(cmd_state));
    cmd_state_init(c, &o);   /* every desired_* only shapes what an ENCODER would put; the decoder takes the coded nibble */
    orc_ans_decoder cd, ld;
    orc_ans_decoder_init(&cd, m.s[0].buf ? m.s[0].buf + m.s[0].start : (const uint8_t *)"", m.s[0].end - m.s[0].start);
    orc_ans_decoder_init(&ld, m.s[1].buf ? m.s[1].buf + m.s[1].start : (const uint8_t *)"", m.s[1].end - m.s[1].start);
    cmd_coder cc = {NULL, &cd, 0};
    orc_lit_config *cfg = (orc_lit_config *)calloc(1, sizeof(orc_lit_config));
    for (int i = 0; i < 4; ++i) cfg->literal_adaptation[i] = SP_MUD;
    orc_lit_state *ls = orc_lit_state_new(cfg);
    size_t produced = 0; int rc = 0; uint8_t btype = 0;
    for (;;) {
        uint8_t code = code_command_type(c, &cc, no_drain, NULL, 0);
        if (cd.starved) { rc = -5; break; }
        if (code == 0xf) break;
        if (code == 7) {
            orc_prediction_mode_result r;
            if (code_prediction_mode(c, &cc, no_drain, NULL, NULL, &r)) { rc = -6; break; }
            lit_config_from_pm(cfg, &r, btype, r.mixing_math);
            orc_lit_state_reconfigure(ls, cfg);
        } else if (code == 4) {
            uint8_t stride;
            code_block_switch_literal(c, &cc, no_drain, NULL, 0, 0, &btype, &stride);
            cfg->btype = btype;
            orc_lit_state_reconfigure(ls, cfg);
        } else if (code == 3) {
            uint32_t len;
            if (code_literal_length(c, &cc, no_drain, NULL, 15, &len)) { rc = -7; break; }
            if (produced + len > cap) { rc = -8; break; }
            orc_lit_decode_bytes(ls, &ld, out + produced, len);
            if (ld.starved) { rc = -9; break; }
            produced += len;
        } else { rc = -10; break; }   /* Copy / Dict / other block switches: outside the literal-only scope */
    }
    if (out_len) *out_len = produced;
    free(cfg); orc_lit_state_free(ls); free(c); mux_free(&m);
    return rc;
}

/* ------------------------------------------------------------------ walking a reference-written CMD stream (test pin) */
/* prior tables the literal-only path never touches: CopyCommandPriors / DictCommandPriors, codec/priors.rs:73-99; rows are
 * (ctype or distance prior) + 256 * index, types in the order the struct lists them (priors.rs:211-259) */
enum { CP_DIST_BEG = 0, CP_DIST_MNEM = CP_DIST_BEG + 256 * 64, CP_DIST_LAST = CP_DIST_MNEM + 256 * 2, CP_DIST_MANT = CP_DIST_LAST + 256,
       CP_COUNT_SMALL = CP_DIST_MANT + 256 * 5, CP_COUNT_BEG = CP_COUNT_SMALL + 256 * 64, CP_COUNT_LAST = CP_COUNT_BEG + 256 * 64,
       CP_COUNT_MANT = CP_COUNT_LAST + 256 * 64, CP_TOTAL = CP_COUNT_MANT + 256 * 64 };
enum { DC_SIZE_BEG = 0, DC_SIZE_LAST = 256, DC_INDEX = 512, DC_TRANSFORM = DC_INDEX + 256 * 5, DC_TOTAL = DC_TRANSFORM + 2 * 25 };
typedef struct {
    cmd_state c;
    orc_cdf16 copy_priors[CP_TOTAL], dict_priors[DC_TOTAL];
    uint32_t distance_lru[4], last_llen;     /* codec/interface.rs:153,161,371-373,396 */
    uint8_t last_clen, last_dlen;
} walk_state;

static uint8_t bit_length32(uint32_t v) { return (uint8_t)(v ? 32 - __builtin_clz(v) : 0); }
static uint32_t walk_distance_prior(const walk_state *s, uint32_t copy_len) {       /* get_distance_prior, codec/interface.rs:426-430 */
    uint32_t l = copy_len < 2 ? 2 : copy_len;
    return s->c.distance_context_map[(uint32_t)s->c.btype_lru[2][0] * 4 + (l - 2 < 3 ? l - 2 : 3)];
}
static void walk_obs_state(walk_state *s, uint8_t bits) { s->c.last_4_states = (uint8_t)((s->c.last_4_states >> 2) | bits); }

/* CopyState::encode_or_decode as a decoder, codec/copy.rs:49-290 */
static int walk_copy(walk_state *s, cmd_coder *cc, orc_walk_command *out) {
    const uint32_t ctype = s->c.btype_lru[1][0];
    uint32_t index = ((s->c.last_4_states >> 4) & 3) + 4 * (s->last_llen - 1 < 3 ? s->last_llen - 1 : 3);
    uint32_t num_bytes;
    uint8_t sc = cmd_nibble(cc, 0, &s->copy_priors[CP_COUNT_SMALL + ctype + 256 * index], SP_MUD);
    if (sc == 15) {
        uint8_t beg = cmd_nibble(cc, 0, &s->copy_priors[CP_COUNT_BEG + ctype], SP_FAST), rem; uint32_t dec;
        if (beg == 15) {
            uint8_t last = cmd_nibble(cc, 0, &s->copy_priors[CP_COUNT_LAST + ctype], SP_FAST);
            s->last_clen = (uint8_t)(last + 19); rem = round_up_mod_4((uint8_t)(last + 18)); dec = (uint32_t)1 << (last + 18);
        } else { s->last_clen = (uint8_t)(beg + 4); rem = round_up_mod_4((uint8_t)(beg + 3)); dec = (uint32_t)1 << (beg + 3); }
        for (uint8_t done = 0;; done += 4) {
            uint8_t next = (uint8_t)(rem - 4);
            uint32_t mi = done == 0 ? (uint32_t)(s->last_clen % 4) + 1 : 0;
            dec |= (uint32_t)cmd_nibble(cc, 0, &s->copy_priors[CP_COUNT_MANT + ctype + 256 * mi], SP_SLOW) << next;
            if (!next) break;
            rem = next;
        }
        num_bytes = dec;
    } else { num_bytes = sc; s->last_clen = bit_length32(num_bytes); }
    const uint32_t prior = walk_distance_prior(s, num_bytes);
    uint32_t distance;
    uint8_t mn = cmd_nibble(cc, 0, &s->copy_priors[CP_DIST_MNEM + prior + 256 * (s->last_llen < 8)], SP_SLOW);
    if (mn != 15) {                                   /* get_distance_from_mnemonic_code, codec/interface.rs:979-1009 */
        int32_t d;
        if (mn < 4) d = (int32_t)s->distance_lru[mn];
        else { int32_t us = mn >> 2, ss = us - (((-(int32_t)(mn & 1)) & us) << 1); d = (int32_t)s->distance_lru[(mn & 2) >> 1] + ss; }
        if (d <= 0) return -12;
        distance = (uint32_t)d; s->last_dlen = bit_length32(distance);
    } else {
        uint8_t beg = cmd_nibble(cc, 0, &s->copy_priors[CP_DIST_BEG + prior + 256 * (bit_length32(num_bytes) >> 2)], SP_SLOW);
        if (beg == 15) { distance = s->distance_lru[1] - 3; s->last_dlen = bit_length32(distance); }
        else {
            uint8_t rem = 0; uint32_t dec = 1;
            if (beg == 14) {
                uint8_t last = cmd_nibble(cc, 0, &s->copy_priors[CP_DIST_LAST + prior], SP_ROCKET);
                s->last_dlen = (uint8_t)(last + 15); rem = round_up_mod_4((uint8_t)(last + 14)); dec = (uint32_t)1 << (last + 14);
            } else { s->last_dlen = (uint8_t)(beg + 1); if (beg) { rem = round_up_mod_4(beg); dec = (uint32_t)1 << beg; } }
            uint8_t done = 0;
            for (int sr2 = ((int)rem + 3) >> 2; sr2-- > 0; done += 4) {
                uint32_t mi = done == 0 ? (uint32_t)(s->last_dlen & 3) + 1 : 0;
                orc_speed sp = {(int16_t)(0x4 << ((mi & 6) << ((mi & 2) >> 1))), 0x4000};
                dec |= (uint32_t)cmd_nibble(cc, 0, &s->copy_priors[CP_DIST_MANT + prior + 256 * mi], sp) << (sr2 << 2);
            }
            distance = dec;
        }
    }
    uint32_t *l = s->distance_lru;                     /* obs_distance, codec/interface.rs:509-527 */
    if (distance == l[1]) { l[1] = l[0]; l[0] = distance; }
    else if (distance == l[2]) { l[2] = l[1]; l[1] = l[0]; l[0] = distance; }
    else if (distance != l[0]) { l[3] = l[2]; l[2] = l[1]; l[1] = l[0]; l[0] = distance; }
    out->x = distance; out->y = num_bytes;
    return 0;
}

/* DictState::encode_or_decode as a decoder, codec/dict.rs:36-190 (nibbles only) */
static int walk_dict(walk_state *s, cmd_coder *cc, orc_walk_command *out) {
    static const uint8_t DICT_BITS[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
    const uint32_t ctype = s->c.btype_lru[1][0];
    uint8_t ws = cmd_nibble(cc, 0, &s->dict_priors[DC_SIZE_BEG + ctype], SP_MUD);
    ws = ws == 15 ? (uint8_t)(cmd_nibble(cc, 0, &s->dict_priors[DC_SIZE_LAST + ctype], SP_MUD) + 19) : (uint8_t)(ws + 4);
    if (ws > 24) return -13;
    uint8_t rem = round_up_mod_4(DICT_BITS[ws]); uint32_t id = 0;
    for (uint8_t done = 0;; done += 4) {
        uint8_t next = (uint8_t)(rem - 4);
        uint32_t mi = done == 0 ? (uint32_t)(DICT_BITS[ws] % 4) + 1 : 0;
        id |= (uint32_t)cmd_nibble(cc, 0, &s->dict_priors[DC_INDEX + walk_distance_prior(s, ws) + 256 * mi], SP_MUD) << next;
        if (!next) break;
        rem = next;
    }
    uint8_t hi = cmd_nibble(cc, 0, &s->dict_priors[DC_TRANSFORM + 0 + 2 * (ws >> 1)], SP_FAST);
    uint8_t lo = cmd_nibble(cc, 0, &s->dict_priors[DC_TRANSFORM + 1 + 2 * hi], SP_FAST);
    out->a = ws; out->b = (uint8_t)((hi << 4) | lo); out->x = id;
    return 0;
}

int orc_cmd_stream_walk(const uint8_t *cmd, size_t n, int wire, orc_cmd_walk *w) {
    memset(w, 0, sizeof(*w));
    walk_state *s = (walk_state *)malloc(sizeof(walk_state));
    if (!s) return -1;
    orc_stream_options o;
    orc_stream_options_default(&o);
    cmd_state_init(&s->c, &o);
    s->c.wire = wire;
    for (size_t i = 0; i < CP_TOTAL; ++i) orc_cdf_default(&s->copy_priors[i]);
    for (size_t i = 0; i < DC_TOTAL; ++i) orc_cdf_default(&s->dict_priors[i]);
    s->distance_lru[0] = 4; s->distance_lru[1] = 11; s->distance_lru[2] = 15; s->distance_lru[3] = 16;
    s->last_llen = 1; s->last_clen = 1; s->last_dlen = 1;
    orc_ans_decoder cd;
    orc_ans_decoder_init(&cd, cmd, n);
    cmd_coder cc = {NULL, &cd, 0};
    int rc = 0;
    for (;;) {
        uint8_t code = cmd_nibble(&cc, 0, &s->c.cc_priors[s->c.last_4_states >> 4], SP_ROCKET); /* codec/mod.rs:662-688 */
        if (cd.starved) { rc = -5; break; }
        if (w->n_cmds == 64) { rc = -11; break; }
        orc_walk_command *k = &w->cmds[w->n_cmds++];
        k->kind = code;
        if (code == 0xf) break;
        if (code == 7) {
            if (code_prediction_mode(&s->c, &cc, no_drain, NULL, NULL, &w->pm)) { rc = -6; break; }
            w->mixing_value_min = 255;
            for (size_t i = 0; i < ORC_NUM_MIXING_VALUES; ++i) {
                if (s->c.pm_mixing[i] < w->mixing_value_min) w->mixing_value_min = s->c.pm_mixing[i];
                if (s->c.pm_mixing[i] > w->mixing_value_max) w->mixing_value_max = s->c.pm_mixing[i];
            }
            for (size_t i = 0; i < ORC_MAX_LITERAL_CONTEXT_MAP_SIZE; ++i) if (s->c.pm_literal_context_map[i]) w->literal_context_map_nonzero = 1;
            w->pm.literal_context_map = NULL; w->pm.mixing_values = NULL;   /* the storage dies with the walk */
        } else if (code == 4) {
            code_block_switch_literal(&s->c, &cc, no_drain, NULL, 0, 0, &k->a, &k->b);
        } else if (code == 5 || code == 6) {            /* BlockTypeState for the command / distance switch, codec/block_type.rs:31-107 */
            const int idx = code - 4;
            uint8_t v = cmd_nibble(&cc, 0, &s->c.btype_priors[BT_MNEMONIC(idx)], SP_SLOW), b;
            if (v == 0) b = s->c.btype_lru[idx][1];
            else if (v == 1) b = (uint8_t)(s->c.btype_max_seen[idx] + 1);
            else if (v == 15) {
                uint8_t first = cmd_nibble(&cc, 0, &s->c.btype_priors[BT_FIRST(idx)], SP_SLOW);
                b = (uint8_t)((cmd_nibble(&cc, 0, &s->c.btype_priors[BT_SECOND(idx)], SP_SLOW) << 4) | first);
            } else b = (uint8_t)(v - 2);
            s->c.last_4_states >>= 2;
            s->c.btype_lru[idx][1] = s->c.btype_lru[idx][0]; s->c.btype_lru[idx][0] = b;
            if (b > s->c.btype_max_seen[idx]) s->c.btype_max_seen[idx] = b;
            k->a = b;
        } else if (code == 3) {
            walk_obs_state(s, 128);
            if (code_literal_length(&s->c, &cc, no_drain, NULL, 15, &k->x)) { rc = -7; break; }
            /* last_llen: set on the count-small and mantissa exits only, codec/literal.rs:587,649 -- the two-nibble lengths 15 / 16 leave it */
            if (k->x < 15 || k->x > 16) s->last_llen = k->x;
        } else if (code == 1) {
            walk_obs_state(s, 64);
            if ((rc = walk_copy(s, &cc, k)) != 0) break;
        } else if (code == 2) {
            walk_obs_state(s, 192);
            if ((rc = walk_dict(s, &cc, k)) != 0) break;
        } else { rc = -10; break; }
        if (cd.starved) { rc = -5; break; }
    }
    w->nibbles = cc.nibbles; w->state_a = cd.state_a; w->state_b = cd.state_b; w->consumed = cd.in_pos; w->starved = cd.starved;
    free(s);
    return rc;
}
