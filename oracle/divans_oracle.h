/*
 * divans_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the literal-coding hot path of dropbox/divans
 * (reference tree mounted at /root/reference, citations are relative to it).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped GPU path (divans_amd/csrc) never links it.
 *
 * PARITY STATUS: pinned on the one compressed vector the reference tree holds
 * (wasm/wasm.html:98-107, 113 bytes from a real Rust build; tests/golden/
 * ref_wasm_example.divans, tests/test_reference_vectors.py): header, Mux
 * framing, end marker, CRC-32C trailer, all 8253 nibbles of its CMD stream
 * (final rANS states = the encoder's start states, every byte consumed) and all
 * 28 literal bytes of its LIT stream decode with this code, and re-encoding the
 * decoded literals gives the example's 36 LIT bytes back.  That build's wire
 * format is older than HEAD in two prior-table rows of the PredictionMode
 * command (ORC_WIRE_WASM_EXAMPLE), so bytes of a HEAD build remain unpinned for
 * those rows, for context maps / mixing, and for streams past one rANS chunk:
 * the reference is 100% Rust and cannot be built here (no rustc/cargo, crates
 * not vendored).  Also pinned against the reference's own tests (tests/test_oracle_*.py):
 *   - CDF identities            src/probability/common_tests.rs:3-126
 *   - exact division            src/probability/numeric.rs:73-85, make_div_lut.rs
 *   - Speed f8 codec            src/probability/interface.rs:590-616
 *   - CRC-32C KATs              src/codec/crc32.rs:95-116
 *   - mux framing KAT           src/test_mux.rs:1192-1207
 *   - round trips + size bounds src/bin/benchmark.rs:409-427, integration_test.rs:235-236
 */
#ifndef DIVANS_ORACLE_H_
#define DIVANS_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/probability/interface.rs:3,426-429 ---- */
typedef int16_t orc_prob;
#define ORC_LOG2_SCALE 15
#define ORC_BLEND_FIXED_POINT_PRECISION 15

/* FrequentistCDF16, src/probability/frequentist_cdf.rs:13-23 */
typedef struct { orc_prob cdf[16]; } orc_cdf16;

/* Speed(inc, lim), src/probability/interface.rs:298-375 */
typedef struct { int16_t inc, lim; } orc_speed;

typedef struct { orc_prob start, freq; uint8_t sym; } orc_sym_start_freq;

void orc_cdf_default(orc_cdf16 *c);
/* frequentist_cdf.rs:74-85 */
void orc_cdf_blend(orc_cdf16 *c, uint8_t sym, orc_speed sp);
/* frequentist_cdf.rs:58-72 : out = self.average(other, mix_rate) */
void orc_cdf_average(const orc_cdf16 *self, const orc_cdf16 *other, int32_t mix_rate, orc_cdf16 *out);
/* probability/interface.rs:97-108 ; returns 0 ok, -1 if max()==0 (reference would panic) */
int orc_cdf_sym_to_start_and_freq(const orc_cdf16 *c, uint8_t sym, orc_sym_start_freq *out);
/* probability/interface.rs:136-198 */
int orc_cdf_offset_to_sym_start_and_freq(const orc_cdf16 *c, orc_prob cdf_offset, orc_sym_start_freq *out);
/* numeric.rs:16-31 with the reciprocal computed by compute_divisor (what div_lut.rs tabulates) */
int32_t orc_fast_divide_30bit_by_16bit(int32_t num, int16_t denom);
/* numeric.rs:50-62 */
int16_t orc_fast_divide_16bit_by_8bit(uint16_t num, uint8_t denom);
/* probability/interface.rs:566-585 */
uint8_t orc_speed_to_u8(int16_t data);
int16_t orc_u8_to_speed(uint8_t data);
/* the 15-entry ENCODER_DEFAULT_PALETTE, probability/interface.rs:303-320 */
orc_speed orc_speed_palette(int index);

/* ---- Weights, src/codec/weights.rs:4-133 ---- */
typedef struct {
    int32_t model_weights[2];
    uint8_t mixing_param;
    int16_t normalized_weight;
} orc_weights;
void orc_weights_init(orc_weights *w);
void orc_weights_update(orc_weights *w, const orc_prob model_probs[2], orc_prob weighted_prob);

/* ---- growable byte buffer used by the coders ---- */
typedef struct { uint8_t *data; size_t len, cap; } orc_bytes;
void orc_bytes_free(orc_bytes *b);

/* ---- ANS, src/ans.rs ---- */
#define ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH 65536u /* ans.rs:57,138 */
typedef struct {
    /* start_freq stack: 4 B per symbol (start lo,hi, freq lo,hi), ans.rs:287-301 */
    uint16_t *start, *freq;  /* pending pairs of the current chunk, oldest first */
    uint32_t n_pending;
    orc_bytes out;           /* finished chunks, in the order the decoder reads them */
    int failed;              /* freq<=0 or start<0 seen (reference debug_asserts) */
    uint32_t *words_scratch; /* renormalisation words of the chunk being flushed (reused between chunks) */
} orc_ans_encoder;
void orc_ans_encoder_init(orc_ans_encoder *e);
void orc_ans_encoder_reset(orc_ans_encoder *e);
void orc_ans_encoder_free(orc_ans_encoder *e);
void orc_ans_put_start_freq(orc_ans_encoder *e, orc_prob start, orc_prob freq); /* ans.rs:287-301 */
void orc_ans_put_nibble(orc_ans_encoder *e, uint8_t sym, const orc_cdf16 *cdf, orc_sym_start_freq *coded); /* ans.rs:279-286 */
void orc_ans_flush_chunk(orc_ans_encoder *e);                                   /* ans.rs:331-378 */

typedef struct {
    uint64_t state_a, state_b; /* ans.rs:142-162 */
    uint16_t sym_count;
    uint8_t buffer_a_bytes_required, buffer_b_bytes_required;
    const uint8_t *in; size_t in_len, in_pos;
    int starved;               /* ran out of input while a refill was required */
} orc_ans_decoder;
void orc_ans_decoder_init(orc_ans_decoder *d, const uint8_t *in, size_t in_len);
/* fill (drain_or_fill_static_buffer for a decoder) + get_nibble, ans.rs:246-252,428-442 */
uint8_t orc_ans_get_nibble(orc_ans_decoder *d, const orc_cdf16 *cdf, orc_sym_start_freq *coded);

/* ---- literal coder configuration: what LiteralBookKeeping holds after the
 *      PredictionMode + BlockSwitchLiteral commands, codec/interface.rs:125-340 ---- */
#define ORC_MAX_LITERAL_CONTEXT_MAP_SIZE (256 * 64)
#define ORC_NUM_MIXING_VALUES 8192
typedef struct {
    uint8_t literal_context_map[ORC_MAX_LITERAL_CONTEXT_MAP_SIZE];
    uint8_t mixing_mask[ORC_NUM_MIXING_VALUES];
    uint8_t prediction_mode;  /* brotli numbering LSB6=0 MSB6=1 UTF8=2 SIGN=3 */
    uint8_t btype;            /* literal block type (btype_last) */
    uint8_t context_mixing;   /* Weights::mixing_param; >1 selects MixingTrait */
    uint8_t reserved;
    orc_speed literal_adaptation[4]; /* [0..2) stride, [2..4) context map */
} orc_lit_config;

/* config 2 of BASELINE.json: reference TestSimple (benchmark.rs:195-206) */
void orc_lit_config_simple(orc_lit_config *cfg);
/* config 3: reference TestContextMixing via bench_no_ir (benchmark.rs:156-167,305-343) */
void orc_lit_config_context_mixing(orc_lit_config *cfg);
void orc_get_lut0(uint8_t mode, uint8_t out[256]);
void orc_get_lut1(uint8_t mode, uint8_t out[256]);

/* Literal-byte coder state (LIT_CODER side only). */
typedef struct orc_lit_state orc_lit_state;
orc_lit_state *orc_lit_state_new(const orc_lit_config *cfg);
void orc_lit_state_reconfigure(orc_lit_state *s, const orc_lit_config *cfg); /* obs_prediction_mode_context_map */
void orc_lit_state_free(orc_lit_state *s);
/* code_nibble_array, literal.rs:261-394 (encoder / decoder instantiation) */
void orc_lit_encode_bytes(orc_lit_state *s, orc_ans_encoder *enc, const uint8_t *in, size_t n);
void orc_lit_decode_bytes(orc_lit_state *s, orc_ans_decoder *dec, uint8_t *out, size_t n);
void orc_lit_set_last8(orc_lit_state *s, uint64_t last8);
uint64_t orc_lit_get_last8(const orc_lit_state *s);

/* One independent literal stream (fresh priors): returns number of LIT-coder bytes written
 * to out (<= cap) or (size_t)-1 on overflow/failure. */
size_t orc_lit_stream_encode(const orc_lit_config *cfg, const uint8_t *in, size_t n, uint8_t *out, size_t cap);
int orc_lit_stream_decode(const orc_lit_config *cfg, const uint8_t *in, size_t in_len, uint8_t *out, size_t n);
/* Debug trace for kernel bring-up: per nibble (sym,start,freq). trace has 2*n entries of 3 int16. */
size_t orc_lit_stream_encode_trace(const orc_lit_config *cfg, const uint8_t *in, size_t n,
                                   uint8_t *out, size_t cap, int16_t *trace);

/* batch helper for the CPU baseline: nthreads>=1 pthreads, each stream independent */
int orc_lit_batch_roundtrip(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len,
                            int nthreads, double *enc_seconds, double *dec_seconds, uint64_t *coded_bytes);

/* literals of a general stream: one segment per Literal command (block type, reloaded last_8_literals) */
size_t orc_lit_segments_encode(const orc_lit_config *cfg, const uint8_t *lit, size_t n, const uint32_t *seg_len,
                               const uint32_t *seg_btype, const uint64_t *seg_last8, size_t nseg, uint8_t *out, size_t cap);
int orc_lit_segments_decode(const orc_lit_config *cfg, const uint8_t *in, size_t in_len, const uint32_t *seg_len,
                            const uint32_t *seg_btype, const uint64_t *seg_last8, size_t nseg, uint8_t *out, size_t n);

/* CPU baseline proper: every worker allocates its coder state, ANS buffers and output slots BEFORE a start barrier;
 * all workers then encode their streams (stream i -> worker i % nthreads), meet at a second barrier and decode them.
 * enc_wall / dec_wall = wall clock from the barrier release to the last worker finishing that direction. */
int orc_lit_batch_bench(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len,
                        int nthreads, double *enc_wall, double *dec_wall, uint64_t *coded_bytes);

/* checker: encode n_streams streams on nthreads workers and compare each with coded[off[i] .. off[i] + size[i]);
 * returns how many differ (-1: allocation failure), *first_bad = index of the first that does (n_streams if none) */
long orc_lit_batch_check(const orc_lit_config *cfg, const uint8_t *in, size_t n_streams, size_t stream_len, int nthreads,
                         const uint8_t *coded, const uint64_t *off, const uint32_t *size, size_t *first_bad);

/* ---- complete literal-only .divans streams (stream.c) ---- */
typedef struct {
    int window_size;                 /* header byte 5, clamped to [10,24] (divans_compressor.rs:89) */
    uint8_t dynamic_context_mixing;  /* DivansCompressorOptions, src/interface.rs:444-484 */
    uint8_t prior_depth;
    int use_context_map;
    uint8_t force_stride;            /* StrideSelection 0..8, 9 = UseBrotliRec */
    int has_literal_adaptation; orc_speed literal_adaptation[4];
    size_t call_buffer_size;         /* size of the output buffer the caller hands to each encode/flush call */
    const size_t *call_inputs;       /* bytes the caller passes to each divans_encode call (NULL: all of them in one call);   */
    size_t n_call_inputs;            /*   decides when the ring buffer fills and commands are emitted, raw_to_cmd/mod.rs:55-104 */
} orc_stream_options;
void orc_stream_options_default(orc_stream_options *o);

/* PredictionModeContextMap as the encoder receives it (brotli::enc::interface, field accessors used by context_map.rs) */
typedef struct {
    uint8_t prediction_mode, is_adv_context_map;
    const uint8_t *literal_context_map; size_t n_literal_context_map;
    const uint8_t *distance_context_map; size_t n_distance_context_map;
    const uint8_t *mixing_values;    /* 8192 or NULL */
    int has_context_speeds;
    uint8_t context_map_speed_f8[2][2], stride_speed_f8[2][2], combined_stride_speed_f8[2][2]; /* (inc,lim) f8 pairs */
} orc_prediction_mode;
typedef struct {                     /* what the codec's own PredictionModeContextMap holds afterwards */
    uint8_t prediction_mode, mixing_math;
    orc_speed literal_adaptation[4];
    const uint8_t *literal_context_map;  /* 16384 */
    const uint8_t *mixing_values;        /* 8192 */
} orc_prediction_mode_result;
enum { ORC_CMD_PREDICTION_MODE = 7, ORC_CMD_BLOCK_SWITCH_LITERAL = 4, ORC_CMD_LITERAL = 3,
       ORC_CMD_NEW_CALL = 100, /* not a command: the current encode()/flush() call returns, the next one brings a fresh output buffer */
       ORC_CMD_INPUT_DONE = 101 /* not a command: the current encode() call has taken all of its input; if it now runs out of output,
                                   the application's NEXT call (the following ORC_CMD_NEW_CALL) is the one that resumes it */ };
typedef struct {
    int kind;
    orc_prediction_mode pm;
    uint8_t btype, stride;
    const uint8_t *data; size_t len;
} orc_stream_command;
/* returns the number of bytes written, (size_t)-1 on failure */
size_t orc_stream_compress(const orc_stream_options *o, const orc_stream_command *cmds, size_t n_cmds, uint8_t *out, size_t cap);
size_t orc_stream_compress_raw(const orc_stream_options *o, const uint8_t *in, size_t n, uint8_t *out, size_t cap);
int orc_stream_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len);
int orc_lit_config_from_prediction_mode(const orc_stream_options *o, const orc_prediction_mode *pm, uint8_t btype, orc_lit_config *cfg);
int orc_mux_demux(const uint8_t *in, size_t n, uint8_t *s0, size_t *n0, uint8_t *s1, size_t *n1, size_t *consumed);

/* ---- walking a CMD-coder stream the REFERENCE wrote (the pin on wasm/wasm.html:98-107, tests/test_reference_vectors.py) ----
 * Decodes every command of a CMD stream -- PredictionMode, the three block switches, Literal lengths, Copy (codec/copy.rs),
 * Dict (codec/dict.rs: nibbles only, no dictionary), end marker -- with the oracle's own CDF / rANS / prior-table code and
 * reports what it saw.  rANS is an exact inverse, so a walk that used the right model for EVERY nibble ends with both decoder
 * states at the encoder's start state 2^31 and all bytes consumed; a single wrong (start, freq) anywhere cannot.
 * `wire`: ORC_WIRE_HEAD = /root/reference as it stands; ORC_WIRE_WASM_EXAMPLE = the older build that wrote the example, which
 * differs in two prior-table rows of the PredictionMode command (stream.c code_prediction_mode). */
enum { ORC_WIRE_HEAD = 0, ORC_WIRE_WASM_EXAMPLE = 1 };
typedef struct { uint8_t kind; uint8_t a, b; uint32_t x, y; } orc_walk_command; /* kind = command nibble (15 = end);
    Literal: x = length.  Copy: x = distance, y = length.  Dict: a = word size, b = transform, x = word id.
    Block switch: a = block type, b = stride (literal switch only). */
typedef struct {
    orc_walk_command cmds[64]; uint32_t n_cmds;
    uint32_t nibbles;                 /* symbols decoded */
    uint64_t state_a, state_b;        /* decoder states after the last symbol */
    size_t consumed;                  /* bytes of the stream the decoder took */
    int starved;
    orc_prediction_mode_result pm;    /* of the last PredictionMode command (pointers into the walk's own storage: copied below) */
    uint8_t literal_context_map_nonzero, mixing_value_min, mixing_value_max;
} orc_cmd_walk;
/* returns 0 when the walk reached the end marker, <0 where it stopped (-10 unknown command nibble, -11 more than 64 commands, -5 starved) */
int orc_cmd_stream_walk(const uint8_t *cmd, size_t n, int wire, orc_cmd_walk *w);

/* ---- CRC-32C, src/codec/crc32.rs ---- */
uint32_t orc_crc32c_update(uint32_t crc, const uint8_t *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif
