/*
 * divans_gpu.h -- batch C ABI of the MI355X literal-stream coder (NEW entry points).
 *
 * The reference's per-stream C ABI (c/divans/ffi.h, mirrored in include/divans_ffi.h) feeds one
 * stream at a time through divans_encode()/divans_decode(); it cannot express "65 536 independent
 * streams in one launch".  These entry points are what a reference-side binding would call where
 * it today drives the literal coder:
 *   encode : LiteralState::encode_or_decode_content_bytes -> code_nibble_array
 *            (src/codec/literal.rs:404-494, 261-394) + ANSEncoder::put_nibble/flush_chunk
 *            (src/ans.rs:279-378)
 *   decode : the same generic function instantiated with ANSDecoder (src/ans.rs:225-252,428-442)
 * Each stream is an independent LIT_CODER byte stream (src/codec/interface.rs:48-50) with fresh
 * priors: exactly the bytes ANSEncoder hands to the Mux for stream 1, chunk after chunk.
 *
 * All buffers named d_* are DEVICE pointers (HBM); nothing here touches torch types.
 * Every function returns 0 on success, a negative DIVANS_GPU_E* code otherwise.
 */
#ifndef DIVANS_GPU_H_
#define DIVANS_GPU_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIVANS_GPU_EINVAL (-1)   /* bad argument / unsupported configuration */
#define DIVANS_GPU_ENOMEM (-2)   /* device allocation failed */
#define DIVANS_GPU_EHIP (-3)     /* HIP runtime error (see divans_gpu_last_error) */
#define DIVANS_GPU_ECAP (-4)     /* an output slot was too small */
#define DIVANS_GPU_ECORRUPT (-5) /* a coded stream failed the decoder's integrity check (host-buffer decode entry point) */

#define DIVANS_GPU_MAX_LITERAL_CONTEXT_MAP_SIZE (256 * 64) /* brotli MAX_LITERAL_CONTEXT_MAP_SIZE */
#define DIVANS_GPU_NUM_MIXING_VALUES 8192                  /* codec/interface.rs:137 */

/* Speed(inc, lim): src/probability/interface.rs:298-375 */
typedef struct divans_speed { int16_t inc, lim; } divans_speed;

/* What LiteralBookKeeping (src/codec/interface.rs:125-140) holds once the stream's PredictionMode
 * and BlockSwitchLiteral commands have been observed (obs_prediction_mode_context_map :285-319). */
typedef struct divans_lit_config {
    uint8_t literal_context_map[DIVANS_GPU_MAX_LITERAL_CONTEXT_MAP_SIZE];
    uint8_t mixing_mask[DIVANS_GPU_NUM_MIXING_VALUES];
    uint8_t prediction_mode;   /* LSB6=0 MSB6=1 UTF8=2 SIGN=3 (LiteralPredictionModeNibble) */
    uint8_t btype;             /* literal block type of the stream's BlockSwitchLiteral */
    uint8_t context_mixing;    /* dynamic_context_mixing; >1 selects MixingTrait (specializations.rs:27-47) */
    uint8_t reserved;
    divans_speed literal_adaptation[4]; /* [0..2) stride, [2..4) context map (codec/interface.rs:303-308) */
} divans_lit_config;

/* reference TestSimple (src/bin/benchmark.rs:195-206) and TestContextMixing via bench_no_ir (:156-167,305-343) */
void divans_lit_config_simple(divans_lit_config *cfg);
void divans_lit_config_context_mixing(divans_lit_config *cfg);

typedef struct divans_gpu_codec divans_gpu_codec;

/* Creates a codec bound to `device`.  All launches go to `hip_stream` (a hipStream_t, may be NULL for
 * the default stream).  `max_stream_len` bounds the byte length of any one stream in later calls. */
int divans_gpu_codec_create(divans_gpu_codec **out, const divans_lit_config *cfg, int device, void *hip_stream,
                            uint32_t max_stream_len);
void divans_gpu_codec_destroy(divans_gpu_codec *c);
const char *divans_gpu_last_error(void);

/* Worst-case coded bytes for an n-byte stream (16 B of state per 65 536-symbol chunk plus at most
 * one 32-bit renormalisation word per symbol... see DESIGN.md): size your output slots with this. */
size_t divans_gpu_lit_encode_bound(size_t n);

/* Encode n_streams independent streams.
 *   d_in + in_offsets[i] .. +in_sizes[i]   : literal bytes of stream i (d_in_offsets/d_in_sizes may be NULL:
 *                                            then stream i is d_in + i*stream_len, stream_len bytes)
 *   d_out + i*out_slot                     : output slot of stream i (out_slot % 16 == 0,
 *                                            out_slot >= divans_gpu_lit_encode_bound(len))
 * On return (stream-ordered): d_out_offsets[i] = byte offset in d_out where stream i's coded bytes
 * start (they are RIGHT-aligned in the slot because rANS emits last-byte-first), d_out_sizes[i] = length. */
int divans_gpu_lit_encode_batch(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                                const uint32_t *d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                uint8_t *d_out, uint64_t out_slot, uint64_t *d_out_offsets, uint32_t *d_out_sizes);

/* As divans_gpu_lit_encode_batch, additionally d_chunk_bytes[i * max_chunks + k] = coded bytes of the k-th 65 536-symbol
 * chunk of stream i (the unit ANSEncoder hands to the Mux, src/ans.rs:331-378; callers zero the array first). */
int divans_gpu_lit_encode_batch_chunks(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                                       const uint32_t *d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                       uint8_t *d_out, uint64_t out_slot, uint64_t *d_out_offsets, uint32_t *d_out_sizes,
                                       uint32_t *d_chunk_bytes, uint32_t max_chunks);

/* Encoder pass 1 only: the adaptive model's output before entropy coding.  d_pairs receives, for stream i at
 * [i * 2 * M, i * 2 * M + 2 * len_i) with M = the codec's max_stream_len rounded up to even, one word
 * (start | freq << 16) per nibble in coding order (high nibble first) -- what ANSEncoder::put_start_freq is handed,
 * ans.rs:287-301.  Same input conventions as divans_gpu_lit_encode_batch.  For tests and diagnostics. */
int divans_gpu_lit_model_batch(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                               const uint32_t *d_in_sizes, uint32_t stream_len, uint32_t n_streams, uint32_t *d_pairs);

/* Decode n_streams streams: coded bytes at d_in + d_in_offsets[i] (4-byte aligned), d_in_sizes[i] long;
 * stream i decodes to d_out + (d_out_offsets ? d_out_offsets[i] : i*stream_len), d_out_sizes[i] (or stream_len) bytes. */
int divans_gpu_lit_decode_batch(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                                const uint32_t *d_in_sizes, uint32_t n_streams, uint8_t *d_out,
                                const uint64_t *d_out_offsets, const uint32_t *d_out_sizes, uint32_t stream_len);

/* ---- general streams: literals of one stream split into segments ------------------------------------------------
 * In a stream with Copy / Dict commands the literal coder's 8-byte context (last_8_literals) is reloaded from the ring
 * buffer after every command (src/codec/mod.rs:771-783) and BlockSwitchLiteral commands change the literal block type
 * between Literal commands (src/codec/interface.rs:289-292), while the priors, the mixing Weights and the LIT rANS coder
 * run on: all literal bytes of the stream are ONE LIT_CODER byte stream.  One segment = one Literal command. */
typedef struct divans_lit_segment {
    uint32_t len;      /* literal bytes of this command (> 0) */
    uint32_t btype;    /* literal block type in force (< the count given to divans_gpu_codec_set_block_types) */
    uint64_t last8;    /* the 8 output bytes before the command, oldest in bits 0..7, newest in bits 56..63 */
} divans_lit_segment;
/* context tables for literal block types 0 .. n_btypes-1 (1..8) instead of the single divans_lit_config::btype.  A grid / cache
 * organisation / decoder generation chosen before with the tuning calls below is kept (clamped to what the LDS holds next to
 * the extra tables).  A segment naming a block type >= n_btypes raises DIVANS_GPU_STATUS_BAD_SEGMENT in the status word. */
int divans_gpu_codec_set_block_types(divans_gpu_codec *c, uint32_t n_btypes);
/* As divans_gpu_lit_encode_batch / _decode_batch; stream i's literal bytes (in command order, concatenated) are split
 * by the segments d_segs[d_seg_begin[i] .. d_seg_begin[i+1]) (device arrays; d_seg_begin has n_streams + 1 entries). */
int divans_gpu_lit_encode_segments_batch(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                                         const uint32_t *d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                         const uint32_t *d_seg_begin, const divans_lit_segment *d_segs,
                                         uint8_t *d_out, uint64_t out_slot, uint64_t *d_out_offsets, uint32_t *d_out_sizes);
int divans_gpu_lit_decode_segments_batch(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets,
                                         const uint32_t *d_in_sizes, uint32_t n_streams,
                                         const uint32_t *d_seg_begin, const divans_lit_segment *d_segs, uint8_t *d_out,
                                         const uint64_t *d_out_offsets, const uint32_t *d_out_sizes, uint32_t stream_len);

/* One stream coded piece by piece from host memory, with bounded memory: what the per-stream encoder of include/divans_ffi.h runs
 * for every Literal command the moment the reference would code it (src/divans_compressor.rs:276-337; the LIT coder of
 * src/codec/literal.rs:261-394 with its 65 536-symbol chunks, src/ans.rs:287-378).
 *   _begin   starts a stream on the codec (one workgroup, no row cache; the codec serves this stream until _finish).
 *   _encode  codes `len` more bytes (1 .. max_stream_len) whose predecessors end in `last8` (the ring buffer's last 8 bytes, byte 7 the
 *            most recent, as divans_lit_segment::last8).  `out` receives the bytes of every chunk these bytes COMPLETE, back to back,
 *            their sizes in chunk_sizes[0 .. *n_chunks); the symbols of the open chunk stay on the device.
 *   _finish  codes the open chunk (ANSEncoder::close); 0 bytes when the stream ended on a chunk boundary.
 * The concatenation of everything returned is the stream divans_gpu_lit_encode_host produces for the same bytes. */
int divans_gpu_lit_stream_begin(divans_gpu_codec *c);
int divans_gpu_lit_stream_encode(divans_gpu_codec *c, const uint8_t *in, uint32_t len, uint64_t last8, uint8_t *out, size_t out_cap,
                                 uint32_t *chunk_sizes, uint32_t max_chunks, uint32_t *n_chunks, size_t *out_len);
int divans_gpu_lit_stream_finish(divans_gpu_codec *c, uint8_t *out, size_t out_cap, size_t *out_len);
/* The other direction, whole 65 536-symbol chunks per call (a chunk starts from fresh rANS states: besides the CDF tables, the Weights
 * and the history nothing carries over).  `coded` = the LIT-coder bytes from the current chunk boundary on, as many as have arrived
 * (multiples of 4 are used); out_len = 32768 * k bytes of output, or fewer for the last chunk of the stream; *consumed_bytes = what those
 * chunks occupied.  The caller makes sure the chunks it asks for are complete -- divans_gpu_lit_encode_bound(32768) bytes per chunk
 * always are enough; DIVANS_GPU_ECORRUPT if a chunk does not end in the start states or reads past `coded_bytes`. */
int divans_gpu_lit_stream_decode_begin(divans_gpu_codec *c);
int divans_gpu_lit_stream_decode(divans_gpu_codec *c, const uint8_t *coded, size_t coded_bytes, uint32_t out_len, uint64_t last8,
                                 uint8_t *out, size_t *consumed_bytes);

/* Status of the device-pointer batch calls (they are asynchronous and return before the kernels ran).  Waits for the
 * codec's stream, stores the bits set since the previous call in *status and clears them:
 *   DIVANS_GPU_STATUS_BAD_MODEL  (1): an encode pass met a (start,freq) outside 15 bits / freq == 0 -- a CDF state the
 *                                     supported speeds cannot produce (the reference would debug_assert, ans.rs:305-309)
 *   DIVANS_GPU_STATUS_BAD_STREAM (2): a decoded stream did not end every 65 536-symbol chunk with both rANS states at
 *                                     2^31 or did not consume exactly its coded words: truncated / corrupt input, or coded
 *                                     under another configuration (the reference would stall on NeedsMoreInput, ans.rs:173-223).
 * The host-buffer wrappers check the same word themselves and return DIVANS_GPU_EINVAL / DIVANS_GPU_ECORRUPT. */
#define DIVANS_GPU_STATUS_BAD_MODEL 1u
#define DIVANS_GPU_STATUS_BAD_STREAM 2u
#define DIVANS_GPU_STATUS_BAD_SEGMENT 4u   /* a segment's literal block type lies outside the tables divans_gpu_codec_set_block_types built */
int divans_gpu_codec_status(divans_gpu_codec *c, uint32_t *status);
/* Clears the word without waiting (stream-ordered, before the launches that follow): for callers that keep a codec across calls
 * and may have abandoned a launch sequence without reading its status. */
int divans_gpu_codec_clear_status(divans_gpu_codec *c);
/* Enqueues a copy of the word into PAGE-LOCKED host memory behind the launches enqueued so far (no wait, no clear): read it once an
 * event recorded after this call has completed.  (divans_gpu_codec_status copies into pageable memory and synchronises the stream, which
 * costs ~10 ms per call while other streams keep the device busy.) */
int divans_gpu_codec_status_async(divans_gpu_codec *c, uint32_t *h_pinned_status);
/* Which stream: `d_flags` (device memory, >= n_streams bytes, zeroed by the caller; NULL = off) receives a 1 for every stream
 * of the following decode calls that fails that integrity check. */
int divans_gpu_codec_set_stream_flags(divans_gpu_codec *c, uint8_t *d_flags);

/* Compacts the right-aligned slots into one contiguous buffer (4-byte aligned starts):
 * d_packed_offsets[i] = exclusive prefix sum of round_up(sizes,4); returns total via *d_total (device u64). */
int divans_gpu_pack_streams(divans_gpu_codec *c, const uint8_t *d_slots, const uint64_t *d_offsets,
                            const uint32_t *d_sizes, uint32_t n_streams, uint8_t *d_packed,
                            uint64_t *d_packed_offsets, uint64_t *d_total);

/* Encode n_streams streams and leave them CONTIGUOUS in d_packed, stream after stream (4-byte aligned starts): d_packed_offsets[i],
 * d_sizes[i], *d_total = bytes used (device u64).  The work is done in sub-batches of `sub_batch` streams (0 = the codec's default,
 * 32768) whose right-aligned output slots and bucket work arrays the codec owns and reuses, so that a 65 536-stream batch holds the
 * slots and work arrays of half of it instead of divans_gpu_lit_encode_bound() per stream for all of them.
 * When the coded streams do not fit `packed_cap` the ones past the end are left out and DIVANS_GPU_STATUS_OUTPUT_FULL is raised
 * (offsets, sizes and *d_total still say what was needed: call again with that much).  Input conventions as divans_gpu_lit_encode_batch. */
#define DIVANS_GPU_STATUS_OUTPUT_FULL 8u
int divans_gpu_lit_encode_packed(divans_gpu_codec *c, const uint8_t *d_in, const uint64_t *d_in_offsets, const uint32_t *d_in_sizes,
                                 uint32_t stream_len, uint32_t n_streams, uint8_t *d_packed, uint64_t packed_cap,
                                 uint64_t *d_packed_offsets, uint32_t *d_sizes, uint64_t *d_total, uint32_t sub_batch);

/* Convenience wrappers over host memory (H2D, launch, D2H, synchronous).  out_offsets/out_sizes are host arrays. */
int divans_gpu_lit_encode_host(divans_gpu_codec *c, const uint8_t *in, uint32_t stream_len, uint32_t n_streams,
                               uint8_t *out_packed, size_t out_cap, uint64_t *out_offsets, uint32_t *out_sizes,
                               size_t *out_total);
/* same, additionally returning out_chunk_bytes[i*max_chunks + k] = coded bytes of the k-th 65 536-symbol chunk of stream i
 * (the unit ANSEncoder hands to the Mux, src/ans.rs:331-378) */
int divans_gpu_lit_encode_host_chunks(divans_gpu_codec *c, const uint8_t *in, uint32_t stream_len, uint32_t n_streams,
                                      uint8_t *out_packed, size_t out_cap, uint64_t *out_offsets, uint32_t *out_sizes,
                                      size_t *out_total, uint32_t *out_chunk_bytes, uint32_t max_chunks);
int divans_gpu_lit_decode_host(divans_gpu_codec *c, const uint8_t *in_packed, const uint64_t *in_offsets,
                               const uint32_t *in_sizes, uint32_t n_streams, uint8_t *out, uint32_t stream_len);

/* The same through a three-stage pipeline: the batch is cut into slices of `slice_streams` streams (0 = automatic: at
 * least the persistent grid's worth, at most a quarter of the batch) and slice i+1 travels to the device / slice i-1 back
 * while slice i is coded.  Results are identical to the wrappers above.  The copies only overlap with the kernels when
 * the caller's buffers are page-locked (divans_gpu_host_alloc below, hipHostMalloc or hipHostRegister); pageable memory
 * works but the runtime then serialises the copies.  The decode variant wants the coded streams in offset order (what the
 * encode wrappers produce) and otherwise falls back to divans_gpu_lit_decode_host. */
int divans_gpu_lit_encode_host_pipelined(divans_gpu_codec *c, const uint8_t *in, uint32_t stream_len, uint32_t n_streams,
                                         uint8_t *out_packed, size_t out_cap, uint64_t *out_offsets, uint32_t *out_sizes,
                                         size_t *out_total, uint32_t slice_streams);
int divans_gpu_lit_decode_host_pipelined(divans_gpu_codec *c, const uint8_t *in_packed, const uint64_t *in_offsets,
                                         const uint32_t *in_sizes, uint32_t n_streams, uint8_t *out, uint32_t stream_len,
                                         uint32_t slice_streams);
/* page-locked host memory for the wrappers above (hipHostMalloc / hipHostFree); NULL on failure */
void *divans_gpu_host_alloc(size_t bytes);
void divans_gpu_host_free(void *p);

/* Introspection for benchmarks/tests. */
typedef struct divans_gpu_info {
    uint32_t rows_per_stream;      /* 32-byte CDF rows held per in-flight stream */
    uint32_t resident_groups;      /* streams decoded concurrently (16-lane groups in the persistent grid) */
    uint32_t blocks, threads;      /* launch geometry of the model/decode kernels */
    uint64_t table_bytes;          /* HBM bytes of the CDF tables */
    uint64_t scratch_bytes;        /* HBM bytes of the encoder's work memory: bucketed-pass work arrays, start/freq spill, rANS chunk scratch */
    float last_model_ms, last_rans_ms, last_decode_ms; /* hipEvent timings of the last batch calls */
    float last_pack_ms;            /* the pack launches inside the last divans_gpu_lit_encode_packed call */
} divans_gpu_info;
int divans_gpu_codec_info(divans_gpu_codec *c, divans_gpu_info *info);
/* The kernel instance the last decode call launched, spelt as rocprofv3 --kernel-trace reports it
 * (e.g. "divans_hip::lit_decode2_kernel_w7<4, true, false, false, 17>"); "" before the first decode. */
int divans_gpu_codec_last_decode_kernel(divans_gpu_codec *c, char *buf, size_t cap);
/* tuning knobs: `blocks` = persistent grid of 256-thread workgroups (0 keeps the current value);
 * `cache_rows` = rows of one unified per-stream LDS row cache (0 = off, power of two in [16,256], 0xffffffff keeps). */
int divans_gpu_codec_set_geometry(divans_gpu_codec *c, uint32_t blocks, uint32_t cache_rows);
/* Encoder model pass: 0 = automatic, 1 = streaming kernels (one walk per stream against its CDF table in HBM),
 * 2 = bucketed (positions grouped by the byte / context that selects their rows, one lane per bucket, rows in LDS;
 * lit_bucket.hip, lit_bucket_mix.hip).  The bucketed pass exists for configurations whose every mixing value is 4
 * (stride 1) and streams of at most 65536 bytes: without mixing when the context is constant or follows from the previous
 * byte alone (divans_lit_config_simple; LSB6 / MSB6 prediction modes with any context map, i.e. what the literal-only
 * compressor emits), or with a context map and dynamic mixing for one literal block type (divans_lit_config_context_mixing).  There it is what
 * "automatic" picks; asking for it elsewhere is DIVANS_GPU_EINVAL.  Both produce the same bytes. */
int divans_gpu_codec_set_encode_path(divans_gpu_codec *c, uint32_t path);
/* Streams the bucketed two-model pass takes per launch sequence (default 32768, halved until its work arrays -- 1.9 MB
 * per 64 KiB stream -- fit the device).  A tuning / test knob: the coded bytes do not depend on it. */
int divans_gpu_codec_set_bucket_batch(divans_gpu_codec *c, uint32_t streams);

/* Tuning: which decode kernel runs -- generation 2 / 3 (lit_decode2.hip: LDS row caches read in one round trip, direct mapped (2)
 * or 2-way (3), coded words through an LDS ring) or 1 (lit_kernels.hip) -- and for generations 2 / 3 the rows of the four per-stream caches
 * {high stride rows, high context-map rows, low stride rows, low context-map rows} (0 = not cached, else a power of two in
 * [4, 256]), their hash shifts (set = (row ^ (row >> shift)) & (rows - 1)) and the persistent grid (0 = keep; clamped to
 * what the LDS holds).  rows / shifts may be NULL to keep the current ones.  Both generations produce the same bytes.
 * Generations 1 and 4 (one lane per stream, lit_decode_t.hip) measured 3-8 % and 30-45 % slower than 2 / 3 and are selectable only in
 * a library built with DIVANS_WITH_EXPERIMENTAL_DECODERS=1 (divans_gpu_experimental_decoders() != 0); the default library answers
 * DIVANS_GPU_EINVAL.  It still uses generation 1 by itself where no later generation runs (speeds whose row totals leave i16, the
 * call-by-call stream decoder); there divans_gpu_codec_set_geometry / _set_split_cache only shape the streaming encoder pass. */
int divans_gpu_codec_set_decoder(divans_gpu_codec *c, uint32_t generation, const uint32_t rows[4], const uint32_t shifts[4],
                                 uint32_t blocks);
/* The rANS pass of streams of at most two 65 536-symbol chunks: 1 = one lane per chunk, 2 = two lanes per chunk, one per rANS state
 * (ans.rs:302-329: the states alternate symbol by symbol and only share the output order), 0 = automatic (default): two lanes while
 * that still gives every SIMD at most one wave -- batches up to 64 streams per CU, where it halves the pass (a lane's chain is half as
 * long); larger batches fill the SIMDs with whole-chunk lanes and the total work is the same.  The coded bytes do not depend on it. */
int divans_gpu_codec_set_rans_split(divans_gpu_codec *c, uint32_t mode);
/* Stride-1 configurations (every mixing value 4): the order in which the decoder's private tables lay out the rows of the 256 previous-byte
 * values, i.e. which rows share a 128-byte line -- the byte values that are hot together should (on text -4..5 % decode time against numeric
 * order, profiles/r03c_byte_rank_layout_same_box.txt, r05_byte_order_ab.txt).
 *   0 (default) = LEARNED from the codec's own data: the first batch the codec sees -- the literal bytes of an encode call, or what a decode
 *       call produced -- is sampled on the device (64 streams x 2 KiB, one small kernel behind that call's launches, nothing synchronised)
 *       and the byte values are ranked by frequency; every later launch uses that rank.  Until then (a decode-only codec's first call) the
 *       order is numeric.  Calling this function with 0 again makes the next batch re-learn it.
 *   1 = numeric order.
 *   2 = a fixed English-text rank (lower-case letters by frequency, separators, capitals, digits, the rest numerically): a caller's hint,
 *       what rounds 3-5 used as the default.
 * The decoded bytes do not depend on it (a private layout of each launch). */
int divans_gpu_codec_set_byte_order(divans_gpu_codec *c, uint32_t order);
/* What is in force: *mode = the order set; *ready = 1 when a rank table is in use (mode 2, or mode 0 after the first batch);
 * rank256 (optional, 256 bytes) = rank of every byte value (identity while none is in use).  Synchronises the stream when rank256 is given. */
int divans_gpu_codec_byte_order(divans_gpu_codec *c, uint32_t *mode, uint32_t *ready, uint8_t *rank256);
/* Measurement aid: the memory side's own time for a batch's CDF-row traffic.  `d_literals` (+ offsets / sizes, or NULL for n_streams x stream_len
 * contiguous) are the streams' LITERAL bytes; every row the decoder of this configuration touches for them is loaded, blended with the byte's
 * nibble and stored through the same per-stream LDS caches, table layout, byte order and persistent grid divans_gpu_lit_decode_batch would use
 * for a batch of this size -- with no entropy decoding and nothing that makes a byte wait for the one before it (the low-nibble rows, which a
 * decoder can only name once the high nibble is decoded, are requested a byte ahead).  *ms = that launch's duration: what the decode
 * kernel would take if the memory system were its only limit.  Synchronous.  Stride-1 configurations (divans_lit_config_simple /
 * _context_mixing and the like), default cache organisation; DIVANS_GPU_EINVAL elsewhere. */
int divans_gpu_codec_row_replay(divans_gpu_codec *c, const uint8_t *d_literals, const uint64_t *d_offsets, const uint32_t *d_sizes,
                                uint32_t n_streams, uint32_t stream_len, float *ms);
/* 1 if this library was built with the decoders that lost their measurements (generation 4; generation 1 with unified / split caches) */
int divans_gpu_experimental_decoders(void);

/* Where the CDF tables' pages lie in device memory moves the decode time by up to 10-20 % from one allocation to the next (DESIGN.md
 * section 5; profiles/r04e_table_placement.txt), so the library measures placements -- alternately one hipMalloc block and 32 MiB chunks
 * mapped side by side; which kind is faster differs from box to box -- and keeps the fastest; the same bytes come out whatever the placement.
 * `candidates`:
 *   0 = the library's policy, the default: tables of 2 GiB and more (whole-GPU batches) are tried on 12 placements, ONE PER CALL -- each of
 *       the codec's first divans_gpu_lit_decode_batch calls whose batch fills at least half the persistent grid decodes ONCE, on one
 *       placement (a new candidate, or the best so far), and the next such call reads that launch's time before it picks its own.  No call
 *       decodes twice; a call waits at most for the decode launch of the PREVIOUS such call (never for its own), only while the search
 *       lasts (13 calls), and spends one allocation of the tables on the host.  Calls of another shape (stream count / length) than the
 *       first are decoded on the placement in use and not compared.  A caller that stops after fewer calls keeps what was in use.
 *       Smaller tables are not tuned.
 *   1 = off: the first allocation as it comes (the lanes of divans_batch_* use this).
 *   2..16 = the EAGER form: the first qualifying call decodes its batch that many times, once per placement, synchronises the stream
 *       and returns with the fastest in place -- seconds, for a caller that wants the search over before its first timed call.
 * Memory while a search runs (either form): two copies of the tables (the best so far and the candidate; a rejected copy is released before
 * the next is allocated).  Address space: every chunk-mapped candidate reserves a range that is never returned (the ROCm remap defect,
 * scripts/probes/README.md) -- see divans_gpu_table_memory.  The search starts again after the tables had to grow. */
int divans_gpu_codec_tune_tables(divans_gpu_codec *c, uint32_t candidates);
/* The call-by-call search (what `0` above gives tables of 2 GiB and more) with `candidates` placements whatever the tables' size:
 * 0 = the library's policy, 1 = off, 2..16 = that many, one per qualifying decode call. */
int divans_gpu_codec_search_tables(divans_gpu_codec *c, uint32_t candidates);
typedef struct divans_gpu_table_placement {
    uint32_t policy_candidates;    /* what a search tries at most (the library's policy or divans_gpu_codec_tune_tables); valid before the tables exist */
    uint32_t tried;                /* placements measured so far (0 = no search yet / none at all) */
    float first_ms, best_ms, worst_ms;   /* decode kernel time on the first placement, on the best so far (the one kept once the search is over), on the slowest seen */
    uint32_t kept_chunks;          /* 1 = the kept tables are chunks mapped into a reserved range, 0 = one hipMalloc block (once the search is over) */
    uint32_t searching;            /* 1 = the call-by-call search is still running: the next qualifying call tries another placement */
} divans_gpu_table_placement;
int divans_gpu_codec_table_placement(divans_gpu_codec *c, divans_gpu_table_placement *out);
/* Memory the library keeps beyond its codecs.  A destroyed codec's chunk-mapped tables (2 GiB and more) stay MAPPED -- at most two ranges
 * per process -- for the next codec that fits: idle_ranges / idle_bytes; divans_gpu_trim() gives that memory back now, and every other
 * device allocation of the library does so before it would fail.  Their ADDRESS RANGES are never handed back to the driver (on ROCm 7.2 a
 * range that is unmapped and mapped again reads and writes through stale translations, scripts/probes/README.md): va_reserved_bytes counts
 * them, and past va_cap_bytes (default 256 GiB, divans_gpu_set_table_va_cap) new tables are plain hipMalloc blocks, which hipFree returns
 * in full.  A process that creates and destroys big codecs for days therefore ends on the allocator's ordinary behaviour. */
typedef struct divans_gpu_table_memory_info { uint64_t va_reserved_bytes, va_cap_bytes, idle_bytes; uint32_t idle_ranges; } divans_gpu_table_memory_info;
int divans_gpu_table_memory(divans_gpu_table_memory_info *out);
void divans_gpu_set_table_va_cap(uint64_t bytes);
void divans_gpu_trim(void);

/* separate caches for the rows of the high-nibble and of the low-nibble table (0 = that table goes to HBM/L2 directly) */
int divans_gpu_codec_set_split_cache(divans_gpu_codec *c, uint32_t high_rows, uint32_t low_rows);

/* Exhaustive self-check of the reciprocal division used by the kernels against integer '/':
 * returns the number of mismatches over every (cdf<<15)/max with 1<=max<32768, 0<=cdf<=max. */
int divans_gpu_selftest_division(divans_gpu_codec *c, uint64_t *mismatches);

/* 1 when no count of a row can ever leave i16 under Speed(inc, lim): FrequentistCDF16::blend (src/probability/frequentist_cdf.rs:74-85)
 * adds inc to a row's total on every update and renormalises once it reached lim, so the total follows one trajectory from 64; such
 * speeds run on every kernel.  Host-only, needs no device. */
int divans_gpu_speed_supported(int32_t inc, int32_t lim);
/* 1 when divans_gpu_codec_create takes (inc, lim) at all: every i16 pair with inc >= 0 -- whatever the f8 pairs of a PredictionMode
 * command decode to (src/probability/interface.rs:577-585), short of a negative increment.  Under an accepted speed that is not
 * "supported" the reference's i16 row total wraps negative at some point, and the next nibble coded with that row is not a
 * distribution any more (src/ans.rs:281-285; its own decoder cannot read such a stream back).  A stream in which no wrapped row is
 * coded with again -- every short stream -- is coded bit for bit as the reference codes it; one in which that happens raises
 * DIVANS_GPU_STATUS_BAD_MODEL (encoding) / DIVANS_GPU_STATUS_BAD_STREAM and the per-stream flag (decoding).  These speeds run on the
 * streaming kernels without row caches: correct, not fast. */
int divans_gpu_speed_accepted(int32_t inc, int32_t lim);

/* Test entry points: device primitives in isolation, so that the reference's own unit tests can be run against them.
 * cdf_ops: a script of n_ops operations {kind, a, b, c} (4 x u32 each) on two CDF rows and one Weights object, one
 *   16 x i32 record per operation in `out`:  0/1 blend row 0/1 (a = symbol, b = inc, c = lim) -> the row;
 *   2 row0.average(row1, a) -> the mixed row;  3 sym_to_start_and_freq(row0, a) / 4 cdf_offset_to_sym_start_and_freq(row0, a)
 *   -> {start, freq, sym};  5 Weights::update([a, b], c) -> {w0, w1, normalized_weight as u16};  6 reset;  7 = 0 through
 *   the blend variant of the pipelined paths.
 * rans_pairs: the LIFO rANS pass on n_pairs (even) caller-supplied (start | freq << 16) words of one stream. */
int divans_gpu_selftest_cdf_ops(divans_gpu_codec *c, const uint32_t *ops, uint32_t n_ops, int32_t *out);
int divans_gpu_selftest_rans_pairs(divans_gpu_codec *c, const uint32_t *pairs, uint32_t n_pairs, uint8_t *out, size_t cap, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
