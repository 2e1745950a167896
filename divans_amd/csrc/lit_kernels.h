// lit_kernels.h -- shared between the HIP kernels and the C-ABI host code (not a public header).
#ifndef DIVANS_LIT_KERNELS_H_
#define DIVANS_LIT_KERNELS_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace divans_hip {

constexpr int LIT_THREADS = 256;    // 4 waves = 16 streams per workgroup
constexpr int RANS_THREADS = 64;    // one wave, one lane per stream

// per-batch constant tables staged in LDS by every workgroup
constexpr uint32_t LIT_BLOB_LUT1CLASS = 0;  // class (<8) of literal_lut1[b]: index of its value among the distinct lut1 values
constexpr uint32_t LIT_BLOB_CTXF = 256;     // [block type][prev][class]: literal_context_map[(lut0[prev] | lut1 value) + 64*btype], literal.rs:97-115
constexpr uint32_t LIT_CTXF_BYTES = 2048;   // one block type's table; n_btypes of them follow each other
constexpr uint32_t LIT_MAX_BTYPES = 8;      // block types one codec keeps context tables for (LDS: 2 KB each)
// then mixing_mask[8192] at LitGeometry::mix_off = 256 + 2048 * n_btypes
constexpr uint32_t LIT_BLOB_MAX_BYTES = 256 + LIT_CTXF_BYTES * LIT_MAX_BTYPES + 8192;

// How the (3 x 256 x 256) prior cube of LiteralNibblePriors (codec/priors.rs:35-37) is compacted for
// one configuration: only the planes / context columns the configuration can reach are materialised.
struct LitGeometry {
    uint32_t nctx;        // number of context-map output values (columns of the high-nibble table)
    uint32_t low_width;   // 16, or 256 when some mixing value is 1 (index_c carries context bits)
    uint32_t plane0, plane1, plane2;  // compact plane index of t = (mm>>7)^(opt_1_f_mask>>2)
    uint32_t low_base;    // first row of the low-nibble table
    uint32_t cm_base;     // first row of LiteralCommandPriorsCM (FirstNibble then SecondNibble)
    uint32_t total_rows;  // rows per stream
    int32_t mm_uniform;   // the mixing value when every reachable entry is equal, else -1
    int32_t ctx_const;    // the context when the context map is constant, else -1
    int32_t inc0, lim0, inc1, lim1, inc2, lim2, inc3, lim3;  // literal_adaptation Speeds (scalars: no dynamic indexing of kernargs)
    uint32_t bt_first, n_btypes;   // context tables exist for literal block types [bt_first, bt_first + n_btypes)
    uint32_t mix_off;              // byte offset of mixing_mask inside the configuration blob
    uint32_t lut1_classes;         // distinct literal_lut1 values of the prediction mode (1 for LSB6 / MSB6: the context is a function of prev alone)
    uint32_t hs_classes;           // > 0: the high-nibble stride table holds [row slot of the lut1 class of prev_prev][prev] (that many slots) instead of
                                   // [ctx][prev] -- the rows a stride-1 configuration can reach; the slots are in LIT_BLOB_CTXF[prev][4 + class]
    uint32_t wrap_check;           // some speed lets a row total leave i16 (divans_gpu_speed_supported is false): streaming kernels without
                                   // row caches, which keep such a row recognisable and report a stream that codes with one (lit_kernels.hip blend_row)
};

// One Literal command of a general stream (codec/mod.rs:711-792): `len` literal bytes coded with the literal block type
// `btype` (BlockSwitchLiteral, codec/interface.rs:289-292) after last_8_literals has been reloaded from the ring buffer
// (the 8 output bytes before the command, newest in bits 56..63; codec/mod.rs:771-783).  Priors, Weights and the rANS
// coder run on across segments: a stream's literals are ONE LIT_CODER byte stream.
struct LitSegment { uint32_t len, btype, last8_lo, last8_hi; };

struct LitBatch {
    const uint8_t* blob;        // LIT_BLOB_BYTES of configuration tables
    LitGeometry geom;
    int16_t* tables;            // [resident groups][total_rows][16]
    uint32_t n_streams, stream_len, max_stream_len;
    // encode: literal bytes in, (start|freq<<16) out.  decode: coded bytes in, literal bytes out.
    const uint8_t* in; const uint64_t* in_offsets; const uint32_t* in_sizes;
    uint8_t* out; const uint64_t* out_offsets; const uint32_t* out_sizes;
    uint32_t* sf;               // encode only: [n_streams][2*max_stream_len]
    uint32_t cache_rows_high;   // rows of the per-stream LDS cache for high-nibble rows (power of two >= 16, or 0)
    uint32_t cache_rows_low;    // same for low-nibble rows (ignored when cache_unified)
    uint32_t cache_mode;        // 0 none, 1 unified (cache_rows_high rows serve both tables), 2 high-nibble rows only, 3 separate high / low
    uint32_t cache_bytes_per_wg;  // 16 * (rows_high + rows_low) * (32 + 2)
    const uint32_t* seg_begin;  // [n_streams + 1] first segment of every stream in `segs`, or null: each stream is one segment
    const LitSegment* segs;     //   with zero context and the block type the tables were built for
    uint32_t* status;           // device word: bit 0 = encoder saw an invalid (start,freq), bit 1 = decoder integrity check failed
    // lit_decode2.hip: four direct-mapped row caches per stream (high stride rows, high context-map rows, low stride rows, low
    // context-map rows), one byte each: log2(rows) + 1, 0 = that table is not cached; and the hash shift of each
    // (set = (row ^ (row >> shift)) & (rows - 1)).  cache_bytes_per_wg then covers the word rings too.
    uint32_t dm_log2, dm_shift;
    uint8_t* stream_bad;        // optional [n_streams]: set to 1 for every stream whose decode failed its integrity check
    // lit_model_encode_kernel, one stream coded piece by piece (divans_gpu_lit_stream_*): `resume` = this launch continues the
    // stream of the previous one -- its CDF tables are kept (no row cache: cache_mode 0), the two Weights objects come back from
    // `wstate` ([2][3] ints: model_weights[1], [0]) -- and every launch leaves them there; the history arrives as a segment's last8
    uint32_t resume; int32_t* wstate;
    // lit_decode_kernel under `resume` / `wstate` likewise (whole chunks per launch: a chunk starts from fresh rANS states, so nothing
    // else carries over); `consumed` (optional, [n_streams]) receives the coded words a stream read, and with it set the words
    // offered may outnumber the words read (the caller passes what has arrived so far)
    uint32_t* consumed;
    // lit_decode2.hip, stride-1 instances: the order in which a table lays out the rows of the 256 previous-byte values -- a device
    // array of 256 ranks (a permutation: the codec learns it from the bytes it has seen, divans_gpu_codec_set_byte_order), or null =
    // numeric.  A private layout of the launch: any order decodes the same bytes.
    const uint8_t* byte_rank;
};
constexpr uint32_t LIT_STATUS_BAD_MODEL = 1u;     // rANS pass: freq == 0 or start/freq outside 15 bits
constexpr uint32_t LIT_STATUS_BAD_SEGMENT = 4u;   // a segment names a literal block type outside the codec's context tables
constexpr uint32_t LIT_STATUS_OUTPUT_FULL = 8u;   // divans_gpu_lit_encode_packed: the coded streams did not fit the caller's buffer
constexpr uint32_t LIT_STATUS_BAD_STREAM = 2u;    // decode: a chunk did not end with both states at 2^31, or the coded words were not consumed exactly

struct RansBatch {
    const uint32_t* sf; uint32_t n_streams, stream_len, max_stream_len; const uint32_t* in_sizes;
    uint32_t sf_stride;         // u32 elements per stream in sf (a multiple of 4): 2 * max_stream_len, or the work-array slot when the
                                // model pass left the pairs in place (BucketBatch::sfs / MixBucketBatch::xs[0])
    uint64_t out_base;          // added to the offsets written to out_offsets: `out` points at the first slot of a sub-batch
    uint8_t* out; uint64_t out_slot; uint64_t* out_offsets; uint32_t* out_sizes; uint32_t* status;
    uint32_t* chunk_bytes; uint32_t max_chunks;   // optional [n_streams][max_chunks] coded size of every 65 536-symbol chunk
    // chunk-parallel variant (streams of at most two chunks): chunk 0 is coded into scratch + (s + 1) * scratch_stride
    // (right-aligned) and then moved in front of chunk 1; null scratch selects the one-lane-per-stream kernel
    uint8_t* scratch; uint64_t scratch_stride; uint32_t* chunk0_sizes;
    uint32_t split_states;      // chunk-parallel variant: two lanes per chunk, one per rANS state (rans_encode2_split_kernel) -- for batches too small to fill the SIMDs
};

typedef unsigned int bk_u32x2 __attribute__((ext_vector_type(2)));
// Bucketed encoder model pass (lit_bucket.hip): mixing value 4, no mixing, context constant or a function of the previous byte.
struct BucketBatch {
    const uint8_t* in; const uint64_t* in_offsets; const uint32_t* in_sizes;
    uint32_t n_streams, stream_len, max_stream_len;
    uint32_t pieces;            // 8 KiB pieces per stream slot = ceil(max_stream_len / 8192), at most 8
    uint32_t slot;              // elements per stream in sorted / inv / sfs: pieces * 8192 (+ an optional pad)
    uint32_t sf_stride;         // u32 elements per stream in sf
    uint8_t* sorted;            // [n_streams][pieces * 8192] literal bytes, every piece ordered by (previous byte, position)
    uint16_t* inv;              // [n_streams][pieces * 8192] slot of a position inside its sorted piece
    uint32_t* desc;             // [n_streams][256 previous-byte values][8 pieces] first slot | count << 16
    bk_u32x2* sfs;              // [n_streams][pieces * 8192] (high, low) start|freq<<16 pairs in sorted order
    uint32_t* sf;               // [n_streams][2 * max_stream_len] the same pairs in position order (what rans_encode_kernel reads)
    uint32_t* tasks;            // [6 size classes][n_streams * 256] stream * 256 + previous byte
    uint32_t* counters;         // [0..5] tasks per class, [8] next unclaimed task
    int32_t inc, lim;           // literal_adaptation[0]
};
hipError_t launch_bucket_model(const BucketBatch& b, uint32_t chain_blocks, hipStream_t st);
void launch_bucket_tasks(const BucketBatch& b, hipStream_t st);
void launch_bucket_unsort(const BucketBatch& b, hipStream_t st);
void launch_bucket_unsort32(const BucketBatch& b, hipStream_t st);   // the same for a plane of 4-byte elements

// Bucketed encoder model pass for the two-model configuration (lit_bucket_mix.hip): context map on, every mixing value 4
// (stride 1), dynamic mixing (context_mixing >= 2), one literal block type, no segment lists, streams <= 64 KiB.
struct MixBucketBatch {
    const uint8_t* in; const uint64_t* in_offsets; const uint32_t* in_sizes;
    uint32_t n_streams, stream_len, max_stream_len;
    uint32_t pieces;            // 8 KiB pieces per stream slot, at most 8
    uint32_t slot;              // elements per stream in sorted / inv / the record planes (see BucketBatch::slot)
    uint32_t pos_stride;        // elements per stream in xs[] / maxes[]
    uint32_t sf_stride;         // u32 elements per stream in sf
    const uint8_t* blob;        // configuration tables (LIT_BLOB_LUT1CLASS, LIT_BLOB_CTXF of the one block type)
    uint16_t* sorted;           // [n_streams][pieces * 8192] byte | high-row slot << 8, every piece ordered by (key, position)
    uint16_t* inv;              // [n_streams][pieces * 8192] slot of a position inside its sorted piece
    uint32_t* desc;             // [n_streams][256 keys][8 pieces] first slot | count << 16
    uint32_t* tasks;            // [6 size classes][n_streams * 256]
    uint32_t* counters;
    // What the chains leave per position and model (0 stride, 1 context map), taken from the two rows BEFORE they are blended with the
    // position's nibbles: xs[m][n_streams][slot] = {high: cdf[sym] | cdf[sym-1] << 16, low: the same}, maxes[m][n_streams][slot] =
    // cdf[15] of the high row | of the low row << 16 -- 12 bytes, split so that each plane keeps power-of-two elements.  The chains
    // write them in sorted order; the unsort kernels put every piece back into position order in place (a whole piece sits in LDS
    // before it is written).
    bk_u32x2* xs[2];
    uint32_t* maxes[2];
    uint32_t* sf;               // [n_streams][sf_stride] what rans_encode_kernel reads; may be xs[0] (mix_weights_kernel reads a
                                // chunk of all four planes before it writes that chunk's pairs)
    int32_t inc0, lim0, inc2, lim2, inc3, lim3;   // literal_adaptation[0] (stride rows), [2] (cm low), [3] (cm high)
};
hipError_t launch_bucket_mix_model(const MixBucketBatch& b, uint32_t num_cus, hipStream_t st);

uint32_t lit_lds_bytes(const LitBatch& b);
hipError_t launch_model_encode(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st);
hipError_t launch_rans_encode(const RansBatch& b, hipStream_t st);
hipError_t launch_decode(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st);
hipError_t launch_decode2(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st);
void lit_decode_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap);    // the instances the two launchers pick, as rocprofv3 spells them
void lit_decode2_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap);
uint32_t lit_decode2_effective_caches(uint32_t dm_log2, bool mix, bool seg);   // the caches of dm_log2 a kernel instance exists for
uint32_t lit_decode2_stream_lds(uint32_t dm_log2);   // LDS bytes one stream takes in lit_decode2_kernel (word ring + row caches)
// Decoders that lost their measurement and are NOT part of the default library (build with DIVANS_WITH_EXPERIMENTAL_DECODERS=1 in the
// environment of divans_amd/build.py to have them): generation 4 = lit_decode_t.hip, one lane per stream, 30-45 % slower everywhere
// (profiles/r04e_lane_per_stream_decoder.txt), and the instances of generation 1's lit_decode_kernel no product path needs (the unified and
// split cache organisations; 3-8 % behind generation 3, profiles/r03b_bench_decoder_generations_same_box.txt).  Generation 1 without a
// cache / with the high-row cache stays: it decodes under speeds whose row totals leave i16 (wrap-checked) and call by call (resumable).
#ifndef DIVANS_WITH_EXPERIMENTAL_DECODERS
#define DIVANS_WITH_EXPERIMENTAL_DECODERS 0
#endif
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
// lit_decode_t.hip: one lane per stream (decoder generation 4); 64 streams per workgroup
hipError_t launch_decode_t(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st);
void lit_decode_t_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap);
uint32_t lit_decode_t_stream_lds(uint32_t dm_log2, bool mix);   // LDS bytes one stream takes there
#endif
hipError_t launch_pack(const uint8_t* slots, const uint64_t* src_off, const uint32_t* sizes, uint32_t n, uint8_t* packed,
                       uint64_t* dst_off, uint64_t* total, hipStream_t st, bool accumulate = false, uint64_t cap = ~0ull, uint32_t* status = nullptr);
// rank[b] = position of byte value b when the 256 values are ordered by how often they occur in a sample of the batch (the first
// `sample_len` bytes of `sample_streams` streams spread over it), ties by value: one workgroup, enqueued behind the launch that
// produced / consumed the bytes (lit_decode2.hip)
hipError_t launch_learn_byte_rank(const uint8_t* data, const uint64_t* offsets, const uint32_t* sizes, uint32_t n_streams, uint32_t stream_len,
                                  uint32_t sample_streams, uint32_t sample_len, uint8_t* rank, hipStream_t st);
// The decoder's row traffic without the decoder: the streams' bytes are KNOWN (b.in = the literal bytes, in_offsets / in_sizes theirs),
// every row the decoder of this configuration would touch is loaded, blended with the byte's nibble and stored through the same
// per-stream LDS caches, table layout and persistent grid -- no rANS state, no search, no division, nothing that makes one byte wait
// for the previous one except the rows themselves.  Its time is the memory side's own ceiling for this access stream (bench.py:
// roofline.request_ceiling).  Leaves the tables as a decode of the same bytes would; writes nothing else.
hipError_t launch_row_replay(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st);
hipError_t launch_selftest_division(unsigned long long* d_mismatches, hipStream_t st);
hipError_t launch_selftest_cdf_ops(const uint32_t* d_ops, uint32_t n, int32_t* d_out, hipStream_t st);

}  // namespace divans_hip
#endif
