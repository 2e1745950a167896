/*
 * divans_batch.h -- many complete .divans streams per call, with the two halves of a stream overlapped the way the
 * reference's two-thread decoder overlaps them (src/parallel_decompressor.rs:55-141, src/threading.rs:88-100: a worker
 * thread decodes the CMD stream while the main thread decodes the LIT stream): here the LIT coder of EVERY stream of the
 * batch runs on the GPU in one launch sequence while host threads run the CMD coders (command types, PredictionMode,
 * literal lengths, src/codec/mod.rs:652-792) and the Mux / header / CRC framing (src/mux.rs, src/codec/mod.rs:409-560).
 * SURVEY.md section 8 row f4.  Streams are literal-only (the internal command selection, raw_to_cmd/mod.rs:105-181); every
 * container is byte-identical to what divans_encode / divans_encode_flush of include/divans_ffi.h produce for the same
 * input handed over in one divans_encode call.
 *
 * How a call runs: the streams are binned into length classes (<= 64 KiB, then powers of two).  Compression cuts every class
 * into two slices (the encoder passes fill the GPU whatever the slice size): the host assembles the first while the GPU codes
 * the second.  Decompression cuts the batch into eight slices in stream order, all in flight at once, each on its own HIP stream
 * with its own codec and page-locked staging buffers -- a decoded stream is a serial chain of tens of milliseconds however few
 * of them run, so the slices' kernels run side by side -- while the host threads parse the next slice and copy out the finished
 * ones.  Device memory is sized per slice from
 * its class bound -- roughly 64 bytes per byte of the bound and stream, never more than a sixteenth of the free device memory per
 * slice -- so a long stream among many short ones only costs its own slice.  A single stream whose class does not fit that
 * budget fails with DIVANS_GPU_ENOMEM.
 */
#ifndef DIVANS_BATCH_H_
#define DIVANS_BATCH_H_
#include <stddef.h>
#include <stdint.h>

#include "divans_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct divans_batch_options {      /* DivansCompressorOptions fields the internal compressor reads, src/interface.rs:444-484 */
    int32_t window_size;                   /* default 22 */
    uint8_t dynamic_context_mixing;        /* default 1 */
    uint8_t use_context_map;               /* default 1 */
    uint8_t force_stride;                  /* 0..8, 9 = UseBrotliRec (default) */
    uint8_t has_prior_depth, prior_depth;
    uint8_t has_literal_adaptation;
    divans_speed literal_adaptation[4];
    uint32_t call_buffer_size;             /* output buffer a per-stream caller would pass to each call (Mux slicing); default 65536 */
    int32_t device;                        /* HIP device */
    int32_t host_threads;                  /* threads for the CMD coders and the framing; 0 = hardware concurrency */
    uint8_t skip_crc;                      /* decompress only */
} divans_batch_options;
void divans_batch_options_default(divans_batch_options *o);

typedef struct divans_batch_timing {       /* milliseconds, wall clock */
    double total_ms;
    double gpu_ms;            /* first enqueue .. last slice finished, as seen from the host (launches, copies, kernels of all slices) */
    double host_overlapped_ms;/* host work done while at least one slice was in flight on the GPU: staging, CMD coders, Mux replay +
                                 framing of finished slices (compress) / parsing, staging, output copies (decompress) */
    double host_serial_ms;    /* host work with nothing in flight: the first slice's staging / parsing, the last slice's assembly or
                                 copy-out, the final gather of the containers */
} divans_batch_timing;

/* upper bound of the container size for an n-byte input */
size_t divans_batch_compress_bound(size_t n);

/* n_streams inputs -> n_streams containers written back to back into `out` (out_offsets[i], out_sizes[i]); `timing` may be NULL.
 * 0 on success, a DIVANS_GPU_E* code otherwise (divans_gpu_last_error()). */
int divans_batch_compress(const divans_batch_options *opt, const uint8_t *const *inputs, const size_t *sizes, size_t n_streams,
                          uint8_t *out, size_t out_cap, size_t *out_offsets, size_t *out_sizes, divans_batch_timing *timing);

/* n_streams complete containers -> their payloads back to back in `out`.  Streams whose PredictionMode / block type differ
 * are grouped and decoded group by group.  Returns DIVANS_GPU_ECORRUPT when a container fails its CRC, framing or the LIT
 * decoder's integrity check; out_sizes[i] of the first bad stream found is set to (size_t)-1 (for an integrity failure: the
 * first flagged stream of the slice that was being completed; streams of earlier slices have been written to `out` by then). */
int divans_batch_decompress(const divans_batch_options *opt, const uint8_t *const *containers, const size_t *sizes, size_t n_streams,
                            uint8_t *out, size_t out_cap, size_t *out_offsets, size_t *out_sizes, divans_batch_timing *timing);

/* The batch calls keep their eight lanes (HIP streams, codecs with their tables and scratch, page-locked staging buffers) alive
 * between calls -- creating them costs more than coding a few thousand streams -- and run one call at a time per process.
 * This returns everything; the next call builds the lanes again. */
void divans_batch_release(void);

#ifdef __cplusplus
}
#endif
#endif
