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
 * the second.  Decompression cuts the batch into slices in stream order, FOUR in flight at once, each on its own HIP stream
 * with its own codec and page-locked staging buffers -- a decoded stream is a serial chain of tens of milliseconds however few
 * of them run, so the slices' kernels run side by side (four: more HIP streams than the runtime's four hardware queues take turns,
 * and a slice that waits behind another costs a whole chain) -- while the host threads parse the next slice and copy out the
 * finished ones.  Device memory is sized per slice from
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
    int32_t device;                        /* HIP device, or DIVANS_BATCH_ALL_DEVICES: the call shards its streams over every visible device (below) */
    int32_t host_threads;                  /* threads for the CMD coders and the framing; 0 = what the process is granted (affinity, cgroup quota).
                                              With DIVANS_BATCH_ALL_DEVICES: the total, divided evenly between the devices' driving threads */
    uint8_t skip_crc;                      /* decompress only */
} divans_batch_options;
void divans_batch_options_default(divans_batch_options *o);
/* divans_batch_options::device = DIVANS_BATCH_ALL_DEVICES: ONE call drives every visible HIP device (north_star: "work shards naturally by
 * metablock").  The n streams are cut into contiguous ranges, range r of D devices = [n * r / D, n * (r + 1) / D)
 * (D = min(devices, n)); each range runs on its own device through that device's lanes, driven by its own thread inside the call,
 * with its share of the host threads; outputs, offsets and sizes come back in stream order exactly as from a one-device call -- the
 * containers / payloads are bit-identical to D calls on the D ranges, and to one call on one device.  No data moves between devices.
 * divans_batch_timing then describes the slowest device's share (total_ms: the whole call).  A failure names the device and the range. */
#define DIVANS_BATCH_ALL_DEVICES (-1)
/* The same, but through the sharded code path even when there is one device (where DIVANS_BATCH_ALL_DEVICES simply makes the one-device
 * call): for tests and measurements of that path on a one-GPU box. */
#define DIVANS_BATCH_ALL_DEVICES_SHARDED (-2)

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
 * between calls -- creating them costs more than coding a few thousand streams -- and, per device, the host vectors the containers of a
 * compress call are assembled in (about the size of the largest batch's containers).  There is one set of lanes PER DEVICE: calls that
 * name different devices (divans_batch_options::device) run concurrently from different host threads -- one process drives all of a
 * node's GPUs this way, like independent states of the reference (src/ffi/interface.rs:49-50) -- and calls on one device take turns.
 * Naming another device tears nothing down.  divans_batch_release returns every device's lanes, divans_batch_release_device one
 * device's (both wait for a call that is running there); the next call builds them again. */
void divans_batch_release(void);
void divans_batch_release_device(int device);

/* Diagnostic: where the calling thread's time went in the last batch call, milliseconds: out[0] CMD coders (plans / container
 * parsing), [1] staging into page-locked memory + enqueueing, [2] waiting for the GPU, [3] container assembly / copy-out,
 * [4] final gather of the containers (compress only); parts of [1], compress: [5] reserving the lane's buffers, [6] the copy of the inputs
 * into page-locked memory (the rest of [1] is enqueueing copies and launches); decompress: [5] set-up before the first parse.  Per calling
 * thread (with DIVANS_BATCH_ALL_DEVICES: the slowest device's thread); overwritten by that thread's next call. */
void divans_batch_last_phases(double *out, int n);

/* Why does (or does not) this library take a container?  Host only, no GPU work: header, Mux framing, end marker, CRC-32C
 * trailer (src/codec/mod.rs:518-554), then the CMD coder's commands (src/codec/mod.rs:652-792) as far as this library decodes them.
 * `wire`: DIVANS_WIRE_HEAD = the reference tree as it stands; DIVANS_WIRE_WASM_EXAMPLE = the older build that wrote the one
 * compressed vector the tree holds (wasm/wasm.html:98-107), which coded the context-map mnemonics and the mixing values of the
 * PredictionMode command under other prior-table rows -- the probe reads both, the decoders read HEAD. */
enum { DIVANS_WIRE_HEAD = 0, DIVANS_WIRE_WASM_EXAMPLE = 1 };
typedef struct divans_container_probe {
    int32_t status;               /* 0 walked to the end marker (a stream divans_decode takes); 1 input ends early; 2 damaged (framing, trailer,
                                     CRC or CMD coder); 3 stopped at a command outside the literal-only scope (stopped_at_command) */
    uint8_t window;               /* header byte 5 */
    uint8_t crc_ok;               /* CRC-32C of header + body == the trailer's */
    uint8_t have_prediction_mode;
    uint8_t stopped_at_command;   /* the command nibble at which status 3 stopped: 1 Copy, 2 Dict, 5 / 6 command / distance block switch, 7 a second PredictionMode */
    uint32_t cmd_bytes, lit_bytes;/* the two demultiplexed coder streams */
    uint32_t commands;            /* commands decoded before the stop, the end marker not counted */
    uint32_t cmd_nibbles;         /* CMD-coder symbols decoded, the stopping command nibble included */
    uint32_t first_literal_length;
    uint64_t literal_bytes;       /* sum of the Literal lengths decoded */
    divans_lit_config cfg;        /* what the LIT coder runs under after the commands read (divans_gpu.h) */
} divans_container_probe;
int divans_probe_container(const uint8_t *in, size_t n, int wire, divans_container_probe *out);

#ifdef __cplusplus
}
#endif
#endif
