/*
 * divans_ffi.h -- per-stream C ABI, a drop-in for the reference's c/divans/ffi.h:6-66
 * (Rust side: src/ffi/mod.rs, src/ffi/interface.rs, src/ffi/compressor.rs, src/ffi/decompressor.rs).
 * Same names, argument meaning and return codes.  The literal bytes of the stream are coded by the
 * MI355X kernels (include/divans_gpu.h); command stream, Mux, header and CRC trailer are host code.
 *
 * Scope: the literal-only internal compressor, i.e. what the reference does with
 * DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION = 0 (src/ffi/compressor.rs:73-78,168-178).  Option values that
 * select the brotli front end (1, 2 = the reference's default) are accepted and code the input with the same
 * internal command selection: a valid .divans stream, larger than a brotli-assisted one (brotli command
 * generation is out of scope).  There is no CPU fallback for the literal coder: without a HIP device the first
 * divans_encode() / divans_decode() that needs it returns DIVANS_FAILURE.
 * The compressor works call by call as the reference's does (src/divans_compressor.rs:276-426): input goes into a ring of 2^window
 * bytes, a lap is coded inside the divans_encode() call that completes it (its literals on the GPU, resuming the stream's model),
 * container bytes leave in that call as far as the Mux releases them, and divans_encode() returns DIVANS_NEEDS_MORE_OUTPUT with part
 * of the input untaken exactly where the reference does.  The decompressor works the same way (src/divans_decompressor.rs:356-397):
 * it demultiplexes the container as it arrives, reads the CMD coder as far as its bytes reach and decodes the literals one or two
 * 65 536-symbol chunks at a time on the GPU as soon as the commands read so far cover them, handing the bytes out in the same call;
 * it accepts literal-only streams (one PredictionMode before the first Literal).
 * examples/divans_io.hpp wraps this ABI in the reference's writer / reader adaptors (src/writer.rs, src/reader.rs).
 */
#ifndef DIVANS_FFI_H_
#define DIVANS_FFI_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t DivansResult;
#define DIVANS_SUCCESS ((uint8_t)0)
#define DIVANS_NEEDS_MORE_INPUT ((uint8_t)1)
#define DIVANS_NEEDS_MORE_OUTPUT ((uint8_t)2)
#define DIVANS_FAILURE ((uint8_t)3)

typedef uint8_t DivansOptionSelect;          /* src/ffi/interface.rs:16-37 */
#define DIVANS_OPTION_QUALITY 1
#define DIVANS_OPTION_WINDOW_SIZE 2
#define DIVANS_OPTION_LGBLOCK 3
#define DIVANS_OPTION_DYNAMIC_CONTEXT_MIXING 4
#define DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION 5
#define DIVANS_OPTION_USE_BROTLI_BITSTREAM 6
#define DIVANS_OPTION_USE_CONTEXT_MAP 7
#define DIVANS_OPTION_LITERAL_ADAPTATION_CM_HIGH 8
#define DIVANS_OPTION_FORCE_STRIDE_VALUE 9
#define DIVANS_OPTION_STRIDE_DETECTION_QUALITY 10
#define DIVANS_OPTION_PRIOR_DEPTH 11
#define DIVANS_OPTION_LITERAL_ADAPTATION_STRIDE_HIGH 12
#define DIVANS_OPTION_LITERAL_ADAPTATION_CM_LOW 13
#define DIVANS_OPTION_LITERAL_ADAPTATION_STRIDE_LOW 14
#define DIVANS_OPTION_BROTLI_LITERAL_BYTE_SCORE 15
#define DIVANS_OPTION_SPEED_DETECTION_QUALITY 16
#define DIVANS_OPTION_PRIOR_BITMASK_DETECTION 17
#define DIVANS_OPTION_Q9_5 18
#define DIVANS_OPTION_FORCE_LITERAL_CONTEXT_MODE 19
#define DIVANS_OPTION_IR_OPTIMIZER 20        /* present in src/ffi/interface.rs:37, missing from c/divans/ffi.h */

/* src/ffi/interface.rs:40-47: all three NULL => libc malloc/free */
struct CAllocator {
    void *(*alloc_func)(void *opaque, size_t length);
    void (*free_func)(void *opaque, void *mfd);
    void *opaque;
};
struct DivansDecompressorState;
struct DivansCompressorState;

struct DivansCompressorState *divans_new_compressor(void);                                           /* mod.rs:18-24 */
struct DivansCompressorState *divans_new_compressor_with_custom_alloc(struct CAllocator alloc);       /* mod.rs:27-52 */
DivansResult divans_set_option(struct DivansCompressorState *state, DivansOptionSelect selector, uint32_t value); /* mod.rs:58-67 */
DivansResult divans_encode(struct DivansCompressorState *state, const uint8_t *input_buf_ptr, size_t input_size,
                           size_t *input_offset, uint8_t *output_buf_ptr, size_t output_size, size_t *output_offset); /* mod.rs:70-91 */
DivansResult divans_encode_flush(struct DivansCompressorState *state, uint8_t *output_buf_ptr, size_t output_size,
                                 size_t *output_offset);                                              /* mod.rs:94-108 */
void divans_free_compressor(struct DivansCompressorState *mfd);                                       /* mod.rs:159-169 */
/* Extension (not in c/divans/ffi.h): 1 when the state's options select the brotli front end -- DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION
 * 1 (the reference's default) or 2 -- which this library replaces by the internal command selection (see the scope note above):
 * the caller gets a valid .divans stream, not the one the reference would have produced.  0 after option 5 was set to 0. */
uint8_t divans_compressor_uses_internal_command_selection_instead_of_brotli(const struct DivansCompressorState *state);

struct DivansDecompressorState *divans_new_decompressor(void);                                        /* mod.rs:178-187 */
struct DivansDecompressorState *divans_new_serial_decompressor(void);                                 /* mod.rs:189-198 */
/* The Rust takes (alloc, skip_crc, multithread) (mod.rs:213-232) although c/divans/ffi.h:61 declares two
 * arguments; c/example.c (built against that header) passes two, so the third arrives as register garbage:
 * it is ignored here (there is no worker thread to start -- the literal stream is decoded on the GPU). */
struct DivansDecompressorState *divans_new_decompressor_with_custom_alloc(struct CAllocator alloc, uint8_t skip_crc,
                                                                          uint8_t multithread);
DivansResult divans_decode(struct DivansDecompressorState *state, const uint8_t *input_buf_ptr, size_t input_size,
                           size_t *input_offset, uint8_t *output_buf_ptr, size_t output_size, size_t *output_offset); /* mod.rs:236-262 */
void divans_free_decompressor(struct DivansDecompressorState *mfd);                                   /* mod.rs:312-323 */
/* Extension (not in the reference): the decoded size is the stream's own claim (literal lengths on the CMD coder); a
 * stream that claims more than `max_bytes` (default 1 GiB) is refused with DIVANS_FAILURE before anything is allocated. */
void divans_decompressor_set_max_output_size(struct DivansDecompressorState *state, size_t max_bytes);

/* helper allocators through the state's CAllocator, src/ffi/mod.rs:111-145,276-309 */
uint8_t *divans_compressor_malloc_u8(struct DivansCompressorState *state, size_t size);
void divans_compressor_free_u8(struct DivansCompressorState *state, uint8_t *data, size_t size);
size_t *divans_compressor_malloc_usize(struct DivansCompressorState *state, size_t size);
void divans_compressor_free_usize(struct DivansCompressorState *state, size_t *data, size_t size);
uint8_t *divans_decompressor_malloc_u8(struct DivansDecompressorState *state, size_t size);
void divans_decompressor_free_u8(struct DivansDecompressorState *state, uint8_t *data, size_t size);
size_t *divans_decompressor_malloc_usize(struct DivansDecompressorState *state, size_t size);
void divans_decompressor_free_usize(struct DivansDecompressorState *state, size_t *data, size_t size);

#ifdef __cplusplus
}
#endif
#endif
