// ffi.cpp -- the reference's per-stream C ABI (include/divans_ffi.h <-> c/divans/ffi.h, src/ffi/*.rs) on top of the
// host stream layer + HIP literal coder.  Error behaviour follows src/ffi/mod.rs: NULL state/offset -> FAILURE,
// every internal error collapses to DIVANS_FAILURE, options only before the first encode (OptionStage).
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/divans_ffi.h"
#include "host_stream.h"

using divans_host::StreamOptions;

namespace {

void* state_alloc(const CAllocator& a, size_t n) { return a.alloc_func ? a.alloc_func(a.opaque, n) : std::malloc(n); }
void state_free(const CAllocator& a, void* p) { if (a.alloc_func) { if (a.free_func) a.free_func(a.opaque, p); } else std::free(p); }

// ENCODER_DEFAULT_PALETTE, src/probability/interface.rs:303-320
const divans_speed kPalette[15] = {{0, 1024}, {2, 1024}, {1, 128}, {1, 16384}, {2, 2048}, {4, 1024}, {8, 8192}, {16, 48},
                                   {16, 8192}, {32, 4096}, {64, 16384}, {128, 256}, {128, 16384}, {512, 16384}, {1664, 16384}};

}  // namespace

struct DivansCompressorState {
    CAllocator alloc;
    StreamOptions opt;
    bool started = false;        // CompressorState::OptionStage -> constructed compressor (ffi/compressor.rs:212-236)
    bool failed = false;
    divans_host::StreamEncoder* enc = nullptr;   // DivansCompressor: ring, CMD coder, Mux, GPU literal coder (host_stream.h)
    bool flushing = false;
    ~DivansCompressorState() { delete enc; }
};

struct DivansDecompressorState {
    CAllocator alloc;
    bool skip_crc = false;
    bool failed = false;
    size_t max_output = (size_t)1 << 30;   // bound on the decoded size a stream may claim (divans_decompressor_set_max_output_size)
    divans_host::StreamDecoder* dec = nullptr;   // DivansDecompressor: Mux, CMD coder, GPU literal decoder (host_stream.h); built by the first decode
    ~DivansDecompressorState() { delete dec; }
};

extern "C" {

struct DivansCompressorState* divans_new_compressor_with_custom_alloc(struct CAllocator alloc) {
    void* mem = state_alloc(alloc, sizeof(DivansCompressorState));
    if (!mem) return nullptr;
    DivansCompressorState* s = new (mem) DivansCompressorState();
    s->alloc = alloc;
    return s;
}
struct DivansCompressorState* divans_new_compressor(void) {
    CAllocator a = {nullptr, nullptr, nullptr};
    return divans_new_compressor_with_custom_alloc(a);
}

// src/ffi/compressor.rs:63-166
DivansResult divans_set_option(struct DivansCompressorState* s, DivansOptionSelect selector, uint32_t value) {
    if (!s || s->started) return DIVANS_FAILURE;
    StreamOptions& o = s->opt;
    auto set_speed = [&](int index) -> DivansResult {
        if (value >= 15) return DIVANS_FAILURE;
        if (!o.has_literal_adaptation) { o.has_literal_adaptation = true; for (auto& sp : o.literal_adaptation) sp = kPalette[value]; }
        else o.literal_adaptation[index] = kPalette[value];
        return DIVANS_SUCCESS;
    };
    switch (selector) {
    case DIVANS_OPTION_QUALITY: case DIVANS_OPTION_LGBLOCK: case DIVANS_OPTION_STRIDE_DETECTION_QUALITY:
    case DIVANS_OPTION_PRIOR_BITMASK_DETECTION: case DIVANS_OPTION_SPEED_DETECTION_QUALITY:
    case DIVANS_OPTION_BROTLI_LITERAL_BYTE_SCORE: case DIVANS_OPTION_Q9_5: case DIVANS_OPTION_FORCE_LITERAL_CONTEXT_MODE:
    case DIVANS_OPTION_IR_OPTIMIZER:
        return DIVANS_SUCCESS;   // brotli front-end knobs: stored by the reference, never read by the literal-only compressor
    case DIVANS_OPTION_WINDOW_SIZE: o.window_size = (int)value; return DIVANS_SUCCESS;
    case DIVANS_OPTION_DYNAMIC_CONTEXT_MIXING: o.dynamic_context_mixing = (uint8_t)value; return DIVANS_SUCCESS;
    case DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION: if (value > 2) return DIVANS_FAILURE; o.use_brotli = (int)value; return DIVANS_SUCCESS;
    case DIVANS_OPTION_USE_BROTLI_BITSTREAM: if (value != 1) return DIVANS_FAILURE; o.use_brotli = 2; return DIVANS_SUCCESS;
    case DIVANS_OPTION_USE_CONTEXT_MAP: if (value > 1) return DIVANS_FAILURE; o.use_context_map = value == 1; return DIVANS_SUCCESS;
    case DIVANS_OPTION_FORCE_STRIDE_VALUE: if (value > 8) return DIVANS_FAILURE; o.force_stride = (uint8_t)value; return DIVANS_SUCCESS;
    case DIVANS_OPTION_LITERAL_ADAPTATION_STRIDE_HIGH: return set_speed(1);
    case DIVANS_OPTION_LITERAL_ADAPTATION_CM_HIGH: return set_speed(3);
    case DIVANS_OPTION_LITERAL_ADAPTATION_STRIDE_LOW: return set_speed(0);
    case DIVANS_OPTION_LITERAL_ADAPTATION_CM_LOW: return set_speed(2);
    case DIVANS_OPTION_PRIOR_DEPTH: o.has_prior_depth = true; o.prior_depth = (uint8_t)value; return DIVANS_SUCCESS;
    default: return DIVANS_FAILURE;
    }
}

// Not in the reference's ABI: tells a caller that its settings asked for the brotli front end (the reference's default does) and
// that this library codes the stream with the internal command selection instead -- valid, decodable by the reference, larger.
uint8_t divans_compressor_uses_internal_command_selection_instead_of_brotli(const struct DivansCompressorState* s) {
    return s && s->opt.use_brotli != 0 ? 1 : 0;
}

static bool start(DivansCompressorState* s) {
    s->started = true;
    // BrotliCompressionSetting (src/ffi/compressor.rs:168-210): the reference's default (UseBrotliCommandSelection) and
    // UseBrotliBitstream run the brotli front-end to choose Copy / Dict / Literal commands.  That front-end is outside this
    // library; every setting codes the input with the internal command selection
    // (raw_to_cmd/mod.rs:105-181: one PredictionMode + Literal commands).  The result is a valid .divans stream any
    // reference decoder accepts -- larger than a brotli-assisted one -- so c/example.c, which sets no
    // options, round-trips through this library unchanged.
    if (s->opt.dynamic_context_mixing >= 15) s->failed = true;   // codec/interface.rs:359 assert
    // (the object's members allocate too -- the 2^window ring, vectors: no exception may leave an extern "C" function)
    if (!s->failed) { try { s->enc = new divans_host::StreamEncoder(s->opt, 0); } catch (...) { s->enc = nullptr; } }
    if (!s->enc) s->failed = true;
    return !s->failed;
}

// src/ffi/mod.rs:70-91 + divans_compressor.rs:276-337: input goes into the 2^window ring; whenever it fills, its commands are coded
// and container bytes leave in THIS call, as far as the Mux lets them and `out` has room (DIVANS_NEEDS_MORE_OUTPUT: call again
// with the rest of the input and an empty buffer)
DivansResult divans_encode(struct DivansCompressorState* s, const uint8_t* in, size_t in_size, size_t* in_off,
                           uint8_t* out, size_t out_size, size_t* out_off) {
    if (!s || !in_off || !out_off || *in_off > in_size || *out_off > out_size) return DIVANS_FAILURE;
    if (!s->started && !start(s)) return DIVANS_FAILURE;
    if (s->failed || s->flushing) return DIVANS_FAILURE;       // NotAllowedToEncodeAfterFlush
    int rc;
    try { rc = s->enc->encode(in, in_size, in_off, out, out_size, out_off); } catch (...) { rc = -1; }
    if (rc < 0) { s->failed = true; return DIVANS_FAILURE; }
    return rc == 1 ? DIVANS_NEEDS_MORE_OUTPUT : DIVANS_NEEDS_MORE_INPUT;
}

// src/ffi/mod.rs:94-108 + divans_compressor.rs:363-426
DivansResult divans_encode_flush(struct DivansCompressorState* s, uint8_t* out, size_t out_size, size_t* out_off) {
    if (!s || !out_off || *out_off > out_size) return DIVANS_FAILURE;
    if (!s->started && !start(s)) return DIVANS_FAILURE;
    if (s->failed) return DIVANS_FAILURE;
    s->flushing = true;
    int rc;
    try { rc = s->enc->flush(out, out_size, out_off); } catch (...) { rc = -1; }
    if (rc < 0) { s->failed = true; return DIVANS_FAILURE; }
    return rc == 1 ? DIVANS_NEEDS_MORE_OUTPUT : DIVANS_SUCCESS;
}

void divans_free_compressor(struct DivansCompressorState* s) {
    if (!s) return;
    const CAllocator a = s->alloc;
    s->~DivansCompressorState();
    state_free(a, s);
}

struct DivansDecompressorState* divans_new_decompressor_with_custom_alloc(struct CAllocator alloc, uint8_t skip_crc, uint8_t multithread) {
    (void)multithread;
    void* mem = state_alloc(alloc, sizeof(DivansDecompressorState));
    if (!mem) return nullptr;
    DivansDecompressorState* s = new (mem) DivansDecompressorState();
    s->alloc = alloc;
    s->skip_crc = skip_crc != 0;
    return s;
}
struct DivansDecompressorState* divans_new_decompressor(void) {
    CAllocator a = {nullptr, nullptr, nullptr};
    return divans_new_decompressor_with_custom_alloc(a, 0, 1);
}
struct DivansDecompressorState* divans_new_serial_decompressor(void) {
    CAllocator a = {nullptr, nullptr, nullptr};
    return divans_new_decompressor_with_custom_alloc(a, 0, 0);
}

// src/ffi/mod.rs:236-262 + divans_decompressor.rs:356-397
DivansResult divans_decode(struct DivansDecompressorState* s, const uint8_t* in, size_t in_size, size_t* in_off,
                           uint8_t* out, size_t out_size, size_t* out_off) {
    if (!s || !in_off || !out_off || *in_off > in_size || *out_off > out_size) return DIVANS_FAILURE;
    if (s->failed) return DIVANS_FAILURE;
    if (!s->dec) {
        try { s->dec = new divans_host::StreamDecoder(s->skip_crc, s->max_output, 0); } catch (...) { s->dec = nullptr; }
        if (!s->dec) { s->failed = true; return DIVANS_FAILURE; }
    }
    int rc;
    try { rc = s->dec->decode(in, in_size, in_off, out, out_size, out_off); } catch (...) { rc = -1; }
    if (rc < 0) { s->failed = true; return DIVANS_FAILURE; }
    return rc == 0 ? DIVANS_SUCCESS : (rc == 1 ? DIVANS_NEEDS_MORE_INPUT : DIVANS_NEEDS_MORE_OUTPUT);
}

// extension (not in c/divans/ffi.h): the literal lengths of a stream are its own claim; refuse streams that claim more
void divans_decompressor_set_max_output_size(struct DivansDecompressorState* s, size_t max_bytes) { if (s) s->max_output = max_bytes; }

void divans_free_decompressor(struct DivansDecompressorState* s) {
    if (!s) return;
    const CAllocator a = s->alloc;
    s->~DivansDecompressorState();
    state_free(a, s);
}

// src/ffi/mod.rs:111-145,276-309
uint8_t* divans_compressor_malloc_u8(struct DivansCompressorState* s, size_t n) { return (uint8_t*)state_alloc(s->alloc, n); }
void divans_compressor_free_u8(struct DivansCompressorState* s, uint8_t* p, size_t) { state_free(s->alloc, p); }
size_t* divans_compressor_malloc_usize(struct DivansCompressorState* s, size_t n) { return (size_t*)state_alloc(s->alloc, n * sizeof(size_t)); }
void divans_compressor_free_usize(struct DivansCompressorState* s, size_t* p, size_t) { state_free(s->alloc, p); }
uint8_t* divans_decompressor_malloc_u8(struct DivansDecompressorState* s, size_t n) { return (uint8_t*)state_alloc(s->alloc, n); }
void divans_decompressor_free_u8(struct DivansDecompressorState* s, uint8_t* p, size_t) { state_free(s->alloc, p); }
size_t* divans_decompressor_malloc_usize(struct DivansDecompressorState* s, size_t n) { return (size_t*)state_alloc(s->alloc, n * sizeof(size_t)); }
void divans_decompressor_free_usize(struct DivansDecompressorState* s, size_t* p, size_t) { state_free(s->alloc, p); }

}  // extern "C"
