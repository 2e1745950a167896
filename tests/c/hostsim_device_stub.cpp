// TEST HARNESS, never shipped and never loaded by the divans_amd package.
//
// The per-stream container logic of the product (divans_amd/csrc/host_stream.cpp + ffi.cpp: ring-buffer command emission, CMD coder,
// Mux, CRC, the call-by-call NEEDS_MORE_INPUT / NEEDS_MORE_OUTPUT semantics, the parser of untrusted containers) reaches the GPU through
// nine entry points of include/divans_gpu.h (and divans_amd/csrc/batch.cpp through six more, at the end of this file).  This file stands
// in for them with the CPU oracle (oracle/*.c), so that the host logic
// can be run where there is no GPU -- the "not gpu" test tier -- and under AddressSanitizer / UBSan with mutated input.  What it proves is
// about the HOST code only; the kernels are compared with the oracle by the "-m gpu" tests through the real library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/divans_gpu.h"
extern "C" {
#include "../../oracle/divans_oracle.h"
}

static_assert(sizeof(divans_lit_config) == sizeof(orc_lit_config), "the product's and the oracle's literal configuration are one layout");

static thread_local std::string g_err;
namespace divans_host { int set_last_error(int code, const std::string& msg) { g_err = msg; return code; } }
extern "C" const char* divans_gpu_last_error(void) { return g_err.c_str(); }

struct divans_gpu_codec {
    orc_lit_config cfg;
    uint32_t max_len;
    orc_lit_state* st = nullptr;
    orc_ans_encoder enc; bool enc_live = false;
    bool begun = false;
    uint8_t* flags = nullptr; uint32_t status = 0, blocks = 1024;      // what the batch entry points at the end of the file keep
};

static int fail(int code, const char* msg) { g_err = msg; return code; }

extern "C" int divans_gpu_codec_create(divans_gpu_codec** out, const divans_lit_config* cfg, int device, void* hip_stream, uint32_t max_stream_len) {
    (void)hip_stream;
    if (!out || !cfg || device < 0 || device > 7 || max_stream_len == 0) return fail(DIVANS_GPU_EINVAL, "bad argument");     // up to eight stand-in devices (fakehip)
    for (const auto& s : cfg->literal_adaptation)
        if (s.inc < 0) return fail(DIVANS_GPU_EINVAL, "negative increment");      // divans_gpu_speed_accepted
    divans_gpu_codec* c = new divans_gpu_codec();
    std::memcpy(&c->cfg, cfg, sizeof(c->cfg));
    c->max_len = max_stream_len;
    *out = c;
    return 0;
}

extern "C" void divans_gpu_codec_destroy(divans_gpu_codec* c) {
    if (!c) return;
    if (c->st) orc_lit_state_free(c->st);
    if (c->enc_live) orc_ans_encoder_free(&c->enc);
    delete c;
}

extern "C" int divans_gpu_codec_set_geometry(divans_gpu_codec* c, uint32_t blocks, uint32_t) { if (!c) return DIVANS_GPU_EINVAL; if (blocks) c->blocks = blocks; return 0; }
extern "C" int divans_gpu_codec_tune_tables(divans_gpu_codec* c, uint32_t) { return c ? 0 : DIVANS_GPU_EINVAL; }     // (batch.cpp switches the placement search of its lanes off)

// the rule of divans_amd/csrc/capi.cpp (the one trajectory of cdf[15] from 64 under FrequentistCDF16::blend), so that the stub refuses
// what the library refuses
extern "C" int divans_gpu_speed_supported(int32_t inc, int32_t lim) {
    if (inc < 0 || inc > 0x7fff || lim < -0x8000 || lim > 0x7fff) return 0;
    std::string seen(32768, 0);
    int32_t v = 64;
    while (!seen[v]) {
        seen[v] = 1;
        int32_t a = v + inc;
        if (a > 0x7fff) return 0;
        if (a >= lim) {
            if (a + 16 > 0x7fff) return 0;
            a += 16; a -= a >> 2;
        }
        v = a;
    }
    return 1;
}

extern "C" size_t divans_gpu_lit_encode_bound(size_t n) {
    const size_t nsym = 2 * n;
    const size_t nchunks = (nsym + 65535) / 65536;
    const size_t words = (15 * nsym + 31) / 32 + 2 * nchunks;
    return (16 * nchunks + 4 * words + 15) & ~(size_t)15;
}

static void fresh_state(divans_gpu_codec* c) {
    if (c->st) orc_lit_state_free(c->st);
    c->st = orc_lit_state_new(&c->cfg);
    if (c->enc_live) orc_ans_encoder_free(&c->enc);
    orc_ans_encoder_init(&c->enc); c->enc_live = true;
    c->begun = true;
}

extern "C" int divans_gpu_lit_stream_begin(divans_gpu_codec* c) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    fresh_state(c);
    return 0;
}

extern "C" int divans_gpu_lit_stream_encode(divans_gpu_codec* c, const uint8_t* in, uint32_t len, uint64_t last8, uint8_t* out, size_t out_cap,
                                            uint32_t* chunk_sizes, uint32_t max_chunks, uint32_t* n_chunks, size_t* out_len) {
    if (!c || !in || !out || !n_chunks || !out_len || !c->begun) return fail(DIVANS_GPU_EINVAL, "bad argument (divans_gpu_lit_stream_begin first)");
    if (len == 0 || len > c->max_len) return fail(DIVANS_GPU_EINVAL, "a piece is 1 .. max_stream_len bytes");
    orc_lit_set_last8(c->st, last8);
    *n_chunks = 0; *out_len = 0;
    uint32_t pos = 0;
    while (pos < len) {
        const uint32_t room = (ORC_ANS_NUM_SYMBOLS_BEFORE_FLUSH - c->enc.n_pending) / 2u;     // bytes until the chunk closes
        const uint32_t take = len - pos < room ? len - pos : room;
        orc_lit_encode_bytes(c->st, &c->enc, in + pos, take);
        pos += take;
        if (c->enc.out.len) {
            if (*n_chunks >= max_chunks || !chunk_sizes) return fail(DIVANS_GPU_ECAP, "chunk_sizes too small");
            if (*out_len + c->enc.out.len > out_cap) return fail(DIVANS_GPU_ECAP, "out_cap too small");
            std::memcpy(out + *out_len, c->enc.out.data, c->enc.out.len);
            chunk_sizes[(*n_chunks)++] = (uint32_t)c->enc.out.len;
            *out_len += c->enc.out.len;
            c->enc.out.len = 0;
        }
    }
    if (c->enc.failed) return fail(DIVANS_GPU_EINVAL, "invalid (start,freq) pair");
    return 0;
}

extern "C" int divans_gpu_lit_stream_finish(divans_gpu_codec* c, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!c || !out || !out_len || !c->begun) return fail(DIVANS_GPU_EINVAL, "null argument");
    *out_len = 0;
    if (c->enc.n_pending == 0) return 0;
    orc_ans_flush_chunk(&c->enc);
    if (c->enc.out.len > out_cap) return fail(DIVANS_GPU_ECAP, "out_cap too small");
    std::memcpy(out, c->enc.out.data, c->enc.out.len);
    *out_len = c->enc.out.len; c->enc.out.len = 0;
    return 0;
}

extern "C" int divans_gpu_lit_stream_decode_begin(divans_gpu_codec* c) { return divans_gpu_lit_stream_begin(c); }

extern "C" int divans_gpu_lit_stream_decode(divans_gpu_codec* c, const uint8_t* coded, size_t coded_bytes, uint32_t out_len, uint64_t last8,
                                            uint8_t* out, size_t* consumed_bytes) {
    if (!c || !coded || !out || !consumed_bytes || !c->begun) return fail(DIVANS_GPU_EINVAL, "bad argument (divans_gpu_lit_stream_decode_begin first)");
    if (out_len == 0 || out_len > c->max_len) return fail(DIVANS_GPU_EINVAL, "a call decodes 1 .. max_stream_len bytes");
    orc_ans_decoder d;
    orc_ans_decoder_init(&d, coded, coded_bytes & ~(size_t)3);
    orc_lit_set_last8(c->st, last8);
    orc_lit_decode_bytes(c->st, &d, out, out_len);
    // the kernels' integrity rule (lit_kernels.hip, lit_decode_kernel): words may not run out, and a stream's last chunk ends in the
    // encoder's start states
    if (d.starved || d.state_a != (1ull << 31) || d.state_b != (1ull << 31)) return fail(DIVANS_GPU_ECORRUPT, "the literal stream fails its integrity check");
    *consumed_bytes = d.in_pos;
    return 0;
}

// ---- the batch entry points divans_amd/csrc/batch.cpp uses (with tests/c/fakehip "device" pointers are host pointers) -----------
extern "C" int divans_gpu_codec_info(divans_gpu_codec* c, divans_gpu_info* info) {
    if (!c || !info) return fail(DIVANS_GPU_EINVAL, "null argument");
    std::memset(info, 0, sizeof(*info));
    info->blocks = c->blocks; info->threads = 256; info->rows_per_stream = 4352; info->resident_groups = 16 * info->blocks;
    return 0;
}
extern "C" int divans_gpu_codec_status(divans_gpu_codec* c, uint32_t* status) {
    if (!c || !status) return fail(DIVANS_GPU_EINVAL, "null argument");
    *status = c->status; c->status = 0;
    return 0;
}
extern "C" int divans_gpu_codec_status_async(divans_gpu_codec* c, uint32_t* h) {
    if (!c || !h) return fail(DIVANS_GPU_EINVAL, "null argument");
    *h = c->status;
    return 0;
}
extern "C" int divans_gpu_codec_clear_status(divans_gpu_codec* c) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    c->status = 0;
    return 0;
}
extern "C" int divans_gpu_codec_set_stream_flags(divans_gpu_codec* c, uint8_t* d_flags) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    c->flags = d_flags;
    return 0;
}

extern "C" int divans_gpu_lit_encode_batch_chunks(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets, const uint32_t* d_in_sizes,
                                                  uint32_t stream_len, uint32_t n_streams, uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets,
                                                  uint32_t* d_out_sizes, uint32_t* d_chunk_bytes, uint32_t max_chunks) {
    if (!c || !d_in || !d_out || !d_out_offsets || !d_out_sizes) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (stream_len > c->max_len || (out_slot & 15u) || out_slot < divans_gpu_lit_encode_bound(stream_len)) return fail(DIVANS_GPU_EINVAL, "bad geometry");
    for (uint32_t i = 0; i < n_streams; ++i) {
        const uint32_t len = d_in_sizes ? d_in_sizes[i] : stream_len;
        const uint8_t* in = d_in + (d_in_offsets ? d_in_offsets[i] : (uint64_t)i * stream_len);
        if (len > stream_len) return fail(DIVANS_GPU_EINVAL, "stream longer than stream_len");
        orc_lit_state* st = orc_lit_state_new(&c->cfg);
        orc_ans_encoder enc; orc_ans_encoder_init(&enc);
        uint32_t k = 0;
        for (uint32_t pos = 0; pos < len; ) {
            const uint32_t take = len - pos < 32768u ? len - pos : 32768u;
            const size_t before = enc.out.len;
            orc_lit_encode_bytes(st, &enc, in + pos, take);
            pos += take;
            if (enc.out.len != before && d_chunk_bytes && k < max_chunks) d_chunk_bytes[(size_t)i * max_chunks + k++] = (uint32_t)(enc.out.len - before);
        }
        if (enc.n_pending) { const size_t before = enc.out.len; orc_ans_flush_chunk(&enc); if (d_chunk_bytes && k < max_chunks) d_chunk_bytes[(size_t)i * max_chunks + k++] = (uint32_t)(enc.out.len - before); }
        if (enc.failed) c->status |= DIVANS_GPU_STATUS_BAD_MODEL;
        if (enc.out.len > out_slot) { orc_ans_encoder_free(&enc); orc_lit_state_free(st); return fail(DIVANS_GPU_ECAP, "slot too small"); }
        uint8_t* slot_end = d_out + (uint64_t)(i + 1) * out_slot;                  // right-aligned, as the rANS pass leaves it
        if (enc.out.len) std::memcpy(slot_end - enc.out.len, enc.out.data, enc.out.len);
        d_out_offsets[i] = (uint64_t)(i + 1) * out_slot - enc.out.len;
        d_out_sizes[i] = (uint32_t)enc.out.len;
        orc_ans_encoder_free(&enc); orc_lit_state_free(st);
    }
    return 0;
}

extern "C" int divans_gpu_pack_streams(divans_gpu_codec* c, const uint8_t* d_slots, const uint64_t* d_offsets, const uint32_t* d_sizes, uint32_t n_streams,
                                       uint8_t* d_packed, uint64_t* d_packed_offsets, uint64_t* d_total) {
    if (!c || !d_slots || !d_offsets || !d_sizes || !d_packed || !d_packed_offsets || !d_total) return fail(DIVANS_GPU_EINVAL, "null argument");
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        d_packed_offsets[i] = pos;
        if (d_sizes[i]) std::memcpy(d_packed + pos, d_slots + d_offsets[i], d_sizes[i]);
        pos += ((uint64_t)d_sizes[i] + 3u) & ~(uint64_t)3;
    }
    *d_total = pos;
    return 0;
}

extern "C" int divans_gpu_lit_decode_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets, const uint32_t* d_in_sizes, uint32_t n_streams,
                                           uint8_t* d_out, const uint64_t* d_out_offsets, const uint32_t* d_out_sizes, uint32_t stream_len) {
    if (!c || !d_in || !d_in_offsets || !d_in_sizes || !d_out) return fail(DIVANS_GPU_EINVAL, "null argument");
    for (uint32_t i = 0; i < n_streams; ++i) {
        const uint32_t len = d_out_sizes ? d_out_sizes[i] : stream_len;
        uint8_t* out = d_out + (d_out_offsets ? d_out_offsets[i] : (uint64_t)i * stream_len);
        if (len > c->max_len) return fail(DIVANS_GPU_EINVAL, "stream longer than the codec's bound");
        orc_lit_state* st = orc_lit_state_new(&c->cfg);
        orc_ans_decoder d;
        orc_ans_decoder_init(&d, d_in + d_in_offsets[i], d_in_sizes[i] & ~3u);
        orc_lit_decode_bytes(st, &d, out, len);
        const bool bad = d.starved || (len && (d.state_a != (1ull << 31) || d.state_b != (1ull << 31))) || d.in_pos != (d_in_sizes[i] & ~3u);
        if (bad) { c->status |= DIVANS_GPU_STATUS_BAD_STREAM; if (c->flags) c->flags[i] = 1; }
        orc_lit_state_free(st);
    }
    return 0;
}
