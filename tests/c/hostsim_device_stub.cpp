// TEST HARNESS, never shipped and never loaded by the divans_amd package.
//
// The per-stream container logic of the product (divans_amd/csrc/host_stream.cpp + ffi.cpp: ring-buffer command emission, CMD coder,
// Mux, CRC, the call-by-call NEEDS_MORE_INPUT / NEEDS_MORE_OUTPUT semantics, the parser of untrusted containers) reaches the GPU through
// nine entry points of include/divans_gpu.h.  This file stands in for those nine with the CPU oracle (oracle/*.c), so that the host logic
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
};

static int fail(int code, const char* msg) { g_err = msg; return code; }

extern "C" int divans_gpu_codec_create(divans_gpu_codec** out, const divans_lit_config* cfg, int device, void* hip_stream, uint32_t max_stream_len) {
    (void)hip_stream;
    if (!out || !cfg || device != 0 || max_stream_len == 0) return fail(DIVANS_GPU_EINVAL, "bad argument");
    for (const auto& s : cfg->literal_adaptation)
        if (!divans_gpu_speed_supported(s.inc, s.lim)) return fail(DIVANS_GPU_EINVAL, "unsupported adaptation speed");
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

extern "C" int divans_gpu_codec_set_geometry(divans_gpu_codec* c, uint32_t, uint32_t) { return c ? 0 : DIVANS_GPU_EINVAL; }

// the rule of divans_amd/csrc/capi.cpp (the one trajectory of cdf[15] from 64 under FrequentistCDF16::blend), so that the stub refuses
// what the library refuses
extern "C" int divans_gpu_speed_supported(int32_t inc, int32_t lim) {
    if (inc < 0 || lim <= 0 || inc > 0x4000 || lim > 0x4000) return 0;
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
