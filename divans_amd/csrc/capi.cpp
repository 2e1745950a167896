// capi.cpp -- host side of the batch C ABI declared in include/divans_gpu.h.
// Owns device memory (CDF tables, start/freq spill, configuration blob), derives the compact table
// geometry from the stream configuration and launches the kernels in lit_kernels.hip.
// There is NO CPU fallback: every entry point fails loudly when HIP is unavailable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/divans_gpu.h"
#include "lit_kernels.h"
#include "lit_device.h"      // BytePerm (host side: the English-text hint of divans_gpu_codec_set_byte_order)

using namespace divans_hip;

static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(DIVANS_GPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

extern "C" const char* divans_gpu_last_error(void) { return g_last_error.c_str(); }
namespace divans_host { int set_last_error(int code, const std::string& msg) { return fail(code, msg); } }

// Measurement switches read from the environment exist only in experiment builds (-DDIVANS_EXPERIMENT_SWITCHES=1, scripts/build_variants.sh):
// a shipped library's behaviour does not depend on variables a caller's process happens to carry.
#ifndef DIVANS_EXPERIMENT_SWITCHES
#define DIVANS_EXPERIMENT_SWITCHES 0
#endif
static const char* exp_env(const char* name) {
#if DIVANS_EXPERIMENT_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}


// ---- configuration helpers -------------------------------------------------------------------
static const divans_speed kSpeedMud = {0x10, 0x2000};  // probability/interface.rs:323 / codec/interface.rs:188-190

extern "C" void divans_lit_config_simple(divans_lit_config* cfg) {
    // TestSimple, bin/benchmark.rs:195-206: context map off => map stays zero, mixing values all 4
    // (codec/context_map.rs:269-276, 386-387); lsb6; default MUD speeds
    std::memset(cfg, 0, sizeof(*cfg));
    std::memset(cfg->mixing_mask, 4, sizeof(cfg->mixing_mask));
    for (auto& s : cfg->literal_adaptation) s = kSpeedMud;
}

extern "C" void divans_lit_config_context_mixing(divans_lit_config* cfg) {
    // TestContextMixing through bench_no_ir, bin/benchmark.rs:156-167,305-343
    std::memset(cfg, 0, sizeof(*cfg));
    for (int i = 0; i < 256; ++i) cfg->literal_context_map[i] = (uint8_t)(i & 63);
    std::memset(cfg->mixing_mask, 4, sizeof(cfg->mixing_mask));
    cfg->prediction_mode = 2;
    cfg->btype = 1;
    cfg->context_mixing = 2;
    for (auto& s : cfg->literal_adaptation) s = kSpeedMud;
}

// RFC 7932 section 7.1 context lookups (what constants.rs tabulates), by structure
static uint8_t utf8_lut0(int b) {
    static const uint8_t punct[32] = {8, 12, 16, 12, 12, 20, 12, 16, 24, 28, 12, 12, 32, 12, 36, 12,
                                      44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 32, 32, 24, 40, 28, 12};
    if (b < 32) return (b == 9 || b == 10 || b == 13) ? 4 : 0;
    if (b < 64) return punct[b - 32];
    if (b < 128) {
        const bool lower = b >= 96;
        const int c = b & 31;
        if (c == 0) return 12;
        if (c <= 26) return (uint8_t)((lower ? 56 : 48) + ((c == 1 || c == 5 || c == 9 || c == 15 || c == 21) ? 0 : 4));
        if (c == 27) return 24;
        if (c == 29) return 28;
        if (c == 31) return lower ? 0 : 12;
        return 12;
    }
    return (uint8_t)((b < 192 ? 0 : 2) + (b & 1));
}
static uint8_t utf8_lut1(int b) {
    if (b <= 32 || b == 127) return 0;
    if (b < 128) {
        if ((b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z')) return 2;
        if (b >= 'a' && b <= 'z') return 3;
        return 1;
    }
    return b >= 224 ? 2 : 0;
}
static uint8_t signed3(int b) {
    return b == 0 ? 0 : b < 16 ? 1 : b < 64 ? 2 : b < 128 ? 3 : b < 192 ? 4 : b < 240 ? 5 : b < 255 ? 6 : 7;
}
// codec/interface.rs:199-238 get_lut0 / get_lut1
static void make_luts(uint8_t mode, uint8_t* lut0, uint8_t* lut1) {
    for (int i = 0; i < 256; ++i) {
        switch (mode) {
        case 3: lut0[i] = (uint8_t)(signed3(i) << 3); lut1[i] = signed3(i); break;
        case 2: lut0[i] = utf8_lut0(i); lut1[i] = utf8_lut1(i); break;
        case 1: lut0[i] = (uint8_t)(i >> 2); lut1[i] = 0; break;
        default: lut0[i] = (uint8_t)(i & 0x3f); lut1[i] = 0; break;
        }
    }
}

// the CDF tables' memory: a hipMalloc block (chunks empty) or an address range with physical chunks mapped into it (table_alloc)
struct TableMem { int16_t* p = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> chunks; size_t chunk_bytes = 0, va_bytes = 0; int device = 0; };

struct divans_gpu_codec {
    int device = 0;
    hipStream_t stream = nullptr;
    divans_lit_config cfg;
    LitGeometry geom;
    bool mix = false;
    uint32_t max_stream_len = 0;
    uint32_t num_cus = 256;
    uint32_t blocks = 0;          // persistent grid of the model/decode kernels
    uint32_t cache_high = 0, cache_low = 0;   // per-stream LDS row caches (rows; 0 = that table is accessed in HBM/L2 directly)
    bool cache_unified = false;
    // decoder generation: 2 = lit_decode2.hip (direct-mapped row caches, LDS word ring), 1 = lit_decode_kernel of lit_kernels.hip
    uint32_t decode_gen = 2;
#ifndef DIVANS_DM_AUTO_DEFAULT    // experiment switch (scripts/build_variants.sh dm_auto): 0 = always the codec's own organisation (2-way)
#define DIVANS_DM_AUTO_DEFAULT 1
#endif
    bool dm_auto = DIVANS_DM_AUTO_DEFAULT != 0;          // nobody chose between direct-mapped and 2-way caches (divans_gpu_codec_set_decoder): pick per batch
    uint32_t dm_log2 = 0, dm_shift = 0;   // LitBatch::dm_log2 / dm_shift
    uint32_t blocks2 = 0;                 // persistent grid of lit_decode2_kernel
    uint32_t blocks_t = 0, t_log2 = 0, t_shift = 0;   // generation 4 (lit_decode_t.hip, one lane per stream): grid of 64-stream workgroups, cache rows / shifts
    bool user_geometry = false;           // set_geometry / set_split_cache / set_decoder were called: set_block_types keeps their choices
    char last_decode_kernel[128] = "";    // divans_gpu_codec_last_decode_kernel
    uint32_t last_decode_grid = 0;
    uint8_t* d_stream_flags = nullptr;    // caller-owned per-stream failure flags of the decode entry points (divans_gpu_codec_set_stream_flags)
    uint8_t* d_blob = nullptr;
    int16_t* d_tables = nullptr;  size_t tables_bytes = 0;
    TableMem tm;                          // what d_tables points into (table_alloc)
    uint32_t byte_order = 0;       // divans_gpu_codec_set_byte_order: 0 learned from the codec's own data, 1 numeric, 2 the English-text hint
    uint8_t* d_rank = nullptr;     // [256] rank of every byte value in that order (device); what LitBatch::byte_rank points at once rank_ready
    bool rank_ready = false;       // d_rank holds a permutation (in stream order: the launch that fills it is enqueued before any that reads it)
    uint32_t rans_split = 0;       // divans_gpu_codec_set_rans_split
    uint32_t table_candidates = 0; bool tables_tuned = false;   // divans_gpu_codec_tune_tables / _search_tables: 0 = the library's policy (table_candidates_of)
    bool eager_tune = false;                                    // divans_gpu_codec_tune_tables(c, k >= 2): every placement on the first qualifying call
    divans_gpu_table_placement placement = {0u, 0u, 0.f, 0.f, 0.f, 0u, 0u};   // what the tuning saw (divans_gpu_codec_table_placement)
    // The library's own placement policy explores ACROSS calls (placement_step): every qualifying decode call runs on one placement -- a new
    // candidate or the best so far -- and the call after it reads that launch's time.  `best_tm` holds the best placement while c->tm is a
    // candidate under test (empty: c->tm is the best); two copies alive at most.
    struct PlacementSearch {
        bool active = false, pending = false;       // pending: the launch of the previous qualifying call is the measurement in flight
        uint32_t sig_streams = 0, sig_len = 0;      // only calls of this shape are compared with each other
        TableMem best_tm;
        float best_ms = 0.f, worst_ms = 0.f;
        hipEvent_t e0 = nullptr, e1 = nullptr;      // around the measured launch (ev[3] / ev[4] are re-recorded by every decode call)
        uint64_t seq_at_measure = 0;                // table_launch_seq right after the measured launch
    } ps;
    uint64_t table_launch_seq = 0;                  // launches that read or write the CDF tables (may a placement be freed without a stream sync?)
    uint32_t* d_sf = nullptr;     size_t sf_bytes = 0;
    uint32_t* d_status = nullptr;
    // bucketed encoder model pass (lit_bucket.hip)
    bool bucket_ok = false;       // the configuration allows it: order-1 rows (see configure_from_geometry), no mixing, streams <= 64 KiB
    bool bucket_mix_ok = false;   // two-model configuration the bucketed pass of lit_bucket_mix.hip covers
    uint32_t bucket_mix_batch = 32768;   // streams per launch sequence of the bucketed passes (work arrays are sized for this many)
    uint8_t* d_slots = nullptr; size_t slots_bytes = 0;      // divans_gpu_lit_encode_packed: right-aligned output slots of ONE sub-batch
    uint64_t* d_slot_off = nullptr; size_t slot_off_cap = 0;
    std::vector<hipEvent_t> ev_span; size_t enc_spans = 0;   // ... and its events: per sub-batch (start, before the pack, end)
    float last_pack_ms = 0;
    uint32_t encode_path = 0;     // 0 automatic (bucketed when bucket_ok), 1 streaming kernels, 2 bucketed
    uint8_t* d_bk = nullptr;      size_t bk_bytes = 0; uint32_t bk_streams = 0;
    uint8_t* d_rs = nullptr;      size_t rs_bytes = 0;
    void* host_scratch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // device buffers of the host-buffer entry points, grow-only
    size_t host_scratch_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // chunk-parallel rANS scratch when the bucket arrays are not there to reuse
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // pipelined host-buffer entry points: copy-in / copy-out streams, per-slice events, pinned per-slice totals
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev_in, ev_done;
    uint64_t* h_totals = nullptr; size_t h_totals_cap = 0;
    float last_model_ms = 0, last_rans_ms = 0, last_decode_ms = 0;
    std::vector<hipEvent_t> ev_rans; size_t rans_pairs = 0;   // around the rANS launches of the last encode call
    // one stream coded piece by piece (divans_gpu_lit_stream_*): (start | freq << 16) pairs not yet in a complete chunk, the Weights
    uint32_t* d_sp = nullptr; size_t sp_cap = 0; uint32_t sp_pending = 0; int32_t* d_wstate = nullptr; bool sp_started = false, sd_started = false;
    bool timing_pending_enc = false, timing_pending_dec = false;
};

// A speed whose every row count stays inside i16 for ever.  blend (probability/frequentist_cdf.rs:74-85) adds `inc` to cdf[15] on
// every update whatever the symbol and then takes a quarter off once if the total reached `lim`, so the total follows ONE trajectory
// from the default row's 64 -- with a large `inc` it settles near 4 * inc, far above `lim` -- and the other entries stay below it.
// Such speeds run on every kernel (the reference's debug_asserts `inc, lim <= 0x4000`, probability/interface.rs:341-365, are not
// part of a release build and no arithmetic needs them: only the trajectory counts).
extern "C" int divans_gpu_speed_supported(int32_t inc, int32_t lim) {
    if (inc < 0 || inc > 0x7fff || lim < -0x8000 || lim > 0x7fff) return 0;
    std::vector<uint8_t> seen(32768, 0);
    int32_t v = 64;
    while (!seen[v]) {
        seen[v] = 1;
        int32_t a = v + inc;
        if (a > 0x7fff) return 0;
        if (a >= lim) {
            if (a + 16 > 0x7fff) return 0;              // the bias of entry 15 (CDF_BIAS)
            a += 16; a -= a >> 2;
        }
        v = a;
    }
    return 1;
}
// A speed the coder takes at all: every i16 pair with inc >= 0, i.e. everything the wire format's f8 pairs decode to except a negative
// increment (rows that count DOWN leave the valid range at their first update; nothing writes such a stream).  Where the trajectory
// above leaves i16 the reference's total wraps negative and the NEXT nibble coded with that row is no longer a distribution
// (ans.rs:281-285): a stream in which that happens cannot be read back by the reference itself, one in which it does not -- short
// streams, rows that wrap on their last update -- is coded exactly as the reference codes it.  These speeds run on the streaming
// kernels without row caches (LitGeometry::wrap_check), which report a stream that codes with a wrapped row: status bit BAD_MODEL
// when encoding, BAD_STREAM (+ the per-stream flag) when decoding.
extern "C" int divans_gpu_speed_accepted(int32_t inc, int32_t lim) {
    return inc >= 0 && inc <= 0x7fff && lim >= -0x8000 && lim <= 0x7fff;
}

// Tables are built for the literal block types [bt_first, bt_first + n_btypes): one for a batch of single-segment
// streams (cfg.btype), all of 0..max for streams with BlockSwitchLiteral commands between their segments.
static int derive_geometry(const divans_lit_config& cfg, uint32_t bt_first, uint32_t n_btypes, LitGeometry& g, std::vector<uint8_t>& blob) {
    if (n_btypes == 0 || n_btypes > LIT_MAX_BTYPES || bt_first + n_btypes > 256u)
        return fail(DIVANS_GPU_EINVAL, "a codec keeps context tables for 1..8 consecutive literal block types");
    if (cfg.prediction_mode > 3) return fail(DIVANS_GPU_EINVAL, "prediction_mode must be 0..3 (codec/interface.rs:252-256)");
    if (cfg.context_mixing >= 15) return fail(DIVANS_GPU_EINVAL, "context_mixing must be < 15 (codec/interface.rs:359)");
    for (int i = 0; i < 4; ++i) {
        const divans_speed s = cfg.literal_adaptation[i];
        if (!divans_gpu_speed_accepted(s.inc, s.lim))
            return fail(DIVANS_GPU_EINVAL, "literal_adaptation speed with a negative increment (divans_gpu_speed_accepted)");
    }
    const uint32_t mix_off = LIT_BLOB_CTXF + LIT_CTXF_BYTES * n_btypes;
    blob.assign(mix_off + DIVANS_GPU_NUM_MIXING_VALUES, 0);
    uint8_t lut0[256], lut1[256];
    make_luts(cfg.prediction_mode, lut0, lut1);
    std::memcpy(&blob[mix_off], cfg.mixing_mask, DIVANS_GPU_NUM_MIXING_VALUES);
    // lut1 takes at most 8 distinct values (SIGN: 0..7, UTF8: 0..3, MSB6/LSB6: 0): class them so the kernel
    // can fold lut0 | lut1 | context map into one LDS read keyed by (prev, class(prev_prev))
    int class_of_value[256]; std::memset(class_of_value, -1, sizeof(class_of_value));
    uint8_t class_value[8] = {0}; int nclass = 0;
    for (int b = 0; b < 256; ++b) {
        if (class_of_value[lut1[b]] < 0) {
            if (nclass == 8) return fail(DIVANS_GPU_EINVAL, "literal_lut1 has more than 8 distinct values");
            class_value[nclass] = lut1[b]; class_of_value[lut1[b]] = nclass++;
        }
        blob[LIT_BLOB_LUT1CLASS + b] = (uint8_t)class_of_value[lut1[b]];
    }
    bool ctx_seen[256] = {false};
    uint32_t maxctx = 0; int first = -1; bool constant = true;
    for (uint32_t bt = 0; bt < n_btypes; ++bt) {
        const uint8_t* cmap64 = cfg.literal_context_map + 64u * (bt_first + bt);  // literal.rs:114 cmap_index = sel + (btype << 6)
        for (int prev = 0; prev < 256; ++prev) {
            for (int k = 0; k < 8; ++k) {
                const uint8_t sel = (uint8_t)((lut0[prev] | class_value[k < nclass ? k : 0]) & 63);
                const uint8_t c = cmap64[sel];
                blob[LIT_BLOB_CTXF + LIT_CTXF_BYTES * bt + prev * 8 + k] = c;
                if (k >= nclass) continue;
                ctx_seen[c] = true;
                maxctx = std::max<uint32_t>(maxctx, c);
                if (first < 0) first = c; else if (c != first) constant = false;
            }
        }
    }
    std::memset(&g, 0, sizeof(g));
    g.bt_first = bt_first; g.n_btypes = n_btypes; g.mix_off = mix_off; g.lut1_classes = (uint32_t)nclass;
    g.nctx = maxctx + 1;
    g.ctx_const = constant ? first : -1;
    // reachable mixing values: index = ctx | nibble << 8 | (low ? 4096 : 0)  (literal.rs:176-183)
    bool t_used[3] = {false, false, false};
    bool any1 = false; int mmfirst = -1; bool mmconst = true;
    for (int idx = 0; idx < DIVANS_GPU_NUM_MIXING_VALUES; ++idx) {
        if (!ctx_seen[idx & 0xff]) continue;
        const uint8_t m = cfg.mixing_mask[idx];
        if (mmfirst < 0) mmfirst = m; else if (m != mmfirst) mmconst = false;
        const int t = (m == 0 || m == 3) ? 0 : (m == 1 ? 2 : 1);
        t_used[t] = true;
        any1 |= (m == 1);
    }
    g.mm_uniform = mmconst ? mmfirst : -1;
    uint32_t nplanes = 0;
    uint32_t plane[3] = {0, 0, 0};
    for (int t = 0; t < 3; ++t) if (t_used[t]) plane[t] = nplanes++;
    g.plane0 = plane[0]; g.plane1 = plane[1]; g.plane2 = plane[2];
    g.low_width = any1 ? 256u : 16u;
    // The high-nibble stride row of code_nibble is [ctx][prev] (literal.rs:205-208 with every mixing value 4), and ctx itself is a
    // function of (prev, lut1 class of prev_prev): of the nctx x 256 rows only (classes) x 256 can ever be touched.  Where that function
    // table is laid out [row slot of the class][prev] instead: config 3 (cm[i] = i & 63, UTF8) 16 384 -> 1024 rows, 690 KB -> 199 KB per
    // resident stream, a third of the table fill per stream.
#ifndef DIVANS_HS_BY_CLASS
#define DIVANS_HS_BY_CLASS 1
#endif
    g.hs_classes = 0;
    if (DIVANS_HS_BY_CLASS && g.mm_uniform == 4 && !constant && n_btypes == 1u && nclass <= 4 && (uint32_t)nclass < g.nctx) {
        // two classes of one prev may select the same context (UTF8: lut0 | lut1 collides for the non-ASCII bytes) and then share a
        // row: the free half of the fused table ([prev][4 + class], the prediction modes with at most 4 classes) names the row slot
        // of every class -- the first class with that context
        uint32_t most = 0;
        for (int prev = 0; prev < 256; ++prev) {
            uint32_t used = 0;
            for (int k = 0; k < nclass; ++k) {
                uint32_t slot = used;
                for (int j = 0; j < k; ++j)
                    if (blob[LIT_BLOB_CTXF + prev * 8 + j] == blob[LIT_BLOB_CTXF + prev * 8 + k]) { slot = blob[LIT_BLOB_CTXF + prev * 8 + 4 + j]; break; }
                if (slot == used) ++used;
                blob[LIT_BLOB_CTXF + prev * 8 + 4 + k] = (uint8_t)slot;
            }
            most = std::max(most, used);
        }
        if (most < g.nctx) g.hs_classes = most;
    }
    const uint32_t high_rows = g.hs_classes ? 256u * g.hs_classes : nplanes * 256u * g.nctx;
    const uint32_t low_rows = nplanes * 256u * g.low_width;
    g.low_base = high_rows;
    g.cm_base = high_rows + low_rows;
    const bool mix = cfg.context_mixing > 1;  // Weights::should_mix, weights.rs:44-46
    g.total_rows = g.cm_base + (mix ? 17u * g.nctx : 0u);
    if (const char* e = exp_env("DIVANS_SLAB_ROWS_MOD")) {      // experiment builds only: pad a stream's slab to r rows mod m ("m:r")
        unsigned m = 0, r = 0;
        if (sscanf(e, "%u:%u", &m, &r) == 2 && m > 1 && r < m) while (g.total_rows % m != r) ++g.total_rows;
    }
    g.inc0 = cfg.literal_adaptation[0].inc; g.lim0 = cfg.literal_adaptation[0].lim;
    g.inc1 = cfg.literal_adaptation[1].inc; g.lim1 = cfg.literal_adaptation[1].lim;
    g.inc2 = cfg.literal_adaptation[2].inc; g.lim2 = cfg.literal_adaptation[2].lim;
    g.inc3 = cfg.literal_adaptation[3].inc; g.lim3 = cfg.literal_adaptation[3].lim;
    // speeds [1] is never read by the literal coder (literal.rs:320,354 use [0] for both nibbles); [2] / [3] only with mixing
    g.wrap_check = 0;
    const bool mixing = cfg.context_mixing > 1;
    for (int i = 0; i < 4; ++i) {
        if (i == 1 || (!mixing && i >= 2)) continue;
        if (!divans_gpu_speed_supported(cfg.literal_adaptation[i].inc, cfg.literal_adaptation[i].lim)) g.wrap_check = 1;
    }
    return 0;
}

static void set_cache_fields(const divans_gpu_codec* c, LitBatch& b) {
    // 0 none, 1 unified, 2 high-nibble rows only, 3 separate high / low caches; wrap_check: the post-stream scan reads the table in HBM
    if (c->cache_high == 0 || c->geom.wrap_check) { b.cache_mode = 0; b.cache_rows_high = b.cache_rows_low = 0; }   // (a low-only cache is not offered)
    else if (c->cache_unified) { b.cache_mode = 1; b.cache_rows_high = c->cache_high; b.cache_rows_low = 0; }
    else if (c->cache_low == 0) { b.cache_mode = 2; b.cache_rows_high = c->cache_high; b.cache_rows_low = 0; }
    else { b.cache_mode = 3; b.cache_rows_high = c->cache_high; b.cache_rows_low = c->cache_low; }
    b.cache_bytes_per_wg = (LIT_THREADS / 16) * (b.cache_rows_high + b.cache_rows_low) * 34u;
}

static uint32_t groups_per_block(const divans_gpu_codec*) { return LIT_THREADS / 16; }
static bool use_decode2(const divans_gpu_codec* c) { return c->decode_gen == 2u && c->blocks2 != 0u && !c->geom.wrap_check; }
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
static bool use_decode_t(const divans_gpu_codec* c) { return c->decode_gen == 4u && c->blocks_t != 0u && !c->geom.wrap_check; }
#else
static bool use_decode_t(const divans_gpu_codec*) { return false; }
#endif
extern "C" int divans_gpu_experimental_decoders(void) { return DIVANS_WITH_EXPERIMENTAL_DECODERS; }
// streams that own a table slab at once
static uint32_t resident_groups(const divans_gpu_codec* c) {
    return std::max(std::max(c->blocks, use_decode2(c) ? c->blocks2 : 0u) * groups_per_block(c), use_decode_t(c) ? c->blocks_t * 64u : 0u);
}

// The CDF tables' memory.  How the driver places and maps it changes the decode time by 10-20 % (profiles/r04e_table_placement.txt): one
// physically contiguous block (hipDeviceMallocContiguous) is always the slowest (the L2 loses two thirds of its hits), a hipMalloc block lands
// anywhere in a 10 % band from one allocation -- and one box -- to the next (on some boxes always at the slow end), physical chunks created one
// by one and mapped side by side into a reserved address range (hipMemCreate / hipMemMap) have the better worst case.  Big tables use the
// latter (table_alloc), and divans_gpu_codec_tune_tables measures several placements of both kinds and keeps the fastest.
// One rule comes with the mapping calls: an address range is NEVER handed back (hipMemAddressFree).  On ROCm 7.2 a range that is unmapped,
// freed, reserved again and mapped to new chunks reads and writes through stale translations -- scripts/probes/vmm_remap_probe.hip shows it
// with nothing but the runtime API, and the decode kernels returned wrong bytes that way.  So a range a codec is done with stays
// mapped and waits in a small per-process pool for the next codec that fits; a range the pool gives up is unmapped and its chunks
// released, but its addresses stay reserved for the life of the process (address space, not memory).
// DIVANS_TABLES_ALLOC = "hipmalloc" / "contiguous" / "scattered:<chunk MiB>[:noshuffle]" are measurement switches of experiment builds (exp_env).
//
// Address space is accounted for (g_va_reserved): past kTableVaCap of reserved-and-never-returned ranges new tables are plain hipMalloc blocks,
// which hipFree does give back, so a process that cycles big codecs for days ends on the allocator's ordinary behaviour instead of running the
// device's address space down (divans_gpu_table_memory reports both numbers; the defect's description for the vendor: scripts/probes/README.md).
static std::mutex g_table_pool_mu;
static std::vector<TableMem> g_table_pool;            // mapped ranges no codec uses, oldest first
constexpr size_t kTablePoolRanges = 2;
static std::atomic<uint64_t> g_va_reserved{0};        // bytes of address ranges reserved for tables and never handed back (the rule above)
static std::atomic<uint64_t> g_va_cap{(uint64_t)256 << 30};   // kTableVaCap: 256 GiB = 64 four-GiB tables' worth; divans_gpu_set_table_va_cap

static void table_release_chunks(TableMem& t) {        // unmap and give the memory back; the address range stays reserved (see above)
    for (size_t i = 0; i < t.chunks.size(); ++i) (void)hipMemUnmap((uint8_t*)t.p + i * t.chunk_bytes, t.chunk_bytes);
    for (auto h : t.chunks) (void)hipMemRelease(h);
    t.chunks.clear();
    t.p = nullptr; t.bytes = 0;
}

static void table_free(TableMem& t, bool keep_mapped = true) {
    if (!t.p) return;
    if (t.chunks.empty()) { (void)hipFree(t.p); t.p = nullptr; t.bytes = 0; return; }
    if (!keep_mapped) { table_release_chunks(t); t = TableMem(); return; }      // (the tuning loop: memory back now, nothing kept for a later codec)
    std::lock_guard<std::mutex> lock(g_table_pool_mu);
    g_table_pool.push_back(std::move(t));
    t = TableMem();
    while (g_table_pool.size() > kTablePoolRanges) { table_release_chunks(g_table_pool.front()); g_table_pool.erase(g_table_pool.begin()); }
}

static void table_pool_drop(int device) {      // device < 0: every device
    std::lock_guard<std::mutex> lock(g_table_pool_mu);
    for (size_t i = 0; i < g_table_pool.size();) {
        if (device < 0 || g_table_pool[i].device == device) { table_release_chunks(g_table_pool[i]); g_table_pool.erase(g_table_pool.begin() + i); } else ++i;
    }
}

// hipMalloc for the codec's other big arrays: before it fails for lack of memory, the idle table ranges of the pool are given back
static hipError_t device_alloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess) table_pool_drop(dev);
        e = hipMalloc(p, bytes);
    }
    return e;
}

// *over_cap: nothing was tried because the address-space budget (g_va_cap) is spent -- not an out-of-memory condition: the caller goes
// straight to one hipMalloc block and leaves the pool's idle ranges alone
static hipError_t table_alloc_chunks(int device, size_t need, size_t chunk_mib, bool shuffle, TableMem& t, bool* over_cap) {
    *over_cap = false;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorUnknown;
    size_t chunk = std::max<size_t>(gran, chunk_mib << 20);
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (need + chunk - 1) / chunk;
    // the budget is taken before the reservation (callers on several devices run concurrently) and handed back if there is none to keep
    const uint64_t want_va = (uint64_t)n * chunk;
    if (g_va_reserved.fetch_add(want_va) + want_va > g_va_cap.load()) { g_va_reserved.fetch_sub(want_va); *over_cap = true; return hipErrorOutOfMemory; }
    void* va = nullptr;
    e = hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0);
    if (e != hipSuccess) { g_va_reserved.fetch_sub(want_va); return e; }
    // from here on the range is never returned, whether the mapping below succeeds or not
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e == hipSuccess) hs.push_back(h);
    }
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    if (shuffle) { uint64_t x = 0x9e3779b97f4a7c15ull; for (size_t i = n; i > 1; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[i - 1], order[x % i]); } }
    size_t mapped = 0;
    for (; mapped < n && e == hipSuccess; ++mapped) {
        e = hipMemMap((uint8_t*)va + mapped * chunk, chunk, 0, hs[order[mapped]], 0);
        if (e != hipSuccess) break;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc desc;
        std::memset(&desc, 0, sizeof(desc));
        desc.location.type = hipMemLocationTypeDevice; desc.location.id = device; desc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, n * chunk, &desc, 1);
    }
    if (e != hipSuccess) {       // (the range stays reserved here too)
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((uint8_t*)va + i * chunk, chunk);
        for (auto h : hs) (void)hipMemRelease(h);
        (void)hipGetLastError();
        return e;
    }
    t.chunks.resize(n);
    for (size_t i = 0; i < n; ++i) t.chunks[i] = hs[order[i]];      // in mapping order: table_release_chunks walks them
    t.chunk_bytes = chunk; t.va_bytes = n * chunk;
    t.p = (int16_t*)va; t.bytes = need; t.device = device;
    return hipSuccess;
}

// fresh = never from the pool (divans_gpu_codec_tune_tables compares placements: it must not be handed the range it just put aside)
// plain = one hipMalloc block whatever the mode (the tuning tries both kinds: which is faster differs from box to box)
static hipError_t table_alloc(int device, size_t need, TableMem& t, bool fresh = false, bool plain = false) {
    const char* mode = exp_env("DIVANS_TABLES_ALLOC");      // experiment builds only (see exp_env)
    hipError_t e = hipErrorUnknown;
    // Every mapped chunk is a buffer object of its own, and a process that holds thousands of them pays for it in every other runtime call:
    // with 2 MiB chunks under the tables of the eight lanes of divans_batch_* the many-containers ABI fell from 8.0 / 8.0 to 6.7 / 4.9 GB/s
    // (profiles/r04f_batch_container_rate_ab.txt).  So: 32 MiB chunks (as fast as 2 MiB ones, profiles/r04e_table_placement.txt), and only for
    // tables of 2 GiB and more -- the persistent grids of whole-GPU batches, where the placement is worth 10-20 %; smaller tables are one block.
    if (plain || (!mode && need < ((size_t)2 << 30))) {
    } else if (!mode || mode[0] == 's') {
        unsigned mib = 32; char tail[32] = "";
        if (mode) (void)sscanf(mode, "scattered:%u:%31s", &mib, tail);
        if (!fresh) {
            std::lock_guard<std::mutex> lock(g_table_pool_mu);
            size_t pick = g_table_pool.size();
            for (size_t i = 0; i < g_table_pool.size(); ++i) {
                const TableMem& m = g_table_pool[i];
                if (m.device == device && m.va_bytes >= need && m.va_bytes <= need + need / 2 + (64u << 20) && m.chunk_bytes == ((size_t)mib << 20) &&
                    (pick == g_table_pool.size() || m.va_bytes < g_table_pool[pick].va_bytes)) pick = i;
            }
            if (pick < g_table_pool.size()) {
                t = std::move(g_table_pool[pick]);
                g_table_pool.erase(g_table_pool.begin() + pick);
                t.bytes = need;
                return hipSuccess;
            }
        }
        bool over_cap = false;
        e = table_alloc_chunks(device, need, mib ? mib : 32, std::strcmp(tail, "noshuffle") != 0, t, &over_cap);
        if (e != hipSuccess && !over_cap) { table_pool_drop(device); e = table_alloc_chunks(device, need, mib ? mib : 32, std::strcmp(tail, "noshuffle") != 0, t, &over_cap); }   // out of memory: the idle ranges first
    } else if (mode[0] == 'c') {
        e = hipExtMallocWithFlags((void**)&t.p, need, hipDeviceMallocContiguous);
        if (e != hipSuccess) { (void)hipGetLastError(); t.p = nullptr; }
    }
    if (e != hipSuccess) { e = device_alloc((void**)&t.p, need); if (e != hipSuccess) t.p = nullptr; else { t.chunks.clear(); t.device = device; } }
    if (e == hipSuccess) t.bytes = need;
    return e;
}

static void free_tables(divans_gpu_codec* c) {     // (callers have synchronised the stream when kernels may still use the tables)
    table_free(c->tm); c->d_tables = nullptr; c->tables_bytes = 0; c->tables_tuned = false;
    table_free(c->ps.best_tm, false);
    c->ps.active = c->ps.pending = false;
}

static int ensure_tables(divans_gpu_codec* c) {
    // big geometries (many context columns / planes) shrink the persistent grid instead of asking for hundreds of GB:
    // at most a quarter of the device memory that is free right now, and never less than one workgroup
    size_t free_b = 0, total_b = 0;
    const size_t per_block = (size_t)groups_per_block(c) * c->geom.total_rows * 32u;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && c->tables_bytes == 0) {
        const size_t budget = std::max<size_t>(free_b / 4, per_block);
        if ((size_t)c->blocks * per_block > budget) c->blocks = (uint32_t)std::max<size_t>(1, budget / per_block);
        if ((size_t)c->blocks2 * per_block > budget) c->blocks2 = (uint32_t)std::max<size_t>(1, budget / per_block);
        if ((size_t)c->blocks_t * 4u * per_block > budget) c->blocks_t = (uint32_t)std::max<size_t>(1, budget / (4u * per_block));
    }
    const size_t need = (size_t)resident_groups(c) * c->geom.total_rows * 32u;
    if (need <= c->tables_bytes) return 0;
    if (c->d_tables) { HIP_TRY(hipStreamSynchronize(c->stream)); free_tables(c); }
    if (table_alloc(c->device, need, c->tm) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(CDF tables) failed");
    c->d_tables = c->tm.p;
    if (exp_env("DIVANS_DEBUG_ALLOC")) fprintf(stderr, "[divans] tables %p + %zu in %zu chunks\n", (void*)c->d_tables, need, c->tm.chunks.size());
    c->tables_bytes = need;
    return 0;
}

static int ensure_sf(divans_gpu_codec* c, uint32_t n_streams) {
    const size_t need = (size_t)n_streams * 2u * c->max_stream_len * sizeof(uint32_t);
    if (need <= c->sf_bytes) return 0;
    if (c->d_sf) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_sf)); c->d_sf = nullptr; c->sf_bytes = 0; }
    if (device_alloc((void**)&c->d_sf, need) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(start/freq spill) failed");
    c->sf_bytes = need;
    return 0;
}

// launch geometry, LDS cache organisation and encoder path that follow from the table geometry
static void configure_from_geometry(divans_gpu_codec* c) {
    c->blocks = c->num_cus * 4u;  // without a row cache (more than 32766 rows per stream): 16 waves = 64 streams per CU
    c->cache_high = c->cache_low = 0u; c->cache_unified = false;
    // row cache: high-nibble rows only -- few and hot (32 rows take ~90 % of their accesses)
    if (c->geom.total_rows < 0x7fffu) {
        // the kernels gain more from a seventh wave per SIMD than from the second half of the row cache (DESIGN.md
        // section 7; the non-mixing decode kernel needs 62 VGPRs): 7 workgroups per CU with 32-row caches, fewer when
        // the context tables of a generic configuration take more of the CU's 160 KB of LDS (the kernels stage the
        // mixing mask unless the mixing value is one they are specialised for: 0 or 4, see effective_mm)
        c->cache_high = 32u;
        const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
        const uint32_t lds_per_wg = (LIT_THREADS / 16) * c->cache_high * 34u +
                                    (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
        const uint32_t fit = (160u * 1024u) / lds_per_wg;
        c->blocks = c->num_cus * std::max(1u, std::min(7u, fit));
    }
    // lit_decode2_kernel: 2-way caches, 15-bit row ids in the tags.  Defaults (profiles/r03_*): the high stride rows get 32
    // rows -- sets indexed by the previous byte itself when the context is constant, by row ^ (row >> 5) otherwise -- and with prior
    // mixing the FirstNibble context-map rows 16 of their own; the low-nibble rows are many and stay in HBM / L2.
    c->blocks2 = 0; c->dm_log2 = 0; c->dm_shift = 0;
    if (c->geom.total_rows < 0x7fffu) {
        if (c->mix) { c->dm_log2 = 5u | (5u << 8); c->dm_shift = 5u | (5u << 8); }
        else { c->dm_log2 = 6u; c->dm_shift = c->geom.ctx_const >= 0 ? 31u : 5u; }
    }
    // organisation: 2-way sets (generation 3) without mixing -- fewer misses at the same time per byte --, direct mapped (generation 2) with:
    // since the high stride rows are laid out [class slot][prev] (hs_classes) the direct-mapped lookup is 5-8 % ahead there on tables
    // placed alike (profiles/r04e_table_placement.txt: 433-442 vs 467-480 ms for 65 536 streams)
    if (!c->mix) c->dm_shift |= 0x80000000u;
    {
        const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
        const uint32_t lds_per_wg = (LIT_THREADS / 16) * lit_decode2_stream_lds(c->dm_log2) + 256u /* byte ranks */ +
                                    (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
        const uint32_t fit = (160u * 1024u) / lds_per_wg;
        c->blocks2 = c->num_cus * std::max(1u, std::min(7u, fit));
    }
    // order-1 rows: high row [ctx][prev], low row [prev][hi] with the context constant, or a function of prev alone (one lut1
    // class and one block type: LSB6 / MSB6 -- what the reference's literal-only compressor emits, raw_to_cmd/mod.rs:115-140)
    const bool ctx_from_prev = c->geom.ctx_const >= 0 || (c->geom.lut1_classes == 1u && c->geom.n_btypes == 1u);
    c->bucket_ok = !c->mix && c->geom.mm_uniform == 4 && ctx_from_prev && c->max_stream_len <= 65536u;
    // both models' rows depend on (prev, ctx, high nibble) only when every mixing value is 4 (stride 1, literal.rs:184-192)
    c->bucket_mix_ok = c->mix && c->geom.mm_uniform == 4 && c->geom.n_btypes == 1u && c->max_stream_len <= 65536u;
}

extern "C" void divans_gpu_codec_destroy(divans_gpu_codec* c);

extern "C" int divans_gpu_codec_create(divans_gpu_codec** out, const divans_lit_config* cfg, int device, void* hip_stream,
                                       uint32_t max_stream_len) {
    if (!out || !cfg || max_stream_len == 0) return fail(DIVANS_GPU_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DIVANS_GPU_EHIP, "no HIP device: the divans literal coder has no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    divans_gpu_codec* c = new divans_gpu_codec();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->cfg = *cfg;
    c->max_stream_len = (max_stream_len + 1u) & ~1u;   // even: keeps every stream's start/freq spill 16-byte aligned
    std::vector<uint8_t> blob;
    int rc = derive_geometry(*cfg, cfg->btype, 1, c->geom, blob);
    if (rc) { divans_gpu_codec_destroy(c); return rc; }
    c->mix = cfg->context_mixing > 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = (uint32_t)prop.multiProcessorCount;
    configure_from_geometry(c);
    hipError_t he = hipMalloc(&c->d_blob, LIT_BLOB_MAX_BYTES);
    if (he == hipSuccess) he = hipMalloc(&c->d_status, 64);
    if (he == hipSuccess) he = hipMalloc(&c->d_rank, 256);
    if (he == hipSuccess) he = hipEventCreate(&c->ps.e0);
    if (he == hipSuccess) he = hipEventCreate(&c->ps.e1);
    if (he != hipSuccess) { divans_gpu_codec_destroy(c); return fail(DIVANS_GPU_ENOMEM, "hipMalloc(config) failed"); }
    if (he == hipSuccess) he = hipMemcpy(c->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset(c->d_status, 0, 64);
    for (auto& e : c->ev) if (he == hipSuccess) he = hipEventCreate(&e);
    if (he != hipSuccess) { divans_gpu_codec_destroy(c); return fail(DIVANS_GPU_EHIP, std::string("codec setup: ") + hipGetErrorString(he)); }
    *out = c;
    return 0;
}

extern "C" void divans_gpu_codec_destroy(divans_gpu_codec* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->d_blob) (void)hipFree(c->d_blob);
    free_tables(c);
    if (c->d_sf) (void)hipFree(c->d_sf);
    if (c->d_bk) (void)hipFree(c->d_bk);
    if (c->d_slots) (void)hipFree(c->d_slots);
    if (c->d_slot_off) (void)hipFree(c->d_slot_off);
    for (hipEvent_t e : c->ev_span) (void)hipEventDestroy(e);
    if (c->d_rs) (void)hipFree(c->d_rs);
    if (c->d_sp) (void)hipFree(c->d_sp);
    if (c->d_wstate) (void)hipFree(c->d_wstate);
    for (void* q : c->host_scratch) if (q) (void)hipFree(q);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_rank) (void)hipFree(c->d_rank);
    if (c->ps.e0) (void)hipEventDestroy(c->ps.e0);
    if (c->ps.e1) (void)hipEventDestroy(c->ps.e1);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_rans) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_in) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    delete c;
}

static uint32_t bucket_pieces(const divans_gpu_codec* c) { return (c->max_stream_len + 8191u) / 8192u; }
// Optional pad (a multiple of 16 elements) between the stream slots of the bucketed passes' work arrays.  Measured: breaking
// the power-of-two stride changes nothing (profiles/r02c), so it stays 0.
constexpr uint32_t BUCKET_SLOT_PAD = 0;
static uint32_t bucket_slot(const divans_gpu_codec* c) { return bucket_pieces(c) * 8192u + BUCKET_SLOT_PAD; }
static bool use_bucket_mix(const divans_gpu_codec* c) { return c->bucket_mix_ok && c->encode_path != 1u && !c->geom.wrap_check; }
static bool use_bucket(const divans_gpu_codec* c) { return c->bucket_ok && c->encode_path != 1u && !c->geom.wrap_check; }   // task ids are stream * 256 + byte in 32 bits: callers keep n_streams < 2^24

// one allocation carved into the five work arrays of BucketBatch
static int ensure_bucket(divans_gpu_codec* c, uint32_t n_streams, BucketBatch& b) {
    const size_t pl = bucket_slot(c);
    const size_t n = n_streams;
    const size_t sz_sfs = n * pl * 8u, sz_desc = n * 256u * 8u * 4u, sz_tasks = n * 256u * 6u * 4u, sz_inv = n * pl * 2u, sz_sorted = n * pl;
    const size_t need = sz_sfs + sz_desc + sz_tasks + sz_inv + sz_sorted + 256u;
    if (need > c->bk_bytes) {
        if (c->d_bk) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_bk)); c->d_bk = nullptr; c->bk_bytes = 0; }
        if (device_alloc((void**)&c->d_bk, need) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(bucketed encoder work arrays) failed");
        c->bk_bytes = need;
    }
    uint8_t* p = c->d_bk;
    b.counters = (uint32_t*)p; p += 256;
    b.sfs = (bk_u32x2*)p; p += sz_sfs;
    b.desc = (uint32_t*)p; p += sz_desc;
    b.tasks = (uint32_t*)p; p += sz_tasks;
    b.inv = (uint16_t*)p; p += sz_inv;
    b.sorted = p;
    return 0;
}

static int ensure_bucket_mix(divans_gpu_codec* c, uint32_t n_streams, MixBucketBatch& b) {
    const size_t pl = bucket_slot(c);
    const size_t n = n_streams;
    const size_t sz_xs = n * pl * 8u, sz_max = n * pl * 4u, sz_desc = n * 256u * 8u * 4u, sz_tasks = n * 256u * 6u * 4u, sz_inv = n * pl * 2u, sz_sorted = n * pl * 2u;
    const size_t need = 256u + 2u * (sz_xs + sz_max) + sz_desc + sz_tasks + sz_inv + sz_sorted;
    if (need > c->bk_bytes) {
        if (c->d_bk) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_bk)); c->d_bk = nullptr; c->bk_bytes = 0; }
        if (device_alloc((void**)&c->d_bk, need) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(bucketed two-model encoder work arrays) failed");
        c->bk_bytes = need;
    }
    uint8_t* p = c->d_bk;
    b.counters = (uint32_t*)p; p += 256;
    for (int i = 0; i < 2; ++i) { b.xs[i] = (bk_u32x2*)p; p += sz_xs; }      // 12 bytes per position and model: the entries, then the row totals
    for (int i = 0; i < 2; ++i) { b.maxes[i] = (uint32_t*)p; p += sz_max; }
    b.desc = (uint32_t*)p; p += sz_desc;
    b.tasks = (uint32_t*)p; p += sz_tasks;
    b.inv = (uint16_t*)p; p += sz_inv;       // inv and sorted stay adjacent: together they are the rANS pass's scratch (SfView::spare)
    b.sorted = (uint16_t*)p;
    return 0;
}

// What a model pass leaves for the rANS pass: the (start | freq << 16) pairs of `count` streams, `stride` u32 apart, and
// work memory of the pass that is dead by then (the bucketed passes' inv + sorted arrays).
struct SfView { uint32_t* sf = nullptr; uint32_t stride = 0; uint8_t* spare = nullptr; size_t spare_bytes = 0; };

// Scratch of the chunk-parallel rANS pass: one chunk bound per stream plus a size word.  After a bucketed model pass it
// lives in that pass's dead work arrays; otherwise in a separate allocation.
static int ensure_rans_scratch(divans_gpu_codec* c, uint32_t n_streams, const SfView& v, RansBatch& r) {
    const uint64_t stride = divans_gpu_lit_encode_bound(32768);
    const size_t need = (size_t)n_streams * stride + (size_t)n_streams * 4u + 64u;
    uint8_t* base = v.spare_bytes >= need ? v.spare : nullptr;
    if (!base) {
        if (need > c->rs_bytes) {
            if (c->d_rs) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_rs)); c->d_rs = nullptr; c->rs_bytes = 0; }
            if (device_alloc((void**)&c->d_rs, need) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(rANS chunk scratch) failed");
            c->rs_bytes = need;
        }
        base = c->d_rs;
    }
    r.chunk0_sizes = (uint32_t*)base;
    r.scratch = base + (((size_t)n_streams * 4u + 63u) & ~(size_t)63u);
    r.scratch_stride = stride;
    return 0;
}

// General streams switch the literal block type between Literal commands (BlockSwitchLiteral): rebuild the context
// tables for block types 0 .. n_btypes-1 (the segment entry points index them by divans_lit_segment::btype).
extern "C" int divans_gpu_codec_set_block_types(divans_gpu_codec* c, uint32_t n_btypes) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    HIP_TRY(hipSetDevice(c->device));
    LitGeometry g; std::vector<uint8_t> blob;
    int rc = derive_geometry(c->cfg, 0, n_btypes, g, blob); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    c->geom = g;
    free_tables(c);
    // the block types change the LDS the context tables take, not what the caller chose before: keep an explicitly set
    // grid / cache organisation / decoder generation, clamped to what still fits the LDS (and to caches the new row count allows)
    const uint32_t blocks = c->blocks, blocks2 = c->blocks2, ch = c->cache_high, cl = c->cache_low, gen = c->decode_gen, lg = c->dm_log2, sh = c->dm_shift;
    const bool uni = c->cache_unified, user = c->user_geometry;
    configure_from_geometry(c);
    if (user) {
        const uint32_t fit_blocks = c->blocks, fit_blocks2 = c->blocks2;
        if (c->geom.total_rows < 0x7fffu) {
            c->cache_high = ch; c->cache_low = cl; c->cache_unified = uni; c->dm_log2 = lg; c->dm_shift = sh;
            const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
            const uint32_t extra = (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
            const uint32_t lds1 = (LIT_THREADS / 16) * (ch + (uni ? 0u : cl)) * 34u + extra, lds2 = (LIT_THREADS / 16) * lit_decode2_stream_lds(lg) + extra;
            if (lds1 > 160u * 1024u || lds2 > 160u * 1024u) return fail(DIVANS_GPU_EINVAL, "the row caches chosen before do not fit the LDS next to this many context tables");
            c->blocks = std::min(blocks, c->num_cus * std::max(1u, std::min(8u, (160u * 1024u) / std::max(lds1, 1u))));
            c->blocks2 = std::min(blocks2, c->num_cus * std::max(1u, std::min(8u, (160u * 1024u) / std::max(lds2, 1u))));
        } else { c->blocks = std::min(blocks, fit_blocks); c->blocks2 = std::min(blocks2 ? blocks2 : fit_blocks2, fit_blocks2); }
        c->decode_gen = gen; c->user_geometry = true;
    }
    return 0;
}

extern "C" int divans_gpu_codec_set_encode_path(divans_gpu_codec* c, uint32_t path) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (path > 2u) return fail(DIVANS_GPU_EINVAL, "path must be 0 (automatic), 1 (streaming) or 2 (bucketed)");
    if (path == 2u && !c->bucket_ok && !c->bucket_mix_ok)
        return fail(DIVANS_GPU_EINVAL, "the bucketed encoder needs mixing value 4 everywhere, streams of at most 65536 bytes and either no context map and no mixing or dynamic mixing with one literal block type");
    c->encode_path = path;
    return 0;
}

extern "C" int divans_gpu_codec_set_bucket_batch(divans_gpu_codec* c, uint32_t streams) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (streams == 0 || streams >= (1u << 24)) return fail(DIVANS_GPU_EINVAL, "bucket batch must be in [1, 2^24)");
    c->bucket_mix_batch = streams;
    return 0;
}

// Decoder generation and the geometry of lit_decode2_kernel: rows[i] of the four direct-mapped caches (high stride, high
// context-map, low stride, low context-map rows; 0 = not cached, else a power of two in [4, 256]), their hash shifts, and
// the persistent grid (0 = keep).
static int set_decoder_impl(divans_gpu_codec* c, uint32_t generation, const uint32_t rows[4], const uint32_t shifts[4], uint32_t blocks, bool by_user);
extern "C" int divans_gpu_codec_set_decoder(divans_gpu_codec* c, uint32_t generation, const uint32_t rows[4], const uint32_t shifts[4], uint32_t blocks) {
    return set_decoder_impl(c, generation, rows, shifts, blocks, true);
}
// by_user = false: the grid follows divans_gpu_codec_set_geometry; who chooses between direct-mapped and 2-way caches per batch
// (dm_auto) and whether a later set_block_types keeps the geometry (user_geometry) stay as they were
static int set_decoder_impl(divans_gpu_codec* c, uint32_t generation, const uint32_t rows[4], const uint32_t shifts[4], uint32_t blocks, bool by_user) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (generation < 1u || generation > 4u) return fail(DIVANS_GPU_EINVAL, "decoder generation must be 1, 2 (second generation, direct-mapped caches), 3 (second generation, 2-way caches) or 4 (one lane per stream)");
    HIP_TRY(hipSetDevice(c->device));
#if !DIVANS_WITH_EXPERIMENTAL_DECODERS
    // generations 1 and 4 lost their measurements (lit_kernels.h); this build keeps generation 1 only as the decoder of wrap-checked
    // speeds and of the call-by-call interface, where the library selects it by itself
    if (generation == 4u || (generation == 1u && by_user))
        return fail(DIVANS_GPU_EINVAL, "decoder generations 1 and 4 are experiment builds (DIVANS_WITH_EXPERIMENTAL_DECODERS=1 python divans_amd/build.py --force)");
    if (by_user) c->user_geometry = true;        // (only a call that is accepted changes the codec)
#else
    if (by_user) c->user_geometry = true;
    if (generation == 4u) {
        // lit_decode_t.hip: rows[i] slots of the four direct-mapped per-stream caches (0 = one slot, the staging buffer the kernel needs anyway)
        if (c->geom.wrap_check) return fail(DIVANS_GPU_EINVAL, "generation 4 does not run speeds whose row totals leave i16");
        if (c->geom.total_rows >= 0x7fffu) return fail(DIVANS_GPU_EINVAL, "generation 4 needs fewer than 32767 rows per stream");
        uint32_t lg2 = c->t_log2, sh = c->t_shift;
        if (rows && shifts) {
            lg2 = 0; sh = 0;
            for (int i = 0; i < 4; ++i) {
                if (rows[i]) {
                    if (rows[i] > 256u || (rows[i] & (rows[i] - 1u))) return fail(DIVANS_GPU_EINVAL, "cache rows must be 0 or a power of two up to 256");
                    uint32_t l = 0; while ((1u << l) < rows[i]) ++l;
                    lg2 |= (l + 1u) << (8 * i);
                }
                if (shifts[i] > 31u) return fail(DIVANS_GPU_EINVAL, "hash shift must be below 32");
                sh |= shifts[i] << (8 * i);
            }
        }
        const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
        const uint32_t per_wg = 64u * lit_decode_t_stream_lds(lg2, c->mix) + 256u +
                                (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
        const uint32_t fit = (160u * 1024u) / per_wg;
        if (fit == 0u) return fail(DIVANS_GPU_EINVAL, "these caches do not fit the 160 KB of LDS");
        uint32_t nb = blocks ? blocks : (c->blocks_t ? c->blocks_t : c->num_cus * 4u);      // default: one wave per SIMD
        nb = std::min(nb, c->num_cus * std::min(32u, fit));
        c->t_log2 = lg2; c->t_shift = sh; c->blocks_t = std::max(1u, nb);
        c->decode_gen = 4u;
        if (by_user) c->dm_auto = false;
        return 0;
    }
#endif
    const bool two_way = generation == 3u;
    if (by_user) c->dm_auto = false;
    if (generation == 3u) generation = 2u;
    c->dm_shift = (c->dm_shift & 0x7fffffffu) | (two_way ? 0x80000000u : 0u);
    if (generation == 2u && rows && shifts) {
        uint32_t lg2 = 0, sh = 0;
        for (int i = 0; i < 4; ++i) {
            if (rows[i]) {
                if (rows[i] < 4u || rows[i] > 256u || (rows[i] & (rows[i] - 1u))) return fail(DIVANS_GPU_EINVAL, "cache rows must be 0 or a power of two in [4, 256]");
                if (c->geom.total_rows >= 0x7fffu) return fail(DIVANS_GPU_EINVAL, "row caches need fewer than 32767 rows per stream");
                uint32_t l = 0; while ((1u << l) < rows[i]) ++l;
                lg2 |= (l + 1u) << (8 * i);
            }
            if (shifts[i] > 31u) return fail(DIVANS_GPU_EINVAL, "hash shift must be below 32");
            sh |= shifts[i] << (8 * i);
        }
        c->dm_log2 = lit_decode2_effective_caches(lg2, c->mix, false);   // only the cache sets that exist as kernel instances
        c->dm_shift = sh | (two_way ? 0x80000000u : 0u);
    }
    const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
    const uint32_t lds_per_wg = (LIT_THREADS / 16) * lit_decode2_stream_lds(c->dm_log2) + 256u /* byte ranks */ +
                                (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
    const uint32_t fit = (160u * 1024u) / lds_per_wg;
    if (fit == 0u) return fail(DIVANS_GPU_EINVAL, "these caches do not fit the 160 KB of LDS");
    uint32_t nb = blocks ? blocks : c->blocks2;
    nb = std::min(nb, c->num_cus * std::min(8u, fit));
    // (a larger grid than the tables were sized for: ensure_tables grows them at the next launch; they never shrink, so a codec
    // that alternates between small and large batches -- the lanes of divans_batch_* do -- does not free and reallocate gigabytes)
    if (generation == 2u) c->blocks2 = std::max(1u, nb);
    c->decode_gen = generation;
    return 0;
}

extern "C" void divans_gpu_trim(void) { table_pool_drop(-1); }

extern "C" int divans_gpu_codec_set_rans_split(divans_gpu_codec* c, uint32_t mode) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (mode > 2u) return fail(DIVANS_GPU_EINVAL, "rANS split must be 0 (automatic), 1 (one lane per chunk) or 2 (two lanes per chunk)");
    c->rans_split = mode;
    return 0;
}

extern "C" int divans_gpu_codec_set_byte_order(divans_gpu_codec* c, uint32_t order) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (order > 2u) return fail(DIVANS_GPU_EINVAL, "byte order must be 0 (learned from the codec's data), 1 (numeric) or 2 (English-text hint)");
    HIP_TRY(hipSetDevice(c->device));
    c->byte_order = order;
    c->rank_ready = false;             // 0: learn again from the next batch
    if (order == 2u) {
        static const BytePerm kTextOrder{};
        HIP_TRY(hipMemcpyAsync(c->d_rank, kTextOrder.rank, 256, hipMemcpyHostToDevice, c->stream));
        c->rank_ready = true;
    }
    return 0;
}

extern "C" int divans_gpu_codec_byte_order(divans_gpu_codec* c, uint32_t* mode, uint32_t* ready, uint8_t* rank256) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (mode) *mode = c->byte_order;
    if (ready) *ready = (c->byte_order != 1u && c->rank_ready) ? 1u : 0u;
    if (rank256) {
        if (c->byte_order != 1u && c->rank_ready) {
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipMemcpyAsync(rank256, c->d_rank, 256, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        } else for (int i = 0; i < 256; ++i) rank256[i] = (uint8_t)i;
    }
    return 0;
}

// Order 0: the first batch the codec sees -- the literal bytes an encode call is given, or the bytes a decode call produced -- is sampled
// (64 streams spread over the batch, their first 2 KiB) and the byte values are ranked by frequency; launches enqueued after that index
// the stride-1 tables by the rank.  Nothing is synchronised and nothing is ever read back: the order is a private layout of each launch.
static int maybe_learn_rank(divans_gpu_codec* c, const uint8_t* d_data, const uint64_t* d_offsets, const uint32_t* d_sizes, uint32_t n_streams, uint32_t stream_len) {
    if (c->byte_order != 0u || c->rank_ready || !c->d_rank || n_streams == 0u || c->geom.mm_uniform != 4) return 0;
    HIP_TRY(launch_learn_byte_rank(d_data, d_offsets, d_sizes, n_streams, stream_len, 64u, 2048u, c->d_rank, c->stream));
    c->rank_ready = true;
    return 0;
}

// A new policy while a call-by-call search is running: the search ends on the best placement it has seen (the candidate under test, whose
// time is not known yet, is given back) and the next qualifying call starts over under the new policy.
static int abort_search(divans_gpu_codec* c) {
    if (!c->ps.active) return 0;
    if (c->ps.best_tm.p) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        table_free(c->tm, false);
        c->tm = std::move(c->ps.best_tm); c->ps.best_tm = TableMem();
        c->d_tables = c->tm.p;
    }
    c->ps.active = c->ps.pending = false;
    return 0;
}

extern "C" int divans_gpu_codec_tune_tables(divans_gpu_codec* c, uint32_t candidates) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (candidates > 16u) return fail(DIVANS_GPU_EINVAL, "candidates must be in [0, 16] (0 = the library's policy, 1 = off)");
    { const int rc = abort_search(c); if (rc) return rc; }
    c->table_candidates = candidates;
    c->eager_tune = candidates >= 2u;
    c->tables_tuned = false;
    return 0;
}

extern "C" int divans_gpu_codec_search_tables(divans_gpu_codec* c, uint32_t candidates) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (candidates > 16u) return fail(DIVANS_GPU_EINVAL, "candidates must be in [0, 16] (0 = the library's policy, 1 = off)");
    { const int rc = abort_search(c); if (rc) return rc; }
    c->table_candidates = candidates;
    c->eager_tune = false;
    c->tables_tuned = false;
    return 0;
}

// The one placement policy: tables of 2 GiB and more -- the persistent grids of whole-GPU batches, where a
// slow placement costs 10-20 % of every later decode -- are tried on up to kDefaultTableCandidates placements, ONE PER CALL: each of the
// codec's first decode calls that fill half the grid runs on one placement and the next such call reads its time (placement_step), so no
// call decodes twice and the first one costs what every later one costs.  divans_gpu_codec_tune_tables(c, k >= 2) is the eager form: all
// k on the first such call.  Smaller tables (tests) are not tuned, and the lanes of divans_batch_* switch it off.
constexpr uint32_t kDefaultTableCandidates = 12u;
static uint32_t table_candidates_of(const divans_gpu_codec* c) {
    if (c->table_candidates) return c->table_candidates;
    // (before the tables exist: the size they will have)
    const size_t bytes = c->tm.bytes ? c->tm.bytes : (size_t)resident_groups(c) * c->geom.total_rows * 32u;
    return bytes >= ((size_t)2 << 30) ? kDefaultTableCandidates : 1u;
}

extern "C" int divans_gpu_codec_table_placement(divans_gpu_codec* c, divans_gpu_table_placement* out) {
    if (!c || !out) return fail(DIVANS_GPU_EINVAL, "null argument");
    *out = c->placement;
    out->policy_candidates = table_candidates_of(c);
    out->searching = (c->ps.active && !c->tables_tuned) ? 1u : 0u;
    return 0;
}

extern "C" int divans_gpu_table_memory(divans_gpu_table_memory_info* out) {
    if (!out) return fail(DIVANS_GPU_EINVAL, "null argument");
    out->va_reserved_bytes = g_va_reserved.load(); out->va_cap_bytes = g_va_cap.load();
    std::lock_guard<std::mutex> lock(g_table_pool_mu);
    out->idle_ranges = (uint32_t)g_table_pool.size(); out->idle_bytes = 0;
    for (const TableMem& m : g_table_pool) out->idle_bytes += m.va_bytes;
    return 0;
}
extern "C" void divans_gpu_set_table_va_cap(uint64_t bytes) { g_va_cap.store(bytes); }

static bool valid_cache_rows(uint32_t r) { return r == 0 || (r >= 16 && r <= 256 && (r & (r - 1)) == 0); }

extern "C" int divans_gpu_codec_set_geometry(divans_gpu_codec* c, uint32_t blocks, uint32_t cache_rows) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    c->user_geometry = true;
    if (cache_rows != 0xffffffffu) {
        if (!valid_cache_rows(cache_rows)) return fail(DIVANS_GPU_EINVAL, "cache_rows must be 0 or a power of two in [16, 256]");
        if (cache_rows && c->geom.total_rows >= 0x7fffu) return fail(DIVANS_GPU_EINVAL, "row cache needs fewer than 32767 rows per stream");
        c->cache_high = cache_rows; c->cache_low = 0; c->cache_unified = cache_rows != 0;
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
        c->decode_gen = 1;   // these are the caches of the first-generation decoder (and of the streaming model kernel)
#endif                       // (default build: of the streaming model kernel only; the decoder stays generation 2 / 3)
    }
    if (blocks) {
        c->blocks = blocks;
        if (c->blocks2) {    // the second-generation decoder follows, as far as its LDS use allows
            const uint32_t gen = c->decode_gen;
            int rc = set_decoder_impl(c, (c->dm_shift >> 31) ? 3 : 2, nullptr, nullptr, blocks, false); if (rc) return rc;   // grid only: dm_auto stays
            c->decode_gen = gen;
        }
    }
    return 0;
}

extern "C" int divans_gpu_codec_set_split_cache(divans_gpu_codec* c, uint32_t high_rows, uint32_t low_rows) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    if (!valid_cache_rows(high_rows) || !valid_cache_rows(low_rows)) return fail(DIVANS_GPU_EINVAL, "cache rows must be 0 or a power of two in [16, 256]");
    if (high_rows == 0 && low_rows != 0) return fail(DIVANS_GPU_EINVAL, "a low-nibble cache needs a high-nibble cache");
    if ((high_rows || low_rows) && c->geom.total_rows >= 0x7fffu) return fail(DIVANS_GPU_EINVAL, "row cache needs fewer than 32767 rows per stream");
    c->cache_high = high_rows; c->cache_low = low_rows; c->cache_unified = false;
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
    c->decode_gen = 1;
#endif
    c->user_geometry = true;
    return 0;
}

extern "C" size_t divans_gpu_lit_encode_bound(size_t n) {
    // per 65 536-symbol chunk: 16 bytes of final states; every symbol grows a state by at most 15 bits
    // (freq >= 1 of 2^15) and each 32-bit word emitted removes 32, so words <= ceil(15 * nsym / 32) + 2 per chunk
    const size_t nsym = 2 * n;
    const size_t nchunks = (nsym + 65535) / 65536;
    const size_t words = (15 * nsym + 31) / 32 + 2 * nchunks;
    const size_t bytes = 16 * nchunks + 4 * words;
    return (bytes + 15) & ~(size_t)15;
}

// Encoder pass 1: (start | freq << 16) per nibble, position order.  `after(first, count, view)` is called once the model kernels
// of streams [first, first + count) are enqueued (the two-model bucketed pass works through the batch in sub-batches whose
// work arrays it reuses, so whatever consumes the pairs has to be enqueued in between); the bucketed passes leave the pairs in
// their own work arrays (the unsort is in place), the streaming kernels in c->d_sf.  Records ev[0] before the first kernel.
template <class After>
static int model_pass(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets, const uint32_t* d_in_sizes,
                      uint32_t stream_len, uint32_t n_streams, const uint32_t* d_seg_begin, const divans_lit_segment* d_segs, After&& after) {
    int rc = maybe_learn_rank(c, d_in, d_in_offsets, d_in_sizes, n_streams, stream_len);     // byte order 0: the decoder's table order follows the data the codec sees
    if (rc) return rc;
    SfView view;
    if (use_bucket(c) && n_streams < (1u << 24) && !d_segs) {   // segment lists (context reloads between Literal commands) go through the streaming kernels
        BucketBatch k;
        std::memset(&k, 0, sizeof(k));
        // launch sequences of at most bucket_mix_batch streams (work arrays are sized for one of them: 11.2 bytes per input byte of
        // 32 768 streams instead of the whole batch's; whatever consumes the pairs is enqueued in between, as in the two-model pass)
        const uint32_t sub = std::min(n_streams, c->bucket_mix_batch);
        rc = ensure_bucket(c, sub, k); if (rc) return rc;
        k.stream_len = stream_len; k.max_stream_len = c->max_stream_len;
        k.pieces = bucket_pieces(c); k.slot = bucket_slot(c);
        k.sf = (uint32_t*)k.sfs; k.sf_stride = 2u * k.slot;     // bucket_unsort_kernel works in place
        k.inc = c->geom.inc0; k.lim = c->geom.lim0;
        view.sf = k.sf; view.stride = k.sf_stride;
        view.spare = (uint8_t*)k.inv; view.spare_bytes = (size_t)sub * k.slot * 3u;
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        for (uint32_t s0 = 0; s0 < n_streams; s0 += sub) {
            k.n_streams = std::min(sub, n_streams - s0);
            k.in = d_in_offsets ? d_in : d_in + (size_t)s0 * stream_len;
            k.in_offsets = d_in_offsets ? d_in_offsets + s0 : nullptr;
            k.in_sizes = d_in_sizes ? d_in_sizes + s0 : nullptr;
            HIP_TRY(launch_bucket_model(k, c->num_cus * 4u, c->stream));
            rc = after(s0, k.n_streams, view); if (rc) return rc;
        }
        return 0;
    }
    if (use_bucket_mix(c) && !d_segs) {
        MixBucketBatch k;
        std::memset(&k, 0, sizeof(k));
        // as many streams per launch sequence as the device has room for: a bucket is a serial chain, and the longest
        // ones (half a stream under one context) only stop dominating a sequence when it holds streams by the ten thousand
        uint32_t sub = std::min(n_streams, c->bucket_mix_batch);
        while ((rc = ensure_bucket_mix(c, sub, k)) == DIVANS_GPU_ENOMEM && sub > 1024u) { (void)hipGetLastError(); sub = (sub + 1u) / 2u; }
        if (rc) return rc;
        k.blob = c->d_blob; k.stream_len = stream_len; k.max_stream_len = c->max_stream_len; k.pieces = bucket_pieces(c);
        k.slot = bucket_slot(c); k.pos_stride = k.slot;
        k.sf = (uint32_t*)k.xs[0]; k.sf_stride = 2u * k.slot;  // mix_weights_kernel writes the pairs over the stride model's entries
        k.inc0 = c->geom.inc0; k.lim0 = c->geom.lim0; k.inc2 = c->geom.inc2; k.lim2 = c->geom.lim2; k.inc3 = c->geom.inc3; k.lim3 = c->geom.lim3;
        view.sf = k.sf; view.stride = k.sf_stride;
        view.spare = (uint8_t*)k.inv;
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        for (uint32_t s0 = 0; s0 < n_streams; s0 += sub) {
            k.n_streams = std::min(sub, n_streams - s0);
            k.in = d_in_offsets ? d_in : d_in + (size_t)s0 * stream_len;
            k.in_offsets = d_in_offsets ? d_in_offsets + s0 : nullptr;
            k.in_sizes = d_in_sizes ? d_in_sizes + s0 : nullptr;
            HIP_TRY(launch_bucket_mix_model(k, c->num_cus, c->stream));
            view.spare_bytes = (size_t)sub * k.slot * 4u;
            rc = after(s0, k.n_streams, view); if (rc) return rc;
        }
        return 0;
    }
    rc = ensure_sf(c, n_streams); if (rc) return rc;
    rc = ensure_tables(c); if (rc) return rc;
    LitBatch b;
    std::memset(&b, 0, sizeof(b));
    b.blob = c->d_blob; b.geom = c->geom; b.tables = c->d_tables;
    b.n_streams = n_streams; b.stream_len = stream_len; b.max_stream_len = c->max_stream_len;
    b.in = d_in; b.in_offsets = d_in_offsets; b.in_sizes = d_in_sizes;
    b.sf = c->d_sf; b.status = c->d_status;
    b.seg_begin = d_seg_begin; b.segs = (const LitSegment*)d_segs;
    set_cache_fields(c, b);
    if (d_segs && b.cache_mode != 2u && b.cache_mode != 0u) return fail(DIVANS_GPU_EINVAL, "segment lists need the default (high-nibble-row) cache or none");
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    ++c->table_launch_seq;
    HIP_TRY(launch_model_encode(b, c->mix, c->blocks, c->stream));
    view.sf = c->d_sf; view.stride = 2u * c->max_stream_len;
    return after(0u, n_streams, view);
}

// event pairs around the rANS launches of one encode call (one pair per sub-batch of the model pass)
static int rans_event_pair(divans_gpu_codec* c, size_t i, hipEvent_t*& pair) {
    while (c->ev_rans.size() < 2u * (i + 1u)) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        c->ev_rans.push_back(e);
    }
    pair = &c->ev_rans[2u * i];
    return 0;
}

// Encoder pass 2 for `count` streams whose pairs the model pass left in `v`: coded bytes right-aligned in their slots at `out`
// (`out_base` = where `out` lies in the buffer the reported offsets refer to).
static int rans_pass(divans_gpu_codec* c, const SfView& v, uint32_t count, uint32_t stream_len, const uint32_t* d_in_sizes, uint8_t* out, uint64_t out_base,
                     uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes, uint32_t* d_chunk_bytes, uint32_t max_chunks, size_t& pairs) {
    const bool chunk_lanes = c->max_stream_len > 32768u && c->max_stream_len <= 65536u;   // two chunks per stream slot: one lane per chunk
    RansBatch r;
    r.sf = v.sf; r.sf_stride = v.stride; r.n_streams = count; r.stream_len = stream_len; r.max_stream_len = c->max_stream_len;
    r.in_sizes = d_in_sizes;
    r.out = out; r.out_base = out_base; r.out_slot = out_slot;
    r.out_offsets = d_out_offsets; r.out_sizes = d_out_sizes;
    r.status = c->d_status; r.chunk_bytes = d_chunk_bytes; r.max_chunks = max_chunks;
    r.scratch = nullptr; r.scratch_stride = 0; r.chunk0_sizes = nullptr;
    if (chunk_lanes) { int rr = ensure_rans_scratch(c, count, v, r); if (rr) return rr; }
    // two lanes per chunk (one per rANS state) when whole-chunk lanes would leave SIMDs without a wave: 4 lanes per stream on at most
    // one wave per SIMD (rans_encode2_split_kernel; c->rans_split: 0 automatic, 1 never, 2 always -- a test / measurement knob)
    r.split_states = (c->rans_split == 2u || (c->rans_split == 0u && (uint64_t)count * 4u <= (uint64_t)c->num_cus * 4u * 64u)) ? 1u : 0u;
    hipEvent_t* pair = nullptr;
    int rr = rans_event_pair(c, pairs, pair); if (rr) return rr;
    HIP_TRY(hipEventRecord(pair[0], c->stream));
    HIP_TRY(launch_rans_encode(r, c->stream));
    HIP_TRY(hipEventRecord(pair[1], c->stream));
    ++pairs;
    return 0;
}

static int encode_batch_impl(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                             const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                             uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes,
                             uint32_t* d_chunk_bytes, uint32_t max_chunks,
                             const uint32_t* d_seg_begin = nullptr, const divans_lit_segment* d_segs = nullptr);

extern "C" int divans_gpu_lit_encode_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                           const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                           uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes) {
    return encode_batch_impl(c, d_in, d_in_offsets, d_in_sizes, stream_len, n_streams, d_out, out_slot, d_out_offsets,
                             d_out_sizes, nullptr, 0);
}

static int encode_batch_impl(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                             const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                             uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes,
                             uint32_t* d_chunk_bytes, uint32_t max_chunks,
                             const uint32_t* d_seg_begin, const divans_lit_segment* d_segs) {
    if (!c || !d_in || !d_out || !d_out_offsets || !d_out_sizes) return fail(DIVANS_GPU_EINVAL, "null argument");
    if ((d_seg_begin == nullptr) != (d_segs == nullptr)) return fail(DIVANS_GPU_EINVAL, "seg_begin and segs go together");
    if (n_streams == 0) return 0;
    if ((d_in_offsets == nullptr) != (d_in_sizes == nullptr)) return fail(DIVANS_GPU_EINVAL, "offsets and sizes go together");
    if (stream_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "stream_len exceeds the codec's max_stream_len");
    if (out_slot % 16 != 0 || out_slot < divans_gpu_lit_encode_bound(d_in_sizes ? c->max_stream_len : stream_len))
        return fail(DIVANS_GPU_ECAP, "out_slot must be a multiple of 16 and >= divans_gpu_lit_encode_bound()");
    HIP_TRY(hipSetDevice(c->device));
    size_t pairs = 0;
    int rc = model_pass(c, d_in, d_in_offsets, d_in_sizes, stream_len, n_streams, d_seg_begin, d_segs,
                        [&](uint32_t first, uint32_t count, const SfView& v) -> int {
        return rans_pass(c, v, count, stream_len, d_in_sizes ? d_in_sizes + first : nullptr, d_out + (uint64_t)first * out_slot, (uint64_t)first * out_slot, out_slot,
                         d_out_offsets + first, d_out_sizes + first, d_chunk_bytes ? d_chunk_bytes + (size_t)first * max_chunks : nullptr, max_chunks, pairs);
    });
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    c->rans_pairs = pairs; c->enc_spans = 0;
    c->timing_pending_enc = true;
    return 0;
}

// Coded streams contiguous, through slots that hold ONE sub-batch (include/divans_gpu.h).
extern "C" int divans_gpu_lit_encode_packed(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets, const uint32_t* d_in_sizes,
                                            uint32_t stream_len, uint32_t n_streams, uint8_t* d_packed, uint64_t packed_cap,
                                            uint64_t* d_packed_offsets, uint32_t* d_sizes, uint64_t* d_total, uint32_t sub_batch) {
    if (!c || !d_in || !d_packed || !d_packed_offsets || !d_sizes || !d_total) return fail(DIVANS_GPU_EINVAL, "null argument");
    if ((d_in_offsets == nullptr) != (d_in_sizes == nullptr)) return fail(DIVANS_GPU_EINVAL, "offsets and sizes go together");
    if (stream_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "stream_len exceeds the codec's max_stream_len");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(uint64_t), c->stream));
    if (n_streams == 0) return 0;
    const uint64_t slot = divans_gpu_lit_encode_bound(d_in_sizes ? c->max_stream_len : stream_len);
    uint32_t sub = std::min(n_streams, sub_batch ? sub_batch : c->bucket_mix_batch);
    while ((uint64_t)sub * slot > ((uint64_t)1 << 36) && sub > 1024u) sub = (sub + 1u) / 2u;      // at most 64 GiB of slots
    if ((size_t)sub * slot > c->slots_bytes) {
        if (c->d_slots) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_slots)); c->d_slots = nullptr; c->slots_bytes = 0; }
        if (device_alloc((void**)&c->d_slots, (size_t)sub * slot) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(output slots of a sub-batch) failed");
        c->slots_bytes = (size_t)sub * slot;
    }
    if (sub > c->slot_off_cap) {
        if (c->d_slot_off) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_slot_off)); c->d_slot_off = nullptr; c->slot_off_cap = 0; }
        if (hipMalloc(&c->d_slot_off, (size_t)sub * sizeof(uint64_t)) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(slot offsets) failed");
        c->slot_off_cap = sub;
    }
    const size_t n_sub = (n_streams + sub - 1u) / sub;
    while (c->ev_span.size() < 3u * n_sub) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreate(&e)); c->ev_span.push_back(e); }
    size_t pairs = 0, span = 0;
    for (uint32_t b0 = 0; b0 < n_streams; b0 += sub, ++span) {
        const uint32_t m = std::min(sub, n_streams - b0);
        HIP_TRY(hipEventRecord(c->ev_span[3u * span], c->stream));
        int rc = model_pass(c, d_in_offsets ? d_in : d_in + (size_t)b0 * stream_len, d_in_offsets ? d_in_offsets + b0 : nullptr, d_in_sizes ? d_in_sizes + b0 : nullptr,
                            stream_len, m, nullptr, nullptr, [&](uint32_t first, uint32_t count, const SfView& v) -> int {
            return rans_pass(c, v, count, stream_len, d_in_sizes ? d_in_sizes + b0 + first : nullptr, c->d_slots + (uint64_t)first * slot, (uint64_t)first * slot, slot,
                             c->d_slot_off + first, d_sizes + b0 + first, nullptr, 0, pairs);
        });
        if (rc) return rc;
        HIP_TRY(hipEventRecord(c->ev_span[3u * span + 1u], c->stream));
        HIP_TRY(launch_pack(c->d_slots, c->d_slot_off, d_sizes + b0, m, d_packed, d_packed_offsets + b0, d_total, c->stream, true, packed_cap, c->d_status));
        HIP_TRY(hipEventRecord(c->ev_span[3u * span + 2u], c->stream));
    }
    c->rans_pairs = pairs; c->enc_spans = span;
    c->timing_pending_enc = true;
    return 0;
}

extern "C" int divans_gpu_lit_encode_batch_chunks(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                                  const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                                  uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes,
                                                  uint32_t* d_chunk_bytes, uint32_t max_chunks) {
    if (!d_chunk_bytes || max_chunks < (2u * (uint64_t)stream_len + 65535u) / 65536u) return fail(DIVANS_GPU_EINVAL, "max_chunks too small");
    return encode_batch_impl(c, d_in, d_in_offsets, d_in_sizes, stream_len, n_streams, d_out, out_slot, d_out_offsets, d_out_sizes,
                             d_chunk_bytes, max_chunks);
}

extern "C" int divans_gpu_lit_encode_segments_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                                    const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams,
                                                    const uint32_t* d_seg_begin, const divans_lit_segment* d_segs,
                                                    uint8_t* d_out, uint64_t out_slot, uint64_t* d_out_offsets, uint32_t* d_out_sizes) {
    if (!d_seg_begin || !d_segs) return fail(DIVANS_GPU_EINVAL, "null segment list");
    return encode_batch_impl(c, d_in, d_in_offsets, d_in_sizes, stream_len, n_streams, d_out, out_slot, d_out_offsets,
                             d_out_sizes, nullptr, 0, d_seg_begin, d_segs);
}

extern "C" int divans_gpu_lit_model_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                          const uint32_t* d_in_sizes, uint32_t stream_len, uint32_t n_streams, uint32_t* d_pairs) {
    if (!c || !d_in || !d_pairs) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    if ((d_in_offsets == nullptr) != (d_in_sizes == nullptr)) return fail(DIVANS_GPU_EINVAL, "offsets and sizes go together");
    if (stream_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "stream_len exceeds the codec's max_stream_len");
    HIP_TRY(hipSetDevice(c->device));
    const size_t row = (size_t)2u * c->max_stream_len * sizeof(uint32_t);
    int rc = model_pass(c, d_in, d_in_offsets, d_in_sizes, stream_len, n_streams, nullptr, nullptr,
                        [&](uint32_t first, uint32_t count, const SfView& v) -> int {
        HIP_TRY(hipMemcpy2DAsync(d_pairs + (size_t)first * 2u * c->max_stream_len, row, v.sf, (size_t)v.stride * sizeof(uint32_t), row, count,
                                 hipMemcpyDeviceToDevice, c->stream));
        return 0;
    });
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    c->rans_pairs = 0; c->enc_spans = 0; c->last_pack_ms = 0.f;      // (not the spans of an earlier divans_gpu_lit_encode_packed call)
    c->timing_pending_enc = true;
    return 0;
}

static int decode_batch_impl(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                             const uint32_t* d_in_sizes, uint32_t n_streams, uint8_t* d_out,
                             const uint64_t* d_out_offsets, const uint32_t* d_out_sizes, uint32_t stream_len,
                             const uint32_t* d_seg_begin, const divans_lit_segment* d_segs);

extern "C" int divans_gpu_lit_decode_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                           const uint32_t* d_in_sizes, uint32_t n_streams, uint8_t* d_out,
                                           const uint64_t* d_out_offsets, const uint32_t* d_out_sizes, uint32_t stream_len) {
    return decode_batch_impl(c, d_in, d_in_offsets, d_in_sizes, n_streams, d_out, d_out_offsets, d_out_sizes, stream_len, nullptr, nullptr);
}

extern "C" int divans_gpu_lit_decode_segments_batch(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                                                    const uint32_t* d_in_sizes, uint32_t n_streams,
                                                    const uint32_t* d_seg_begin, const divans_lit_segment* d_segs, uint8_t* d_out,
                                                    const uint64_t* d_out_offsets, const uint32_t* d_out_sizes, uint32_t stream_len) {
    if (!d_seg_begin || !d_segs) return fail(DIVANS_GPU_EINVAL, "null segment list");
    return decode_batch_impl(c, d_in, d_in_offsets, d_in_sizes, n_streams, d_out, d_out_offsets, d_out_sizes, stream_len, d_seg_begin, d_segs);
}

// Cache organisation and grid of lit_decode2_kernel for a batch of n_streams (also what divans_gpu_codec_row_replay launches with)
static void decode2_shape(divans_gpu_codec* c, uint32_t n_streams, bool segs, LitBatch& b, uint32_t& grid) {
    b.dm_log2 = lit_decode2_effective_caches(c->dm_log2, c->mix, segs); b.dm_shift = c->dm_shift;
    // A batch that is resident all at once runs at the latency of a stream's dependency chain, and the direct-mapped lookup is
    // the shorter chain (16 384 streams: 60.3 vs 63.5 ms, mixing 138 vs 142); only a batch that keeps the grid busy for several
    // rounds gains from the 2-way sets' fewer misses (profiles/r03c_small_batch_geometry.txt)
    grid = c->blocks2;
    if (c->dm_auto && n_streams <= c->blocks2 * groups_per_block(c)) {
        b.dm_shift &= 0x7fffffffu;
        // ... and such a batch leaves LDS unused that shortens the chain further: the largest high-row caches under which every
        // stream is still resident at once (profiles/r04c_small_batch_caches.txt: 16 384 streams 60.1 -> 54.2 ms with 64 instead of
        // 32 high rows, 8192: 44.5 -> 40.4; mixing 20 480 streams 162.8 -> 152.5 ms with 32 + 16, 16 384: 134.4 -> 123.4 with 32 + 32;
        // low-row caches do not pay even there)
        if (!segs && c->geom.total_rows < 0x7fffu) {
            const bool mask_in_lds = !(c->geom.mm_uniform == 0 || c->geom.mm_uniform == 4);
            const uint32_t fixed = 256u + (c->geom.ctx_const < 0 ? LIT_BLOB_CTXF + LIT_CTXF_BYTES * c->geom.n_btypes : 0u) + (mask_in_lds ? 8192u : 0u);
            const uint32_t mix_cands[2] = {6u | (6u << 8), 6u | (5u << 8)};      // (log2 rows + 1) per table: high stride | FirstNibble << 8
            const uint32_t plain_cands[1] = {7u};
            const uint32_t* cands = c->mix ? mix_cands : plain_cands;
            for (uint32_t i = 0; i < (c->mix ? 2u : 1u); ++i) {
                const uint32_t lg = lit_decode2_effective_caches(cands[i], c->mix, false);
                const uint32_t per_wg = (LIT_THREADS / 16) * lit_decode2_stream_lds(lg) + fixed;
                const uint32_t fit = std::min(8u, (160u * 1024u) / per_wg);
                if (lg == cands[i] && (uint64_t)n_streams <= (uint64_t)c->num_cus * fit * groups_per_block(c)) { b.dm_log2 = lg; break; }
            }
        }
        grid = std::min(grid, (n_streams + groups_per_block(c) - 1u) / groups_per_block(c));
    }
    b.cache_bytes_per_wg = (LIT_THREADS / 16) * lit_decode2_stream_lds(b.dm_log2);
}

static void placement_finish(divans_gpu_codec* c) {
    c->placement.best_ms = c->ps.best_ms; c->placement.worst_ms = c->ps.worst_ms; c->placement.kept_chunks = c->tm.chunks.empty() ? 0u : 1u;
    c->tables_tuned = true; c->ps.active = false; c->ps.pending = false;
}

// One step of the library's placement search, run by a qualifying decode call BEFORE its launch.  Reads the time of the previous
// qualifying call's launch (the one wait of the search: for a launch of an EARLIER call -- this call's own launch is never waited for),
// keeps the faster of {best so far, that candidate}, gives the loser's memory back and maps the next candidate for this call's launch;
// after the last candidate the best stays for good.  *measure: this call's launch is a measurement (the caller records ps.e0 / ps.e1).
static int placement_step(divans_gpu_codec* c, uint32_t want, uint32_t n_streams, uint32_t stream_len, bool* measure) {
    auto& ps = c->ps;
    *measure = false;
    if (!ps.active) {        // the first qualifying call: the tables as ensure_tables allocated them are placement 0
        ps.active = true; ps.pending = false; ps.sig_streams = n_streams; ps.sig_len = stream_len;
        ps.best_ms = ps.worst_ms = 0.f;
        c->placement = {0u, 0u, 0.f, 0.f, 0.f, 0u, 0u};
        *measure = true;
        return 0;
    }
    if (n_streams != ps.sig_streams || stream_len != ps.sig_len) return 0;      // another shape: decoded on the placement in use, not compared
    if (ps.pending) {
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(ps.e1));
        HIP_TRY(hipEventElapsedTime(&t, ps.e0, ps.e1));
        ps.pending = false;
        // a table is only given back once nothing enqueued can touch it: the measured launch is complete; anything enqueued on the tables
        // after it (decode calls of other shapes, streaming encoder passes) is waited for
        if (c->table_launch_seq != ps.seq_at_measure) HIP_TRY(hipStreamSynchronize(c->stream));
        const bool first = c->placement.tried == 0u;
        if (first) c->placement.first_ms = t;
        ++c->placement.tried;
        ps.worst_ms = first ? t : std::max(ps.worst_ms, t);
        if (first || t < ps.best_ms) { ps.best_ms = t; table_free(ps.best_tm, false); }
        else { table_free(c->tm, false); c->tm = std::move(ps.best_tm); ps.best_tm = TableMem(); c->d_tables = c->tm.p; }
        c->placement.best_ms = ps.best_ms; c->placement.worst_ms = ps.worst_ms;
    }
    if (c->placement.tried >= want) { placement_finish(c); return 0; }
    TableMem cand;
    // alternately one hipMalloc block and chunks mapped side by side: which kind is faster differs from box to box
    if (table_alloc(c->device, c->tm.bytes, cand, true, (c->placement.tried & 1u) != 0u) != hipSuccess) { (void)hipGetLastError(); placement_finish(c); return 0; }   // no room for a second copy
    ps.best_tm = std::move(c->tm);
    c->tm = std::move(cand);
    c->d_tables = c->tm.p;
    *measure = true;
    return 0;
}

static int decode_batch_impl(divans_gpu_codec* c, const uint8_t* d_in, const uint64_t* d_in_offsets,
                             const uint32_t* d_in_sizes, uint32_t n_streams, uint8_t* d_out,
                             const uint64_t* d_out_offsets, const uint32_t* d_out_sizes, uint32_t stream_len,
                             const uint32_t* d_seg_begin, const divans_lit_segment* d_segs) {
    if (!c || !d_in || !d_in_offsets || !d_in_sizes || !d_out) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    if ((d_out_offsets == nullptr) != (d_out_sizes == nullptr)) return fail(DIVANS_GPU_EINVAL, "offsets and sizes go together");
    if (stream_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "stream_len exceeds the codec's max_stream_len");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_tables(c); if (rc) return rc;
    // The table placements are compared on launches that differ in nothing else: the byte order is settled first (order 0 learns it from
    // the first batch -- an earlier encode call's input or this call's output)
    const bool order_settled = !(c->byte_order == 0u && !c->rank_ready && c->geom.mm_uniform == 4);
    const uint32_t want = table_candidates_of(c);
    const bool qualifies = want > 1u && !c->tables_tuned && 2u * (uint64_t)n_streams >= resident_groups(c) && order_settled &&
                           !c->sp_started && !c->sd_started;      // (a stream coded call by call keeps its state IN the tables)
    const bool eager = c->eager_tune;                   // divans_gpu_codec_tune_tables(c, k >= 2): every placement on this call
    bool measure = false;
    if (qualifies && !eager) { rc = placement_step(c, want, n_streams, stream_len, &measure); if (rc) return rc; }
    LitBatch b;
    std::memset(&b, 0, sizeof(b));
    b.blob = c->d_blob; b.geom = c->geom; b.tables = c->d_tables;
    b.n_streams = n_streams; b.stream_len = stream_len; b.max_stream_len = c->max_stream_len;
    b.in = d_in; b.in_offsets = d_in_offsets; b.in_sizes = d_in_sizes;
    b.out = d_out; b.out_offsets = d_out_offsets; b.out_sizes = d_out_sizes; b.status = c->d_status;
    b.stream_bad = c->d_stream_flags;
    b.seg_begin = d_seg_begin; b.segs = (const LitSegment*)d_segs;
    b.byte_rank = (c->byte_order != 1u && c->rank_ready) ? c->d_rank : nullptr;
    set_cache_fields(c, b);
#if !DIVANS_WITH_EXPERIMENTAL_DECODERS
    if (!use_decode2(c) && (b.cache_mode == 1u || b.cache_mode == 3u)) {     // the generation-1 fallback of this build knows the high-row cache or none
        b.cache_mode = 2u; b.cache_rows_low = 0u;
        b.cache_bytes_per_wg = (LIT_THREADS / 16) * b.cache_rows_high * 34u;
    }
#endif
    if (d_segs && !use_decode2(c) && b.cache_mode != 2u && b.cache_mode != 0u) return fail(DIVANS_GPU_EINVAL, "segment lists need the default (high-nibble-row) cache or none");
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
    const bool transposed = use_decode_t(c) && !d_segs;      // (segment lists: the first generation's kernel)
#endif
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
    if (transposed) {
        b.dm_log2 = c->t_log2; b.dm_shift = c->t_shift;
        b.cache_bytes_per_wg = 64u * lit_decode_t_stream_lds(c->t_log2, c->mix);
        c->last_decode_grid = std::min(c->blocks_t, (n_streams + 63u) / 64u);
    } else
#endif
    if (use_decode2(c)) decode2_shape(c, n_streams, d_segs != nullptr, b, c->last_decode_grid);
    auto launch = [&]() -> int {
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        ++c->table_launch_seq;
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
        if (transposed) { lit_decode_t_kernel_name(b, c->mix, c->last_decode_kernel, sizeof(c->last_decode_kernel)); HIP_TRY(launch_decode_t(b, c->mix, c->last_decode_grid, c->stream)); }
        else
#endif
        if (use_decode2(c)) { lit_decode2_kernel_name(b, c->mix, c->last_decode_kernel, sizeof(c->last_decode_kernel)); HIP_TRY(launch_decode2(b, c->mix, c->last_decode_grid, c->stream)); }
        else { lit_decode_kernel_name(b, c->mix, c->last_decode_kernel, sizeof(c->last_decode_kernel)); HIP_TRY(launch_decode(b, c->mix, c->blocks, c->stream)); }
        HIP_TRY(hipEventRecord(c->ev[4], c->stream));
        return 0;
    };
    if (measure) HIP_TRY(hipEventRecord(c->ps.e0, c->stream));
    rc = launch(); if (rc) return rc;
    if (measure) { HIP_TRY(hipEventRecord(c->ps.e1, c->stream)); c->ps.pending = true; c->ps.seq_at_measure = c->table_launch_seq; }
    if (!order_settled) { rc = maybe_learn_rank(c, d_out, d_out_offsets, d_out_sizes, n_streams, stream_len); if (rc) return rc; }
    // divans_gpu_codec_tune_tables(c, k >= 2), the eager form: this batch is decoded once per candidate placement of the tables (the same
    // bytes come out every time) and the fastest placement stays.  Two copies of the tables are alive at most -- the best so far and the
    // candidate under test; a rejected one gives its memory back before the next is allocated.  (Stopping at the first placement
    // that is 5 % ahead of the slowest seen was tried and costs the mixing configurations 5 %: their times spread over three clusters, and a very
    // slow placement ends the search on a middling one -- profiles/r05a_bench_line_early_stop.json: 436.6 ms kept after 4, best of 12 is 415.)
    // This path synchronises the stream and decodes k times; the library's own policy (placement_step) does neither.
    if (qualifies && eager) {
        float best = 0.f;
        HIP_TRY(hipEventSynchronize(c->ev[4]));
        HIP_TRY(hipEventElapsedTime(&best, c->ev[3], c->ev[4]));
        float worst = best;
        c->placement = {0u, 1u, best, 0.f, 0.f, 0u, 0u};
        for (uint32_t k = 1; k < want; ++k) {
            TableMem cand;
            if (table_alloc(c->device, c->tm.bytes, cand, true, (k & 1u) != 0u) != hipSuccess) { (void)hipGetLastError(); break; }     // no room for a second copy: keep what we have
            std::swap(c->tm, cand);
            c->d_tables = c->tm.p; b.tables = c->d_tables;
            rc = launch();
            float t = 0.f;
            if (!rc && (hipEventSynchronize(c->ev[4]) != hipSuccess || hipEventElapsedTime(&t, c->ev[3], c->ev[4]) != hipSuccess)) rc = fail(DIVANS_GPU_EHIP, "timing a candidate table placement failed");
            if (exp_env("DIVANS_DEBUG_ALLOC")) fprintf(stderr, "[divans] table placement %u (%s): %.2f ms (best so far %.2f)\n", k, c->tm.chunks.empty() ? "one block" : "chunks", t, best);
            if (rc || t >= best) { std::swap(c->tm, cand); c->d_tables = c->tm.p; b.tables = c->d_tables; }    // the earlier one stays
            else best = t;
            if (!rc) { worst = std::max(worst, t); ++c->placement.tried; }
            if (rc) (void)hipStreamSynchronize(c->stream);      // a launch that failed half way may still be running on the candidate
            table_free(cand, false);      // the loser's memory goes back now (its address range stays reserved: the remap defect)
            if (rc) break;
        }
        if (rc) return rc;
        c->placement.best_ms = best; c->placement.worst_ms = worst; c->placement.kept_chunks = c->tm.chunks.empty() ? 0u : 1u;
        c->last_decode_ms = best; c->timing_pending_dec = false;
        c->tables_tuned = true;
        return 0;       // (the events hold the last candidate's time; last_decode_ms the kept one's)
    }
    c->timing_pending_dec = true;
    return 0;
}

// The memory side's own time for a batch's row traffic (launch_row_replay, lit_kernels.h): the literal bytes are given, every row the
// decoder of this configuration touches is read, blended and written back through the same caches, table layout, byte order and grid
// as divans_gpu_lit_decode_batch would use for the batch -- no entropy decoding, nothing that makes a byte wait for the one before it.
// A measurement aid (bench.py's roofline.request_ceiling), synchronous; stride-1 configurations without segment lists.
extern "C" int divans_gpu_codec_row_replay(divans_gpu_codec* c, const uint8_t* d_literals, const uint64_t* d_offsets, const uint32_t* d_sizes,
                                           uint32_t n_streams, uint32_t stream_len, float* ms) {
    if (!c || !d_literals || !ms) return fail(DIVANS_GPU_EINVAL, "null argument");
    if ((d_offsets == nullptr) != (d_sizes == nullptr)) return fail(DIVANS_GPU_EINVAL, "offsets and sizes go together");
    if (n_streams == 0 || stream_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "bad batch shape");
    if (!use_decode2(c) || c->geom.mm_uniform != 4) return fail(DIVANS_GPU_EINVAL, "row replay exists for the stride-1 configurations of the second-generation decoder");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_tables(c); if (rc) return rc;
    LitBatch b;
    std::memset(&b, 0, sizeof(b));
    b.blob = c->d_blob; b.geom = c->geom; b.tables = c->d_tables;
    b.n_streams = n_streams; b.stream_len = stream_len; b.max_stream_len = c->max_stream_len;
    b.in = d_literals; b.in_offsets = d_offsets; b.in_sizes = d_sizes;
    b.byte_rank = (c->byte_order != 1u && c->rank_ready) ? c->d_rank : nullptr;
    uint32_t grid = 0;
    decode2_shape(c, n_streams, false, b, grid);
    HIP_TRY(hipEventRecord(c->ev[3], c->stream));
    ++c->table_launch_seq;
    if (launch_row_replay(b, c->mix, grid, c->stream) != hipSuccess) { (void)hipGetLastError(); return fail(DIVANS_GPU_EINVAL, "no row-replay instance for this cache organisation"); }
    HIP_TRY(hipEventRecord(c->ev[4], c->stream));
    HIP_TRY(hipEventSynchronize(c->ev[4]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev[3], c->ev[4]));
    c->timing_pending_dec = false;
    return 0;
}

extern "C" int divans_gpu_pack_streams(divans_gpu_codec* c, const uint8_t* d_slots, const uint64_t* d_offsets,
                                       const uint32_t* d_sizes, uint32_t n_streams, uint8_t* d_packed,
                                       uint64_t* d_packed_offsets, uint64_t* d_total) {
    if (!c || !d_slots || !d_offsets || !d_sizes || !d_packed || !d_packed_offsets || !d_total) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_pack(d_slots, d_offsets, d_sizes, n_streams, d_packed, d_packed_offsets, d_total, c->stream));
    return 0;
}

extern "C" int divans_gpu_codec_info(divans_gpu_codec* c, divans_gpu_info* info) {
    if (!c || !info) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->timing_pending_enc && c->enc_spans) {      // divans_gpu_lit_encode_packed: per sub-batch (start, before the pack, end)
        HIP_TRY(hipEventSynchronize(c->ev_span[3u * c->enc_spans - 1u]));
        float coding = 0.f, pack = 0.f, rans = 0.f;
        for (size_t i = 0; i < c->enc_spans; ++i) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->ev_span[3u * i], c->ev_span[3u * i + 1u])); coding += t;
            HIP_TRY(hipEventElapsedTime(&t, c->ev_span[3u * i + 1u], c->ev_span[3u * i + 2u])); pack += t;
        }
        for (size_t i = 0; i < c->rans_pairs; ++i) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->ev_rans[2u * i], c->ev_rans[2u * i + 1u]));
            rans += t;
        }
        c->last_rans_ms = rans; c->last_model_ms = coding - rans; c->last_pack_ms = pack;
        c->timing_pending_enc = false;
    }
    if (c->timing_pending_enc) {
        HIP_TRY(hipEventSynchronize(c->ev[2]));
        float total = 0.f, rans = 0.f;
        HIP_TRY(hipEventElapsedTime(&total, c->ev[0], c->ev[2]));
        for (size_t i = 0; i < c->rans_pairs; ++i) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->ev_rans[2u * i], c->ev_rans[2u * i + 1u]));
            rans += t;
        }
        c->last_rans_ms = rans; c->last_model_ms = total - rans;
        c->timing_pending_enc = false;
    }
    if (c->timing_pending_dec) {
        HIP_TRY(hipEventSynchronize(c->ev[4]));
        HIP_TRY(hipEventElapsedTime(&c->last_decode_ms, c->ev[3], c->ev[4]));
        c->timing_pending_dec = false;
    }
    info->rows_per_stream = c->geom.total_rows;
    info->resident_groups = resident_groups(c);
    info->blocks = c->blocks; info->threads = LIT_THREADS;
    info->table_bytes = (uint64_t)resident_groups(c) * c->geom.total_rows * 32u;
    info->scratch_bytes = c->sf_bytes + c->bk_bytes + c->rs_bytes + c->slots_bytes;
    info->last_pack_ms = c->last_pack_ms;
    info->last_model_ms = c->last_model_ms; info->last_rans_ms = c->last_rans_ms; info->last_decode_ms = c->last_decode_ms;
    return 0;
}

// Which stream failed: a device array of at least n_streams bytes the decode entry points set to 1 for every stream that fails
// its integrity check (they never clear it; null switches it off).  The caller owns the array and its zeroing.
extern "C" int divans_gpu_codec_set_stream_flags(divans_gpu_codec* c, uint8_t* d_flags) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    c->d_stream_flags = d_flags;
    return 0;
}

// Sticky device status word of the batch entry points: waits for the codec's stream, returns the bits set since the last
// call and clears them.  bit 0: the rANS pass met an invalid (start,freq); bit 1: a decoded stream failed its integrity check.
extern "C" int divans_gpu_codec_status(divans_gpu_codec* c, uint32_t* status) {
    if (!c || !status) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t h = 0;
    HIP_TRY(hipMemcpyAsync(&h, c->d_status, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h) HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(h), c->stream));
    *status = h;
    return 0;
}

extern "C" int divans_gpu_codec_status_async(divans_gpu_codec* c, uint32_t* h_pinned_status) {
    if (!c || !h_pinned_status) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h_pinned_status, c->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    return 0;
}

extern "C" int divans_gpu_codec_last_decode_kernel(divans_gpu_codec* c, char* buf, size_t cap) {
    if (!c || !buf || !cap) return fail(DIVANS_GPU_EINVAL, "null argument");
    snprintf(buf, cap, "%s", c->last_decode_kernel);
    return 0;
}

extern "C" int divans_gpu_codec_clear_status(divans_gpu_codec* c) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    return 0;
}

extern "C" int divans_gpu_selftest_division(divans_gpu_codec* c, uint64_t* mismatches) {
    if (!c || !mismatches) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(*d)));
    HIP_TRY(hipMemsetAsync(d, 0, sizeof(*d), c->stream));
    HIP_TRY(launch_selftest_division(d, c->stream));
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d));
    *mismatches = h;
    return 0;
}

// Test entry points: the device primitives in isolation (tests/test_gpu_reference_unit_tests.py ports the reference's
// own unit tests onto them).
extern "C" int divans_gpu_selftest_cdf_ops(divans_gpu_codec* c, const uint32_t* ops, uint32_t n_ops, int32_t* out) {
    if (!c || !ops || !out || n_ops == 0) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t* d_ops = nullptr; int32_t* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_ops, (size_t)n_ops * 16u));
    if (hipMalloc(&d_out, (size_t)n_ops * 64u) != hipSuccess) { (void)hipFree(d_ops); return fail(DIVANS_GPU_ENOMEM, "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(d_ops, ops, (size_t)n_ops * 16u, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_selftest_cdf_ops(d_ops, n_ops, d_out, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n_ops * 64u, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_ops); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(DIVANS_GPU_EHIP, hipGetErrorString(e));
    return 0;
}

// The rANS pass alone on caller-supplied (start | freq << 16) pairs of ONE stream (n_pairs even, oldest first): what
// ANSEncoder::put_start_freq + flush_chunk produce (ans.rs:287-378), chunk after chunk.
extern "C" int divans_gpu_selftest_rans_pairs(divans_gpu_codec* c, const uint32_t* pairs, uint32_t n_pairs, uint8_t* out, size_t cap, size_t* out_len) {
    if (!c || !pairs || !out || !out_len || (n_pairs & 1u)) return fail(DIVANS_GPU_EINVAL, "bad argument (the pair count must be even)");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t len = n_pairs / 2u;
    const uint64_t slot = divans_gpu_lit_encode_bound(len);
    uint32_t* d_sf = nullptr; uint8_t* d_out = nullptr; uint64_t* d_off = nullptr; uint32_t* d_sz = nullptr;
    hipError_t e = hipMalloc(&d_sf, (size_t)n_pairs * 4u + 64u);
    if (e == hipSuccess) e = hipMalloc(&d_out, slot + 64u);
    if (e == hipSuccess) e = hipMalloc(&d_off, 8);
    if (e == hipSuccess) e = hipMalloc(&d_sz, 4);
    uint64_t off = 0; uint32_t sz = 0; uint32_t status = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(d_sf, pairs, (size_t)n_pairs * 4u, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_status, 0, 4, c->stream);
    if (e == hipSuccess) {
        RansBatch r;
        std::memset(&r, 0, sizeof(r));
        r.sf = d_sf; r.n_streams = 1; r.stream_len = len; r.max_stream_len = len; r.sf_stride = 2u * len; r.in_sizes = nullptr;
        r.out = d_out; r.out_slot = slot; r.out_offsets = d_off; r.out_sizes = d_sz; r.status = c->d_status;
        e = launch_rans_encode(r, c->stream);   // scratch == null: the one-lane-per-stream kernel, any number of chunks
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&off, d_off, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&sz, d_sz, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&status, c->d_status, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && sz <= cap) e = hipMemcpy(out, d_out + off, sz, hipMemcpyDeviceToHost);
    (void)hipFree(d_sf); (void)hipFree(d_out); (void)hipFree(d_off); (void)hipFree(d_sz);
    if (e != hipSuccess) return fail(DIVANS_GPU_EHIP, hipGetErrorString(e));
    if (status & LIT_STATUS_BAD_MODEL) { (void)hipMemsetAsync(c->d_status, 0, 4, c->stream); return fail(DIVANS_GPU_EINVAL, "invalid (start,freq) pair"); }
    if (sz > cap) return fail(DIVANS_GPU_ECAP, "output buffer too small");
    *out_len = sz;
    return 0;
}

// ---- host-memory convenience wrappers ----------------------------------------------------------
extern "C" int divans_gpu_lit_encode_host(divans_gpu_codec* c, const uint8_t* in, uint32_t stream_len, uint32_t n_streams,
                                          uint8_t* out_packed, size_t out_cap, uint64_t* out_offsets, uint32_t* out_sizes,
                                          size_t* out_total) {
    return divans_gpu_lit_encode_host_chunks(c, in, stream_len, n_streams, out_packed, out_cap, out_offsets, out_sizes,
                                             out_total, nullptr, 0);
}

// The host-buffer entry points keep their device buffers between calls (allocating and freeing gigabytes per call
// costs more than the kernels); `which` names the buffer, capacity only grows.
template <typename T>
static int host_scratch(divans_gpu_codec* c, int which, size_t bytes, T** out) {
    if (bytes > c->host_scratch_cap[which]) {
        if (c->host_scratch[which]) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->host_scratch[which])); c->host_scratch[which] = nullptr; c->host_scratch_cap[which] = 0; }
        if (device_alloc((void**)&c->host_scratch[which], bytes) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(host-buffer staging) failed");
        c->host_scratch_cap[which] = bytes;
    }
    *out = (T*)c->host_scratch[which];
    return 0;
}

// ---- one stream, piece by piece -------------------------------------------------------------------------------------------
// The per-stream ABI's encoder (include/divans_ffi.h) emits container bytes while input is still arriving, as the reference does
// (src/divans_compressor.rs:276-337): each Literal command of a ring lap is coded when it is emitted.  The model pass of a piece
// continues the previous one -- the stream's CDF tables stay in the codec's table slab, the Weights in d_wstate, the history
// comes with the call (last8) -- its pairs join the ones still waiting for their 65 536-symbol chunk to complete, and every
// complete chunk goes through the rANS pass (one lane per chunk) and back to the host.  Memory: the piece, 8 bytes per byte
// of it, one open chunk.
static int stream_rans(divans_gpu_codec* c, uint32_t n_chunks, uint32_t syms_last, uint8_t* out, size_t out_cap, uint32_t* sizes, size_t* out_len) {
    // chunks 0 .. n_chunks-1 of d_sp, 65 536 pairs each except the last (syms_last), as independent one-chunk streams
    const uint64_t slot = divans_gpu_lit_encode_bound(32768);
    uint8_t* d_out = nullptr; uint64_t* d_off = nullptr; uint32_t* d_sz = nullptr; uint32_t* d_len = nullptr;
    int rc;
    if ((rc = host_scratch(c, 4, (size_t)n_chunks * slot + 64, &d_out))) return rc;
    if ((rc = host_scratch(c, 5, (size_t)n_chunks * 8u + 64, &d_off))) return rc;
    if ((rc = host_scratch(c, 6, (size_t)n_chunks * 8u + 64, &d_sz))) return rc;
    d_len = d_sz + n_chunks;
    std::vector<uint32_t> lens(n_chunks, 32768u); lens[n_chunks - 1] = syms_last / 2u;
    HIP_TRY(hipMemcpyAsync(d_len, lens.data(), (size_t)n_chunks * 4u, hipMemcpyHostToDevice, c->stream));
    RansBatch r;
    std::memset(&r, 0, sizeof(r));
    r.sf = c->d_sp; r.n_streams = n_chunks; r.stream_len = 32768u; r.max_stream_len = 32768u; r.sf_stride = 65536u; r.in_sizes = d_len;
    r.out = d_out; r.out_slot = slot; r.out_offsets = d_off; r.out_sizes = d_sz; r.status = c->d_status;
    HIP_TRY(launch_rans_encode(r, c->stream));
    std::vector<uint64_t> offs(n_chunks); std::vector<uint32_t> szs(n_chunks);
    HIP_TRY(hipMemcpyAsync(offs.data(), d_off, (size_t)n_chunks * 8u, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(szs.data(), d_sz, (size_t)n_chunks * 4u, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    size_t total = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) total += szs[k];
    if (total > out_cap) return fail(DIVANS_GPU_ECAP, "output buffer too small");
    size_t pos = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) {
        HIP_TRY(hipMemcpyAsync(out + pos, d_out + offs[k], szs[k], hipMemcpyDeviceToHost, c->stream));
        pos += szs[k]; if (sizes) sizes[k] = szs[k];
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out_len = total;
    return 0;
}

extern "C" int divans_gpu_lit_stream_begin(divans_gpu_codec* c) {
    if (!c) return fail(DIVANS_GPU_EINVAL, "null codec");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_wstate && hipMalloc(&c->d_wstate, 64) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(stream state) failed");
    c->sp_pending = 0; c->sp_started = false;
    // the tables of the stream live in the first slab of the table area from call to call: one workgroup, no row cache (a cached
    // row would have to be written back at the end of every piece)
    c->blocks = 1; c->cache_high = c->cache_low = 0; c->cache_unified = false; c->user_geometry = true;
    return 0;
}

extern "C" int divans_gpu_lit_stream_encode(divans_gpu_codec* c, const uint8_t* in, uint32_t len, uint64_t last8, uint8_t* out, size_t out_cap,
                                            uint32_t* chunk_sizes, uint32_t max_chunks, uint32_t* n_chunks, size_t* out_len) {
    if (!c || !in || !out || !n_chunks || !out_len || !c->d_wstate) return fail(DIVANS_GPU_EINVAL, "bad argument (divans_gpu_lit_stream_begin first)");
    if (len == 0 || len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "a piece is 1 .. max_stream_len bytes");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_tables(c); if (rc) return rc;
    const size_t need = (size_t)65536u + 2u * (size_t)c->max_stream_len + 64u;
    if (need > c->sp_cap) {
        uint32_t* p = nullptr;
        if (hipMalloc(&p, need * 4u) != hipSuccess) return fail(DIVANS_GPU_ENOMEM, "hipMalloc(stream pairs) failed");
        if (c->d_sp) { HIP_TRY(hipMemcpy(p, c->d_sp, (size_t)c->sp_pending * 4u, hipMemcpyDeviceToDevice)); (void)hipFree(c->d_sp); }
        c->d_sp = p; c->sp_cap = need;
    }
    uint8_t* d_in = nullptr; uint32_t* d_seg = nullptr;
    if ((rc = host_scratch(c, 0, (size_t)len + 64, &d_in))) return rc;
    if ((rc = host_scratch(c, 7, 64, &d_seg))) return rc;
    const uint32_t seg_host[8] = {0u, 1u, 0u, 0u, len, (uint32_t)c->cfg.btype, (uint32_t)last8, (uint32_t)(last8 >> 32)};   // seg_begin[2], pad, divans_lit_segment
    HIP_TRY(hipMemcpyAsync(d_in, in, len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_seg, seg_host, sizeof(seg_host), hipMemcpyHostToDevice, c->stream));
    LitBatch b;
    std::memset(&b, 0, sizeof(b));
    b.blob = c->d_blob; b.geom = c->geom; b.tables = c->d_tables;
    b.n_streams = 1; b.stream_len = len; b.max_stream_len = c->max_stream_len;
    b.in = d_in; b.sf = c->d_sp + c->sp_pending; b.status = c->d_status;
    b.seg_begin = d_seg; b.segs = (const LitSegment*)(d_seg + 4);
    b.resume = c->sp_started ? 1u : 0u; b.wstate = c->d_wstate;
    set_cache_fields(c, b);
    ++c->table_launch_seq;
    HIP_TRY(launch_model_encode(b, c->mix, 1u, c->stream));
    c->sp_started = true;
    c->sp_pending += 2u * len;
    const uint32_t full = c->sp_pending / 65536u;
    *n_chunks = full; *out_len = 0;
    if (full == 0) { HIP_TRY(hipStreamSynchronize(c->stream)); return 0; }
    if (full > max_chunks || !chunk_sizes) return fail(DIVANS_GPU_ECAP, "chunk_sizes too small");
    rc = stream_rans(c, full, 65536u, out, out_cap, chunk_sizes, out_len); if (rc) return rc;
    const uint32_t rest = c->sp_pending - full * 65536u;     // < 65536 pairs, behind at least one coded chunk: the ranges do not overlap
    if (rest) HIP_TRY(hipMemcpyAsync(c->d_sp, c->d_sp + (size_t)full * 65536u, (size_t)rest * 4u, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sp_pending = rest;
    return 0;
}

// The decode direction: whole 65 536-symbol chunks per call (a chunk starts from fresh rANS states, so besides the tables, the
// Weights and the history nothing carries over).  `coded` = the LIT-coder bytes from the current chunk boundary on, as many as have
// arrived (a multiple of 4 is used); `out_len` = 32768 * k bytes of output, or less for the last chunk of the stream; the bytes the
// chunks occupied come back in *consumed_bytes.  The caller makes sure the chunks are complete: divans_gpu_lit_encode_bound(32768)
// bytes per chunk always are enough.
extern "C" int divans_gpu_lit_stream_decode_begin(divans_gpu_codec* c) {
    const int rc = divans_gpu_lit_stream_begin(c);
    if (rc) return rc;
    c->sd_started = false;
    return 0;
}

extern "C" int divans_gpu_lit_stream_decode(divans_gpu_codec* c, const uint8_t* coded, size_t coded_bytes, uint32_t out_len, uint64_t last8,
                                            uint8_t* out, size_t* consumed_bytes) {
    if (!c || !coded || !out || !consumed_bytes || !c->d_wstate) return fail(DIVANS_GPU_EINVAL, "bad argument (divans_gpu_lit_stream_decode_begin first)");
    if (out_len == 0 || out_len > c->max_stream_len) return fail(DIVANS_GPU_EINVAL, "a call decodes 1 .. max_stream_len bytes");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_tables(c); if (rc) return rc;
    const size_t use = std::min<size_t>(coded_bytes & ~(size_t)3, (size_t)0xfffffff0u);
    uint8_t* d_in = nullptr; uint8_t* d_out = nullptr; uint32_t* d_meta = nullptr;
    if ((rc = host_scratch(c, 0, use + 64, &d_in))) return rc;
    if ((rc = host_scratch(c, 1, (size_t)out_len + 64, &d_out))) return rc;
    if ((rc = host_scratch(c, 7, 128, &d_meta))) return rc;
    // d_meta: [0..1] in_offset (u64 0), [2] in_size, [3] consumed, [4..5] seg_begin, [8..11] divans_lit_segment
    const uint32_t meta[12] = {0u, 0u, (uint32_t)use, 0u, 0u, 1u, 0u, 0u, out_len, (uint32_t)c->cfg.btype, (uint32_t)last8, (uint32_t)(last8 >> 32)};
    HIP_TRY(hipMemcpyAsync(d_in, coded, use, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, 4, c->stream));
    LitBatch b;
    std::memset(&b, 0, sizeof(b));
    b.blob = c->d_blob; b.geom = c->geom; b.tables = c->d_tables;
    b.n_streams = 1; b.stream_len = out_len; b.max_stream_len = c->max_stream_len;
    b.in = d_in; b.in_offsets = (const uint64_t*)d_meta; b.in_sizes = d_meta + 2; b.consumed = d_meta + 3;
    b.out = d_out; b.status = c->d_status;
    b.seg_begin = d_meta + 4; b.segs = (const LitSegment*)(d_meta + 8);
    b.resume = c->sd_started ? 1u : 0u; b.wstate = c->d_wstate;
    set_cache_fields(c, b);
    ++c->table_launch_seq;
    HIP_TRY(launch_decode(b, c->mix, 1u, c->stream));
    c->sd_started = true;
    uint32_t words = 0, status = 0;
    HIP_TRY(hipMemcpyAsync(&words, d_meta + 3, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&status, c->d_status, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, out_len, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (status & LIT_STATUS_BAD_STREAM) { (void)hipMemsetAsync(c->d_status, 0, 4, c->stream); return fail(DIVANS_GPU_ECORRUPT, "the literal stream fails its integrity check"); }
    *consumed_bytes = (size_t)words * 4u;
    return 0;
}

// the open chunk, as ANSEncoder::close flushes it (ans.rs:331-378); 0 bytes when the stream ended on a chunk boundary
extern "C" int divans_gpu_lit_stream_finish(divans_gpu_codec* c, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!c || !out || !out_len) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    *out_len = 0;
    if (c->sp_pending == 0) return 0;
    int rc = stream_rans(c, 1u, c->sp_pending, out, out_cap, nullptr, out_len); if (rc) return rc;
    c->sp_pending = 0;
    uint32_t status = 0;
    HIP_TRY(hipMemcpy(&status, c->d_status, 4, hipMemcpyDeviceToHost));
    if (status & LIT_STATUS_BAD_MODEL) { (void)hipMemsetAsync(c->d_status, 0, 4, c->stream); return fail(DIVANS_GPU_EINVAL, "invalid (start,freq) pair"); }
    return 0;
}


extern "C" int divans_gpu_lit_encode_host_chunks(divans_gpu_codec* c, const uint8_t* in, uint32_t stream_len, uint32_t n_streams,
                                                 uint8_t* out_packed, size_t out_cap, uint64_t* out_offsets, uint32_t* out_sizes,
                                                 size_t* out_total, uint32_t* out_chunk_bytes, uint32_t max_chunks) {
    if (!c || !in || !out_packed || !out_offsets || !out_sizes || !out_total) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (out_chunk_bytes && max_chunks < (2u * stream_len + 65535u) / 65536u) return fail(DIVANS_GPU_EINVAL, "max_chunks too small");
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)stream_len * n_streams;
    const uint64_t slot = divans_gpu_lit_encode_bound(stream_len);
    uint8_t *d_in = nullptr, *d_slots = nullptr, *d_packed = nullptr;
    uint64_t *d_off = nullptr, *d_poff = nullptr, *d_total = nullptr; uint32_t* d_sz = nullptr; uint32_t* d_chunks = nullptr;
    int rc = 0;
    auto cleanup = [&]() {};
#define TRY_OR_CLEAN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(DIVANS_GPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    if ((rc = host_scratch(c, 0, in_bytes + 64, &d_in))) return rc;
    if ((rc = host_scratch(c, 1, slot * n_streams + 64, &d_slots))) return rc;
    if ((rc = host_scratch(c, 2, slot * n_streams + 64, &d_packed))) return rc;
    if ((rc = host_scratch(c, 3, sizeof(uint64_t) * n_streams, &d_off))) return rc;
    if ((rc = host_scratch(c, 4, sizeof(uint64_t) * n_streams, &d_poff))) return rc;
    if ((rc = host_scratch(c, 5, sizeof(uint64_t), &d_total))) return rc;
    if ((rc = host_scratch(c, 6, sizeof(uint32_t) * n_streams, &d_sz))) return rc;
    if (out_chunk_bytes) {
        if ((rc = host_scratch(c, 7, sizeof(uint32_t) * (size_t)n_streams * max_chunks, &d_chunks))) return rc;
        TRY_OR_CLEAN(hipMemsetAsync(d_chunks, 0, sizeof(uint32_t) * (size_t)n_streams * max_chunks, c->stream));
    }
    TRY_OR_CLEAN(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));   // this call reports only its own status
    TRY_OR_CLEAN(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    rc = encode_batch_impl(c, d_in, nullptr, nullptr, stream_len, n_streams, d_slots, slot, d_off, d_sz, d_chunks, max_chunks);
    if (rc) { cleanup(); return rc; }
    rc = divans_gpu_pack_streams(c, d_slots, d_off, d_sz, n_streams, d_packed, d_poff, d_total);
    if (rc) { cleanup(); return rc; }
    uint64_t total = 0; uint32_t status = 0;
    TRY_OR_CLEAN(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(&status, c->d_status, sizeof(status), hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(out_offsets, d_poff, sizeof(uint64_t) * n_streams, hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(out_sizes, d_sz, sizeof(uint32_t) * n_streams, hipMemcpyDeviceToHost, c->stream));
    if (out_chunk_bytes)
        TRY_OR_CLEAN(hipMemcpyAsync(out_chunk_bytes, d_chunks, sizeof(uint32_t) * (size_t)n_streams * max_chunks, hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipStreamSynchronize(c->stream));
    if (status) { (void)hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream); cleanup(); return fail(DIVANS_GPU_EINVAL, "model produced an invalid (start,freq): unsupported speed/CDF state"); }
    if (total > out_cap) { cleanup(); return fail(DIVANS_GPU_ECAP, "out_cap too small for the packed streams"); }
    TRY_OR_CLEAN(hipMemcpy(out_packed, d_packed, total, hipMemcpyDeviceToHost));
    *out_total = total;
    cleanup();
    return 0;
}

extern "C" int divans_gpu_lit_decode_host(divans_gpu_codec* c, const uint8_t* in_packed, const uint64_t* in_offsets,
                                          const uint32_t* in_sizes, uint32_t n_streams, uint8_t* out, uint32_t stream_len) {
    if (!c || !in_packed || !in_offsets || !in_sizes || !out) return fail(DIVANS_GPU_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    size_t total = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (in_offsets[i] % 4) return fail(DIVANS_GPU_EINVAL, "coded streams must start 4-byte aligned");
        total = std::max<size_t>(total, in_offsets[i] + in_sizes[i]);
    }
    uint8_t *d_in = nullptr, *d_out = nullptr; uint64_t* d_off = nullptr; uint32_t* d_sz = nullptr;
    auto cleanup = [&]() {};
    int rc0 = 0;
    if ((rc0 = host_scratch(c, 0, total + 128, &d_in))) return rc0;
    if ((rc0 = host_scratch(c, 1, (size_t)stream_len * n_streams + 64, &d_out))) return rc0;
    if ((rc0 = host_scratch(c, 3, sizeof(uint64_t) * n_streams, &d_off))) return rc0;
    if ((rc0 = host_scratch(c, 6, sizeof(uint32_t) * n_streams, &d_sz))) return rc0;
    TRY_OR_CLEAN(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));   // this call reports only its own status
    TRY_OR_CLEAN(hipMemcpyAsync(d_in, in_packed, total, hipMemcpyHostToDevice, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(d_off, in_offsets, sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(d_sz, in_sizes, sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice, c->stream));
    int rc = divans_gpu_lit_decode_batch(c, d_in, d_off, d_sz, n_streams, d_out, nullptr, nullptr, stream_len);
    if (rc) { cleanup(); return rc; }
    uint32_t status = 0;
    TRY_OR_CLEAN(hipMemcpyAsync(&status, c->d_status, sizeof(status), hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(out, d_out, (size_t)stream_len * n_streams, hipMemcpyDeviceToHost, c->stream));
    TRY_OR_CLEAN(hipStreamSynchronize(c->stream));
    cleanup();
    if (status & LIT_STATUS_BAD_STREAM) {
        (void)hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream);
        return fail(DIVANS_GPU_ECORRUPT, "a coded stream is truncated, corrupt or was coded under another configuration (final rANS states / word count mismatch)");
    }
    return 0;
}

// ---- pipelined host-buffer entry points ------------------------------------------------------------
// The batch is cut into slices of `slice_streams` streams; slice i+1 travels to the device and slice i-1 back to the host
// while slice i is coded.  Copies only overlap when the caller's buffers are page-locked (divans_gpu_host_alloc,
// hipHostMalloc, hipHostRegister); with pageable memory the calls are correct but the runtime serialises the copies.
extern "C" void* divans_gpu_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)fail(DIVANS_GPU_ENOMEM, "hipHostMalloc failed"); return nullptr; }
    return p;
}
extern "C" void divans_gpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

static int ensure_pipeline(divans_gpu_codec* c, uint32_t slices) {
    if (!c->s_in) HIP_TRY(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    if (!c->s_out) HIP_TRY(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    while (c->ev_in.size() < slices) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_in.push_back(e); }
    while (c->ev_done.size() < slices) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_done.push_back(e); }
    if (c->h_totals_cap < 2u * slices) {
        if (c->h_totals) HIP_TRY(hipHostFree(c->h_totals));
        c->h_totals = nullptr; c->h_totals_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&c->h_totals, sizeof(uint64_t) * 2u * slices, hipHostMallocDefault));
        c->h_totals_cap = 2u * slices;
    }
    return 0;
}

static uint32_t default_slice(const divans_gpu_codec* c, uint32_t n_streams, uint32_t slice_streams) {
    // a slice should keep the persistent decode grid full; four slices are enough to hide the copies
    if (slice_streams == 0) slice_streams = std::max<uint32_t>(resident_groups(c), (n_streams + 3u) / 4u);
    return std::min(slice_streams, n_streams);
}

extern "C" int divans_gpu_lit_encode_host_pipelined(divans_gpu_codec* c, const uint8_t* in, uint32_t stream_len, uint32_t n_streams,
                                                    uint8_t* out_packed, size_t out_cap, uint64_t* out_offsets, uint32_t* out_sizes,
                                                    size_t* out_total, uint32_t slice_streams) {
    if (!c || !in || !out_packed || !out_offsets || !out_sizes || !out_total) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) { *out_total = 0; return 0; }
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t S = default_slice(c, n_streams, slice_streams);
    const uint32_t slices = (n_streams + S - 1u) / S;
    const uint64_t slot = divans_gpu_lit_encode_bound(stream_len);
    uint8_t *d_in = nullptr, *d_slots = nullptr, *d_packed = nullptr;
    uint64_t *d_off = nullptr, *d_poff = nullptr, *d_total = nullptr; uint32_t* d_sz = nullptr;
    int rc = 0;
    if ((rc = ensure_pipeline(c, slices))) return rc;
    if ((rc = host_scratch(c, 0, (size_t)stream_len * n_streams + 64, &d_in))) return rc;
    if ((rc = host_scratch(c, 1, slot * S + 64, &d_slots))) return rc;                       // one slice of right-aligned slots
    if ((rc = host_scratch(c, 2, slot * n_streams + 64, &d_packed))) return rc;              // every slice's packed bytes, slice i at i * S * slot
    if ((rc = host_scratch(c, 3, sizeof(uint64_t) * n_streams, &d_off))) return rc;
    if ((rc = host_scratch(c, 4, sizeof(uint64_t) * n_streams, &d_poff))) return rc;
    if ((rc = host_scratch(c, 5, sizeof(uint64_t) * slices, &d_total))) return rc;
    if ((rc = host_scratch(c, 6, sizeof(uint32_t) * n_streams, &d_sz))) return rc;
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));      // earlier work on the codec's stream may still read the staging buffers
    // From here on copies from `in` and into the caller's output arrays are in flight on three streams: every exit -- also the
    // early ones -- goes through drain() so that the caller can free or reuse its buffers as soon as the call returns.
    auto drain = [&]() { (void)hipStreamSynchronize(c->s_in); (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->s_out); };
    rc = [&]() -> int {
    for (uint32_t i = 0; i < slices; ++i) {
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        HIP_TRY(hipMemcpyAsync(d_in + (size_t)s0 * stream_len, in + (size_t)s0 * stream_len, (size_t)ns * stream_len, hipMemcpyHostToDevice, c->s_in));
        HIP_TRY(hipEventRecord(c->ev_in[i], c->s_in));
    }
    for (uint32_t i = 0; i < slices; ++i) {
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in[i], 0));
        int r = encode_batch_impl(c, d_in + (size_t)s0 * stream_len, nullptr, nullptr, stream_len, ns, d_slots, slot, d_off + s0, d_sz + s0, nullptr, 0);
        if (r) return r;
        r = divans_gpu_pack_streams(c, d_slots, d_off + s0, d_sz + s0, ns, d_packed + (size_t)s0 * slot, d_poff + s0, d_total + i);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync(&c->h_totals[i], d_total + i, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->ev_done[i], c->stream));
    }
    // sizes / offsets of a slice are small; they follow their slice's bytes on the copy-out stream
    uint64_t base = 0;
    std::vector<uint64_t> bases(slices);
    for (uint32_t i = 0; i < slices; ++i) {
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        HIP_TRY(hipEventSynchronize(c->ev_done[i]));
        const uint64_t total = c->h_totals[i];
        if (base + total > out_cap) return fail(DIVANS_GPU_ECAP, "out_cap too small for the packed streams");
        bases[i] = base;
        HIP_TRY(hipMemcpyAsync(out_packed + base, d_packed + (size_t)s0 * slot, total, hipMemcpyDeviceToHost, c->s_out));
        HIP_TRY(hipMemcpyAsync(out_offsets + s0, d_poff + s0, sizeof(uint64_t) * ns, hipMemcpyDeviceToHost, c->s_out));
        HIP_TRY(hipMemcpyAsync(out_sizes + s0, d_sz + s0, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, c->s_out));
        base += total;
    }
    uint32_t status = 0;
    HIP_TRY(hipMemcpyAsync(&status, c->d_status, sizeof(status), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->s_out));
    if (status) { (void)hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream); return fail(DIVANS_GPU_EINVAL, "model produced an invalid (start,freq): unsupported speed/CDF state"); }
    for (uint32_t i = 1; i < slices; ++i) {            // slice-local packed offsets -> offsets in out_packed
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        for (uint32_t k = 0; k < ns; ++k) out_offsets[s0 + k] += bases[i];
    }
    *out_total = base;
    return 0;
    }();
    if (rc) drain();
    return rc;
}

extern "C" int divans_gpu_lit_decode_host_pipelined(divans_gpu_codec* c, const uint8_t* in_packed, const uint64_t* in_offsets,
                                                    const uint32_t* in_sizes, uint32_t n_streams, uint8_t* out, uint32_t stream_len,
                                                    uint32_t slice_streams) {
    if (!c || !in_packed || !in_offsets || !in_sizes || !out) return fail(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t total = 0; bool ordered = true;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (in_offsets[i] % 4) return fail(DIVANS_GPU_EINVAL, "coded streams must start 4-byte aligned");
        if (in_offsets[i] < total) ordered = false;             // a slice's bytes must be one range of the input
        total = std::max<size_t>(total, in_offsets[i] + in_sizes[i]);
    }
    if (!ordered) return divans_gpu_lit_decode_host(c, in_packed, in_offsets, in_sizes, n_streams, out, stream_len);
    const uint32_t S = default_slice(c, n_streams, slice_streams);
    const uint32_t slices = (n_streams + S - 1u) / S;
    uint8_t *d_in = nullptr, *d_out = nullptr; uint64_t* d_off = nullptr; uint32_t* d_sz = nullptr;
    int rc = 0;
    if ((rc = ensure_pipeline(c, slices))) return rc;
    if ((rc = host_scratch(c, 0, total + 128, &d_in))) return rc;
    if ((rc = host_scratch(c, 1, (size_t)stream_len * n_streams + 64, &d_out))) return rc;
    if ((rc = host_scratch(c, 3, sizeof(uint64_t) * n_streams, &d_off))) return rc;
    if ((rc = host_scratch(c, 6, sizeof(uint32_t) * n_streams, &d_sz))) return rc;
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    auto drain = [&]() { (void)hipStreamSynchronize(c->s_in); (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->s_out); };
    rc = [&]() -> int {     // see divans_gpu_lit_encode_host_pipelined: no return with a copy still in flight
    HIP_TRY(hipMemcpyAsync(d_off, in_offsets, sizeof(uint64_t) * n_streams, hipMemcpyHostToDevice, c->s_in));
    HIP_TRY(hipMemcpyAsync(d_sz, in_sizes, sizeof(uint32_t) * n_streams, hipMemcpyHostToDevice, c->s_in));
    for (uint32_t i = 0; i < slices; ++i) {
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        const size_t lo = in_offsets[s0], hi = in_offsets[s0 + ns - 1u] + in_sizes[s0 + ns - 1u];
        if (hi > lo) HIP_TRY(hipMemcpyAsync(d_in + lo, in_packed + lo, hi - lo, hipMemcpyHostToDevice, c->s_in));
        HIP_TRY(hipEventRecord(c->ev_in[i], c->s_in));
    }
    for (uint32_t i = 0; i < slices; ++i) {
        const uint32_t s0 = i * S, ns = std::min(S, n_streams - s0);
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in[i], 0));
        const int r = divans_gpu_lit_decode_batch(c, d_in, d_off + s0, d_sz + s0, ns, d_out + (size_t)s0 * stream_len, nullptr, nullptr, stream_len);
        if (r) return r;
        HIP_TRY(hipEventRecord(c->ev_done[i], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->s_out, c->ev_done[i], 0));
        HIP_TRY(hipMemcpyAsync(out + (size_t)s0 * stream_len, d_out + (size_t)s0 * stream_len, (size_t)ns * stream_len, hipMemcpyDeviceToHost, c->s_out));
    }
    uint32_t status = 0;
    HIP_TRY(hipMemcpyAsync(&status, c->d_status, sizeof(status), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->s_out));
    if (status & LIT_STATUS_BAD_STREAM) {
        (void)hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream);
        return fail(DIVANS_GPU_ECORRUPT, "a coded stream is truncated, corrupt or was coded under another configuration (final rANS states / word count mismatch)");
    }
    return 0;
    }();
    if (rc) drain();
    return rc;
}

