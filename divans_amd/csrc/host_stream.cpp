// host_stream.cpp -- CMD-coder side, Mux, header and trailer of a literal-only .divans stream (host, C++).
// Product code: it links the HIP literal coder (capi.cpp) for every literal byte and never the CPU oracle.
//   command type nibble / flush / trailer : src/codec/mod.rs:143-158,409-560,652-792
//   PredictionMode                         : src/codec/context_map.rs:105-428
//   BlockSwitchLiteral                     : src/codec/block_type.rs:31-194
//   literal length                         : src/codec/literal.rs:565-661
//   Mux                                    : src/mux.rs
//   header / internal compressor           : src/divans_compressor.rs:126-174,276-426, src/raw_to_cmd/mod.rs:105-181
#include "host_stream.h"

#include <algorithm>
#include <array>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace divans_host {

// ---------------------------------------------------------------- CRC-32C (Castagnoli, reflected), src/codec/crc32.rs
// The x86 crc32 instruction computes exactly this polynomial; the table walk is the portable path and the checker of the other.
static uint32_t crc32c_table(uint32_t crc, const uint8_t* p, size_t n) {
    static const struct Table {
        uint32_t t[256];
        Table() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
                t[i] = c;
            }
        }
    } table;
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table.t[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}
#if defined(__x86_64__)
// The crc32 instruction has a latency of three cycles and a throughput of one per cycle: one dependent chain runs at a third of what the core
// can do.  Three chains over three adjacent blocks, each then advanced over the blocks behind it (CRC(A || B) = shift(CRC(A), |B|) ^ CRC(B);
// shifting over a FIXED number of zero bytes is a linear map of the 32 register bits, tabulated byte by byte).  A 30 KB container: 3.8 -> 1.4 us
// -- it was two thirds of parse_container_host and half of assemble_container (round 6).
struct CrcShift {
    uint32_t t[4][256];
    static uint32_t times(const uint32_t* mat, uint32_t vec) { uint32_t sum = 0; for (; vec; vec >>= 1, ++mat) if (vec & 1u) sum ^= *mat; return sum; }
    static void square(uint32_t* sq, const uint32_t* mat) { for (int n = 0; n < 32; ++n) sq[n] = times(mat, mat[n]); }
    explicit CrcShift(size_t len) {      // the operator that advances a (raw, un-inverted) register over `len` zero bytes; len a power of two >= 4
        uint32_t even[32], odd[32];
        odd[0] = 0x82F63B78u;            // one zero BIT: reflected polynomial in row 0, then the shift
        uint32_t row = 1;
        for (int n = 1; n < 32; ++n) { odd[n] = row; row <<= 1; }
        square(even, odd);               // two bits
        square(odd, even);               // four bits
        // from four bits: every squaring doubles; len bytes = 8 len bits
        uint32_t* cur = odd; uint32_t* nxt = even;
        for (size_t bits = 4; bits < 8 * len; bits <<= 1) { square(nxt, cur); std::swap(cur, nxt); }
        for (uint32_t n = 0; n < 256; ++n) { t[0][n] = times(cur, n); t[1][n] = times(cur, n << 8); t[2][n] = times(cur, n << 16); t[3][n] = times(cur, n << 24); }
    }
    uint32_t operator()(uint32_t c) const { return t[0][c & 0xffu] ^ t[1][(c >> 8) & 0xffu] ^ t[2][(c >> 16) & 0xffu] ^ t[3][c >> 24]; }
};
// three blocks of `block` bytes at a time while they last; returns the register, advances p / n
__attribute__((target("sse4.2"))) static uint64_t crc32c_hw_three(uint64_t c, const uint8_t*& p, size_t& n, size_t block, const CrcShift& shift) {
    while (n >= 3 * block) {
        uint64_t c1 = 0, c2 = 0;
        for (size_t i = 0; i < block; i += 8) {
            uint64_t v0, v1, v2;
            std::memcpy(&v0, p + i, 8); std::memcpy(&v1, p + block + i, 8); std::memcpy(&v2, p + 2 * block + i, 8);
            c = __builtin_ia32_crc32di(c, v0); c1 = __builtin_ia32_crc32di(c1, v1); c2 = __builtin_ia32_crc32di(c2, v2);
        }
        c = shift((uint32_t)c) ^ (uint32_t)c1;
        c = shift((uint32_t)c) ^ (uint32_t)c2;
        p += 3 * block; n -= 3 * block;
    }
    return c;
}
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(uint32_t crc, const uint8_t* p, size_t n) {
    constexpr size_t kLong = 4096, kShort = 256;
    static const CrcShift shift_long(kLong), shift_short(kShort);
    uint64_t c = (uint32_t)~crc;
    while (n && ((uintptr_t)p & 7u)) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); --n; }
    c = crc32c_hw_three(c, p, n, kLong, shift_long);
    c = crc32c_hw_three(c, p, n, kShort, shift_short);
    for (; n >= 8; n -= 8, p += 8) { uint64_t v; std::memcpy(&v, p, 8); c = __builtin_ia32_crc32di(c, v); }
    while (n) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); --n; }
    return ~(uint32_t)c;
}
#endif
uint32_t crc32c(uint32_t crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(crc, p, n);
#endif
    return crc32c_table(crc, p, n);
}
uint32_t crc32c_portable(uint32_t crc, const uint8_t* p, size_t n) { return crc32c_table(crc, p, n); }

// ---------------------------------------------------------------- 16-symbol CDF on the host (CMD coder only)
struct Speed { int16_t inc, lim; };
static const Speed kMud{0x10, 0x2000}, kSlow{0x20, 0x1000}, kMed{0x30, 0x4000}, kFast{0x60, 0x4000},
    kPlane{0x80, 0x4000}, kRocket{0x180, 0x4000};   // probability/interface.rs:321-328

static inline int16_t wrap16(int v) { return (int16_t)(uint16_t)v; }

#if defined(__SSE2__)
// lanes >= sym of a 16 x i16 row: sixteen zero words, then sixteen all-ones words, read 16 - sym words in
alignas(16) static const int16_t kFromSym[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
alignas(16) static const int16_t kBias[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
#endif

struct alignas(16) Cdf {
    int16_t c[16];
    Cdf() { for (int i = 0; i < 16; ++i) c[i] = (int16_t)(4 * (i + 1)); }        // frequentist_cdf.rs:17-23
    void blend(int sym, Speed s) {                                               // frequentist_cdf.rs:74-85
#if defined(__SSE2__)
        // the same wrapping i16 arithmetic, eight entries at a time (the SIMD twin of the reference does the same, simd_frequentist_cdf.rs:213-224)
        const __m128i inc = _mm_set1_epi16(s.inc);
        __m128i lo = _mm_load_si128((const __m128i*)c), hi = _mm_load_si128((const __m128i*)(c + 8));
        lo = _mm_add_epi16(lo, _mm_and_si128(inc, _mm_loadu_si128((const __m128i*)(kFromSym + 16 - sym))));
        hi = _mm_add_epi16(hi, _mm_and_si128(inc, _mm_loadu_si128((const __m128i*)(kFromSym + 24 - sym))));
        if ((int16_t)_mm_extract_epi16(hi, 7) >= s.lim) {
            lo = _mm_add_epi16(lo, _mm_load_si128((const __m128i*)kBias)); hi = _mm_add_epi16(hi, _mm_load_si128((const __m128i*)(kBias + 8)));
            lo = _mm_sub_epi16(lo, _mm_srai_epi16(lo, 2)); hi = _mm_sub_epi16(hi, _mm_srai_epi16(hi, 2));
        }
        _mm_store_si128((__m128i*)c, lo); _mm_store_si128((__m128i*)(c + 8), hi);
#else
        for (int i = sym; i < 16; ++i) c[i] = wrap16(c[i] + s.inc);
        if (c[15] >= s.lim)
            for (int i = 0; i < 16; ++i) { int16_t t = wrap16(c[i] + i + 1); c[i] = wrap16(t - (t >> 2)); }
#endif
    }
    bool range(int sym, int& start, int& freq) const {                           // probability/interface.rs:97-108
        const int mx = c[15];
        if (mx == 0) return false;
        const int hi = ((int)c[sym] << 15) / mx;
        const int lo = sym ? ((int)c[sym - 1] << 15) / mx : 0;
        start = wrap16(wrap16(lo) + 1);
        freq = wrap16(wrap16(hi - lo) - 1);
        return true;
    }
    int find(int offset) const {                                                 // probability/interface.rs:136-198
        const int16_t r = wrap16(((int)(int16_t)offset * (int)c[15]) >> 15);
#if defined(__SSE2__)
        const __m128i rr = _mm_set1_epi16(r);
        const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpgt_epi16(_mm_load_si128((const __m128i*)c), rr))
                         | (unsigned)_mm_movemask_epi8(_mm_cmpgt_epi16(_mm_load_si128((const __m128i*)(c + 8)), rr)) << 16;
        return m ? __builtin_ctz(m) >> 1 : 15;     // the first entry above r; entry 15 and "none" both mean symbol 15
#else
        for (int i = 0; i < 15; ++i) if (r < c[i]) return i;
        return 15;
#endif
    }
};

// ---------------------------------------------------------------- rANS on the host (CMD coder only), src/ans.rs
class RansEncoder {
  public:
    std::vector<uint8_t> out;            // finished chunks in decoder order
    bool failed = false;
    void put(int start, int freq) {      // ans.rs:287-301
        pairs_.push_back(((uint32_t)(uint16_t)freq << 16) | (uint16_t)start);
        if (pairs_.size() == 65536) flush();
    }
    void flush() {                       // ans.rs:331-378 + 302-329
        if (pairs_.empty()) return;
        std::vector<uint32_t> words;
        uint64_t a = 1ull << 31, b = 1ull << 31;
        for (size_t k = pairs_.size(); k-- > 0;) {
            const int16_t start = (int16_t)(pairs_[k] & 0xffffu), freq = (int16_t)(pairs_[k] >> 16);
            if (freq <= 0 || start < 0) { failed = true; pairs_.clear(); return; }
            const uint64_t f = (uint64_t)freq;
            uint64_t st = a;
            if (st >= (f << 48)) { words.push_back((uint32_t)st); st >>= 32; }
            const uint64_t x = ((st / f) << 15) + (st % f) + (uint64_t)start;
            a = b; b = x;
        }
        std::swap(a, b);
        for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(a >> (8 * i)));
        for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(b >> (8 * i)));
        for (size_t w = words.size(); w-- > 0;)
            for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(words[w] >> (8 * i)));
        pairs_.clear();
    }
  private:
    std::vector<uint32_t> pairs_;
};

class RansDecoder {
  public:
    RansDecoder(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    bool starved = false;
    int get(const Cdf& cdf) {            // ans.rs:246-252, 230-244, refill :428-442
        fill();
        const int offset = (int)(a_ & 0x7fffu);
        const int sym = cdf.find(offset);
        int start = 1, freq = 1;
        if (!cdf.range(sym, start, freq)) starved = true;
        need_a_ = need_b_ | ((count_ == 65535u) ? 8u : 0u);
        const uint64_t x = (uint64_t)(int64_t)freq * (a_ >> 15) + (a_ & 0x7fffu) - (uint64_t)(int64_t)start;
        count_ = (uint16_t)(count_ + 1);
        need_b_ = x < (1ull << 31) ? 1u : 0u;
        a_ = b_; b_ = x;
        return sym;
    }
  private:
    void fill() {
        if (need_a_ == 0) return;
        if (need_a_ == 1) {
            if (n_ - pos_ < 4) { starved = true; return; }
            uint32_t w = 0;
            for (int i = 0; i < 4; ++i) w |= (uint32_t)p_[pos_ + i] << (8 * i);
            a_ = (a_ << 32) | w; pos_ += 4; need_a_ = 0;
            return;
        }
        if (n_ - pos_ < 16) { starved = true; return; }
        a_ = b_ = 0; count_ = 0;
        for (int i = 0; i < 8; ++i) a_ |= (uint64_t)p_[pos_ + i] << (8 * i);
        for (int i = 0; i < 8; ++i) b_ |= (uint64_t)p_[pos_ + 8 + i] << (8 * i);
        pos_ += 16; need_a_ = 0;
    }
    const uint8_t* p_; size_t n_, pos_ = 0;
    uint64_t a_ = 0, b_ = 0; uint16_t count_ = 0; unsigned need_a_ = 8, need_b_ = 0;
};

// get_or_put_nibble + blend for either direction
struct NibbleCoder {
    RansEncoder* enc = nullptr; RansDecoder* dec = nullptr;
    std::function<void()> before;   // drain_or_fill_internal_buffer_cmd runs before every CMD nibble (e.g. codec/mod.rs:663)
    uint32_t nibbles = 0;
    int code(int v, Cdf& prior, Speed sp) {
        ++nibbles;
        if (before) before();
        if (enc) { int s, f; if (!prior.range(v, s, f)) enc->failed = true; else enc->put(s, f); }
        else v = dec->get(prior);
        prior.blend(v, sp);
        return v;
    }
};

// 5.3 mini-float of the speeds, probability/interface.rs:566-585
static uint8_t speed_to_u8(int16_t d) {
    const uint16_t u = (uint16_t)d;
    int length = 0; while (length < 16 && (u >> length) != 0) ++length;
    int mant = 0;
    if (d != 0) { const int16_t rem = (int16_t)(d - (int16_t)(1 << (length - 1))); mant = (int16_t)(rem << 3) >> (length - 1); }
    return (uint8_t)((length << 3) | mant);
}
uint8_t speed_to_f8(int16_t v) { return speed_to_u8(v); }
static int16_t u8_to_speed(uint8_t d) {
    if (d < 8) return 0;
    const int lg = ((d >> 3) - 1) & 15;       // exponents past 15 only come from damaged streams; an i16 shift in a release build of the reference takes the amount mod 16
    const int16_t rem = (int16_t)(uint16_t)(((uint32_t)d & 7u) << lg);
    return (int16_t)((int16_t)(uint16_t)(1u << lg) | (rem >> 3));
}

// ---------------------------------------------------------------- command-stream model
class CommandModel {
  public:
    explicit CommandModel(const StreamOptions& o) {
        lit_len.resize(4096 + 3 * 256);
        last_4_states = 3 << 4;                                   // codec/interface.rs:374
        for (auto& l : btype_lru) { l[0] = 0; l[1] = 1; }
        mixing = o.dynamic_context_mixing;
        if (o.force_stride != 0 && mixing == 0 && o.use_context_map) mixing = 1;   // codec/interface.rs:360-365
        prior_depth = o.has_prior_depth ? o.prior_depth : 0;
        do_context_map = o.use_context_map; force_stride = o.force_stride; wire = o.wire;
        has_adaptation = o.has_literal_adaptation;
        for (int i = 0; i < 4; ++i) adaptation[i] = Speed{o.literal_adaptation[i].inc, o.literal_adaptation[i].lim};
        pm_cmap.assign(DIVANS_GPU_MAX_LITERAL_CONTEXT_MAP_SIZE, 0);
        pm_mixing.assign(DIVANS_GPU_NUM_MIXING_VALUES, 0);
        pm_dmap.assign(1024, 0);
    }
    // priors (layouts of codec/priors.rs, linearised as priors.rs:211-259)
    std::array<Cdf, 16> cc;                 // CrossCommandPriors FullSelection(16,1)
    std::vector<Cdf> lit_len;               // CountSmall(256,16) | SizeBegNib | SizeLastNib | SizeMantissaNib
    std::array<Cdf, 31> pred;               // Only, LiteralSpeed, FirstNibble(2), SecondNibble(2), Mnemonic(4), PriorMixingValue(17), ContextMapSpeedPalette(4)
    std::array<Cdf, 10> btype;              // Mnemonic(3) FirstNibble(3) SecondNibble(3) StrideNibble(1)
    std::array<uint8_t, 13> lru{};          // cmap_lru
    std::array<std::array<uint8_t, 2>, 3> btype_lru{};
    std::array<uint8_t, 3> btype_max{};
    uint8_t last_4_states = 0;
    uint8_t mixing = 0, prior_depth = 0, force_stride = 9;
    int wire = 0;                           // DIVANS_WIRE_*: 1 = the build behind wasm/wasm.html's example (two prior rows differ, below)
    bool do_context_map = true, has_adaptation = false;
    Speed adaptation[4];
    std::vector<uint8_t> pm_cmap, pm_mixing, pm_dmap;     // the codec's own PredictionModeContextMap (persists, context_map.rs:84-94)
    // result of the last PredictionMode command
    uint8_t pm_mode = 0, pm_mixing_math = 0;
    Speed pm_speeds[4] = {kMud, kMud, kMud, kMud};

    int command_type(NibbleCoder& nc, int code) {           // codec/mod.rs:662-688
        code = nc.code(code, cc[last_4_states >> 4], kRocket);
        if (code == 3) { last_4_states >>= 2; last_4_states |= 128; }
        return code;
    }

    bool prediction_mode(NibbleCoder& nc, const PredictionModeIn* in) {   // context_map.rs:105-428
        Speed desired[4] = {kMud, kMud, kMud, kMud};
        if (in && in->has_context_speeds) {
            for (int i = 0; i < 2; ++i) {
                if (in->cm_speed[i][0] || in->cm_speed[i][1]) desired[2 + i] = Speed{u8_to_speed(in->cm_speed[i][0]), u8_to_speed(in->cm_speed[i][1])};
                const uint8_t* st = mixing != 0 ? in->combined_speed[i] : in->stride_speed[i];
                if (st[0] || st[1]) desired[i] = Speed{u8_to_speed(st[0]), u8_to_speed(st[1])};
            }
        }
        if (has_adaptation) for (int i = 0; i < 4; ++i) desired[i] = adaptation[i];
        for (int i = 0; i < 13; ++i) lru[i] = (uint8_t)i;
        const int mode = nc.code(in ? in->prediction_mode : 0, pred[0], kMed);
        if (mode > 3) return false;
        pm_mode = (uint8_t)mode;
        const int is_adv = in ? in->is_adv : 0;
        // DynamicContextMixingSpeed and PriorDepth are not members of PredictionModePriors: they alias the last entry (offset 27)
        const int mixnib = nc.code(mixing | (is_adv << 3), pred[27], kMed);
        pm_mixing_math = (uint8_t)(mixnib & 3);
        const bool combine = mixnib != 0;
        nc.code(prior_depth, pred[27], kFast);
        uint8_t f8[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int index = 0; index < 16; ++index) {
            const int si = index >> 2, pt = index & 3;
            const uint8_t f0 = speed_to_u8(desired[si].inc), f1 = speed_to_u8(desired[si].lim);
            int nib = pt == 0 ? (f0 & 0x7f) >> 3 : pt == 1 ? (f0 & 7) : pt == 2 ? (f1 & 0x7f) >> 3 : (f1 & 7);
            nib = nc.code(nib, pred[27 + pt], kFast);
            if (pt == 0) f8[si][0] |= (uint8_t)(nib << 3);
            if (pt == 1) f8[si][0] |= (uint8_t)nib;
            if (pt == 2) f8[si][1] |= (uint8_t)(nib << 3);
            if (pt == 3) f8[si][1] |= (uint8_t)nib;
        }
        for (int i = 0; i < 4; ++i)   // stored as f8 again and read back through u8_to_speed (codec/interface.rs:303-308)
            pm_speeds[i] = Speed{u8_to_speed(speed_to_u8(u8_to_speed(f8[i][0]))), u8_to_speed(speed_to_u8(u8_to_speed(f8[i][1])))};
        for (int type = 0; type < 2; ++type) {
            const std::vector<uint8_t>* cur = nullptr;
            if (in && do_context_map) { if (type == 0) cur = &in->literal_context_map; else if (in->has_context_speeds) cur = &in->distance_context_map; }
            std::vector<uint8_t>& outmap = type == 0 ? pm_cmap : pm_dmap;
            for (uint32_t index = 0;; ++index) {
                int mn = 14;
                const bool have = cur && index < cur->size();
                if (have) {
                    const uint8_t target = (*cur)[index];
                    mn = 15;
                    for (int i = 0; i < 13; ++i) if (lru[i] == target) mn = i;
                    if (target == (uint8_t)(*std::max_element(lru.begin(), lru.end()) + 1)) mn = 13;
                }
                // HEAD: Mnemonic is a member of PredictionModePriors (offset 6 + type).  The example's build coded the mnemonics of both
                // maps under row 27 -- its two mnemonic nibbles read "end of map" there and nowhere else
                mn = nc.code(mn, pred[wire == 1 ? 27 : 6 + type], kMed);
                if (mn == 14) { if (type == 0) for (int i = 0; i < 13; ++i) lru[i] = (uint8_t)i; break; }
                uint8_t val;
                if (mn == 15) {
                    const int msn = nc.code(have ? (*cur)[index] >> 4 : 0, pred[2 + type], kMed);
                    const int lsn = nc.code(have ? (*cur)[index] & 0xf : 0, pred[4 + type], kMed);
                    val = (uint8_t)((msn << 4) | lsn);
                } else {
                    val = mn == 13 ? (uint8_t)(*std::max_element(lru.begin(), lru.end()) + 1) : lru[(size_t)mn];
                }
                if (index >= outmap.size()) return false;
                outmap[index] = val;
                touch_lru(val);
            }
        }
        for (uint32_t index = 0; index < DIVANS_GPU_NUM_MIXING_VALUES; ++index) {
            int nib = !do_context_map ? 4 : (!combine ? 0 : ((in && in->has_context_speeds && !in->mixing_values.empty()) ? in->mixing_values[index] : 0));
            const int prior = (index >= 256 && wire != 1) ? (pm_mixing[index - 256] & 0xf) : 16;   // the example's build: row 16 throughout
            nib = nc.code(nib, pred[10 + prior], kPlane);
            pm_mixing[index] = (uint8_t)nib;
        }
        return true;
    }

    void block_switch_literal(NibbleCoder& nc, uint8_t in_btype, uint8_t in_stride, uint8_t& out_btype) {   // block_type.rs:31-194
        int varint = in_btype == btype_lru[0][1] ? 0 : in_btype == (uint8_t)(btype_max[0] + 1) ? 1 : in_btype <= 12 ? in_btype + 2 : 15;
        varint = nc.code(varint, btype[0], kSlow);
        uint8_t bt;
        if (varint == 0) bt = btype_lru[0][1];
        else if (varint == 1) bt = (uint8_t)(btype_max[0] + 1);
        else if (varint == 15) {
            const int first = nc.code(in_btype & 0xf, btype[3], kSlow);
            const int second = nc.code(in_btype >> 4, btype[6], kSlow);
            bt = (uint8_t)((second << 4) | first);
        } else bt = (uint8_t)(varint - 2);
        nc.code(force_stride == 9 ? in_stride : force_stride, btype[9], kSlow);
        last_4_states >>= 2;                                    // obs_btypel, codec/interface.rs:527-537
        btype_lru[0][1] = btype_lru[0][0]; btype_lru[0][0] = bt;
        btype_max[0] = std::max(btype_max[0], bt);
        out_btype = bt;
    }

    bool literal_length(NibbleCoder& nc, uint32_t len_in, uint32_t& len_out) {   // literal.rs:565-661
        const uint32_t ctype = btype_lru[1][0], MN = 14;
        const uint32_t serialized = len_in - (MN + 1);
        int lllen = 0; while (lllen < 32 && (serialized >> lllen) != 0) ++lllen;
        const int shortcut = nc.code((int)std::min<uint32_t>(MN, len_in - 1), lit_len[ctype], kMed);
        if (shortcut == (int)MN + 1) return false;
        if (shortcut != (int)MN) { len_out = (uint32_t)shortcut + 1; return true; }
        const int beg = nc.code(std::min(15, lllen), lit_len[4096 + ctype], kMud);
        int remaining; uint32_t decoded;
        auto round_up_mod_4 = [](int v) { return ((v - 1) | 3) + 1; };
        if (beg == 15) {
            const int last = nc.code((lllen - 15) & 0xf, lit_len[4096 + 256 + ctype], kMud);
            remaining = round_up_mod_4(last + 14); decoded = 1u << (last + 14);
        } else if (beg <= 1) { len_out = MN + 1 + (uint32_t)beg; return true; }
        else { remaining = round_up_mod_4(beg - 1); decoded = 1u << (beg - 1); }
        for (;;) {
            const int next = remaining - 4;
            const int nib = nc.code((int)(((serialized ^ decoded) >> next) & 0xf), lit_len[4096 + 512 + ctype], kMud);
            decoded |= (uint32_t)nib << next;
            if (next == 0) { len_out = decoded + MN + 1; return true; }
            remaining = next;
        }
    }

    void fill_lit_config(divans_lit_config& cfg, uint8_t bt) const {   // obs_prediction_mode_context_map, codec/interface.rs:285-319
        std::memcpy(cfg.literal_context_map, pm_cmap.data(), sizeof(cfg.literal_context_map));
        std::memcpy(cfg.mixing_mask, pm_mixing.data(), sizeof(cfg.mixing_mask));
        cfg.prediction_mode = pm_mode; cfg.btype = bt; cfg.context_mixing = pm_mixing_math; cfg.reserved = 0;
        for (int i = 0; i < 4; ++i) cfg.literal_adaptation[i] = divans_speed{pm_speeds[i].inc, pm_speeds[i].lim};
    }

  private:
    void touch_lru(uint8_t val) {          // obs_context_map_for_lru, codec/interface.rs:421-455
        int found = -1;
        for (int i = 0; i < 13; ++i) if (lru[i] == val) { found = i; break; }
        const int shift = found < 0 ? 12 : found;
        for (int i = shift; i > 0; --i) lru[i] = lru[i - 1];
        lru[0] = val;
    }
};

// ---------------------------------------------------------------- Mux, src/mux.rs
class Mux {
  public:
    struct Stream { std::vector<uint8_t> buf; size_t start = 0, end = 0; size_t avail() const { return end - start; } };
    Stream s[2];
    uint32_t leftover = 0; int leftover_stream = 0;
    size_t last_flush[2] = {0, 0}, bytes_flushed = 0;
    int eof = 0;

    void prep(int id, size_t len) {       // prep_push_for_n_bytes :285-329
        Stream& b = s[id];
        if (b.buf.size() - b.end >= len) return;
        const size_t have = b.avail();
        if (b.buf.size() >= have + len + 3 && (b.start == b.end || (b.start >= 16384 && b.start > have + 3))) {
            std::memmove(b.buf.data() + 3, b.buf.data() + b.start, have);
            b.end = 3 + have; b.start = 3;
            return;
        }
        const uint64_t desired = 3 + len + have;
        int lg = 0; while ((desired >> lg) != 0) ++lg;
        lg = std::max(lg + 1, 9);
        std::vector<uint8_t> nb((size_t)1 << lg, 0);
        if (have) std::memcpy(nb.data() + 3, b.buf.data() + b.start, have);
        b.buf.swap(nb); b.end = 3 + have; b.start = 3;
    }
    void push(int id, const uint8_t* p, size_t n) { if (!n) return; prep(id, n); std::memcpy(s[id].buf.data() + s[id].end, p, n); s[id].end += n; }

    size_t serialize(uint8_t* out, size_t cap) {   // :445-476
        size_t off = 0;
        if (leftover) off += copy_leftover(out, cap);
        while (off < cap) {
            bool any = false;
            const size_t lo = std::min(last_flush[0], last_flush[1]), hi = std::max(last_flush[0], last_flush[1]);
            for (int i = 0; i < 2; ++i) {
                const bool lagging = hi > kVariance + last_flush[i];
                if (s[i].avail() >= chunk_size(last_flush[i], lagging) && last_flush[i] <= lo + kVariance) {
                    any = true;
                    emit(i, out, cap, off, lagging);
                    if (leftover) break;
                }
            }
            if (!any) break;
        }
        return off;
    }
    size_t close(uint8_t* out, size_t cap) {       // serialize_close :477-518 + flush_internal :519-561
        if (eof == 3) return 0;
        size_t off = 0;
        if (leftover) off += copy_leftover(out, cap);
        while (off < cap) {
            bool any = false, have = false; size_t lf = 0;
            for (int i = 0; i < 2; ++i) {
                const bool nonempty = s[i].avail() != 0;
                if (!have ? nonempty : (last_flush[i] < lf && nonempty)) { lf = last_flush[i]; have = true; }
            }
            for (int i = 0; i < 2; ++i) {
                if (!have || last_flush[i] <= lf + kVariance) {
                    const size_t before = off;
                    if (s[i].avail()) emit(i, out, cap, off, true);
                    if (off != before) any = true;
                    if (leftover) break;
                }
            }
            if (!any) break;
        }
        static const uint8_t marker[3] = {0xff, 0xfe, 0xff};
        for (int st = 0; st < 3; ++st) {
            if (off == cap) return off;
            if (eof == st) { out[off++] = marker[st]; eof = st + 1; }
        }
        return off;
    }
    // deserialize :384-444.  Returns bytes consumed; stops after the EOF marker.
    size_t deserialize(const uint8_t* in, size_t n) {
        size_t pos = 0;
        while (pos < n && eof != 3) {
            const uint8_t c = in[pos];
            if (pending == 2) { lsb = c; pending = 3; ++pos; }
            else if (pending == 3) { count = ((uint32_t)lsb | ((uint32_t)c << 8)) + 1; pending = 1; ++pos; }
            else if (pending == 1) {
                const size_t take = std::min<size_t>(count, n - pos);
                push(pstream, in + pos, take); pos += take; count -= (uint32_t)take;
                if (count == 0) pending = 0;
            } else {
                if (c == 0xff || (c == 0xfe && eof != 0)) {
                    static const uint8_t marker[3] = {0xff, 0xfe, 0xff};
                    bool progressed = false;
                    while (pos < n && eof < 3 && in[pos] == marker[eof]) { ++eof; ++pos; progressed = true; }
                    if (!progressed || eof == 3 || pos < n) return pos;
                    continue;
                }
                pstream = c & 1;
                if (c < 16) { pending = 2; ++pos; }
                else { count = 1024u << ((c >> 4) << 1); pending = 1; ++pos; }
            }
        }
        return pos;
    }

  private:
    static constexpr size_t kVariance = 131073;
    int pending = 0, pstream = 0; uint32_t count = 0; uint8_t lsb = 0;
    static size_t chunk_size(size_t last, bool lagging) { return lagging ? 16 : last <= 1024 ? 4096 : last <= 65536 ? 16384 : 65536; }
    size_t copy_leftover(uint8_t* out, size_t cap) {
        const size_t n = std::min<size_t>(leftover, cap);
        Stream& b = s[leftover_stream];
        std::memcpy(out, b.buf.data() + b.start, n); b.start += n; leftover -= (uint32_t)n;
        return n;
    }
    static int header_for(int id, size_t n, bool lagging, uint8_t hdr[3], size_t& count) {   // get_code :55-78
        if (!lagging || n == 4096 || n == 16384 || n >= 65536) {
            if (n < 4096) return header_for(id, n, true, hdr, count);
            if (n < 16384) { hdr[0] = (uint8_t)(id | 0x10); count = 4096; return 1; }
            if (n < 65536) { hdr[0] = (uint8_t)(id | 0x20); count = 16384; return 1; }
            hdr[0] = (uint8_t)(id | 0x30); count = 65536; return 1;
        }
        hdr[0] = (uint8_t)id; hdr[1] = (uint8_t)((n - 1) & 0xff); hdr[2] = (uint8_t)((n - 1) >> 8); count = n;
        return 3;
    }
    void emit(int id, uint8_t* out, size_t cap, size_t& off, bool lagging) {   // serialize_stream_id :339-382
        Stream& b = s[id];
        uint8_t hdr[3]; size_t count;
        const int hl = header_for(id, b.avail(), lagging, hdr, count);
        bytes_flushed += count;
        const size_t total = count + (size_t)hl;
        b.start -= (size_t)hl;
        std::memcpy(b.buf.data() + b.start, hdr, (size_t)hl);
        last_flush[id] = bytes_flushed;
        const size_t n = std::min(total, cap - off);
        std::memcpy(out + off, b.buf.data() + b.start, n);
        b.start += n;
        if (b.start == b.end) { b.start = 3; b.end = 3; }
        off += n;
        if (n != total) { leftover = (uint32_t)(total - n); leftover_stream = id; }
    }
};

// ---------------------------------------------------------------- output as the caller sees it: one buffer per call
class CallSink {
  public:
    CallSink(std::vector<uint8_t>& out, size_t call_buffer) : out_(out), buf_(call_buffer ? call_buffer : 65536), len_(out.size()) {}
    // A full buffer does not by itself end the call: the codec goes back to its caller only where a drain cannot finish or the
    // Mux / header need room (new_call() at exactly those points); until then it keeps coding and the Mux keeps the bytes.
    size_t room() const { return buf_ - used_; }
    // n writable bytes behind what has been committed.  The vector only grows here and finish() cuts it to the committed length: growing and
    // shrinking it per reservation zero-filled up to a whole call buffer about ten times per container -- two thirds of the time a container's
    // assembly took (17 -> 6 us per 64 KiB stream, round 6).
    uint8_t* reserve(size_t n) { if (out_.size() < len_ + n) out_.resize(std::max(len_ + n, 2 * out_.size())); return out_.data() + len_; }
    void commit(size_t /*reserved*/, size_t used) { len_ += used; used_ += used; }
    void finish() { out_.resize(len_); }
    void new_call() { used_ = 0; }
  private:
    std::vector<uint8_t>& out_; size_t buf_, used_ = 0, len_;
};

// The application around the compressor (c/example.c:26-60): a call that returns NEEDS_MORE_OUTPUT is followed by another call with
// an empty buffer.  Once an encode call has taken all of its input, that next call is the next piece's divans_encode or
// divans_encode_flush, which first finishes the frozen commands (divans_compressor.rs:189-207): the NewCall step that follows in
// the plan has then already happened.
struct Calls {
    CallSink& sink; bool input_done = false, next_call_started = false;
    explicit Calls(CallSink& s) : sink(s) {}
    void returns_for_output() { sink.new_call(); if (input_done) { input_done = false; next_call_started = true; } }
    void next_call() { if (next_call_started) next_call_started = false; else sink.new_call(); input_done = false; }
};

// drain_or_fill_static_buffer for an encoder, codec/interface.rs:868-895: linearize what the Mux will give, make room in the
// stream's buffer, pop; with coder bytes left and the caller's buffer full it reports NeedsMoreOutput -- which every caller hands
// to the application (`retry`: the re-entered call repeats the drain) except code_nibble_array after the last byte of a Literal
// (literal.rs:376-390: the command completes and the LIT coder keeps its bytes until the next LIT drain).
static void drain(Mux& mux, Calls& calls, int id, const uint8_t* coder_out, size_t avail_end, size_t& drained, bool retry = true) {
    CallSink& sink = calls.sink;
    while (drained < avail_end) {
        const size_t room = sink.room();
        uint8_t* p = sink.reserve(room);
        const size_t n = mux.serialize(p, room);
        sink.commit(room, n);
        mux.prep(0, 16); mux.prep(1, 16);                                 // write_buffer :184-204
        Mux::Stream& b = mux.s[id];
        const size_t take = std::min(avail_end - drained, b.buf.size() - b.end);
        if (take) std::memcpy(b.buf.data() + b.end, coder_out + drained, take);
        b.end += take; drained += take;
        if (drained < avail_end && sink.room() == 0) {
            if (!retry) return;
            calls.returns_for_output();
        }
    }
}

// One divans_gpu_codec per configuration is kept alive between per-stream states (creating one allocates its CDF
// tables, spill and staging buffers on the device): a released handle parks its codec here and the next stream with the
// same divans_lit_config on the same device picks it up if it is large enough.
struct CodecCache {
    std::mutex mu;
    divans_gpu_codec* codec = nullptr; int device = -1; uint32_t max_len = 0;
    std::unique_ptr<divans_lit_config> cfg;
    ~CodecCache() { /* the HIP runtime may already be gone at process exit: leave the codec to the OS */ }
};
static CodecCache& codec_cache() { static CodecCache* c = new CodecCache(); return *c; }

struct GpuCodecHandle {
    divans_gpu_codec* c = nullptr; int device = 0; uint32_t max_len = 0;
    std::unique_ptr<divans_lit_config> cfg;
    int acquire(const divans_lit_config& want, int dev, uint32_t need_len) {
        CodecCache& cc = codec_cache();
        {
            std::lock_guard<std::mutex> g(cc.mu);
            if (cc.codec && cc.device == dev && cc.max_len >= need_len && std::memcmp(cc.cfg.get(), &want, sizeof(want)) == 0) {
                c = cc.codec; cc.codec = nullptr; device = dev; max_len = cc.max_len; cfg = std::move(cc.cfg);
                return 0;
            }
        }
        cfg = std::make_unique<divans_lit_config>(want);
        device = dev; max_len = std::max<uint32_t>(need_len, 65536u);
        int rc = divans_gpu_codec_create(&c, cfg.get(), dev, nullptr, max_len);
        if (rc) return rc;
        (void)divans_gpu_codec_set_geometry(c, 1, 0xffffffffu);   // one stream: one workgroup's worth of tables is plenty
        return 0;
    }
    ~GpuCodecHandle() {
        if (!c) return;
        CodecCache& cc = codec_cache();
        divans_gpu_codec* old = nullptr;
        {
            std::lock_guard<std::mutex> g(cc.mu);
            old = cc.codec;
            cc.codec = c; cc.device = device; cc.max_len = max_len; cc.cfg = std::move(cfg);
        }
        if (old) divans_gpu_codec_destroy(old);
    }
};

int lit_config_from_prediction_mode(const StreamOptions& opt, const PredictionModeIn* pm, divans_lit_config& cfg) {
    std::memset(&cfg, 0, sizeof(cfg));
    if (!pm) {   // LiteralBookKeeping::new, codec/interface.rs:244-262: zero maps, lsb6, default speeds
        for (auto& s : cfg.literal_adaptation) s = divans_speed{0x10, 0x2000};
        return 0;
    }
    CommandModel probe(opt);
    RansEncoder scratch;
    NibbleCoder pn; pn.enc = &scratch;
    probe.command_type(pn, 7);
    if (!probe.prediction_mode(pn, pm)) return DIVANS_GPU_EINVAL;
    probe.fill_lit_config(cfg, 0);
    return 0;
}

// ---------------------------------------------------------------- when the internal compressor emits which command
// RawToCmdState (raw_to_cmd/mod.rs:32-181) buffers the caller's bytes in a ring of 2^window bytes and hands out commands
// only when the ring is full or at flush.  `decode` / `output` are its two ring indices; the byte counts of the Literal
// commands fall out of how they chase each other (first lap 2^w bytes, second 2^w - 1, then pairs of k-2 and 2^w-k+1).
// Where the reference would emit ring bytes it never refilled (input ending inside the span a lap first writes at the end
// of the ring: the index is reset regardless, :70-72), only the bytes actually written are emitted.
struct RingEvents {
    enum Kind { PredictionMode, Literal, NewCall, InputDone };
    struct Event { Kind kind; size_t len; };
    std::vector<Event> events;
    size_t ring, decode = 0, output = 0, fresh_tail = 0;
    bool header_done = false;
    explicit RingEvents(size_t ring_bytes) : ring(ring_bytes) {}
    void literal(size_t len) { if (len) events.push_back({Literal, len}); }
    void flush() {                                       // RawToCmdState::flush
        if (!header_done) { header_done = true; events.push_back({PredictionMode, 0}); }
        if (decode < output) {
            literal(fresh_tail); fresh_tail = 0;
            if (decode == ring) decode = 0;
            output = 0;
        }
        if (decode != output) { literal(decode - output); output = decode; }
    }
    void feed(size_t bytes) {                            // one divans_encode call: DivansCompressor::encode looping over stream()
        while (true) {
            if (decode >= output) {
                const size_t k = std::min(ring - decode, bytes);
                bytes -= k; decode += k;
                if (output != 0) { fresh_tail = decode - output; decode = 0; }
            }
            if (decode < output) {
                const size_t k = std::min(output - 1 - decode, bytes);
                bytes -= k; decode += k;
            }
            const bool full = decode == ring || decode + 1 == output;
            if (!full) break;
            if (bytes == 0) events.push_back({InputDone, 0});   // what this flush emits is coded with the call's input all taken
            flush();
            if (bytes == 0) break;
        }
    }
};

// The default PredictionMode of the internal compressor, raw_to_cmd/mod.rs:115-143
static PredictionModeIn internal_prediction_mode() {
    PredictionModeIn pm;
    pm.literal_context_map.resize(64); for (int i = 0; i < 64; ++i) pm.literal_context_map[i] = (uint8_t)(i & 0x3f);
    pm.distance_context_map = {0, 1, 2, 3};
    pm.mixing_values.assign(DIVANS_GPU_NUM_MIXING_VALUES, 4);
    pm.has_context_speeds = true;
    return pm;
}

// ---- phase A: everything about a stream that does not need its LIT-coder bytes -------------------------------------
// For literal-only streams the CMD coder sees only lengths and options, never the data: its bytes, and the order in
// which coder bytes become available to the Mux, can be worked out while the GPU is still coding the literals
// (the reference overlaps the same two halves with a worker thread, threading.rs:88-100).
// The internal compressor's PredictionMode command depends on the options alone and is every stream's first command: its 8.2 k nibbles
// through the model (three quarters of a plan's time) are the same for every stream of a batch.
struct PlanPrefix { CommandModel model; RansEncoder cmd; divans_lit_config cfg; explicit PlanPrefix(const StreamOptions& o) : model(o) {} };
std::shared_ptr<const PlanPrefix> make_plan_prefix(const StreamOptions& opt) {
    std::shared_ptr<PlanPrefix> p(new PlanPrefix(opt));
    const PredictionModeIn pm = internal_prediction_mode();
    NibbleCoder nc; nc.enc = &p->cmd;
    p->model.command_type(nc, 7);
    if (!p->model.prediction_mode(nc, &pm) || p->cmd.failed || !p->cmd.out.empty()) return nullptr;
    p->model.fill_lit_config(p->cfg, 0);
    return p;
}

int plan_stream(const StreamOptions& opt, size_t n, const std::vector<size_t>* call_inputs, StreamPlan& plan, const PlanPrefix* prefix) {
    plan = StreamPlan();
    if (n > 0x7fffffffu) return DIVANS_GPU_EINVAL;
    plan.n = n;
    plan.window = std::min(24, std::max(10, opt.window_size));
    RingEvents ring((size_t)1 << plan.window);
    {
        size_t left = n, calls = call_inputs ? call_inputs->size() : 1;
        for (size_t k = 0; k < calls; ++k) {               // every encode call brings a fresh output buffer; the first also carries the header
            const size_t m = std::min(left, call_inputs ? (*call_inputs)[k] : n);
            if (k) ring.events.push_back({RingEvents::NewCall, 0});
            ring.feed(m); left -= m;
        }
        if (left) return DIVANS_GPU_EINVAL;
        ring.events.push_back({RingEvents::NewCall, 0});   // the flush calls
        ring.flush();                                      // an empty input still flushes a PredictionMode command
    }
    const PredictionModeIn pm = internal_prediction_mode();
    CommandModel model(opt);
    RansEncoder cmd;
    NibbleCoder nc; nc.enc = &cmd;
    size_t cmd_logged = 0;
    // drain_or_fill_internal_buffer_cmd runs before every CMD nibble; it only does something when the coder has new bytes
    nc.before = [&]() { if (cmd.out.size() != cmd_logged) { cmd_logged = cmd.out.size(); plan.steps.push_back({StreamPlan::CmdAvail, (uint32_t)cmd_logged}); } };
    uint64_t lit_syms = 0; uint32_t chunk_idx = 0;
    for (const RingEvents::Event& ev : ring.events) {
        if (ev.kind == RingEvents::NewCall) { plan.steps.push_back({StreamPlan::NewCall, 0}); continue; }
        if (ev.kind == RingEvents::InputDone) { plan.steps.push_back({StreamPlan::InputDone, 0}); continue; }
        if (ev.kind == RingEvents::PredictionMode) {
            if (prefix) {     // always the stream's first command: model and coder are still as constructed, take over the prefix's
                model = prefix->model; cmd = prefix->cmd; plan.cfg = prefix->cfg;
                continue;
            }
            model.command_type(nc, 7);
            if (!model.prediction_mode(nc, &pm)) return DIVANS_GPU_EINVAL;
            model.fill_lit_config(plan.cfg, 0);
            continue;
        }
        model.command_type(nc, 3);
        uint32_t len_out;
        if (!model.literal_length(nc, (uint32_t)ev.len, len_out)) return DIVANS_GPU_EINVAL;
        nc.before();
        plan.steps.push_back({StreamPlan::LitDrain, 0});   // encode_or_decode_content_bytes drains the LIT coder before the first nibble (:426-433)
        // every 65 536th LIT symbol flushes a chunk, drained right after that nibble (literal.rs:309-315,368-374)
        const uint64_t end_syms = lit_syms + 2 * (uint64_t)ev.len;
        while ((lit_syms / 65536 + 1) * 65536 <= end_syms) {
            lit_syms = (lit_syms / 65536 + 1) * 65536;
            plan.steps.push_back({lit_syms == end_syms ? StreamPlan::LitChunkLast : StreamPlan::LitChunk, chunk_idx++});
        }
        lit_syms = end_syms;
    }
    // DivansCodec::flush, codec/mod.rs:424-554: EOF nibble, EncodedShutdownNode, ShutdownCoder(0), ShutdownCoder(1), CoderBufferDrain
    model.command_type(nc, 0xf);
    nc.before();
    plan.steps.push_back({StreamPlan::LitDrain, 0});   // EncodedShutdownNode: both coders drained
    cmd.flush();
    nc.before();
    if (lit_syms % 65536) plan.steps.push_back({StreamPlan::LitChunk, chunk_idx++});   // the partial last chunk
    plan.lit_chunks = chunk_idx;
    if (cmd.failed) return DIVANS_GPU_EINVAL;
    plan.cmd.swap(cmd.out);
    return 0;
}

// ---- phase B: header, Mux replay in event order, EOF marker, CRC trailer --------------------------------------------
int assemble_container(const StreamPlan& plan, const uint8_t* lit, size_t lit_size, const uint32_t* chunk_bytes, size_t call_buffer,
                       std::vector<uint8_t>& out) {
    out.clear();
    Mux mux;
    CallSink sink(out, call_buffer);
    Calls calls(sink);
    {   // header in the first encode() call, divans_compressor.rs:126-131,150-174
        uint8_t hdr[16] = {0xff, 0xe5, 0x8c, 0x9f, 0, (uint8_t)plan.window, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        size_t done = 0;
        while (done < 16) {
            if (sink.room() == 0) calls.returns_for_output();
            const size_t room = sink.room(), k = std::min<size_t>(16 - done, room); std::memcpy(sink.reserve(k), hdr + done, k); sink.commit(k, k); done += k;
        }
    }
    size_t cmd_drained = 0, lit_drained = 0, lit_avail = 0;      // (the coders' bytes are read where they lie: no per-container copy of the LIT stream)
    for (const StreamPlan::Step& st : plan.steps) {
        if (st.kind == StreamPlan::NewCall) calls.next_call();
        else if (st.kind == StreamPlan::InputDone) calls.input_done = true;
        else if (st.kind == StreamPlan::CmdAvail) drain(mux, calls, 0, plan.cmd.data(), st.value, cmd_drained);
        else if (st.kind == StreamPlan::LitDrain) drain(mux, calls, 1, lit, lit_avail, lit_drained);
        else {
            lit_avail += chunk_bytes[st.value]; if (lit_avail > lit_size) return DIVANS_GPU_EINVAL;
            drain(mux, calls, 1, lit, lit_avail, lit_drained, st.kind != StreamPlan::LitChunkLast);
        }
    }
    if (lit_avail != lit_size || cmd_drained != plan.cmd.size()) return DIVANS_GPU_EINVAL;
    while (mux.eof != 3) {                          // MuxDrain
        if (sink.room() == 0) calls.returns_for_output();
        const size_t room = sink.room();
        uint8_t* p = sink.reserve(room);
        const size_t k = mux.close(p, room);
        sink.commit(room, k);
    }
    sink.finish();
    const uint32_t crc = crc32c(0, out.data(), out.size());
    const uint8_t tr[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), 'a', 'n', 's', '~'};
    out.insert(out.end(), tr, tr + 8);
    return 0;
}

// ---------------------------------------------------------------- the same compressor, one call at a time
// What plan_stream + assemble_container do for a whole stream at once, as a machine that stops wherever the reference returns to its
// caller.  Commands are coded the moment the ring emits them (CMD nibbles on the host, literal bytes on the GPU); what that leaves for
// the Mux is a queue of steps, and a step whose drain cannot finish because the caller's buffer is full stays at the head of the
// queue: the next call starts by running it again, which is exactly the re-entry of divans_compressor.rs:189-207.
struct StreamEncoder::Impl {
    StreamOptions opt; int device; int window;
    std::vector<uint8_t> ring; size_t decode = 0, output = 0, fresh_tail = 0; bool header_done = false;   // RawToCmdState
    int header_sent = 0;
    CommandModel model; RansEncoder cmd; NibbleCoder nc; size_t cmd_logged = 0;
    GpuCodecHandle gpu; bool lit_started = false; divans_lit_config cfg;
    uint64_t last8 = 0; uint64_t lit_syms = 0;
    struct Step { StreamPlan::Kind kind; size_t value; bool started; };
    std::deque<Step> steps;
    std::deque<std::vector<uint8_t>> chunks;          // LIT chunks coded but not yet handed to the Mux
    std::vector<uint8_t> lit; size_t lit_avail = 0, lit_drained = 0, cmd_drained = 0;
    Mux mux;
    uint32_t crc = 0;
    int phase = 0;                                    // flush: 0 commands still to come, 1 steps, 2 MuxDrain, 3 trailer, 4 done
    int trailer_sent = 0; uint8_t trailer[8];
    int error = 0;

    Impl(const StreamOptions& o, int dev) : opt(o), device(dev), window(std::min(24, std::max(10, o.window_size))), model(o) {
        ring.resize((size_t)1 << window);
        nc.enc = &cmd;
        nc.before = [this]() { if (cmd.out.size() != cmd_logged) { cmd_logged = cmd.out.size(); steps.push_back({StreamPlan::CmdAvail, cmd_logged, false}); } };
    }
    void put(uint8_t* out, size_t* off, const uint8_t* src, size_t n) { std::memcpy(out + *off, src, n); crc = crc32c(crc, src, n); *off += n; }

    // drain_or_fill_static_buffer against the caller's real buffer: false = NeedsMoreOutput (the step runs again in the next call)
    bool drain(int id, const std::vector<uint8_t>& coder_out, size_t avail_end, size_t& drained, bool retry, uint8_t* out, size_t cap, size_t* off) {
        while (drained < avail_end) {
            const size_t room = cap - *off;
            const size_t k = mux.serialize(out + *off, room);
            crc = crc32c(crc, out + *off, k); *off += k;
            mux.prep(0, 16); mux.prep(1, 16);
            Mux::Stream& b = mux.s[id];
            const size_t take = std::min(avail_end - drained, b.buf.size() - b.end);
            std::memcpy(b.buf.data() + b.end, coder_out.data() + drained, take);
            b.end += take; drained += take;
            if (drained < avail_end && cap == *off) return !retry;      // status dropped (LitChunkLast) or handed to the caller
        }
        return true;
    }
    bool run_steps(uint8_t* out, size_t cap, size_t* off) {          // false = NeedsMoreOutput
        while (!steps.empty()) {
            Step& st = steps.front();
            if (st.kind == StreamPlan::CmdAvail) { if (!drain(0, cmd.out, st.value, cmd_drained, true, out, cap, off)) return false; }
            else {
                if (st.kind != StreamPlan::LitDrain && !st.started) {
                    st.started = true;
                    if (lit_drained == lit.size()) { lit.clear(); lit_drained = 0; lit_avail = 0; }   // everything before this chunk has left
                    lit.insert(lit.end(), chunks.front().begin(), chunks.front().end());
                    lit_avail = lit.size();
                    chunks.pop_front();
                }
                if (!drain(1, lit, lit_avail, lit_drained, st.kind != StreamPlan::LitChunkLast, out, cap, off)) return false;
            }
            steps.pop_front();
        }
        return true;
    }
    // one Literal command of the ring: CMD nibbles, then its bytes through the GPU coder, then the steps its chunks mean for the Mux
    int code_literal(const uint8_t* data, size_t len) {
        model.command_type(nc, 3);
        uint32_t len_out;
        if (!model.literal_length(nc, (uint32_t)len, len_out)) return DIVANS_GPU_EINVAL;
        nc.before();
        steps.push_back({StreamPlan::LitDrain, 0, false});
        if (!lit_started) {
            int rc = gpu.acquire(cfg, device, (uint32_t)ring.size()); if (rc) return rc;
            rc = divans_gpu_lit_stream_begin(gpu.c); if (rc) return rc;
            lit_started = true;
        }
        std::vector<uint8_t> buf((size_t)divans_gpu_lit_encode_bound((uint32_t)len) + 65536u);
        const uint32_t max_chunks = (uint32_t)(len / 32768u + 2u);
        std::vector<uint32_t> sizes(max_chunks); uint32_t n_chunks = 0; size_t got = 0;
        int rc = divans_gpu_lit_stream_encode(gpu.c, data, (uint32_t)len, last8, buf.data(), buf.size(), sizes.data(), max_chunks, &n_chunks, &got);
        if (rc) return rc;
        for (size_t i = len < 8 ? 0 : len - 8; i < len; ++i) last8 = (last8 >> 8) | ((uint64_t)data[i] << 56);
        const uint64_t end_syms = lit_syms + 2 * (uint64_t)len;
        size_t pos = 0; uint32_t k = 0;
        while ((lit_syms / 65536 + 1) * 65536 <= end_syms) {
            lit_syms = (lit_syms / 65536 + 1) * 65536;
            if (k >= n_chunks) return DIVANS_GPU_EINVAL;
            chunks.emplace_back(buf.begin() + pos, buf.begin() + pos + sizes[k]); pos += sizes[k]; ++k;
            steps.push_back({lit_syms == end_syms ? StreamPlan::LitChunkLast : StreamPlan::LitChunk, 0, false});
        }
        lit_syms = end_syms;
        return (k == n_chunks && pos == got) ? 0 : DIVANS_GPU_EINVAL;
    }
    int code_prediction_mode() {
        const PredictionModeIn pm = internal_prediction_mode();
        model.command_type(nc, 7);
        if (!model.prediction_mode(nc, &pm)) return DIVANS_GPU_EINVAL;
        model.fill_lit_config(cfg, 0);
        return 0;
    }
    int ring_flush() {                                    // RawToCmdState::flush, raw_to_cmd/mod.rs:105-181 (fresh bytes only, see RingEvents)
        int rc = 0;
        if (!header_done) { header_done = true; if ((rc = code_prediction_mode())) return rc; }
        if (decode < output) {
            if (fresh_tail && (rc = code_literal(ring.data() + output, fresh_tail))) return rc;
            fresh_tail = 0;
            if (decode == ring.size()) decode = 0;
            output = 0;
        }
        if (decode != output) { if ((rc = code_literal(ring.data() + output, decode - output))) return rc; output = decode; }
        return 0;
    }
    bool header(uint8_t* out, size_t cap, size_t* off) {
        const uint8_t hdr[16] = {0xff, 0xe5, 0x8c, 0x9f, 0, (uint8_t)window, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const size_t n = std::min<size_t>(16 - header_sent, cap - *off);
        put(out, off, hdr + header_sent, n); header_sent += (int)n;
        return header_sent == 16;
    }
};

StreamEncoder::StreamEncoder(const StreamOptions& opt, int device) : p_(new Impl(opt, device)) {}
StreamEncoder::~StreamEncoder() { delete p_; }

int StreamEncoder::encode(const uint8_t* in, size_t n, size_t* in_off, uint8_t* out, size_t cap, size_t* out_off) {   // divans_compressor.rs:276-337
    Impl& s = *p_;
    if (s.error) return s.error;
    if (s.phase != 0) return s.error = DIVANS_GPU_EINVAL;                // NotAllowedToEncodeAfterFlush
    if (!s.header(out, cap, out_off) || !s.run_steps(out, cap, out_off)) return 1;
    for (;;) {                                                          // RawToCmdState::stream, raw_to_cmd/mod.rs:55-104
        if (s.decode >= s.output) {
            const size_t k = std::min(s.ring.size() - s.decode, n - *in_off);
            std::memcpy(s.ring.data() + s.decode, in + *in_off, k);
            *in_off += k; s.decode += k;
            if (s.output != 0) { s.fresh_tail = s.decode - s.output; s.decode = 0; }
        }
        if (s.decode < s.output) {
            const size_t k = std::min(s.output - 1 - s.decode, n - *in_off);
            std::memcpy(s.ring.data() + s.decode, in + *in_off, k);
            *in_off += k; s.decode += k;
        }
        if (!(s.decode == s.ring.size() || s.decode + 1 == s.output)) return 0;      // ring not full: all input taken
        if (int rc = s.ring_flush()) return s.error = rc;
        if (!s.run_steps(out, cap, out_off)) return 1;
        if (*in_off == n) return 0;
    }
}

int StreamEncoder::flush(uint8_t* out, size_t cap, size_t* out_off) {   // divans_compressor.rs:362-426 + codec/mod.rs:424-560
    Impl& s = *p_;
    if (s.error) return s.error;
    if (!s.header(out, cap, out_off) || !s.run_steps(out, cap, out_off)) return 1;
    if (s.phase == 0) {
        if (int rc = s.ring_flush()) return s.error = rc;
        s.model.command_type(s.nc, 0xf);                                // the end marker through the command-type prior
        s.nc.before();
        s.steps.push_back({StreamPlan::LitDrain, 0, false});            // EncodedShutdownNode
        s.cmd.flush();                                                  // ShutdownCoder(0)
        s.nc.before();
        if (s.cmd.failed) return s.error = DIVANS_GPU_EINVAL;
        if (s.lit_started) {                                            // ShutdownCoder(1), CoderBufferDrain
            std::vector<uint8_t> buf((size_t)divans_gpu_lit_encode_bound(32768u) + 64u); size_t got = 0;
            if (int rc = divans_gpu_lit_stream_finish(s.gpu.c, buf.data(), buf.size(), &got)) return s.error = rc;
            if (got) { s.chunks.emplace_back(buf.begin(), buf.begin() + got); s.steps.push_back({StreamPlan::LitChunk, 0, false}); }
        }
        s.phase = 1;
    }
    if (s.phase == 1) { if (!s.run_steps(out, cap, out_off)) return 1; s.phase = 2; }
    while (s.phase == 2) {                                              // MuxDrain
        if (cap == *out_off) return 1;
        const size_t k = s.mux.close(out + *out_off, cap - *out_off);
        s.crc = crc32c(s.crc, out + *out_off, k); *out_off += k;
        if (s.mux.eof == 3 && s.mux.s[0].avail() == 0 && s.mux.s[1].avail() == 0 && s.mux.leftover == 0) {
            const uint32_t c = s.crc;
            const uint8_t tr[8] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24), 'a', 'n', 's', '~'};
            std::memcpy(s.trailer, tr, 8);
            s.phase = 3;
        }
    }
    if (s.phase == 3) {                                                 // WriteChecksum
        const size_t k = std::min<size_t>(8 - s.trailer_sent, cap - *out_off);
        std::memcpy(out + *out_off, s.trailer + s.trailer_sent, k); *out_off += k; s.trailer_sent += (int)k;
        if (s.trailer_sent < 8) return 1;
        s.phase = 4;
    }
    return 0;
}

struct ParseMemo::Impl {
    struct Entry { std::string cmd; uint64_t total; int cfg_id; };
    std::mutex mu;
    // keyed by a 64-bit hash of the CMD bytes computed OUTSIDE the lock (a mixing configuration's CMD stream is 4-5 KB: hashing it and copying a
    // 25 KB configuration under the lock serialised the parsing threads of a batch, round 6); the bytes themselves decide a hit
    std::unordered_multimap<uint64_t, Entry> map;
    std::vector<std::shared_ptr<const divans_lit_config>> cfgs;   // the distinct LIT configurations seen (25 KB each): entries name one by index
    static constexpr size_t kMaxEntries = 65536;    // ~100 bytes each: a batch of all-different lengths still fits
    static constexpr size_t kMaxConfigs = 64;
    static uint64_t hash(const uint8_t* p, size_t n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
        for (; i < n; ++i) { h = (h ^ p[i]) * 0x100000001b3ull; }
        return h ^ (h >> 29);
    }
    const Entry* find(uint64_t h, const uint8_t* p, size_t n) const {     // under mu
        const auto range = map.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) if (it->second.cmd.size() == n && std::memcmp(it->second.cmd.data(), p, n) == 0) return &it->second;
        return nullptr;
    }
    int intern(std::shared_ptr<const divans_lit_config>& c) {        // under mu; c becomes the interned object
        for (size_t i = 0; i < cfgs.size(); ++i) if (std::memcmp(cfgs[i].get(), c.get(), sizeof(divans_lit_config)) == 0) { c = cfgs[i]; return (int)i; }
        if (cfgs.size() >= kMaxConfigs) return -1;
        cfgs.push_back(c);
        return (int)cfgs.size() - 1;
    }
};
ParseMemo::ParseMemo() : p_(new Impl()) {}
ParseMemo::~ParseMemo() { delete p_; }

// Host half of decoding: framing, CRC, CMD coder.  Leaves the LIT coder's bytes, the decoded size and its configuration.
// Demultiplexing a COMPLETE container without copying its LIT slices (mux.rs:445-561 read back): the CMD stream (a few hundred bytes) is
// collected, every LIT slice is recorded as (offset in `in`, length).  Returns the bytes used up to and including the end marker,
// 0 when the framing is damaged or the input ends before the marker (the caller then takes the copying path, which tells the two apart).
static size_t scan_container_body(const uint8_t* in, size_t n, size_t base, std::vector<uint8_t>& cmd, std::vector<std::pair<uint32_t, uint32_t>>& spans, size_t& lit_size) {
    size_t pos = 0;
    lit_size = 0;
    while (pos < n) {
        const uint8_t c = in[pos];
        if (c == 0xff) {
            if (n - pos < 3 || in[pos + 1] != 0xfe || in[pos + 2] != 0xff) return 0;
            return pos + 3;
        }
        size_t count;
        if (c < 16) {
            if (n - pos < 3) return 0;
            count = ((size_t)in[pos + 1] | ((size_t)in[pos + 2] << 8)) + 1; pos += 3;
        } else { count = (size_t)1024u << ((c >> 4) << 1); pos += 1; }
        if (count > n - pos || base + pos > 0xffffffffu) return 0;
        if (c & 1) { spans.emplace_back((uint32_t)(base + pos), (uint32_t)count); lit_size += count; }
        else cmd.insert(cmd.end(), in + pos, in + pos + count);
        pos += count;
    }
    return 0;
}

ParseStatus parse_container_host(const uint8_t* in, size_t n, bool skip_crc, size_t max_output, ParsedStream& ps, size_t* consumed, ParseMemo* memo, bool spans_only) {
    ps = ParsedStream();
    if (n < 16) return PARSE_NEED_MORE;
    if (in[0] != 0xff || in[1] != 0xe5 || in[2] != 0x8c || in[3] != 0x9f) return PARSE_CORRUPT;   // divans_decompressor.rs:38-52
    if (in[5] < 10 || in[5] >= 25) return PARSE_CORRUPT;
    Mux mux;
    size_t used = 0;
    std::vector<uint8_t> cmd_bytes;
    bool scanned = false;
    if (spans_only) {
        used = scan_container_body(in + 16, n - 16, 16, cmd_bytes, ps.lit_spans, ps.lit_size);
        scanned = used != 0;
        if (!scanned) { ps.lit_spans.clear(); ps.lit_size = 0; cmd_bytes.clear(); }
    }
    if (!scanned) {
        used = mux.deserialize(in + 16, n - 16);
        if (mux.eof != 3) return (16 + used < n) ? PARSE_CORRUPT : PARSE_NEED_MORE;
    }
    if (n - 16 - used < 8) return PARSE_NEED_MORE;
    const uint8_t* tr = in + 16 + used;
    const uint32_t crc = skip_crc ? 0u : crc32c(0, in, 16 + used);      // (not even computed when the caller does not want it checked)
    const uint8_t want[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), 'a', 'n', 's', '~'};
    if (std::memcmp(tr + 4, want + 4, 4) != 0) return PARSE_CORRUPT;               // codec/mod.rs:949-1017
    if (!skip_crc && std::memcmp(tr, want, 4) != 0) return PARSE_CORRUPT;
    if (consumed) *consumed = 16 + used + 8;
    const uint8_t* cmd_ptr = scanned ? cmd_bytes.data() : mux.s[0].buf.data() + mux.s[0].start;
    const size_t cmd_len = scanned ? cmd_bytes.size() : mux.s[0].avail();
    const size_t lit_len = scanned ? ps.lit_size : mux.s[1].avail();
    // the literal lengths are the stream's own claim: bound them before anything is allocated for them.  A literal
    // byte costs the LIT coder at least ~0.0007 bytes (two nibbles at the largest probability the fastest
    // speed allows), so a stream claiming more than 4096 bytes per coded LIT byte is lying.  (The claims only grow along the
    // CMD stream, so the same test on a remembered final sum refuses exactly the streams the walk would have refused on the way.)
    const uint64_t most = std::min<uint64_t>(max_output, (uint64_t)lit_len * 4096u + 65536u);
    auto finish = [&](uint64_t total) -> ParseStatus {
        ps.total = (size_t)total;
        // the kernels read whole 32-bit words; LIT streams are 16 + 4k bytes per chunk by construction
        if (!scanned) {
            ps.lit.assign(mux.s[1].buf.begin() + (long)mux.s[1].start, mux.s[1].buf.begin() + (long)mux.s[1].end);
            ps.lit_size = ps.lit.size();
        }
        if (ps.lit_size % 4) return PARSE_CORRUPT;
        if (total == 0 && ps.lit_size != 0) return PARSE_CORRUPT;
        return PARSE_OK;
    };
    uint64_t key = 0;
    if (memo) {
        key = ParseMemo::Impl::hash(cmd_ptr, cmd_len);
        uint64_t hit_total = 0; bool hit = false;
        {
            std::lock_guard<std::mutex> g(memo->p_->mu);
            if (const ParseMemo::Impl::Entry* e = memo->p_->find(key, cmd_ptr, cmd_len)) {
                hit = true; hit_total = e->total;
                ps.cfg = memo->p_->cfgs[(size_t)e->cfg_id]; ps.cfg_id = e->cfg_id;
            }
        }
        if (hit) {
            if (hit_total > 0x7fffffffu) return PARSE_UNSUPPORTED;
            if (hit_total > most) return PARSE_CORRUPT;
            return finish(hit_total);
        }
    }
    // CMD stream on the host
    StreamOptions o;
    CommandModel model(o);
    RansDecoder cd(cmd_ptr, cmd_len);
    NibbleCoder nc; nc.dec = &cd;
    uint8_t btype = 0; bool have_pm = false, seen_literal = false;
    uint64_t total = 0;
    for (;;) {
        const int code = model.command_type(nc, 0);
        if (cd.starved) return PARSE_CORRUPT;
        if (code == 0xf) break;
        if (code == 7) {
            // a second PredictionMode after literal bytes would change the literal coder's tables mid-stream:
            // the literal-only path codes one configuration per stream
            if (seen_literal) return PARSE_UNSUPPORTED;
            if (!model.prediction_mode(nc, nullptr)) return PARSE_CORRUPT;
            have_pm = true;
        } else if (code == 4) {
            if (seen_literal) return PARSE_UNSUPPORTED;
            model.block_switch_literal(nc, 0, 0, btype);
        } else if (code == 3) {
            uint32_t len;
            if (!model.literal_length(nc, 15, len)) return PARSE_CORRUPT;
            total += len; seen_literal = true;
            if (total > 0x7fffffffu) return PARSE_UNSUPPORTED;
            if (total > most) return PARSE_CORRUPT;
        } else return PARSE_UNSUPPORTED;   // Copy / Dict / command- and distance- block switches
        if (cd.starved) return PARSE_CORRUPT;
    }
    auto fresh = std::make_shared<divans_lit_config>();
    if (have_pm) model.fill_lit_config(*fresh, btype);
    else {   // LiteralBookKeeping::new defaults, codec/interface.rs:244-262
        std::memset(fresh.get(), 0, sizeof(divans_lit_config));
        fresh->btype = btype;
        for (auto& s : fresh->literal_adaptation) s = divans_speed{0x10, 0x2000};
    }
    ps.cfg = fresh;
    if (memo) {
        std::lock_guard<std::mutex> g(memo->p_->mu);
        const int id = memo->p_->intern(ps.cfg);
        ps.cfg_id = id;
        if (id >= 0 && memo->p_->map.size() < ParseMemo::Impl::kMaxEntries && !memo->p_->find(key, cmd_ptr, cmd_len))
            memo->p_->map.emplace(key, ParseMemo::Impl::Entry{std::string((const char*)cmd_ptr, cmd_len), total, id});
    }
    return finish(total);
}

void probe_container_host(const uint8_t* in, size_t n, int wire, divans_container_probe_fields& pr) {
    pr = divans_container_probe_fields();
    std::memset(&pr.cfg, 0, sizeof(pr.cfg));
    for (auto& s : pr.cfg.literal_adaptation) s = divans_speed{0x10, 0x2000};
    if (n < 16) { pr.status = 1; return; }
    if (in[0] != 0xff || in[1] != 0xe5 || in[2] != 0x8c || in[3] != 0x9f || in[5] < 10 || in[5] >= 25) { pr.status = 2; return; }
    pr.window = in[5];
    Mux mux;
    const size_t used = mux.deserialize(in + 16, n - 16);
    pr.cmd_bytes = (uint32_t)mux.s[0].avail(); pr.lit_bytes = (uint32_t)mux.s[1].avail();
    if (mux.eof != 3) { pr.status = (16 + used < n) ? 2 : 1; return; }
    if (n - 16 - used < 8) { pr.status = 1; return; }
    const uint8_t* tr = in + 16 + used;
    const uint32_t crc = crc32c(0, in, 16 + used);
    const uint8_t want[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), 'a', 'n', 's', '~'};
    if (std::memcmp(tr + 4, want + 4, 4) != 0) { pr.status = 2; return; }
    pr.crc_ok = std::memcmp(tr, want, 4) == 0;
    StreamOptions o; o.wire = wire;
    CommandModel model(o);
    RansDecoder cd(mux.s[0].buf.data() + mux.s[0].start, mux.s[0].avail());
    NibbleCoder nc; nc.dec = &cd;
    uint8_t btype = 0;
    pr.status = 0;
    for (;;) {
        const int code = model.command_type(nc, 0);
        if (cd.starved) { pr.status = 2; break; }
        if (code == 0xf) break;
        if (code == 7 && !pr.literal_bytes && !pr.have_pm) {
            if (!model.prediction_mode(nc, nullptr) || cd.starved) { pr.status = 2; break; }
            pr.have_pm = 1;
        } else if (code == 4 && !pr.literal_bytes) {
            model.block_switch_literal(nc, 0, 0, btype);
        } else if (code == 3) {
            uint32_t len;
            if (!model.literal_length(nc, 15, len) || cd.starved) { pr.status = 2; break; }
            if (!pr.literal_bytes) pr.first_literal_length = len;
            pr.literal_bytes += len;
        } else { pr.status = 3; pr.stopped_at = (uint8_t)code; break; }
        if (cd.starved) { pr.status = 2; break; }
        pr.commands += 1;
    }
    pr.cmd_nibbles = nc.nibbles;
    if (pr.have_pm) model.fill_lit_config(pr.cfg, btype); else pr.cfg.btype = btype;
}

// ---------------------------------------------------------------- the decompressor, one call at a time
struct StreamDecoder::Impl {
    bool skip_crc; size_t max_output; int device;
    uint8_t hdr[16]; int hdr_have = 0;
    Mux mux; uint32_t crc = 0;
    uint8_t trailer[8]; int trailer_have = 0; bool trailer_ok = false;
    // what the CMD coder has told so far (re-read from its first byte whenever more of it has arrived: it is a few hundred bytes)
    size_t cmd_seen = 0; bool cmd_done = false, have_cfg = false; uint64_t known = 0; divans_lit_config cfg;
    GpuCodecHandle gpu; bool lit_started = false;
    uint64_t done = 0; uint64_t last8 = 0;
    std::vector<uint8_t> ready; size_t ready_pos = 0;      // decoded, not yet handed out
    int error = 0;

    int read_commands() {       // codec/mod.rs:652-792 on the CMD bytes that are in; a command whose bytes are not all there is not counted
        Mux::Stream& c = mux.s[0];
        if (cmd_done || c.avail() == cmd_seen) return 0;
        cmd_seen = c.avail();
        StreamOptions o;
        CommandModel model(o);
        RansDecoder cd(c.buf.data() + c.start, c.avail());
        NibbleCoder nc; nc.dec = &cd;
        uint8_t btype = 0; bool have_pm = false, seen_literal = false;
        uint64_t total = 0;
        for (;;) {
            const int code = model.command_type(nc, 0);
            if (cd.starved) break;
            if (code == 0xf) { cmd_done = true; break; }
            if (code == 7) {
                if (seen_literal) return -101;
                if (!model.prediction_mode(nc, nullptr)) { if (cd.starved) break; return -100; }
                if (cd.starved) break;
                have_pm = true;
            } else if (code == 4) {
                if (seen_literal) return -101;
                model.block_switch_literal(nc, 0, 0, btype);
                if (cd.starved) break;
            } else if (code == 3) {
                uint32_t len;
                if (!model.literal_length(nc, 15, len)) { if (cd.starved) break; return -100; }
                if (cd.starved) break;
                if (!have_cfg) {         // the first Literal fixes the configuration the LIT coder runs under
                    if (have_pm) model.fill_lit_config(cfg, btype);
                    else { std::memset(&cfg, 0, sizeof(cfg)); cfg.btype = btype; for (auto& sp : cfg.literal_adaptation) sp = divans_speed{0x10, 0x2000}; }
                    have_cfg = true;
                }
                total += len; seen_literal = true;
                if (total > 0x7fffffffu) return -101;
                if (total > max_output) return -100;
            } else return -101;   // Copy / Dict / command- and distance- block switches
        }
        known = total;
        return 0;
    }
    // decodes what can be decoded now: whole chunks the commands cover and whose coded bytes are certainly complete, everything once
    // the container has ended
    int decode_chunks() {
        Mux::Stream& l = mux.s[1];
        const size_t chunk_bound = (size_t)divans_gpu_lit_encode_bound(32768);
        for (;;) {
            if (ready.size() - ready_pos > (1u << 20)) return 0;          // the caller first takes what is there
            uint32_t want = 0;
            if (mux.eof == 3 && cmd_done) want = (uint32_t)std::min<uint64_t>(known - done, 65536u);
            else if (known >= done + 32768u && l.avail() >= chunk_bound) want = (known >= done + 65536u && l.avail() >= 2 * chunk_bound) ? 65536u : 32768u;
            if (want == 0) return 0;
            if (!lit_started) {
                int rc = gpu.acquire(cfg, device, 65536u); if (rc) return rc;
                rc = divans_gpu_lit_stream_decode_begin(gpu.c); if (rc) return rc;
                lit_started = true;
            }
            if (ready_pos == ready.size()) { ready.clear(); ready_pos = 0; }
            const size_t at = ready.size();
            ready.resize(at + want);
            size_t used = 0;
            const size_t show = std::min<size_t>(l.avail(), (size_t)((want + 32767u) / 32768u) * chunk_bound);   // no chunk needs more than its bound
            const int rc = divans_gpu_lit_stream_decode(gpu.c, l.buf.data() + l.start, show, want, last8, ready.data() + at, &used);
            if (rc) return rc == DIVANS_GPU_ECORRUPT ? -100 : rc;
            if (used > l.avail()) return -100;
            l.start += used;
            for (size_t i = want < 8 ? 0 : want - 8; i < want; ++i) last8 = (last8 >> 8) | ((uint64_t)ready[at + i] << 56);
            done += want;
        }
    }
};

StreamDecoder::StreamDecoder(bool skip_crc, size_t max_output, int device) : p_(new Impl()) { p_->skip_crc = skip_crc; p_->max_output = max_output; p_->device = device; }
StreamDecoder::~StreamDecoder() { delete p_; }

int StreamDecoder::decode(const uint8_t* in, size_t n, size_t* in_off, uint8_t* out, size_t cap, size_t* out_off) {
    Impl& s = *p_;
    if (s.error) return s.error;
    auto fail_with = [&](int e) { s.error = e; return e; };
    for (;;) {
        // hand out what is decoded
        if (s.ready_pos < s.ready.size()) {
            const size_t k = std::min(s.ready.size() - s.ready_pos, cap - *out_off);
            std::memcpy(out + *out_off, s.ready.data() + s.ready_pos, k);
            s.ready_pos += k; *out_off += k;
            if (s.ready_pos < s.ready.size()) return 2;
        }
        if (s.trailer_ok && s.cmd_done && s.done == s.known) {
            if (s.mux.s[1].avail() != 0) return fail_with(-100);          // LIT bytes nobody asked for
            return 0;
        }
        // take input: header, Mux slices, trailer
        bool progressed = false;
        if (s.hdr_have < 16 && *in_off < n) {
            const size_t k = std::min<size_t>(16 - s.hdr_have, n - *in_off);
            std::memcpy(s.hdr + s.hdr_have, in + *in_off, k); s.crc = crc32c(s.crc, in + *in_off, k);
            s.hdr_have += (int)k; *in_off += k; progressed = true;
            const int h = s.hdr_have;     // magic / window are checkable as soon as they are in (divans_decompressor.rs:38-52)
            if ((h > 0 && s.hdr[0] != 0xff) || (h > 1 && s.hdr[1] != 0xe5) || (h > 2 && s.hdr[2] != 0x8c) || (h > 3 && s.hdr[3] != 0x9f) ||
                (h > 5 && (s.hdr[5] < 10 || s.hdr[5] >= 25))) return fail_with(-100);
        }
        if (s.hdr_have == 16 && s.mux.eof != 3 && *in_off < n) {
            const size_t k = s.mux.deserialize(in + *in_off, n - *in_off);
            s.crc = crc32c(s.crc, in + *in_off, k);
            *in_off += k; progressed = progressed || k != 0;
            if (k == 0 && s.mux.eof != 3) return fail_with(-100);          // bytes that are neither a slice nor the end marker
        }
        if (s.mux.eof == 3 && !s.trailer_ok && *in_off < n) {
            const size_t k = std::min<size_t>(8 - s.trailer_have, n - *in_off);
            std::memcpy(s.trailer + s.trailer_have, in + *in_off, k);
            s.trailer_have += (int)k; *in_off += k; progressed = true;
            if (s.trailer_have == 8) {                                     // codec/mod.rs:949-1017
                const uint8_t want[8] = {(uint8_t)s.crc, (uint8_t)(s.crc >> 8), (uint8_t)(s.crc >> 16), (uint8_t)(s.crc >> 24), 'a', 'n', 's', '~'};
                if (std::memcmp(s.trailer + 4, want + 4, 4) != 0 || (!s.skip_crc && std::memcmp(s.trailer, want, 4) != 0)) return fail_with(-100);
                s.trailer_ok = true;
                if (s.mux.s[1].avail() % 4) return fail_with(-100);
            }
        }
        if (int rc = s.read_commands()) return fail_with(rc);
        if (s.mux.eof == 3 && !s.cmd_done) return fail_with(-100);          // the container ended before its command stream did
        if (s.have_cfg || s.known == 0) {
            const uint64_t before = s.done;
            if (s.known) { if (int rc = s.decode_chunks()) return fail_with(rc); }
            progressed = progressed || s.done != before;
        }
        if (s.ready_pos < s.ready.size()) continue;
        if (s.trailer_ok && s.cmd_done && s.done == s.known) continue;
        if (!progressed) return 1;                                          // nothing more can be done with what has arrived
    }
}


}  // namespace divans_host

// Test hook (tests/test_abi_cpu.py): both CRC-32C paths on the same bytes.
extern "C" void divans_host_selftest_crc32c(const uint8_t* p, size_t n, uint32_t* accelerated, uint32_t* portable) {
    if (accelerated) *accelerated = divans_host::crc32c(0, p, n);
    if (portable) *portable = divans_host::crc32c_portable(0, p, n);
}
