// lit_bucket_dev.h -- device helpers shared by the bucketed encoder passes (lit_bucket.hip, lit_bucket_mix.hip).
#ifndef DIVANS_LIT_BUCKET_DEV_H_
#define DIVANS_LIT_BUCKET_DEV_H_
#include "lit_device.h"

namespace divans_hip {

constexpr uint32_t BK_PIECE = 8192;          // positions sorted together
constexpr uint32_t BK_SORT_THREADS = 256;
constexpr uint32_t BK_LANE_DWORDS = 148;     // 17 rows x 8 dwords + 8 descriptors, padded: 16-byte aligned and the
                                             // 64 lanes' b128 accesses at equal offsets cover all 32 banks
constexpr uint32_t BK_DESC_DW = 136;
constexpr uint32_t BK_TAB_DW = 64 * BK_LANE_DWORDS;
constexpr uint32_t BK_VALID = 1u << 31;
constexpr uint32_t BK_WINDOW = 256;          // tasks a wave reserves per atomic
constexpr uint32_t BK_CLASSES = 6;           // task lists by bucket size, longest first (a bucket is a serial chain: the long ones must start early)
constexpr uint32_t BK_CLAIM = 8;             // counters[0..5] = tasks per class, counters[BK_CLAIM] = next unclaimed task

__device__ __forceinline__ int bk_class_of(uint32_t tot) {
    return tot == 0u ? -1 : (tot >= 16384u ? 0 : (tot >= 8192u ? 1 : (tot >= 4096u ? 2 : (tot >= 2048u ? 3 : (tot >= 64u ? 4 : 5)))));
}
// the t-th task overall: class lists are [BK_CLASSES][cap], `ends` their cumulative sizes
struct BkTaskLists {
    uint32_t ends[BK_CLASSES];
    __device__ __forceinline__ void load(const uint32_t* counters) {
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t c = 0; c < BK_CLASSES; ++c) { acc += counters[c]; ends[c] = acc; }
    }
    __device__ __forceinline__ uint32_t total() const { return ends[BK_CLASSES - 1u]; }
    __device__ __forceinline__ const uint32_t* at(const uint32_t* tasks, uint32_t cap, uint32_t t) const {
        uint32_t cls = 0, base = 0;
#pragma unroll
        for (uint32_t c = 0; c + 1u < BK_CLASSES; ++c) if (t >= ends[c]) { cls = c + 1u; base = ends[c]; }
        return tasks + (size_t)cls * cap + (t - base);
    }
};

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// One nibble against a row of 8 dwords (16 x u16) in LDS, split into phases so that the two nibbles of a byte
// (different rows) can be in flight together: read, pack (start | freq << 16), blend, write.
struct BkRow { u32x4 w0, w1, a0, a1; int chi, cprev; };

__device__ __forceinline__ BkRow bk_read(const uint32_t* row, const uint32_t* tab, uint32_t sym) {
    BkRow r;
    r.w0 = *(const u32x4*)row; r.w1 = *(const u32x4*)(row + 4);
    const uint16_t* r16 = (const uint16_t*)row;
    r.chi = r16[sym];
    r.cprev = r16[sym ? sym - 1u : 0u];
    r.a0 = *(const u32x4*)(tab + sym * 8u); r.a1 = *(const u32x4*)(tab + sym * 8u + 4u);
    return r;
}
__device__ __forceinline__ uint32_t bk_pack(const BkRow& r, uint32_t sym) {     // probability/interface.rs:97-108
    const int mx = (int)(r.w1.w >> 16);
    const int clo = sym ? r.cprev : 0;
    const float rcp = biased_rcp15(mx);
    const uint32_t dhi = scaled_div(r.chi, mx, rcp), dlo = scaled_div(clo, mx, rcp);
    return (dlo + 1u) | ((dhi - dlo - 1u) << 16);
}
__device__ __forceinline__ void bk_renorm(BkRow& r) {                           // frequentist_cdf.rs:79-84, both halves at once
    const u32x4 b0 = {1u | (2u << 16), 3u | (4u << 16), 5u | (6u << 16), 7u | (8u << 16)};
    const u32x4 b1 = {9u | (10u << 16), 11u | (12u << 16), 13u | (14u << 16), 15u | (16u << 16)};
    const u32x4 t0 = r.w0 + b0, t1 = r.w1 + b1;
    r.w0 = t0 - ((t0 >> 2) & 0x3fff3fffu);
    r.w1 = t1 - ((t1 >> 2) & 0x3fff3fffu);
}

__device__ __forceinline__ void bk_store_quad(u32x4* p, u32x4 v) {     // 8-byte aligned is enough for a global store
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bk_store_pair(u32x2* p, u32x2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void bk_store_word(uint32_t* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

}  // namespace divans_hip
#endif
