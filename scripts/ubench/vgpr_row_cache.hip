// Micro-benchmark (design aid, not product): what a row cache held in the VECTOR REGISTER FILE costs per access
// against the same cache in LDS, in the decoders' mapping (16 lanes = one stream, lane i holds cdf[i], four streams per wave64).
//
// A cached row is one element of a 16-dword register vector (two u16 rows per dword: 32 rows); the slot index is uniform over a
// stream's 16 lanes but differs between the four streams of a wave, and gfx950 indexes registers with a SCALAR index
// (s_set_gpr_idx_on + v_mov), so every access is a four-step waterfall: v_readlane of each stream's index, one indexed move
// per stream, one select per stream.  The LDS form is what lit_decode2.hip does today: ds_read_u16 / ds_write_b16 at a per-lane
// address.  Both loops carry the dependency the decoder has (the next slot depends on the value just read).
//
//   hipcc --offload-arch=gfx950 -O3 -o vgpr_row_cache vgpr_row_cache.hip && ./vgpr_row_cache
// Output: ns and SIMD cycles per read-modify-write step for 1..8 waves per SIMD, VGPR form (packed halves, read + write),
// VGPR read only, LDS form, and the arithmetic both loops share (the baseline to subtract).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 4096;

__device__ __forceinline__ uint32_t vg_read(const u32x16& v, uint32_t idx, int row) {
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sj = __builtin_amdgcn_readlane(idx, 16 * j) & 15u;
        const uint32_t x = v[sj];
        out = row == j ? x : out;
    }
    return out;
}
__device__ __forceinline__ void vg_write(u32x16& v, uint32_t idx, int row, uint32_t val) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t sj = __builtin_amdgcn_readlane(idx, 16 * j) & 15u;
        uint32_t t = v[sj];
        t = row == j ? val : t;
        v[sj] = t;
    }
}

// MODE 0: arithmetic only; 1: VGPR read + write (packed halves); 2: VGPR read only; 3: LDS read + write
template <int MODE>
__global__ __launch_bounds__(256) void k_cache(uint32_t* out, uint32_t seed) {
    __shared__ uint16_t lds[16 * 32 * 16];       // 16 streams per workgroup x 32 rows x 16 entries
    u32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * (uint32_t)(i + 1) + seed;
    for (int i = threadIdx.x; i < 16 * 32 * 16; i += 256) lds[i] = (uint16_t)(i + seed);
    __syncthreads();
    const int row = (threadIdx.x >> 4) & 3, li = threadIdx.x & 15;
    uint16_t* mine = lds + (threadIdx.x >> 4) * 32 * 16 + li;
    uint32_t slot = (threadIdx.x >> 4) & 31u, acc = seed;
    for (int i = 0; i < ITERS; ++i) {
        uint32_t x;
        if (MODE == 1 || MODE == 2) {
            const uint32_t d = vg_read(v, slot & 15u, row);
            const bool hi = (slot & 16u) != 0u;
            x = hi ? d >> 16 : d & 0xffffu;
            if (MODE == 1) {
                const uint32_t nv = (x + 16u) & 0xffffu;
                vg_write(v, slot & 15u, row, hi ? (d & 0xffffu) | (nv << 16) : (d & 0xffff0000u) | nv);
            }
        } else if (MODE == 3) {
            x = mine[slot * 16u];
            mine[slot * 16u] = (uint16_t)(x + 16u);
        } else {
            x = acc * 3u;
        }
        acc += x;
        // the next slot is uniform over the stream's 16 lanes and depends on what was read (lane 15's value, like cdf[15])
        const uint32_t top = (uint32_t)__builtin_amdgcn_mov_dpp((int)acc, 0x15f, 0xf, 0xf, false);
        slot = (slot * 5u + 1u + (top & 3u)) & 31u;
    }
    uint32_t s = acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= v[i];
    if (s == 0xdeadbeefu) out[0] = s;
}

typedef void (*K)(uint32_t*, uint32_t);
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    uint32_t* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; K k; } vs[] = {{"arithmetic only", k_cache<0>}, {"VGPR cache read+write", k_cache<1>}, {"VGPR cache read only", k_cache<2>}, {"LDS cache read+write", k_cache<3>}};
    printf("CUs %d, clock %.2f GHz; per step = one row read-modify-write of each of a wave's four streams\n", ncu, ghz);
    for (int wps = 1; wps <= 8; ++wps) {
        for (auto& v : vs) {
            v.k<<<ncu * wps, 256>>>(out, 1u);
            CK(hipEventRecord(e0));
            v.k<<<ncu * wps, 256>>>(out, 2u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double ns_step = ms * 1e6 / ITERS;     // chain latency of one wave's step when wps waves share the SIMD
            printf("waves/SIMD %d  %-24s %8.1f ns per step of a wave = %7.1f cycles; SIMD cycles per wave-step %7.1f\n", wps, v.name, ns_step, ns_step * ghz, ns_step * ghz / wps);
        }
    }
    return 0;
}
