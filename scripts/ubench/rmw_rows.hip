// Micro-benchmark (design aid, not product): rate of random table-row accesses through L2 / MALL / HBM as a function of
// how a row is accessed (lanes x bytes per lane), row size, read / write / read-modify-write, footprint per stream,
// resident waves and number of CUs.  Each group of LANES lanes owns a private slab (like one stream's CDF table); one
// SRD per workgroup (uniform), per-lane offsets -- the access form of lit_decode_kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -o rmw_rows rmw_rows.hip && ./rmw_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
constexpr uint32_t SLAB = 139264u;   // bytes per group: 4352 rows x 32 B (BASELINE configs[1] table)

template <int BPL> struct Acc;
template <> struct Acc<2> {
    static __device__ uint32_t ld(__amdgpu_buffer_rsrc_t r, uint32_t o) { return __builtin_amdgcn_raw_buffer_load_b16(r, o, 0, 0); }
    static __device__ void st(__amdgpu_buffer_rsrc_t r, uint32_t o, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, r, o, 0, 0); }
};
template <> struct Acc<4> {
    static __device__ uint32_t ld(__amdgpu_buffer_rsrc_t r, uint32_t o) { return __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0); }
    static __device__ void st(__amdgpu_buffer_rsrc_t r, uint32_t o, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, o, 0, 0); }
};
template <> struct Acc<8> {
    static __device__ uint32_t ld(__amdgpu_buffer_rsrc_t r, uint32_t o) { v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0); return v.x + v.y; }
    static __device__ void st(__amdgpu_buffer_rsrc_t r, uint32_t o, uint32_t v) { v2u w = {v, v + 1}; __builtin_amdgcn_raw_buffer_store_b64(w, r, o, 0, 0); }
};
template <> struct Acc<16> {
    static __device__ uint32_t ld(__amdgpu_buffer_rsrc_t r, uint32_t o) { v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0); return v.x + v.w; }
    static __device__ void st(__amdgpu_buffer_rsrc_t r, uint32_t o, uint32_t v) { v4u w = {v, v + 1, v + 2, v + 3}; __builtin_amdgcn_raw_buffer_store_b128(w, r, o, 0, 0); }
};

// OP: 0 = read-modify-write (dependent), 1 = read only (dependent address), 2 = write only
template <int LANES, int BPL, int OP>
__global__ __launch_bounds__(256) void rows_kernel(uint8_t* base, uint32_t hot_rows, uint32_t iters, uint32_t* sink) {
    constexpr uint32_t ROWB = LANES * BPL, GPB = 256 / LANES, NROWS = SLAB / ROWB;
    const uint32_t lg = threadIdx.x / LANES, j = threadIdx.x % LANES;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)blockIdx.x * GPB * SLAB, 0, GPB * SLAB, 0x00020000);
    const uint32_t g = blockIdx.x * GPB + lg;
    uint32_t x = g * 2654435761u + 12345u, acc = 0;
    const uint32_t lane_off = lg * SLAB + j * BPL;
    for (uint32_t i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t r = ((x >> 8) + (OP != 2 ? (acc & 1u) : 0u)) % hot_rows;
        r = (r * 37u) % NROWS;   // spread the hot rows over the slab
        const uint32_t off = lane_off + r * ROWB;
        uint32_t v = 0;
        if (OP != 2) { v = Acc<BPL>::ld(rsrc, off); acc += v; }
        if (OP != 1) Acc<BPL>::st(rsrc, off, v + i);
    }
    if (acc == 0xdeadbeef) *sink = acc;
}

typedef void (*K)(uint8_t*, uint32_t, uint32_t, uint32_t*);
struct V { const char* name; K k; int lanes; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    V vs[] = {
        {"32B row 16x2B rmw", rows_kernel<16, 2, 0>, 16}, {"32B row 16x2B rd ", rows_kernel<16, 2, 1>, 16}, {"32B row 16x2B wr ", rows_kernel<16, 2, 2>, 16},
        {"32B row  8x4B rmw", rows_kernel<8, 4, 0>, 8},   {"32B row  2x16B rmw", rows_kernel<2, 16, 0>, 2}, {"32B row  2x16B rd ", rows_kernel<2, 16, 1>, 2},
        {"64B row 16x4B rmw", rows_kernel<16, 4, 0>, 16}, {"64B row 16x4B rd ", rows_kernel<16, 4, 1>, 16}, {"64B row 16x4B wr ", rows_kernel<16, 4, 2>, 16},
        {"128B row 16x8B rmw", rows_kernel<16, 8, 0>, 16}, {"128B row 16x8B rd ", rows_kernel<16, 8, 1>, 16}, {"128B row 16x8B wr ", rows_kernel<16, 8, 2>, 16},
        {"128B row 8x16B rmw", rows_kernel<8, 16, 0>, 8},
    };
    const int max_blocks = ncu * 8;
    // the LANES = 2 variants have 128 groups per block: cap the buffer by the largest group count used
    size_t max_groups = (size_t)max_blocks * 128;
    uint8_t* buf; CK(hipMalloc(&buf, max_groups * SLAB)); CK(hipMemset(buf, 0, max_groups * SLAB));
    for (int cus : {ncu, 32}) {
        for (int wg_per_cu : {2, 4, 8}) {
            if (cus != ncu && wg_per_cu != 4) continue;
            for (auto& v : vs) {
                const uint32_t gpb = 256 / v.lanes;
                int blocks = cus * wg_per_cu;
                if ((size_t)blocks * gpb > max_groups) blocks = (int)(max_groups / gpb);   // LANES = 2 / 8 variants: fewer blocks, same buffer
                for (uint32_t hot : {1u, 4u, 16u, 64u, 256u, 1024u}) {
                    if (wg_per_cu != 4 && hot != 16u && hot != 256u) continue;
                    const uint32_t iters = 2048;
                    v.k<<<blocks, 256>>>(buf, hot, 128, sink);
                    CK(hipEventRecord(e0));
                    v.k<<<blocks, 256>>>(buf, hot, iters, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double groups = (double)blocks * gpb, ops = groups * iters;
                    printf("CUs %3d wg/cu %d blocks %5d groups %7.0f hot_rows %4u (%8.1f MB hot) %s: %7.2f G rows/s  %6.0f ns per step\n", cus, wg_per_cu, blocks,
                           groups, hot, groups * hot * (256.0 / v.lanes == gpb ? 1 : 1) * (v.lanes == 16 && v.name[0] == '6' ? 64 : (v.name[0] == '1' ? 128 : 32)) / 1e6,
                           v.name, ops / ms / 1e6, ms * 1e6 / iters);
                }
            }
        }
    }
    return 0;
}
