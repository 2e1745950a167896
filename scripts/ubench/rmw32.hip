// Micro-benchmark (design aid): throughput of random 32-byte row read-modify-writes, 8 lanes x 4 B per row,
// as a function of the footprint and of the cache policy bits.  Each 8-lane group owns a slab (like a stream's table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int AUX_LD, int AUX_ST, bool DEP>
__global__ __launch_bounds__(256) void rmw_kernel(uint8_t* base, uint32_t slab_rows, uint32_t hot_rows, uint32_t iters, uint32_t* sink) {
    const uint32_t g = (blockIdx.x * 256 + threadIdx.x) >> 3, j = threadIdx.x & 7;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)g * slab_rows * 32u, 0, slab_rows * 32u, 0x00020000);
    uint32_t x = g * 2654435761u + 12345u, acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t r = ((x >> 8) + (DEP ? (acc & 1u) : 0u)) % hot_rows;
        r = (r * 37u) % slab_rows;            // spread the hot rows over the slab
        uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, r * 32u + j * 4u, 0, AUX_LD);
        acc += v;
        __builtin_amdgcn_raw_buffer_store_b32(v + 1u, rsrc, r * 32u + j * 4u, 0, AUX_ST);
    }
    if (acc == 0xdeadbeef) *sink = acc;
}

typedef void (*K)(uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t*);
int main() {
    int ncu = 256;
    const uint32_t slab_rows = 4352;
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; K k; } variants[] = {
        {"dep default      ", rmw_kernel<0, 0, true>},
        {"dep ld nt        ", rmw_kernel<2, 0, true>},
        {"dep ld/st nt     ", rmw_kernel<2, 2, true>},
        {"dep ld sc0sc1    ", rmw_kernel<17, 0, true>},
        {"dep ld/st sc0sc1 ", rmw_kernel<17, 17, true>},
        {"dep ld/st sc1    ", rmw_kernel<16, 16, true>},
        {"indep default    ", rmw_kernel<0, 0, false>},
    };
    for (int wg_per_cu : {4, 8}) {
        const uint32_t groups = ncu * wg_per_cu * 32;
        uint8_t* buf; CK(hipMalloc(&buf, (size_t)groups * slab_rows * 32));
        CK(hipMemset(buf, 0, (size_t)groups * slab_rows * 32));
        for (uint32_t hot : {16u, 64u, 128u, 256u, 1024u, 4352u}) {
            for (auto& v : variants) {
                const uint32_t iters = 4096;
                v.k<<<ncu * wg_per_cu, 256>>>(buf, slab_rows, hot, 256, sink);
                CK(hipEventRecord(e0));
                v.k<<<ncu * wg_per_cu, 256>>>(buf, slab_rows, hot, iters, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double rmw = (double)groups * iters;
                printf("wg/cu %d groups %u hot_rows %4u (%6.1f MB hot) %s: %7.2f G rmw/s  %6.0f ns per dependent step\n", wg_per_cu, groups, hot,
                       groups * (double)hot * 32 / 1e6, v.name, rmw / ms / 1e6, ms * 1e6 / iters);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
