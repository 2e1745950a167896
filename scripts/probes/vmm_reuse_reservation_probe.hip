// Follow-up to vmm_remap_probe.hip: is a RESERVATION THAT IS KEPT (no hipMemAddressFree / hipMemAddressReserve in between) safe to map to new
// chunks after its old ones were unmapped and released?  If it is, the tables' address ranges can be recycled instead of piling up.
//   hipcc --offload-arch=gfx950 -O2 -o vmm_reuse_reservation_probe vmm_reuse_reservation_probe.hip && ./vmm_reuse_reservation_probe [MiB] [chunk MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void rmw(uint32_t* p, size_t words, uint32_t tag, int rounds) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] = tag * 2654435761u + (uint32_t)i;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] += 1u;
}
__global__ void check(const uint32_t* p, size_t words, uint32_t tag, int rounds, unsigned long long* bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride)
        if (p[i] != tag * 2654435761u + (uint32_t)i + (uint32_t)rounds) atomicAdd(bad, 1ull);
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 4096) * (size_t)(1 << 20);
    const size_t chunk = (argc > 2 ? atol(argv[2]) : 32) * (size_t)(1 << 20);
    const size_t n = (bytes + chunk - 1) / chunk;
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc d; memset(&d, 0, sizeof(d)); d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    // a second buffer that stays alive and is written between the rounds: stale translations of the recycled range would land in it or miss their own pages
    uint32_t* other; CK(hipMalloc(&other, bytes));
    unsigned long long total_bad = 0;
    for (int round = 0; round < 6; ++round) {
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
        // map in a different order every round, as the library's shuffled mapping does
        for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, hs[(i * (2 * round + 1) + round) % n], 0));
        CK(hipMemSetAccess(va, n * chunk, &d, 1));
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(rmw, dim3(2048), dim3(256), 0, 0, (uint32_t*)va, bytes / 4, 100u + round, 3);
        hipLaunchKernelGGL(rmw, dim3(2048), dim3(256), 0, 0, other, bytes / 4, 200u + round, 2);
        hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, 0, (const uint32_t*)va, bytes / 4, 100u + round, 3, bad);
        hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, 0, (const uint32_t*)other, bytes / 4, 200u + round, 2, bad);
        unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("round %d: same reservation %p, %zu new chunks of %zu MiB: %llu bad words of %zu (range + the other buffer)\n", round, va, n, chunk >> 20, h, bytes / 2);
        total_bad += h;
        CK(hipDeviceSynchronize());
        for (size_t i = 0; i < n; ++i) CK(hipMemUnmap((char*)va + i * chunk, chunk));
        for (auto hnd : hs) CK(hipMemRelease(hnd));
    }
    printf("%s\n", total_bad ? "REUSING A KEPT RESERVATION IS NOT SAFE" : "a kept reservation can be mapped to new chunks again: no bad words in 6 rounds");
    return total_bad ? 1 : 0;
}
