// Does a device range mapped with hipMemMap right after another range was unmapped see its own pages?  (The decode kernels produced
// wrong bytes on CDF tables allocated that way during divans_gpu_codec_tune_tables.)   hipcc --offload-arch=gfx950 -O2 -o vmm_remap_probe vmm_remap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Range { void* va = nullptr; size_t chunk = 0, n = 0; std::vector<hipMemGenericAllocationHandle_t> hs; };

static Range make(size_t bytes, size_t chunk, void* hint) {
    Range r;
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    r.chunk = chunk; r.n = (bytes + chunk - 1) / chunk;
    CK(hipMemAddressReserve(&r.va, r.n * chunk, 0, hint, 0));
    for (size_t i = 0; i < r.n; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); r.hs.push_back(h); }
    for (size_t i = 0; i < r.n; ++i) CK(hipMemMap((char*)r.va + i * chunk, chunk, 0, r.hs[(i * 7919) % r.n == i ? i : i], 0));
    hipMemAccessDesc d; memset(&d, 0, sizeof(d)); d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(r.va, r.n * chunk, &d, 1));
    return r;
}
static void drop(Range& r) {
    for (size_t i = 0; i < r.n; ++i) CK(hipMemUnmap((char*)r.va + i * r.chunk, r.chunk));
    for (auto h : r.hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(r.va, r.n * r.chunk));
    r.va = nullptr;
}

// every thread owns a strided set of 32-byte rows: write a value derived from (tag, row), then read-modify-write it `rounds` times
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
    const size_t chunk = 2u << 20;
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    auto run = [&](Range& r, uint32_t tag, const char* what) {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(rmw, dim3(2048), dim3(256), 0, 0, (uint32_t*)r.va, bytes / 4, tag, 3);
        hipLaunchKernelGGL(check, dim3(2048), dim3(256), 0, 0, (const uint32_t*)r.va, bytes / 4, tag, 3, bad);
        unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("%-60s va %p: %llu bad words of %zu\n", what, r.va, h, bytes / 4);
    };
    Range a = make(bytes, chunk, nullptr); run(a, 1, "A (first range)");
    void* va_a = a.va;
    Range b = make(bytes, chunk, nullptr); run(b, 2, "B (second range, A still mapped)");
    drop(a);
    Range c = make(bytes, chunk, nullptr); run(c, 3, c.va == va_a ? "C right after A was dropped (SAME address as A)" : "C right after A was dropped (other address)");
    run(b, 4, "B again");
    drop(c);
    Range d = make(bytes, chunk, va_a); run(d, 5, d.va == va_a ? "D at A's address (hint honoured)" : "D (hint not honoured)");
    for (int k = 0; k < 4; ++k) { drop(d); d = make(bytes, chunk, nullptr); char nm[64]; snprintf(nm, sizeof nm, "drop + make, round %d", k); run(d, 6 + k, nm); }
    drop(b); drop(d);
    return 0;
}
