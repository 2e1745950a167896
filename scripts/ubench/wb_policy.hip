// wb_policy.hip -- does the L2 of gfx950 hold a dirty row until its line is evicted (write back), or does every row store leave
// for the fabric (write through)?  tests/tools/sim_hierarchy.c replays the decoders' row traffic against a write-back L2 and
// reproduces their fills (TCC_EA0_RDREQ) but only 70 % of their TCC_EA0_WRREQ; this program settles which model is right
// (design aid, not product).
//
// rmw_private<K>: each of 28 672 16-lane groups (the decoders' grid) owns K consecutive 32-byte rows and read-modify-writes them
// round robin, `iters` accesses in all, the decoders' way (16 lanes x 2 bytes through plain global accesses).  Footprint =
// groups x K x 32 B:  K = 1 -> 0.9 MB, 4 -> 3.7 MB, 32 -> 29 MB (the eight L2s hold 32 MB), 256 -> 235 MB (Infinity Cache), 1024 -> 940 MB.
//   write back   => WRREQ ~ footprint / 64 B while the footprint fits the L2, whatever `iters`
//   write through => WRREQ ~ groups x iters
// The program prints the known counts (stores, footprint lines) per kernel; the PMC passes are in scripts/r06_wb_policy.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int K>
__global__ __launch_bounds__(256) void rmw_private(uint8_t* base, uint32_t iters, uint32_t* sink) {
    const uint32_t j = threadIdx.x & 15u;
    const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 4;
    uint8_t* mine = base + (size_t)g * K * 32u + 2u * j;
    uint32_t acc = 0, r = g % K;
    for (uint32_t i = 0; i < iters; ++i) {
        uint16_t* q = (uint16_t*)(mine + 32u * r);
        const uint16_t v = *q;
        *q = (uint16_t)(v + 1u);
        acc += v;
        r = (r * 5u + 1u) % K;     // a full-period walk over the K rows when K is a power of two
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}

int main() {
    const size_t BYTES = 1ull << 30;
    uint8_t* buf; CK(hipMalloc(&buf, BYTES)); CK(hipMemset(buf, 1, BYTES));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    CK(hipDeviceSynchronize());
    const uint32_t blocks = 256 * 7, iters = 2048;
    const unsigned long long groups = (unsigned long long)blocks * 16, n = groups * iters;
    printf("kernel,bytes_streamed,lines_128B,halves_64B,rows_32B\n");
#define RUN(K) rmw_private<K><<<blocks, 256>>>(buf, iters, sink); CK(hipDeviceSynchronize()); \
    printf("rmw_private_%d,0,%llu,%llu,%llu\n", K, (groups * K * 32ull + 127ull) / 128ull, (groups * K * 32ull + 63ull) / 64ull, n);
    RUN(1) RUN(4) RUN(32) RUN(256) RUN(1024)
    return 0;
}
