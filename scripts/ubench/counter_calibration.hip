// Calibration of rocprofv3's memory-side counters (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ / _32B, TCC_EA0_WRREQ / _64B) on
// access patterns whose request and byte counts are known by construction (design aid, not product).
//
//   hipcc --offload-arch=gfx950 -O3 -o counter_calibration counter_calibration.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc --output-format csv -- ./counter_calibration     (one pass per counter set)
//
// Every kernel is launched exactly once; the program prints, per kernel, what it touched: bytes streamed, 128-byte lines,
// 64-byte halves and 32-byte rows.  scripts/ubench/calibration_table.py divides the counters of each dispatch by these.
//
//   stream_read16 / stream_read1   coalesced reads, 16 bytes / 1 byte per lane (bucket_sort_kernel reads its input like the latter)
//   stream_write16                 coalesced 16-byte-per-lane writes
//   row_read<MASK>                 N random 128-byte lines of a 4 GiB buffer; of each line the 32-byte rows in MASK are read the way
//                                  the decoders read a CDF row: 16 lanes x 2 bytes through a buffer descriptor.  MASK 1 = one row,
//                                  3 = both rows of one 64-byte half, 5 = one row in either half, 15 = all four.  If a miss fills the
//                                  whole 128-byte line the request count per line is 1 whatever MASK is; if it fills a 64-byte half,
//                                  MASK 5 costs two; if 32-byte sectors, MASK 3 costs two as well.
//   row_write<MASK>                the same with 2-byte-per-lane row stores
//   row_rmw                        load, add, store of one random row (the decoders' low-nibble row access)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read16(const v4u* in, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const v4u v = __builtin_nontemporal_load(in + i); acc += v.x ^ v.w; }
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ __launch_bounds__(256) void stream_read1(const uint8_t* in, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc == 0xdeadbeefu) *sink = acc;
}
__global__ __launch_bounds__(256) void stream_write16(v4u* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const v4u v = {(uint32_t)i, 1u, 2u, 3u}; out[i] = v; }
}

// one 16-lane group per random line sequence; the line index comes from a 64-bit LCG so that no line repeats soon
__device__ __forceinline__ uint64_t lcg(uint64_t x) { return x * 6364136223846793005ull + 1442695040888963407ull; }

template <int MASK, int OP>   // OP 0 read, 1 write, 2 read-modify-write
__global__ __launch_bounds__(256) void rows_kernel(uint8_t* base, uint64_t lines, uint32_t iters, uint32_t* sink) {
    const uint32_t j = threadIdx.x & 15u;
    const uint32_t g = (blockIdx.x * 256u + threadIdx.x) >> 4;
    uint64_t x = (uint64_t)g * 0x9E3779B97F4A7C15ull + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        x = lcg(x);
        const uint64_t line = (x >> 20) % lines;
        uint8_t* p = base + line * 128u + 2u * j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!(MASK & (1 << r))) continue;
            uint16_t* q = (uint16_t*)(p + 32 * r);
            if (OP == 0) acc += *q;
            else if (OP == 1) *q = (uint16_t)(i + j);
            else { const uint16_t v = *q; *q = (uint16_t)(v + 1u); }
        }
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}

int main() {
    const size_t BYTES = 4ull << 30;
    uint8_t* buf; CK(hipMalloc(&buf, BYTES)); CK(hipMemset(buf, 1, BYTES));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    CK(hipDeviceSynchronize());
    const size_t stream_bytes = 2ull << 30;
    printf("kernel,bytes_streamed,lines_128B,halves_64B,rows_32B\n");
    stream_read16<<<4096, 256>>>((const v4u*)buf, stream_bytes / 16, sink);
    printf("stream_read16,%zu,%zu,%zu,%zu\n", stream_bytes, stream_bytes / 128, stream_bytes / 64, stream_bytes / 32);
    stream_read1<<<4096, 256>>>(buf + stream_bytes, stream_bytes / 2, sink);
    printf("stream_read1,%zu,%zu,%zu,%zu\n", stream_bytes / 2, stream_bytes / 256, stream_bytes / 128, stream_bytes / 64);
    stream_write16<<<4096, 256>>>((v4u*)buf, stream_bytes / 16);
    printf("stream_write16,%zu,%zu,%zu,%zu\n", stream_bytes, stream_bytes / 128, stream_bytes / 64, stream_bytes / 32);
    CK(hipDeviceSynchronize());
    const uint32_t blocks = 256 * 7, iters = 1024;       // the decoders' grid: 28 672 groups of 16 lanes
    const uint64_t groups = (uint64_t)blocks * 16, n = groups * iters;
    const uint64_t lines = BYTES / 128;
#define ROWS(MASK, OP, NAME, H, R) rows_kernel<MASK, OP><<<blocks, 256>>>(buf, lines, iters, sink); printf(NAME ",0,%llu,%llu,%llu\n", (unsigned long long)n, (unsigned long long)(n * H), (unsigned long long)(n * R)); CK(hipDeviceSynchronize());
    ROWS(1, 0, "row_read_mask1", 1, 1)
    ROWS(3, 0, "row_read_mask3", 1, 2)
    ROWS(5, 0, "row_read_mask5", 2, 2)
    ROWS(15, 0, "row_read_mask15", 2, 4)
    ROWS(1, 1, "row_write_mask1", 1, 1)
    ROWS(3, 1, "row_write_mask3", 1, 2)
    ROWS(5, 1, "row_write_mask5", 2, 2)
    ROWS(15, 1, "row_write_mask15", 2, 4)
    ROWS(1, 2, "row_rmw_mask1", 1, 1)
    CK(hipDeviceSynchronize());
    return 0;
}
