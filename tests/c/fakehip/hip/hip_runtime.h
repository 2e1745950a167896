// TEST HARNESS: the handful of HIP runtime calls divans_amd/csrc/batch.cpp makes, on host memory and in program order (copies
// and "kernels" complete inside the call that enqueues them -- one of the orders a real device may produce).  It lets the many-
// containers interface (slices, lanes, persistent thread pool, container parsing / assembly under the "GPU work") run under
// AddressSanitizer / ThreadSanitizer with the oracle standing in for the literal kernels (hostsim_device_stub.cpp).  Only g++
// builds under tests/c see this directory; the product is built by hipcc against the real header.
#ifndef DIVANS_TESTS_FAKE_HIP_RUNTIME_H_
#define DIVANS_TESTS_FAKE_HIP_RUNTIME_H_
#include <cstddef>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct fake_hip_stream { int id; }* hipStream_t;
typedef struct fake_hip_event { int recorded; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
#define hipHostMallocDefault 0u
#define hipEventDisableTiming 2u

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "invalid value"; }
// FAKEHIP_DEVICES "devices" (environment, default 2, at most 8), the current one per thread as in the HIP runtime (the batch interface
// keeps one set of lanes per device and can shard one call over all of them)
static inline int fake_hip_device_count() { static const int n = [] { const char* e = std::getenv("FAKEHIP_DEVICES"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }(); return n; }
static inline int& fake_hip_current_device() { static thread_local int d = 0; return d; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = fake_hip_device_count(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= fake_hip_device_count()) return hipErrorInvalidValue; fake_hip_current_device() = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = fake_hip_current_device(); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)288 << 30; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new fake_hip_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new fake_hip_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->recorded = 1; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { return e->recorded ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(dst, src, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) { if (n) std::memset(dst, v, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
#endif
