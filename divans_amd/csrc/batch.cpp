// batch.cpp -- many complete .divans streams per call (include/divans_batch.h): the LIT coder of every stream on the GPU,
// the CMD coders and the framing on host threads, the two overlapped (the reference overlaps
// the same halves of ONE stream with a worker thread, src/parallel_decompressor.rs:55-141, src/threading.rs:88-100).
//
// Both directions run as a pipeline of SLICES over a few lanes (lane = HIP stream + codecs + page-locked staging buffers):
// while the GPU codes the slices that are in flight, the host threads stage the next one and assemble / copy out the one
// that has just finished.  Streams are binned into LENGTH CLASSES first (<= 64 KiB, then powers of two): device memory is
// sized per slice from the class bound, so one long stream among many short ones costs its own slice and nothing else, and
// the <= 64 KiB class keeps the bucketed encoder passes.
#include "../../include/divans_batch.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "host_stream.h"

namespace divans_host { int set_last_error(int code, const std::string& msg); }
using divans_host::set_last_error;

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

divans_host::StreamOptions to_stream_options(const divans_batch_options& o) {
    divans_host::StreamOptions so;
    so.window_size = o.window_size; so.dynamic_context_mixing = o.dynamic_context_mixing; so.use_context_map = o.use_context_map != 0;
    so.force_stride = o.force_stride; so.has_prior_depth = o.has_prior_depth != 0; so.prior_depth = o.prior_depth;
    so.has_literal_adaptation = o.has_literal_adaptation != 0;
    for (int i = 0; i < 4; ++i) so.literal_adaptation[i] = o.literal_adaptation[i];
    so.use_brotli = 0;
    return so;
}

// How many threads the host half may use: what the process is actually granted -- CPU affinity and the cgroup's CPU quota --
// not the machine's logical CPU count (a container that shows 256 CPUs and grants 16 would oversubscribe 16-fold).
int usable_threads() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0}; long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
            const long q = std::atol(quota);
            if (q > 0) n = std::min<long>(n, std::max<long>(1, (q + period - 1) / period));
        }
        std::fclose(f);
    }
    return std::min(n, 64);
}

// A persistent pool: the calls below run dozens of short parallel sections per call (per slice: staging, assembly, copy-out),
// and creating the threads each time costs more than the sections themselves.  The pool is the process's cores, shared by every
// device: SEVERAL sections may be open at once (one per calling thread -- the device threads of a call that names all devices, or a
// caller's own threads on different devices), each with its own width, i.e. its budget of threads; an idle worker joins whichever
// open section still has items and fewer helpers than its width.  (Until round 6 a section held the whole pool: eight device
// threads took turns for their host halves.)
class ThreadPool {
public:
    explicit ThreadPool(int workers) { for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); }); }
    ~ThreadPool() {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int workers() const { return (int)threads_.size(); }
    // body(i) for i in [0, n) on at most `width` threads (the caller is one of them); returns when all are done
    void run(size_t n, int width, const std::function<void(size_t)>& body) {
        if (n == 0) return;
        const int helpers = (int)std::min<size_t>((size_t)std::max(0, std::min(width, workers() + 1) - 1), n - 1);
        if (helpers == 0) { for (size_t i = 0; i < n; ++i) body(i); return; }
        Section sec; sec.body = &body; sec.n = n; sec.want = helpers;
        { std::lock_guard<std::mutex> l(mu_); open_.push_back(&sec); }
        if (helpers == 1) cv_.notify_one(); else cv_.notify_all();
        for (size_t i = sec.next.fetch_add(1); i < n; i = sec.next.fetch_add(1)) body(i);
        std::unique_lock<std::mutex> l(mu_);
        open_.erase(std::find(open_.begin(), open_.end(), &sec));      // no worker joins from here on
        done_cv_.wait(l, [&] { return sec.active == 0; });              // the ones that did have left `sec` (it lives on this stack frame)
    }
private:
    struct Section { const std::function<void(size_t)>* body = nullptr; size_t n = 0; std::atomic<size_t> next{0}; int want = 0, joined = 0, active = 0; };
    Section* pick() {          // mu_ held: an open section with items left and room for another helper
        for (Section* s : open_) if (s->joined < s->want && s->next.load(std::memory_order_relaxed) < s->n) return s;
        return nullptr;
    }
    void loop() {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            Section* sec = nullptr;
            cv_.wait(l, [&] { return stop_ || (sec = pick()) != nullptr; });
            if (stop_) return;
            sec->joined += 1; sec->active += 1;
            l.unlock();
            for (size_t i = sec->next.fetch_add(1); i < sec->n; i = sec->next.fetch_add(1)) (*sec->body)(i);
            l.lock();
            sec->active -= 1;
            if (sec->active == 0) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_; std::condition_variable cv_, done_cv_;
    std::vector<Section*> open_;
    bool stop_ = false;
};
ThreadPool& thread_pool() { static ThreadPool p(usable_threads() - 1); return p; }

template <typename F>
void parallel_for(size_t n, int threads, F&& body) {
    const int width = threads > 0 ? threads : usable_threads();
    const std::function<void(size_t)> fn = std::forward<F>(body);
    thread_pool().run(n, width, fn);
}

// grow-only buffers: a lane reuses them from slice to slice
struct DeviceBuf {
    void* p = nullptr; size_t cap = 0;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() const { return (T*)p; }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};
struct PinnedBuf {
    void* p = nullptr; size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    template <typename T> T* as() const { return (T*)p; }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 4, 256);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
};

#define HIP_OR_FAIL(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_last_error(DIVANS_GPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// ---- length classes and slices ---------------------------------------------------------------------------------------
// class bound of a stream length: 65536 for everything up to 64 KiB (one codec with the bucketed encoder passes), then the
// next power of two
// A copy whose destination will not be read again by this thread (decoded payloads into the caller's buffer, coded bytes into
// page-locked staging memory): streaming stores spare the read-for-ownership of every destination line -- a third of the memory
// traffic of a plain memcpy for blocks far larger than any cache share.
void stream_copy(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__SSE2__)
    if (n >= 4096) {
        const size_t head = (size_t)(-(intptr_t)dst) & 63u;
        std::memcpy(dst, src, head); dst += head; src += head; n -= head;
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; ++i) {
            const __m128i a = _mm_loadu_si128((const __m128i*)(src) + 0), b = _mm_loadu_si128((const __m128i*)(src) + 1);
            const __m128i c = _mm_loadu_si128((const __m128i*)(src) + 2), d = _mm_loadu_si128((const __m128i*)(src) + 3);
            _mm_stream_si128((__m128i*)(dst) + 0, a); _mm_stream_si128((__m128i*)(dst) + 1, b);
            _mm_stream_si128((__m128i*)(dst) + 2, c); _mm_stream_si128((__m128i*)(dst) + 3, d);
            src += 64; dst += 64;
        }
        _mm_sfence();
        n -= blocks * 64;
    }
#endif
    std::memcpy(dst, src, n);
}

uint32_t class_bound(size_t len) {
    uint32_t b = 65536u;
    while ((size_t)b < len) b <<= 1;
    return b;
}
struct Slice { uint32_t bound = 0; std::vector<size_t> members; size_t bytes = 0; };

// device bytes one stream of a class costs the codec and this file (slots, packed copy, start/freq spill, bucket arrays of the
// two-model pass: DESIGN.md section 2), rounded up
size_t device_bytes_per_stream(uint32_t bound) { return (size_t)bound * 64u + (1u << 16); }

constexpr int kLanes = 8;

// Compression: the encoder passes are organised by throughput (sort / bucket chains / unsort / rANS fill the GPU whatever the
// slice size, so concurrent slices would only queue behind each other): a class is cut into TWO slices (at least 256, at most
// 16384 streams each, never more than the device budget allows) -- the host assembles the first while the GPU codes the second.
void make_slices(const std::vector<size_t>& order, const std::vector<uint32_t>& bound_of, const size_t* sizes, size_t budget_bytes,
                 std::vector<Slice>& slices) {
    size_t i = 0;
    while (i < order.size()) {
        const uint32_t bound = bound_of[order[i]];
        size_t j = i;
        while (j < order.size() && bound_of[order[j]] == bound) ++j;
        const size_t n_class = j - i;
#ifndef DIVANS_BATCH_ENCODE_SLICES      // experiment knob (scripts/r06_decode_slices.sh): slices a length class is cut into by a compress call
#define DIVANS_BATCH_ENCODE_SLICES 2
#endif
        size_t per = std::min<size_t>(16384, std::max<size_t>(256, (n_class + DIVANS_BATCH_ENCODE_SLICES - 1) / DIVANS_BATCH_ENCODE_SLICES));
        per = std::max<size_t>(1, std::min(per, budget_bytes / device_bytes_per_stream(bound)));
        for (size_t b = i; b < j; b += per) {
            Slice s; s.bound = bound;
            for (size_t k = b; k < std::min(j, b + per); ++k) { s.members.push_back(order[k]); s.bytes += sizes[order[k]]; }
            slices.push_back(std::move(s));
        }
        i = j;
    }
}

// Per lane, half of what is free in all -- asked ONCE per device, when its lanes are built (LanePool::acquire): the lanes' own buffers and
// codecs then eat into the free memory, and a budget re-read on every call shrank with them, cut the same batch into different slices
// from one call to the next and sent the odd last slice to a lane whose page-locked and device buffers had to be re-allocated -- 100 ms of
// hipHostMalloc / hipMalloc per compress call of 16 384 containers in "steady state" (round 6, profiles/r06_batch_container_rate_*).
size_t device_budget_now() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)8 << 30;
    return std::max<size_t>((size_t)1 << 30, free_b / (2 * kLanes));
}

struct CodecKey {
    divans_lit_config cfg; uint32_t bound;
    bool operator<(const CodecKey& o) const { const int c = std::memcmp(&cfg, &o.cfg, sizeof(cfg)); return c != 0 ? c < 0 : bound < o.bound; }
};

struct Lane {
    hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    struct Entry { divans_gpu_codec* c; uint32_t full_grid, cur_grid; uint64_t used; };   // full_grid: the persistent grid the codec chose for a full GPU; cur_grid: what it is set to
    std::map<CodecKey, Entry> codecs;     // at most kCodecsPerLane: a long-running process that sees many PredictionModes / length classes
    uint64_t tick = 0;                    // would otherwise pile up tables and scratch until hipMalloc fails
    static constexpr size_t kCodecsPerLane = 6;
    PinnedBuf h_in, h_off, h_sz, h_ooff, h_osz, h_out, h_chunks, h_total, h_flags, h_status;
    DeviceBuf d_in, d_off, d_sz, d_slots, d_ooff, d_osz, d_packed, d_poff, d_total, d_chunks, d_out, d_flags;
    long slice = -1;                 // slice in flight on this lane
    ~Lane() {
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& kv : codecs) divans_gpu_codec_destroy(kv.second.c);
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int init() {
        HIP_OR_FAIL(hipStreamCreate(&stream));
        HIP_OR_FAIL(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        return 0;
    }
    int codec_for(const divans_lit_config& cfg, uint32_t bound, int device, size_t n_streams, divans_gpu_codec** out) {
        CodecKey key; key.cfg = cfg; key.bound = bound;
        auto it = codecs.find(key);
        if (it == codecs.end()) {
            while (codecs.size() >= kCodecsPerLane) {          // evict the least recently used one; its launches are behind us after the sync
                auto lru = codecs.begin();
                for (auto j = codecs.begin(); j != codecs.end(); ++j) if (j->second.used < lru->second.used) lru = j;
                HIP_OR_FAIL(hipStreamSynchronize(stream));
                divans_gpu_codec_destroy(lru->second.c);
                codecs.erase(lru);
            }
            divans_gpu_codec* c = nullptr;
            const int rc = divans_gpu_codec_create(&c, &cfg, device, stream, bound);
            if (rc) return rc;
            // a lane's codec is never placement-tuned, whatever its configuration's table size: the search would hold a second copy of
            // the tables per lane and compare launches of slices that differ from call to call
            (void)divans_gpu_codec_tune_tables(c, 1);
            divans_gpu_info info;
            if (divans_gpu_codec_info(c, &info)) { divans_gpu_codec_destroy(c); return DIVANS_GPU_EHIP; }
            it = codecs.emplace(key, Entry{c, info.blocks, info.blocks, 0}).first;
        }
        it->second.used = ++tick;
        // a small slice does not need the full persistent grid's worth of CDF tables; a later, larger one gets the grid back
        const uint32_t want = (uint32_t)std::max<size_t>(1, std::min<size_t>(it->second.full_grid, (n_streams + 15) / 16));
        // (the grid is remembered here: asking the codec -- divans_gpu_codec_info -- waits for the events of its last launches)
        if (it->second.cur_grid != want && divans_gpu_codec_set_geometry(it->second.c, want, 0xffffffffu) == 0) it->second.cur_grid = want;
        *out = it->second.c;
        return 0;
    }
};

// The lanes live from call to call (streams, codecs with their tables and scratch, page-locked staging buffers are expensive to
// create): ONE POOL PER DEVICE, each handed to one call at a time.  Calls on different devices run concurrently -- the reference's
// states are independent of each other (src/ffi/interface.rs:49-50, src/parallel_decompressor.rs:55-141), and so are a process's
// GPUs: eight host threads drive eight devices through this interface without waiting for each other's device work (they share
// the host thread pool).  Nothing is torn down when a call names another device; divans_batch_release_device / divans_batch_release
// return a device's / every device's lanes.
struct LanePool {
    std::mutex mu;                       // held for the whole of a call on this device
    int device = -1;
    std::unique_ptr<Lane[]> lanes;
    size_t budget = 0;                   // device bytes a slice may take on one lane: fixed while the lanes live (device_budget_now)
    // the containers of a compress call between their assembly and their way home: kept (cleared, capacity intact) from call to call -- a
    // fresh vector per container meant 16 384 allocations whose first-touch page faults made up 40 % of the assembly (round 6); host memory
    // of about the size of the largest batch's containers, returned with the lanes (divans_batch_release*)
    std::vector<std::vector<uint8_t>> containers;
    int acquire(Lane** out) {
        if (!lanes) {
            std::unique_ptr<Lane[]> fresh(new Lane[kLanes]);
            for (int i = 0; i < kLanes; ++i) { const int rc = fresh[i].init(); if (rc) return rc; }
            budget = device_budget_now();
            lanes = std::move(fresh);
        }
        *out = lanes.get();
        return 0;
    }
};
struct PoolRegistry {
    std::mutex mu;                       // guards the map only; never held while a pool's own mutex is waited for by a call
    std::map<int, std::unique_ptr<LanePool>> by_device;
    LanePool& of(int device) {
        std::lock_guard<std::mutex> l(mu);
        std::unique_ptr<LanePool>& p = by_device[device];
        if (!p) { p.reset(new LanePool); p->device = device; }
        return *p;                       // pools are never erased: the reference stays valid for the life of the process
    }
    void release(int device) {           // device < 0: all
        std::vector<LanePool*> pools;
        { std::lock_guard<std::mutex> l(mu); for (auto& kv : by_device) if (device < 0 || kv.first == device) pools.push_back(kv.second.get()); }
        int before = -1;
        const bool have_before = hipGetDevice(&before) == hipSuccess;
        for (LanePool* p : pools) {
            std::lock_guard<std::mutex> l(p->mu);       // waits for a call that is running on that device
            if (p->lanes) { (void)hipSetDevice(p->device); p->lanes.reset(); }
            std::vector<std::vector<uint8_t>>().swap(p->containers);
        }
        if (have_before) (void)hipSetDevice(before);    // the caller's current device is not ours to change

    }
};
PoolRegistry& registry() { static PoolRegistry r; return r; }

// Where the calling thread's time went in its last call (divans_batch_last_phases): a diagnostic, per thread, overwritten by every call
thread_local double g_phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};
enum { PH_PARSE_OR_PLAN = 0, PH_STAGE = 1, PH_WAIT = 2, PH_FINISH = 3, PH_GATHER = 4, PH_SETUP = 5, PH_TEARDOWN = 6 };

// wall-clock bookkeeping of the overlap: host work counts as overlapped while at least one slice is in flight on the GPU
struct Overlap {
    int in_flight = 0; double overlapped = 0, serial = 0, gpu_first = -1, gpu_last = 0;
    void host(double t0, double t1) { (in_flight > 0 ? overlapped : serial) += t1 - t0; }
};

}  // namespace

extern "C" {

void divans_batch_options_default(divans_batch_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->window_size = 22; o->dynamic_context_mixing = 1; o->use_context_map = 1; o->force_stride = 9; o->call_buffer_size = 65536;
}

size_t divans_batch_compress_bound(size_t n) {
    // header + trailer + EOF marker, both coders' worst cases, one 3-byte Mux header per 4096-byte... per 16 bytes at worst
    const size_t lit = divans_gpu_lit_encode_bound(n);
    const size_t cmd = 16 + 4 * (16 + 8192 + 64 * 4 + 64 + (n >> 10) * 12 + 64);   // PredictionMode + a literal length per ring span, at < 2 bytes a nibble
    const size_t payload = lit + cmd;
    return 16 + 8 + 3 + payload + 3 * (payload / 16 + 8) + 64;
}

}  // extern "C"

namespace {

// One device's share of a compress call (the whole call when it names one device).  `keep` != null: the containers stay in *keep
// (one vector per stream) and the caller gathers them -- the all-devices form, whose output offsets depend on every device's sizes.
int compress_on_device(const divans_batch_options* opt, const uint8_t* const* inputs, const size_t* sizes, size_t n_streams,
                       uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing,
                       std::vector<std::vector<uint8_t>>* keep) {
    if (n_streams >= (1u << 24)) return set_last_error(DIVANS_GPU_EINVAL, "too many streams in one batch");
    const double t_begin = now_ms();
    const divans_host::StreamOptions so = to_stream_options(*opt);
    std::vector<uint32_t> bound_of(n_streams);
    for (size_t i = 0; i < n_streams; ++i) {
        if (sizes[i] > 0x7fffffffu) return set_last_error(DIVANS_GPU_EINVAL, "stream too long");
        bound_of[i] = class_bound(sizes[i]);
    }
    // the LIT configuration follows from the options alone (the PredictionMode of the internal compressor)
    // ... and so does the first command of every stream: its trip through the CMD model is made once (host_stream.h, PlanPrefix)
    const std::shared_ptr<const divans_host::PlanPrefix> prefix = divans_host::make_plan_prefix(so);
    if (!prefix) return set_last_error(DIVANS_GPU_EINVAL, "options cannot be coded");
    divans_host::StreamPlan probe;
    int rc = divans_host::plan_stream(so, 0, nullptr, probe, prefix.get());
    if (rc) return set_last_error(rc, "options cannot be coded");
    HIP_OR_FAIL(hipSetDevice(opt->device));
    std::vector<size_t> order(n_streams);
    for (size_t i = 0; i < n_streams; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bound_of[a] < bound_of[b]; });
    LanePool& lane_pool = registry().of(opt->device);
    std::lock_guard<std::mutex> pool_lock(lane_pool.mu);
    Lane* lanes = nullptr;
    rc = lane_pool.acquire(&lanes); if (rc) return rc;
    std::vector<Slice> slices;
    make_slices(order, bound_of, sizes, lane_pool.budget, slices);
    struct Drain { Lane* l; ~Drain() { for (int i = 0; i < kLanes; ++i) if (l[i].stream) (void)hipStreamSynchronize(l[i].stream); } } drain{lanes};   // no early return with copies in flight
    Overlap ov;

    // the CMD coder of every stream sees lengths and options only: one plan per distinct length, made while the first slices run
    std::map<size_t, std::unique_ptr<divans_host::StreamPlan>> plans;
    for (size_t i = 0; i < n_streams; ++i) plans.emplace(sizes[i], nullptr);
    std::vector<std::unique_ptr<divans_host::StreamPlan>*> plan_slots; std::vector<size_t> plan_len;
    for (auto& kv : plans) { plan_slots.push_back(&kv.second); plan_len.push_back(kv.first); }
    std::atomic<int> plan_rc{0};
    bool plans_done = false;
    for (double& v : g_phases) v = 0;
    auto make_plans = [&]() {
        const double t0 = now_ms();
        parallel_for(plan_slots.size(), opt->host_threads, [&](size_t k) {
            auto p = std::make_unique<divans_host::StreamPlan>();
            const int r = divans_host::plan_stream(so, plan_len[k], nullptr, *p, prefix.get());
            if (r) plan_rc = r;
            *plan_slots[k] = std::move(p);
        });
        plans_done = true;
        ov.host(t0, now_ms()); g_phases[PH_PARSE_OR_PLAN] += now_ms() - t0;
    };

    auto issue = [&](size_t k) -> int {
        Lane& L = lanes[k % kLanes];
        const Slice& s = slices[k];
        const size_t m = s.members.size();
        const uint32_t max_chunks = (uint32_t)std::max<uint64_t>(1, (2ull * s.bound + 65535ull) / 65536ull);
        const uint64_t slot = divans_gpu_lit_encode_bound(s.bound);
        const double t0 = now_ms();
        HIP_OR_FAIL(L.h_in.reserve(s.bytes + 64)); HIP_OR_FAIL(L.h_off.reserve(8 * m)); HIP_OR_FAIL(L.h_sz.reserve(4 * m));
        HIP_OR_FAIL(L.h_ooff.reserve(8 * m)); HIP_OR_FAIL(L.h_osz.reserve(4 * m)); HIP_OR_FAIL(L.h_total.reserve(8));
        HIP_OR_FAIL(L.h_chunks.reserve(4ull * m * max_chunks));
        const size_t guess = std::min<size_t>(slot * m, s.bytes / 8 * 5 + 64 * m + 4096);    // page-locked memory is expensive: what packed streams usually need (5/8 of the input)
        HIP_OR_FAIL(L.h_out.reserve(guess + 64));
        HIP_OR_FAIL(L.d_in.reserve(s.bytes + 64)); HIP_OR_FAIL(L.d_off.reserve(8 * m)); HIP_OR_FAIL(L.d_sz.reserve(4 * m));
        HIP_OR_FAIL(L.d_slots.reserve(slot * m + 64)); HIP_OR_FAIL(L.d_ooff.reserve(8 * m)); HIP_OR_FAIL(L.d_osz.reserve(4 * m));
        HIP_OR_FAIL(L.d_packed.reserve(slot * m + 64)); HIP_OR_FAIL(L.d_poff.reserve(8 * m)); HIP_OR_FAIL(L.d_total.reserve(8));
        HIP_OR_FAIL(L.d_chunks.reserve(4ull * m * max_chunks));
        {
            uint64_t* off = L.h_off.as<uint64_t>(); uint32_t* sz = L.h_sz.as<uint32_t>(); uint8_t* dst = L.h_in.as<uint8_t>();
            uint64_t pos = 0;
            for (size_t j = 0; j < m; ++j) { off[j] = pos; sz[j] = (uint32_t)sizes[s.members[j]]; pos += sizes[s.members[j]]; }
            const double tc0 = now_ms();
            g_phases[PH_SETUP] += tc0 - t0;                  // (compress: buffer reservations + offsets, part of [1])
            parallel_for(m, opt->host_threads, [&](size_t j) { if (sz[j]) stream_copy(dst + off[j], inputs[s.members[j]], sz[j]); });
            g_phases[PH_TEARDOWN] += now_ms() - tc0;         // (compress: the copy into page-locked memory, part of [1]; the rest of [1] is enqueueing)
        }
        divans_gpu_codec* codec = nullptr;
        int r = L.codec_for(probe.cfg, s.bound, opt->device, m, &codec); if (r) return r;
        HIP_OR_FAIL(hipMemcpyAsync(L.d_in.p, L.h_in.p, s.bytes, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_off.p, L.h_off.p, 8 * m, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_sz.p, L.h_sz.p, 4 * m, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemsetAsync(L.d_chunks.p, 0, 4ull * m * max_chunks, L.stream));
        r = divans_gpu_codec_clear_status(codec); if (r) return r;          // the codec outlives the call: no bit of an earlier, abandoned slice
        const double te0 = now_ms();
        r = divans_gpu_lit_encode_batch_chunks(codec, L.d_in.as<uint8_t>(), L.d_off.as<uint64_t>(), L.d_sz.as<uint32_t>(), s.bound, (uint32_t)m,
                                               L.d_slots.as<uint8_t>(), slot, L.d_ooff.as<uint64_t>(), L.d_osz.as<uint32_t>(), L.d_chunks.as<uint32_t>(), max_chunks);
        if (r) return r;
        r = divans_gpu_pack_streams(codec, L.d_slots.as<uint8_t>(), L.d_ooff.as<uint64_t>(), L.d_osz.as<uint32_t>(), (uint32_t)m,
                                    L.d_packed.as<uint8_t>(), L.d_poff.as<uint64_t>(), L.d_total.as<uint64_t>());
        if (r) return r;
        g_phases[7] += now_ms() - te0;        // (compress: inside the codec's encode + pack entry points, part of [1])
        HIP_OR_FAIL(L.h_status.reserve(64));
        r = divans_gpu_codec_status_async(codec, L.h_status.as<uint32_t>()); if (r) return r;     // read in complete(), behind the lane's event
        // the packed streams are at most slot * m bytes, in practice about half the input: copy what a stream can be at most only
        // when the slice is tiny, otherwise the first bytes that can hold the whole slice at the input's size (checked on completion)
        HIP_OR_FAIL(hipMemcpyAsync(L.h_ooff.p, L.d_poff.p, 8 * m, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.h_osz.p, L.d_osz.p, 4 * m, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.h_chunks.p, L.d_chunks.p, 4ull * m * max_chunks, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.h_total.p, L.d_total.p, 8, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.h_out.p, L.d_packed.p, guess, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipEventRecord(L.done, L.stream));
        L.slice = (long)k;
        const double t1 = now_ms();
        ov.host(t0, t1); g_phases[PH_STAGE] += t1 - t0;
        if (ov.gpu_first < 0) ov.gpu_first = t0;
        ov.in_flight += 1;
        return 0;
    };

    std::vector<std::vector<uint8_t>> own_results;
    if (keep) own_results.resize(n_streams);                    // (the all-devices form hands them to the caller's gather)
    else if (lane_pool.containers.size() < n_streams) lane_pool.containers.resize(n_streams);
    std::vector<std::vector<uint8_t>>& results = keep ? own_results : lane_pool.containers;
    const size_t call_buffer = opt->call_buffer_size ? opt->call_buffer_size : 65536;
    // The containers go home in stream order, back to back: one can be placed as soon as every container in front of it has been assembled.
    // A batch of one length class is sliced in stream order, so the first slice's containers are copied out while the GPU codes the second
    // (round 5 gathered everything after the last slice: 15 of 83 ms per 16 384 containers).
    std::vector<char> ready(n_streams, 0);
    size_t next_out = 0, out_pos = 0;
    auto drain_ready = [&]() -> int {
        if (keep) return 0;
        const size_t first = next_out;
        while (next_out < n_streams && ready[next_out]) { out_offsets[next_out] = out_pos; out_sizes[next_out] = results[next_out].size(); out_pos += results[next_out].size(); ++next_out; }
        if (out_pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
        if (next_out == first) return 0;
        const double t0 = now_ms();
        parallel_for(next_out - first, opt->host_threads, [&](size_t j) { stream_copy(out + out_offsets[first + j], results[first + j].data(), results[first + j].size()); });
        ov.host(t0, now_ms()); g_phases[PH_GATHER] += now_ms() - t0;
        return 0;
    };
    auto complete = [&](size_t k) -> int {
        Lane& L = lanes[k % kLanes];
        const Slice& s = slices[k];
        const size_t m = s.members.size();
        const uint32_t max_chunks = (uint32_t)std::max<uint64_t>(1, (2ull * s.bound + 65535ull) / 65536ull);
        const double tw = now_ms();
        HIP_OR_FAIL(hipEventSynchronize(L.done));
        g_phases[PH_WAIT] += now_ms() - tw;
        const uint64_t packed_total = *L.h_total.as<uint64_t>();
        const size_t guess = std::min<size_t>(divans_gpu_lit_encode_bound(s.bound) * m, s.bytes / 8 * 5 + 64 * m + 4096);
        if (packed_total > guess) {   // incompressible input: a larger staging buffer, the whole slice again
            HIP_OR_FAIL(L.h_out.reserve(packed_total + 64));
            HIP_OR_FAIL(hipMemcpyAsync(L.h_out.p, L.d_packed.p, packed_total, hipMemcpyDeviceToHost, L.stream));
            HIP_OR_FAIL(hipStreamSynchronize(L.stream));
        }
        ov.in_flight -= 1; ov.gpu_last = now_ms();
        divans_gpu_codec* codec = nullptr;
        int r = L.codec_for(probe.cfg, s.bound, opt->device, m, &codec); if (r) return r;
        if (L.h_status.as<uint32_t>()[0]) return set_last_error(DIVANS_GPU_EINVAL, "the literal coder reported an invalid model state");
        if (!plans_done) make_plans();
        if (plan_rc) return set_last_error(plan_rc, "a stream's command stream cannot be coded");
        // framing: Mux replay, EOF marker, CRC trailer per stream
        const double t0 = now_ms();
        std::atomic<int> asm_rc{0};
        parallel_for(m, opt->host_threads, [&](size_t j) {
            const size_t i = s.members[j];
            const divans_host::StreamPlan& p = *plans[sizes[i]];
            const int rr = divans_host::assemble_container(p, L.h_out.as<uint8_t>() + L.h_ooff.as<uint64_t>()[j], L.h_osz.as<uint32_t>()[j],
                                                           L.h_chunks.as<uint32_t>() + j * max_chunks, call_buffer, results[i]);
            if (rr) asm_rc = rr;
        });
        ov.host(t0, now_ms()); g_phases[PH_FINISH] += now_ms() - t0;
        L.slice = -1;
        if (asm_rc) return set_last_error(asm_rc, "container assembly failed");
        for (size_t j = 0; j < m; ++j) ready[s.members[j]] = 1;
        return drain_ready();
    };

    // software pipeline: up to kLanes slices in flight; the plans are made under the first of them
    const size_t ns = slices.size();
    for (size_t k = 0; k < std::min<size_t>(ns, kLanes); ++k) { rc = issue(k); if (rc) return rc; }
    if (!plans_done) make_plans();
    for (size_t k = 0; k < ns; ++k) {
        rc = complete(k); if (rc) return rc;
        if (k + kLanes < ns) { rc = issue(k + kLanes); if (rc) return rc; }
    }
    if (keep) *keep = std::move(own_results);
    else if (next_out != n_streams) return set_last_error(DIVANS_GPU_EHIP, "internal: a container was never assembled");
    const double t_end = now_ms();
    if (timing) {
        timing->total_ms = t_end - t_begin; timing->gpu_ms = ov.gpu_last - ov.gpu_first;
        timing->host_overlapped_ms = ov.overlapped; timing->host_serial_ms = ov.serial;
    }
    return 0;
}

// A decompress call that names all devices parses every container before the devices start (the output offsets of a device's range are
// the decoded sizes of everything in front of it): the device shares then take the parsed streams and their first output offset from here.
struct PreParsed { divans_host::ParsedStream* parsed; size_t pos0; size_t index_base; };

int decompress_on_device(const divans_batch_options* opt, const uint8_t* const* containers, const size_t* sizes, size_t n_streams,
                         uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing, const PreParsed* pre);

// ---- all devices behind one call (divans_batch_options::device = DIVANS_BATCH_ALL_DEVICES) --------------------------------------------
// The streams are cut into contiguous ranges, one per device -- range r of D is [n * r / D, n * (r + 1) / D), the rule of
// divans_amd/sharding.py shard_bounds and SURVEY.md 8e -- every range is driven by its own thread through that device's lanes, exactly
// as a single-device call would, and the outputs land in stream order.  The host threads are divided between the device threads
// (a budget each, open at the same time: ThreadPool); nothing moves between devices.  The reference's analogue: independent
// compressor / decompressor states, one per thread (src/ffi/interface.rs:49-50, src/parallel_decompressor.rs:55-141).
struct DeviceShare { size_t b = 0, e = 0; int device = 0; int rc = 0; std::string err; divans_batch_timing t = {0, 0, 0, 0}; double phases[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };

int plan_shares(const divans_batch_options* opt, size_t n_streams, std::vector<DeviceShare>& shares, divans_batch_options& per_device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_last_error(DIVANS_GPU_EHIP, "no HIP device: the divans batch interface has no CPU fallback");
    const size_t D = std::max<size_t>(1, std::min<size_t>((size_t)ndev, n_streams));
    shares.resize(D);
    for (size_t r = 0; r < D; ++r) { shares[r].b = n_streams * r / D; shares[r].e = n_streams * (r + 1) / D; shares[r].device = (int)r; }
    per_device = *opt;
    const int total = opt->host_threads > 0 ? opt->host_threads : usable_threads();
    per_device.host_threads = std::max(1, total / (int)D);       // a device thread's budget; the budgets are open at the same time
    return 0;
}

template <typename Body>
void run_shares(std::vector<DeviceShare>& shares, Body&& body) {
    int before = -1;
    const bool have_before = hipGetDevice(&before) == hipSuccess;
    auto one = [&](DeviceShare& sh) {
        sh.rc = body(sh);
        if (sh.rc) sh.err = divans_gpu_last_error();      // the message lives in the device thread: bring it home
        divans_batch_last_phases(sh.phases, 8);
    };
    std::vector<std::thread> ts;
    for (size_t r = 1; r < shares.size(); ++r) ts.emplace_back([&, r] { one(shares[r]); });
    one(shares[0]);                                       // the calling thread drives the first device itself
    for (auto& t : ts) t.join();
    if (have_before) (void)hipSetDevice(before);          // the caller's current device is not ours to change
}

int finish_shares(const std::vector<DeviceShare>& shares, double t_begin, divans_batch_timing* timing) {
    const DeviceShare* slowest = &shares[0];
    for (const DeviceShare& sh : shares) {
        if (sh.rc) return set_last_error(sh.rc, "device " + std::to_string(sh.device) + " (streams " + std::to_string(sh.b) + ".." + std::to_string(sh.e) + "): " + sh.err);
        if (sh.t.total_ms > slowest->t.total_ms) slowest = &sh;
    }
    for (int i = 0; i < 8; ++i) g_phases[i] = slowest->phases[i];       // divans_batch_last_phases: the slowest device's thread
    if (timing) { *timing = slowest->t; timing->total_ms = now_ms() - t_begin; }
    return 0;
}

}  // namespace

extern "C" {

int divans_batch_compress(const divans_batch_options* opt, const uint8_t* const* inputs, const size_t* sizes, size_t n_streams,
                          uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing) {
    if (!opt || (!inputs && n_streams) || (!sizes && n_streams) || !out || !out_offsets || !out_sizes) return set_last_error(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    if (opt->device >= 0) return compress_on_device(opt, inputs, sizes, n_streams, out, out_cap, out_offsets, out_sizes, timing, nullptr);
    if (opt->device != DIVANS_BATCH_ALL_DEVICES && opt->device != DIVANS_BATCH_ALL_DEVICES_SHARDED) return set_last_error(DIVANS_GPU_EINVAL, "device must be a HIP device or DIVANS_BATCH_ALL_DEVICES");
    const double t_begin = now_ms();
    std::vector<DeviceShare> shares; divans_batch_options po;
    int rc = plan_shares(opt, n_streams, shares, po); if (rc) return rc;
    if (shares.size() == 1 && opt->device == DIVANS_BATCH_ALL_DEVICES) {        // one device (or one stream): the one-device call, with its pipeline intact
        divans_batch_options o = *opt; o.device = shares[0].device;
        return compress_on_device(&o, inputs, sizes, n_streams, out, out_cap, out_offsets, out_sizes, timing, nullptr);
    }
    std::vector<std::vector<std::vector<uint8_t>>> kept(shares.size());
    run_shares(shares, [&](DeviceShare& sh) {
        divans_batch_options o = po; o.device = sh.device;
        return compress_on_device(&o, inputs + sh.b, sizes + sh.b, sh.e - sh.b, nullptr, 0, out_offsets + sh.b, out_sizes + sh.b, &sh.t, &kept[&sh - shares.data()]);
    });
    rc = finish_shares(shares, t_begin, nullptr); if (rc) return rc;
    // the containers of all devices, in stream order
    const double tg = now_ms();
    size_t pos = 0;
    std::vector<const std::vector<uint8_t>*> flat(n_streams);
    for (size_t r = 0; r < shares.size(); ++r)
        for (size_t i = shares[r].b; i < shares[r].e; ++i) { flat[i] = &kept[r][i - shares[r].b]; out_offsets[i] = pos; out_sizes[i] = flat[i]->size(); pos += flat[i]->size(); }
    if (pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
    parallel_for(n_streams, opt->host_threads, [&](size_t i) { stream_copy(out + out_offsets[i], flat[i]->data(), flat[i]->size()); });
    const double gather_ms = now_ms() - tg;
    rc = finish_shares(shares, t_begin, timing);      // (sets the phases to the slowest share's; the gather is this thread's own)
    g_phases[PH_GATHER] += gather_ms;
    if (timing) timing->host_serial_ms += gather_ms;
    return rc;
}

int divans_batch_decompress(const divans_batch_options* opt, const uint8_t* const* containers, const size_t* sizes, size_t n_streams,
                            uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing) {
    if (!opt || (!containers && n_streams) || (!sizes && n_streams) || !out || !out_offsets || !out_sizes) return set_last_error(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    if (opt->device >= 0) return decompress_on_device(opt, containers, sizes, n_streams, out, out_cap, out_offsets, out_sizes, timing, nullptr);
    if (opt->device != DIVANS_BATCH_ALL_DEVICES && opt->device != DIVANS_BATCH_ALL_DEVICES_SHARDED) return set_last_error(DIVANS_GPU_EINVAL, "device must be a HIP device or DIVANS_BATCH_ALL_DEVICES");
    const double t_begin = now_ms();
    std::vector<DeviceShare> shares; divans_batch_options po;
    int rc = plan_shares(opt, n_streams, shares, po); if (rc) return rc;
    if (shares.size() == 1 && opt->device == DIVANS_BATCH_ALL_DEVICES) {        // one device: no range needs another's sizes, the one-device pipeline parses under its GPU work
        divans_batch_options o = *opt; o.device = shares[0].device;
        return decompress_on_device(&o, containers, sizes, n_streams, out, out_cap, out_offsets, out_sizes, timing, nullptr);
    }
    // every container's framing, CRC and CMD coder first: the decoded sizes place each device's range in the output
    std::vector<divans_host::ParsedStream> parsed(n_streams);
    std::vector<int> status(n_streams, 0);
    {
        divans_host::ParseMemo memo;
        parallel_for(n_streams, opt->host_threads, [&](size_t i) {
            status[i] = (int)divans_host::parse_container_host(containers[i], sizes[i], opt->skip_crc != 0, (size_t)1 << 30, parsed[i], nullptr, &memo, true);
        });
    }
    size_t pos = 0;
    for (size_t i = 0; i < n_streams; ++i) {
        if (status[i] != divans_host::PARSE_OK) { out_sizes[i] = (size_t)-1; return set_last_error(DIVANS_GPU_ECORRUPT, "container " + std::to_string(i) + " is truncated, corrupt or not a literal-only stream"); }
        out_offsets[i] = pos; pos += parsed[i].total;
    }
    if (pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
    const double parse_ms = now_ms() - t_begin;
    run_shares(shares, [&](DeviceShare& sh) {
        divans_batch_options o = po; o.device = sh.device;
        const PreParsed pre = {parsed.data() + sh.b, out_offsets[sh.b], sh.b};
        return decompress_on_device(&o, containers + sh.b, sizes + sh.b, sh.e - sh.b, out, out_cap, out_offsets + sh.b, out_sizes + sh.b, &sh.t, &pre);
    });
    rc = finish_shares(shares, t_begin, timing);
    if (rc == 0) { g_phases[PH_PARSE_OR_PLAN] += parse_ms; if (timing) timing->host_serial_ms += parse_ms; }
    return rc;
}

}  // extern "C"

namespace {

int decompress_on_device(const divans_batch_options* opt, const uint8_t* const* containers, const size_t* sizes, size_t n_streams,
                         uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing, const PreParsed* pre) {
    const double t_begin = now_ms();
    const size_t index_base = pre ? pre->index_base : 0;      // (error messages name the caller's container index)
    HIP_OR_FAIL(hipSetDevice(opt->device));
    LanePool& lane_pool = registry().of(opt->device);
    std::lock_guard<std::mutex> pool_lock(lane_pool.mu);
    Lane* lanes = nullptr;
    int rc = lane_pool.acquire(&lanes); if (rc) return rc;
    struct Drain { Lane* l; ~Drain() { for (int i = 0; i < kLanes; ++i) if (l[i].stream) (void)hipStreamSynchronize(l[i].stream); } } drain{lanes};   // no early return with copies in flight
    Overlap ov;
    for (double& v : g_phases) v = 0;
    const double t_setup0 = now_ms();
    const size_t budget = lane_pool.budget;
    // Slices in stream order (the output offsets are the running sum of the decoded sizes): one per lane, 128 .. 8192
    // containers.  While the GPU decodes the slices in flight (concurrently: a stream is a serial chain of tens of
    // milliseconds), host threads parse the next one (framing, CRC, CMD coder -> decoded sizes and LIT configuration, which
    // the launch needs) and copy out the ones that have finished.
    // FOUR slices in flight, not one per lane: the decode kernels of more than four HIP streams do not run at the same time (the runtime gives a
    // process four hardware queues by default, GPU_MAX_HW_QUEUES), and a slice whose kernel waits behind another's costs a whole decode chain --
    // 16 384 x 64 KiB: 2 slices 11.4 GB/s, 3: 10.5, 4: 13.1, 5 / 6 / 8: 8.7-8.9, 16: 6.0 (profiles/r06_decode_slices.txt; rounds 3-5 ran eight)
#ifndef DIVANS_BATCH_DECODE_SLICES      // (experiment knob, scripts/r06_decode_slices.sh)
#define DIVANS_BATCH_DECODE_SLICES 4
#endif
    constexpr size_t kDecodeInFlight = DIVANS_BATCH_DECODE_SLICES;
    const size_t per = std::min<size_t>(8192, std::max<size_t>(128, (n_streams + kDecodeInFlight - 1) / kDecodeInFlight));
    const size_t n_blocks = (n_streams + per - 1) / per;      // parse blocks; a block is then cut into the slices that are issued
    struct Range { size_t b, e; };
    std::vector<Range> slices;                                // in issue order; slice k runs on lane k % kLanes
    std::vector<divans_host::ParsedStream> own_parsed(pre ? 0 : n_streams);
    divans_host::ParsedStream* const parsed = pre ? pre->parsed : own_parsed.data();
    std::vector<int> status(n_streams, 0);
    divans_host::ParseMemo memo;     // equal-length streams of one producer carry the same CMD bytes: decode them once (host_stream.h)
    size_t pos = pre ? pre->pos0 : 0;
    g_phases[PH_SETUP] += now_ms() - t_setup0;

    struct Group { divans_lit_config cfg; int cfg_id; uint32_t bound; std::vector<size_t> members; size_t in_bytes = 0, out_bytes = 0, in_base = 0, out_base = 0, idx_base = 0; };
    std::vector<std::vector<Group>> slice_groups;

    // What a decoding slice costs on the device: its coded bytes in, its decoded bytes out, offsets / sizes / flags.  (The codec's
    // tables belong to the persistent grid, not to the slice.)  A parse block is cut into slices that stay under the lane's budget;
    // only a single stream that does not fit by itself fails the call.
    auto decode_bytes_of = [&](size_t i) -> size_t { return parsed[i].lit_size + parsed[i].total + 192; };
    auto cut_block = [&](size_t b, size_t e) -> int {
        size_t sb = b, acc = 0;
        for (size_t i = b; i < e; ++i) {
            const size_t need = decode_bytes_of(i);
            if (need > budget * 2) { out_sizes[i] = (size_t)-1; return set_last_error(DIVANS_GPU_ENOMEM, "container " + std::to_string(index_base + i) + " does not fit the device by itself"); }
            if (i > sb && acc + need > budget) { slices.push_back({sb, i}); sb = i; acc = 0; }
            acc += need;
        }
        if (e > sb) slices.push_back({sb, e});
        slice_groups.resize(slices.size());
        return 0;
    };

    auto parse = [&](size_t kb) -> int {
        const size_t b = kb * per, e = std::min(n_streams, b + per);
        if (pre) return cut_block(b, e);        // parsed (and found whole) before the devices started
        const double t0 = now_ms();
        parallel_for(e - b, opt->host_threads, [&](size_t j) {
            const size_t i = b + j;
            status[i] = (int)divans_host::parse_container_host(containers[i], sizes[i], opt->skip_crc != 0, (size_t)1 << 30, parsed[i], nullptr, &memo, true);
        });
        ov.host(t0, now_ms()); g_phases[PH_PARSE_OR_PLAN] += now_ms() - t0;
        for (size_t i = b; i < e; ++i)
            if (status[i] != divans_host::PARSE_OK) { out_sizes[i] = (size_t)-1; return set_last_error(DIVANS_GPU_ECORRUPT, "container " + std::to_string(index_base + i) + " is truncated, corrupt or not a literal-only stream"); }
        return cut_block(b, e);
    };

    auto issue = [&](size_t k) -> int {
        Lane& L = lanes[k % kLanes];
        const size_t b = slices[k].b, e = slices[k].e;
        const double t0 = now_ms();
        // group the slice by LIT configuration and length class (one codec = one configuration and one table / scratch size)
        std::vector<Group>& groups = slice_groups[k];
        for (size_t i = b; i < e; ++i) {
            out_offsets[i] = pos; out_sizes[i] = parsed[i].total; pos += parsed[i].total;
            if (parsed[i].total == 0) continue;
            const uint32_t bound = class_bound(parsed[i].total);
            Group* g = nullptr;
            for (auto& q : groups)    // the memo's configuration ids spare the 25 KB comparison per container
                if (q.bound == bound && ((q.cfg_id >= 0 && q.cfg_id == parsed[i].cfg_id) || ((q.cfg_id < 0 || parsed[i].cfg_id < 0) && std::memcmp(&q.cfg, parsed[i].cfg.get(), sizeof(divans_lit_config)) == 0))) { g = &q; break; }
            if (!g) { groups.emplace_back(); g = &groups.back(); g->cfg = *parsed[i].cfg; g->cfg_id = parsed[i].cfg_id; g->bound = bound; }
            g->members.push_back(i); g->in_bytes += parsed[i].lit_size; g->out_bytes += parsed[i].total;
        }
        if (pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
        // the slice's decoded bytes sit in d_out exactly as they will in `out` (stream order, back to back): one copy brings them home
        const size_t slice_out_base = out_offsets[b];
        size_t in_total = 0, m_total = 0;
        const size_t out_total = pos - slice_out_base;
        for (auto& g : groups) {
            g.in_base = in_total; g.out_base = 0; g.idx_base = m_total;
            in_total += (g.in_bytes + 127) & ~(size_t)63; m_total += g.members.size();
        }
        HIP_OR_FAIL(L.h_in.reserve(in_total + 128)); HIP_OR_FAIL(L.h_off.reserve(8 * m_total)); HIP_OR_FAIL(L.h_sz.reserve(4 * m_total));
        HIP_OR_FAIL(L.h_ooff.reserve(8 * m_total)); HIP_OR_FAIL(L.h_osz.reserve(4 * m_total));
        HIP_OR_FAIL(L.h_out.reserve(out_total + 64));
        HIP_OR_FAIL(L.h_flags.reserve(m_total + 64)); HIP_OR_FAIL(L.h_status.reserve(4 * groups.size() + 64));
        HIP_OR_FAIL(L.d_in.reserve(in_total + 128)); HIP_OR_FAIL(L.d_off.reserve(8 * m_total)); HIP_OR_FAIL(L.d_sz.reserve(4 * m_total));
        HIP_OR_FAIL(L.d_ooff.reserve(8 * m_total)); HIP_OR_FAIL(L.d_osz.reserve(4 * m_total)); HIP_OR_FAIL(L.d_out.reserve(out_total + 64));
        HIP_OR_FAIL(L.d_flags.reserve(m_total + 64));
        // the streams of a group lie back to back (whole 32-bit words each); only the padding behind a group and behind the slice is not
        // overwritten below and is zeroed -- clearing the whole buffer (half a gigabyte per 16 384 containers, on the calling thread, in
        // front of every slice's launch) was a third of the time the slices took to get under way
        for (const auto& g : groups) std::memset(L.h_in.as<uint8_t>() + g.in_base + g.in_bytes, 0, ((g.in_bytes + 127) & ~(size_t)63) - g.in_bytes);
        std::memset(L.h_in.as<uint8_t>() + in_total, 0, 128);
        for (auto& g : groups) {
            uint64_t ip = 0;
            std::vector<uint64_t> ioff(g.members.size());
            for (size_t j = 0; j < g.members.size(); ++j) {
                const divans_host::ParsedStream& ps = parsed[g.members[j]];
                ioff[j] = ip;
                L.h_off.as<uint64_t>()[g.idx_base + j] = ip; L.h_sz.as<uint32_t>()[g.idx_base + j] = (uint32_t)ps.lit_size;
                L.h_ooff.as<uint64_t>()[g.idx_base + j] = out_offsets[g.members[j]] - slice_out_base; L.h_osz.as<uint32_t>()[g.idx_base + j] = (uint32_t)ps.total;
                ip += ps.lit_size;
            }
            parallel_for(g.members.size(), opt->host_threads, [&](size_t j) {
                const divans_host::ParsedStream& ps = parsed[g.members[j]];
                // straight from the caller's container into page-locked memory (the parser left spans, not a copy)
                if (!ps.lit_spans.empty()) ps.copy_lit(containers[g.members[j]], L.h_in.as<uint8_t>() + g.in_base + ioff[j]);
                else if (ps.lit_size) std::memcpy(L.h_in.as<uint8_t>() + g.in_base + ioff[j], ps.lit.data(), ps.lit_size);
            });
        }
        HIP_OR_FAIL(hipMemcpyAsync(L.d_in.p, L.h_in.p, in_total + 128, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_off.p, L.h_off.p, 8 * m_total, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_sz.p, L.h_sz.p, 4 * m_total, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_ooff.p, L.h_ooff.p, 8 * m_total, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.d_osz.p, L.h_osz.p, 4 * m_total, hipMemcpyHostToDevice, L.stream));
        HIP_OR_FAIL(hipMemsetAsync(L.d_flags.p, 0, m_total + 64, L.stream));
        for (auto& g : groups) {
            divans_gpu_codec* codec = nullptr;
            int r = L.codec_for(g.cfg, g.bound, opt->device, g.members.size(), &codec); if (r) return r;
            r = divans_gpu_codec_set_stream_flags(codec, L.d_flags.as<uint8_t>() + g.idx_base); if (r) return r;
            r = divans_gpu_codec_clear_status(codec); if (r) return r;      // the codec outlives the call: no bit of an earlier, abandoned slice
            r = divans_gpu_lit_decode_batch(codec, L.d_in.as<uint8_t>() + g.in_base, L.d_off.as<uint64_t>() + g.idx_base, L.d_sz.as<uint32_t>() + g.idx_base,
                                            (uint32_t)g.members.size(), L.d_out.as<uint8_t>() + g.out_base, L.d_ooff.as<uint64_t>() + g.idx_base,
                                            L.d_osz.as<uint32_t>() + g.idx_base, g.bound);
            if (r) return r;
            r = divans_gpu_codec_status_async(codec, L.h_status.as<uint32_t>() + (&g - groups.data())); if (r) return r;
        }
        // (copying straight into a caller's page-locked buffer instead was measured and is no faster: 186 vs 140 ms for 16 384 x 64 KiB,
        // profiles/r04d_batch_container_rate_summary.txt -- the host copy below runs at 54 GB/s under the GPU work of the later slices)
        if (out_total) HIP_OR_FAIL(hipMemcpyAsync(L.h_out.p, L.d_out.p, out_total, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipMemcpyAsync(L.h_flags.p, L.d_flags.p, m_total, hipMemcpyDeviceToHost, L.stream));
        HIP_OR_FAIL(hipEventRecord(L.done, L.stream));
        const double t1 = now_ms();
        ov.host(t0, t1); g_phases[PH_STAGE] += t1 - t0;
        if (ov.gpu_first < 0) ov.gpu_first = t0;
        ov.in_flight += 1;
        return 0;
    };

    auto complete = [&](size_t k) -> int {
        Lane& L = lanes[k % kLanes];
        const size_t b = slices[k].b, e = slices[k].e;
        const double tw = now_ms();
        HIP_OR_FAIL(hipEventSynchronize(L.done));
        g_phases[PH_WAIT] += now_ms() - tw;
        ov.in_flight -= 1; ov.gpu_last = now_ms();
        const double t0 = now_ms();
        std::vector<Group>& groups = slice_groups[k];
        for (auto& g : groups) {
            const uint32_t st = L.h_status.as<uint32_t>()[&g - groups.data()];     // copied behind the group's launch, in page-locked memory by now
            // the per-stream flags are the kernels' own record (the status word is the codec's, and a codec may have been rebuilt)
            bool flagged = false;
            for (size_t j = 0; j < g.members.size() && !flagged; ++j) flagged = L.h_flags.as<uint8_t>()[g.idx_base + j] != 0;
            if ((st & DIVANS_GPU_STATUS_BAD_STREAM) || flagged) {
                size_t first = g.members.front();
                for (size_t j = 0; j < g.members.size(); ++j) if (L.h_flags.as<uint8_t>()[g.idx_base + j]) { first = g.members[j]; out_sizes[first] = (size_t)-1; break; }
                return set_last_error(DIVANS_GPU_ECORRUPT, "the LIT stream of container " + std::to_string(index_base + first) + " failed the decoder's integrity check");
            }
        }
        if (e > b) {      // the host threads bring the slice home, 256 KiB at a time
            const size_t slice_out_base = out_offsets[b], bytes = out_offsets[e - 1] + parsed[e - 1].total - slice_out_base;
            const size_t piece = (size_t)256 << 10, pieces = (bytes + piece - 1) / piece;
            parallel_for(pieces, opt->host_threads, [&](size_t q) {
                stream_copy(out + slice_out_base + q * piece, L.h_out.as<uint8_t>() + q * piece, std::min(piece, bytes - q * piece));
            });
        }
        for (size_t i = b; i < e; ++i) { std::vector<uint8_t>().swap(parsed[i].lit); std::vector<std::pair<uint32_t, uint32_t>>().swap(parsed[i].lit_spans); }
        groups.clear();
        ov.host(t0, now_ms()); g_phases[PH_FINISH] += now_ms() - t0;
        return 0;
    };

    // software pipeline: parse a block, issue its slices; the oldest slice is completed once kLanes are in flight
    size_t issued = 0, completed = 0;
    for (size_t kb = 0; kb < n_blocks; ++kb) {
        rc = parse(kb); if (rc) return rc;
        while (issued < slices.size()) {
            if (issued - completed >= std::min<size_t>(kLanes, kDecodeInFlight)) { rc = complete(completed++); if (rc) return rc; }
            rc = issue(issued++); if (rc) return rc;
        }
    }
    while (completed < issued) { rc = complete(completed++); if (rc) return rc; }
    if (timing) {
        timing->total_ms = now_ms() - t_begin; timing->gpu_ms = ov.gpu_last - ov.gpu_first;
        timing->host_overlapped_ms = ov.overlapped; timing->host_serial_ms = ov.serial;
    }
    return 0;
}

}  // namespace

extern "C" {

// Frees what the batch calls keep between calls (HIP streams, codecs, device scratch, page-locked staging buffers): of every device /
// of one.  Waits for a call that is running on a device it releases.
void divans_batch_release(void) { registry().release(-1); }
void divans_batch_release_device(int device) { if (device >= 0) registry().release(device); }

// Where the calling thread's time went in the last batch call (milliseconds): [0] CMD coders -- plans (compress) / container parsing
// (decompress), [1] staging into page-locked memory + enqueueing, [2] waiting for the GPU, [3] container assembly (compress) / copy-out
// (decompress), [4] final gather of the containers (compress).  A diagnostic: per calling thread, overwritten by that thread's next call.
void divans_batch_last_phases(double* out, int n) {
    for (int i = 0; i < n && i < 8; ++i) out[i] = g_phases[i];
}

// Host-only report on one container (divans_batch.h).
int divans_probe_container(const uint8_t* in, size_t n, int wire, divans_container_probe* out) {
    if (!out || (!in && n) || (wire != DIVANS_WIRE_HEAD && wire != DIVANS_WIRE_WASM_EXAMPLE)) return DIVANS_GPU_EINVAL;
    divans_host::divans_container_probe_fields f;
    divans_host::probe_container_host(in, n, wire, f);
    std::memset(out, 0, sizeof(*out));
    out->status = f.status; out->window = f.window; out->crc_ok = f.crc_ok; out->have_prediction_mode = f.have_pm;
    out->stopped_at_command = f.stopped_at; out->cmd_bytes = f.cmd_bytes; out->lit_bytes = f.lit_bytes; out->commands = f.commands;
    out->cmd_nibbles = f.cmd_nibbles; out->first_literal_length = f.first_literal_length; out->literal_bytes = f.literal_bytes;
    out->cfg = f.cfg;
    return 0;
}

}  // extern "C"
