// TEST HARNESS (see hostsim_device_stub.cpp, fakehip/hip/hip_runtime.h): the many-containers interface (include/divans_batch.h,
// divans_amd/csrc/batch.cpp: length classes, slices on lanes, persistent thread pool, plans / parsing under the "GPU work", container
// assembly) with the oracle standing in for the literal kernels, under AddressSanitizer + UBSan or ThreadSanitizer.
//
//   hostsim_batch <file> <seed> <rounds> [largest batch, default 700] [devices, default 1]
//
// devices = 2: the rounds run on two threads at once, one per stand-in device (fakehip knows two): the per-device lane pools of batch.cpp
//  -- concurrent calls on different devices, divans_batch_release / _release_device from one thread while the other is
// inside a call -- under ThreadSanitizer.
// devices = -D (D = 2 .. 8): D stand-in devices; one thread makes every call with divans_batch_options::device = DIVANS_BATCH_ALL_DEVICES --
// the call cuts its streams into D contiguous ranges and drives every device from its own thread -- and compares
// what comes back with D one-device calls on the D ranges, while a second thread keeps device D - 1 busy with calls of its own.
//
// Per round: a batch of streams of mixed lengths (empty, tiny, around the 64 KiB class bound, now and then several hundred KB; every
// fourth round incompressible ones) under
// random options and a random number of host threads -> divans_batch_compress; containers compared with the oracle's
// (orc_stream_compress_raw, same options, one call, same call buffer); a second batch under other options is mixed in and everything
// goes through divans_batch_decompress (grouping by configuration and length class) and must come back exact; then a damaged copy of one
// container -- the call must fail and name it -- an output buffer one byte short (DIVANS_GPU_ECAP), and now and then
// divans_batch_release() between calls.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/divans_batch.h"
extern "C" {
#include "../../oracle/divans_oracle.h"
}

static thread_local uint64_t rng_state;
static thread_local int t_device = 0;
static uint64_t rnd() { rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27; return rng_state * 2685821237909765ull; }
static size_t rnd_below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }
typedef std::vector<uint8_t> Bytes;

static divans_batch_options random_options() {
    static const divans_speed palette[6] = {{0x10, 0x2000}, {2, 1024}, {64, 16384}, {128, 16384}, {1, 16384}, {4, 1024}};
    divans_batch_options o; divans_batch_options_default(&o);
    o.window_size = 10 + (int)rnd_below(13);
    o.dynamic_context_mixing = (uint8_t)rnd_below(3);
    o.use_context_map = (uint8_t)(rnd() & 1);
    o.force_stride = (uint8_t)((rnd() & 1) ? rnd_below(9) : 9);
    o.has_prior_depth = (rnd() & 3) == 0; o.prior_depth = (uint8_t)rnd_below(3);
    o.has_literal_adaptation = (rnd() & 3) == 0;
    for (auto& sp : o.literal_adaptation) sp = palette[rnd_below(6)];
    o.call_buffer_size = (rnd() & 1) ? 65536u : (uint32_t)(1 + rnd_below(70000));
    o.host_threads = (int)rnd_below(9);          // 0 = all the process is granted
    o.device = t_device;
    return o;
}

static bool oracle_container(const divans_batch_options& b, const uint8_t* in, size_t n, Bytes& out) {
    orc_stream_options o; orc_stream_options_default(&o);
    o.window_size = b.window_size; o.dynamic_context_mixing = b.dynamic_context_mixing;
    o.prior_depth = b.has_prior_depth ? b.prior_depth : 0; o.use_context_map = b.use_context_map; o.force_stride = b.force_stride;
    o.has_literal_adaptation = b.has_literal_adaptation;
    for (int i = 0; i < 4; ++i) { o.literal_adaptation[i].inc = b.literal_adaptation[i].inc; o.literal_adaptation[i].lim = b.literal_adaptation[i].lim; }
    o.call_buffer_size = b.call_buffer_size ? b.call_buffer_size : 65536;
    out.resize(2 * n + 70000);
    const size_t r = orc_stream_compress_raw(&o, in, n, out.data(), out.size());
    if (r == (size_t)-1) return false;
    out.resize(r);
    return true;
}

struct Batch { std::vector<const uint8_t*> ptr; std::vector<size_t> len; };

static Batch random_batch(const Bytes& data, size_t max_streams, size_t from = 0) {    // from: streams start at or after this offset of `data`
    Batch b;
    const size_t n = 1 + rnd_below(max_streams);
    for (size_t i = 0; i < n; ++i) {
        size_t len;
        switch (rnd() % 8) {
            case 0: len = rnd_below(3); break;
            case 1: len = 65536 - 2 + rnd_below(5); break;                      // around the first class bound
            case 2: len = (rnd() % 16 == 0) ? 100000 + rnd_below(data.size() - from - 100000) : rnd_below(70000); break;
            default: len = 1 + rnd_below(9000); break;
        }
        len = std::min(len, data.size() - from);
        b.ptr.push_back(data.data() + from + rnd_below(data.size() - from - len + 1)); b.len.push_back(len);
    }
    return b;
}

#define FAIL(...) do { std::fprintf(stderr, "device %d: ", t_device); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, " (%s)\n", divans_gpu_last_error()); return 1; } while (0)

static std::atomic<size_t> n_containers{0}, bytes_in{0};
static int g_fake_devices = 2;

// the rounds of one device (its own thread when there are two)
static int run_rounds(const Bytes& data, size_t noise_from, uint64_t seed, long rounds, size_t largest, int device) {
    rng_state = seed; t_device = device;
    for (long round = 0; round < rounds; ++round) {
        const divans_batch_options oa = random_options(), ob = random_options();
        const Batch a = random_batch(data, round % 4 == 0 ? largest : std::min<size_t>(60, largest), round % 4 == 2 ? noise_from : 0), b = random_batch(data, 20);
        std::vector<Bytes> containers;                      // of a, then of b
        for (int which = 0; which < 2; ++which) {
            const Batch& x = which ? b : a; const divans_batch_options& o = which ? ob : oa;
            size_t cap = 0; for (size_t l : x.len) cap += divans_batch_compress_bound(l);
            Bytes out(cap); std::vector<size_t> off(x.len.size()), sz(x.len.size());
            divans_batch_timing t;
            if (divans_batch_compress(&o, x.ptr.data(), x.len.data(), x.len.size(), out.data(), out.size(), off.data(), sz.data(), &t) != 0)
                FAIL("round %ld: divans_batch_compress failed", round);
            const size_t step = x.len.size() > 80 ? 7 : 1;                    // every container of a small batch, every seventh of a large one
            for (size_t i = 0; i < x.len.size(); ++i) {
                if (off[i] + sz[i] > out.size()) FAIL("round %ld: container %zu lies outside the output", round, i);
                if (i % step == 0) {
                    Bytes ref;
                    if (!oracle_container(o, x.ptr[i], x.len[i], ref)) FAIL("round %ld: the oracle failed", round);
                    if (ref.size() != sz[i] || std::memcmp(ref.data(), out.data() + off[i], sz[i]) != 0)
                        FAIL("round %ld: container %zu (%zu bytes, window %d, mixing %u, buffer %u) differs from the oracle's: %zu vs %zu bytes",
                             round, i, x.len[i], o.window_size, o.dynamic_context_mixing, o.call_buffer_size, sz[i], ref.size());
                }
                containers.emplace_back(out.begin() + off[i], out.begin() + off[i] + sz[i]);
            }
            // one byte short of what the containers need
            size_t total = 0; for (size_t s : sz) total += s;
            if (total > 0) {
                Bytes small(total - 1);
                if (divans_batch_compress(&o, x.ptr.data(), x.len.data(), x.len.size(), small.data(), small.size(), off.data(), sz.data(), nullptr) != DIVANS_GPU_ECAP)
                    FAIL("round %ld: a short output buffer was not refused", round);
            }
            n_containers += x.len.size(); for (size_t l : x.len) bytes_in += l;
        }
        if (device == DIVANS_BATCH_ALL_DEVICES) {
            // one call over all devices == one call per device on the contiguous ranges [n r / D, n (r + 1) / D), container for container
            const size_t n = a.len.size(), D = std::min<size_t>((size_t)g_fake_devices, n);
            for (size_t r = 0; r < D; ++r) {
                const size_t b0 = n * r / D, e0 = n * (r + 1) / D;
                divans_batch_options o1 = oa; o1.device = (int)r;
                size_t cap = 0; for (size_t i = b0; i < e0; ++i) cap += divans_batch_compress_bound(a.len[i]);
                Bytes out(cap + 1); std::vector<size_t> off(e0 - b0), sz(e0 - b0);
                if (divans_batch_compress(&o1, a.ptr.data() + b0, a.len.data() + b0, e0 - b0, out.data(), out.size(), off.data(), sz.data(), nullptr) != 0)
                    FAIL("round %ld: the one-device call on range %zu failed", round, r);
                for (size_t i = b0; i < e0; ++i)
                    if (sz[i - b0] != containers[i].size() || std::memcmp(out.data() + off[i - b0], containers[i].data(), sz[i - b0]) != 0)
                        FAIL("round %ld: container %zu of the all-devices call differs from device %zu's own call", round, i, r);
            }
        }
        if (round % 3 == 1) { if ((round / 3) & 1) divans_batch_release_device(device < 0 ? 0 : device); else divans_batch_release(); }    // (the other thread may be inside a call)
        // both batches interleaved through one decompress call
        std::vector<size_t> order;                               // index into `containers`
        { size_t ia = 0, ib = 0; while (ia < a.len.size() || ib < b.len.size()) { if (ib < b.len.size() && (ia >= a.len.size() || (rnd() & 3) == 0)) order.push_back(a.len.size() + ib++); else order.push_back(ia++); } }
        std::vector<const uint8_t*> cp; std::vector<size_t> cl; size_t total = 0;
        auto original = [&](size_t k, const uint8_t*& p, size_t& l) { if (k < a.len.size()) { p = a.ptr[k]; l = a.len[k]; } else { p = b.ptr[k - a.len.size()]; l = b.len[k - a.len.size()]; } };
        for (size_t k : order) { cp.push_back(containers[k].data()); cl.push_back(containers[k].size()); const uint8_t* p; size_t l; original(k, p, l); total += l; }
        divans_batch_options od; divans_batch_options_default(&od); od.host_threads = (int)rnd_below(9); od.device = device;
        Bytes back(total + 1); std::vector<size_t> off(order.size()), sz(order.size());
        if (divans_batch_decompress(&od, cp.data(), cl.data(), order.size(), back.data(), back.size(), off.data(), sz.data(), nullptr) != 0)
            FAIL("round %ld: divans_batch_decompress failed", round);
        for (size_t j = 0; j < order.size(); ++j) {
            const uint8_t* p; size_t l; original(order[j], p, l);
            if (sz[j] != l || off[j] + l > back.size() || std::memcmp(back.data() + off[j], p, l) != 0) FAIL("round %ld: stream %zu came back wrong", round, j);
        }
        if (total > 0 && divans_batch_decompress(&od, cp.data(), cl.data(), order.size(), back.data(), total - 1, off.data(), sz.data(), nullptr) != DIVANS_GPU_ECAP)
            FAIL("round %ld: a short decompress buffer was not refused", round);
        // a damaged container: the call fails and out_sizes names it (or an earlier slice refused it while parsing)
        {
            const size_t victim = rnd_below(order.size());
            Bytes dmg = containers[order[victim]];
            if (dmg.size() > 30) {
                dmg[16 + rnd_below(dmg.size() - 24)] ^= (uint8_t)(1u << (rnd() % 8));
                cp[victim] = dmg.data();
                od.skip_crc = (uint8_t)(rnd() & 1);
                std::fill(sz.begin(), sz.end(), 0);
                const int rc = divans_batch_decompress(&od, cp.data(), cl.data(), order.size(), back.data(), back.size(), off.data(), sz.data(), nullptr);
                if (rc == 0) {
                    // only acceptable when the flipped bit did not matter to anything that is checked (skip_crc and a harmless spot): the bytes must be right then
                    const uint8_t* p; size_t l; original(order[victim], p, l);
                    if (!od.skip_crc || sz[victim] != l || std::memcmp(back.data() + off[victim], p, l) != 0) FAIL("round %ld: a damaged container was accepted", round);
                } else {
                    for (size_t j = 0; j < order.size(); ++j) if (sz[j] == (size_t)-1 && j != victim) FAIL("round %ld: the call blames container %zu, damaged was %zu", round, j, victim);
                }
                cp[victim] = containers[order[victim]].data();
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: hostsim_batch <file> <seed> <rounds> [largest] [devices]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    Bytes data; { uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + n); }
    std::fclose(f);
    if (data.size() < 200000) { std::fprintf(stderr, "the input file should hold at least 200 000 bytes\n"); return 2; }
    const uint64_t seed = std::strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull + 1;
    rng_state = seed;
    const size_t noise_from = data.size();                      // incompressible tail: packed streams larger than the staging guess (5/8 of the input)
    for (size_t i = 0; i < 300000; ++i) data.push_back((uint8_t)rnd());
    const long rounds = std::strtol(argv[3], nullptr, 0);
    const size_t largest = argc > 4 ? (size_t)std::strtoul(argv[4], nullptr, 0) : 700;
    const int devices = argc > 5 ? std::atoi(argv[5]) : 1;
    if (devices < 0) { g_fake_devices = -devices; setenv("FAKEHIP_DEVICES", std::to_string(-devices).c_str(), 1); }      // before the first HIP call
    // the oracle is single-threaded test infrastructure with lazily built tables (CRC-32C, context lookups): build them before the threads start
    { Bytes warm; divans_batch_options wo; divans_batch_options_default(&wo); wo.dynamic_context_mixing = 2; (void)oracle_container(wo, data.data(), 3000, warm); }
    int rc = 0;
    if (devices < 0) {
        std::atomic<int> failed{0};
        std::thread other([&]() { if (run_rounds(data, noise_from, seed ^ 0x9e3779b9u, rounds, std::min<size_t>(largest, 40), -devices - 1)) failed = 1; });
        if (run_rounds(data, noise_from, seed ^ 0x5bd1e995u, rounds, largest, DIVANS_BATCH_ALL_DEVICES)) failed = 1;
        other.join();
        rc = failed.load();
    }
    else if (devices <= 1) rc = run_rounds(data, noise_from, seed ^ 0x5bd1e995u, rounds, largest, 0);
    else {
        std::atomic<int> failed{0};
        std::vector<std::thread> ts;
        for (int d = 0; d < 2; ++d) ts.emplace_back([&, d]() { if (run_rounds(data, noise_from, seed ^ (0x5bd1e995u * (uint64_t)(d + 1)), rounds, largest, d)) failed = 1; });
        for (auto& t : ts) t.join();
        rc = failed.load();
    }
    divans_batch_release();
    if (rc) return rc;
    std::printf("%ld rounds on %d device(s)%s: %zu containers, %zu bytes, all equal to the oracle's and back\n", rounds, devices < 0 ? -devices : (devices > 1 ? 2 : 1),
                devices < 0 ? " through all-devices calls (+ a thread of its own on the last device)" : "", n_containers.load(), bytes_in.load());
    return 0;
}
