// TEST HARNESS (see hostsim_device_stub.cpp): independent compressor / decompressor states on concurrent threads under ThreadSanitizer.
// The reference promises that distinct states are independent (c/divans/ffi.h, src/ffi/interface.rs:49-50: a state is `Send`, not `Sync`);
// the host code shares a codec cache (host_stream.cpp) and a per-thread error string between them.
//   hostsim_threads <file> <threads> <rounds>      exit 0 = every round trip exact, nothing for TSan to report
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/divans_ffi.h"

typedef std::vector<uint8_t> Bytes;

static bool round_trip(const Bytes& in, unsigned mixing, unsigned window, size_t buf_size) {
    DivansCompressorState* cs = divans_new_compressor();
    if (divans_set_option(cs, DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION, 0) != DIVANS_SUCCESS) return false;
    if (divans_set_option(cs, DIVANS_OPTION_DYNAMIC_CONTEXT_MIXING, mixing) != DIVANS_SUCCESS) return false;
    if (divans_set_option(cs, DIVANS_OPTION_WINDOW_SIZE, window) != DIVANS_SUCCESS) return false;
    Bytes coded, buf(buf_size);
    size_t off = 0;
    while (off < in.size()) {
        size_t ro = 0, wo = 0;
        if (divans_encode(cs, in.data() + off, in.size() - off, &ro, buf.data(), buf.size(), &wo) == DIVANS_FAILURE) return false;
        off += ro; coded.insert(coded.end(), buf.begin(), buf.begin() + wo);
    }
    for (;;) {
        size_t wo = 0;
        const DivansResult r = divans_encode_flush(cs, buf.data(), buf.size(), &wo);
        if (r == DIVANS_FAILURE) return false;
        coded.insert(coded.end(), buf.begin(), buf.begin() + wo);
        if (r == DIVANS_SUCCESS) break;
    }
    divans_free_compressor(cs);
    DivansDecompressorState* ds = divans_new_decompressor();
    Bytes back; off = 0;
    for (;;) {
        size_t ro = 0, wo = 0;
        const size_t feed = coded.size() - off < 5000 ? coded.size() - off : 5000;
        const DivansResult r = divans_decode(ds, coded.data() + off, feed, &ro, buf.data(), buf.size(), &wo);
        if (r == DIVANS_FAILURE) return false;
        off += ro; back.insert(back.end(), buf.begin(), buf.begin() + wo);
        if (r == DIVANS_SUCCESS) break;
        if (r == DIVANS_NEEDS_MORE_INPUT && feed == 0) return false;
    }
    divans_free_decompressor(ds);
    // a damaged copy on the same thread: the failure path (error strings) runs concurrently too
    if (coded.size() > 64) {
        coded[coded.size() / 2] ^= 0x20;
        ds = divans_new_decompressor();
        size_t ro = 0, wo = 0; Bytes big(in.size() + 65536);
        const DivansResult r = divans_decode(ds, coded.data(), coded.size(), &ro, big.data(), big.size(), &wo);
        divans_free_decompressor(ds);
        if (r == DIVANS_SUCCESS) return false;
    }
    return back == in;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    Bytes data; { uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + n); }
    std::fclose(f);
    const int threads = std::atoi(argv[2]), rounds = std::atoi(argv[3]);
    std::atomic<int> failures{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (int r = 0; r < rounds; ++r) {
                const size_t from = (size_t)(t * 7919 + r * 104729) % (data.size() / 2), len = 1000 + (size_t)(t * 3571 + r * 7907) % (data.size() / 2);
                const Bytes piece(data.begin() + from, data.begin() + from + len);
                // odd threads flip between the two literal configurations, so the shared codec cache keeps being handed over and rebuilt
                if (!round_trip(piece, (t + r) % 2 ? 2u : 0u, 10u + (unsigned)((t + r) % 7), 777u + 1000u * (unsigned)(r % 5))) failures++;
            }
        });
    for (auto& th : pool) th.join();
    std::printf("%d threads x %d rounds: %d failures\n", threads, rounds, failures.load());
    return failures.load() ? 1 : 0;
}
