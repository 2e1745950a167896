// batch.cpp -- many complete .divans streams per call (include/divans_batch.h): the LIT coder of every stream on the GPU,
// the CMD coders and the framing on host threads, the two overlapped (the reference overlaps
// the same halves of ONE stream with a worker thread, src/parallel_decompressor.rs:55-141, src/threading.rs:88-100).
#include "../../include/divans_batch.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

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

template <typename F>
void parallel_for(size_t n, int threads, F&& body) {
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(n, 1));
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) body(i); };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

struct DeviceBuf {
    void* p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() const { return (T*)p; }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    template <typename T> T* as() const { return (T*)p; }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault); }
};
struct StreamGuard { hipStream_t s = nullptr; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } };
struct CodecGuard { divans_gpu_codec* c = nullptr; ~CodecGuard() { if (c) divans_gpu_codec_destroy(c); } };

#define HIP_OR_FAIL(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_last_error(DIVANS_GPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

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

int divans_batch_compress(const divans_batch_options* opt, const uint8_t* const* inputs, const size_t* sizes, size_t n_streams,
                          uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing) {
    if (!opt || (!inputs && n_streams) || (!sizes && n_streams) || !out || !out_offsets || !out_sizes) return set_last_error(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    if (n_streams >= (1u << 24)) return set_last_error(DIVANS_GPU_EINVAL, "too many streams in one batch");
    const double t_begin = now_ms();
    const divans_host::StreamOptions so = to_stream_options(*opt);
    size_t longest = 0, total_in = 0;
    for (size_t i = 0; i < n_streams; ++i) {
        if (sizes[i] > 0x7fffffffu) return set_last_error(DIVANS_GPU_EINVAL, "stream too long");
        longest = std::max(longest, sizes[i]); total_in += sizes[i];
    }
    // the LIT configuration follows from the options alone (the PredictionMode of the internal compressor)
    divans_host::StreamPlan probe;
    int rc = divans_host::plan_stream(so, 0, nullptr, probe);
    if (rc) return set_last_error(rc, "options cannot be coded");
    HIP_OR_FAIL(hipSetDevice(opt->device));
    StreamGuard stream; HIP_OR_FAIL(hipStreamCreate(&stream.s));
    CodecGuard codec;
    const uint32_t max_len = (uint32_t)std::max<size_t>(longest, 16);
    rc = divans_gpu_codec_create(&codec.c, &probe.cfg, opt->device, stream.s, max_len);
    if (rc) return rc;
    {   // a small batch does not need the full persistent grid's worth of CDF tables
        divans_gpu_info info;
        if (divans_gpu_codec_info(codec.c, &info) == 0) (void)divans_gpu_codec_set_geometry(codec.c, std::max<uint32_t>(1, std::min<uint32_t>(info.blocks, (uint32_t)((n_streams + 15) / 16))), 0xffffffffu);
    }
    const uint32_t max_chunks = (uint32_t)std::max<uint64_t>(1, (2ull * max_len + 65535ull) / 65536ull);
    const uint64_t slot = divans_gpu_lit_encode_bound(max_len);
    // ---- GPU half: enqueue everything, no host synchronisation until the CMD coders are done -------------------------
    const double t_gpu0 = now_ms();
    PinnedBuf h_in, h_off, h_sz, h_packed, h_poff, h_psz, h_chunks, h_total;
    DeviceBuf d_in, d_off, d_sz, d_slots, d_ooff, d_osz, d_packed, d_poff, d_total, d_chunks;
    HIP_OR_FAIL(h_in.alloc(total_in + 64)); HIP_OR_FAIL(h_off.alloc(8 * n_streams)); HIP_OR_FAIL(h_sz.alloc(4 * n_streams));
    HIP_OR_FAIL(h_poff.alloc(8 * n_streams)); HIP_OR_FAIL(h_psz.alloc(4 * n_streams)); HIP_OR_FAIL(h_total.alloc(8));
    HIP_OR_FAIL(h_chunks.alloc(4ull * n_streams * max_chunks));
    {
        uint64_t* off = h_off.as<uint64_t>(); uint32_t* sz = h_sz.as<uint32_t>(); uint8_t* dst = h_in.as<uint8_t>();
        uint64_t pos = 0;
        for (size_t i = 0; i < n_streams; ++i) { off[i] = pos; sz[i] = (uint32_t)sizes[i]; pos += sizes[i]; }
        parallel_for(n_streams, opt->host_threads, [&](size_t i) { if (sizes[i]) std::memcpy(dst + off[i], inputs[i], sizes[i]); });
    }
    HIP_OR_FAIL(d_in.alloc(total_in + 64)); HIP_OR_FAIL(d_off.alloc(8 * n_streams)); HIP_OR_FAIL(d_sz.alloc(4 * n_streams));
    HIP_OR_FAIL(d_slots.alloc(slot * n_streams + 64)); HIP_OR_FAIL(d_ooff.alloc(8 * n_streams)); HIP_OR_FAIL(d_osz.alloc(4 * n_streams));
    HIP_OR_FAIL(d_packed.alloc(slot * n_streams + 64)); HIP_OR_FAIL(d_poff.alloc(8 * n_streams)); HIP_OR_FAIL(d_total.alloc(8));
    HIP_OR_FAIL(d_chunks.alloc(4ull * n_streams * max_chunks));
    HIP_OR_FAIL(hipMemcpyAsync(d_in.p, h_in.p, total_in, hipMemcpyHostToDevice, stream.s));
    HIP_OR_FAIL(hipMemcpyAsync(d_off.p, h_off.p, 8 * n_streams, hipMemcpyHostToDevice, stream.s));
    HIP_OR_FAIL(hipMemcpyAsync(d_sz.p, h_sz.p, 4 * n_streams, hipMemcpyHostToDevice, stream.s));
    HIP_OR_FAIL(hipMemsetAsync(d_chunks.p, 0, 4ull * n_streams * max_chunks, stream.s));
    rc = divans_gpu_lit_encode_batch_chunks(codec.c, d_in.as<uint8_t>(), d_off.as<uint64_t>(), d_sz.as<uint32_t>(), max_len, (uint32_t)n_streams,
                                            d_slots.as<uint8_t>(), slot, d_ooff.as<uint64_t>(), d_osz.as<uint32_t>(), d_chunks.as<uint32_t>(), max_chunks);
    if (rc) return rc;
    rc = divans_gpu_pack_streams(codec.c, d_slots.as<uint8_t>(), d_ooff.as<uint64_t>(), d_osz.as<uint32_t>(), (uint32_t)n_streams,
                                 d_packed.as<uint8_t>(), d_poff.as<uint64_t>(), d_total.as<uint64_t>());
    if (rc) return rc;
    HIP_OR_FAIL(hipMemcpyAsync(h_poff.p, d_poff.p, 8 * n_streams, hipMemcpyDeviceToHost, stream.s));
    HIP_OR_FAIL(hipMemcpyAsync(h_psz.p, d_osz.p, 4 * n_streams, hipMemcpyDeviceToHost, stream.s));
    HIP_OR_FAIL(hipMemcpyAsync(h_chunks.p, d_chunks.p, 4ull * n_streams * max_chunks, hipMemcpyDeviceToHost, stream.s));
    HIP_OR_FAIL(hipMemcpyAsync(h_total.p, d_total.p, 8, hipMemcpyDeviceToHost, stream.s));
    // ---- host half, overlapped: the CMD coder of every stream (it sees lengths and options only, never the data) --------
    const double t_host0 = now_ms();
    std::map<size_t, std::unique_ptr<divans_host::StreamPlan>> plans;
    for (size_t i = 0; i < n_streams; ++i) plans.emplace(sizes[i], nullptr);
    std::vector<size_t> distinct; for (auto& kv : plans) distinct.push_back(kv.first);
    std::atomic<int> plan_rc{0};
    parallel_for(distinct.size(), opt->host_threads, [&](size_t k) {
        auto p = std::make_unique<divans_host::StreamPlan>();
        const int r = divans_host::plan_stream(so, distinct[k], nullptr, *p);
        if (r) plan_rc = r;
        plans[distinct[k]] = std::move(p);     // the map's nodes exist already: no rebalancing, distinct keys per thread
    });
    const double t_host1 = now_ms();
    if (plan_rc) return set_last_error(plan_rc, "a stream's command stream cannot be coded");
    // ---- join --------------------------------------------------------------------------------------------------------
    HIP_OR_FAIL(hipStreamSynchronize(stream.s));
    uint32_t status = 0;
    if (divans_gpu_codec_status(codec.c, &status) || status) return set_last_error(DIVANS_GPU_EINVAL, "the literal coder reported an invalid model state");
    const uint64_t packed_total = *h_total.as<uint64_t>();
    HIP_OR_FAIL(h_packed.alloc(packed_total + 64));
    HIP_OR_FAIL(hipMemcpy(h_packed.p, d_packed.p, packed_total, hipMemcpyDeviceToHost));
    const double t_gpu1 = now_ms();
    // ---- framing: Mux replay, EOF marker, CRC trailer per stream ---------------------------------------------------------
    std::vector<std::vector<uint8_t>> results(n_streams);
    std::atomic<int> asm_rc{0};
    const size_t call_buffer = opt->call_buffer_size ? opt->call_buffer_size : 65536;
    parallel_for(n_streams, opt->host_threads, [&](size_t i) {
        const divans_host::StreamPlan& p = *plans[sizes[i]];
        const int r = divans_host::assemble_container(p, h_packed.as<uint8_t>() + h_poff.as<uint64_t>()[i], h_psz.as<uint32_t>()[i],
                                                      h_chunks.as<uint32_t>() + i * max_chunks, call_buffer, results[i]);
        if (r) asm_rc = r;
    });
    if (asm_rc) return set_last_error(asm_rc, "container assembly failed");
    size_t pos = 0;
    for (size_t i = 0; i < n_streams; ++i) { out_offsets[i] = pos; out_sizes[i] = results[i].size(); pos += results[i].size(); }
    if (pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
    parallel_for(n_streams, opt->host_threads, [&](size_t i) { std::memcpy(out + out_offsets[i], results[i].data(), results[i].size()); });
    const double t_end = now_ms();
    if (timing) {
        timing->total_ms = t_end - t_begin; timing->gpu_ms = t_gpu1 - t_gpu0;
        timing->host_overlapped_ms = t_host1 - t_host0; timing->host_serial_ms = t_end - t_gpu1;
    }
    return 0;
}

int divans_batch_decompress(const divans_batch_options* opt, const uint8_t* const* containers, const size_t* sizes, size_t n_streams,
                            uint8_t* out, size_t out_cap, size_t* out_offsets, size_t* out_sizes, divans_batch_timing* timing) {
    if (!opt || (!containers && n_streams) || (!sizes && n_streams) || !out || !out_offsets || !out_sizes) return set_last_error(DIVANS_GPU_EINVAL, "null argument");
    if (n_streams == 0) return 0;
    const double t_begin = now_ms();
    HIP_OR_FAIL(hipSetDevice(opt->device));
    StreamGuard stream; HIP_OR_FAIL(hipStreamCreate(&stream.s));
    // Pipeline over slices of the batch: while the GPU decodes the LIT streams of slice k, host threads parse slice k + 1
    // (framing, CRC, CMD coder -> decoded sizes and LIT configuration, which the GPU launch needs).
    // a slice should still fill the persistent decode grid (28 672 streams on MI355X): small batches are one slice
    const size_t n_slices = std::max<size_t>(1, std::min<size_t>(4, n_streams / 16384));
    std::vector<divans_host::ParsedStream> parsed(n_streams);
    std::vector<int> status(n_streams, 0);
    double host_overlapped = 0, host_serial = 0, gpu_ms = 0;
    auto parse_slice = [&](size_t k) {
        const size_t b = k * n_streams / n_slices, e = (k + 1) * n_streams / n_slices;
        parallel_for(e - b, opt->host_threads, [&](size_t j) {
            const size_t i = b + j;
            status[i] = (int)divans_host::parse_container_host(containers[i], sizes[i], opt->skip_crc != 0, (size_t)1 << 30, parsed[i], nullptr);
        });
    };
    struct Group {
        divans_lit_config cfg; std::vector<size_t> members;
        CodecGuard codec; PinnedBuf h_in, h_off, h_sz, h_ooff, h_osz, h_out; DeviceBuf d_in, d_off, d_sz, d_ooff, d_osz, d_out;
        size_t out_bytes = 0;
    };
    size_t pos = 0;
    double t0 = now_ms();
    parse_slice(0);
    host_serial += now_ms() - t0;
    for (size_t k = 0; k < n_slices; ++k) {
        const size_t b = k * n_streams / n_slices, e = (k + 1) * n_streams / n_slices;
        for (size_t i = b; i < e; ++i)
            if (status[i] != divans_host::PARSE_OK) { out_sizes[i] = (size_t)-1; return set_last_error(DIVANS_GPU_ECORRUPT, "container " + std::to_string(i) + " is truncated, corrupt or not a literal-only stream"); }
        // group the slice by LIT configuration (one codec = one configuration)
        std::vector<std::unique_ptr<Group>> groups;
        for (size_t i = b; i < e; ++i) {
            out_offsets[i] = pos; out_sizes[i] = parsed[i].total; pos += parsed[i].total;
            if (parsed[i].total == 0) continue;
            Group* g = nullptr;
            for (auto& q : groups) if (std::memcmp(&q->cfg, &parsed[i].cfg, sizeof(divans_lit_config)) == 0) { g = q.get(); break; }
            if (!g) { groups.emplace_back(new Group()); g = groups.back().get(); g->cfg = parsed[i].cfg; }
            g->members.push_back(i);
        }
        if (pos > out_cap) return set_last_error(DIVANS_GPU_ECAP, "output buffer too small");
        const double tg0 = now_ms();
        for (auto& gp : groups) {
            Group& g = *gp;
            const size_t m = g.members.size();
            size_t in_bytes = 0, longest = 0; g.out_bytes = 0;
            for (size_t i : g.members) { in_bytes += parsed[i].lit.size(); g.out_bytes += parsed[i].total; longest = std::max(longest, parsed[i].total); }
            HIP_OR_FAIL(g.h_in.alloc(in_bytes + 64)); HIP_OR_FAIL(g.h_off.alloc(8 * m)); HIP_OR_FAIL(g.h_sz.alloc(4 * m));
            HIP_OR_FAIL(g.h_ooff.alloc(8 * m)); HIP_OR_FAIL(g.h_osz.alloc(4 * m)); HIP_OR_FAIL(g.h_out.alloc(g.out_bytes + 64));
            HIP_OR_FAIL(g.d_in.alloc(in_bytes + 128)); HIP_OR_FAIL(g.d_off.alloc(8 * m)); HIP_OR_FAIL(g.d_sz.alloc(4 * m));
            HIP_OR_FAIL(g.d_ooff.alloc(8 * m)); HIP_OR_FAIL(g.d_osz.alloc(4 * m)); HIP_OR_FAIL(g.d_out.alloc(g.out_bytes + 64));
            uint64_t ip = 0, op = 0;
            for (size_t j = 0; j < m; ++j) {
                const divans_host::ParsedStream& ps = parsed[g.members[j]];
                g.h_off.as<uint64_t>()[j] = ip; g.h_sz.as<uint32_t>()[j] = (uint32_t)ps.lit.size();
                g.h_ooff.as<uint64_t>()[j] = op; g.h_osz.as<uint32_t>()[j] = (uint32_t)ps.total;
                std::memcpy(g.h_in.as<uint8_t>() + ip, ps.lit.data(), ps.lit.size());
                ip += ps.lit.size(); op += ps.total;
            }
            int rc = divans_gpu_codec_create(&g.codec.c, &g.cfg, opt->device, stream.s, (uint32_t)std::max<size_t>(longest, 16));
            if (rc) return rc;
            {
                divans_gpu_info info;
                if (divans_gpu_codec_info(g.codec.c, &info) == 0) (void)divans_gpu_codec_set_geometry(g.codec.c, std::max<uint32_t>(1, std::min<uint32_t>(info.blocks, (uint32_t)((m + 15) / 16))), 0xffffffffu);
            }
            HIP_OR_FAIL(hipMemsetAsync(g.d_in.as<uint8_t>() + in_bytes, 0, 64, stream.s));
            HIP_OR_FAIL(hipMemcpyAsync(g.d_in.p, g.h_in.p, in_bytes, hipMemcpyHostToDevice, stream.s));
            HIP_OR_FAIL(hipMemcpyAsync(g.d_off.p, g.h_off.p, 8 * m, hipMemcpyHostToDevice, stream.s));
            HIP_OR_FAIL(hipMemcpyAsync(g.d_sz.p, g.h_sz.p, 4 * m, hipMemcpyHostToDevice, stream.s));
            HIP_OR_FAIL(hipMemcpyAsync(g.d_ooff.p, g.h_ooff.p, 8 * m, hipMemcpyHostToDevice, stream.s));
            HIP_OR_FAIL(hipMemcpyAsync(g.d_osz.p, g.h_osz.p, 4 * m, hipMemcpyHostToDevice, stream.s));
            rc = divans_gpu_lit_decode_batch(g.codec.c, g.d_in.as<uint8_t>(), g.d_off.as<uint64_t>(), g.d_sz.as<uint32_t>(), (uint32_t)m,
                                             g.d_out.as<uint8_t>(), g.d_ooff.as<uint64_t>(), g.d_osz.as<uint32_t>(), (uint32_t)std::max<size_t>(longest, 16));
            if (rc) return rc;
            HIP_OR_FAIL(hipMemcpyAsync(g.h_out.p, g.d_out.p, g.out_bytes, hipMemcpyDeviceToHost, stream.s));
        }
        // overlapped host work: the next slice's containers
        const double th0 = now_ms();
        if (k + 1 < n_slices) parse_slice(k + 1);
        host_overlapped += now_ms() - th0;
        HIP_OR_FAIL(hipStreamSynchronize(stream.s));
        gpu_ms += now_ms() - tg0;
        const double ts0 = now_ms();
        for (auto& gp : groups) {
            Group& g = *gp;
            uint32_t st = 0;
            if (divans_gpu_codec_status(g.codec.c, &st)) return DIVANS_GPU_EHIP;
            if (st & DIVANS_GPU_STATUS_BAD_STREAM) return set_last_error(DIVANS_GPU_ECORRUPT, "a LIT stream failed the decoder's integrity check");
            parallel_for(g.members.size(), opt->host_threads, [&](size_t j) {
                const size_t i = g.members[j];
                std::memcpy(out + out_offsets[i], g.h_out.as<uint8_t>() + g.h_ooff.as<uint64_t>()[j], parsed[i].total);
            });
        }
        for (size_t i = b; i < e; ++i) { std::vector<uint8_t>().swap(parsed[i].lit); }
        host_serial += now_ms() - ts0;
    }
    if (timing) { timing->total_ms = now_ms() - t_begin; timing->gpu_ms = gpu_ms; timing->host_overlapped_ms = host_overlapped; timing->host_serial_ms = host_serial; }
    return 0;
}

}  // extern "C"
