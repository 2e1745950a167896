// TEST HARNESS (see hostsim_device_stub.cpp): the per-stream C ABI's host logic under AddressSanitizer / UBSan.
//
//   hostsim_fuzz <file> <seed> <iterations> [selector=value ...]
//   hostsim_fuzz ir <file.ir> <seed> <iterations>
//   hostsim_fuzz plan <file> <seed> <iterations>
//
// 1. compresses <file> through divans_encode / divans_encode_flush with seeded random input pieces and output buffer sizes (1 byte ..
//    64 KiB), decompresses it the same way and compares;
// 2. `iterations` times: damages the container (bit flips, overwritten runs, truncation, deleted / duplicated / inserted spans, a splice
//    of two containers) and decodes it with random piece and buffer sizes, with and without the CRC check.  The decoder may answer
//    DIVANS_FAILURE, may ask for input that does not exist, or -- when the damage was harmless -- succeed; what it may not do is touch
//    memory it does not own (the sanitizers abort), spin without progress, hand out more than max_output, or report success with wrong
//    bytes while the CRC is checked.
// 3. every damaged container also goes through divans_host::parse_container_host, the whole-container parser of the batch interface
//    (divans_amd/csrc/batch.cpp), which must agree with the streaming decoder on what is acceptable.
// `ir` mode: the textual command IR (include/divans_ir.h) -- the file must parse and expand; damaged copies (bytes flipped, tokens
// deleted / duplicated, numbers replaced by huge ones) may be refused or accepted, within bounds.
// `plan` mode: the two-phase container builder of the batch interface (divans_host::plan_stream -- everything that needs no literal
// data -- and assemble_container, divans_amd/csrc/batch.cpp's host half) with random lengths, options, call buffers and call
// patterns; the literal bytes come from the oracle's literal coder under the plan's configuration, and the container must be the
// oracle's (orc_stream_compress_raw) byte for byte.
// exit 0 = all held; prints one summary line.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/divans_ffi.h"
#include "../../include/divans_ir.h"
#include "../../divans_amd/csrc/host_stream.h"
extern "C" {
#include "../../oracle/divans_oracle.h"
}

static uint64_t rng_state;
static uint64_t rnd() {                       // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return rng_state * 2685821237909765ull;
}
static size_t rnd_below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }
static size_t rnd_size() {                   // skewed: tiny, small and buffer-sized values all occur
    switch (rnd() % 4) {
        case 0: return 1 + rnd_below(16);
        case 1: return 1 + rnd_below(700);
        case 2: return 1 + rnd_below(8192);
        default: return 1 + rnd_below(65536);
    }
}

typedef std::vector<uint8_t> Bytes;

static bool compress(const Bytes& in, const std::vector<std::pair<unsigned, unsigned>>& opts, bool fixed_sizes, Bytes& out) {
    DivansCompressorState* st = divans_new_compressor();
    for (auto& o : opts) if (divans_set_option(st, (DivansOptionSelect)o.first, o.second) != DIVANS_SUCCESS) { std::fprintf(stderr, "option %u=%u refused\n", o.first, o.second); return false; }
    Bytes buf(65536);
    size_t off = 0;
    while (off < in.size()) {
        const size_t piece = fixed_sizes ? in.size() - off : std::min(in.size() - off, rnd_size());
        size_t done = 0; int idle = 0;
        while (done < piece) {
            const size_t cap = fixed_sizes ? buf.size() : rnd_size();
            size_t ro = 0, wo = 0;
            const DivansResult r = divans_encode(st, in.data() + off + done, piece - done, &ro, buf.data(), cap, &wo);
            if (r == DIVANS_FAILURE || ro > piece - done || wo > cap) { std::fprintf(stderr, "encode failed (%d)\n", (int)r); return false; }
            done += ro; out.insert(out.end(), buf.begin(), buf.begin() + wo);
            idle = (ro | wo) ? 0 : idle + 1;
            if (idle > 4) { std::fprintf(stderr, "encode makes no progress\n"); return false; }
        }
        off += piece;
    }
    for (int idle = 0;;) {
        const size_t cap = fixed_sizes ? buf.size() : rnd_size();
        size_t wo = 0;
        const DivansResult r = divans_encode_flush(st, buf.data(), cap, &wo);
        if (r == DIVANS_FAILURE || wo > cap) { std::fprintf(stderr, "flush failed\n"); return false; }
        out.insert(out.end(), buf.begin(), buf.begin() + wo);
        if (r == DIVANS_SUCCESS) break;
        idle = wo ? 0 : idle + 1;
        if (idle > 4) { std::fprintf(stderr, "flush makes no progress\n"); return false; }
    }
    divans_free_compressor(st);
    return true;
}

enum Outcome { OK = 0, FAILED = 1, WANTS_INPUT = 2, STUCK = 3, OVERRUN = 4 };

static Outcome decompress(const Bytes& coded, bool skip_crc, bool fixed_sizes, size_t limit, Bytes& out) {
    struct CAllocator none = {nullptr, nullptr, nullptr};
    DivansDecompressorState* st = divans_new_decompressor_with_custom_alloc(none, skip_crc ? 1 : 0, 0);
    divans_decompressor_set_max_output_size(st, limit);       // a stream that claims more is refused, not followed
    Bytes buf(65536);
    size_t off = 0; int idle = 0; Outcome res = STUCK;
    for (;;) {
        const size_t feed = std::min(coded.size() - off, fixed_sizes ? (size_t)100000 : rnd_size());
        const size_t cap = fixed_sizes ? buf.size() : rnd_size();
        size_t ro = 0, wo = 0;
        const DivansResult r = divans_decode(st, coded.data() + off, feed, &ro, buf.data(), cap, &wo);
        if (ro > feed || wo > cap) { res = OVERRUN; break; }
        off += ro; out.insert(out.end(), buf.begin(), buf.begin() + wo);
        if (out.size() > limit) { res = OVERRUN; break; }
        if (r == DIVANS_SUCCESS) { res = OK; break; }
        if (r == DIVANS_FAILURE) { res = FAILED; break; }
        if (r == DIVANS_NEEDS_MORE_INPUT && off == coded.size() && feed == 0) { res = WANTS_INPUT; break; }
        idle = (ro | wo) ? 0 : idle + 1;
        if (idle > 8) { res = STUCK; break; }
    }
    divans_free_decompressor(st);
    return res;
}

static void damage(Bytes& c, const Bytes& other) {
    const int kinds = 1 + (int)(rnd() % 3);
    for (int k = 0; k < kinds && !c.empty(); ++k) {
        const size_t at = (rnd() % 4 == 0) ? rnd_below(std::min<size_t>(c.size(), 64)) : rnd_below(c.size());   // the header and the first slices get their share
        switch (rnd() % 8) {
            case 0: c[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 1: { const size_t n = std::min(c.size() - at, 1 + rnd_below(32)); for (size_t i = 0; i < n; ++i) c[at + i] = (uint8_t)rnd(); break; }
            case 2: c.resize(at); break;
            case 3: { const size_t n = std::min(c.size() - at, 1 + rnd_below(300)); c.erase(c.begin() + at, c.begin() + at + n); break; }
            case 4: { const size_t n = std::min(c.size() - at, 1 + rnd_below(300)); Bytes d(c.begin() + at, c.begin() + at + n); c.insert(c.begin() + at, d.begin(), d.end()); break; }
            case 5: { Bytes d(1 + rnd_below(64)); for (auto& b : d) b = (uint8_t)rnd(); c.insert(c.begin() + at, d.begin(), d.end()); break; }
            case 6: { const size_t n = std::min(c.size() - at, 1 + rnd_below(8)); for (size_t i = 0; i < n; ++i) c[at + i] = (rnd() & 1) ? 0xff : 0x00; break; }
            default: if (!other.empty()) { const size_t from = rnd_below(other.size()); c.resize(at); c.insert(c.end(), other.begin() + from, other.end()); } break;
        }
    }
}

static int fuzz_ir(const char* path, long iterations) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return 2;
    std::string text; { char tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) text.append(tmp, n); }
    std::fclose(f);
    const size_t cap = (size_t)64 << 20;
    Bytes raw(cap), lit(cap); std::vector<divans_lit_segment> segs(1 << 20);
    auto run = [&](const std::string& t, bool must) -> int {
        divans_ir* ir = nullptr;
        const int rc = divans_ir_parse(t.data(), t.size(), &ir);
        if (rc) { if (ir) return 10; return must ? 11 : 0; }
        if (!ir) return 12;
        const size_t n = divans_ir_raw_size(ir);
        const int e = divans_ir_expand(ir, raw.data(), cap);
        if ((n <= cap) != (e == 0) && must) { divans_ir_free(ir); return 13; }
        const size_t nl = divans_ir_literal_size(ir), ns = divans_ir_num_segments(ir);
        const int l = divans_ir_literal_segments(ir, lit.data(), cap, segs.data(), segs.size());
        if (must && (l != 0 || nl > n || ns > divans_ir_num_commands(ir))) { divans_ir_free(ir); return 14; }
        divans_ir_options o; divans_ir_options_default(&o);
        divans_lit_config cfg;
        (void)divans_ir_lit_config(ir, &o, &cfg);
        (void)divans_ir_num_block_types(ir); (void)divans_ir_count(ir, DIVANS_IR_COPY);
        divans_ir_free(ir);
        return 1;
    };
    int r = run(text, true);
    if (r != 1) { std::fprintf(stderr, "the IR file itself: %d (%s)\n", r, divans_gpu_last_error()); return 3; }
    long accepted = 0;
    for (long it = 0; it < iterations; ++it) {
        std::string t = text;
        if (rnd() % 4 == 0 && t.size() > 4096) { const size_t from = rnd_below(t.size() - 4096); t = t.substr(from, 4096 + rnd_below(60000)); }   // a window of it: most damage then lands near live commands
        const int kinds = 1 + (int)(rnd() % 4);
        for (int k = 0; k < kinds && !t.empty(); ++k) {
            const size_t at = rnd_below(t.size());
            switch (rnd() % 7) {
                case 0: t[at] = (char)rnd(); break;
                case 1: t.erase(at, 1 + rnd_below(40)); break;
                case 2: t.insert(at, t.substr(rnd_below(t.size()), 1 + rnd_below(80))); break;
                case 3: t.insert(at, " 99999999999999999999 "); break;
                case 4: t.insert(at, " 4294967295 "); break;
                case 5: t.insert(at, (rnd() & 1) ? "\n" : " "); break;
                default: t.resize(at); break;
            }
        }
        r = run(t, false);
        if (r != 0 && r != 1) { std::fprintf(stderr, "iteration %ld: %d\n", it, r); return 6; }
        accepted += r;
    }
    std::printf("IR of %zu characters; %ld damaged: %ld still parse\n", text.size(), iterations, accepted);
    return 0;
}

static int fuzz_plan(const char* path, long iterations) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return 2;
    Bytes data; { uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + n); }
    std::fclose(f);
    static const divans_speed palette[6] = {{0x10, 0x2000}, {2, 1024}, {64, 16384}, {128, 16384}, {1, 16384}, {4, 1024}};
    size_t total_in = 0, total_out = 0;
    for (long it = 0; it < iterations; ++it) {
        const size_t n = (rnd() % 8 == 0) ? rnd_below(8) : std::min(data.size(), (rnd() % 3 == 0) ? rnd_below(data.size() + 1) : rnd_size() * (1 + rnd_below(4)));
        const size_t from = rnd_below(data.size() - n + 1);
        const uint8_t* in = data.data() + from;
        divans_host::StreamOptions opt;
        opt.window_size = 10 + (int)rnd_below(13);
        opt.dynamic_context_mixing = (uint8_t)rnd_below(3);
        opt.use_context_map = (rnd() & 1) != 0;
        opt.force_stride = (uint8_t)((rnd() & 1) ? rnd_below(9) : 9);
        opt.has_prior_depth = (rnd() & 3) == 0; opt.prior_depth = (uint8_t)rnd_below(3);
        opt.has_literal_adaptation = (rnd() & 3) == 0;
        for (auto& sp : opt.literal_adaptation) sp = palette[rnd_below(6)];
        opt.use_brotli = 0;
        const size_t call_buffer = rnd_size();
        std::vector<size_t> calls;
        if (rnd() & 1) { size_t left = n; while (left) { const size_t k = std::min(left, rnd_size() * (1 + rnd_below(3))); calls.push_back(k); left -= k; } }
        // the product's two phases
        divans_host::StreamPlan plan;
        if (divans_host::plan_stream(opt, n, calls.empty() ? nullptr : &calls, plan) != 0) { std::fprintf(stderr, "iteration %ld: plan_stream failed\n", it); return 3; }
        {   // the same plan through the shared PredictionMode prefix (what divans_batch_compress does for every stream of a batch)
            const std::shared_ptr<const divans_host::PlanPrefix> prefix = divans_host::make_plan_prefix(opt);
            divans_host::StreamPlan q;
            if (!prefix || divans_host::plan_stream(opt, n, calls.empty() ? nullptr : &calls, q, prefix.get()) != 0) { std::fprintf(stderr, "iteration %ld: plan with prefix failed\n", it); return 3; }
            bool same = q.cmd == plan.cmd && q.steps.size() == plan.steps.size() && q.lit_chunks == plan.lit_chunks && q.n == plan.n && q.window == plan.window
                        && std::memcmp(&q.cfg, &plan.cfg, sizeof(plan.cfg)) == 0;
            for (size_t k = 0; same && k < q.steps.size(); ++k) same = q.steps[k].kind == plan.steps[k].kind && q.steps[k].value == plan.steps[k].value;
            if (!same) { std::fprintf(stderr, "iteration %ld: the prefix changes the plan\n", it); return 3; }
        }
        orc_lit_config cfg; std::memcpy(&cfg, &plan.cfg, sizeof(cfg));
        orc_lit_state* st = orc_lit_state_new(&cfg);
        orc_ans_encoder enc; orc_ans_encoder_init(&enc);
        std::vector<uint32_t> chunk_bytes;
        for (size_t pos = 0; pos < n; ) {                            // 32 768 bytes = one 65 536-symbol chunk
            const size_t take = std::min<size_t>(n - pos, 32768);
            const size_t before = enc.out.len;
            orc_lit_encode_bytes(st, &enc, in + pos, take);
            pos += take;
            if (enc.out.len != before) chunk_bytes.push_back((uint32_t)(enc.out.len - before));
        }
        if (enc.n_pending) { const size_t before = enc.out.len; orc_ans_flush_chunk(&enc); chunk_bytes.push_back((uint32_t)(enc.out.len - before)); }
        if (chunk_bytes.size() != plan.lit_chunks) { std::fprintf(stderr, "iteration %ld: %zu chunks, the plan expects %u\n", it, chunk_bytes.size(), plan.lit_chunks); return 4; }
        Bytes mine;
        const int rc = divans_host::assemble_container(plan, enc.out.data, enc.out.len, chunk_bytes.data(), call_buffer, mine);
        orc_ans_encoder_free(&enc); orc_lit_state_free(st);
        if (rc != 0) { std::fprintf(stderr, "iteration %ld: assemble_container failed (%d)\n", it, rc); return 5; }
        // the oracle's container for the same caller
        orc_stream_options o; orc_stream_options_default(&o);
        o.window_size = opt.window_size; o.dynamic_context_mixing = opt.dynamic_context_mixing;
        o.prior_depth = opt.has_prior_depth ? opt.prior_depth : 0; o.use_context_map = opt.use_context_map ? 1 : 0; o.force_stride = opt.force_stride;
        o.has_literal_adaptation = opt.has_literal_adaptation ? 1 : 0;
        for (int i = 0; i < 4; ++i) { o.literal_adaptation[i].inc = opt.literal_adaptation[i].inc; o.literal_adaptation[i].lim = opt.literal_adaptation[i].lim; }
        o.call_buffer_size = call_buffer; o.call_inputs = calls.empty() ? nullptr : calls.data(); o.n_call_inputs = calls.size();
        Bytes ref(2 * n + 70000);
        const size_t rn = orc_stream_compress_raw(&o, in, n, ref.data(), ref.size());
        if (rn == (size_t)-1) { std::fprintf(stderr, "iteration %ld: the oracle failed\n", it); return 6; }
        if (rn != mine.size() || std::memcmp(ref.data(), mine.data(), rn) != 0) {
            size_t d = 0; while (d < rn && d < mine.size() && ref[d] == mine[d]) ++d;
            std::fprintf(stderr, "iteration %ld: n %zu window %d mixing %u buffer %zu calls %zu: %zu vs %zu bytes, first difference at %zu\n",
                         it, n, opt.window_size, opt.dynamic_context_mixing, call_buffer, calls.size(), mine.size(), rn, d);
            return 7;
        }
        // and the whole-container parser takes it apart again
        divans_host::ParsedStream ps; size_t used = 0;
        if (divans_host::parse_container_host(mine.data(), mine.size(), false, n + 16, ps, &used) != divans_host::PARSE_OK || ps.total != n || used != mine.size()
            || !ps.cfg || std::memcmp(ps.cfg.get(), &plan.cfg, sizeof(plan.cfg)) != 0) { std::fprintf(stderr, "iteration %ld: the parser disagrees with the plan\n", it); return 8; }
        total_in += n; total_out += mine.size();
    }
    std::printf("%ld planned containers: %zu bytes in, %zu out, all equal to the oracle's\n", iterations, total_in, total_out);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 5 && std::strcmp(argv[1], "plan") == 0) {
        rng_state = std::strtoull(argv[3], nullptr, 0) * 0x9E3779B97F4A7C15ull + 1;
        return fuzz_plan(argv[2], std::strtol(argv[4], nullptr, 0));
    }
    if (argc >= 5 && std::strcmp(argv[1], "ir") == 0) {
        rng_state = std::strtoull(argv[3], nullptr, 0) * 0x9E3779B97F4A7C15ull + 1;
        return fuzz_ir(argv[2], std::strtol(argv[4], nullptr, 0));
    }
    if (argc < 4) { std::fprintf(stderr, "usage: hostsim_fuzz <file> <seed> <iterations> [selector=value ...]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    Bytes data; { uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + n); }
    std::fclose(f);
    rng_state = std::strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull + 1;
    const long iterations = std::strtol(argv[3], nullptr, 0);
    std::vector<std::pair<unsigned, unsigned>> opts;
    for (int i = 4; i < argc; ++i) { unsigned s, v; if (std::sscanf(argv[i], "%u=%u", &s, &v) == 2) opts.push_back({s, v}); }

    // 1. the same bytes whatever the caller's buffer sizes are NOT expected (the Mux's slices depend on them); the same CONTENT is
    Bytes plain, ragged, back;
    if (!compress(data, opts, true, plain) || !compress(data, opts, false, ragged)) return 3;
    for (const Bytes* c : {&plain, &ragged})
        for (int fixed = 0; fixed < 2; ++fixed) {
            back.clear();
            const Outcome o = decompress(*c, false, fixed != 0, data.size(), back);
            if (o != OK || back != data) { std::fprintf(stderr, "round trip failed: outcome %d, %zu of %zu bytes\n", (int)o, back.size(), data.size()); return 4; }
        }
    // trailing bytes behind a complete container are not consumed and do not turn success into failure
    { Bytes t = plain; t.insert(t.end(), 100, 0x5a); back.clear(); if (decompress(t, false, true, data.size(), back) != OK || back != data) { std::fprintf(stderr, "trailing bytes changed the result\n"); return 5; } }

    long n_ok = 0, n_failed = 0, n_wants = 0;
    divans_host::ParseMemo memo;
    for (long it = 0; it < iterations; ++it) {
        Bytes c = (rnd() & 1) ? plain : ragged;
        damage(c, (rnd() & 1) ? plain : ragged);
        const bool skip_crc = (rnd() % 3) == 0;
        back.clear();
        const Outcome o = decompress(c, skip_crc, (rnd() % 4) == 0, data.size() + (1u << 20), back);
        if (o == STUCK || o == OVERRUN) { std::fprintf(stderr, "iteration %ld: decoder %s\n", it, o == STUCK ? "makes no progress" : "overran a bound"); return 6; }
        if (o == OK && !skip_crc && back != data) { std::fprintf(stderr, "iteration %ld: success with wrong bytes under the CRC\n", it); return 7; }
        n_ok += o == OK; n_failed += o == FAILED; n_wants += o == WANTS_INPUT;
        // the whole-container parser of the batch interface on the same bytes
        divans_host::ParsedStream ps; size_t used = 0;
        const divans_host::ParseStatus st = divans_host::parse_container_host(c.data(), c.size(), skip_crc, data.size() + (1u << 20), ps, &used);
        {   // and with the memo of CMD streams seen before (divans_batch_decompress uses one per call): the same answer
            divans_host::ParsedStream pm; size_t used_m = 0;
            const divans_host::ParseStatus sm = divans_host::parse_container_host(c.data(), c.size(), skip_crc, data.size() + (1u << 20), pm, &used_m, &memo);
            if (sm != st || (st == divans_host::PARSE_OK && (used_m != used || pm.total != ps.total || pm.lit != ps.lit || !pm.cfg || !ps.cfg || std::memcmp(pm.cfg.get(), ps.cfg.get(), sizeof(divans_lit_config)) != 0))) {
                std::fprintf(stderr, "iteration %ld: the memo changes the parser's answer (%d vs %d)\n", it, (int)sm, (int)st); return 10;
            }
        }
        {   // and without copying the LIT slices out (the batch interface stages them from where they lie): the same answer, the same bytes
            divans_host::ParsedStream pp; size_t used_p = 0;
            const divans_host::ParseStatus sp = divans_host::parse_container_host(c.data(), c.size(), skip_crc, data.size() + (1u << 20), pp, &used_p, nullptr, true);
            bool same = sp == st;
            if (same && st == divans_host::PARSE_OK) {
                std::vector<uint8_t> gathered(pp.lit_size + 1);
                for (const auto& span : pp.lit_spans) if ((size_t)span.first + span.second > c.size()) same = false;
                if (same) {
                    if (!pp.lit_spans.empty()) pp.copy_lit(c.data(), gathered.data()); else if (pp.lit_size) std::memcpy(gathered.data(), pp.lit.data(), pp.lit_size);
                    gathered.resize(pp.lit_size);
                    same = used_p == used && pp.total == ps.total && pp.lit_size == ps.lit.size() && gathered == ps.lit && pp.cfg && ps.cfg && std::memcmp(pp.cfg.get(), ps.cfg.get(), sizeof(divans_lit_config)) == 0;
                }
            }
            if (!same) { std::fprintf(stderr, "iteration %ld: the span parser disagrees with the copying parser (%d vs %d)\n", it, (int)sp, (int)st); return 11; }
        }
        if (st == divans_host::PARSE_OK && (used > c.size() || ps.total > data.size() + (1u << 20))) { std::fprintf(stderr, "iteration %ld: parser out of bounds\n", it); return 8; }
        if (st == divans_host::PARSE_OK && o != OK) {
            // framing, CMD stream and CRC held, so only the literal decoder can have refused it (its final-state check) -- impossible while the CRC is checked
            if (!skip_crc) { std::fprintf(stderr, "iteration %ld: the parser accepts what the decoder (%d) refused under the CRC\n", it, (int)o); return 9; }
        }
        if (o == OK && st != divans_host::PARSE_OK) { std::fprintf(stderr, "iteration %ld: the decoder accepts what the parser (%d) refused\n", it, (int)st); return 9; }
    }
    std::printf("containers %zu / %zu bytes for %zu; %ld damaged: %ld refused, %ld starved, %ld decoded\n", plain.size(), ragged.size(), data.size(), iterations, n_failed, n_wants, n_ok);
    return 0;
}
