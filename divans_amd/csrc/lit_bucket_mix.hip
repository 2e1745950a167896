// lit_bucket_mix.hip -- bucketed model pass of the ENCODER for the two-model configuration (BASELINE configs[2],
// reference TestContextMixing, bin/benchmark.rs:156-167): context map on, every mixing value 4, dynamic mixing.
//
// Each nibble is coded with the weighted average of two rows (codec/literal.rs:209-241):
//   stride model   high nibble: row [ctx][prev]      low nibble: row [prev][hi]       blended with literal_adaptation[0]
//   context model  high nibble: row First[ctx]       low nibble: row Second[hi][ctx]  blended with literal_adaptation[3] / [2]
// and the weights of the average follow from how well each model did on the symbols before (codec/weights.rs:23-38).
// A row's CDF depends only on the earlier positions that used the same row -- never on the weights -- so the rows
// can be walked bucket by bucket exactly as in lit_bucket.hip, once per model:
//   stride model:  buckets keyed by prev  (up to 8 high rows, one per class of prev_prev that reaches a different ctx, + 16 low rows)
//   context model: buckets keyed by ctx   (1 high row + 16 low rows)
// Instead of (start, freq) a chain lane leaves the three raw counts mixing needs per nibble -- cdf[sym], cdf[sym-1], cdf[15] --
// as 12 bytes per position: {high: cdf[sym] | cdf[sym-1] << 16, low: the same} in one plane, the two cdf[15] in another.
// After both models' records are back in position order, mix_weights_kernel runs the only serial part that is left --
// the per-stream Weights recursion, one lane per (stream, nibble half): average (probability/frequentist_cdf.rs:58-72)
// of the three entries, the six divisions of sym_to_start_and_freq (probability/interface.rs:97-108), Weights::update.
#include "lit_bucket_dev.h"

namespace divans_hip {

// Chain waves per CU.  The stride model's 24 rows per lane allow three; the context model's 17 would allow four, but three
// were faster in round 2 (30.9 vs 35.7 ms per 32 768 streams, two: 32.2): with 16 bytes of records leaving per lane and step the
// waves of a CU queue up behind its vector-memory path (TA busy 80 %), and a fourth wave only lengthens the queue.  With the
// 16-byte payload loads of round 3 three and four measure the same.
constexpr uint32_t MX_CHAIN_WAVES = 3;

template <int MODEL> struct MxGeom {
    static constexpr uint32_t NH = MODEL == 0 ? 8u : 1u;            // high-nibble rows of a bucket
    static constexpr uint32_t NR = NH + 16u;
    static constexpr uint32_t DESC_DW = NR * 8u;
    // rows + 8 descriptors, padded so that the 64 lanes' b128 accesses at equal offsets cover all 32 banks
    static constexpr uint32_t LANE_DW = MODEL == 0 ? 204u : 148u;
    static constexpr uint32_t LDS_BYTES = (64u * LANE_DW + 256u) * 4u;
};

// ---------------------------------------------------------------------------------------------
// 1. per (stream, piece): sorted[slot] = byte | high-row slot << 8, inv[pos] = slot, desc[stream][key][piece]
//    MODEL 0 (stride): key = prev.  MODEL 1 (context map): key = ctx (literal.rs:87-117 through the fused table).
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(BK_SORT_THREADS) void mix_sort_kernel(const MixBucketBatch b) {
    __shared__ __attribute__((aligned(16))) uint16_t staging[BK_PIECE];
    __shared__ __attribute__((aligned(16))) uint8_t piece_in[16 + BK_PIECE];   // piece_in[14], [15] = the two bytes before the piece
    __shared__ uint16_t kp[BK_PIECE];                                          // key | high-row slot << 8 of every position
    __shared__ uint32_t hist[4][256];
    __shared__ uint32_t scan[256];
    __shared__ uint8_t lut1c[256], ctxf[2048], slot_of[2048];
    const uint32_t s = blockIdx.x / b.pieces, piece = blockIdx.x % b.pieces;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
    const uint8_t* in = b.in + (b.in_offsets ? b.in_offsets[s] : (uint64_t)s * b.stream_len);
    uint32_t* desc = b.desc + ((size_t)s * 256u + tid) * 8u + piece;
    const uint32_t base = piece * BK_PIECE;
    if (base >= len) { *desc = 0u; return; }
    const uint32_t n = len - base < BK_PIECE ? len - base : BK_PIECE;
    const size_t pl = b.slot;
    for (uint32_t i = tid; i < 1024u; i += BK_SORT_THREADS) (&hist[0][0])[i] = 0u;
    lut1c[tid] = b.blob[LIT_BLOB_LUT1CLASS + tid];
    for (uint32_t i = tid; i < 2048u; i += BK_SORT_THREADS) ctxf[i] = b.blob[LIT_BLOB_CTXF + i];
    {
        const uint8_t* src = in + base;
        if ((((uintptr_t)src) & 15u) == 0u) {
            for (uint32_t i = tid * 16u; i < n; i += BK_SORT_THREADS * 16u) {
                if (i + 16u <= n) *(u32x4*)(piece_in + 16u + i) = *(const u32x4*)(src + i);
                else for (uint32_t k = i; k < n; ++k) piece_in[16u + k] = src[k];
            }
        } else {
            for (uint32_t i = tid; i < n; i += BK_SORT_THREADS) piece_in[16u + i] = src[i];
        }
        if (tid == 0u) { piece_in[15] = base ? in[base - 1u] : 0u; piece_in[14] = base ? in[base - 2u] : 0u; }   // last_8_literals starts at zero
    }
    __syncthreads();
    // classes of prev_prev that reach the same context share a high row: slot = the first such class
    if (MODEL == 0) {
        for (uint32_t i = tid; i < 2048u; i += BK_SORT_THREADS) {
            const uint32_t row = i & ~7u; const uint8_t c = ctxf[i];
            uint32_t k = 0; while (ctxf[row + k] != c) ++k;
            slot_of[i] = (uint8_t)k;
        }
        __syncthreads();
    }
    // wave w owns positions [2048 w, 2048 w + 2048) of the piece and visits them in order, 64 at a time
    for (uint32_t bt = 0; bt < 32u; ++bt) {
        const uint32_t p = w * 2048u + bt * 64u + lane;
        if (p < n) {
            const uint32_t prev = piece_in[15u + p], e = (prev << 3) + lut1c[piece_in[14u + p]];
            const uint32_t key = MODEL == 0 ? prev : ctxf[e];
            kp[p] = (uint16_t)(key | (MODEL == 0 ? (uint32_t)slot_of[e] << 8 : 0u));
            atomicAdd(&hist[w][key], 1u);
        }
    }
    __syncthreads();
    const uint32_t c0 = hist[0][tid], c1 = hist[1][tid], c2 = hist[2][tid], c3 = hist[3][tid];
    const uint32_t tot = c0 + c1 + c2 + c3;
    scan[tid] = tot;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t v = tid >= d ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const uint32_t start = scan[tid] - tot;
    hist[0][tid] = start; hist[1][tid] = start + c0; hist[2][tid] = start + c0 + c1; hist[3][tid] = start + c0 + c1 + c2;
    *desc = start | (tot << 16);
    __syncthreads();
    uint16_t* inv = b.inv + (size_t)s * pl + base;
    for (uint32_t bt = 0; bt < 32u; ++bt) {
        const uint32_t p = w * 2048u + bt * 64u + lane;
        const bool valid = p < n;
        const uint32_t k = valid ? kp[p] : 0u;
        const uint32_t key = k & 0xffu;
        const uint32_t pay = valid ? (uint32_t)piece_in[16u + p] | (k & 0xff00u) : 0u;
        unsigned long long same = __ballot(valid);
        for (uint32_t bit = 0; bit < 8u; ++bit) {
            const bool set = (key >> bit) & 1u;
            const unsigned long long bb = __ballot(set);
            same &= set ? bb : ~bb;
        }
        const uint32_t rank = lanes_below(same), cnt = (uint32_t)__popcll(same);
        if (valid) {
            const uint32_t off = hist[w][key];
            staging[off + rank] = (uint16_t)pay;
            inv[p] = (uint16_t)(off + rank);
            if (rank == cnt - 1u) hist[w][key] = off + cnt;   // the wave's LDS accesses stay in program order
        }
    }
    __syncthreads();
    uint16_t* sorted = b.sorted + (size_t)s * pl + base;
    for (uint32_t i = tid * 8u; i < n; i += BK_SORT_THREADS * 8u) {
        if (i + 8u <= n) *(u32x4*)(sorted + i) = *(const u32x4*)(staging + i);
        else for (uint32_t k = i; k < n; ++k) sorted[k] = staging[k];
    }
}

// ---------------------------------------------------------------------------------------------
// 3. chains: one lane per bucket, rows in LDS, raw counts out.  Same skeleton as bucket_chain_kernel (task window,
//    one 16-byte payload load per iteration, stores hidden from the compiler and delayed by an iteration); see there for why.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(64) void mix_chain_kernel(const MixBucketBatch b) {
    using G = MxGeom<MODEL>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    const uint32_t lane = threadIdx.x;
    uint32_t* my = lds32 + lane * G::LANE_DW;
    uint32_t* mydesc = my + G::DESC_DW;
    uint32_t* tabh = lds32 + 64u * G::LANE_DW;
    uint32_t* tabl = tabh + 128u;
    const uint32_t inch = (uint32_t)(MODEL == 0 ? b.inc0 : b.inc3), incl = (uint32_t)(MODEL == 0 ? b.inc0 : b.inc2);   // literal.rs:320,354 / :242
    const int limh = MODEL == 0 ? b.lim0 : b.lim3, liml = MODEL == 0 ? b.lim0 : b.lim2;
    for (uint32_t i = lane; i < 128u; i += 64u) {
        const uint32_t sym = i >> 3, k = i & 7u;
        tabh[i] = (2u * k >= sym ? inch : 0u) | (2u * k + 1u >= sym ? inch << 16 : 0u);
        tabl[i] = (2u * k >= sym ? incl : 0u) | (2u * k + 1u >= sym ? incl << 16 : 0u);
    }
    __syncthreads();
    const size_t pl = b.slot;
    const uint32_t cap = b.n_streams * 256u;
    BkTaskLists lists; lists.load(b.counters);
    const uint32_t total = lists.total();
    const u32x4 def0 = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
    const u32x4 def1 = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};

    // Same loop as bucket_chain_kernel (see there): a bucket's runs compacted to the non-empty ones, ONE aligned 16-byte load of
    // eight sorted payloads per iteration requested an iteration ahead, the records of a group stored at the top of the next one.
    bool has_task = false, exhausted = false;
    uint32_t run_i = 0, nruns = 0, left = 0, idx = 0;
    u32x2* const rec_x = b.xs[MODEL]; uint32_t* const rec_m = b.maxes[MODEL];
    u32x2* cur_x = rec_x; uint32_t* cur_m = rec_m; const uint16_t* cur_sorted = b.sorted;
    uint32_t nt_stage = 0, nt_tid = 0;
    u32x4 nd0 = {0u, 0u, 0u, 0u}, nd1 = {0u, 0u, 0u, 0u};
    uint32_t win_cur = 0, win_end = 0, nxt_val = 0, nxt_w = 0;
    const uint32_t long_end = lists.ends[3];     // tasks of at least 2048 positions come first
    bool nxt_pending = false, drained = false;
    u32x4 e_next = {0u, 0u, 0u, 0u}; uint32_t m_next = 0;   // meta: base | first << 16 | cnt << 20 | BK_VALID
    u32x2 x0 = {0u, 0u}, x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;     // {high, low} entries of the group's positions
    uint32_t t0 = 0u, t1 = 0u, t2 = 0u, t3 = 0u, t4 = 0u, t5 = 0u, t6 = 0u, t7 = 0u;       // their row totals, high | low << 16
    uint32_t m_prev = 0; u32x2* x_prev = rec_x; uint32_t* t_prev = rec_m;

#define MX_OPAQUE(X) asm volatile("" : "+v"(X))
#define MX_POS(K, WORD, RX, RT)                                                                         \
    if (((K - first) & 15u) < cnt) {                                                                    \
        const uint32_t pay = (WORD >> (16u * (K & 1u))) & 0xffffu;                                      \
        const uint32_t hi = (pay >> 4) & 15u, lo = pay & 15u;                                           \
        uint32_t* rowh = MODEL == 0 ? my + ((pay >> 5) & 0x38u) : my;   /* 8 dwords x slot (bits 8..10) */ \
        uint32_t* rowl = my + 8u * (G::NH + hi);                                                        \
        BkRow H = bk_read(rowh, tabh, hi), L = bk_read(rowl, tabl, lo);                                 \
        const u32x2 vx = {(uint32_t)H.chi | (hi ? (uint32_t)H.cprev << 16 : 0u),                        \
                          (uint32_t)L.chi | (lo ? (uint32_t)L.cprev << 16 : 0u)};                       \
        const uint32_t vt = (H.w1.w >> 16) | (L.w1.w & 0xffff0000u);                                    \
        H.w0 += H.a0; H.w1 += H.a1; L.w0 += L.a0; L.w1 += L.a1;   /* frequentist_cdf.rs:75-78 */        \
        if ((int)(H.w1.w >> 16) >= limh) bk_renorm(H);                                                  \
        if ((int)(L.w1.w >> 16) >= liml) bk_renorm(L);                                                  \
        *(u32x4*)rowh = H.w0; *(u32x4*)(rowh + 4) = H.w1; *(u32x4*)rowl = L.w0; *(u32x4*)(rowl + 4) = L.w1; \
        RX = vx; RT = vt;                                                                               \
    }
#define MX_STORE8(DST, R0, R1, R2, R3, R4, R5, R6, R7)                                                  \
    {                                                                                                   \
        u32x2* dst = DST + (m_prev & 0xffffu);                                                          \
        if (pc == 8u) {                                                                                 \
            const u32x4 q0 = {R0.x, R0.y, R1.x, R1.y}, q1 = {R2.x, R2.y, R3.x, R3.y};                   \
            const u32x4 q2 = {R4.x, R4.y, R5.x, R5.y}, q3 = {R6.x, R6.y, R7.x, R7.y};                   \
            bk_store_quad((u32x4*)dst, q0); bk_store_quad((u32x4*)(dst + 2), q1);                       \
            bk_store_quad((u32x4*)(dst + 4), q2); bk_store_quad((u32x4*)(dst + 6), q3);                 \
        } else {                                                                                        \
            if (((0u - pf) & 15u) < pc) bk_store_pair(dst + 0, R0);                                     \
            if (((1u - pf) & 15u) < pc) bk_store_pair(dst + 1, R1);                                     \
            if (((2u - pf) & 15u) < pc) bk_store_pair(dst + 2, R2);                                     \
            if (((3u - pf) & 15u) < pc) bk_store_pair(dst + 3, R3);                                     \
            if (((4u - pf) & 15u) < pc) bk_store_pair(dst + 4, R4);                                     \
            if (((5u - pf) & 15u) < pc) bk_store_pair(dst + 5, R5);                                     \
            if (((6u - pf) & 15u) < pc) bk_store_pair(dst + 6, R6);                                     \
            if (((7u - pf) & 15u) < pc) bk_store_pair(dst + 7, R7);                                     \
        }                                                                                               \
    }
#define MX_STORE8W(DST, R0, R1, R2, R3, R4, R5, R6, R7)                                                 \
    {                                                                                                   \
        uint32_t* dst = DST + (m_prev & 0xffffu);                                                       \
        if (pc == 8u) {                                                                                 \
            const u32x4 q0 = {R0, R1, R2, R3}, q1 = {R4, R5, R6, R7};                                   \
            bk_store_quad((u32x4*)dst, q0); bk_store_quad((u32x4*)(dst + 4), q1);                       \
        } else {                                                                                        \
            if (((0u - pf) & 15u) < pc) bk_store_word(dst + 0, R0);                                     \
            if (((1u - pf) & 15u) < pc) bk_store_word(dst + 1, R1);                                     \
            if (((2u - pf) & 15u) < pc) bk_store_word(dst + 2, R2);                                     \
            if (((3u - pf) & 15u) < pc) bk_store_word(dst + 3, R3);                                     \
            if (((4u - pf) & 15u) < pc) bk_store_word(dst + 4, R4);                                     \
            if (((5u - pf) & 15u) < pc) bk_store_word(dst + 5, R5);                                     \
            if (((6u - pf) & 15u) < pc) bk_store_word(dst + 6, R6);                                     \
            if (((7u - pf) & 15u) < pc) bk_store_word(dst + 7, R7);                                     \
        }                                                                                               \
    }

    for (;;) {
        u32x4 e = e_next; const uint32_t m = m_next;
        if (m_prev & BK_VALID) {                        // 1. the previous group's records leave
            const uint32_t pf = (m_prev >> 16) & 15u, pc = (m_prev >> 20) & 15u;
            MX_STORE8(x_prev, x0, x1, x2, x3, x4, x5, x6, x7)
            MX_STORE8W(t_prev, t0, t1, t2, t3, t4, t5, t6, t7)
        }
        {                                               // 2. the next group of the run is requested, the next run taken
            const bool adv = has_task && left == 0u, more = run_i < nruns;
            const uint32_t d = mydesc[run_i & 7u];
            if (adv && more) { left = d >> 16; idx = d & 0xffffu; ++run_i; }
            if (adv && !more) has_task = false;
        }
        {
            const bool fetch_ = has_task && left != 0u;
            const uint32_t base = idx & ~7u, first_ = idx & 7u;
            const uint32_t cnt_ = left < 8u - first_ ? left : 8u - first_;
            const uint16_t* lp = fetch_ ? cur_sorted + base : b.sorted;
            e_next = *(const u32x4*)lp;
            m_next = fetch_ ? (base | (first_ << 16) | (cnt_ << 20) | BK_VALID) : 0u;
            idx += fetch_ ? cnt_ : 0u; left -= fetch_ ? cnt_ : 0u;
        }
        MX_OPAQUE(e);                                   // 3. this iteration's group
        m_prev = m; x_prev = cur_x; t_prev = cur_m;
        if (m & BK_VALID) {
            const uint32_t first = (m >> 16) & 15u, cnt = (m >> 20) & 15u;
            MX_POS(0u, e.x, x0, t0) MX_POS(1u, e.x, x1, t1) MX_POS(2u, e.y, x2, t2) MX_POS(3u, e.y, x3, t3)
            MX_POS(4u, e.z, x4, t4) MX_POS(5u, e.z, x5, t5) MX_POS(6u, e.w, x6, t6) MX_POS(7u, e.w, x7, t7)
        }
        if (!has_task && !(m_next & BK_VALID) && nt_stage == 3u) {   // 4. a finished lane takes its prefetched task
            MX_OPAQUE(nd0); MX_OPAQUE(nd1);
            uint32_t n = 0;
            const uint32_t dsc[8] = {nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, nd1.z, nd1.w};
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) if (dsc[j] >> 16) { mydesc[n] = dsc[j] + j * BK_PIECE; ++n; }
            nruns = n; run_i = 0u;
            for (uint32_t r = 0; r < G::NR; ++r) { *(u32x4*)(my + 8u * r) = def0; *(u32x4*)(my + 8u * r + 4u) = def1; }
            const size_t slot = (size_t)(nt_tid >> 8) * pl;
            cur_x = rec_x + slot; cur_m = rec_m + slot; cur_sorted = b.sorted + slot;
            left = 0u; has_task = true; nt_stage = 0u;
        }
        const bool want = nt_stage == 0u && !exhausted;
        if (nt_stage == 2u) nt_stage = 3u;
        else if (nt_stage == 1u) {
            MX_OPAQUE(nt_tid);
            const u32x4* dp = (const u32x4*)(b.desc + (size_t)nt_tid * 8u);
            nd0 = dp[0]; nd1 = dp[1];
            nt_stage = 2u;
        }
        const unsigned long long wm = __ballot(want);
        if (wm) {
            if (win_cur == win_end && nxt_pending) {
                MX_OPAQUE(nxt_val);
                const uint32_t basev = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt_val);
                nxt_pending = false;
                if (basev >= total) { drained = true; win_cur = win_end = total; }
                else { win_cur = basev; win_end = basev + nxt_w < total ? basev + nxt_w : total; }
            }
            const uint32_t avail = win_end - win_cur, asked = (uint32_t)__popcll(wm);
            const uint32_t rank = lanes_below(wm);
            if (want) {
                if (rank < avail) {
                    const uint32_t t = win_cur + rank;
                    nt_tid = *lists.at(b.tasks, cap, t);
                    nt_stage = 1u;
                } else if (drained) exhausted = true;
            }
            win_cur += asked < avail ? asked : avail;
        }
        // long buckets are handed out 64 at a time: a wave that reserved 256 of them would run four per lane back to back
        const uint32_t want_w = win_end < long_end ? 64u : BK_WINDOW;
        if (!nxt_pending && !drained && win_end - win_cur < want_w / 2u) {
            nxt_w = want_w;
            if (lane == 0u) nxt_val = atomicAdd(&b.counters[BK_CLAIM], want_w);
            nxt_pending = true;
        }
        const bool done = !has_task && !(m_next & BK_VALID) && !(m_prev & BK_VALID) && nt_stage == 0u && exhausted;
        if (__ballot(!done) == 0ull) break;
    }
#undef MX_POS
#undef MX_STORE8
#undef MX_STORE8W
#undef MX_OPAQUE
}

// ---------------------------------------------------------------------------------------------
// 5. the Weights recursion: one lane per (stream, nibble half).  model_weights[1] belongs to the high nibbles and
//    model_weights[0] to the low nibbles (literal.rs:230), so the two halves of a stream are independent.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix_nibble(Weights& w, uint32_t st_x, uint32_t st_max, uint32_t cm_x, uint32_t cm_max) {
    const int mix_rate = w.norm;
    const int cm_s = (int)(cm_x & 0xffffu), cm_p = (int)(cm_x >> 16), st_s = (int)(st_x & 0xffffu), st_p = (int)(st_x >> 16);
    const int cmax = (int)cm_max, smax = (int)st_max;
    const int p_s = average_rows(cm_s, st_s, cmax, smax, mix_rate);       // frequentist_cdf.rs:58-72, entry sym
    const int p_p = average_rows(cm_p, st_p, cmax, smax, mix_rate);       //   entry sym-1 (0 | 0 -> 0 when sym == 0)
    // entry 15: both rows contribute their own total, so the two products are equal, rs = ro = (cmax * smax) >> sh, and
    // (rs * mix + rs * (2^15 - mix) + 1) >> 15 = rs whatever the weight -- the third average costs a multiply and a shift
    int pmax;
    {
        const uint32_t prod = __umul24((uint32_t)cmax, (uint32_t)smax);
        int lz = __clz((int)prod);
        lz = lz > 17 ? 17 : lz;
        pmax = (int)(prod >> (17 - lz));
    }
    const float rp = biased_rcp15(pmax), rc = biased_rcp15(cmax), rs = biased_rcp15(smax);
    const uint32_t qp = scaled_div(p_p, pmax, rp);
    const uint32_t freq = scaled_div(p_s, pmax, rp) - qp - 1u;            // probability/interface.rs:97-108
    const uint32_t fcm = scaled_div(cm_s, cmax, rc) - scaled_div(cm_p, cmax, rc) - 1u;
    const uint32_t fst = scaled_div(st_s, smax, rs) - scaled_div(st_p, smax, rs) - 1u;
    weights_update(w, (int)(short)fcm, (int)(short)fst, (int)(short)freq);   // literal.rs:236-239
    return (qp + 1u) | (freq << 16);
}

// One wave = 32 streams x 2 halves.  The records of a stream are contiguous in memory, so the wave fetches them
// together -- every 16-byte load instruction covers 128-byte (xs) / 64-byte (maxes) runs of 8 / 16 streams -- into LDS, 16 positions
// at a time, double-buffered; each lane then reads its own stream's records from LDS, and the (start, freq) pairs go back out the
// same way (a lane-per-stream walk straight over global memory costs one 16-byte request per lane and load).
constexpr uint32_t MW_CHUNK = 16;                       // positions per LDS buffer (8 and 4 measured 13 % / 25 % worse on the model pass -- before the
                                                        // one-wave-per-SIMD attribute below, so possibly for its reason; not re-measured)
constexpr uint32_t MW_XP = MW_CHUNK / 2u, MW_TP = MW_CHUNK / 4u;     // 16-byte pieces of a stream's chunk: two xs records, four maxes
constexpr uint32_t MW_XS = 64u / MW_XP, MW_TS = 64u / MW_TP;         // streams one load instruction of the wave covers
constexpr uint32_t MW_XG = 32u / MW_XS, MW_TG = 32u / MW_TS;         // such loads per plane
constexpr uint32_t MW_PER_MODEL = MW_XG + MW_TG;
constexpr uint32_t MW_NLOAD = 2u * MW_PER_MODEL;
constexpr uint32_t MW_MODEL_BYTES = MW_CHUNK * 12u;                  // a stream's chunk of one model in LDS: xs (8 B / position), then maxes (4 B)
constexpr uint32_t MW_IN_STRIDE = 2u * MW_MODEL_BYTES + 16u;         // bytes per stream, padded against bank conflicts
constexpr uint32_t MW_OUT_STRIDE = 2u * MW_CHUNK * 4u + 16u;
constexpr uint32_t MW_IN_BYTES = 32u * MW_IN_STRIDE;

// One wave per SIMD, enforced (amdgpu_waves_per_eu): a wave walks its 32 streams' recursions at the SIMD's issue rate and a second
// wave on the same SIMD only halves both, while a 32 768-stream sequence is exactly one wave for each of the chip's 1024 SIMDs.  Left
// to the dispatcher, a CU's four workgroups do not always land on four different SIMDs: round 2's version happened to be safe because
// its 257 registers allowed one wave per SIMD anyway; with 169 the kernel took 39.6 instead of 24.5 ms until this attribute came.
// A sequence of more than 32 768 streams (round 6: 65 536 where the device has room for its work arrays) is two waves for every SIMD --
// the WPE = 2 instance, pinned to exactly that: the second wave issues into the first one's dependency stalls.
template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void mix_weights_kernel(const MixBucketBatch b) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_in[2u * MW_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t lds_out[32u * MW_OUT_STRIDE];
    const uint32_t lane = threadIdx.x;
    const uint32_t s0 = blockIdx.x * 32u;
    // consumer role: stream s0 + lane / 2, nibble half lane & 1
    const uint32_t half = lane & 1u;
    // mover role, xs: stream s0 + 8 g + lane / 8 of load g, records 2 (lane & 7), + 1 of the chunk; maxes: stream s0 + 16 g + lane / 4,
    // totals 4 (lane & 3) .. + 3; pairs out: as xs
    const uint32_t xq = lane % MW_XP, xj = lane / MW_XP, tq = lane % MW_TP, tj = lane / MW_TP;
    const uint32_t chunks = (b.stream_len + MW_CHUNK - 1u) / MW_CHUNK;     // stream_len = the longest stream of the batch
    Weights w; w.w0 = 1; w.w1 = 1; w.norm = 1 << 14;               // weights.rs:15-21
    u32x4 r[MW_NLOAD];
#define MW_FETCH(C)                                                                                     \
    _Pragma("unroll") for (uint32_t i = 0; i < MW_NLOAD; ++i) {                                         \
        const uint32_t model = i / MW_PER_MODEL, g = i % MW_PER_MODEL;                                  \
        if (g < MW_XG) {                                                                                \
            const uint32_t j = g * MW_XS + xj, p = (C) * MW_CHUNK + 2u * xq;                            \
            const bool ok = s0 + j < b.n_streams && p < b.max_stream_len;                               \
            const u32x2* src = b.xs[model] + (ok ? (size_t)(s0 + j) * b.pos_stride + p : 0u);           \
            r[i] = __builtin_nontemporal_load((const u32x4*)src);                                       \
        } else {                                                                                        \
            const uint32_t j = (g - MW_XG) * MW_TS + tj, p = (C) * MW_CHUNK + 4u * tq;                  \
            const bool ok = s0 + j < b.n_streams && p < b.max_stream_len;                               \
            const uint32_t* src = b.maxes[model] + (ok ? (size_t)(s0 + j) * b.pos_stride + p : 0u);     \
            r[i] = __builtin_nontemporal_load((const u32x4*)src);                                       \
        }                                                                                               \
    }
#define MW_STAGE(BUF)                                                                                   \
    _Pragma("unroll") for (uint32_t i = 0; i < MW_NLOAD; ++i) {                                         \
        const uint32_t model = i / MW_PER_MODEL, g = i % MW_PER_MODEL;                                  \
        uint8_t* dst = lds_in + (BUF) * MW_IN_BYTES + model * MW_MODEL_BYTES;                           \
        if (g < MW_XG) *(u32x4*)(dst + (g * MW_XS + xj) * MW_IN_STRIDE + xq * 16u) = r[i];              \
        else *(u32x4*)(dst + ((g - MW_XG) * MW_TS + tj) * MW_IN_STRIDE + MW_CHUNK * 8u + tq * 16u) = r[i]; \
    }
    if (chunks == 0u) return;
    MW_FETCH(0u)
    MW_STAGE(0u)
    __syncthreads();
    const uint32_t hs = 16u * half;
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint32_t buf = c & 1u;
        if (c + 1u < chunks) { MW_FETCH(c + 1u) }
        const uint8_t* mine = lds_in + buf * MW_IN_BYTES + (lane >> 1) * MW_IN_STRIDE;
        uint32_t* outp = (uint32_t*)(lds_out + (lane >> 1) * MW_OUT_STRIDE) + half;
        const uint32_t p0 = c * MW_CHUNK;
#pragma unroll
        for (uint32_t k = 0; k < MW_CHUNK / 2u; ++k) {
            const u32x4 st = *(const u32x4*)(mine + k * 16u), cm = *(const u32x4*)(mine + MW_MODEL_BYTES + k * 16u);
            const u32x2 stt = *(const u32x2*)(mine + MW_CHUNK * 8u + k * 8u), cmt = *(const u32x2*)(mine + MW_MODEL_BYTES + MW_CHUNK * 8u + k * 8u);
            // No test against the stream's length: past its end the walk chews on whatever the staging buffers hold (loads are clamped to
            // mapped memory, integer arithmetic does not trap), its Weights are never used again and the pairs are not stored (the
            // store-out below tests the length) -- and without 32 branches per chunk the compiler schedules across positions.
            outp[4u * k] = mix_nibble(w, half ? st.y : st.x, (stt.x >> hs) & 0xffffu, half ? cm.y : cm.x, (cmt.x >> hs) & 0xffffu);
            outp[4u * k + 2u] = mix_nibble(w, half ? st.w : st.z, (stt.y >> hs) & 0xffffu, half ? cm.w : cm.z, (cmt.y >> hs) & 0xffffu);
        }
        __syncthreads();
        // pairs out: load-shaped again, 16 bytes = both nibbles of two positions per lane
#pragma unroll
        for (uint32_t i = 0; i < MW_XG; ++i) {
            const uint32_t j = i * MW_XS + xj, p = p0 + 2u * xq;
            const uint32_t slen = s0 + j < b.n_streams ? (b.in_sizes ? b.in_sizes[s0 + j] : b.stream_len) : 0u;
            if (p < slen)   // an odd stream's last quad carries one stale pair: it stays inside the (even) slot and is never read
                *(u32x4*)(b.sf + (size_t)(s0 + j) * b.sf_stride + 2u * p) = *(const u32x4*)(lds_out + j * MW_OUT_STRIDE + xq * 16u);
        }
        if (c + 1u < chunks) { MW_STAGE(buf ^ 1u) }
        __syncthreads();
    }
#undef MW_FETCH
#undef MW_STAGE
}

hipError_t launch_bucket_mix_model(const MixBucketBatch& b, uint32_t num_cus, hipStream_t st) {
    BucketBatch v;                       // the view the shared task-list and unsort kernels take
    v.in = b.in; v.in_offsets = b.in_offsets; v.in_sizes = b.in_sizes;
    v.n_streams = b.n_streams; v.stream_len = b.stream_len; v.max_stream_len = b.max_stream_len; v.pieces = b.pieces;
    v.slot = b.slot; v.sf_stride = 2u * b.pos_stride;   // pos_stride == slot: the unsort below is in place
    v.sorted = nullptr; v.inv = b.inv; v.desc = b.desc; v.sfs = nullptr; v.sf = nullptr; v.tasks = b.tasks; v.counters = b.counters;
    v.inc = 0; v.lim = 0;
    for (int model = 0; model < 2; ++model) {
        hipError_t e = hipMemsetAsync(b.counters, 0, 64, st);
        if (e != hipSuccess) return e;
        if (b.pieces < 8u) {
            e = hipMemsetAsync(b.desc, 0, (size_t)b.n_streams * 256u * 8u * sizeof(uint32_t), st);
            if (e != hipSuccess) return e;
        }
        if (model == 0) {
            hipLaunchKernelGGL(mix_sort_kernel<0>, dim3(b.n_streams * b.pieces), dim3(BK_SORT_THREADS), 0, st, b);
            launch_bucket_tasks(v, st);
            hipLaunchKernelGGL(mix_chain_kernel<0>, dim3(num_cus * MX_CHAIN_WAVES), dim3(64), MxGeom<0>::LDS_BYTES, st, b);
        } else {
            hipLaunchKernelGGL(mix_sort_kernel<1>, dim3(b.n_streams * b.pieces), dim3(BK_SORT_THREADS), 0, st, b);
            launch_bucket_tasks(v, st);
            hipLaunchKernelGGL(mix_chain_kernel<1>, dim3(num_cus * MX_CHAIN_WAVES), dim3(64), MxGeom<1>::LDS_BYTES, st, b);
        }
        v.sfs = b.xs[model]; v.sf = (uint32_t*)b.xs[model]; v.sf_stride = 2u * b.pos_stride;
        launch_bucket_unsort(v, st);
        v.sfs = (bk_u32x2*)b.maxes[model]; v.sf = b.maxes[model]; v.sf_stride = b.pos_stride;
        launch_bucket_unsort32(v, st);
    }
    if ((b.n_streams + 31u) / 32u > num_cus * 4u) hipLaunchKernelGGL(mix_weights_kernel<2>, dim3((b.n_streams + 31u) / 32u), dim3(64), 0, st, b);
    else hipLaunchKernelGGL(mix_weights_kernel<1>, dim3((b.n_streams + 31u) / 32u), dim3(64), 0, st, b);
    return hipGetLastError();
}

}  // namespace divans_hip
