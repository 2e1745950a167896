// lit_bucket.hip -- bucketed model pass of the ENCODER for the order-1, context-map-off configuration
// (mixing value 4 everywhere, constant context: BASELINE configs[1], reference TestSimple, bin/benchmark.rs:195-206).
//
// In that configuration the high nibble of byte i is coded with row [prev] and its low nibble with row [prev][hi]
// (codec/literal.rs:176-208 with mm_opts == 4), both blended with literal_adaptation[0] (literal.rs:320,354).  A row's
// CDF therefore only depends on the earlier positions that share its `prev` byte.  The encoder knows every byte
// up front, so instead of walking the stream serially against a 139 KB table in HBM (lit_model_encode_kernel) it
//   1. bucket_sort_kernel    stable counting sort of every 8 KiB piece of a stream by prev byte (LDS),
//   2. bucket_tasks_kernel   turns the non-empty (stream, prev) buckets into a task list, longest first,
//   3. bucket_chain_kernel   ONE LANE per bucket walks its positions in order with the bucket's 17 rows
//                            (1 high + 16 low) in LDS -- no table in HBM, no row cache, no cross-lane traffic,
//   4. bucket_unsort_kernel  puts the (start,freq) pairs back into position order for the rANS pass.
// The arithmetic per nibble is the same as everywhere else (probability/interface.rs:97-108, frequentist_cdf.rs:74-85).
// The decoder cannot do this (it learns the bytes one at a time) and keeps the streaming kernels.
#include "lit_bucket_dev.h"

namespace divans_hip {

// ---------------------------------------------------------------------------------------------
// 1. per (stream, piece): sorted[slot] = byte, inv[pos] = slot, desc[stream][prev][piece] = start | count << 16
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BK_SORT_THREADS) void bucket_sort_kernel(const BucketBatch b) {
    __shared__ __attribute__((aligned(16))) uint8_t staging[BK_PIECE];
    __shared__ __attribute__((aligned(16))) uint8_t piece_in[16 + BK_PIECE];   // piece_in[15] = the byte before the piece
    __shared__ uint32_t hist[4][256];
    __shared__ uint32_t scan[256];
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
    // the piece (and the byte before it: the first key) once into LDS, 16 bytes per lane where alignment allows
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
        if (tid == 0u) piece_in[15] = base ? in[base - 1u] : 0u;
    }
    __syncthreads();
    const uint8_t* key_of = piece_in + 15;      // key_of[p] = previous byte of position p, key_of[p + 1] = its own byte
    // wave w owns positions [2048 w, 2048 w + 2048) of the piece and visits them in order, 64 at a time
    for (uint32_t bt = 0; bt < 32u; ++bt) {
        const uint32_t p = w * 2048u + bt * 64u + lane;
        if (p < n) atomicAdd(&hist[w][key_of[p]], 1u);
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
        const uint32_t key = valid ? key_of[p] : 0u;
        const uint32_t byte = valid ? key_of[p + 1u] : 0u;
        unsigned long long same = __ballot(valid);
        for (uint32_t bit = 0; bit < 8u; ++bit) {
            const bool set = (key >> bit) & 1u;
            const unsigned long long bb = __ballot(set);
            same &= set ? bb : ~bb;
        }
        const uint32_t rank = lanes_below(same), cnt = (uint32_t)__popcll(same);
        if (valid) {
            const uint32_t off = hist[w][key];
            staging[off + rank] = (uint8_t)byte;
            inv[p] = (uint16_t)(off + rank);
            if (rank == cnt - 1u) hist[w][key] = off + cnt;   // the wave's LDS accesses stay in program order
        }
    }
    __syncthreads();
    uint8_t* sorted = b.sorted + (size_t)s * pl + base;
    for (uint32_t i = tid * 16u; i < n; i += BK_SORT_THREADS * 16u) {
        if (i + 16u <= n) *(u32x4*)(sorted + i) = *(const u32x4*)(staging + i);
        else for (uint32_t k = i; k < n; ++k) sorted[k] = staging[k];
    }
}

// ---------------------------------------------------------------------------------------------
// 2. task lists by bucket size class, so that the long chains start first
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bucket_tasks_kernel(const BucketBatch b) {
    __shared__ uint32_t cnt[BK_CLASSES], basev[BK_CLASSES];
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x;      // stream * 256 + prev
    const uint32_t cap = b.n_streams * 256u;
    if (threadIdx.x < BK_CLASSES) cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t tot = 0;
    if (t < cap) {
        const u32x4* d = (const u32x4*)(b.desc + (size_t)t * 8u);
        const u32x4 d0 = d[0], d1 = d[1];
        tot = (d0.x >> 16) + (d0.y >> 16) + (d0.z >> 16) + (d0.w >> 16) + (d1.x >> 16) + (d1.y >> 16) + (d1.z >> 16) + (d1.w >> 16);
    }
    const int cls = bk_class_of(tot);
    // slot inside the block: one LDS atomic per wave and class, then one global atomic per block and class
    uint32_t local = 0;
    for (int c = 0; c < (int)BK_CLASSES; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t wbase = 0;
        if ((int)(threadIdx.x & 63u) == leader) wbase = atomicAdd(&cnt[c], (uint32_t)__popcll(m));
        wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, leader);
        if (cls == c) local = wbase + lanes_below(m);
    }
    __syncthreads();
    if (threadIdx.x < BK_CLASSES) basev[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&b.counters[threadIdx.x], cnt[threadIdx.x]) : 0u;
    __syncthreads();
    if (cls >= 0) b.tasks[(size_t)cls * cap + basev[cls] + local] = t;
}

// ---------------------------------------------------------------------------------------------
// 3. chains
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bucket_chain_kernel(const BucketBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    const uint32_t lane = threadIdx.x;
    uint32_t* my = lds32 + lane * BK_LANE_DWORDS;
    uint32_t* mydesc = my + BK_DESC_DW;
    uint32_t* tab = lds32 + BK_TAB_DW;
    const uint32_t inc = (uint32_t)b.inc; const int lim = b.lim;
    for (uint32_t i = lane; i < 128u; i += 64u) {
        const uint32_t sym = i >> 3, k = i & 7u;
        tab[i] = (2u * k >= sym ? inc : 0u) | (2u * k + 1u >= sym ? inc << 16 : 0u);
    }
    __syncthreads();
    const size_t pl = b.slot;
    const uint32_t cap = b.n_streams * 256u;
    BkTaskLists lists; lists.load(b.counters);
    const uint32_t total = lists.total();
    const u32x4 def0 = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
    const u32x4 def1 = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};

    // per-lane chain state.  A bucket is up to eight runs of consecutive sorted slots, one per 8 KiB piece that holds some of its
    // positions; the task's descriptors are compacted to the non-empty ones when the lane takes it (mydesc[0 .. nruns)).
    bool has_task = false, exhausted = false;
    uint32_t run_i = 0, nruns = 0, left = 0, idx = 0;
    u32x2* cur_sfs = b.sfs; const uint8_t* cur_sorted = b.sorted;   // the current bucket's stream slot
    uint32_t nt_stage = 0, nt_tid = 0;
    u32x4 nd0 = {0u, 0u, 0u, 0u}, nd1 = {0u, 0u, 0u, 0u};
    // wave-uniform task window
    uint32_t win_cur = 0, win_end = 0, nxt_val = 0, nxt_w = 0;
    const uint32_t long_end = lists.ends[3];     // tasks of at least 2048 positions come first
    bool nxt_pending = false, drained = false;
    // The chain is bound by the vector-memory path, not by its arithmetic (profiles/r03c_chain_role_experiment.txt), so a lane
    // moves its bytes eight at a time: ONE aligned 8-byte load per iteration covers the sorted slots [base, base + 8), of which
    // the run owns [first, first + cnt); it is requested one iteration before it is coded.  The (start, freq) pairs of a group
    // leave at the top of the NEXT iteration -- four 16-byte stores for a full group -- through inline asm: the compiler then sees
    // one load per iteration and waits for it with vmcnt(0) at a point where the only other operations in flight are stores a whole
    // iteration old (a store is acknowledged out of order with loads, so no smaller count would prove the load complete).
    u32x2 e_next = {0u, 0u}; uint32_t m_next = 0;           // meta: base | first << 16 | cnt << 20 | BK_VALID
    u32x2 pv0 = {0u, 0u}, pv1 = pv0, pv2 = pv0, pv3 = pv0, pv4 = pv0, pv5 = pv0, pv6 = pv0, pv7 = pv0;
    uint32_t m_prev = 0; u32x2* sfs_prev = b.sfs;

#define BK_OPAQUE(X) asm volatile("" : "+v"(X))
#define BK_BYTE(K, WORD, PV)                                                                            \
    if (((K - first) & 15u) < cnt) {                                                                    \
        const uint32_t byte_ = (WORD >> (8u * (K & 3u))) & 0xffu;                                       \
        const uint32_t hi = byte_ >> 4, lo = byte_ & 15u;                                               \
        uint32_t* rowl = my + 8u * (1u + hi);                                                           \
        BkRow H = bk_read(my, tab, hi), L = bk_read(rowl, tab, lo);                                     \
        const u32x2 v = {bk_pack(H, hi), bk_pack(L, lo)};                                               \
        H.w0 += H.a0; H.w1 += H.a1; L.w0 += L.a0; L.w1 += L.a1;   /* frequentist_cdf.rs:75-78 */        \
        if ((int)(H.w1.w >> 16) >= lim) bk_renorm(H);                                                   \
        if ((int)(L.w1.w >> 16) >= lim) bk_renorm(L);                                                   \
        *(u32x4*)my = H.w0; *(u32x4*)(my + 4) = H.w1; *(u32x4*)rowl = L.w0; *(u32x4*)(rowl + 4) = L.w1; \
        PV = v;                                                                                         \
    }

    for (;;) {
        u32x2 e = e_next; const uint32_t m = m_next;
        // 1. the previous group's pairs leave (their registers are free again below)
        if (m_prev & BK_VALID) {
            u32x2* dst = sfs_prev + (m_prev & 0xffffu);
            const uint32_t pf = (m_prev >> 16) & 15u, pc = (m_prev >> 20) & 15u;
            if (pc == 8u) {
                const u32x4 q0 = {pv0.x, pv0.y, pv1.x, pv1.y}, q1 = {pv2.x, pv2.y, pv3.x, pv3.y};
                const u32x4 q2 = {pv4.x, pv4.y, pv5.x, pv5.y}, q3 = {pv6.x, pv6.y, pv7.x, pv7.y};
                bk_store_quad((u32x4*)dst, q0); bk_store_quad((u32x4*)(dst + 2), q1);
                bk_store_quad((u32x4*)(dst + 4), q2); bk_store_quad((u32x4*)(dst + 6), q3);
            } else {
                if (((0u - pf) & 15u) < pc) bk_store_pair(dst + 0, pv0);
                if (((1u - pf) & 15u) < pc) bk_store_pair(dst + 1, pv1);
                if (((2u - pf) & 15u) < pc) bk_store_pair(dst + 2, pv2);
                if (((3u - pf) & 15u) < pc) bk_store_pair(dst + 3, pv3);
                if (((4u - pf) & 15u) < pc) bk_store_pair(dst + 4, pv4);
                if (((5u - pf) & 15u) < pc) bk_store_pair(dst + 5, pv5);
                if (((6u - pf) & 15u) < pc) bk_store_pair(dst + 6, pv6);
                if (((7u - pf) & 15u) < pc) bk_store_pair(dst + 7, pv7);
            }
        }
        // 2. the next group of the run is requested (every lane issues exactly one load, from a harmless address if it has nothing
        //    to fetch), the next run of the bucket taken when this one is used up
        {
            const bool adv = has_task && left == 0u, more = run_i < nruns;
            const uint32_t d = mydesc[run_i & 7u];
            if (adv && more) { left = d >> 16; idx = d & 0xffffu; ++run_i; }
            if (adv && !more) has_task = false;
        }
        {
            const bool fetch_ = has_task && left != 0u;
            const uint32_t base = idx & ~7u, first_ = idx & 7u;
            const uint32_t cnt_ = left < 8u - first_ ? left : 8u - first_;
            const uint8_t* lp = fetch_ ? cur_sorted + base : b.sorted;
            e_next = *(const u32x2*)lp;
            m_next = fetch_ ? (base | (first_ << 16) | (cnt_ << 20) | BK_VALID) : 0u;
            idx += fetch_ ? cnt_ : 0u; left -= fetch_ ? cnt_ : 0u;
        }
        // 3. this iteration's group
        BK_OPAQUE(e);      /* keeps the compiler from touching the bytes (and waiting for them) before this point */
        m_prev = m; sfs_prev = cur_sfs;
        if (m & BK_VALID) {
            const uint32_t first = (m >> 16) & 15u, cnt = (m >> 20) & 15u;
            BK_BYTE(0u, e.x, pv0) BK_BYTE(1u, e.x, pv1) BK_BYTE(2u, e.x, pv2) BK_BYTE(3u, e.x, pv3)
            BK_BYTE(4u, e.y, pv4) BK_BYTE(5u, e.y, pv5) BK_BYTE(6u, e.y, pv6) BK_BYTE(7u, e.y, pv7)
        }
        // 4. a lane whose bucket is finished -- the group coded above was its last: nothing of it is still to be requested or
        //    coded, only the pairs of that group wait for step 1 (with sfs_prev) -- takes its prefetched task
        if (!has_task && !(m_next & BK_VALID) && nt_stage == 3u) {
            BK_OPAQUE(nd0); BK_OPAQUE(nd1);
            uint32_t n = 0;
            const uint32_t dsc[8] = {nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, nd1.z, nd1.w};
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) if (dsc[j] >> 16) { mydesc[n] = dsc[j] + j * BK_PIECE; ++n; }   // first slot + piece base < 65536
            nruns = n; run_i = 0u;
            for (uint32_t r = 0; r < 17u; ++r) { *(u32x4*)(my + 8u * r) = def0; *(u32x4*)(my + 8u * r + 4u) = def1; }
            cur_sfs = b.sfs + (size_t)(nt_tid >> 8) * pl; cur_sorted = b.sorted + (size_t)(nt_tid >> 8) * pl;
            left = 0u; has_task = true; nt_stage = 0u;
        }
        // task prefetch pipeline, one stage per iteration so that no load is waited for in the iteration that issued it
        const bool want = nt_stage == 0u && !exhausted;
        if (nt_stage == 2u) nt_stage = 3u;
        else if (nt_stage == 1u) {
            BK_OPAQUE(nt_tid);
            const u32x4* dp = (const u32x4*)(b.desc + (size_t)nt_tid * 8u);
            nd0 = dp[0]; nd1 = dp[1];
            nt_stage = 2u;
        }
        const unsigned long long wm = __ballot(want);
        if (wm) {
            if (win_cur == win_end && nxt_pending) {
                BK_OPAQUE(nxt_val);
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
#undef BK_BYTE
#undef BK_OPAQUE
}

// ---------------------------------------------------------------------------------------------
// 4. back to position order: sf[stream][pos] = sfs[stream][piece][inv[pos]]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bucket_unsort_kernel(const BucketBatch b) {
    __shared__ u32x2 buf[BK_PIECE];
    const uint32_t s = blockIdx.x / b.pieces, piece = blockIdx.x % b.pieces;
    const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
    const uint32_t base = piece * BK_PIECE;
    if (base >= len) return;
    const uint32_t n = len - base < BK_PIECE ? len - base : BK_PIECE;
    const size_t pl = b.slot;
    const u32x2* src = b.sfs + (size_t)s * pl + base;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) buf[i] = __builtin_nontemporal_load(src + i);
    __syncthreads();
    const uint16_t* inv = b.inv + (size_t)s * pl + base;
    u32x2* dst = (u32x2*)(b.sf + (size_t)s * b.sf_stride) + base;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) dst[i] = buf[inv[i]];
}

// the same for a plane of 4-byte elements (the two-model pass's maxes): sfs / sf are read as uint32_t arrays, `slot` / sf_stride apart
__global__ __launch_bounds__(1024) void bucket_unsort32_kernel(const BucketBatch b) {
    __shared__ uint32_t buf[BK_PIECE];
    const uint32_t s = blockIdx.x / b.pieces, piece = blockIdx.x % b.pieces;
    const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
    const uint32_t base = piece * BK_PIECE;
    if (base >= len) return;
    const uint32_t n = len - base < BK_PIECE ? len - base : BK_PIECE;
    const size_t pl = b.slot;
    const uint32_t* src = (const uint32_t*)b.sfs + (size_t)s * pl + base;
    const uint32_t n2 = (n + 1u) & ~1u;                    // pieces start on even elements and the slot is even: whole pairs
    for (uint32_t i = 2u * threadIdx.x; i < n2; i += 2048u) *(u32x2*)(buf + i) = __builtin_nontemporal_load((const u32x2*)(src + i));
    __syncthreads();
    const uint16_t* inv = b.inv + (size_t)s * pl + base;
    uint32_t* dst = b.sf + (size_t)s * b.sf_stride + base;
    for (uint32_t i = 2u * threadIdx.x; i < n2; i += 2048u) {
        const uint32_t iv = *(const uint32_t*)(inv + i);
        const u32x2 v = {buf[iv & 0xffffu], i + 1u < n ? buf[iv >> 16] : 0u};
        *(u32x2*)(dst + i) = v;
    }
}

uint32_t bucket_chain_lds_bytes() { return (64u * BK_LANE_DWORDS + 128u) * 4u; }

// steps 2 and 4 on their own, for the two-model pass (lit_bucket_mix.hip) that brings its own sort and chain kernels
void launch_bucket_tasks(const BucketBatch& b, hipStream_t st) {
    hipLaunchKernelGGL(bucket_tasks_kernel, dim3((b.n_streams + 3u) / 4u), dim3(1024), 0, st, b);
}
void launch_bucket_unsort(const BucketBatch& b, hipStream_t st) {
    hipLaunchKernelGGL(bucket_unsort_kernel, dim3(b.n_streams * b.pieces), dim3(1024), 0, st, b);
}

void launch_bucket_unsort32(const BucketBatch& b, hipStream_t st) {
    hipLaunchKernelGGL(bucket_unsort32_kernel, dim3(b.n_streams * b.pieces), dim3(1024), 0, st, b);
}

hipError_t launch_bucket_model(const BucketBatch& b, uint32_t chain_blocks, hipStream_t st) {
    hipError_t e = hipMemsetAsync(b.counters, 0, 64, st);
    if (e != hipSuccess) return e;
    if (b.pieces < 8u) {
        e = hipMemsetAsync(b.desc, 0, (size_t)b.n_streams * 256u * 8u * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(b.n_streams * b.pieces), dim3(BK_SORT_THREADS), 0, st, b);
    hipLaunchKernelGGL(bucket_tasks_kernel, dim3((b.n_streams + 3u) / 4u), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(bucket_chain_kernel, dim3(chain_blocks), dim3(64), bucket_chain_lds_bytes(), st, b);
    hipLaunchKernelGGL(bucket_unsort_kernel, dim3(b.n_streams * b.pieces), dim3(1024), 0, st, b);
    return hipGetLastError();
}

}  // namespace divans_hip
