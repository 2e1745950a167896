// lit_kernels_p8.hip -- packed variant of the non-mixing literal coder: EIGHT lanes own one stream and each lane
// holds two consecutive CDF entries as one dword (cdf[2j] | cdf[2j+1] << 16), so a wave64 carries 8 streams.
// The per-stream scalar work (rANS state, row addressing, context) is then shared by twice as many streams per
// wave instruction, and blend runs on both halves of the dword at once.  Same arithmetic, same tables in HBM
// (a row is 16 x i16 = 8 dwords), same results as lit_kernels.hip; see that file for the reference citations.
//
//   lane j of a stream (j = lane & 7): entries 2j and 2j+1;  a DPP row (16 lanes) holds two streams.
#include "lit_device.h"

namespace divans_hip {

constexpr int P8_GROUPS = LIT_THREADS / 8;   // streams per 256-thread workgroup

// cdf[15] of the stream (= high half of its last lane), broadcast to its 8 lanes: two masked row_newbcast
__device__ __forceinline__ int p8_max(int c) {
    int m = __builtin_amdgcn_update_dpp(0, c, DPP_ROW_BCAST(7), 0xf, 0x3, false);     // lanes 0-7  <- lane 7
    m = __builtin_amdgcn_update_dpp(m, c, DPP_ROW_BCAST(15), 0xf, 0xc, false);        // lanes 8-15 <- lane 15
    return (int)((uint32_t)m >> 16);
}

__device__ __forceinline__ int p8_default_row(int j) { return (8 * j + 4) | ((8 * j + 8) << 16); }

struct P8Ref { uint32_t row; uint32_t slot_addr; };

// table access: one dword per lane; CACHE 0 = none, 2 = LDS cache for high-nibble rows only (same 2-way write-back
// organisation and tag format as Table<> in lit_kernels.hip)
template <int CACHE>
struct Table8 {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane_off;          // stream slab offset + 4 * lane-in-stream
    uint8_t* lds;
    uint32_t data_off, tag_off, set_mask;
    __device__ __forceinline__ int gload(uint32_t row) const {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off + (row << 5), 0, 0);
    }
    __device__ __forceinline__ void gstore(uint32_t row, int v) const {
        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, rsrc, lane_off + (row << 5), 0, 0);
    }
    template <bool HIGH>
    __device__ __forceinline__ int load(uint32_t row, P8Ref& ref) const {
        ref.row = row;
        if (CACHE == 0 || !HIGH) return gload(row);
        const uint32_t set = (row ^ (row >> 4) ^ (row >> 9)) & set_mask;
        uint32_t* tagp = (uint32_t*)(lds + tag_off + (set << 2));
        uint32_t tp = *tagp;
        const uint32_t t0 = tp & 0x7fffu, t1 = (tp >> 15) & 0x7fffu;
        const bool h0 = t0 == row, h1 = t1 == row;
        const uint32_t way = h0 ? 0u : (h1 ? 1u : ((tp >> 31) ^ 1u));
        ref.slot_addr = data_off + (((set << 1) + way) << 5);
        int v = (int)*(uint32_t*)(lds + ref.slot_addr);
        if (!(h0 || h1)) {
            const uint32_t victim = way ? t1 : t0;
            if (victim != 0x7fffu) gstore(victim, v);
            v = gload(row);
            tp = way ? ((tp & ~(0x7fffu << 15)) | (row << 15)) : ((tp & ~0x7fffu) | row);
        }
        *tagp = (tp & 0x7fffffffu) | (way << 31);
        return v;
    }
    template <bool HIGH>
    __device__ __forceinline__ void store(const P8Ref& ref, int v) const {
        if (CACHE != 0 && HIGH) *(uint32_t*)(lds + ref.slot_addr) = (uint32_t)v;
        else gstore(ref.row, v);
    }
    __device__ __forceinline__ void reset_cache(int j) const {
        if (CACHE == 0) return;
        for (uint32_t s = (uint32_t)j; s <= set_mask; s += 8u) *(uint32_t*)(lds + tag_off + (s << 2)) = 0x3fffffffu;
    }
};

template <int CACHE>
__device__ __forceinline__ Table8<CACHE> p8_make_table(const LitBatch& b, uint8_t* lds, int j) {
    const uint32_t slab = b.geom.total_rows * 32u;
    const uint32_t g = threadIdx.x >> 3;
    Table8<CACHE> t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((uint8_t*)b.tables + (size_t)blockIdx.x * P8_GROUPS * slab), 0,
                                               P8_GROUPS * slab, 0x00020000);
    t.lane_off = g * slab + 4u * (uint32_t)j;
    t.lds = lds;
    const uint32_t per_stream = b.cache_rows_high * 34u;
    t.data_off = g * per_stream + 4u * (uint32_t)j;
    t.tag_off = g * per_stream + b.cache_rows_high * 32u;
    t.set_mask = (b.cache_rows_high >> 1) - 1u;
    return t;
}

template <int CACHE>
__device__ __forceinline__ void p8_init_table(const Table8<CACHE>& t, uint32_t rows, int j) {
    // a default row is 32 bytes = two 16-byte halves; even lanes write first halves, odd lanes second halves
    const u32x4 lo = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
    const u32x4 hi = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};
    const u32x4 v = (j & 1) ? hi : lo;
    const uint32_t base = t.lane_off - 4u * (uint32_t)j + 16u * (uint32_t)j;
    for (uint32_t i = 0; i < rows * 32u; i += 128u)
        if (i + 16u * (uint32_t)j < rows * 32u) __builtin_amdgcn_raw_buffer_store_b128(v, t.rsrc, base + i, 0, 0);
    t.reset_cache(j);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
}

struct Fetched8 { P8Ref ref; int value; bool is_default; };

template <bool HIGH, int MM, int CACHE>
__device__ __forceinline__ Fetched8 p8_fetch(const LitGeometry& g, const LdsView& lv, const Table8<CACHE>& tb,
                                             uint32_t ctx, uint64_t last8, uint32_t hi_nib) {
    const RowSel rs = select_rows<HIGH, MM>(g, lv.mix, ctx, last8, hi_nib);
    Fetched8 f;
    f.value = tb.template load<HIGH>(rs.stride_row, f.ref);
    f.is_default = (MM < 0 || MM == 2) && rs.is_default;
    return f;
}

// probability/interface.rs:136-198 on a packed row: number of entries i < 15 with rescaled >= cdf[i]
__device__ __forceinline__ int p8_search(int c, int mx, uint32_t slot, int j, int sbase) {
    const int rescaled = (int)((uint32_t)__umul24(slot, (uint32_t)mx) >> 15);
    const int lo = c & 0xffff, hi = (int)((uint32_t)c >> 16);
    const unsigned long long blo = __ballot(rescaled >= lo);
    const unsigned long long bhi = __ballot(rescaled >= hi && j != 7);
    return __popc((uint32_t)(blo >> sbase) & 0xffu) + __popc((uint32_t)(bhi >> sbase) & 0xffu);
}

// probability/interface.rs:97-108 for the coded symbol: (start | freq << 16), uniform over the stream's lanes
__device__ __forceinline__ uint32_t p8_start_freq(int c, int mx, int sym, int j, int sbase) {
    const float rcp = biased_rcp15(mx);
    const int lo = c & 0xffff, hi = (int)((uint32_t)c >> 16);
    const int dlo = (int)scaled_div(lo, mx, rcp), dhi = (int)scaled_div(hi, mx, rcp);
    int dprev = row_prev_or_zero(dhi);          // scaled cdf of entry 2j-1 (previous lane's high half)
    dprev = j == 0 ? 0 : dprev;                 // lane 8 of the DPP row starts the second stream
    const bool odd = (sym & 1) != 0;
    const int base = odd ? dlo : dprev, top = odd ? dhi : dlo;
    const uint32_t sf = (uint32_t)(base + 1) | ((uint32_t)(top - base - 1) << 16);
    return (uint32_t)__builtin_amdgcn_ds_bpermute((sbase + (sym >> 1)) << 2, (int)sf);
}

// frequentist_cdf.rs:74-85 on both halves of the dword (no half can carry or borrow: values stay below 2^15 + 16)
__device__ __forceinline__ int p8_blend(int c, int j, int sym, int inc, int lim) {
    const int e = 2 * j + 1 - sym;              // entry 2j+1 >= sym  <=>  e >= 0 ; entry 2j >= sym  <=>  e >= 1
    const int add = (e >= 1 ? inc : 0) | (e >= 0 ? (inc << 16) : 0);
    c += add;
    const int mx = p8_max(c);
    const int t = c + ((2 * j + 1) | ((2 * j + 2) << 16));
    const int renorm = t - (int)(((uint32_t)t >> 2) & 0x3fff3fffu);
    return mx >= lim ? renorm : c;
}

// ---------------------------------------------------------------------------------------------
// Encode, pass 1 (model), packed
// ---------------------------------------------------------------------------------------------
template <bool HIGH, int CACHE>
__device__ __forceinline__ uint32_t p8_model_finish(const LitGeometry& g, const Table8<CACHE>& tb, int j, int sbase,
                                                    const Fetched8& f, int sym) {
    const int cv = f.is_default ? p8_default_row(j) : f.value;
    const int mx = p8_max(cv);
    const uint32_t packed = p8_start_freq(cv, mx, sym, j, sbase);
    int st = f.value;
    if (!f.is_default) st = p8_blend(st, j, sym, g.inc0, g.lim0);
    if ((CACHE != 0 && HIGH) || !f.is_default) tb.template store<HIGH>(f.ref, st);
    return packed;
}

template <int MM, bool CTXC, int CACHE>
__global__ __launch_bounds__(LIT_THREADS) void lit_model_encode_p8_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, j = lane & 7, sbase = lane & 56;
    const uint32_t gg = blockIdx.x * P8_GROUPS + (threadIdx.x >> 3);
    const uint32_t G = gridDim.x * P8_GROUPS;
    const Table8<CACHE> tb = p8_make_table<CACHE>(b, lds, j);
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint8_t* in = b.in + (b.in_offsets ? b.in_offsets[s] : (uint64_t)s * b.stream_len);
        const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
        uint32_t* sf = b.sf + (size_t)s * 2u * b.max_stream_len;
        p8_init_table(tb, g.total_rows, j);
        uint64_t last8 = 0;
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS];
        // each lane holds one literal byte of the current and of the next 8-byte window
        uint32_t mine = ((uint32_t)j < len) ? in[j] : 0u;
        uint32_t nxt = (8u + j < len) ? in[8u + j] : 0u;
        uint32_t cur = (uint32_t)__builtin_amdgcn_ds_bpermute(sbase << 2, (int)mine);
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, 0u, k1);
        Fetched8 rowH = p8_fetch<true, MM, CACHE>(g, lv, tb, ctx_cur, 0ull, 0u);
        Fetched8 rowL = p8_fetch<false, MM, CACHE>(g, lv, tb, ctx_cur, 0ull, cur >> 4);
        for (uint32_t base = 0; base < len; base += 8) {
            const uint32_t cnt = len - base < 8u ? len - base : 8u;
            uint32_t pend_a = 0, pend_b = 0;   // lane k keeps the two (start,freq) pairs of byte base+k
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t byte = cur;
                const uint32_t nb = (uint32_t)__builtin_amdgcn_ds_bpermute((sbase + (int)((k + 1u) & 7u)) << 2, (int)(k + 1u < 8u ? mine : nxt));
                const uint32_t ph = p8_model_finish<true, CACHE>(g, tb, j, sbase, rowH, (int)(byte >> 4));
                const uint32_t prev = (uint32_t)(last8 >> 56);
                if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + prev];
                last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, byte, k1);
                rowH = p8_fetch<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);
                const uint32_t pl = p8_model_finish<false, CACHE>(g, tb, j, sbase, rowL, (int)(byte & 15u));
                rowL = p8_fetch<false, MM, CACHE>(g, lv, tb, ctx_cur, last8, nb >> 4);
                cur = nb;
                pend_a = (uint32_t)j == k ? ph : pend_a;
                pend_b = (uint32_t)j == k ? pl : pend_b;
            }
            if ((uint32_t)j < cnt) {
                u32x2 v = {pend_a, pend_b};
                __builtin_nontemporal_store(v, (u32x2*)(sf + 2u * (size_t)(base + j)));
            }
            mine = nxt;
            nxt = (base + 16u + j < len) ? in[base + 16u + j] : 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decode, packed
// ---------------------------------------------------------------------------------------------
struct WordWindow8 {   // 8 upcoming coded words per stream (one per lane) + the 8 after them
    const uint32_t* in; uint32_t nwords; uint32_t base, pos; uint32_t w, wn;
    __device__ __forceinline__ uint32_t fetch(uint32_t first, int j) const { return (first + j < nwords) ? in[first + j] : 0u; }
    __device__ __forceinline__ void start(int j) { base = 0; pos = 0; w = fetch(0, j); wn = fetch(8, j); }
    __device__ __forceinline__ uint32_t next(int j, int sbase) {
        uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((sbase + (int)(pos - base)) << 2, (int)w);
        pos += 1;
        if (pos - base == 8u) { base = pos; w = wn; wn = fetch(base + 8u, j); }
        return v;
    }
};

template <bool HIGH, int CACHE>
__device__ __forceinline__ void p8_finish_nibble(const LitGeometry& g, const Table8<CACHE>& tb, int j, int sbase,
                                                 const Fetched8& f, int cv, int mx, int sym, uint64_t& S) {
    const uint32_t slot = (uint32_t)S & 0x7fffu;
    const uint32_t packed = p8_start_freq(cv, mx, sym, j, sbase);
    const uint32_t start = packed & 0xffffu, freq = packed >> 16;
    S = (uint64_t)freq * (S >> 15) + (uint64_t)slot - (uint64_t)start;
    int st = f.value;
    if (!f.is_default) st = p8_blend(st, j, sym, g.inc0, g.lim0);
    if ((CACHE != 0 && HIGH) || !f.is_default) tb.template store<HIGH>(f.ref, st);
}

template <int MM, bool CTXC, int CACHE>
__global__ __launch_bounds__(LIT_THREADS) void lit_decode_p8_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, j = lane & 7, sbase = lane & 56;
    const uint32_t gg = blockIdx.x * P8_GROUPS + (threadIdx.x >> 3);
    const uint32_t G = gridDim.x * P8_GROUPS;
    const Table8<CACHE> tb = p8_make_table<CACHE>(b, lds, j);
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint32_t len = b.out_sizes ? b.out_sizes[s] : b.stream_len;
        uint8_t* out = b.out + (b.out_offsets ? b.out_offsets[s] : (uint64_t)s * b.stream_len);
        WordWindow8 ww;
        ww.in = (const uint32_t*)(b.in + b.in_offsets[s]);
        ww.nwords = b.in_sizes[s] >> 2;
        ww.start(j);
        p8_init_table(tb, g.total_rows, j);
        uint64_t last8 = 0;
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS];
        uint64_t SA = 0, SB = 0;
        bool corrupt = false;
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, 0u, k1);
        Fetched8 rowH = p8_fetch<true, MM, CACHE>(g, lv, tb, ctx_cur, 0ull, 0u);
        for (uint32_t cbeg = 0; cbeg < len; cbeg += 32768u) {
            {
                uint32_t a0 = ww.next(j, sbase), a1 = ww.next(j, sbase), b0 = ww.next(j, sbase), b1 = ww.next(j, sbase);
                SA = ((uint64_t)a1 << 32) | a0;
                SB = ((uint64_t)b1 << 32) | b0;
            }
            const uint32_t cend = cbeg + 32768u < len ? cbeg + 32768u : len;
            for (uint32_t base = cbeg; base < cend; base += 8u) {
                const uint32_t cnt = cend - base < 8u ? cend - base : 8u;
                uint32_t outb = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    if (SA < (1ull << 31)) SA = (SA << 32) | ww.next(j, sbase);
                    const int cvh = rowH.is_default ? p8_default_row(j) : rowH.value;
                    const int mxh = p8_max(cvh);
                    const uint32_t hi = (uint32_t)p8_search(cvh, mxh, (uint32_t)SA & 0x7fffu, j, sbase);
                    const Fetched8 rowL = p8_fetch<false, MM, CACHE>(g, lv, tb, ctx_cur, last8, hi);
                    p8_finish_nibble<true, CACHE>(g, tb, j, sbase, rowH, cvh, mxh, (int)hi, SA);
                    if (SB < (1ull << 31)) SB = (SB << 32) | ww.next(j, sbase);
                    const int cvl = rowL.is_default ? p8_default_row(j) : rowL.value;
                    const int mxl = p8_max(cvl);
                    const uint32_t lo = (uint32_t)p8_search(cvl, mxl, (uint32_t)SB & 0x7fffu, j, sbase);
                    const uint32_t byte = (hi << 4) | lo;
                    const uint32_t prev = (uint32_t)(last8 >> 56);
                    if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + prev];
                    last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                    ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, byte, k1);
                    rowH = p8_fetch<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);
                    p8_finish_nibble<false, CACHE>(g, tb, j, sbase, rowL, cvl, mxl, (int)lo, SB);
                    outb = (uint32_t)j == k ? byte : outb;
                }
                if ((uint32_t)j < cnt) __builtin_nontemporal_store((uint8_t)outb, out + base + j);
            }
            corrupt |= (SA != (1ull << 31)) | (SB != (1ull << 31));   // see lit_decode_kernel
        }
        corrupt |= ww.pos != ww.nwords;
        if (corrupt && j == 0 && b.status) atomicOr(b.status, LIT_STATUS_BAD_STREAM);
    }
}

// ---------------------------------------------------------------------------------------------
typedef void (*LitKernel)(const LitBatch);

#define P8_PICK(KERNEL)                                                                          \
    template <int CACHE>                                                                         \
    static LitKernel pick_##KERNEL(int mm, bool ctxc) {                                          \
        const int key = (mm == 4 ? 2 : (mm == 0 ? 1 : 0)) * 2 + (ctxc ? 1 : 0);                  \
        switch (key) {                                                                           \
        case 0: return KERNEL<-1, false, CACHE>; case 1: return KERNEL<-1, true, CACHE>;         \
        case 2: return KERNEL<0, false, CACHE>;  case 3: return KERNEL<0, true, CACHE>;          \
        case 4: return KERNEL<4, false, CACHE>;  default: return KERNEL<4, true, CACHE>;         \
        }                                                                                        \
    }
P8_PICK(lit_model_encode_p8_kernel)
P8_PICK(lit_decode_p8_kernel)

static int p8_effective_mm(int mm) { return (mm == 0 || mm == 4) ? mm : -1; }

uint32_t lit_lds_bytes_p8(const LitBatch& b) {
    uint32_t bytes = b.cache_mode ? P8_GROUPS * b.cache_rows_high * 34u : 0u;
    if (b.geom.ctx_const < 0) bytes += LIT_BLOB_CTXF + LIT_CTXF_BYTES * b.geom.n_btypes;
    if (p8_effective_mm(b.geom.mm_uniform) < 0) bytes += 8192u;
    return bytes;
}

static hipError_t launch_p8(LitKernel k, const LitBatch& b_in, uint32_t blocks, hipStream_t st) {
    LitBatch b = b_in;
    b.cache_bytes_per_wg = b.cache_mode ? P8_GROUPS * b.cache_rows_high * 34u : 0u;   // load_config_to_lds places the tables after the caches
    const uint32_t lds = lit_lds_bytes_p8(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LIT_THREADS), lds, st, b);
    return hipGetLastError();
}

// cache_mode: 0 = none, anything else = high-nibble-row cache of cache_rows_high rows
hipError_t launch_model_encode_p8(const LitBatch& b, uint32_t blocks, hipStream_t st) {
    const int mm = p8_effective_mm(b.geom.mm_uniform);
    LitKernel k = b.cache_mode ? pick_lit_model_encode_p8_kernel<2>(mm, b.geom.ctx_const >= 0)
                               : pick_lit_model_encode_p8_kernel<0>(mm, b.geom.ctx_const >= 0);
    return launch_p8(k, b, blocks, st);
}
hipError_t launch_decode_p8(const LitBatch& b, uint32_t blocks, hipStream_t st) {
    const int mm = p8_effective_mm(b.geom.mm_uniform);
    LitKernel k = b.cache_mode ? pick_lit_decode_p8_kernel<2>(mm, b.geom.ctx_const >= 0)
                               : pick_lit_decode_p8_kernel<0>(mm, b.geom.ctx_const >= 0);
    return launch_p8(k, b, blocks, st);
}

}  // namespace divans_hip
