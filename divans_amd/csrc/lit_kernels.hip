// lit_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the divans literal coder.
//
// Mapping: one 16-lane DPP row owns one stream; lane i of the row holds cdf[i] of
// the CDF row being coded, so blend / search / start-freq are one VALU op per step for all 16
// entries, and the stream's scalar state (rANS states, context bytes, weights) is replicated
// across the 16 lanes.  Four streams share a wave64, sixteen a 256-thread workgroup.  Each stream's
// CDF rows live in an HBM table private to its row of lanes (L2 / Infinity-Cache resident while the
// stream is in flight) and are reached with buffer_load/store_short through one per-workgroup SRD.
//
// What each device function restates (paths relative to the reference tree):
//   exact_div            probability/numeric.rs:25-31 (any exact division is bit-compatible, make_div_lut.rs:37-39)
//   start/freq           probability/interface.rs:97-108 sym_to_start_and_freq
//   symbol search        probability/interface.rs:136-198 cdf_offset_to_sym_start_and_freq
//   blend_row            probability/frequentist_cdf.rs:74-85
//   average_rows         probability/frequentist_cdf.rs:58-72
//   weights_update       codec/weights.rs:23-133
//   select_rows          codec/literal.rs:154-259 (index math of code_nibble)
//   decode_nibble        ans.rs:225-252 (get_nibble + helper_advance_sym), refill ans.rs:428-442
//   rans_encode_kernel   ans.rs:302-378 (reverse_put_sym / flush_chunk)
//
// Template parameters specialise the per-nibble index math at compile time:
//   MM   : the mixing value when the whole reachable mixing_mask is uniform (0..8), or -1 = look it up
//   CTXC : the context map is constant (context = geom.ctx_const), else context comes from the LDS tables
//   MIX  : CodecTraits::MIXING_PRIORS (specializations.rs:27-36)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#include "lit_device.h"

namespace divans_hip {

// Per-row view of the stream's CDF table: SRD of the workgroup's table slab + this lane's byte offset,
// fronted (CACHE) by private 2-way set-associative write-back row caches in LDS, one for the rows of the
// high-nibble table and one for the low-nibble table (they may alias = one unified cache, or the low one may be
// absent: high-nibble rows are few and hot -- 32 ways hold ~90 % of their accesses -- while low-nibble rows are many):
//   tag word per set = way0 row id (15 bits) | way1 row id << 15 | MRU way << 31, 0x7fff = empty way.
// Every table access of the coder is a read-modify-write of one whole row, so a cached row is always dirty;
// a miss writes the victim row back to HBM and fetches the new one.
struct RowRef { uint32_t row; uint32_t slot_addr; };
constexpr uint32_t kNoSlot = 0xffffffffu;

struct CacheDesc {
    uint32_t data_off;   // LDS byte offset of this stream's cached rows + 2 * lane-in-row
    uint32_t tag_off;    // LDS byte offset of this stream's tag words
    uint32_t set_mask;   // sets - 1, or 0xffffffff when this table is not cached
};

template <int CACHE>   // 0: no LDS cache, 1: one unified cache, 2: high-nibble rows only, 3: separate high / low caches
struct Table {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane_off;   // row-of-lanes slab offset + 2 * lane-in-row
    uint8_t* lds;        // workgroup LDS base
    CacheDesc ch, cl;    // high-nibble / low-nibble table
    __device__ __forceinline__ int gload(uint32_t row) const {
        return (int)__builtin_amdgcn_raw_buffer_load_b16(rsrc, lane_off + (row << 5), 0, 0);
    }
    __device__ __forceinline__ void gstore(uint32_t row, int v) const {
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, rsrc, lane_off + (row << 5), 0, 0);
    }
    __device__ __forceinline__ int load(uint32_t row, RowRef& ref, bool high) const {
        ref.row = row;
        ref.slot_addr = kNoSlot;
        if (CACHE == 0 || (CACHE == 2 && !high)) return gload(row);
        const CacheDesc& d = (CACHE == 1 || high) ? ch : cl;
        const uint32_t set = (row ^ (row >> 4) ^ (row >> 9)) & d.set_mask;
        uint32_t* tagp = (uint32_t*)(lds + d.tag_off + (set << 2));
        uint32_t tp = *tagp;
        const uint32_t t0 = tp & 0x7fffu, t1 = (tp >> 15) & 0x7fffu;
        const bool h0 = t0 == row, h1 = t1 == row;
        const uint32_t way = h0 ? 0u : (h1 ? 1u : ((tp >> 31) ^ 1u));
        ref.slot_addr = d.data_off + (((set << 1) + way) << 5);
        int v = (int)*(uint16_t*)(lds + ref.slot_addr);
        if (!(h0 || h1)) {
            const uint32_t victim = way ? t1 : t0;
            if (victim != 0x7fffu) gstore(victim, v);
            v = gload(row);
            tp = way ? ((tp & ~(0x7fffu << 15)) | (row << 15)) : ((tp & ~0x7fffu) | row);
        }
        *tagp = (tp & 0x7fffffffu) | (way << 31);
        return v;
    }
    __device__ __forceinline__ void store(const RowRef& ref, int v) const {
        if (CACHE == 0) gstore(ref.row, v);
        else if (CACHE == 1 || CACHE == 3) *(uint16_t*)(lds + ref.slot_addr) = (uint16_t)v;
        else if (ref.slot_addr != kNoSlot) *(uint16_t*)(lds + ref.slot_addr) = (uint16_t)v;   // CACHE == 2: slot_addr is compile-time known per call site
        else gstore(ref.row, v);
    }
    // start of a stream: every way empty
    __device__ __forceinline__ void reset_cache(int li) const {
        if (CACHE == 0) return;
        for (uint32_t s = (uint32_t)li; s <= ch.set_mask; s += 16u) *(uint32_t*)(lds + ch.tag_off + (s << 2)) = 0x3fffffffu;
        if (CACHE == 3)
            for (uint32_t s = (uint32_t)li; s <= cl.set_mask; s += 16u) *(uint32_t*)(lds + cl.tag_off + (s << 2)) = 0x3fffffffu;
    }
};

// frequentist_cdf.rs:74-85 on one entry per lane (li = lane index in row).  For speeds whose row totals never leave i16
// (divans_gpu_speed_supported) plain 32-bit arithmetic is the reference's.  For the others (`wc`, LitGeometry::wrap_check) the
// reference's i16 total wraps negative when it passes 0x7fff: `cdf[15] >= lim` is then false -- no renormalisation -- and the next
// nibble coded with that row gets a (start, freq) that is no distribution (ans.rs:281-285's debug_asserts; a release build divides by a
// zero freq or writes a stream its own decoder cannot read).  The lanes keep such a row recognisable instead of wrapping it: no
// renormalisation, its total stays above 0x7fff (saturating at 0xffff, rows are stored as u16), and wrap_scan() after the stream tells
// a row that merely ended there (harmless: the reference never looks at it again) from one that was coded with again.
// `_known_max`: the row's total before the update is already at hand (cdf[15] always takes the increment, so the renormalisation
// test needs no second broadcast)
// i16 leaves its range in two places of blend: the increment carries the total past 0x7fff, or -- the total still inside -- the
// renormalisation's `cdf[15] + 16` does (t = c + i + 1 is largest in entry 15); either way the reference's row is garbage from there on
__device__ __forceinline__ int blend_wrap_checked(int c, int li, int lim, int renorm, int nt) {
    c = c > 0xffff ? 0xffff : c;
    if (nt > 0x7fff) return c;                                   // wrapped by the increment: no renormalisation (the i16 total is negative)
    if (nt < lim) return c;
    if (nt + 16 > 0x7fff) return li == 15 ? 0x8000 : c;          // wrapped inside the renormalisation: mark the total
    return renorm;
}
__device__ __forceinline__ int blend_row_known_max(int c, int li, int sym, int inc, int lim, int old_max, bool wc = false) {
    c = (li >= sym) ? c + inc : c;
    int t = c + li + 1;
    int renorm = t - (t >> 2);
    const int nt = old_max + inc;
    if (wc) return blend_wrap_checked(c, li, lim, renorm, nt);
    return nt >= lim ? renorm : c;
}
__device__ __forceinline__ int blend_row(int c, int li, int sym, int inc, int lim, bool wc = false) {
    c = (li >= sym) ? c + inc : c;
    int c15 = row_bcast<15>(c);
    int t = c + li + 1;
    int renorm = t - (t >> 2);
    if (wc) return blend_wrap_checked(c, li, lim, renorm, c15);
    return c15 >= lim ? renorm : c;
}

// Fill this stream's table with default rows (ffi/alloc_util.rs:77-79: allocations are default-initialised).
template <int CACHE>
__device__ __forceinline__ void init_table(const Table<CACHE>& t, uint32_t rows, int li) {
    // row = 16 x i16 = two 16-byte halves; even lanes write the first half, odd lanes the second
    u32x4 lo = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
    u32x4 hi = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};
    u32x4 v = (li & 1) ? hi : lo;
    const uint32_t base = t.lane_off - 2u * (uint32_t)li + 16u * (uint32_t)li;
    for (uint32_t i = 0; i < rows * 32u; i += 256u) {
        if (i + 16u * (uint32_t)li < rows * 32u) __builtin_amdgcn_raw_buffer_store_b128(v, t.rsrc, base + i, 0, 0);
    }
    t.reset_cache(li);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // the row loads that follow must see the fill
}

template <int CACHE>
__device__ __forceinline__ Table<CACHE> make_table(const LitBatch& b, uint8_t* lds, int li) {
    const uint32_t slab = b.geom.total_rows * 32u;           // bytes of one stream's table
    Table<CACHE> t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((uint8_t*)b.tables + (size_t)blockIdx.x * (LIT_THREADS / 16) * slab),
                                               0, (LIT_THREADS / 16) * slab, 0x00020000);
    t.lane_off = (threadIdx.x >> 4) * slab + 2u * (uint32_t)li;
    t.lds = lds;
    // per-stream LDS region: [high rows][high tags][low rows][low tags]; a unified cache aliases low onto high
    const uint32_t hi_bytes = b.cache_rows_high * 34u, lo_bytes = CACHE == 3 ? b.cache_rows_low * 34u : 0u;
    const uint32_t base = (threadIdx.x >> 4) * (hi_bytes + lo_bytes);
    t.ch.data_off = base + 2u * (uint32_t)li;
    t.ch.tag_off = base + b.cache_rows_high * 32u;
    t.ch.set_mask = (b.cache_rows_high >> 1) - 1u;
    t.cl.data_off = base + hi_bytes + 2u * (uint32_t)li;
    t.cl.tag_off = base + hi_bytes + b.cache_rows_low * 32u;
    t.cl.set_mask = (b.cache_rows_low >> 1) - 1u;
    return t;
}

// LitGeometry::wrap_check, after a stream: was a row whose total had left i16 coded with again?  Such a row took at least one more
// increment after the one that carried it past 0x7fff (blend_row above keeps it there), so its total is above 0x7fff + inc of its
// table -- stride rows literal_adaptation[0], FirstNibble rows [3], SecondNibble rows [2] (literal.rs:241,320,354).  Lane l of the
// row looks at entry 15 of rows l, l + 16, ...; needs the uncached table (the host launches these speeds with CACHE == 0).
template <int CACHE>
__device__ __forceinline__ bool wrap_scan(const LitGeometry& g, const Table<CACHE>& tb, int li, int rbase) {
    bool bad = false;
    const uint32_t base = tb.lane_off - 2u * (uint32_t)li + 30u;     // entry 15 of row 0 of this stream's slab
    for (uint32_t row = (uint32_t)li; row < g.total_rows; row += 16u) {
        const int total = (int)__builtin_amdgcn_raw_buffer_load_b16(tb.rsrc, base + (row << 5), 0, 0);
        const int inc = row < g.cm_base ? g.inc0 : (row < g.cm_base + g.nctx ? g.inc3 : g.inc2);
        bad |= total > 0x7fff + inc;
    }
    return ((__ballot(bad) >> rbase) & 0xffffull) != 0ull;
}

// ---------------------------------------------------------------------------------------------
// Encode, pass 1: adaptive model.  bytes -> (start | freq << 16) per nibble.
// ---------------------------------------------------------------------------------------------
// Mixing paths: (start | freq << 16) of `sym` under the mixed row p, plus the frequencies of `sym` under the context-map row
// and the stride row alone (wfreqs = cm freq | stride freq << 16, the Weights update's model_probs, literal.rs:236-239).
// That is six quotients -- entries sym and sym-1 of three rows, each by its own row total (probability/interface.rs:97-108)
// -- and they go through ONE division pass instead of three 16-entry ones: lanes 0/1 of the row take p, lanes 2/3 cm,
// lanes 4/5 the stride row; even lanes entry sym, odd lanes entry sym-1 (0 when sym == 0).
__device__ __forceinline__ uint32_t mixed_start_freq(int p, int cm, int st, int pmax, int cmax, int smax, int li, int rbase, int sym,
                                                     uint32_t& wfreqs) {
    const uint32_t pc = (uint32_t)p | ((uint32_t)cm << 16);
    const int src = (rbase + sym - (li & 1)) << 2;
    const uint32_t g_pc = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pc);
    const uint32_t g_st = (uint32_t)__builtin_amdgcn_ds_bpermute(src, st);
    const int which = li >> 1;
    int num = which == 0 ? (int)(g_pc & 0xffffu) : (which == 1 ? (int)(g_pc >> 16) : (int)g_st);
    num = ((li & 1) != 0 && sym == 0) ? 0 : num;
    const int den = which == 0 ? pmax : (which == 1 ? cmax : smax);
    const uint32_t q = scaled_div(num, den, biased_rcp15(den));
    const int f = (int)q - row_next_or_zero((int)q) - 1;          // even lanes: freq of their row's entry
    const uint32_t start = (uint32_t)row_bcast<1>((int)q) + 1u, freq = (uint32_t)row_bcast<0>(f);
    wfreqs = ((uint32_t)row_bcast<2>(f) & 0xffffu) | ((uint32_t)row_bcast<4>(f) << 16);
    return start | (freq << 16);
}

template <bool HIGH, int MM, bool MIX, int CACHE>
__device__ __forceinline__ uint32_t model_nibble(const LitGeometry& g, const LdsView& lv, const Table<CACHE>& tb, int li, int rbase,
                                                 uint32_t ctx, uint64_t last8, uint32_t hi_nib, int sym, int mix_rate, uint32_t& wfreqs) {
    const RowSel rs = select_rows<HIGH, MM>(g, lv.mix, ctx, last8, hi_nib);
    RowRef sref, cref;
    int st = tb.load(rs.stride_row, sref, HIGH);
    uint32_t packed;
    if (MIX) {
        int cm = tb.load(rs.cm_row, cref, HIGH);
        int cmax = row_bcast<15>(cm), smax = row_bcast<15>(st);
        int p = average_rows(cm, st, cmax, smax, mix_rate);
        int pmax = row_bcast<15>(p);
        packed = mixed_start_freq(p, cm, st, pmax, cmax, smax, li, rbase, sym, wfreqs);
        cm = blend_row_known_max(cm, li, sym, HIGH ? g.inc3 : g.inc2, HIGH ? g.lim3 : g.lim2, cmax, g.wrap_check != 0u);
        tb.store(cref, cm);
    } else {
        int cv = ((MM < 0 || MM == 2) && rs.is_default) ? 4 * (li + 1) : st;
        int mx = row_bcast<15>(cv);
        uint32_t d = scaled_div(cv, mx, biased_rcp15(mx));
        int dprev = row_prev_or_zero((int)d);
        uint32_t sf = (uint32_t)(dprev + 1) | ((uint32_t)((int)d - dprev - 1) << 16);
        packed = (uint32_t)row_gather((int)sf, rbase, sym);
    }
    if (!((MM < 0 || MM == 2) && rs.is_default)) st = blend_row(st, li, sym, g.inc0, g.lim0, g.wrap_check != 0u);   // literal_adaptation[0] for both nibbles, literal.rs:320,354
    if (CACHE != 0 || !((MM < 0 || MM == 2) && rs.is_default)) tb.store(sref, st);   // a cached way must hold its row even when it is not blended
    return packed;
}

struct FetchedRow { RowRef ref; int value; bool is_default; };

template <bool HIGH, int MM, int CACHE>
__device__ __forceinline__ FetchedRow fetch_row(const LitGeometry& g, const LdsView& lv, const Table<CACHE>& tb,
                                                uint32_t ctx, uint64_t last8, uint32_t hi_nib) {
    const RowSel rs = select_rows<HIGH, MM>(g, lv.mix, ctx, last8, hi_nib);
    FetchedRow f;
    f.value = tb.load(rs.stride_row, f.ref, HIGH);
    f.is_default = (MM < 0 || MM == 2) && rs.is_default;
    return f;
}

// (start | freq << 16) of `sym` under the fetched row, then blend + store: the non-mixing half of code_nibble
template <int CACHE>
__device__ __forceinline__ uint32_t model_finish(const LitGeometry& g, const Table<CACHE>& tb, int li, int rbase,
                                                 const FetchedRow& f, int sym) {
    const int cv = f.is_default ? 4 * (li + 1) : f.value;
    const int mx = row_bcast<15>(cv);
    const uint32_t d = scaled_div(cv, mx, biased_rcp15(mx));
    const int dprev = row_prev_or_zero((int)d);
    const uint32_t sf = (uint32_t)(dprev + 1) | ((uint32_t)((int)d - dprev - 1) << 16);
    const uint32_t packed = (uint32_t)row_gather((int)sf, rbase, sym);
    int st = f.value;
    if (!f.is_default) st = blend_row_known_max(st, li, sym, g.inc0, g.lim0, mx, g.wrap_check != 0u);   // cv == f.value here, so mx is its total
    if (CACHE != 0 || !f.is_default) tb.store(f.ref, st);
    return packed;
}

template <int MM, bool CTXC, bool MIX, int CACHE, bool SEG>
__global__ __launch_bounds__(LIT_THREADS) void lit_model_encode_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, li = lane & 15, rbase = lane & 48;
    const uint32_t gg = blockIdx.x * (LIT_THREADS / 16) + (threadIdx.x >> 4);
    const uint32_t G = gridDim.x * (LIT_THREADS / 16);
    const Table<CACHE> tb = make_table<CACHE>(b, lds, li);
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint8_t* in = b.in + (b.in_offsets ? b.in_offsets[s] : (uint64_t)s * b.stream_len);
        const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
        uint32_t* sf = b.sf + (size_t)s * 2u * b.max_stream_len;
        if (!b.resume) init_table(tb, g.total_rows, li);
        WeightsPair wp; wp.init();
        if (MIX && b.resume) { const int32_t* p = b.wstate + (li < 8 ? 0 : 3); wp.w.w0 = p[0]; wp.w.w1 = p[1]; wp.w.norm = p[2]; }
        int nh = wp.norm_high(), nl = wp.norm_low();     // normalized_weight of model_weights[1] (high nibble) / [0] (low nibble)
        uint64_t last8 = 0;
        uint32_t ctab = LIT_BLOB_CTXF;      // context table of the current literal block type
        SegCursor sc;
        if (SEG) sc.start(b, s, last8, ctab);
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];   // lut1 class of prev_prev
        // each lane holds one literal byte of the current and of the next 16-byte window (coalesced reads)
        uint32_t mine = ((uint32_t)li < len) ? in[li] : 0u;
        uint32_t nxt = (16u + li < len) ? in[16u + li] : 0u;
        // Non-mixing path: every symbol is known, so each row is requested one nibble ahead (right after the previous
        // row of the SAME table has been stored) and the model math of one nibble runs under the fetch of the next.
        uint32_t cur = (uint32_t)row_gather((int)mine, rbase, 0);
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, (uint32_t)(last8 >> 56), k1);
        FetchedRow rowH = {}, rowL = {};
        if (!MIX) {
            rowH = fetch_row<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);
            rowL = fetch_row<false, MM, CACHE>(g, lv, tb, ctx_cur, last8, cur >> 4);
        }
        for (uint32_t base = 0; base < len; base += 16) {
            const uint32_t cnt = len - base < 16u ? len - base : 16u;
            uint32_t pend_a = 0, pend_b = 0;  // lane k keeps the two pairs of byte base+k
            for (uint32_t k = 0; k < cnt; ++k) {
                if (MIX) {
                    const uint32_t byte = (uint32_t)row_gather((int)mine, rbase, (int)k);
                    const uint32_t prev = (uint32_t)(last8 >> 56);
                    const uint32_t ctx = context_of<CTXC>(g, lv.ctx, ctab, prev, k1);
                    if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + prev];
                    const uint32_t hi = byte >> 4, lo = byte & 15u;
                    uint32_t fh = 0, fl = 0;
                    const uint32_t ph = model_nibble<true, MM, MIX, CACHE>(g, lv, tb, li, rbase, ctx, last8, 0u, (int)hi, nh, fh);
                    const uint32_t pl = model_nibble<false, MM, MIX, CACHE>(g, lv, tb, li, rbase, ctx, last8, hi, (int)lo, nl, fl);
                    wp.update(li, fh, ph >> 16, fl, pl >> 16);
                    nh = wp.norm_high(); nl = wp.norm_low();
                    last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                    if (SEG) {   // the next Literal command starts from the ring buffer's last 8 bytes and its own block type
                        if (--sc.left == 0u) {
                            sc.advance(g, last8, ctab);
                            if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];
                        }
                    }
                    pend_a = (uint32_t)li == k ? ph : pend_a;
                    pend_b = (uint32_t)li == k ? pl : pend_b;
                } else {
                    const uint32_t byte = cur;
                    const uint32_t nb = (uint32_t)row_gather((int)(k + 1u < 16u ? mine : nxt), rbase, (int)((k + 1u) & 15u));
                    const uint32_t ph = model_finish<CACHE>(g, tb, li, rbase, rowH, (int)(byte >> 4));
                    last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                    if (SEG) { if (--sc.left == 0u) sc.advance(g, last8, ctab); }
                    if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];
                    ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, (uint32_t)(last8 >> 56), k1);
                    rowH = fetch_row<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);           // next byte's high row
                    const uint32_t pl = model_finish<CACHE>(g, tb, li, rbase, rowL, (int)(byte & 15u));
                    rowL = fetch_row<false, MM, CACHE>(g, lv, tb, ctx_cur, last8, nb >> 4);     // next byte's low row
                    cur = nb;
                    pend_a = (uint32_t)li == k ? ph : pend_a;
                    pend_b = (uint32_t)li == k ? pl : pend_b;
                }
            }
            // nibble index of byte (base+li) is 2*(base+li): each lane stores its two pairs (8 B; 128 B coalesced per row)
            if ((uint32_t)li < cnt) {
                u32x2 v = {pend_a, pend_b};   // written once, read once by the rANS kernel: keep it out of the L2's way
                __builtin_nontemporal_store(v, (u32x2*)(sf + 2u * (size_t)(base + li)));
            }
            mine = nxt;
            nxt = (base + 32u + li < len) ? in[base + 32u + li] : 0u;
        }
        if (MIX && b.wstate && (li & 7) == 0) {   // lanes 0 and 8 of the row hold the two Weights objects
            int32_t* p = b.wstate + (li ? 3 : 0);
            p[0] = wp.w.w0; p[1] = wp.w.w1; p[2] = wp.w.norm;
        }
        if (g.wrap_check) {     // a speed under which a row total can leave i16: say so if one did and was coded with again
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            if (wrap_scan<CACHE>(g, tb, li, rbase) && li == 0) {
                if (b.status) atomicOr(b.status, LIT_STATUS_BAD_MODEL);
                if (b.stream_bad) b.stream_bad[s] = 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Encode, pass 2: rANS.  One lane per stream, walks the (start,freq) pairs newest -> oldest and
// writes the coded bytes right-aligned into the stream's slot.  ans.rs:302-378.
// ---------------------------------------------------------------------------------------------
// The coded words leave through inline asm so that the compiler only sees in-order loads on vmcnt: with a visible
// store next to them it waits with vmcnt(0) for every prefetched group, i.e. for the newest request as well.
__device__ __forceinline__ void rans_store_word(uint32_t* p, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// (state / freq, state % freq) for state < freq << 48 (so state < 2^63, quotient < 2^48) in ONE double-precision step:
// the product of (double)state and a reciprocal refined to full precision and biased low by 2^-49 is below the true
// quotient x by less than x * 2^-48.4 + rounding < 1, so its integer part is q or q - 1 and a single compare of the
// remainder (which then fits 32 bits) finishes it.  Checked against 64-bit '/' and '%' by selftest_division_kernel.
#ifndef DIVANS_RANS_DIVMOD      // experiment switch: 0 = round 1-4's form (float -> integer through trunc / ldexp / floor / fma / two cvt)
#define DIVANS_RANS_DIVMOD 1
#endif
// 64-bit state / freq in one double-precision step.  The reciprocal is biased LOW by 2^-49 -- sixteen times what the Newton step and the
// roundings on the way can add back (two Newton steps refine v_rcp_f64 to full precision; the state's 63 bits round into 53 at 2^-53) -- so the product never reaches the true quotient, and since the quotient is below 2^48 it falls short of it by less
// than 2^48 * 2^-48 = 1: its integer part is the quotient or one below, which the remainder decides.  The integer part leaves the double
// through its mantissa (trunc, + 2^52: exact for an integer below 2^52) instead of the compiler's six-instruction f64 -> u64 conversion.
// selftest_division_kernel checks every divisor at its boundary states against 64-bit '/' and '%'.
// (Rounding the whole kernel's f64 arithmetic toward zero, so that one fma(state, y, 2^52) IS the floor, does not survive the compiler: its
// mode-register pass restores round-to-nearest in front of the first f64 FMA -- tried in round 5, the self-test caught it.)
__device__ __forceinline__ uint64_t rans_divmod(uint64_t state, uint32_t freq, uint32_t& rem) {
    const double fd = (double)freq;
    double y = __builtin_amdgcn_rcp(fd);
    y = __builtin_fma(__builtin_fma(-fd, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-fd, y, 1.0), y, y);     // (one step is not enough: 2218 of 8e8 self-test divisions came out wrong with it)
#if DIVANS_RANS_DIVMOD
    y *= (1.0 - 0x1p-49);
    const double sd = __builtin_fma((double)(uint32_t)(state >> 32), 0x1p32, (double)(uint32_t)state);
    const double t = __builtin_trunc(sd * y) + 0x1p52;
    uint64_t q = (uint64_t)__builtin_bit_cast(unsigned long long, t) & ((1ull << 52) - 1ull);
#else
    y *= (1.0 - 0x1p-49);
    const double qd = (double)state * y;
    uint64_t q = (uint64_t)qd;
#endif
    uint32_t r = (uint32_t)state - (uint32_t)q * freq;     // low 32 bits are enough: the true remainder is below 2 * freq
    if (r >= freq) { q += 1; r -= freq; }
    rem = r;
    return q;
}

__device__ __forceinline__ uint64_t rans_put(uint64_t state, uint32_t start, uint32_t freq, uint32_t*& wp) {
    // rescale_lim = ((2^31 >> 15) << 32) * freq = freq << 48
    if ((uint32_t)(state >> 32) >= (freq << 16)) { rans_store_word(--wp, (uint32_t)state); state >>= 32; }     // (freq << 48 has no low half)
    uint32_t r;
    const uint64_t q = rans_divmod(state, freq, r);        // ans.rs:318-323: ((state / freq) << 15) + state % freq + start
    return (q << 15) + (uint64_t)r + (uint64_t)start;
}

// one 65 536-symbol chunk [beg, end) of a stream's (start,freq) pairs, newest symbol first; words grow downwards from wp
__device__ __forceinline__ uint32_t* rans_encode_chunk(const uint32_t* sf, uint32_t beg, uint32_t end, uint32_t* wp, uint32_t& bad) {
    uint64_t a = 1ull << 31, bst = 1ull << 31;
    uint32_t i = end;
    // the two states alternate, so consecutive symbols are independent chains
    auto put = [&](uint32_t p) {
        const uint32_t start = p & 0xffffu;
        uint32_t freq = p >> 16;
        bad |= (freq == 0u) | (freq >> 15) | (start >> 15);
        freq = freq ? freq : 1u;
        const uint64_t x = rans_put(a, start, freq, wp);
        a = bst; bst = x;
    };
    while (i > beg && (i & 3u)) put(sf[--i]);             // ragged tail (nsym is even, so 0 or 2 symbols)
    // The pairs arrive in 16-byte groups through a ring of four register quads (4 x 4 groups = 64 symbols), each quad
    // requested three steps before it is coded.  Loads and their waits are inline asm: loads complete in order and
    // every step issues exactly four (from the chunk's first group once nothing is left to request), so "at most 12
    // operations outstanding" proves the quad requested three steps ago has arrived whatever the word stores in
    // between are doing -- and with a step this long those stores are old enough not to be waited for either.
    // The wait takes the registers as read-write operands, which keeps every use of the quad behind it.
    struct Quad { u32x4 a, b, c, d; };
    auto sat = [](uint32_t x, uint32_t k) { return x >= k ? x - k : 0u; };
    auto request1 = [&](u32x4& q, uint32_t top) {
        const uint32_t* p = sf + (top >= beg + 4u ? top - 4u : beg);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
    };
    auto request = [&](Quad& q, uint32_t top) {
        request1(q.a, top); request1(q.b, sat(top, 4u)); request1(q.c, sat(top, 8u)); request1(q.d, sat(top, 12u));
    };
    auto code = [&](const u32x4& g) { if (i > beg) { put(g.w); put(g.z); put(g.y); put(g.x); i -= 4u; } };
    Quad q0, q1, q2, q3;
    request(q0, i); request(q1, sat(i, 16u)); request(q2, sat(i, 32u)); request(q3, sat(i, 48u));
#define RANS_STEP(Q)   /* code the quad, then reuse its registers for the request 64 symbols on */ \
    {                                                                                   \
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(Q.a), "+v"(Q.b), "+v"(Q.c), "+v"(Q.d) : : "memory"); \
        const uint32_t top = i;                                                         \
        code(Q.a); code(Q.b); code(Q.c); code(Q.d);                                     \
        request(Q, sat(top, 64u));                                                      \
    }
    while (i > beg) { RANS_STEP(q0) RANS_STEP(q1) RANS_STEP(q2) RANS_STEP(q3) }
#undef RANS_STEP
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0.a), "+v"(q0.b), "+v"(q0.c), "+v"(q0.d), "+v"(q1.a), "+v"(q1.b), "+v"(q1.c), "+v"(q1.d) : : "memory");
    asm volatile("" : "+v"(q2.a), "+v"(q2.b), "+v"(q2.c), "+v"(q2.d), "+v"(q3.a), "+v"(q3.b), "+v"(q3.c), "+v"(q3.d) : : "memory");
    // unconditional swap (ans.rs:354-356), then [state_a][state_b] little-endian in front of the words
    const uint64_t fa = bst, fb = a;
    rans_store_word(--wp, (uint32_t)(fb >> 32)); rans_store_word(--wp, (uint32_t)fb);
    rans_store_word(--wp, (uint32_t)(fa >> 32)); rans_store_word(--wp, (uint32_t)fa);
    return wp;
}

__global__ __launch_bounds__(RANS_THREADS) void rans_encode_kernel(const RansBatch b) {
    const uint32_t s = blockIdx.x * RANS_THREADS + threadIdx.x;
    if (s >= b.n_streams) return;
    const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
    const uint32_t nsym = 2u * len;
    const uint32_t* sf = b.sf + (size_t)s * b.sf_stride;   // 16-byte aligned: sf_stride is a multiple of 4
    uint8_t* slot_end = b.out + (uint64_t)(s + 1) * b.out_slot;
    uint32_t* wp = (uint32_t*)slot_end;
    uint32_t* chunk_top = wp;
    uint32_t bad = 0;
    // chunk k covers symbols [k*65536, min((k+1)*65536, nsym)); later chunks sit later in the stream
    uint32_t nchunks = (nsym + 65535u) >> 16;
    for (uint32_t ck = nchunks; ck-- > 0;) {
        const uint32_t beg = ck << 16;
        const uint32_t end = beg + 65536u < nsym ? beg + 65536u : nsym;
        wp = rans_encode_chunk(sf, beg, end, wp, bad);
        if (b.chunk_bytes) {   // bytes of chunk ck (the host replays the reference's per-chunk Mux drains with these)
            b.chunk_bytes[(size_t)s * b.max_chunks + ck] = (uint32_t)((uint8_t*)chunk_top - (uint8_t*)wp);
            chunk_top = wp;
        }
    }
    uint64_t off = (uint64_t)((uint8_t*)wp - b.out) + b.out_base;
    b.out_offsets[s] = off;
    b.out_sizes[s] = (uint32_t)(slot_end - (uint8_t*)wp);
    if (bad) atomicOr(b.status, LIT_STATUS_BAD_MODEL);
}

// Streams of one or two chunks (at most 65 536 bytes): the chunks are independent rANS runs (states restart at 2^31,
// ans.rs:331-378), so each gets its own lane.  Lane pair (2k, 2k+1) = chunks (0, 1) of stream k; chunk 1 lands
// right-aligned in the stream's slot as before, chunk 0 in a scratch area, and rans_stitch_kernel puts it in front.
// At most two waves per SIMD (amdgpu_waves_per_eu): the pass is bound by its own instruction stream -- two waves on a SIMD take twice
// as long as one -- so all that matters is that the one-wave workgroups are spread evenly, and left to itself the dispatcher stacks
// three on some SIMDs of a CU while others hold one: 65 536 streams = 2048 waves = two per SIMD took 19.0 ms, pinned 14.7
// (profiles/r03g_rans_wave_placement.txt).  Larger batches simply run in rounds of 2048 waves.
// ---- the same chunk coded by TWO lanes, one per rANS state (ans.rs:302-329) ----------------------------------------------------------
// ANSEncoder's two states alternate symbol by symbol and never meet: state_a takes the symbols at even distances from the chunk's end,
// state_b the odd ones.  What they share is the output: every symbol step may push one 32-bit word, and the words lie in step order.
// Lanes 2k (state a) and 2k + 1 (state b) of a wave run the same instruction stream, so at every step each lane sees through one DPP
// swap whether its partner pushes a word too, and both keep the same running count: lane a's word goes first, lane b's behind it.  A lane
// walks half the chunk -- 32 768 dependent steps instead of 65 536 -- which is what matters when the batch is too small to give every SIMD
// a wave of whole chunks (a 16 384-stream batch = one of eight GPUs' share of BASELINE configs[4]: 512 such waves on 1024 SIMDs).
// When the one-lane kernel already fills the SIMDs the total work is the same and this form only adds the bookkeeping; launch_rans_encode picks.
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false); }

__device__ __forceinline__ uint32_t rans_encode_chunk_split(const uint32_t* sf, uint32_t beg, uint32_t end, uint32_t* top, uint32_t second, uint32_t& bad) {
    uint64_t st = 1ull << 31;       // this lane's state
    uint32_t cnt = 0;               // words pushed so far by the pair (the same value in both lanes)
    uint32_t i = end;
    // one step of the pair: lane `second` = 0 codes pa (the later symbol), lane 1 codes pb (the one before it)
    auto put2 = [&](uint32_t pa, uint32_t pb) {
        const uint32_t p = second ? pb : pa;
        const uint32_t start = p & 0xffffu;
        uint32_t freq = p >> 16;
        bad |= (freq == 0u) | (freq >> 15) | (start >> 15);
        freq = freq ? freq : 1u;
        // rescale_lim = freq << 48 has no low half: the test needs the state's high word only
        const uint32_t mine = (uint32_t)(st >> 32) >= (freq << 16) ? 1u : 0u;
        const uint32_t theirs = (uint32_t)pair_swap((int)mine);
        if (mine) { rans_store_word(top - 1u - cnt - (second ? theirs : 0u), (uint32_t)st); st >>= 32; }
        cnt += mine + theirs;
        uint32_t r;
        const uint64_t q = rans_divmod(st, freq, r);
        st = (q << 15) + (uint64_t)r + (uint64_t)start;
    };
    if (i > beg && (i & 3u)) { put2(sf[i - 1u], sf[i - 2u]); i -= 2u; }      // ragged tail (nsym is even: 0 or 2 symbols)
    struct Quad { u32x4 a, b, c, d; };
    auto sat = [](uint32_t x, uint32_t k) { return x >= k ? x - k : 0u; };
    auto request1 = [&](u32x4& q, uint32_t t) {
        const uint32_t* p = sf + (t >= beg + 4u ? t - 4u : beg);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q) : "v"(p) : "memory");
    };
    auto request = [&](Quad& q, uint32_t t) { request1(q.a, t); request1(q.b, sat(t, 4u)); request1(q.c, sat(t, 8u)); request1(q.d, sat(t, 12u)); };
    auto code = [&](const u32x4& g) { if (i > beg) { put2(g.w, g.z); put2(g.y, g.x); i -= 4u; } };
    Quad q0, q1, q2, q3;
    request(q0, i); request(q1, sat(i, 16u)); request(q2, sat(i, 32u)); request(q3, sat(i, 48u));
#define RANS_STEP(Q)                                                                    \
    {                                                                                   \
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(Q.a), "+v"(Q.b), "+v"(Q.c), "+v"(Q.d) : : "memory"); \
        const uint32_t t0 = i;                                                          \
        code(Q.a); code(Q.b); code(Q.c); code(Q.d);                                     \
        request(Q, sat(t0, 64u));                                                       \
    }
    while (i > beg) { RANS_STEP(q0) RANS_STEP(q1) RANS_STEP(q2) RANS_STEP(q3) }
#undef RANS_STEP
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0.a), "+v"(q0.b), "+v"(q0.c), "+v"(q0.d), "+v"(q1.a), "+v"(q1.b), "+v"(q1.c), "+v"(q1.d) : : "memory");
    asm volatile("" : "+v"(q2.a), "+v"(q2.b), "+v"(q2.c), "+v"(q2.d), "+v"(q3.a), "+v"(q3.b), "+v"(q3.c), "+v"(q3.d) : : "memory");
    // ans.rs:354-356: after the (even number of) symbols the reference's `a` is the state that coded the chunk's LAST symbol -- this pair's
    // lane 0 -- and the unconditional swap puts the other one first: [state of lane 1][state of lane 0] in front of the words
    uint32_t* wp = top - cnt - (second ? 4u : 2u);
    rans_store_word(wp, (uint32_t)st); rans_store_word(wp + 1, (uint32_t)(st >> 32));
    return 4u * (cnt + 4u);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rans_encode2_split_kernel(const RansBatch b) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t s = g >> 2, ck = (g >> 1) & 1u, second = g & 1u;
    const bool live = s < b.n_streams;
    const uint32_t len = live ? (b.in_sizes ? b.in_sizes[s] : b.stream_len) : 0u;
    const uint32_t nsym = 2u * len, beg = ck << 16;
    uint8_t* slot_end = b.out + (uint64_t)(s + 1) * b.out_slot;
    uint32_t size = 0, bad = 0;
    if (live && beg < nsym) {
        const uint32_t end = beg + 65536u < nsym ? beg + 65536u : nsym;
        const uint32_t* sf = b.sf + (size_t)s * b.sf_stride;
        uint8_t* top = ck ? slot_end : b.scratch + (uint64_t)(s + 1) * b.scratch_stride;
        size = rans_encode_chunk_split(sf, beg, end, (uint32_t*)top, second, bad);
        if (b.chunk_bytes && !second) b.chunk_bytes[(size_t)s * b.max_chunks + ck] = size;
    }
    const uint32_t other = (uint32_t)__shfl_xor((int)size, 2);
    if (live && (g & 3u) == 0u) {
        const uint32_t total = size + other;
        b.out_sizes[s] = total;
        b.out_offsets[s] = (uint64_t)(slot_end - b.out) - total + b.out_base;
        b.chunk0_sizes[s] = size;
    }
    if (bad) atomicOr(b.status, LIT_STATUS_BAD_MODEL);
}

#ifndef DIVANS_RANS2_THREADS    // experiment switch (scripts/build_variants.sh rans2)
#define DIVANS_RANS2_THREADS 256
#endif
constexpr int RANS2_THREADS = DIVANS_RANS2_THREADS;
__global__ __launch_bounds__(RANS2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void rans_encode2_kernel(const RansBatch b) {
    const uint32_t g = blockIdx.x * RANS2_THREADS + threadIdx.x;
    const uint32_t s = g >> 1, ck = g & 1u;
    const bool live = s < b.n_streams;
    const uint32_t len = live ? (b.in_sizes ? b.in_sizes[s] : b.stream_len) : 0u;
    const uint32_t nsym = 2u * len, beg = ck << 16;
    uint8_t* slot_end = b.out + (uint64_t)(s + 1) * b.out_slot;
    uint32_t size = 0, bad = 0;
    if (live && beg < nsym) {
        const uint32_t end = beg + 65536u < nsym ? beg + 65536u : nsym;
        const uint32_t* sf = b.sf + (size_t)s * b.sf_stride;
        uint8_t* top = ck ? slot_end : b.scratch + (uint64_t)(s + 1) * b.scratch_stride;
        uint32_t* wp = rans_encode_chunk(sf, beg, end, (uint32_t*)top, bad);
        size = (uint32_t)(top - (uint8_t*)wp);
        if (b.chunk_bytes) b.chunk_bytes[(size_t)s * b.max_chunks + ck] = size;
    }
    const uint32_t other = (uint32_t)__shfl_xor((int)size, 1);
    if (live && ck == 0u) {
        const uint32_t total = size + other;
        b.out_sizes[s] = total;
        b.out_offsets[s] = (uint64_t)(slot_end - b.out) - total + b.out_base;
        b.chunk0_sizes[s] = size;
    }
    if (bad) atomicOr(b.status, LIT_STATUS_BAD_MODEL);
}

__global__ __launch_bounds__(256) void rans_stitch_kernel(const RansBatch b) {
    // one wave per stream: chunk 0 from the scratch area to just below chunk 1
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t nw = (gridDim.x * 256u) >> 6;
    for (uint32_t s = wave; s < b.n_streams; s += nw) {
        const uint32_t words = b.chunk0_sizes[s] >> 2;
        const uint32_t* src = (const uint32_t*)(b.scratch + (uint64_t)(s + 1) * b.scratch_stride) - words;
        uint32_t* dst = (uint32_t*)(b.out + (b.out_offsets[s] - b.out_base));
        for (uint32_t i = lane; i < words; i += 64u) dst[i] = __builtin_nontemporal_load(src + i);
    }
}

// ---------------------------------------------------------------------------------------------
// Decode: fused rANS decode + CDF search + blend.
// ---------------------------------------------------------------------------------------------
struct WordWindow {
    // 16 upcoming 32-bit words of the coded stream, one per lane of the row (`w`), plus the 16 after them (`wn`),
    // requested one window early so that handing a word to the decoder (ds_bpermute on `w`) never waits on memory.
    const uint32_t* in; uint32_t nwords; uint32_t base, pos; uint32_t w, wn;
    __device__ __forceinline__ uint32_t fetch(uint32_t first, int li) const { return (first + li < nwords) ? in[first + li] : 0u; }
    __device__ __forceinline__ void start(int li) { base = 0; pos = 0; w = fetch(0, li); wn = fetch(16, li); }
    __device__ __forceinline__ uint32_t next(int li, int rbase) {
        uint32_t v = (uint32_t)row_gather((int)w, rbase, (int)(pos - base));
        pos += 1;
        if (pos - base == 16u) { base = pos; w = wn; wn = fetch(base + 16u, li); }
        return v;
    }
};

template <bool HIGH, int MM, bool MIX, int CACHE>
__device__ __forceinline__ uint32_t decode_nibble(const LitGeometry& g, const LdsView& lv, const Table<CACHE>& tb, int li, int rbase,
                                                  uint32_t ctx, uint64_t last8, uint32_t hi_nib, uint64_t& S, int mix_rate,
                                                  uint32_t& wfreqs, uint32_t& wpmix) {
    const RowSel rs = select_rows<HIGH, MM>(g, lv.mix, ctx, last8, hi_nib);
    RowRef sref, cref;
    int st = tb.load(rs.stride_row, sref, HIGH);
    int cm = 0, cmax = 0, smax = 0;
    int cv;
    if (MIX) {
        cm = tb.load(rs.cm_row, cref, HIGH);
        cmax = row_bcast<15>(cm); smax = row_bcast<15>(st);
        cv = average_rows(cm, st, cmax, smax, mix_rate);
    } else {
        cv = ((MM < 0 || MM == 2) && rs.is_default) ? 4 * (li + 1) : st;
    }
    // cdf_offset_to_sym_start_and_freq: first i<15 with rescaled < cdf[i]
    const uint32_t slot = (uint32_t)S & 0x7fffu;
    const int mx = row_bcast<15>(cv);
    const int rescaled = (int)((uint32_t)__umul24(slot, (uint32_t)mx) >> 15);
    const unsigned long long ge = __ballot(rescaled >= cv);
    const int sym = __popc((uint32_t)(ge >> rbase) & 0x7fffu);
    uint32_t packed;
    if (MIX) packed = mixed_start_freq(cv, cm, st, mx, cmax, smax, li, rbase, sym, wfreqs);
    else {
        const uint32_t d = scaled_div(cv, mx, biased_rcp15(mx));
        const int dprev = row_prev_or_zero((int)d);
        const uint32_t sf = (uint32_t)(dprev + 1) | ((uint32_t)((int)d - dprev - 1) << 16);
        packed = (uint32_t)row_gather((int)sf, rbase, sym);
    }
    const uint32_t start = packed & 0xffffu, freq = packed >> 16;
    // helper_advance_sym ans.rs:238: x = freq * (state >> 15) + (state & mask) - start
    S = (uint64_t)freq * (S >> 15) + (uint64_t)slot - (uint64_t)start;
    if (MIX) {
        wpmix = freq;
        cm = blend_row_known_max(cm, li, sym, HIGH ? g.inc3 : g.inc2, HIGH ? g.lim3 : g.lim2, cmax, g.wrap_check != 0u);
        tb.store(cref, cm);
    }
    if (!((MM < 0 || MM == 2) && rs.is_default)) st = blend_row(st, li, sym, g.inc0, g.lim0, g.wrap_check != 0u);
    if (CACHE != 0 || !((MM < 0 || MM == 2) && rs.is_default)) tb.store(sref, st);
    return (uint32_t)sym;
}

// Non-mixing decode, software-pipelined: the symbol search needs only the row and the state's low 15 bits, and the
// NEXT row's address needs only the symbol -- so the next row is requested right after the search, and the
// division / (start,freq) gather / state update / blend / store of the current nibble run under that fetch.
// Two rows are live at a time (the one being finished and the one in flight); they belong to different tables
// (high vs low nibble), and the 2-way cache never evicts the most recently used way, so they cannot collide.
__device__ __forceinline__ int search_symbol(int cv, uint32_t slot, int rbase) {
    const int mx = row_bcast<15>(cv);
    const int rescaled = (int)((uint32_t)__umul24(slot, (uint32_t)mx) >> 15);
    const unsigned long long ge = __ballot(rescaled >= cv);
    return __popc((uint32_t)(ge >> rbase) & 0x7fffu);
}

template <int CACHE>
__device__ __forceinline__ void finish_nibble(const LitGeometry& g, const Table<CACHE>& tb, int li, int rbase,
                                              const FetchedRow& f, int cv, int sym, uint64_t& S) {
    const uint32_t slot = (uint32_t)S & 0x7fffu;
    const int mx = row_bcast<15>(cv);
    const uint32_t d = scaled_div(cv, mx, biased_rcp15(mx));
    const int dprev = row_prev_or_zero((int)d);
    const uint32_t sf = (uint32_t)(dprev + 1) | ((uint32_t)((int)d - dprev - 1) << 16);
    const uint32_t packed = (uint32_t)row_gather((int)sf, rbase, sym);
    const uint32_t start = packed & 0xffffu, freq = packed >> 16;
    S = (uint64_t)freq * (S >> 15) + (uint64_t)slot - (uint64_t)start;     // helper_advance_sym, ans.rs:238
    int st = f.value;
    if (!f.is_default) st = blend_row_known_max(st, li, sym, g.inc0, g.lim0, mx, g.wrap_check != 0u);   // cv == f.value here, so mx is its total
    if (CACHE != 0 || !f.is_default) tb.store(f.ref, st);
}

template <int MM, bool CTXC, bool MIX, int CACHE, bool SEG>
__global__ __launch_bounds__(LIT_THREADS) void lit_decode_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, li = lane & 15, rbase = lane & 48;
    const uint32_t gg = blockIdx.x * (LIT_THREADS / 16) + (threadIdx.x >> 4);
    const uint32_t G = gridDim.x * (LIT_THREADS / 16);
    const Table<CACHE> tb = make_table<CACHE>(b, lds, li);
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint32_t len = b.out_sizes ? b.out_sizes[s] : b.stream_len;
        uint8_t* out = b.out + (b.out_offsets ? b.out_offsets[s] : (uint64_t)s * b.stream_len);
        WordWindow ww;
        ww.in = (const uint32_t*)(b.in + b.in_offsets[s]);
        ww.nwords = b.in_sizes[s] >> 2;
        ww.start(li);
        if (!b.resume) init_table(tb, g.total_rows, li);
        WeightsPair wp; wp.init();
        if (MIX && b.resume) { const int32_t* p = b.wstate + (li < 8 ? 0 : 3); wp.w.w0 = p[0]; wp.w.w1 = p[1]; wp.w.norm = p[2]; }
        int nh = wp.norm_high(), nl = wp.norm_low();     // normalized_weight of model_weights[1] (high nibble) / [0] (low nibble)
        uint64_t last8 = 0;
        uint32_t ctab = LIT_BLOB_CTXF;      // context table of the current literal block type
        SegCursor sc;
        if (SEG) sc.start(b, s, last8, ctab);
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];
        uint64_t SA = 0, SB = 0;      // state_a decodes high nibbles, state_b low nibbles (two symbols per byte)
        bool corrupt = false;
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, (uint32_t)(last8 >> 56), k1);
        FetchedRow rowH = {};
        if (!MIX) rowH = fetch_row<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);
        for (uint32_t cbeg = 0; cbeg < len; cbeg += 32768u) {
            // start of a 65 536-symbol chunk: 16 bytes = state_a, state_b (ans.rs:174-186)
            {
                uint32_t a0 = ww.next(li, rbase), a1 = ww.next(li, rbase), b0 = ww.next(li, rbase), b1 = ww.next(li, rbase);
                SA = ((uint64_t)a1 << 32) | a0;
                SB = ((uint64_t)b1 << 32) | b0;
            }
            const uint32_t cend = cbeg + 32768u < len ? cbeg + 32768u : len;
            for (uint32_t base = cbeg; base < cend; base += 16u) {
                const uint32_t cnt = cend - base < 16u ? cend - base : 16u;
                uint32_t outb = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    if (MIX) {
                        const uint32_t prev = (uint32_t)(last8 >> 56);
                        const uint32_t ctx = context_of<CTXC>(g, lv.ctx, ctab, prev, k1);
                        if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + prev];
                        // a state that dropped below 2^31 takes 4 more bytes right before it is used again (ans.rs:432-440)
                        if (SA < (1ull << 31)) SA = (SA << 32) | ww.next(li, rbase);
                        uint32_t fh = 0, fl = 0, ph = 0, pl = 0;
                        const uint32_t hi = decode_nibble<true, MM, MIX, CACHE>(g, lv, tb, li, rbase, ctx, last8, 0u, SA, nh, fh, ph);
                        if (SB < (1ull << 31)) SB = (SB << 32) | ww.next(li, rbase);
                        const uint32_t lo = decode_nibble<false, MM, MIX, CACHE>(g, lv, tb, li, rbase, ctx, last8, hi, SB, nl, fl, pl);
                        wp.update(li, fh, ph, fl, pl);
                        nh = wp.norm_high(); nl = wp.norm_low();
                        const uint32_t byte = (hi << 4) | lo;
                        last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                        if (SEG) {
                            if (--sc.left == 0u) {
                                sc.advance(g, last8, ctab);
                                if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];
                            }
                        }
                        outb = (uint32_t)li == k ? byte : outb;
                    } else {
                        // rowH (this byte's high-nibble row) was requested while the previous byte was being finished
                        if (SA < (1ull << 31)) SA = (SA << 32) | ww.next(li, rbase);
                        const int cvh = rowH.is_default ? 4 * (li + 1) : rowH.value;
                        const uint32_t hi = (uint32_t)search_symbol(cvh, (uint32_t)SA & 0x7fffu, rbase);
                        const FetchedRow rowL = fetch_row<false, MM, CACHE>(g, lv, tb, ctx_cur, last8, hi);
                        finish_nibble<CACHE>(g, tb, li, rbase, rowH, cvh, (int)hi, SA);
                        if (SB < (1ull << 31)) SB = (SB << 32) | ww.next(li, rbase);
                        const int cvl = rowL.is_default ? 4 * (li + 1) : rowL.value;
                        const uint32_t lo = (uint32_t)search_symbol(cvl, (uint32_t)SB & 0x7fffu, rbase);
                        const uint32_t byte = (hi << 4) | lo;
                        last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                        if (SEG) { if (--sc.left == 0u) sc.advance(g, last8, ctab); }
                        if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + (uint32_t)((last8 >> 48) & 0xffu)];
                        ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, (uint32_t)(last8 >> 56), k1);
                        rowH = fetch_row<true, MM, CACHE>(g, lv, tb, ctx_cur, last8, 0u);   // next byte's row (harmless past the end)
                        finish_nibble<CACHE>(g, tb, li, rbase, rowL, cvl, (int)lo, SB);
                        outb = (uint32_t)li == k ? byte : outb;
                    }
                }
                if ((uint32_t)li < cnt) __builtin_nontemporal_store((uint8_t)outb, out + base + li);
            }
            // rANS is an exact inverse: a chunk that was coded from the start states 2^31 (ans.rs:135-136,331-378) decodes back
            // to exactly those; anything else means a truncated, corrupt or mismatched stream (the reference would stall on
            // NeedsMoreInput or fail its checksum)
            corrupt |= (SA != (1ull << 31)) | (SB != (1ull << 31));
        }
        if (b.consumed) { corrupt |= ww.pos > ww.nwords; if (li == 0) b.consumed[s] = ww.pos; }
        else corrupt |= ww.pos != ww.nwords;     // every coded word consumed, none read past the end
        if (MIX && b.wstate && (li & 7) == 0) {   // lanes 0 and 8 of the row hold the two Weights objects
            int32_t* p = b.wstate + (li ? 3 : 0);
            p[0] = wp.w.w0; p[1] = wp.w.w1; p[2] = wp.w.norm;
        }
        if (g.wrap_check) {     // the encoder of this stream coded with a row whose i16 total had wrapped: whatever it wrote, it is not these bytes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            corrupt |= wrap_scan<CACHE>(g, tb, li, rbase);
        }
        if (corrupt && li == 0) {
            if (b.status) atomicOr(b.status, LIT_STATUS_BAD_STREAM);
            if (b.stream_bad) b.stream_bad[s] = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pack_streams: exclusive scan of the 4-byte-rounded sizes (single block, 3 phases) + coalesced copy
// ---------------------------------------------------------------------------------------------
// `accumulate`: *total on entry is where this batch starts in the packed buffer (a sub-batch appended behind earlier ones) and takes
// this batch's bytes on top; otherwise the batch starts at 0 and *total receives its size
__global__ __launch_bounds__(1024) void scan_sizes_kernel(const uint32_t* sizes, uint32_t n, uint64_t* offsets, uint64_t* total, int accumulate) {
    __shared__ uint64_t partial[1024];
    const uint32_t t = threadIdx.x;
    const uint64_t base = accumulate ? *total : 0ull;       // read by every thread before thread 1023 overwrites it (barriers below)
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t beg = t * per < n ? t * per : n, end = beg + per < n ? beg + per : n;
    uint64_t sum = 0;
    for (uint32_t i = beg; i < end; ++i) sum += (sizes[i] + 3u) & ~3u;
    partial[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        uint64_t v = t >= off ? partial[t - off] : 0;
        __syncthreads();
        partial[t] += v;
        __syncthreads();
    }
    uint64_t run = base + (t ? partial[t - 1] : 0);
    for (uint32_t i = beg; i < end; ++i) { offsets[i] = run; run += (sizes[i] + 3u) & ~3u; }
    if (t == 1023u) *total = base + partial[1023];
}

__global__ __launch_bounds__(256) void pack_copy_kernel(const uint8_t* slots, const uint64_t* src_off, const uint32_t* sizes,
                                                        uint32_t n, uint8_t* packed, const uint64_t* dst_off, uint64_t cap, uint32_t* status) {
    // one wave per stream, 4 bytes per lane per step (coded streams are whole 32-bit words)
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t nw = (gridDim.x * 256u) >> 6;
    for (uint32_t s = wave; s < n; s += nw) {
        const uint32_t* src = (const uint32_t*)(slots + src_off[s]);
        uint32_t* dst = (uint32_t*)(packed + dst_off[s]);
        const uint32_t words = (sizes[s] + 3u) >> 2;
        if (dst_off[s] + 4ull * words > cap) {        // the caller's buffer ends here: the stream stays unwritten (offsets and sizes still say what was needed)
            if (lane == 0 && status) atomicOr(status, LIT_STATUS_OUTPUT_FULL);
            continue;
        }
        for (uint32_t i = lane; i < words; i += 64u) dst[i] = src[i];
    }
}

// exhaustive check of exact_div against '/' (tests): grid-stride over max in [1, 32767]
__global__ void selftest_division_kernel(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (uint32_t mx = blockIdx.x + 1; mx < 32768u; mx += gridDim.x) {
        float rcp = __builtin_amdgcn_rcpf((float)mx);
        for (uint32_t c = threadIdx.x; c <= mx; c += blockDim.x) {
            uint32_t n = c << 15;
            bad += exact_div(n, mx, rcp) != n / mx;
            bad += scaled_div((int)c, (int)mx, biased_rcp15((int)mx)) != n / mx;
        }
        // the rANS step divides a state below mx << 48 by mx: boundary states, multiples of mx +- 1, and a spread of others
        for (uint32_t k = threadIdx.x; k < 4096u; k += blockDim.x) {
            const uint64_t top = (uint64_t)mx << 48;
            uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull + (uint64_t)mx * 0xD1B54A32D192ED03ull;
            x ^= x >> 29;
            const uint64_t cand[6] = {top - 1ull - k, (x % top), (x % top) / mx * mx, ((x % top) / mx * mx) + mx - 1ull,
                                      (1ull << 31) + k * 977ull, (x >> (k & 31)) % top};
            for (int j = 0; j < 6; ++j) {
                uint32_t rem;
                const uint64_t q = rans_divmod(cand[j], mx, rem);
                bad += (q != cand[j] / mx) | (rem != (uint32_t)(cand[j] % mx));
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// The CDF / Weights primitives the coding kernels are built from, driven by a script of operations on two rows held
// one entry per lane -- the GPU side of the reference's own CDF unit tests (probability/common_tests.rs:152-185
// operation_test_helper compares two CDF implementations after every blend and at five mixing rates; here the second
// implementation is the CPU restatement, compared on the host).  Every op writes one 16-entry record.
__global__ __launch_bounds__(64) void cdf_ops_selftest_kernel(const u32x4* ops, uint32_t n, int32_t* out) {
    const int lane = threadIdx.x & 63, li = lane & 15, rbase = lane & 48;
    int c0 = 4 * (li + 1), c1 = 4 * (li + 1);
    Weights w = {1, 1, 1 << 14};
    for (uint32_t k = 0; k < n; ++k) {
        const u32x4 op = ops[k];
        int rec = 0;
        switch (op.x) {
        case 0: c0 = blend_row(c0, li, (int)op.y, (int)op.z, (int)op.w); rec = c0; break;                  // cdf0.blend(sym, Speed(inc, lim))
        case 1: c1 = blend_row(c1, li, (int)op.y, (int)op.z, (int)op.w); rec = c1; break;                  // cdf1.blend
        case 2: rec = average_rows(c0, c1, row_bcast<15>(c0), row_bcast<15>(c1), (int)op.y); break;       // cdf0.average(cdf1, mix_rate)
        case 3: case 4: {                                                                                   // sym_to_start_and_freq / cdf_offset_to_sym_start_and_freq on cdf0
            const int mx = row_bcast<15>(c0);
            const int sym = op.x == 3 ? (int)op.y : search_symbol(c0, op.y & 0x7fffu, rbase);
            const uint32_t d = scaled_div(c0, mx, biased_rcp15(mx));
            const int dprev = row_prev_or_zero((int)d);
            const uint32_t sf = (uint32_t)(dprev + 1) | ((uint32_t)((int)d - dprev - 1) << 16);
            const uint32_t packed = (uint32_t)row_gather((int)sf, rbase, sym);
            rec = li == 0 ? (int)(packed & 0xffffu) : (li == 1 ? (int)(packed >> 16) : (li == 2 ? sym : 0));
            break;
        }
        case 5: weights_update(w, (int)(short)op.y, (int)(short)op.z, (int)(short)op.w);                  // Weights::update([p0, p1], weighted)
                rec = li == 0 ? w.w0 : (li == 1 ? w.w1 : (li == 2 ? w.norm : 0)); break;
        case 6: c0 = 4 * (li + 1); c1 = 4 * (li + 1); w.w0 = 1; w.w1 = 1; w.norm = 1 << 14; rec = c0; break;
        case 7: c0 = blend_row_known_max(c0, li, (int)op.y, (int)op.z, (int)op.w, row_bcast<15>(c0)); rec = c0; break;   // the variant the pipelined paths use
        default: break;
        }
        if (lane < 16) out[(size_t)k * 16u + (uint32_t)li] = rec;
    }
}

// ---------------------------------------------------------------------------------------------
// launch helpers (called from capi.cpp)
// ---------------------------------------------------------------------------------------------
typedef void (*LitKernel)(const LitBatch);

#define LIT_PICK(KERNEL)                                                                                     \
    template <int CACHE, bool SEG>                                                                           \
    static LitKernel pick_##KERNEL(int mm, bool ctxc, bool mix) {                                            \
        const int key = (mm == 4 ? 2 : (mm == 0 ? 1 : 0)) * 4 + (ctxc ? 2 : 0) + (mix ? 1 : 0);              \
        switch (key) {                                                                                       \
        case 0: return KERNEL<-1, false, false, CACHE, SEG>; case 1: return KERNEL<-1, false, true, CACHE, SEG>;   \
        case 2: return KERNEL<-1, true, false, CACHE, SEG>;  case 3: return KERNEL<-1, true, true, CACHE, SEG>;    \
        case 4: return KERNEL<0, false, false, CACHE, SEG>;  case 5: return KERNEL<0, false, true, CACHE, SEG>;    \
        case 6: return KERNEL<0, true, false, CACHE, SEG>;   case 7: return KERNEL<0, true, true, CACHE, SEG>;     \
        case 8: return KERNEL<4, false, false, CACHE, SEG>;  case 9: return KERNEL<4, false, true, CACHE, SEG>;    \
        case 10: return KERNEL<4, true, false, CACHE, SEG>;  default: return KERNEL<4, true, true, CACHE, SEG>;    \
        }                                                                                                    \
    }                                                                                                        \
    static LitKernel pick_mode_##KERNEL(int cache_mode, bool seg, int mm, bool ctxc, bool mix) {             \
        if (seg) return cache_mode == 2 ? pick_##KERNEL<2, true>(mm, ctxc, mix) : pick_##KERNEL<0, true>(mm, ctxc, mix);   /* segment lists: default cache organisation or none */ \
        switch (cache_mode) {                                                                                \
        case 1: return pick_##KERNEL<1, false>(mm, ctxc, mix);                                               \
        case 2: return pick_##KERNEL<2, false>(mm, ctxc, mix);                                               \
        case 3: return pick_##KERNEL<3, false>(mm, ctxc, mix);                                               \
        default: return pick_##KERNEL<0, false>(mm, ctxc, mix);                                              \
        }                                                                                                    \
    }
LIT_PICK(lit_model_encode_kernel)
#if DIVANS_WITH_EXPERIMENTAL_DECODERS
LIT_PICK(lit_decode_kernel)
#else
// The default library keeps generation 1's decoder only where no later generation runs: without a cache (wrap-checked speeds) and with
// the high-nibble-row cache (the resumable call-by-call decoder); the unified / split cache instances are experiment builds (lit_kernels.h)
template <int CACHE, bool SEG>
static LitKernel pick_lit_decode_kernel(int mm, bool ctxc, bool mix) {
    const int key = (mm == 4 ? 2 : (mm == 0 ? 1 : 0)) * 4 + (ctxc ? 2 : 0) + (mix ? 1 : 0);
    switch (key) {
    case 0: return lit_decode_kernel<-1, false, false, CACHE, SEG>; case 1: return lit_decode_kernel<-1, false, true, CACHE, SEG>;
    case 2: return lit_decode_kernel<-1, true, false, CACHE, SEG>;  case 3: return lit_decode_kernel<-1, true, true, CACHE, SEG>;
    case 4: return lit_decode_kernel<0, false, false, CACHE, SEG>;  case 5: return lit_decode_kernel<0, false, true, CACHE, SEG>;
    case 6: return lit_decode_kernel<0, true, false, CACHE, SEG>;   case 7: return lit_decode_kernel<0, true, true, CACHE, SEG>;
    case 8: return lit_decode_kernel<4, false, false, CACHE, SEG>;  case 9: return lit_decode_kernel<4, false, true, CACHE, SEG>;
    case 10: return lit_decode_kernel<4, true, false, CACHE, SEG>;  default: return lit_decode_kernel<4, true, true, CACHE, SEG>;
    }
}
static LitKernel pick_mode_lit_decode_kernel(int cache_mode, bool seg, int mm, bool ctxc, bool mix) {
    if (seg) return cache_mode == 2 ? pick_lit_decode_kernel<2, true>(mm, ctxc, mix) : pick_lit_decode_kernel<0, true>(mm, ctxc, mix);
    if (cache_mode == 2) return pick_lit_decode_kernel<2, false>(mm, ctxc, mix);
    return cache_mode == 0 ? pick_lit_decode_kernel<0, false>(mm, ctxc, mix) : nullptr;
}
#endif

// specialisation that will actually run: only 0 and 4 have dedicated MM instances
static int effective_mm(int mm) { return (mm == 0 || mm == 4) ? mm : -1; }

uint32_t lit_lds_bytes(const LitBatch& b) {
    uint32_t bytes = b.cache_bytes_per_wg;
    if (b.geom.ctx_const < 0) bytes += LIT_BLOB_CTXF + LIT_CTXF_BYTES * b.geom.n_btypes;
    if (effective_mm(b.geom.mm_uniform) < 0) bytes += 8192u;
    return bytes;
}

hipError_t launch_model_encode(const LitBatch& b_in, bool mix, uint32_t blocks, hipStream_t st) {
    const LitBatch& b = b_in;
    const int mm = effective_mm(b.geom.mm_uniform);
    if (b.segs && b.cache_mode != 2u && b.cache_mode != 0u) return hipErrorInvalidValue;
    LitKernel k = pick_mode_lit_model_encode_kernel((int)b.cache_mode, b.segs != nullptr, mm, b.geom.ctx_const >= 0, mix);
    const uint32_t lds = lit_lds_bytes(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LIT_THREADS), lds, st, b);
    return hipGetLastError();
}
hipError_t launch_rans_encode(const RansBatch& b, hipStream_t st) {
    if (b.scratch) {   // one lane per chunk (streams of at most two chunks), then move chunk 0 in front of chunk 1
        // one lane per chunk while that gives every SIMD a wave (2 x n_streams lanes >= 1024 waves on the 256-CU part); two lanes per chunk below that
        if (b.split_states) {
            hipLaunchKernelGGL(rans_encode2_split_kernel, dim3((4u * b.n_streams + 255u) / 256u), dim3(256), 0, st, b);
        } else {
            const uint32_t blocks = (2u * b.n_streams + RANS2_THREADS - 1) / RANS2_THREADS;
            hipLaunchKernelGGL(rans_encode2_kernel, dim3(blocks), dim3(RANS2_THREADS), 0, st, b);
        }
        const uint32_t sblocks = (b.n_streams + 3u) / 4u < 8192u ? (b.n_streams + 3u) / 4u : 8192u;
        hipLaunchKernelGGL(rans_stitch_kernel, dim3(sblocks), dim3(256), 0, st, b);
        return hipGetLastError();
    }
    uint32_t blocks = (b.n_streams + RANS_THREADS - 1) / RANS_THREADS;
    hipLaunchKernelGGL(rans_encode_kernel, dim3(blocks), dim3(RANS_THREADS), 0, st, b);
    return hipGetLastError();
}
void lit_decode_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap) {   // as rocprofv3 spells the instance launch_decode picks
    const int mm = effective_mm(b.geom.mm_uniform);
    const bool seg = b.segs != nullptr;
    const int cache = seg ? (b.cache_mode == 2u ? 2 : 0) : (b.cache_mode >= 1u && b.cache_mode <= 3u ? (int)b.cache_mode : 0);
    snprintf(buf, cap, "divans_hip::lit_decode_kernel<%d, %s, %s, %d, %s>", mm, b.geom.ctx_const >= 0 ? "true" : "false", mix ? "true" : "false", cache,
             seg ? "true" : "false");
}
hipError_t launch_decode(const LitBatch& b_in, bool mix, uint32_t blocks, hipStream_t st) {
    const LitBatch& b = b_in;
    const int mm = effective_mm(b.geom.mm_uniform);
    if (b.segs && b.cache_mode != 2u && b.cache_mode != 0u) return hipErrorInvalidValue;
    LitKernel k = pick_mode_lit_decode_kernel((int)b.cache_mode, b.segs != nullptr, mm, b.geom.ctx_const >= 0, mix);
    if (!k) return hipErrorInvalidValue;     // a cache organisation this build holds no generation-1 decoder for
    const uint32_t lds = lit_lds_bytes(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LIT_THREADS), lds, st, b);
    return hipGetLastError();
}
hipError_t launch_pack(const uint8_t* slots, const uint64_t* src_off, const uint32_t* sizes, uint32_t n, uint8_t* packed,
                       uint64_t* dst_off, uint64_t* total, hipStream_t st, bool accumulate, uint64_t cap, uint32_t* status) {
    hipLaunchKernelGGL(scan_sizes_kernel, dim3(1), dim3(1024), 0, st, sizes, n, dst_off, total, accumulate ? 1 : 0);
    uint32_t blocks = (n + 3) / 4;
    blocks = blocks > 2048 ? 2048 : (blocks ? blocks : 1);
    hipLaunchKernelGGL(pack_copy_kernel, dim3(blocks), dim3(256), 0, st, slots, src_off, sizes, n, packed, dst_off, cap, status);
    return hipGetLastError();
}
hipError_t launch_selftest_cdf_ops(const uint32_t* d_ops, uint32_t n, int32_t* d_out, hipStream_t st) {
    hipLaunchKernelGGL(cdf_ops_selftest_kernel, dim3(1), dim3(64), 0, st, (const u32x4*)d_ops, n, d_out);
    return hipGetLastError();
}
hipError_t launch_selftest_division(unsigned long long* d_mismatches, hipStream_t st) {
    hipLaunchKernelGGL(selftest_division_kernel, dim3(1024), dim3(256), 0, st, d_mismatches);
    return hipGetLastError();
}

}  // namespace divans_hip
