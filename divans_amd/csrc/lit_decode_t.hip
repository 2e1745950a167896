// lit_decode_t.hip -- the literal decoder with ONE LANE PER STREAM (decoder generation 4, "transposed"): the layout VERDICT r03
// asked to have measured against the 16-lanes-per-stream kernels of lit_kernels.hip / lit_decode2.hip.
//
// A lane owns a stream: its rANS states, history, Weights and the 16 entries of every row it touches sit in that lane's
// registers (a row = 8 dwords of two counts each), nothing crosses lanes, and the per-stream scalar work the 16-lane layout
// executes once per 16-lane row (a quarter of a wave) is executed once per lane.  Executed VALU per decoded byte falls from
// ~65 (mixing) / ~25 (plain) per stream to ~15 / ~6; what does NOT change is what DESIGN.md section 5 identifies as the bound:
// the row accesses per byte that leave the CU and the LDS a stream's row caches take.
//
//   * CDF tables: the same per-stream slabs in HBM as the other generations ([resident stream][row][16] i16); a lane reads and
//     writes its rows as two 16-byte halves.
//   * row caches: direct mapped, per stream, in LDS, laid out [slot][half][lane] x 16 bytes so that the 64 lanes of a b128 access
//     never share a bank whatever slots they index; a table without a cache keeps a single slot (it is the staging buffer
//     the entry extraction reads and delays the write-back by one byte).
//   * cdf[sym], cdf[sym-1] of a row (probability/interface.rs:97-108) are read back from the row's LDS slot with a per-lane
//     address -- registers cannot be indexed per lane; the mixed row goes through a 32-byte scratch row per lane for the same reason.
//   * blend (probability/frequentist_cdf.rs:74-85) adds the increment to both counts of a dword at once (the totals of the
//     speeds this generation accepts stay below 2^15, divans_gpu_speed_supported, so no carry crosses the halves).
//   * coded words: read straight from memory, one word ahead of their use.
//
// Arithmetic and results are those of the other generations (the GPU parity tests run all of them against the oracle).
// No segment lists, no wrap-checked speeds: divans_gpu_codec_set_decoder refuses generation 4 there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "lit_device.h"

namespace divans_hip {

namespace {

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_v4;
__device__ __forceinline__ uint32_t lds_read16(uint32_t a) { return *(const lds_u16*)(uintptr_t)a; }
__device__ __forceinline__ void lds_write16(uint32_t a, uint32_t v) { *(lds_u16*)(uintptr_t)a = (uint16_t)v; }
__device__ __forceinline__ uint32_t lds_read8(uint32_t a) { return *(const lds_u8*)(uintptr_t)a; }
__device__ __forceinline__ void lds_write32(uint32_t a, uint32_t v) { *(lds_u32*)(uintptr_t)a = v; }
__device__ __forceinline__ u32x4 lds_read128(uint32_t a) { return *(const lds_v4*)(uintptr_t)a; }
__device__ __forceinline__ void lds_write128(uint32_t a, u32x4 v) { *(lds_v4*)(uintptr_t)a = v; }

__device__ const BytePerm kBytePermT{};

constexpr uint32_t T_LANES = 64u;              // streams per workgroup = lanes of its one wave
constexpr uint32_t T_HALF = T_LANES * 16u;     // LDS distance between the two 16-byte halves of a row
constexpr uint32_t T_SLOT = 2u * T_HALF;       // one cache slot of all 64 streams
constexpr uint32_t T_TAGS = T_LANES * 2u;      // its tags

struct TCache { uint32_t data, tag, mask, shift; };     // data / tag: LDS address of slot 0 of THIS lane
// a row in registers: entries 2j, 2j+1 in dword j of a (j < 4) / b; addr = its LDS slot; fresh = it came from memory and is not in the slot yet
struct TRow { u32x4 a, b; uint32_t addr, row; bool fresh; };

struct TTable {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t off;        // this lane's slab inside the workgroup's 64
    __device__ __forceinline__ TRow lookup(const TCache& c, uint32_t row) const {
        TRow r;
        r.row = row;
        const uint32_t set = (row ^ (row >> c.shift)) & c.mask;
        r.addr = c.data + set * T_SLOT;
        const uint32_t taddr = c.tag + set * T_TAGS;
        r.a = lds_read128(r.addr); r.b = lds_read128(r.addr + T_HALF);
        const uint32_t tag = lds_read16(taddr);
        r.fresh = tag != row;
        if (r.fresh) {
            // every access of the coder is a read-modify-write: a cached row is dirty, the victim goes back to its slab
            if (tag != 0x7fffu) {
                __builtin_amdgcn_raw_buffer_store_b128(r.a, rsrc, off + (tag << 5), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(r.b, rsrc, off + (tag << 5) + 16u, 0, 0);
            }
            r.a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (row << 5), 0, 0);
            r.b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (row << 5) + 16u, 0, 0);
            lds_write16(taddr, row);
        }
        return r;
    }
};

// the row is about to be used: a row that came from memory goes into its slot (the entry extraction reads it there)
__device__ __forceinline__ void consume(const TRow& r) {
    if (r.fresh) { lds_write128(r.addr, r.a); lds_write128(r.addr + T_HALF, r.b); }
}
__device__ __forceinline__ void put_back(const TRow& r) { lds_write128(r.addr, r.a); lds_write128(r.addr + T_HALF, r.b); }

__device__ __forceinline__ uint32_t dword_of(const TRow& r, int j) { return j < 4 ? r.a[j] : r.b[j - 4]; }
__device__ __forceinline__ void set_dword(TRow& r, int j, uint32_t v) { if (j < 4) r.a[j] = v; else r.b[j - 4] = v; }
__device__ __forceinline__ uint32_t entry(const TRow& r, int i) { const uint32_t w = dword_of(r, i >> 1); return (i & 1) ? w >> 16 : w & 0xffffu; }
// LDS byte offset of entry e inside a slot
__device__ __forceinline__ uint32_t entry_off(uint32_t e) { return ((e & 8u) << 7) | ((e & 7u) << 1); }

// which halves of dword j blend increments: entries >= sym
__device__ __forceinline__ void blend_masks(uint32_t sym, uint32_t sel[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (int)sym - 2 * j;
        sel[j] = k <= 0 ? 0xffffffffu : (k == 1 ? 0xffff0000u : 0u);
    }
}

// frequentist_cdf.rs:74-85 on a packed row
__device__ __forceinline__ void blend_t(TRow& r, const uint32_t sel[8], int inc, int lim, uint32_t old_max) {
    const uint32_t incp = (uint32_t)inc * 0x10001u;
#pragma unroll
    for (int j = 0; j < 8; ++j) set_dword(r, j, dword_of(r, j) + (incp & sel[j]));
    if ((int)old_max >= lim - inc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = dword_of(r, j);
            uint32_t lo = (w & 0xffffu) + (uint32_t)(2 * j + 1), hi = (w >> 16) + (uint32_t)(2 * j + 2);
            lo -= lo >> 2; hi -= hi >> 2;
            set_dword(r, j, lo | (hi << 16));
        }
    }
}

// helper_advance_sym (ans.rs:238), as advance_state of lit_decode2.hip
__device__ __forceinline__ void advance_t(uint64_t& S, uint32_t slot, uint32_t d, uint32_t dprev) {
    const uint32_t freq = d - dprev - 1u;
    const int32_t a = (int32_t)slot - (int32_t)dprev - 1;
    const uint32_t xlo = (uint32_t)(S >> 15), xhi = (uint32_t)(S >> 47);
    const uint64_t t = (uint64_t)xlo * freq + (uint64_t)(int64_t)a;
    const uint32_t thi = (uint32_t)(t >> 32) + __umul24(xhi, freq);
    S = ((uint64_t)thi << 32) | (uint32_t)t;
}

struct MixT { uint32_t p[16]; uint32_t sym, slot, cmax, smax; };

// average (frequentist_cdf.rs:58-72) of the two rows under the nibble's normalized weight, and the symbol search
// (probability/interface.rs:136-198): sym = number of entries i < 15 with (max * slot >> 15) >= p[i]
__device__ __forceinline__ void mix_search(MixT& m, const TRow& cm, const TRow& st, int norm, uint64_t S) {
    m.cmax = cm.b[3] >> 16; m.smax = st.b[3] >> 16;
    const uint32_t prod = __umul24(m.cmax, m.smax);
    int lz = __clz((int)prod);
    lz = lz > 17 ? 17 : lz;
    const int sh = 17 - lz;
    const uint32_t inv = (uint32_t)((1 << 15) - norm);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t rs = __umul24(entry(cm, i), m.smax) >> sh;
        const uint32_t ro = __umul24(entry(st, i), m.cmax) >> sh;
        m.p[i] = (uint32_t)((int)(__umul24(rs, (uint32_t)norm) + __umul24(ro, inv) + 1u) >> 15);
    }
    m.slot = (uint32_t)S & 0x7fffu;
    const uint32_t th = __umul24(m.slot, m.p[15]) >> 15;
    uint32_t sym = 0;
#pragma unroll
    for (int i = 0; i < 15; ++i) sym += m.p[i] <= th ? 1u : 0u;
    m.sym = sym;
}

// the three (start, freq) pairs of the symbol, the state update, both blends and the Weights update (literal.rs:209-243, weights.rs:23-38)
__device__ __forceinline__ void mix_finish(const MixT& m, TRow& cm, TRow& st, uint32_t pscr, uint64_t& S, Weights& w, bool is_default,
                                           int inc_cm, int lim_cm, int inc_st, int lim_st) {
    u32x4 pa, pb;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pa[j] = m.p[2 * j] | (m.p[2 * j + 1] << 16); pb[j] = m.p[8 + 2 * j] | (m.p[9 + 2 * j] << 16); }
    lds_write128(pscr, pa); lds_write128(pscr + T_HALF, pb);
    const uint32_t os = entry_off(m.sym), op = entry_off((m.sym - 1u) & 15u);
    const bool nz = m.sym != 0u;
    const int ps = (int)lds_read16(pscr + os), pp = (int)lds_read16(pscr + op);
    const int cs = (int)lds_read16(cm.addr + os), cp = (int)lds_read16(cm.addr + op);
    const int ss = (int)lds_read16(st.addr + os), sp = (int)lds_read16(st.addr + op);
    const int pmax = (int)m.p[15];
    const float rp = biased_rcp15(pmax), rc = biased_rcp15((int)m.cmax), rs = biased_rcp15((int)m.smax);
    const uint32_t d = scaled_div(ps, pmax, rp), dprev = nz ? scaled_div(pp, pmax, rp) : 0u;
    const uint32_t dc = scaled_div(cs, (int)m.cmax, rc), dcp = nz ? scaled_div(cp, (int)m.cmax, rc) : 0u;
    const uint32_t ds = scaled_div(ss, (int)m.smax, rs), dsp = nz ? scaled_div(sp, (int)m.smax, rs) : 0u;
    advance_t(S, m.slot, d, dprev);
    uint32_t sel[8];
    blend_masks(m.sym, sel);
    blend_t(cm, sel, inc_cm, lim_cm, m.cmax);
    put_back(cm);
    if (!is_default) { blend_t(st, sel, inc_st, lim_st, m.smax); put_back(st); }
    weights_update(w, (int)(short)(dc - dcp - 1u), (int)(short)(ds - dsp - 1u), (int)(short)(d - dprev - 1u));
}

// the non-mixing nibble: search under the row (or the default CDF, mixing value 2), then (start, freq), state, blend
__device__ __forceinline__ uint32_t plain_search(const TRow& st, bool is_default, uint64_t S, uint32_t& slot, uint32_t& mx) {
    slot = (uint32_t)S & 0x7fffu;
    mx = is_default ? 64u : st.b[3] >> 16;
    const uint32_t th = __umul24(slot, mx) >> 15;
    uint32_t sym = 0;
#pragma unroll
    for (int i = 0; i < 15; ++i) sym += (is_default ? 4u * (uint32_t)(i + 1) : entry(st, i)) <= th ? 1u : 0u;
    return sym;
}
__device__ __forceinline__ void plain_finish(TRow& st, bool is_default, uint32_t sym, uint32_t slot, uint32_t mx, uint64_t& S, int inc, int lim) {
    const uint32_t os = entry_off(sym), op = entry_off((sym - 1u) & 15u);
    const bool nz = sym != 0u;
    int xs = (int)lds_read16(st.addr + os), xp = (int)lds_read16(st.addr + op);
    if (is_default) { xs = 4 * (int)(sym + 1u); xp = 4 * (int)sym; }
    const float r = biased_rcp15((int)mx);
    const uint32_t d = scaled_div(xs, (int)mx, r), dprev = nz ? scaled_div(xp, (int)mx, r) : 0u;
    advance_t(S, slot, d, dprev);
    if (!is_default) {
        uint32_t sel[8];
        blend_masks(sym, sel);
        blend_t(st, sel, inc, lim, mx);
        put_back(st);
    }
}

// the coded stream, one word ahead of its use
struct WordsT {
    const uint32_t* in; uint32_t nwords, pos, ahead;
    __device__ __forceinline__ void start() { pos = 0; ahead = nwords ? in[0] : 0u; }
    __device__ __forceinline__ uint32_t next() {
        const uint32_t v = ahead;
        pos += 1u;
        ahead = pos < nwords ? in[pos] : 0u;
        return v;
    }
};

}  // namespace

template <int MM, bool CTXC, bool MIX>
__global__ __launch_bounds__(64) void lit_decode_t_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const uint32_t lane = threadIdx.x;
    const uint32_t slab = g.total_rows * 32u;
    TTable tb;
    {
        const uint64_t base = (uint64_t)((uint8_t*)b.tables + (size_t)blockIdx.x * T_LANES * slab);
        const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        tb.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, __builtin_amdgcn_readfirstlane((int)(T_LANES * slab)), 0x00020000);
        tb.off = lane * slab;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr bool PERM = MM == 4;
    const uint32_t perm_off = lds_base + (uint32_t)(lv.mix - lv.base);
    if (PERM) {
        lds_write32(perm_off + 4u * lane, ((const uint32_t*)kBytePermT.rank)[lane]);
        __syncthreads();
    }
    // LDS of the workgroup: [mixed-row scratch 2 x 1 KB][data of the caches][their tags]
    const uint32_t pscr = lds_base + lane * 16u;
    TCache hs, hc, ls, lc;
    {
        TCache* all[4] = {&hs, &hc, &ls, &lc};
        uint32_t off = lds_base + T_SLOT;
        for (int i = 0; i < 4; ++i) {
            const uint32_t lg = (b.dm_log2 >> (8 * i)) & 0xffu;
            const uint32_t slots = (!MIX && (i & 1)) ? 0u : (lg ? 1u << (lg - 1u) : 1u);
            all[i]->data = off + lane * 16u;
            all[i]->mask = slots ? slots - 1u : 0u;
            all[i]->shift = (b.dm_shift >> (8 * i)) & 0x1fu;
            off += slots * T_SLOT;
        }
        for (int i = 0; i < 4; ++i) {
            all[i]->tag = off + lane * 2u;
            off += ((!MIX && (i & 1)) ? 0u : all[i]->mask + 1u) * T_TAGS;
        }
    }
    const uint32_t G = gridDim.x * T_LANES;
    for (uint32_t s0 = blockIdx.x * T_LANES; s0 < b.n_streams; s0 += G) {
        const uint32_t s = s0 + lane;
        const bool live = s < b.n_streams;
        const uint32_t len = live ? (b.out_sizes ? b.out_sizes[s] : b.stream_len) : 0u;
        uint8_t* out = b.out + (live ? (b.out_offsets ? b.out_offsets[s] : (uint64_t)s * b.stream_len) : 0u);
        WordsT ww;
        ww.in = (const uint32_t*)(b.in + (live ? b.in_offsets[s] : 0u));
        ww.nwords = live ? b.in_sizes[s] >> 2 : 0u;
        ww.start();
        // the 64 slabs of the workgroup are one contiguous region: default CDFs (ffi/alloc_util.rs:77-79), written cooperatively
        {
            const u32x4 lo = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
            const u32x4 hi = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};
            const u32x4 v = (lane & 1u) ? hi : lo;
            const uint32_t bytes = T_LANES * slab;
            for (uint32_t i = lane * 16u; i < bytes; i += T_LANES * 16u) __builtin_amdgcn_raw_buffer_store_b128(v, tb.rsrc, i, 0, 0);
            TCache* all[4] = {&hs, &hc, &ls, &lc};
            for (int i = 0; i < 4; ++i) {
                if (!MIX && (i & 1)) continue;
                for (uint32_t k = 0; k <= all[i]->mask; ++k) lds_write16(all[i]->tag + k * T_TAGS, 0x7fffu);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
        }
        Weights wH, wL;      // model_weights[1] (high nibble), [0] (low nibble), literal.rs:230
        wH.w0 = 1; wH.w1 = 1; wH.norm = 1 << 14; wL = wH;
        uint64_t last8 = 0;
        uint32_t p1 = 0, p2 = 0;
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS + p2];
        const uint32_t ctab = LIT_BLOB_CTXF;
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, p1, k1);
        uint64_t sel8 = PERM ? (uint64_t)lds_read8(perm_off + p1) << 56 : last8;     // what select_rows sees of the history
        uint64_t SA = 0, SB = 0;
        bool corrupt = false;
        TRow stH, cmH = {};
        bool defH;
        {
            const RowSel rs = select_rows<true, MM>(g, lv.mix, ctx_cur, sel8, 0u);
            stH = tb.lookup(hs, rs.stride_row);
            if (MIX) cmH = tb.lookup(hc, rs.cm_row);
            defH = (MM < 0 || MM == 2) && rs.is_default;
        }
        const bool aligned = (((uintptr_t)out) & 3u) == 0u;
        uint32_t outw = 0;
        uint32_t pos = 0;
        while (pos < len) {
            if ((pos & 32767u) == 0u) {
                // a 65 536-symbol chunk ends with both states back at 2^31 (ans.rs:135-136,331-378) and starts with 16 bytes = state_a, state_b (ans.rs:174-186)
                if (pos) corrupt |= (SA != (1ull << 31)) | (SB != (1ull << 31));
                const uint32_t a0 = ww.next(), a1 = ww.next(), b0 = ww.next(), b1 = ww.next();
                SA = ((uint64_t)a1 << 32) | a0;
                SB = ((uint64_t)b1 << 32) | b0;
            }
            uint32_t hi, lo;
            TRow stL, cmL = {};
            bool defL;
            if (MIX) {
                if (SA < (1ull << 31)) SA = (SA << 32) | ww.next();
                consume(stH); consume(cmH);
                MixT mh;
                mix_search(mh, cmH, stH, wH.norm, SA);
                hi = mh.sym;
                {
                    const RowSel rs = select_rows<false, MM>(g, lv.mix, ctx_cur, sel8, hi);
                    stL = tb.lookup(ls, rs.stride_row);
                    cmL = tb.lookup(lc, rs.cm_row);
                    defL = (MM < 0 || MM == 2) && rs.is_default;
                }
                mix_finish(mh, cmH, stH, pscr, SA, wH, defH, g.inc3, g.lim3, g.inc0, g.lim0);
                if (SB < (1ull << 31)) SB = (SB << 32) | ww.next();
                consume(stL); consume(cmL);
                MixT ml;
                mix_search(ml, cmL, stL, wL.norm, SB);
                lo = ml.sym;
                const uint32_t byte = (hi << 4) | lo;
                last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                p2 = p1; p1 = byte;
                if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + p2];
                ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, p1, k1);
                sel8 = PERM ? (uint64_t)lds_read8(perm_off + p1) << 56 : last8;
                {
                    const RowSel rs = select_rows<true, MM>(g, lv.mix, ctx_cur, sel8, 0u);     // next byte's rows (harmless past the end)
                    stH = tb.lookup(hs, rs.stride_row);
                    cmH = tb.lookup(hc, rs.cm_row);
                    defH = (MM < 0 || MM == 2) && rs.is_default;
                }
                mix_finish(ml, cmL, stL, pscr, SB, wL, defL, g.inc2, g.lim2, g.inc0, g.lim0);
            } else {
                if (SA < (1ull << 31)) SA = (SA << 32) | ww.next();
                consume(stH);
                uint32_t slot_a, mxa;
                hi = plain_search(stH, defH, SA, slot_a, mxa);
                {
                    const RowSel rs = select_rows<false, MM>(g, lv.mix, ctx_cur, sel8, hi);
                    stL = tb.lookup(ls, rs.stride_row);
                    defL = (MM < 0 || MM == 2) && rs.is_default;
                }
                plain_finish(stH, defH, hi, slot_a, mxa, SA, g.inc0, g.lim0);
                if (SB < (1ull << 31)) SB = (SB << 32) | ww.next();
                consume(stL);
                uint32_t slot_b, mxb;
                lo = plain_search(stL, defL, SB, slot_b, mxb);
                const uint32_t byte = (hi << 4) | lo;
                last8 = (last8 >> 8) | ((uint64_t)byte << 56);
                p2 = p1; p1 = byte;
                if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + p2];
                ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, p1, k1);
                sel8 = PERM ? (uint64_t)lds_read8(perm_off + p1) << 56 : last8;
                {
                    const RowSel rs = select_rows<true, MM>(g, lv.mix, ctx_cur, sel8, 0u);
                    stH = tb.lookup(hs, rs.stride_row);
                    defH = (MM < 0 || MM == 2) && rs.is_default;
                }
                plain_finish(stL, defL, lo, slot_b, mxb, SB, g.inc0, g.lim0);
            }
            const uint32_t q = pos & 3u;
            outw |= p1 << (8u * q);
            if (q == 3u || pos + 1u == len) {
                uint8_t* o = out + (pos - q);
                if (aligned && q == 3u) __builtin_nontemporal_store(outw, (uint32_t*)o);
                else for (uint32_t t = 0; t <= q; ++t) o[t] = (uint8_t)(outw >> (8u * t));
                outw = 0;
            }
            ++pos;
        }
        if (len) corrupt |= (SA != (1ull << 31)) | (SB != (1ull << 31));
        corrupt |= ww.pos != ww.nwords;     // every coded word consumed, none read past the end
        if (live && corrupt) {
            if (b.status) atomicOr(b.status, LIT_STATUS_BAD_STREAM);
            if (b.stream_bad) b.stream_bad[s] = 1;
        }
    }
}

typedef void (*LitKernel)(const LitBatch);

template <bool MIX>
static LitKernel pick_decode_t(int mm, bool ctxc) {
    const int key = (mm == 4 ? 2 : (mm == 0 ? 1 : 0)) * 2 + (ctxc ? 1 : 0);
    switch (key) {
    case 0: return lit_decode_t_kernel<-1, false, MIX>; case 1: return lit_decode_t_kernel<-1, true, MIX>;
    case 2: return lit_decode_t_kernel<0, false, MIX>;  case 3: return lit_decode_t_kernel<0, true, MIX>;
    case 4: return lit_decode_t_kernel<4, false, MIX>;  default: return lit_decode_t_kernel<4, true, MIX>;
    }
}

// LDS bytes one stream takes: its share of the mixed-row scratch, and 34 bytes per cache slot (a table without a cache keeps one)
uint32_t lit_decode_t_stream_lds(uint32_t dm_log2, bool mix) {
    uint32_t slots = 0;
    for (int i = 0; i < 4; ++i) {
        if (!mix && (i & 1)) continue;
        const uint32_t lg = (dm_log2 >> (8 * i)) & 0xffu;
        slots += lg ? 1u << (lg - 1u) : 1u;
    }
    return 32u + slots * 34u;
}

uint32_t lit_lds_bytes_t(const LitBatch& b) {
    uint32_t bytes = b.cache_bytes_per_wg;
    if (b.geom.ctx_const < 0) bytes += LIT_BLOB_CTXF + LIT_CTXF_BYTES * b.geom.n_btypes;
    if (!(b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4)) bytes += 8192u;
    if (b.geom.mm_uniform == 4) bytes += 256u;   // BytePerm's ranks
    return bytes;
}

void lit_decode_t_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap) {
    const int mm = (b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4) ? b.geom.mm_uniform : -1;
    snprintf(buf, cap, "divans_hip::lit_decode_t_kernel<%d, %s, %s>", mm, b.geom.ctx_const >= 0 ? "true" : "false", mix ? "true" : "false");
}

hipError_t launch_decode_t(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st) {
    if (b.segs != nullptr || b.geom.wrap_check) return hipErrorInvalidValue;
    const int mm = (b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4) ? b.geom.mm_uniform : -1;
    LitKernel k = mix ? pick_decode_t<true>(mm, b.geom.ctx_const >= 0) : pick_decode_t<false>(mm, b.geom.ctx_const >= 0);
    const uint32_t lds = lit_lds_bytes_t(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), lds, st, b);
    return hipGetLastError();
}

}  // namespace divans_hip
