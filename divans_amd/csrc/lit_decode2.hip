// lit_decode2.hip -- the literal decoder's second generation: the same 16-lane DPP-row mapping as lit_kernels.hip
// (lane i of a row holds cdf[i], four streams per wave64), rebuilt around the instruction-issue budget of gfx950
// (~2.4 cycles for the add / logic / shift-right / fp32 class, ~4.2 for everything else):
//
//   * symbol search (probability/interface.rs:136-198): the compare runs on (cdf << 15) against max * slot, i.e. on the
//     numerator the division needs anyway; the row's CDF is non-decreasing, so the lanes above the coded slot form a suffix of
//     the row whose first lane is the symbol (v_ffbl), and the same predicate later selects the lanes blend increments;
//   * (start, freq) of the symbol (probability/interface.rs:97-108): one division pass over the 16 entries, then TWO
//     ds_bpermute reads (the scaled entry and its row_shr:1 copy) instead of packing, permuting and unpacking -- the LDS pipe
//     does not cost VALU issue cycles;
//   * blend (probability/frequentist_cdf.rs:74-85): the renormalisation half sits behind a wave-level branch (with the
//     reference's speeds a row renormalises once in hundreds of updates);
//   * row caches: 2-way, both ways' data and the tag word are read in parallel (one LDS round trip), one cache per table so that
//     the rows of one nibble never compete for a set;
//   * coded words: a 32-word ring per stream in LDS, topped up 16 words at a time (ans.rs:428-442 reads them in this order).
//
// Arithmetic and results are those of lit_decode_kernel (the parity tests run both against the oracle).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "lit_device.h"

// experiment switches (scripts/build_variants.sh); the defaults are the product
#ifndef DIVANS_D2_ASYNC
#define DIVANS_D2_ASYNC 1
#endif
#ifndef DIVANS_D2_W7
#define DIVANS_D2_W7 1
#endif
#ifndef DIVANS_D2_PERM          // stride-1 instances: index the tables by a frequency rank of the previous byte (LitBatch::byte_rank)
#define DIVANS_D2_PERM 1
#endif
#ifndef DIVANS_D2_LOAD_AUX      // cache-policy bits of the row loads / stores that go to memory: 1 = sc0, 2 = nt, 16 = sc1
#define DIVANS_D2_LOAD_AUX 0
#endif
#ifndef DIVANS_D2_STORE_AUX
#define DIVANS_D2_STORE_AUX 0
#endif
#ifndef DIVANS_D2_PAD_VALU      // counterfactual: N extra 4-cycle-class VALU instructions per byte (is the kernel bound by VALU issue?)
#define DIVANS_D2_PAD_VALU 0
#endif

namespace divans_hip {

namespace {

// LDS is addressed with 32-bit byte addresses (address space 3): no generic-pointer arithmetic in the byte loop
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_read16(uint32_t a) { return *(const lds_u16*)(uintptr_t)a; }
__device__ __forceinline__ void lds_write16(uint32_t a, uint32_t v) { *(lds_u16*)(uintptr_t)a = (uint16_t)v; }
__device__ __forceinline__ uint32_t lds_read8(uint32_t a) { return *(const lds_u8*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_read32(uint32_t a) { return *(const lds_u32*)(uintptr_t)a; }
__device__ __forceinline__ void lds_write32(uint32_t a, uint32_t v) { *(lds_u32*)(uintptr_t)a = v; }

constexpr uint32_t kRingWords = 32u;
constexpr uint32_t kRingBytes = kRingWords * 4u;

// One per-stream row cache in LDS, write back, in one of two organisations (compile-time, CM_2WAY):
//  * direct mapped: slot = 16 lanes x u16, tag = u16 row id (0x7fff = empty); the slot's data and its tag are read in parallel;
//  * 2-way set associative: a set is 16 lanes x u32 of data -- lane i holds entry i of way 0 in the low half and of way 1 in the
//    high half, so ONE ds_read_b32 per lane fetches both candidates while the set's tag word (way 0 row id | way 1 row id << 15 |
//    most recently used way << 31) is read in parallel.
// Either way a hit costs one LDS round trip.
struct DmCache {
    uint32_t data_off;   // LDS address of slot / set 0 of this stream's cache + (2 or 4) * lane-in-row
    uint32_t tag_off;    // LDS address of its tags
    uint32_t mask;       // slots - 1 / sets - 1
    uint32_t shift;      // index = (row ^ (row >> shift)) & mask
};

struct RowSlot { uint32_t row; uint32_t addr; bool missed; };   // missed: the row was requested with gload_async()

struct Table2 {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane_off;   // row-of-lanes slab offset + 2 * lane-in-row
    __device__ __forceinline__ int gload(uint32_t row) const {
        return (int)__builtin_amdgcn_raw_buffer_load_b16(rsrc, lane_off + (row << 5), 0, DIVANS_D2_LOAD_AUX);
    }
    __device__ __forceinline__ void gstore(uint32_t row, int v) const {
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v, rsrc, lane_off + (row << 5), 0, DIVANS_D2_STORE_AUX);
    }
    // A high-nibble row fetched on a cache miss is requested one nibble ahead and first used at the top of the next byte.  The
    // vector-memory counter covers loads AND stores (and a store may be acknowledged before an older load has returned, so only
    // vmcnt(0) proves a load complete): a compiler-placed wait at the top of the byte would wait for the row stores the previous
    // byte ended with on EVERY byte -- a store acknowledgement on the dependency chain -- although most bytes miss nothing.
    // These loads therefore go out through inline asm (the compiler does not see them), the lookup reports the miss, and the
    // byte loop waits (wait_async) only when some stream of the wave did miss.
    __device__ __forceinline__ int gload_async(uint32_t row) const {
        int v;
        asm volatile("buffer_load_ushort %0, %1, %2, 0 offen" : "=v"(v) : "v"(lane_off + (row << 5)), "s"(rsrc) : "memory");
        return v;
    }
    // Every access of the coder is a read-modify-write of a whole row: a cached row is always dirty, a miss writes the
    // victim's row back (its data is in what the speculative read returned) and fetches the new one.
    template <bool PRESENT, bool TWOWAY, bool ASYNC>
    __device__ __forceinline__ int load(const DmCache& d, uint32_t row, RowSlot& s) const {
        s.row = row;
        s.missed = false;
        if (!PRESENT) return gload(row);
        const uint32_t set = (row ^ (row >> d.shift)) & d.mask;
        if (!TWOWAY) {
            s.addr = d.data_off + (set << 5);
            const uint32_t taddr = d.tag_off + (set << 1);
            int v = (int)lds_read16(s.addr);
            const uint32_t tag = lds_read16(taddr);
            if (tag != row) {
                if (tag != 0x7fffu) gstore(tag, v);
                v = ASYNC ? gload_async(row) : gload(row);
                lds_write16(taddr, row);
                s.missed = ASYNC;
            }
            return v;
        }
        const uint32_t daddr = d.data_off + (set << 6);
        const uint32_t taddr = d.tag_off + (set << 2);
        const uint32_t pair = lds_read32(daddr);
        const uint32_t tp = lds_read32(taddr);
        const uint32_t t0 = tp & 0x7fffu, t1 = (tp >> 15) & 0x7fffu;
        const bool h0 = t0 == row, h1 = t1 == row;
        const uint32_t way = h1 ? 1u : (h0 ? 0u : ((tp >> 31) ^ 1u));
        int v = (int)(way ? pair >> 16 : pair & 0xffffu);
        s.addr = daddr + (way << 1);
        uint32_t ntp = tp;
        if (!(h0 || h1)) {
            const uint32_t victim = way ? t1 : t0;
            if (victim != 0x7fffu) gstore(victim, v);
            v = ASYNC ? gload_async(row) : gload(row);
            s.missed = ASYNC;
            ntp = way ? ((tp & ~(0x7fffu << 15)) | (row << 15)) : ((tp & ~0x7fffu) | row);
        }
        ntp = (ntp & 0x7fffffffu) | (way << 31);
        if (ntp != tp) lds_write32(taddr, ntp);
        return v;
    }
    template <bool PRESENT>
    __device__ __forceinline__ void store(const DmCache& d, const RowSlot& s, int v) const {
        if (!PRESENT) gstore(s.row, v);
        else lds_write16(s.addr, (uint32_t)v);
    }
    template <bool PRESENT, bool TWOWAY>
    __device__ __forceinline__ void reset(const DmCache& d, int li) const {
        if (!PRESENT) return;
        if (TWOWAY) { for (uint32_t s = (uint32_t)li; s <= d.mask; s += 16u) lds_write32(d.tag_off + (s << 2), 0x3fffffffu); }
        else { for (uint32_t s = (uint32_t)li; s <= d.mask; s += 16u) lds_write16(d.tag_off + (s << 1), 0x7fffu); }
    }
};

// wait for the gload_async() requests; the value operands tie the first uses of the loaded registers to this point
__device__ __forceinline__ void wait_async(int& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void wait_async(int& a, int& b) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }

template <int N>
__device__ __forceinline__ void pad_valu(int& x) {
    if constexpr (N > 0) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x)); pad_valu<N - 1>(x); }
}

struct Caches { DmCache hs, hc, ls, lc; };   // high stride rows, high context-map rows (FirstNibble), low stride rows, low context-map rows
// which of them exist is a compile-time mask CM (an absent cache then costs no registers): bit 0 hs, 1 hc, 2 ls, 3 lc
constexpr int CM_HS = 1, CM_HC = 2, CM_LS = 4, CM_LC = 8, CM_2WAY = 16;   // CM_2WAY: the organisation of all of them

__device__ __forceinline__ uint32_t dm_rows(uint32_t packed, int i) { return (packed >> (8 * i)) & 0xffu; }   // log2(rows) + 1, 0 = absent

// per-stream LDS region: [ring][hs data][hc data][ls data][lc data][hs tags][hc tags][ls tags][lc tags]
template <bool TWOWAY>
__device__ __forceinline__ Caches make_caches(const LitBatch& b, uint32_t stream_base, int li) {
    Caches c;
    DmCache* all[4] = {&c.hs, &c.hc, &c.ls, &c.lc};
    uint32_t off = stream_base + kRingBytes;
    for (int i = 0; i < 4; ++i) {
        const uint32_t lg = dm_rows(b.dm_log2, i);
        const uint32_t rows = lg ? 1u << (lg - 1u) : 0u;
        all[i]->data_off = off + (TWOWAY ? 4u : 2u) * (uint32_t)li;
        all[i]->mask = rows ? (TWOWAY ? rows >> 1 : rows) - 1u : 0u;
        all[i]->shift = (b.dm_shift >> (8 * i)) & 0x1fu;   // bit 31 of dm_shift selects the 2-way organisation
        off += rows * 32u;
    }
    for (int i = 0; i < 4; ++i) {
        const uint32_t lg = dm_rows(b.dm_log2, i);
        const uint32_t rows = lg ? 1u << (lg - 1u) : 0u;
        all[i]->tag_off = off;
        off += rows * 2u;
    }
    return c;
}

// The coded stream, 32 words at a time in LDS (one ring per stream): word p sits at ring[p & 31]; when the reader crosses a
// multiple of 16 the half it has just left takes the 16 words after the other half (held in `wn` since the previous crossing).
struct WordRing {
    const uint32_t* in; uint32_t nwords, pos, wn, off;     // off = LDS address of the ring
    __device__ __forceinline__ uint32_t fetch(uint32_t first, int li) const { return (first + li < nwords) ? in[first + li] : 0u; }
    __device__ __forceinline__ void start(int li) {
        pos = 0;
        lds_write32(off + ((uint32_t)li << 2), fetch(0u, li));
        lds_write32(off + ((16u + (uint32_t)li) << 2), fetch(16u, li));
        wn = fetch(32u, li);
    }
    __device__ __forceinline__ uint32_t next(int li) {
        const uint32_t v = lds_read32(off + ((pos & (kRingWords - 1u)) << 2));
        pos += 1u;
        if ((pos & 15u) == 0u) {
            lds_write32(off + ((((pos + 16u) & (kRingWords - 1u)) + (uint32_t)li) << 2), wn);
            wn = fetch(pos + 32u, li);
        }
        return v;
    }
};

// The decoded bytes the context and row selection look back at (last_8_literals, codec/literal.rs:66-85).  Configurations
// whose every mixing value is 0 or 4 (stride <= 1) read the previous byte only, the context tables the one before it as well:
// the 8-byte window is then never assembled.
template <bool NEED8>
struct History {
    uint64_t last8; uint32_t p1, p2;
    uint32_t pb;         // rank of p1 in the launch's byte order (stride-1 instances), read from the LDS copy at perm_off
    uint32_t perm_off;
    __device__ __forceinline__ void set(uint64_t v, bool perm) { last8 = v; p1 = (uint32_t)(v >> 56); p2 = (uint32_t)(v >> 48) & 0xffu; if (perm) pb = lds_read8(perm_off + p1); }
    __device__ __forceinline__ void push(uint32_t byte, bool perm) {
        p2 = p1; p1 = byte;
        if (perm) pb = lds_read8(perm_off + byte);
        if (NEED8) last8 = (last8 >> 8) | ((uint64_t)byte << 56);
    }
    __device__ __forceinline__ uint32_t stride_byte(uint32_t offset_bits) const { return NEED8 ? (uint32_t)(last8 >> (56u - offset_bits)) & 0xffu : p1; }
};

// codec/literal.rs:176-208, as select_rows of lit_device.h but on a History
template <bool HIGH, int MM, bool NEED8>
__device__ __forceinline__ RowSel select_rows2(const LitGeometry& g, const uint8_t* lds_mix, uint32_t ctxk, const History<NEED8>& h, uint32_t hi_nib) {
    const uint32_t ctx = ctxk & 0xffu;      // ctxk = context | row slot of (prev, class of prev_prev) << 8 (context_of)
    uint32_t mm_opts;
    if (MM >= 0) mm_opts = (uint32_t)MM;
    else mm_opts = lds_mix[ctx | (HIGH ? ((h.p1 >> 4) << 8) : ((hi_nib << 8) | 4096u))];
    const uint32_t fast_cm = (mm_opts != 3) ? 0xffu : 0u;
    const uint32_t mm = (mm_opts != 0 && mm_opts != 3) ? 0xffu : 0u;
    const uint32_t opt1 = (mm_opts == 1) ? 0xfu : 0u;
    uint32_t stride_offset = 0;
    if (mm_opts >= 4) { uint32_t x = mm_opts ^ 4u; stride_offset = (x < 7u ? x : 7u) << 3; }
    const uint32_t sb = (DIVANS_D2_PERM && MM == 4) ? h.pb : h.stride_byte(stride_offset);
    uint32_t b, c, width;
    if (HIGH) {
        b = sb & mm & ~opt1 & 0xffu; c = ctx; width = g.nctx;
        if (MM == 4 && g.hs_classes) { c = ctxk >> 8; width = g.hs_classes; }     // [slot of the class of prev_prev][prev]: the reachable rows only
    }
    else { b = ((mm & sb) | (~mm & ctx)) & 0xffu; c = (hi_nib & fast_cm) | ((ctx & opt1) << 4); width = g.low_width; }
    const uint32_t t = (mm >> 7) ^ (opt1 >> 2);
    const uint32_t plane = t == 0 ? g.plane0 : (t == 1 ? g.plane1 : g.plane2);
    RowSel r;
    r.stride_row = (HIGH ? 0u : g.low_base) + ((plane * width + c) << 8) + b;
    r.cm_row = g.cm_base + (HIGH ? ctx : g.nctx + hi_nib + 16u * ctx);
    r.is_default = mm_opts == 2;
    return r;
}

struct Searched {
    int sym;     // the decoded nibble
    int mx;      // cdf[15] of the row that was searched
    int c15;     // this lane's entry << 15
    bool above;  // this lane's entry is above the coded slot  <=>  lane >= sym  (always true in lane 15)
};

// cdf_offset_to_sym_start_and_freq, the search: sym = number of entries i < 15 with (max * slot >> 15) >= cdf[i].
// (max * slot >> 15) >= c  <=>  max * slot >= c << 15, and c[15] << 15 = max << 15 > max * slot.
__device__ __forceinline__ Searched search2(int cv, uint32_t slot, int rbase) {
    Searched r;
    r.mx = row_bcast<15>(cv);
    const uint32_t prod = __umul24(slot, (uint32_t)r.mx);
    r.c15 = cv << 15;
    r.above = (uint32_t)r.c15 > prod;
    const unsigned long long m = __ballot(r.above);
    r.sym = __builtin_ctz((uint32_t)(m >> rbase));
    __builtin_assume(r.sym >= 0 && r.sym < 16);
    return r;
}

// helper_advance_sym (ans.rs:238): x = freq * (state >> 15) + (state & mask) - start, with start = dprev + 1 and
// freq = d - dprev - 1 taken from the two scaled CDF entries around the symbol.
__device__ __forceinline__ void advance_state(uint64_t& S, uint32_t slot, uint32_t d, uint32_t dprev) {
    const uint32_t freq = d - dprev - 1u;
    const int32_t a = (int32_t)slot - (int32_t)dprev - 1;            // slot - start: non-negative for every stream an encoder produced
    const uint32_t xlo = (uint32_t)(S >> 15), xhi = (uint32_t)(S >> 47);
    const uint64_t t = (uint64_t)xlo * freq + (uint64_t)(int64_t)a;
    const uint32_t thi = (uint32_t)(t >> 32) + __umul24(xhi, freq);     // v_mad_u32_u24: xhi < 2^16, freq < 2^16
    S = ((uint64_t)thi << 32) | (uint32_t)t;
}

// frequentist_cdf.rs:74-85 with the search predicate as the increment mask and the row's previous total at hand
__device__ __forceinline__ int blend2(int c, int li1, bool above, int inc, int lim, int old_max) {
    int c2 = c + inc;
    asm("" : "+v"(c2));                    // one add and one select (not a select of the increment followed by the add)
    c = above ? c2 : c;
    const bool renorm = old_max >= lim - inc;       // lim - inc is uniform: a scalar subtraction
    if (__builtin_expect(__ballot(renorm) != 0ull, 0)) {
        asm volatile("" ::: "memory");     // keep this a branch: with the reference's speeds it is taken once in hundreds of updates
        const int t = c + li1;
        c = renorm ? t - (t >> 2) : c;
    }
    return c;
}

// sym_to_start_and_freq for the searched symbol under the row `cv`, state update, blend and store: the non-mixing nibble
template <bool PRESENT>
__device__ __forceinline__ void finish2(const Table2& tb, const DmCache& dc, int li1, int rbase4, const RowSlot& slot_ref, int value,
                                        bool is_default, int cv, const Searched& s, uint32_t slot, uint64_t& S, int inc, int lim) {
    const float rl = biased_rcp15(s.mx);
    const uint32_t q = (uint32_t)((float)cv * rl);
    const int32_t r = s.c15 - __mul24((int)q, s.mx);
    const uint32_t d = r >= s.mx ? q + 1u : q;
    const int dp = row_prev_or_zero((int)d);
    const int addr = rbase4 + (s.sym << 2);
    const uint32_t dsym = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)d);
    const uint32_t dpsym = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, dp);
    advance_state(S, slot, dsym, dpsym);
    int st = value;
    if (!is_default) st = blend2(st, li1, s.above, inc, lim, s.mx);     // cv == value here, so mx is its total and `above` its predicate
    tb.template store<PRESENT>(dc, slot_ref, st);
}

// The mixing nibble's three (start, freq) pairs -- the symbol under the mixed row p, under the context-map row and under the
// stride row, each entry scaled by its own row total -- through ONE division pass: lanes 0/1 take p, lanes 4/5 the context-map
// row, lanes 8/9 the stride row; even lanes entry sym, odd lanes entry sym-1 (lane 15 of the own row when sym == 0: its scaled
// value is 2^15, which the 15-bit mask turns into the 0 the reference uses there).
struct MixLanes { int addr_bias; bool is_p, is_cm, odd; };
__device__ __forceinline__ uint32_t mixed_sf2(int p, int cm, int st, int pmax, int cmax, int smax, const MixLanes& ml, int rbase4, int sym,
                                              uint32_t& dprev_out, uint32_t& wfreqs) {
    const int a = rbase4 | (((sym << 2) + ml.addr_bias) & 60);
    const int gp = __builtin_amdgcn_ds_bpermute(a, p);
    const int gc = __builtin_amdgcn_ds_bpermute(a, cm);
    const int gs = __builtin_amdgcn_ds_bpermute(a, st);
    const int num = ml.is_p ? gp : (ml.is_cm ? gc : gs);
    const int den = ml.is_p ? pmax : (ml.is_cm ? cmax : smax);
    const uint32_t q = scaled_div(num, den, biased_rcp15(den));
    const uint32_t qn = (uint32_t)row_next_or_zero((int)q) & 0x7fffu;      // odd neighbour: scaled entry sym-1
    const int f = (int)q - (int)qn - 1;                                     // even lanes: freq of their row's entry
    dprev_out = (uint32_t)row_bcast<1>((int)q) & 0x7fffu;
    wfreqs = ((uint32_t)row_bcast<4>(f) & 0xffffu) | ((uint32_t)row_bcast<8>(f) << 16);
    return (uint32_t)row_bcast<0>((int)q);
}

// The mixing nibble in three steps, so that the byte loop can request the NEXT nibble's rows as soon as a symbol is known and
// run the rest of the current one (three (start, freq) pairs, state update, two blends) under those fetches.
struct MixRows { RowSlot sref, cref; int st, cm; bool is_default; };
struct MixSearched { Searched s; int cv, cmax, smax; uint32_t slot; };

template <bool HIGH, int MM, bool NEED8, int CM>
__device__ __forceinline__ MixRows fetch_mix2(const LitGeometry& g, const LdsView& lv, const Table2& tb, const Caches& cc, uint32_t ctx,
                                              const History<NEED8>& hist, uint32_t hi_nib) {
    const RowSel rs = select_rows2<HIGH, MM, NEED8>(g, lv.mix, ctx, hist, hi_nib);
    constexpr bool ST_C = (CM & (HIGH ? CM_HS : CM_LS)) != 0, CM_C = (CM & (HIGH ? CM_HC : CM_LC)) != 0;
    MixRows r;
    r.st = tb.template load<ST_C, (CM & CM_2WAY) != 0, HIGH && DIVANS_D2_ASYNC>(HIGH ? cc.hs : cc.ls, rs.stride_row, r.sref);
    r.cm = tb.template load<CM_C, (CM & CM_2WAY) != 0, HIGH && DIVANS_D2_ASYNC>(HIGH ? cc.hc : cc.lc, rs.cm_row, r.cref);
    r.is_default = (MM < 0 || MM == 2) && rs.is_default;
    return r;
}

__device__ __forceinline__ MixSearched search_mix2(const MixRows& r, uint64_t S, int mix_rate, int rbase) {
    MixSearched m;
    m.cmax = row_bcast<15>(r.cm); m.smax = row_bcast<15>(r.st);
    m.cv = average_rows(r.cm, r.st, m.cmax, m.smax, mix_rate);
    m.slot = (uint32_t)S & 0x7fffu;
    m.s = search2(m.cv, m.slot, rbase);
    return m;
}

template <bool HIGH, int CM>
__device__ __forceinline__ void finish_mix2(const LitGeometry& g, const Table2& tb, const Caches& cc, int li1, int rbase4, const MixLanes& ml,
                                            MixRows& r, const MixSearched& m, uint64_t& S, uint32_t& wfreqs, uint32_t& wpmix) {
    constexpr bool ST_C = (CM & (HIGH ? CM_HS : CM_LS)) != 0, CM_C = (CM & (HIGH ? CM_HC : CM_LC)) != 0;
    uint32_t dprev;
    const uint32_t d = mixed_sf2(m.cv, r.cm, r.st, m.s.mx, m.cmax, m.smax, ml, rbase4, m.s.sym, dprev, wfreqs);
    wpmix = d - dprev - 1u;
    advance_state(S, m.slot, d, dprev);
    const int cm = blend2(r.cm, li1, m.s.above, HIGH ? g.inc3 : g.inc2, HIGH ? g.lim3 : g.lim2, m.cmax);
    tb.template store<CM_C>(HIGH ? cc.hc : cc.lc, r.cref, cm);
    int st = r.st;
    if (!r.is_default) st = blend2(st, li1, m.s.above, g.inc0, g.lim0, m.smax);
    tb.template store<ST_C>(HIGH ? cc.hs : cc.ls, r.sref, st);
}

struct Fetched2 { RowSlot ref; int value; bool is_default; };

template <bool HIGH, int MM, bool NEED8, int CM>
__device__ __forceinline__ Fetched2 fetch2(const LitGeometry& g, const LdsView& lv, const Table2& tb, const Caches& cc, uint32_t ctx,
                                           const History<NEED8>& hist, uint32_t hi_nib) {
    const RowSel rs = select_rows2<HIGH, MM, NEED8>(g, lv.mix, ctx, hist, hi_nib);
    Fetched2 f;
    f.value = tb.template load<(CM & (HIGH ? CM_HS : CM_LS)) != 0, (CM & CM_2WAY) != 0, HIGH && DIVANS_D2_ASYNC>(HIGH ? cc.hs : cc.ls, rs.stride_row, f.ref);
    f.is_default = (MM < 0 || MM == 2) && rs.is_default;
    return f;
}

template <int CM>
__device__ __forceinline__ void init_table2(const Table2& t, const Caches& cc, uint32_t rows, int li) {
    // row = 16 x i16 = two 16-byte halves; even lanes write the first half, odd lanes the second (ffi/alloc_util.rs:77-79:
    // allocations are default-initialised = every row the default CDF)
    const u32x4 lo = {4u | (8u << 16), 12u | (16u << 16), 20u | (24u << 16), 28u | (32u << 16)};
    const u32x4 hi = {36u | (40u << 16), 44u | (48u << 16), 52u | (56u << 16), 60u | (64u << 16)};
    const u32x4 v = (li & 1) ? hi : lo;
    const uint32_t base = t.lane_off - 2u * (uint32_t)li + 16u * (uint32_t)li;
    for (uint32_t i = 0; i < rows * 32u; i += 256u) {
        if (i + 16u * (uint32_t)li < rows * 32u) __builtin_amdgcn_raw_buffer_store_b128(v, t.rsrc, base + i, 0, 0);
    }
    constexpr bool W2 = (CM & CM_2WAY) != 0;
    t.template reset<(CM & CM_HS) != 0, W2>(cc.hs, li); t.template reset<(CM & CM_HC) != 0, W2>(cc.hc, li);
    t.template reset<(CM & CM_LS) != 0, W2>(cc.ls, li); t.template reset<(CM & CM_LC) != 0, W2>(cc.lc, li);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // the row loads that follow must see the fill
}

}  // namespace

template <int MM, bool CTXC, bool MIX, bool SEG, int CM>
__device__ __forceinline__ void decode2_body(const LitBatch& b, uint8_t* lds) {
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, li = lane & 15, rbase = lane & 48, rbase4 = rbase << 2, li1 = li + 1;
    const uint32_t gg = blockIdx.x * (LIT_THREADS / 16) + (threadIdx.x >> 4);
    const uint32_t G = gridDim.x * (LIT_THREADS / 16);
    Table2 tb;
    {
        const uint32_t slab = g.total_rows * 32u;           // bytes of one stream's table
        // the slab pointer is uniform but its 64-bit product is computed on the VALU: read it back explicitly, or every buffer
        // access below is wrapped in a waterfall loop (the descriptor would count as divergent)
        const uint64_t base = (uint64_t)((uint8_t*)b.tables + (size_t)blockIdx.x * (LIT_THREADS / 16) * slab);
        const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        tb.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, __builtin_amdgcn_readfirstlane((int)((LIT_THREADS / 16) * slab)), 0x00020000);
        tb.lane_off = (threadIdx.x >> 4) * slab + 2u * (uint32_t)li;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr bool PERM = DIVANS_D2_PERM && MM == 4;
    const uint32_t perm_off = lds_base + (uint32_t)(lv.mix - lv.base);   // behind the configuration tables (lit_lds_bytes2; MM >= 0: no mixing mask there)
    if (PERM) {     // LitBatch::byte_rank: the ranks the codec learned from its data (or was given), or the bytes themselves
        if (threadIdx.x < 64u) lds_write32(perm_off + 4u * threadIdx.x, b.byte_rank ? ((const uint32_t*)b.byte_rank)[threadIdx.x] : 0x03020100u + 0x04040404u * threadIdx.x);
        __syncthreads();
    }
    const uint32_t stream_base = lds_base + (threadIdx.x >> 4) * (b.cache_bytes_per_wg / (LIT_THREADS / 16));
    const Caches cc = make_caches<(CM & CM_2WAY) != 0>(b, stream_base, li);
    MixLanes ml;
    ml.odd = (li & 1) != 0; ml.is_p = li < 4; ml.is_cm = (li & 12) == 4; ml.addr_bias = ml.odd ? -4 : 0;
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint32_t len = b.out_sizes ? b.out_sizes[s] : b.stream_len;
        uint8_t* out = b.out + (b.out_offsets ? b.out_offsets[s] : (uint64_t)s * b.stream_len);
        WordRing ww;
        ww.in = (const uint32_t*)(b.in + b.in_offsets[s]);
        ww.nwords = b.in_sizes[s] >> 2;
        ww.off = stream_base;
        ww.start(li);
        init_table2<CM>(tb, cc, g.total_rows, li);
        WeightsPair wp; wp.init();
        int nh = 1 << 14, nl = 1 << 14;     // normalized_weight of model_weights[1] (high nibble) / [0] (low nibble)
        constexpr bool NEED8 = SEG || !(MM == 0 || MM == 4);
        History<NEED8> hist;
        hist.perm_off = perm_off;
        uint64_t seg_last8 = 0;
        uint32_t ctab = LIT_BLOB_CTXF;      // context table of the current literal block type
        SegCursor sc;
        if (SEG) sc.start(b, s, seg_last8, ctab);
        hist.set(seg_last8, PERM);
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS + hist.p2];   // lut1 class of the byte before the previous one
        uint64_t SA = 0, SB = 0;      // state_a decodes high nibbles, state_b low nibbles (two symbols per byte)
        bool corrupt = false;
        int pad = li;
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, hist.p1, k1);
        Fetched2 rowH = {};
        MixRows mrowH = {};
        if (!MIX) rowH = fetch2<true, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, 0u);
        else mrowH = fetch_mix2<true, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, 0u);
        for (uint32_t cbeg = 0; cbeg < len; cbeg += 32768u) {
            // start of a 65 536-symbol chunk: 16 bytes = state_a, state_b (ans.rs:174-186)
            {
                const uint32_t a0 = ww.next(li), a1 = ww.next(li), b0 = ww.next(li), b1 = ww.next(li);
                SA = ((uint64_t)a1 << 32) | a0;
                SB = ((uint64_t)b1 << 32) | b0;
            }
            const uint32_t cend = cbeg + 32768u < len ? cbeg + 32768u : len;
            for (uint32_t base = cbeg; base < cend; base += 16u) {
                const uint32_t cnt = cend - base < 16u ? cend - base : 16u;
                uint32_t outb = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    if (MIX) {
                        // a state that dropped below 2^31 takes 4 more bytes right before it is used again (ans.rs:432-440)
                        if (SA < (1ull << 31)) SA = (SA << 32) | ww.next(li);
                        // the rows requested one nibble ahead, if a stream of this wave missed its cache
                        if (DIVANS_D2_ASYNC && (CM & (CM_HS | CM_HC)) && __ballot(mrowH.sref.missed || mrowH.cref.missed) != 0ull) wait_async(mrowH.st, mrowH.cm);
                        const MixSearched mh = search_mix2(mrowH, SA, nh, rbase);
                        const uint32_t hi = (uint32_t)mh.s.sym;
                        MixRows mrowL = fetch_mix2<false, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, hi);
                        uint32_t fh = 0, fl = 0, ph = 0, pl = 0;
                        finish_mix2<true, CM>(g, tb, cc, li1, rbase4, ml, mrowH, mh, SA, fh, ph);
                        if (SB < (1ull << 31)) SB = (SB << 32) | ww.next(li);
                        const MixSearched mlo = search_mix2(mrowL, SB, nl, rbase);
                        const uint32_t lo = (uint32_t)mlo.s.sym;
                        const uint32_t byte = (hi << 4) | lo;
                        hist.push(byte, PERM);
                        if (SEG) {   // the next Literal command starts from the ring buffer's last 8 bytes and its own block type
                            if (--sc.left == 0u) { seg_last8 = hist.last8; sc.advance(g, seg_last8, ctab); hist.set(seg_last8, PERM); }
                        }
                        if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + hist.p2];
                        ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, hist.p1, k1);
                        mrowH = fetch_mix2<true, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, 0u);   // next byte's rows (harmless past the end)
                        finish_mix2<false, CM>(g, tb, cc, li1, rbase4, ml, mrowL, mlo, SB, fl, pl);
                        wp.update(li, fh, ph, fl, pl);
                        nh = wp.norm_high(); nl = wp.norm_low();
                        pad_valu<DIVANS_D2_PAD_VALU>(pad);
                        outb = (uint32_t)li == k ? byte : outb;
                    } else {
                        // rowH (this byte's high-nibble row) was requested while the previous byte was being finished
                        if (SA < (1ull << 31)) SA = (SA << 32) | ww.next(li);
                        if (DIVANS_D2_ASYNC && (CM & CM_HS) && __ballot(rowH.ref.missed) != 0ull) wait_async(rowH.value);
                        const int cvh = rowH.is_default ? 4 * li1 : rowH.value;
                        const uint32_t slot_a = (uint32_t)SA & 0x7fffu;
                        const Searched sh = search2(cvh, slot_a, rbase);
                        const uint32_t hi = (uint32_t)sh.sym;
                        const Fetched2 rowL = fetch2<false, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, hi);
                        finish2<(CM & CM_HS) != 0>(tb, cc.hs, li1, rbase4, rowH.ref, rowH.value, rowH.is_default, cvh, sh, slot_a, SA, g.inc0, g.lim0);
                        if (SB < (1ull << 31)) SB = (SB << 32) | ww.next(li);
                        const int cvl = rowL.is_default ? 4 * li1 : rowL.value;
                        const uint32_t slot_b = (uint32_t)SB & 0x7fffu;
                        const Searched sl = search2(cvl, slot_b, rbase);
                        const uint32_t lo = (uint32_t)sl.sym;
                        const uint32_t byte = (hi << 4) | lo;
                        hist.push(byte, PERM);
                        if (SEG) { if (--sc.left == 0u) { seg_last8 = hist.last8; sc.advance(g, seg_last8, ctab); hist.set(seg_last8, PERM); } }
                        if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + hist.p2];
                        ctx_cur = context_of<CTXC>(g, lv.ctx, ctab, hist.p1, k1);
                        rowH = fetch2<true, MM, NEED8, CM>(g, lv, tb, cc, ctx_cur, hist, 0u);   // next byte's row (harmless past the end)
                        finish2<(CM & CM_LS) != 0>(tb, cc.ls, li1, rbase4, rowL.ref, rowL.value, rowL.is_default, cvl, sl, slot_b, SB, g.inc0, g.lim0);
                        pad_valu<DIVANS_D2_PAD_VALU>(pad);
                        outb = (uint32_t)li == k ? byte : outb;
                    }
                }
                if ((uint32_t)li < cnt) __builtin_nontemporal_store((uint8_t)outb, out + base + li);
            }
            // rANS is an exact inverse: a chunk that was coded from the start states 2^31 (ans.rs:135-136,331-378) decodes back
            // to exactly those; anything else means a truncated, corrupt or mismatched stream
            corrupt |= (SA != (1ull << 31)) | (SB != (1ull << 31));
        }
        corrupt |= ww.pos != ww.nwords;     // every coded word consumed, none read past the end
        if (corrupt && li == 0) {
            if (b.status) atomicOr(b.status, LIT_STATUS_BAD_STREAM);
            if (b.stream_bad) b.stream_bad[s] = 1;
        }
    }
}

// Seven waves per SIMD hide the per-byte HBM round trip better than six with a few more registers each: the instances whose
// mixing value is a compile-time constant are held to 72 VGPRs (the mixing ones then park one 64-bit value per stream in
// scratch, outside the byte loop); the table-driven ones keep what they need.
template <int MM, bool CTXC, bool MIX, bool SEG, int CM>
__global__ __launch_bounds__(LIT_THREADS) __attribute__((amdgpu_waves_per_eu(7))) void lit_decode2_kernel_w7(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    decode2_body<MM, CTXC, MIX, SEG, CM>(b, lds);
}
template <int MM, bool CTXC, bool MIX, bool SEG, int CM>
__global__ __launch_bounds__(LIT_THREADS) void lit_decode2_kernel_any(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    decode2_body<MM, CTXC, MIX, SEG, CM>(b, lds);
}


// ---- the decoder's row traffic without the decoder (launch_row_replay, lit_kernels.h) ----
// Same table layout, LDS caches, persistent grid and per-byte row selection as decode2_body for the stride-1 configurations (every
// mixing value 4); the bytes come from memory instead of from the rANS states, so the rows of byte k + 1 -- the low-nibble row
// included, which a decoder can only name once it has decoded the high nibble -- are requested while byte k is being blended.
template <bool CTXC, bool MIX, int CM>
__device__ __forceinline__ void replay_body(const LitBatch& b, uint8_t* lds) {
    constexpr int MM = 4;
    const LdsView lv = load_config_to_lds<MM, CTXC>(lds, b);
    const LitGeometry& g = b.geom;
    const int lane = threadIdx.x & 63, li = lane & 15, rbase = lane & 48, li1 = li + 1;
    const uint32_t gg = blockIdx.x * (LIT_THREADS / 16) + (threadIdx.x >> 4);
    const uint32_t G = gridDim.x * (LIT_THREADS / 16);
    Table2 tb;
    {
        const uint32_t slab = g.total_rows * 32u;
        const uint64_t base = (uint64_t)((uint8_t*)b.tables + (size_t)blockIdx.x * (LIT_THREADS / 16) * slab);
        const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        tb.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, __builtin_amdgcn_readfirstlane((int)((LIT_THREADS / 16) * slab)), 0x00020000);
        tb.lane_off = (threadIdx.x >> 4) * slab + 2u * (uint32_t)li;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr bool PERM = DIVANS_D2_PERM != 0;
    const uint32_t perm_off = lds_base + (uint32_t)(lv.mix - lv.base);
    if (PERM) {
        if (threadIdx.x < 64u) lds_write32(perm_off + 4u * threadIdx.x, b.byte_rank ? ((const uint32_t*)b.byte_rank)[threadIdx.x] : 0x03020100u + 0x04040404u * threadIdx.x);
        __syncthreads();
    }
    const uint32_t stream_base = lds_base + (threadIdx.x >> 4) * (b.cache_bytes_per_wg / (LIT_THREADS / 16));
    const Caches cc = make_caches<(CM & CM_2WAY) != 0>(b, stream_base, li);
    constexpr bool W2 = (CM & CM_2WAY) != 0;
    constexpr bool HS_C = (CM & CM_HS) != 0, HC_C = (CM & CM_HC) != 0, LS_C = (CM & CM_LS) != 0, LC_C = (CM & CM_LC) != 0;
    for (uint32_t s = gg; s < b.n_streams; s += G) {
        const uint32_t len = b.in_sizes ? b.in_sizes[s] : b.stream_len;
        const uint8_t* in = b.in + (b.in_offsets ? b.in_offsets[s] : (uint64_t)s * b.stream_len);
        init_table2<CM>(tb, cc, g.total_rows, li);
        History<false> hist;
        hist.perm_off = perm_off;
        hist.set(0ull, PERM);
        uint32_t k1 = CTXC ? 0u : lv.ctx[LIT_BLOB_LUT1CLASS + hist.p2];
        uint32_t ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, hist.p1, k1);
        // rows of the byte in hand: requested one byte ahead
        RowSlot hs_ref = {}, hc_ref = {}, ls_ref = {}, lc_ref = {};
        int hs_v = 0, hc_v = 0, ls_v = 0, lc_v = 0;
        uint32_t cur = len ? in[0] : 0u;
        {
            const RowSel rh = select_rows2<true, MM, false>(g, lv.mix, ctx_cur, hist, 0u);
            const RowSel rl = select_rows2<false, MM, false>(g, lv.mix, ctx_cur, hist, cur >> 4);
            hs_v = tb.template load<HS_C, W2, false>(cc.hs, rh.stride_row, hs_ref);
            if (MIX) hc_v = tb.template load<HC_C, W2, false>(cc.hc, rh.cm_row, hc_ref);
            ls_v = tb.template load<LS_C, W2, false>(cc.ls, rl.stride_row, ls_ref);
            if (MIX) lc_v = tb.template load<LC_C, W2, false>(cc.lc, rl.cm_row, lc_ref);
        }
        for (uint32_t base = 0; base < len; base += 16u) {
            const uint32_t cnt = len - base < 16u ? len - base : 16u;
            // the 16 bytes AFTER this group's first: byte base + 1 + lane (the first is already in `cur`)
            const uint32_t mine = (base + 1u + (uint32_t)li < len) ? in[base + 1u + (uint32_t)li] : 0u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_ds_bpermute((rbase + (int)k) << 2, (int)mine);
                const uint32_t hi = cur >> 4, lo = cur & 15u;
                // high nibble of the byte in hand
                {
                    const bool above = (uint32_t)li >= hi;
                    const int st = blend2(hs_v, li1, above, g.inc0, g.lim0, row_bcast<15>(hs_v));
                    tb.template store<HS_C>(cc.hs, hs_ref, st);
                    if (MIX) { const int cm = blend2(hc_v, li1, above, g.inc3, g.lim3, row_bcast<15>(hc_v)); tb.template store<HC_C>(cc.hc, hc_ref, cm); }
                }
                hist.push(cur, PERM);
                if (!CTXC) k1 = lv.ctx[LIT_BLOB_LUT1CLASS + hist.p2];
                ctx_cur = context_of<CTXC>(g, lv.ctx, LIT_BLOB_CTXF, hist.p1, k1);
                // every row of the next byte, its low-nibble rows included
                const RowSel rh = select_rows2<true, MM, false>(g, lv.mix, ctx_cur, hist, 0u);
                const RowSel rl = select_rows2<false, MM, false>(g, lv.mix, ctx_cur, hist, nxt >> 4);
                RowSlot nhs = {}, nhc = {}, nls = {}, nlc = {};
                const int nhs_v = tb.template load<HS_C, W2, false>(cc.hs, rh.stride_row, nhs);
                int nhc_v = 0; if (MIX) nhc_v = tb.template load<HC_C, W2, false>(cc.hc, rh.cm_row, nhc);
                // low nibble of the byte in hand
                {
                    const bool above = (uint32_t)li >= lo;
                    const int st = blend2(ls_v, li1, above, g.inc0, g.lim0, row_bcast<15>(ls_v));
                    tb.template store<LS_C>(cc.ls, ls_ref, st);
                    if (MIX) { const int cm = blend2(lc_v, li1, above, g.inc2, g.lim2, row_bcast<15>(lc_v)); tb.template store<LC_C>(cc.lc, lc_ref, cm); }
                }
                const int nls_v = tb.template load<LS_C, W2, false>(cc.ls, rl.stride_row, nls);
                int nlc_v = 0; if (MIX) nlc_v = tb.template load<LC_C, W2, false>(cc.lc, rl.cm_row, nlc);
                hs_ref = nhs; hc_ref = nhc; ls_ref = nls; lc_ref = nlc; hs_v = nhs_v; hc_v = nhc_v; ls_v = nls_v; lc_v = nlc_v;
                cur = nxt;
            }
        }
    }
}

template <bool CTXC, bool MIX, int CM>
__global__ __launch_bounds__(LIT_THREADS) __attribute__((amdgpu_waves_per_eu(7))) void row_replay_kernel(const LitBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    replay_body<CTXC, MIX, CM>(b, lds);
}

// one workgroup: histogram of a sample of the batch in LDS, then every thread ranks its own byte value
__global__ __launch_bounds__(256) void learn_byte_rank_kernel(const uint8_t* data, const uint64_t* offsets, const uint32_t* sizes, uint32_t n_streams,
                                                              uint32_t stream_len, uint32_t sample_streams, uint32_t sample_len, uint8_t* rank) {
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t j = 0; j < sample_streams; ++j) {
        const uint32_t s = (uint32_t)(((uint64_t)j * n_streams) / sample_streams);
        const uint8_t* p = data + (offsets ? offsets[s] : (uint64_t)s * stream_len);
        uint32_t len = sizes ? sizes[s] : stream_len;
        len = len < sample_len ? len : sample_len;
        for (uint32_t i = threadIdx.x; i < len; i += 256u) atomicAdd(&cnt[p[i]], 1u);
    }
    __syncthreads();
    const uint32_t mine = cnt[threadIdx.x];
    uint32_t r = 0;
    for (uint32_t x = 0; x < 256u; ++x) { const uint32_t c = cnt[x]; r += (c > mine || (c == mine && x < threadIdx.x)) ? 1u : 0u; }
    rank[threadIdx.x] = (uint8_t)r;
}

typedef void (*LitKernel)(const LitBatch);

template <bool MIX, bool SEG, int CM>
static LitKernel pick_decode2_cm(int mm, bool ctxc) {
    const int key = (mm == 4 ? 2 : (mm == 0 ? 1 : 0)) * 2 + (ctxc ? 1 : 0);
    switch (key) {
    case 0: return lit_decode2_kernel_any<-1, false, MIX, SEG, CM>; case 1: return lit_decode2_kernel_any<-1, true, MIX, SEG, CM>;
#if DIVANS_D2_W7
    case 2: return lit_decode2_kernel_w7<0, false, MIX, SEG, CM>;  case 3: return lit_decode2_kernel_w7<0, true, MIX, SEG, CM>;
    case 4: return lit_decode2_kernel_w7<4, false, MIX, SEG, CM>;  default: return lit_decode2_kernel_w7<4, true, MIX, SEG, CM>;
#else
    case 2: return lit_decode2_kernel_any<0, false, MIX, SEG, CM>;  case 3: return lit_decode2_kernel_any<0, true, MIX, SEG, CM>;
    case 4: return lit_decode2_kernel_any<4, false, MIX, SEG, CM>;  default: return lit_decode2_kernel_any<4, true, MIX, SEG, CM>;
#endif
    }
}

// The cache sets that exist as kernels: none; the high rows (stride, and context-map with mixing); the high rows plus the
// low stride rows (no mixing) / the low context-map rows (mixing).  Streams with segment lists: none or the high rows.
static uint32_t supported_cache_mask(bool mix, bool seg, uint32_t want) {
    const uint32_t high = mix ? (uint32_t)(CM_HS | CM_HC) : (uint32_t)CM_HS;
    const uint32_t full = high | (mix ? (uint32_t)CM_LC : (uint32_t)CM_LS);
    if (!seg && (want & full) == full) return full;
    if ((want & high) == high) return high;
    return 0u;
}

template <int W2>
static LitKernel pick_decode2_w(bool mix, bool seg, uint32_t cm, int mm, bool ctxc) {
    if (mix) {
        if (seg) return cm ? pick_decode2_cm<true, true, CM_HS | CM_HC | W2>(mm, ctxc) : pick_decode2_cm<true, true, 0>(mm, ctxc);
        if (cm == (uint32_t)(CM_HS | CM_HC | CM_LC)) return pick_decode2_cm<true, false, CM_HS | CM_HC | CM_LC | W2>(mm, ctxc);
        return cm ? pick_decode2_cm<true, false, CM_HS | CM_HC | W2>(mm, ctxc) : pick_decode2_cm<true, false, 0>(mm, ctxc);
    }
    if (seg) return cm ? pick_decode2_cm<false, true, CM_HS | W2>(mm, ctxc) : pick_decode2_cm<false, true, 0>(mm, ctxc);
    if (cm == (uint32_t)(CM_HS | CM_LS)) return pick_decode2_cm<false, false, CM_HS | CM_LS | W2>(mm, ctxc);
    return cm ? pick_decode2_cm<false, false, CM_HS | W2>(mm, ctxc) : pick_decode2_cm<false, false, 0>(mm, ctxc);
}
static LitKernel pick_decode2(bool mix, bool seg, uint32_t cm, bool two_way, int mm, bool ctxc) {
    return two_way ? pick_decode2_w<CM_2WAY>(mix, seg, cm, mm, ctxc) : pick_decode2_w<0>(mix, seg, cm, mm, ctxc);
}

static uint32_t wanted_cache_mask(uint32_t dm_log2) {
    uint32_t m = 0;
    for (int i = 0; i < 4; ++i) if ((dm_log2 >> (8 * i)) & 0xffu) m |= 1u << i;
    return m;
}

// dm_log2 with the caches no kernel instance implements dropped (the host sizes the LDS and the grid from this)
uint32_t lit_decode2_effective_caches(uint32_t dm_log2, bool mix, bool seg) {
    const uint32_t cm = supported_cache_mask(mix, seg, wanted_cache_mask(dm_log2));
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) if (cm & (1u << i)) out |= dm_log2 & (0xffu << (8 * i));
    return out;
}

uint32_t lit_lds_bytes2(const LitBatch& b) {
    uint32_t bytes = b.cache_bytes_per_wg;
    if (b.geom.ctx_const < 0) bytes += LIT_BLOB_CTXF + LIT_CTXF_BYTES * b.geom.n_btypes;
    if (!(b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4)) bytes += 8192u;
    if (DIVANS_D2_PERM && b.geom.mm_uniform == 4) bytes += 256u;   // the byte ranks
    return bytes;
}

uint32_t lit_decode2_stream_lds(uint32_t dm_log2) {
    uint32_t rows = 0;
    for (int i = 0; i < 4; ++i) { const uint32_t lg = (dm_log2 >> (8 * i)) & 0xffu; rows += lg ? 1u << (lg - 1u) : 0u; }
    return kRingBytes + rows * 34u;
}

// The instance launch_decode2 picks, spelt the way rocprofv3 reports it (so that a bench line and a kernel trace name the same thing)
void lit_decode2_kernel_name(const LitBatch& b, bool mix, char* buf, size_t cap) {
    const int mm = (b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4) ? b.geom.mm_uniform : -1;
    const bool seg = b.segs != nullptr, ctxc = b.geom.ctx_const >= 0;
    const uint32_t cm = wanted_cache_mask(b.dm_log2);
    const int w2 = (b.dm_shift >> 31) ? CM_2WAY : 0;
    int cmv;
    if (mix) cmv = (!seg && cm == (uint32_t)(CM_HS | CM_HC | CM_LC)) ? (CM_HS | CM_HC | CM_LC | w2) : (cm ? (CM_HS | CM_HC | w2) : 0);
    else cmv = (!seg && cm == (uint32_t)(CM_HS | CM_LS)) ? (CM_HS | CM_LS | w2) : (cm ? (CM_HS | w2) : 0);
    snprintf(buf, cap, "divans_hip::lit_decode2_kernel_%s<%d, %s, %s, %s, %d>", (mm >= 0 && DIVANS_D2_W7) ? "w7" : "any", mm,
             ctxc ? "true" : "false", mix ? "true" : "false", seg ? "true" : "false", cmv);
}

hipError_t launch_decode2(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st) {
    const int mm = (b.geom.mm_uniform == 0 || b.geom.mm_uniform == 4) ? b.geom.mm_uniform : -1;
    const uint32_t cm = wanted_cache_mask(b.dm_log2);
    if (cm != supported_cache_mask(mix, b.segs != nullptr, cm)) return hipErrorInvalidValue;   // the host passes lit_decode2_effective_caches()
    LitKernel k = pick_decode2(mix, b.segs != nullptr, cm, (b.dm_shift >> 31) != 0u, mm, b.geom.ctx_const >= 0);
    const uint32_t lds = lit_lds_bytes2(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LIT_THREADS), lds, st, b);
    return hipGetLastError();
}


hipError_t launch_learn_byte_rank(const uint8_t* data, const uint64_t* offsets, const uint32_t* sizes, uint32_t n_streams, uint32_t stream_len,
                                  uint32_t sample_streams, uint32_t sample_len, uint8_t* rank, hipStream_t st) {
    if (n_streams == 0u) return hipSuccess;
    if (sample_streams > n_streams) sample_streams = n_streams;
    hipLaunchKernelGGL(learn_byte_rank_kernel, dim3(1), dim3(256), 0, st, data, offsets, sizes, n_streams, stream_len, sample_streams, sample_len, rank);
    return hipGetLastError();
}

// the instances the benchmark configurations use: constant context without mixing (high-row cache, 2-way or direct mapped),
// context table with mixing (high stride + FirstNibble caches)
hipError_t launch_row_replay(const LitBatch& b, bool mix, uint32_t blocks, hipStream_t st) {
    if (b.geom.mm_uniform != 4 || b.segs) return hipErrorInvalidValue;
    const uint32_t cm = wanted_cache_mask(b.dm_log2);
    const bool two_way = (b.dm_shift >> 31) != 0u, ctxc = b.geom.ctx_const >= 0;
    LitKernel k = nullptr;
    if (!mix && cm == (uint32_t)CM_HS) {
        if (ctxc) k = two_way ? row_replay_kernel<true, false, CM_HS | CM_2WAY> : row_replay_kernel<true, false, CM_HS>;
        else k = two_way ? row_replay_kernel<false, false, CM_HS | CM_2WAY> : row_replay_kernel<false, false, CM_HS>;
    } else if (mix && cm == (uint32_t)(CM_HS | CM_HC)) {
        if (ctxc) k = two_way ? row_replay_kernel<true, true, CM_HS | CM_HC | CM_2WAY> : row_replay_kernel<true, true, CM_HS | CM_HC>;
        else k = two_way ? row_replay_kernel<false, true, CM_HS | CM_HC | CM_2WAY> : row_replay_kernel<false, true, CM_HS | CM_HC>;
    }
    if (!k) return hipErrorInvalidValue;
    const uint32_t lds = lit_lds_bytes2(b);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(LIT_THREADS), lds, st, b);
    return hipGetLastError();
}

}  // namespace divans_hip
