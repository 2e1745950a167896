// sim_hierarchy.c -- design aid (NOT product, NOT oracle): what a table LAYOUT of the literal decoder costs at the L2 -> fabric
// interface.  One XCD's share of a decode launch is replayed: R resident streams (28 672 / 8 = 3584 at seven workgroups per CU)
// step through their own 64 KiB blocks byte by byte in lock step; every row access goes through the stream's LDS row caches as
// lit_decode2.hip organises them (direct mapped or 2-way, write back), what misses goes to ONE shared 4 MiB 16-way L2 of
// 128-byte lines with 32-byte dirty sectors (write-allocate WITHOUT fill: profiles/r05_counter_calibration.txt -- a row store
// leaves as a 32-byte request, two rows of a 64-byte half as one 64-byte request, a read miss is one 128-byte fill), and behind it
// a 32 MiB share of the Infinity Cache (memory side, 128-byte lines).  Reported per decoded byte: TCC_EA0_RDREQ (fills),
// TCC_EA0_WRREQ (write-backs, 32 / 64 B), L2 hit rate, LDS hit rates, HBM lines behind the Infinity Cache.
//
// The point is the comparison between layouts -- which rows share a line -- on the decoders' real access streams; the model is
// first checked against the counters of the deployed layout (profiles/r05_simple_summary.txt, r05_mixing_summary.txt).
//
// usage: sim_hierarchy <blocks.bin> <n_streams> <stream_len> <config: 0 plain, 1 mixing> <layout> [key=value ...]
//   keys: l2_kb=4096 l2_ways=16 mall_kb=32768 hs_rows=32 hs_ways=2 hc_rows=16 ls_rows=0 ls_ways=1 lazy_init=0 cm_layout=0 wt=0 ctxf=<file>
//   wt=1: the L2 cleans a dirty sector at once (what gfx950's does under the decoders' footprint, scripts/ubench/wb_policy.hip)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t sets, ways;
    uint32_t* tag;    // [sets][ways], MRU first; 0xffffffff = empty
    uint8_t* meta;    // bit 7: whole line valid (filled); bits 0..3: dirty 32-byte sectors
} Cache;

static void cache_init(Cache* c, uint64_t bytes, uint32_t ways) {
    c->ways = ways; c->sets = (uint32_t)(bytes / 128u / ways);
    c->tag = (uint32_t*)malloc(sizeof(uint32_t) * c->sets * ways);
    c->meta = (uint8_t*)calloc((size_t)c->sets * ways, 1);
    memset(c->tag, 0xff, sizeof(uint32_t) * c->sets * ways);
}
static inline uint32_t set_of(const Cache* c, uint32_t line) {
    uint64_t h = (uint64_t)line * 0x9E3779B97F4A7C15ull;     // the hardware hashes channel / set bits; any even spread will do
    return (uint32_t)((h >> 24) % c->sets);
}
// returns the way index after moving the line to MRU, or -1; *ev_tag / *ev_meta = what was evicted when `alloc`
static inline int cache_find(Cache* c, uint32_t set, uint32_t line) {
    uint32_t* t = c->tag + (size_t)set * c->ways;
    for (uint32_t w = 0; w < c->ways; ++w) if (t[w] == line) return (int)w;
    return -1;
}
static inline void cache_touch(Cache* c, uint32_t set, int w) {
    uint32_t* t = c->tag + (size_t)set * c->ways; uint8_t* m = c->meta + (size_t)set * c->ways;
    const uint32_t tt = t[w]; const uint8_t mm = m[w];
    memmove(t + 1, t, sizeof(uint32_t) * (size_t)w); memmove(m + 1, m, (size_t)w);
    t[0] = tt; m[0] = mm;
}
static inline void cache_insert(Cache* c, uint32_t set, uint32_t line, uint8_t meta, uint32_t* ev_tag, uint8_t* ev_meta) {
    uint32_t* t = c->tag + (size_t)set * c->ways; uint8_t* m = c->meta + (size_t)set * c->ways;
    *ev_tag = t[c->ways - 1]; *ev_meta = m[c->ways - 1];
    memmove(t + 1, t, sizeof(uint32_t) * (c->ways - 1)); memmove(m + 1, m, c->ways - 1);
    t[0] = line; m[0] = meta;
}

static Cache L2, MALL;
static uint64_t n_req, n_hit, n_rdreq, n_wr32, n_wr64, n_hbm_rd, n_hbm_wr, n_mall_rd, n_mall_rdhit;
static int g_write_through;   // wt=1: a store leaves for the fabric at once (the line stays, clean)

static void mall_access(uint32_t line, int write) {
    const uint32_t set = set_of(&MALL, line ^ 0x5bd1e995u);
    int w = cache_find(&MALL, set, line);
    if (!write) { ++n_mall_rd; if (w >= 0) ++n_mall_rdhit; }
    if (w >= 0) { cache_touch(&MALL, set, w); if (write) MALL.meta[(size_t)set * MALL.ways] |= 1; return; }
    if (!write) ++n_hbm_rd;
    uint32_t et; uint8_t em;
    cache_insert(&MALL, set, line, write ? 1 : 0, &et, &em);
    if (et != 0xffffffffu && (em & 1)) ++n_hbm_wr;
}
static void l2_evict(uint32_t et, uint8_t em) {
    if (et == 0xffffffffu) return;
    const uint32_t d = em & 15u;
    if (!d) return;
    for (int half = 0; half < 2; ++half) {
        const uint32_t b = (d >> (2 * half)) & 3u;
        if (b == 3u) ++n_wr64; else if (b) ++n_wr32;
        if (b) mall_access(et, 1);
    }
}
static void l2_read(uint32_t line, uint32_t sector) {
    ++n_req;
    const uint32_t set = set_of(&L2, line);
    int w = cache_find(&L2, set, line);
    if (w >= 0) {
        cache_touch(&L2, set, w);
        uint8_t* m = L2.meta + (size_t)set * L2.ways;
        if ((m[0] & 0x80u) || (m[0] & (1u << sector))) { ++n_hit; return; }
        ++n_rdreq; mall_access(line, 0); m[0] |= 0x80u; return;
    }
    ++n_rdreq; mall_access(line, 0);
    uint32_t et; uint8_t em;
    cache_insert(&L2, set, line, 0x80u, &et, &em);
    l2_evict(et, em);
}
static void l2_write(uint32_t line, uint32_t sector_mask) {
    ++n_req;
    const uint32_t set = set_of(&L2, line);
    int w = cache_find(&L2, set, line);
    if (g_write_through) {
        for (int half = 0; half < 2; ++half) { const uint32_t b = (sector_mask >> (2 * half)) & 3u; if (b == 3u) ++n_wr64; else if (b) ++n_wr32; if (b) mall_access(line, 1); }
        if (w >= 0) { ++n_hit; cache_touch(&L2, set, w); return; }
        uint32_t et0; uint8_t em0;
        cache_insert(&L2, set, line, 0, &et0, &em0);    // allocated with nothing valid (sector validity without dirtiness is not tracked: a later read of the line fills)
        return;
    }
    if (w >= 0) { ++n_hit; cache_touch(&L2, set, w); L2.meta[(size_t)set * L2.ways] |= (uint8_t)sector_mask; return; }
    uint32_t et; uint8_t em;
    cache_insert(&L2, set, line, (uint8_t)sector_mask, &et, &em);
    l2_evict(et, em);
}

// ---- per-stream LDS row cache as lit_decode2.hip has it: keyed by the physical row, write back ----
typedef struct { uint32_t rows, ways, shift; int32_t* tag; uint8_t* mru; uint64_t acc, hit; } Lds;
static void lds_init(Lds* c, uint32_t streams, uint32_t rows, uint32_t ways, uint32_t shift) {
    c->rows = rows; c->ways = ways; c->shift = shift; c->acc = c->hit = 0;
    if (!rows) return;
    c->tag = (int32_t*)malloc(sizeof(int32_t) * (size_t)streams * rows);
    memset(c->tag, 0xff, sizeof(int32_t) * (size_t)streams * rows);
    c->mru = (uint8_t*)calloc((size_t)streams * rows, 1);
}
// returns 1 on hit; on a miss *victim = the row written back (or -1)
static int lds_access(Lds* c, uint32_t stream, uint32_t row, int32_t* victim) {
    ++c->acc; *victim = -1;
    if (!c->rows) return 0;
    int32_t* t = c->tag + (size_t)stream * c->rows;
    if (c->ways == 1) {
        const uint32_t s = (row ^ (c->shift < 32 ? row >> c->shift : 0)) & (c->rows - 1);
        if (t[s] == (int32_t)row) { ++c->hit; return 1; }
        *victim = t[s]; t[s] = (int32_t)row; return 0;
    }
    const uint32_t sets = c->rows / 2, s = (row ^ (c->shift < 32 ? row >> c->shift : 0)) & (sets - 1);
    uint8_t* m = c->mru + (size_t)stream * c->rows;
    if (t[2 * s] == (int32_t)row) { ++c->hit; m[s] = 0; return 1; }
    if (t[2 * s + 1] == (int32_t)row) { ++c->hit; m[s] = 1; return 1; }
    const uint32_t way = m[s] ^ 1u;
    *victim = t[2 * s + way]; t[2 * s + way] = (int32_t)row; m[s] = (uint8_t)way; return 0;
}

// ---- layouts ----
static const char kOrder[] = " etaoinshrdlcumwfgypbvkjxqz\n,.;'\"-!?:()TAISOWHBCMNEPDLFRGYUVKJQXZ0123456789";
static uint8_t g_rank[256];
static void make_rank(void) {
    int used[256] = {0}; uint32_t n = 0;
    for (size_t i = 0; i + 1 < sizeof(kOrder); ++i) { uint8_t b = (uint8_t)kOrder[i]; if (!used[b]) { used[b] = 1; g_rank[b] = (uint8_t)n++; } }
    for (int b = 0; b < 256; ++b) if (!used[b]) g_rank[b] = (uint8_t)n++;
}

enum { LAY_HI_RANK = 0, LAY_HI_NUM = 1, LAY_PREV_HI = 2, LAY_HI_ARANK = 3, LAY_ARANK_AHI = 4, LAY_DENSE = 5, LAY_LINE_DENSE = 6, LAY_COLOC = 7,
       LAY_RANK_AHI = 8, LAY_NUM_AHI = 9, LAY_ORACLE_DENSE = 10, LAY_ORACLE_RANK = 11 };
static const char* kLayName[] = {"[hi][rank(prev)] static text rank (deployed)", "[hi][prev] numeric", "[prev][hi] numeric",
    "[hi][first-touch rank(prev)]", "[first-touch rank(prev)][first-touch pos(hi)]", "dense first-touch rows", "first-touch lines per prev (4 hi each)",
    "unit per prev: high row + 16 low rows, first-touch order", "[static rank(prev)][first-touch pos(hi)]", "[prev numeric][first-touch pos(hi)]",
    "BOUND: low rows dense in the order of the stream's own access counts (two passes: not implementable in a decoder)",
    "BOUND: [hi][rank(prev)] with the rank from the stream's own byte counts"};

typedef struct {
    uint8_t arank[256]; uint16_t n_prev;             // first-touch rank of a previous byte (0xff = none yet)
    uint8_t pos[256][16]; uint8_t npos[256];         // first-touch position of a high nibble under a previous byte
    uint16_t slot[256][16]; uint32_t n_slots;        // dense: row slot of (prev, hi)
    uint16_t pline[256][4];                          // line-dense: the lines a previous byte owns
    uint32_t n_lines;
} Adapt;

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s blocks.bin n_streams stream_len config layout [key=value...]\n", argv[0]); return 2; }
    const uint32_t R = (uint32_t)atoi(argv[2]), L = (uint32_t)atoi(argv[3]);
    const int mixing = atoi(argv[4]), layout = atoi(argv[5]);
    uint32_t l2_kb = 4096, l2_ways = 16, mall_kb = 32768, hs_rows = mixing ? 16 : 32, hs_ways = mixing ? 1 : 2, hc_rows = mixing ? 16 : 0, lazy_init = 0;
    uint32_t ls_rows = 0, ls_ways = 1, cm_layout = 0;
    const char* ctxf_path = NULL;
    for (int i = 6; i < argc; ++i) {
        char* eq = strchr(argv[i], '='); if (!eq) continue; *eq = 0; const char* k = argv[i]; const char* v = eq + 1;
        if (!strcmp(k, "l2_kb")) l2_kb = (uint32_t)atoi(v); else if (!strcmp(k, "l2_ways")) l2_ways = (uint32_t)atoi(v);
        else if (!strcmp(k, "mall_kb")) mall_kb = (uint32_t)atoi(v); else if (!strcmp(k, "hs_rows")) hs_rows = (uint32_t)atoi(v);
        else if (!strcmp(k, "hs_ways")) hs_ways = (uint32_t)atoi(v); else if (!strcmp(k, "hc_rows")) hc_rows = (uint32_t)atoi(v);
        else if (!strcmp(k, "lazy_init")) lazy_init = (uint32_t)atoi(v); else if (!strcmp(k, "ls_rows")) ls_rows = (uint32_t)atoi(v);
        else if (!strcmp(k, "ls_ways")) ls_ways = (uint32_t)atoi(v); else if (!strcmp(k, "ctxf")) ctxf_path = v;
        else if (!strcmp(k, "cm_layout")) cm_layout = (uint32_t)atoi(v);
        else if (!strcmp(k, "wt")) g_write_through = atoi(v);
    }
    make_rank();
    uint8_t* data = (uint8_t*)malloc((size_t)R * L);
    FILE* f = fopen(argv[1], "rb"); if (!f || fread(data, 1, (size_t)R * L, f) != (size_t)R * L) { fprintf(stderr, "short read\n"); return 1; } fclose(f);
    // mixing: ctxf[prev][class] = ctx | slot << 8 (u16) and lut1class[256], written by sim_hierarchy.py from the oracle's tables
    uint16_t ctxf[256][4]; uint8_t lut1class[256]; uint32_t hs_classes = 1, nctx = 1;
    memset(ctxf, 0, sizeof(ctxf)); memset(lut1class, 0, sizeof(lut1class));
    if (mixing) {
        if (!ctxf_path) { fprintf(stderr, "mixing needs ctxf=<file>\n"); return 2; }
        f = fopen(ctxf_path, "rb");
        if (!f || fread(ctxf, 1, sizeof(ctxf), f) != sizeof(ctxf) || fread(lut1class, 1, 256, f) != 256) { fprintf(stderr, "bad ctxf\n"); return 1; }
        fclose(f);
        for (int p = 0; p < 256; ++p) for (int k = 0; k < 4; ++k) { if ((uint32_t)(ctxf[p][k] >> 8) + 1 > hs_classes) hs_classes = (ctxf[p][k] >> 8) + 1; if ((uint32_t)(ctxf[p][k] & 0xff) + 1 > nctx) nctx = (ctxf[p][k] & 0xff) + 1; }
    }
    // table geometry in rows: [high stride][low stride][FirstNibble][SecondNibble]
    const uint32_t high_rows = 256 * hs_classes;
    uint32_t low_rows = 4096;
    const uint32_t unit_rows = 17;                     // LAY_COLOC
    uint32_t low_base = high_rows;
    if (layout == LAY_COLOC) { low_base = 0; low_rows = 0; }
    const uint32_t cm_base = (layout == LAY_COLOC ? 256 * unit_rows + (high_rows - 256) : high_rows + low_rows);
    const uint32_t cm_rows = mixing ? nctx + 16 * nctx : 0;
    uint32_t total_rows = cm_base + cm_rows;
    total_rows = (total_rows + 3u) & ~3u;
    const uint32_t slab_lines = total_rows / 4;
    cache_init(&L2, (uint64_t)l2_kb * 1024u, l2_ways);
    cache_init(&MALL, (uint64_t)mall_kb * 1024u, 16);
    Lds hs, hc, ls;
    lds_init(&hs, R, hs_rows, hs_ways, mixing ? 5 : 31);
    lds_init(&hc, R, hc_rows, 1, 5);
    lds_init(&ls, R, ls_rows, ls_ways, 4);
    Adapt* ad = (Adapt*)calloc(R, sizeof(Adapt));
    for (uint32_t s = 0; s < R; ++s) { memset(ad[s].arank, 0xff, 256); memset(ad[s].pos, 0xff, sizeof(ad[s].pos)); memset(ad[s].slot, 0xff, sizeof(ad[s].slot)); memset(ad[s].pline, 0xff, sizeof(ad[s].pline)); }
    if (layout == LAY_ORACLE_DENSE || layout == LAY_ORACLE_RANK) {
        for (uint32_t s = 0; s < R; ++s) {
            const uint8_t* b = data + (size_t)s * L;
            static uint32_t cnt[4096]; static uint32_t pc[256];
            memset(cnt, 0, sizeof(cnt)); memset(pc, 0, sizeof(pc));
            for (uint32_t t = 0; t < L; ++t) { const uint32_t prev = t ? b[t - 1] : 0; ++cnt[prev * 16 + (b[t] >> 4)]; ++pc[prev]; }
            // selection sort by count (4096 / 256 keys: cheap enough)
            if (layout == LAY_ORACLE_DENSE) {
                static uint8_t done[4096]; memset(done, 0, sizeof(done));
                for (uint32_t k = 0; k < 4096; ++k) {
                    uint32_t best = 0, bi = 0xffffffffu;
                    for (uint32_t i = 0; i < 4096; ++i) if (!done[i] && (bi == 0xffffffffu || cnt[i] > best)) { best = cnt[i]; bi = i; }
                    if (best == 0) break;
                    done[bi] = 1; ad[s].slot[bi >> 4][bi & 15] = (uint16_t)k; ad[s].n_slots = k + 1;
                }
            }
            uint8_t pdone[256]; memset(pdone, 0, 256);
            for (uint32_t k = 0; k < 256; ++k) {
                uint32_t best = 0, bi = 0xffffffffu;
                for (uint32_t i = 0; i < 256; ++i) if (!pdone[i] && (bi == 0xffffffffu || pc[i] > best)) { best = pc[i]; bi = i; }
                pdone[bi] = 1; ad[s].arank[bi] = (uint8_t)k;
            }
            ad[s].n_prev = 256;
        }
    }
    uint64_t init_wr = 0, touched_lines = 0;
    uint8_t* line_touched = (uint8_t*)calloc((size_t)R * slab_lines, 1);

#define ROW_RMW(stream, row, via_lds) do { \
        const uint32_t r_ = (row); const uint32_t ln_ = (stream) * slab_lines + (r_ >> 2); \
        if (!line_touched[ln_]) { line_touched[ln_] = 1; ++touched_lines; } \
        int32_t vict_ = -1; \
        if ((via_lds) && lds_access((via_lds), (stream), r_, &vict_)) break; \
        if ((via_lds) && (via_lds)->rows) { \
            if (vict_ >= 0) l2_write((stream) * slab_lines + ((uint32_t)vict_ >> 2), 1u << ((uint32_t)vict_ & 3u)); \
            l2_read(ln_, r_ & 3u); \
        } else { l2_read(ln_, r_ & 3u); l2_write(ln_, 1u << (r_ & 3u)); } \
    } while (0)

    for (uint32_t t = 0; t < L; ++t) {
        for (uint32_t s = 0; s < R; ++s) {
            const uint8_t* b = data + (size_t)s * L;
            Adapt* a = &ad[s];
            if (t == 0 && !lazy_init) {
                // init_table2: every row of the slab, 16-byte stores -- whole 64-byte halves, one request each
                for (uint32_t ln = 0; ln < slab_lines; ++ln) { l2_write(s * slab_lines + ln, 15u); ++init_wr; }
            }
            const uint32_t prev = t ? b[t - 1] : 0, pp = t > 1 ? b[t - 2] : 0, cur = b[t], hi = cur >> 4;
            uint32_t ctx = 0, hslot = 0;
            if (mixing) { const uint16_t v = ctxf[prev][lut1class[pp] & 3]; ctx = v & 0xff; hslot = v >> 8; }
            // --- physical rows of this byte under the layout ---
            uint32_t prow;     // index of prev among the 256 rows of a [.][prev] table
            switch (layout) {
            case LAY_HI_RANK: case LAY_RANK_AHI: prow = g_rank[prev]; break;
            case LAY_HI_ARANK: case LAY_ARANK_AHI: case LAY_COLOC: case LAY_DENSE: case LAY_LINE_DENSE: case LAY_ORACLE_DENSE: case LAY_ORACLE_RANK:
                if (a->arank[prev] == 0xff) a->arank[prev] = (uint8_t)a->n_prev++;
                prow = a->arank[prev]; break;
            default: prow = prev; break;
            }
            uint32_t hrow, lrow;
            hrow = hslot * 256 + prow;
            uint32_t hpos = hi;
            if (layout == LAY_ARANK_AHI || layout == LAY_COLOC || layout == LAY_RANK_AHI || layout == LAY_NUM_AHI) {
                if (a->pos[prev][hi] == 0xff) a->pos[prev][hi] = a->npos[prev]++;
                hpos = a->pos[prev][hi];
            }
            switch (layout) {
            case LAY_HI_RANK: case LAY_HI_NUM: case LAY_HI_ARANK: case LAY_ORACLE_RANK: lrow = low_base + hi * 256 + prow; break;
            case LAY_PREV_HI: lrow = low_base + prev * 16 + hi; break;
            case LAY_ARANK_AHI: case LAY_RANK_AHI: case LAY_NUM_AHI: lrow = low_base + prow * 16 + hpos; break;
            case LAY_DENSE: case LAY_ORACLE_DENSE:
                if (a->slot[prev][hi] == 0xffff) a->slot[prev][hi] = (uint16_t)a->n_slots++;
                lrow = low_base + a->slot[prev][hi]; break;
            case LAY_LINE_DENSE: {
                if (a->pos[prev][hi] == 0xff) a->pos[prev][hi] = a->npos[prev]++;
                const uint32_t q = a->pos[prev][hi];
                if (a->pline[prev][q >> 2] == 0xffff) a->pline[prev][q >> 2] = (uint16_t)a->n_lines++;
                lrow = low_base + a->pline[prev][q >> 2] * 4 + (q & 3); break;
            }
            default: /* LAY_COLOC */
                hrow = hslot ? 256 * unit_rows + (hslot - 1) * 256 + prow : prow * unit_rows;
                lrow = prow * unit_rows + 1 + hpos; break;
            }
            // lazy_init=1 (layouts with a directory: a row is filled with the default CDF when it is first touched) only drops the table fill at
            // t = 0; the first touch itself is modelled as the read-modify-write below (a slight over-count of fills)
            ROW_RMW(s, hrow, &hs);
            if (mixing) ROW_RMW(s, cm_base + ctx, &hc);
            if (ls_rows) ROW_RMW(s, lrow, &ls); else ROW_RMW(s, lrow, (Lds*)0);
            if (mixing) {
                const uint32_t crow = cm_layout == 0 ? cm_base + nctx + hi + 16 * ctx : cm_base + nctx + ctx + nctx * hi;
                ROW_RMW(s, crow, (Lds*)0);
            }
            // decoded bytes leave 16 at a time (non-temporal): one 64-byte request per 64 bytes; coded words come in as 64-byte reads
            if ((t & 63u) == 63u) { ++n_wr64; }
            if ((t & 127u) == 0u) { ++n_rdreq; }
        }
    }
    const double bytes = (double)R * L;
    printf("layout %d %s | config %s | R %u L %u l2 %u KB mall %u KB hs %u x%u hc %u ls %u lazy %u\n", layout, kLayName[layout], mixing ? "mixing" : "plain", R, L, l2_kb, mall_kb, hs_rows, hs_ways, hc_rows, ls_rows, lazy_init);
    printf("  rows/stream %u (%u KB), lines touched/stream %.1f\n", total_rows, total_rows * 32 / 1024, (double)touched_lines / R);
    printf("  LDS hits: high stride %.1f %%", hs.acc ? 100.0 * hs.hit / hs.acc : 0.0);
    if (mixing) printf(", high cm %.1f %%", hc.acc ? 100.0 * hc.hit / hc.acc : 0.0);
    if (ls_rows) printf(", low stride %.1f %%", ls.acc ? 100.0 * ls.hit / ls.acc : 0.0);
    printf("\n  L2: req/byte %.3f hit %.1f %% | RDREQ/byte %.3f | WRREQ/byte %.3f (32 B %.3f, 64 B %.3f) | fabric req/byte %.3f | fabric bytes/byte %.1f\n",
           n_req / bytes, 100.0 * n_hit / n_req, n_rdreq / bytes, (n_wr32 + n_wr64) / bytes, n_wr32 / bytes, n_wr64 / bytes,
           (n_rdreq + n_wr32 + n_wr64) / bytes, (128.0 * n_rdreq + 32.0 * n_wr32 + 64.0 * n_wr64) / bytes);
    printf("  Infinity Cache: read hit %.1f %% | HBM line reads/byte %.3f writes/byte %.3f\n", n_mall_rd ? 100.0 * n_mall_rdhit / n_mall_rd : 0.0, n_hbm_rd / bytes, n_hbm_wr / bytes);
    printf("  RESULT layout=%d config=%d rdreq=%.4f wrreq=%.4f fabric=%.4f l2hit=%.4f hs=%.4f hbm_rd=%.4f hbm_wr=%.4f\n", layout, mixing, n_rdreq / bytes, (n_wr32 + n_wr64) / bytes,
           (n_rdreq + n_wr32 + n_wr64) / bytes, (double)n_hit / n_req, hs.acc ? (double)hs.hit / hs.acc : 0.0, n_hbm_rd / bytes, n_hbm_wr / bytes);
    return 0;
}
