"""Offline simulation for VERDICT r04 item 1a (design aid, not product): would 16-byte "compact" rows -- sixteen 8-bit symbol counts, valid until a
row's first renormalisation (frequentist_cdf.rs:74-85: before it cdf[i] = 4 (i + 1) + inc * #{coded symbols <= i}) -- buy the decoders' LDS row caches
more hits per byte than today's 32-byte rows?

Per 64 KiB stream of the benchmark workloads, TestSimple rows (high row = [prev], low row = [prev][high nibble]), reference speed MUD (inc 16,
lim 8192: a row renormalises at its 508th update):
  * how the accesses split between rows that are still compact-representable (no renormalisation yet, every symbol count <= 255) and promoted rows;
  * LRU hit rates of an LDS budget spent on full rows only (34 B per row with its tag) against the same bytes split between full rows and
    compact rows (18 B); a compact row that stops being representable moves to the full tier.
usage: python tests/tools/sim_compact_rows.py"""
import collections
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_R, "tests"))
import lzma

import numpy as np
import workload

INC, LIM = 16, 8192


def accesses(block):
    """(row id, symbol) per nibble, in coding order; low rows are offset by 256"""
    out = []
    prev = 0
    for b in block.tolist():
        hi, lo = b >> 4, b & 15
        out.append((prev, hi))
        out.append((256 + prev * 16 + hi, lo))
        prev = b
    return out


class Row:
    __slots__ = ("counts", "n", "compact")

    def __init__(self):
        self.counts = [0] * 16
        self.n = 0
        self.compact = True

    def update(self, sym):
        self.n += 1
        self.counts[sym] += 1
        if 64 + INC * self.n >= LIM or self.counts[sym] > 255:       # first renormalisation, or a count that leaves 8 bits
            self.compact = False


def split(acc):
    rows = collections.defaultdict(Row)
    comp = 0
    for r, s in acc:
        row = rows[r]
        comp += row.compact
        row.update(s)
    return comp / len(acc), len(rows), sum(1 for v in rows.values() if not v.compact)


def lru_full(acc, slots, which):
    """hit rate of an LRU of `slots` full rows over the accesses of one nibble half (which = 0 high, 1 low)"""
    lru = collections.OrderedDict()
    hit = tot = 0
    for r, s in acc:
        if (r >= 256) != bool(which):
            continue
        tot += 1
        if r in lru:
            hit += 1
            lru.move_to_end(r)
        else:
            lru[r] = 1
            if len(lru) > slots:
                lru.popitem(last=False)
    return hit / tot


def lru_two_tier(acc, full_slots, compact_slots, which):
    rows = collections.defaultdict(Row)
    full, comp = collections.OrderedDict(), collections.OrderedDict()
    hit = tot = 0
    for r, s in acc:
        if (r >= 256) != bool(which):
            continue
        tot += 1
        row = rows[r]
        if r in full:
            hit += 1
            full.move_to_end(r)
        elif r in comp:
            hit += 1
            comp.move_to_end(r)
        else:
            (comp if row.compact and compact_slots else full)[r] = 1
        row.update(s)
        if not row.compact and r in comp:       # promoted: from now on it needs a full slot
            del comp[r]
            full[r] = 1
        if len(full) > full_slots:
            full.popitem(last=False)
        if len(comp) > compact_slots:
            comp.popitem(last=False)
    return hit / tot


def main():
    corpus = workload.load_corpus()
    text = workload.make_blocks(corpus, 11, 2)
    with lzma.open(os.path.join(workload.GOLDEN, "random_then_unicode.xz")) as f:
        rtu = np.frombuffer(f.read(), dtype=np.uint8)
    streams = [("text block 11", text[0]), ("text block 12", text[1]), ("random_then_unicode[0:64K] (random bytes)", rtu[:65536]),
               ("random_then_unicode[192K:256K] (multi-script UTF-8)", rtu[196608:262144])]
    for name, blk in streams:
        acc = accesses(blk)
        frac, nrows, promoted = split(acc)
        print(f"== {name}: {nrows} rows touched, {promoted} of them leave the compact form; {100 * frac:.1f} % of all row accesses find their row still compact")
        for which, half in ((0, "high rows"), (1, "low rows")):
            for budget_rows in (32, 64):
                budget = budget_rows * 34
                line = f"   {half}, LDS budget {budget} B: {budget_rows} full rows {100 * lru_full(acc, budget_rows, which):5.1f} % hits"
                for full_slots in (budget_rows * 3 // 4, budget_rows // 2, budget_rows // 4):
                    compact_slots = (budget - full_slots * 34) // 18
                    line += f" | {full_slots} full + {compact_slots} compact {100 * lru_two_tier(acc, full_slots, compact_slots, which):5.1f} %"
                print(line)


if __name__ == "__main__":
    main()
