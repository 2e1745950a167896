"""Offline simulation of LDS row-cache organisations on the benchmark workload (design aid, not product)."""
import sys
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(_R, 'tests')); sys.path.insert(0, os.path.join(_R, 'oracle'))
import numpy as np, workload, pyoracle as po, ctypes

c = workload.load_corpus()
blocks = workload.make_blocks(c, 11, 3)

def rows_simple(b):
    prev = np.concatenate([[0], b[:-1]]).astype(np.int64); hi = (b >> 4).astype(np.int64)
    seq = np.empty(2 * b.size, dtype=np.int64); seq[0::2] = prev; seq[1::2] = 256 + prev * 16 + hi
    return seq

def rows_mixing(b):
    L = po.lib(); l0 = (ctypes.c_uint8 * 256)(); l1 = (ctypes.c_uint8 * 256)()
    L.orc_get_lut0(2, l0); L.orc_get_lut1(2, l1)
    l0 = np.array(l0, dtype=np.int64); l1 = np.array(l1, dtype=np.int64)
    prev = np.concatenate([[0], b[:-1]]).astype(np.int64); pp = np.concatenate([[0, 0], b[:-2]]).astype(np.int64)
    ctx = (l0[prev] | l1[pp]) & 63; hi = (b >> 4).astype(np.int64)
    high = prev * 64 + ctx; low = 16384 + prev * 16 + hi; cmf = 20480 + ctx; cms = 20480 + 64 + hi + 16 * ctx
    seq = np.empty(4 * b.size, dtype=np.int64); seq[0::4] = high; seq[1::4] = cmf; seq[2::4] = low; seq[3::4] = cms
    return seq

def dm(seq, S, h):
    tags = np.full(S, -1, dtype=np.int64); hit = 0
    for r in seq:
        s = h(r) % S
        if tags[s] == r: hit += 1
        else: tags[s] = r
    return hit / len(seq)

def way2(seq, S, h):
    sets = S // 2; t0 = [-1] * sets; t1 = [-1] * sets; hit = 0   # t0 = MRU
    for r in seq:
        s = h(r) % sets
        if t0[s] == r: hit += 1
        elif t1[s] == r: hit += 1; t0[s], t1[s] = t1[s], t0[s]
        else: t1[s] = t0[s]; t0[s] = r
    return hit / len(seq)

h_id = lambda r: int(r)
h_mul = lambda r: (int(r) * 2654435761 & 0xffffffff) >> 12
h_xor = lambda r: int(r) ^ (int(r) >> 4) ^ (int(r) >> 9)
for name, fn in (("simple", rows_simple), ("mixing", rows_mixing)):
    for bi in range(1):
        seq = fn(blocks[bi]).tolist()
        print(name, "distinct rows", len(set(seq)))
        for S in (128, 256, 512, 1024):
            print(f"  S={S:5d} ({S*32//1024:3d} KB): DM id {dm(seq,S,h_id)*100:5.1f}  DM mul {dm(seq,S,h_mul)*100:5.1f}  DM xor {dm(seq,S,h_xor)*100:5.1f}  2way mul {way2(seq,S,h_mul)*100:5.1f} 2way xor {way2(seq,S,h_xor)*100:5.1f}")
