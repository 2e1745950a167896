#!/usr/bin/env python3
"""Measurement aid (round 5): decode kernel time with the stride-1 tables laid out by BytePerm's text-frequency rank (0) and numerically (1),
on the benchmark text and on testdata/random_then_unicode -- ONE codec, one placement of its tables, the order alternating launch by launch."""
import lzma, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import divans_amd as da
import workload
from bench import device_blocks
dev = torch.device("cuda", 0)
N, L = 65536, 65536
corpus = torch.from_numpy(workload.load_corpus()).to(dev)
with lzma.open(os.path.join(ROOT, "tests", "golden", "random_then_unicode.xz")) as f:
    rtu = torch.from_numpy(np.frombuffer(f.read(), dtype=np.uint8).copy()).to(dev)
for name, src in (("text (alice29||asyoulik)", corpus), ("random_then_unicode", rtu)):
    d_in = device_blocks(torch, src, 0, N, L)
    c = da.LiteralCodec(da.config_simple(), L)
    c.tune_tables(1)
    outs = c.alloc_encode_outputs(N, L)
    c.encode_batch(d_in, N, L, outs)
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    line = []
    for order in (0, 1, 0, 1, 0, 1):
        c.set_byte_order(order)
        c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
        torch.cuda.synchronize()
        line.append(f"order {order}: {c.info().last_decode_ms:.1f} ms")
    print(f"{name}: " + ", ".join(line) + f"  round trip {bool(torch.equal(d_back, d_in))}", flush=True)
    c.close(); del outs, d_back, d_in
    torch.cuda.empty_cache()
