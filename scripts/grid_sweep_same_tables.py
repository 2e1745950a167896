#!/usr/bin/env python3
"""Measurement aid (round 5): decode time against the size of the persistent grid ON ONE ALLOCATION of the tables (they only grow, so the largest
grid goes first and the smaller ones reuse its memory: no placement noise between the lines).  usage: grid_sweep_same_tables.py [simple|mixing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import divans_amd as da
import workload
from bench import device_blocks
dev = torch.device("cuda", 0)
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "simple"
N, L = 65536, 65536
corpus = workload.load_corpus()
d_in = device_blocks(torch, torch.from_numpy(corpus).to(dev), 0, N, L)
cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
c = da.LiteralCodec(cfg, L)
c.tune_tables(1)
outs = c.alloc_encode_outputs(N, L)
c.encode_batch(d_in, N, L, outs)
torch.cuda.synchronize()
d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
gen = 3 if cfg_name == "simple" else 2
for blocks in (2048, 1792, 2048, 1920, 1792, 1664, 1536, 1792):
    c.set_decoder(gen, None, None, blocks=blocks)
    ms = []
    for _ in range(2):
        c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
        torch.cuda.synchronize(); ms.append(c.info().last_decode_ms)
    print(f"{cfg_name} blocks {blocks:5d} ({blocks / 256:.2f} per CU): decode {min(ms):7.2f} ms  resident {c.info().resident_groups} ok={bool(torch.equal(d_back, d_in))}", flush=True)
