#!/usr/bin/env python3
"""Tuning aid: decode time of configs[1] over persistent-grid sizes that are NOT multiples of the CU count -- 65 536 equal streams on G
resident stream slots take ceil(65536 / G) rounds, the last one partly empty."""
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
enc = da.LiteralCodec(cfg, L)
outs = enc.alloc_encode_outputs(N, L)
enc.encode_batch(d_in, N, L, outs)
torch.cuda.synchronize(); enc.close()
d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
rows = (32, 0, 0, 0) if cfg_name == "simple" else (16, 16, 0, 0)
shifts = (31, 5, 5, 5) if cfg_name == "simple" else (5, 5, 5, 5)
for blocks in [int(x) for x in sys.argv[2:]]:
    c = da.LiteralCodec(cfg, L)
    c.set_decoder(3, rows, shifts, blocks=blocks)
    ms = []
    for _ in range(2):
        c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
        torch.cuda.synchronize(); ms.append(c.info().last_decode_ms)
    ok = bool(torch.equal(d_back, d_in))
    print(f"{cfg_name} blocks {blocks} ({blocks * 16} slots, {N / (blocks * 16):.2f} rounds): decode {min(ms):.1f} ms ok={ok}", flush=True)
    c.close()
