#!/usr/bin/env python3
"""Measurement aid (round 5): do LOW-row caches pay for batches small enough to be latency-bound (a stream's dependency chain, not VALU issue or
the fabric)?  Decodes n streams of one coded batch under several cache geometries of lit_decode2 (rows of the high stride / high context-map /
low stride / low context-map caches).  usage: small_batch_low_cache.py [simple|mixing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import divans_amd as da
import workload
from bench import device_blocks
dev = torch.device("cuda", 0)
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "simple"
N, L = 16384, 65536
corpus = workload.load_corpus()
d_in = device_blocks(torch, torch.from_numpy(corpus).to(dev), 0, N, L)
cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
enc = da.LiteralCodec(cfg, L)
outs = enc.alloc_encode_outputs(N, L)
enc.encode_batch(d_in, N, L, outs)
torch.cuda.synchronize(); enc.close()
d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
cus = torch.cuda.get_device_properties(0).multi_processor_count
geoms = ([None, (64, 0, 0, 0), (64, 0, 32, 0), (64, 0, 64, 0), (64, 0, 128, 0), (32, 0, 64, 0), (128, 0, 128, 0)] if cfg_name == "simple"
         else [None, (32, 32, 0, 0), (32, 32, 0, 32), (32, 32, 0, 64), (64, 32, 0, 64), (32, 16, 0, 32)])
for n in (16384, 8192, 4096, 2048):
    for g in geoms:
        c = da.LiteralCodec(cfg, L)
        c.tune_tables(1)
        try:
            if g is not None:
                c.set_decoder(2, g, (31 if cfg_name == "simple" else 5, 5, 5, 5), blocks=max(1, (n + 15) // 16))
            ms = []
            for _ in range(3):
                c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, L, d_back)
                torch.cuda.synchronize(); ms.append(c.info().last_decode_ms)
            ok = bool(torch.equal(d_back[:n], d_in[:n]))
            print(f"{cfg_name} n {n:6d} rows {str(g):18s}: decode {min(ms):7.2f} ms  ({n * L / 1e6 / min(ms):6.2f} GB/s) resident {c.info().resident_groups} {c.last_decode_kernel().split('<')[1]} ok={ok}", flush=True)
        except Exception as e:
            print(f"{cfg_name} n {n} rows {g}: {e}", flush=True)
        c.close()
