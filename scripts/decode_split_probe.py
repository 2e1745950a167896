#!/usr/bin/env python3
"""Measurement aid (round 5): is a 65 536-stream decode cheaper as whole rounds of the persistent grid + a separately launched tail?
The persistent grid holds 28 672 streams; 65 536 = 2 x 28 672 + 8 192, so in one launch 8 192 slots decode a third stream while the
other 20 480 are done, at the latency of a lone stream's dependency chain.  Decodes n streams of the same coded batch for several n and
prints the decode kernel's time (divans_gpu_codec_info, hipEvents on the launch stream).  usage: decode_split_probe.py [simple|mixing]"""
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
res = c.info().resident_groups
print(f"{cfg_name}: resident {res}")
for n in (65536, 2 * res, 65536 - 2 * res, res, 65536, 3 * res // 2, res // 2, 16384, 8192, 4096):
    ms = []
    for _ in range(3):
        c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], n, L, d_back)
        torch.cuda.synchronize(); ms.append(c.info().last_decode_ms)
    ok = bool(torch.equal(d_back[:n], d_in[:n]))
    print(f"  n = {n:6d} ({n / res:5.2f} grids): decode {min(ms):7.2f} ms  ({n * L / 1e6 / min(ms):6.2f} GB/s)  kernel {c.last_decode_kernel()} ok={ok}", flush=True)
