#!/usr/bin/env python3
"""Measurement aid (round 5): would the encoder gain from running its two 32 768-stream sub-batches on two HIP streams (model pass of one under
the rANS pass of the other)?  Two codecs, each with its own work arrays and its own stream, code one half of the batch each:
(a) one after the other, (b) at once, (c) at once with the second delayed by a sort + chain's worth of time.  Wall time by events on the default stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import divans_amd as da
import workload
from bench import device_blocks
dev = torch.device("cuda", 0)
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "simple"
H, L = 32768, 65536
corpus = workload.load_corpus()
d_in = device_blocks(torch, torch.from_numpy(corpus).to(dev), 0, 2 * H, L)
cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c1 = da.LiteralCodec(cfg, L, stream=s1); c2 = da.LiteralCodec(cfg, L, stream=s2)
o1 = c1.alloc_encode_outputs(H, L); o2 = c2.alloc_encode_outputs(H, L)
def both(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c1.encode_batch(d_in[:H], H, L, o1)
    if mode == "serial":
        s1.synchronize()
    c2.encode_batch(d_in[H:], H, L, o2)
    s1.synchronize(); s2.synchronize()
    return (time.perf_counter() - t0) * 1e3
for mode in ("serial", "concurrent", "serial", "concurrent", "serial", "concurrent"):
    ms = both(mode)
    i1, i2 = c1.info(), c2.info()
    print(f"{cfg_name} {mode:10s}: wall {ms:7.2f} ms   codec 1 model {i1.last_model_ms:6.2f} rans {i1.last_rans_ms:5.2f}   codec 2 model {i2.last_model_ms:6.2f} rans {i2.last_rans_ms:5.2f}", flush=True)
