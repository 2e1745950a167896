#!/usr/bin/env python3
"""Serial rate of the bucketed two-model chains: streams of one repeated byte are ONE bucket each, so the chain kernels'
duration is (stream length) x (time per byte of a lane); run under rocprofv3 --kernel-trace --stats to read it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import divans_amd as da
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "const"
dev = torch.device("cuda", 0)
if kind == "const":
    d_in = torch.full((n * L + 64,), 0x65, dtype=torch.uint8, device=dev)
else:   # two alternating letters: two buckets per stream in the stride model, one or two in the context model
    d_in = torch.from_numpy(np.resize(np.frombuffer(b"et", dtype=np.uint8), n * L + 64)).to(dev)
codec = da.LiteralCodec(da.config_context_mixing(), L)
outs = codec.alloc_encode_outputs(n)
for _ in range(2):
    codec.encode_batch(d_in, n, L, outs)
    torch.cuda.synchronize()
print(codec.info().last_model_ms if hasattr(codec, "info") else "")
