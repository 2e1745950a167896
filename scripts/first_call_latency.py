#!/usr/bin/env python3
"""Measurement aid (round 6): what does the FIRST divans_gpu_lit_decode_batch call of a fresh codec with >= 2 GiB of tables cost,
wall clock from the call to its output being complete, under the library's placement policy (one candidate per call) and under the eager form
(divans_gpu_codec_tune_tables(c, 12)) -- against the steady state of the same codec.  usage: python scripts/first_call_latency.py [simple|mixing]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import divans_amd as da  # noqa: E402
import workload  # noqa: E402
from bench import device_blocks, alloc_packed_outputs  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "simple"
N, L = 65536, 65536
dev = torch.device("cuda", 0)
corpus_t = torch.from_numpy(workload.load_corpus()).to(dev)
d_in = device_blocks(torch, corpus_t, 0, N, L)
cfg = (da.config_simple if name == "simple" else da.config_context_mixing)()
enc = da.LiteralCodec(cfg, L)
outs = alloc_packed_outputs(torch, N, L, dev)
enc.encode_packed(d_in, N, L, outs["packed"], outs["packed_offsets"], outs["sizes"], outs["packed_total"])
torch.cuda.synchronize()
enc.close()
d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)


def timed_call(codec):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codec.decode_batch(outs["packed"], outs["packed_offsets"], outs["sizes"], N, L, d_back)
    t_return = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_return * 1e3, (time.perf_counter() - t0) * 1e3


print(f"{name}: {N} x {L} B streams, decode-only codecs (the byte order is learned from the first call's output: calls 1 and 2 are outside the search)")
for label, setup in (("library policy (one placement per call)", lambda c: None), ("eager form, divans_gpu_codec_tune_tables(c, 12)", lambda c: c.tune_tables(12)),
                     ("no search, divans_gpu_codec_tune_tables(c, 1)", lambda c: c.tune_tables(1))):
    codec = da.LiteralCodec(cfg, L)
    setup(codec)
    rows = []
    for call in range(18):
        ret, done = timed_call(codec)
        rows.append((ret, done, codec.table_placement()))
    assert torch.equal(d_back, d_in)
    steady = np.median([r[1] for r in rows[-3:]])
    print(f"  {label}")
    print("    call:            " + " ".join(f"{i + 1:7d}" for i in range(18)))
    print("    returned after:  " + " ".join(f"{r[0]:7.1f}" for r in rows) + "  ms")
    print("    output complete: " + " ".join(f"{r[1]:7.1f}" for r in rows) + "  ms")
    print(f"    steady state {steady:.1f} ms; first call {rows[0][1] / steady:.2f} x, slowest call {max(r[1] for r in rows) / steady:.2f} x steady state; placement {rows[-1][2]}")
    codec.close()
    torch.cuda.empty_cache()
