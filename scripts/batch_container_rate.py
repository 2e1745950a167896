#!/usr/bin/env python3
"""Informational: complete-container rate of the batch ABI (include/divans_batch.h) from and to HOST memory, with the
host/GPU overlap it achieves.  Never the bench `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import divans_amd as da
import workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"    # 1500 distinct lengths between 1 KiB and 64 KiB instead of n x 64 KiB
device = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # -1 = DIVANS_BATCH_ALL_DEVICES: one call over every visible GPU
corpus = workload.load_corpus()
blocks = workload.make_blocks(corpus, 0, n)
if mixed:
    lens = [1024 + (i % 1500) * 43 for i in range(n)]
    inputs = [blocks[i][:lens[i]] for i in range(n)]
else:
    inputs = [blocks[i] for i in range(n)]
raw = int(sum(x.size for x in inputs))
order = tuple(int(x) for x in sys.argv[5].split(",")) if len(sys.argv) > 5 else (0, 2)   # e.g. "2,0,0": which configurations, in which order
for mixing in order:
    opts = da.batch_options(dynamic_context_mixing=mixing, use_context_map=0 if mixing == 0 else 1, force_stride=1 if mixing == 0 else 0, host_threads=threads, device=device)
    da.batch_compress(inputs[:64], opts)          # module load
    # first call of this size: the lanes' codecs, device scratch and page-locked staging buffers are created inside it;
    # the library keeps them (divans_batch_release() frees them), so later calls -- the steady state reported below -- do not
    t0 = time.time(); cont, tc1 = da.batch_compress(inputs, opts); first_c = time.time() - t0
    t0 = time.time(); back, td1 = da.batch_decompress(cont, raw, opts); first_d = time.time() - t0
    t0 = time.time(); cont, tc = da.batch_compress(inputs, opts); wall_c = time.time() - t0; pc = da.batch_last_phases()
    t0 = time.time(); back, td = da.batch_decompress(cont, raw, opts); wall_d = time.time() - t0; pd = da.batch_last_phases()
    ok = all((back[i] == inputs[i]).all() for i in range(0, n, max(1, n // 64)))
    print(json.dumps({"streams": n, "device": device, "lengths": "1500 distinct, 1024 .. 65501 B" if mixed else "65536 B", "input_bytes": raw, "dynamic_context_mixing": mixing, "round_trip_ok": ok, "container_bytes": int(sum(c.size for c in cont)),
                      "compress": {k: round(v, 2) for k, v in tc.items()}, "decompress": {k: round(v, 2) for k, v in td.items()},
                      "compress_host_phases": pc, "decompress_host_phases": pd,
                      "compress_MBps": round(raw / 1e6 / (tc["total_ms"] / 1e3), 1), "decompress_MBps": round(raw / 1e6 / (td["total_ms"] / 1e3), 1),
                      "first_call_ms": [round(tc1["total_ms"], 1), round(td1["total_ms"], 1)],
                      "python_wall_s": [round(wall_c, 2), round(wall_d, 2)]}))
