#!/usr/bin/env python3
"""One codec, one batch, a few decode launches -- the workload of scripts/placement_counters.sh (rocprofv3 --pmc on the decode kernel under
different placements of the CDF tables, DIVANS_TABLES_ALLOC).   python scripts/decode_once.py [--config simple|mixing] [--streams 65536] [--reps 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="simple")
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--codecs", type=int, default=1, help="this many codecs alive at once (each with its own tables), decoding in turn")
    args = ap.parse_args()
    import numpy as np
    import torch
    import divans_amd as da
    import workload
    dev = torch.device("cuda", 0)
    N, L = args.streams, 65536
    # the blocks are built on the host and kept in a file: no torch kernels run under the profiler (an elementwise kernel of the on-GPU
    # generator crashes inside rocprofv3 --pmc)
    os.makedirs("/tmp/divans_cache", exist_ok=True)
    cache = f"/tmp/divans_cache/divans_blocks_0_{N}_{L}.npy"
    d_in = torch.empty((N, L), dtype=torch.uint8, device=dev)
    if os.path.exists(cache):
        host = np.load(cache, mmap_mode="r")
        for c0 in range(0, N, 4096):
            d_in[c0:min(N, c0 + 4096)].copy_(torch.from_numpy(np.ascontiguousarray(host[c0:min(N, c0 + 4096)])))
    else:
        corpus = workload.load_corpus()
        host = np.lib.format.open_memmap(cache, mode="w+", dtype=np.uint8, shape=(N, L))
        for c0 in range(0, N, 2048):
            c1 = min(N, c0 + 2048)
            blk = workload.make_blocks(corpus, c0, c1 - c0, block_len=L)
            host[c0:c1] = blk
            d_in[c0:c1].copy_(torch.from_numpy(blk))
        host.flush()
    cfg = da.config_simple() if args.config == "simple" else da.config_context_mixing()
    codec = da.LiteralCodec(cfg, L)
    outs = codec.alloc_encode_outputs(N, L)
    codec.encode_batch(d_in, N, L, outs)
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    codecs = [codec] + [da.LiteralCodec(cfg, L) for _ in range(args.codecs - 1)]
    for k, c in enumerate(codecs):
        c.set_decoder(2)
        ms = []
        for _ in range(args.reps):
            c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
            torch.cuda.synchronize()
            ms.append(c.info().last_decode_ms)
        ok = c.status() == 0 and bytes(d_back[N // 3, 1000:1100].cpu().numpy()) == bytes(d_in[N // 3, 1000:1100].cpu().numpy())
        print(f"codec {k}: {args.config} {N} streams, tables {os.environ.get('DIVANS_TABLES_ALLOC', 'chunks')}: decode {' '.join('%.2f' % m for m in ms)} ms  ok={ok}", flush=True)
    for c in codecs:
        c.close()


if __name__ == "__main__":
    main()
