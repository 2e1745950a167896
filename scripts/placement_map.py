#!/usr/bin/env python3
"""Decode time against WHERE in device memory the tables lie: K codecs created one after the other and all kept alive (so every one gets the
next free stretch of memory), each decoding the same one-round batch.   python scripts/placement_map.py [--codecs 40] [--config simple]
(DIVANS_TABLES_ALLOC selects how the tables are allocated.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codecs", type=int, default=40)
    ap.add_argument("--config", default="simple")
    ap.add_argument("--streams", type=int, default=28672)
    args = ap.parse_args()
    import torch
    import divans_amd as da
    import workload
    from bench import device_blocks
    dev = torch.device("cuda", 0)
    N, L = args.streams, 65536
    d_in = device_blocks(torch, torch.from_numpy(workload.load_corpus()).to(dev), 0, N, L)
    cfg = da.config_simple() if args.config == "simple" else da.config_context_mixing()
    enc = da.LiteralCodec(cfg, L)
    outs = enc.alloc_encode_outputs(N, L)
    enc.encode_batch(d_in, N, L, outs)
    torch.cuda.synchronize()
    enc.close(); da.trim(); torch.cuda.empty_cache()
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    codecs, times = [], []
    for k in range(args.codecs):
        free = torch.cuda.mem_get_info()[0]
        if free < 12 * 2**30:
            break
        c = da.LiteralCodec(cfg, L); c.set_decoder(2)
        ms = []
        for _ in range(2):
            c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
            torch.cuda.synchronize()
            ms.append(c.info().last_decode_ms)
        codecs.append(c); times.append(min(ms))
        print(f"{k:3d}  free before {free / 2**30:6.1f} GiB  decode {min(ms):7.2f} ms", flush=True)
    s = sorted(times)
    print(f"{len(times)} placements: min {s[0]:.2f} median {s[len(s) // 2]:.2f} max {s[-1]:.2f}; below min + 3 %: {sum(t < s[0] * 1.03 for t in times)}", flush=True)
    ok = bool(torch.equal(d_back, d_in))
    print("last output ok:", ok)
    for c in codecs:
        c.close()


if __name__ == "__main__":
    main()
