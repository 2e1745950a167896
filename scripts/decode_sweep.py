#!/usr/bin/env python3
"""Tuning aid: decode-kernel time over persistent-grid / LDS-cache geometries, one process, data built once.

    python scripts/decode_sweep.py [--streams 65536] [--config simple|mixing] [--data corpus|zeros]
Prints one line per geometry: lanes, cache organisation, workgroups per CU, decode kernel ms (hipEvent inside the
C ABI), decode GB/s, round trip ok.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--config", default="simple")
    ap.add_argument("--data", default="corpus")
    ap.add_argument("--geoms", default="")
    args = ap.parse_args()
    import torch
    import divans_amd as da
    import workload
    from bench import device_blocks
    dev = torch.device("cuda", 0)
    N, L = args.streams, 65536
    corpus = workload.load_corpus()
    d_in = torch.zeros((N, L), dtype=torch.uint8, device=dev) if args.data == "zeros" else device_blocks(torch, torch.from_numpy(corpus).to(dev), 0, N, L)
    cfg = da.config_simple() if args.config == "simple" else da.config_context_mixing()
    enc = da.LiteralCodec(cfg, L)
    outs = enc.alloc_encode_outputs(N, L)
    enc.encode_batch(d_in, N, L, outs)
    torch.cuda.synchronize()
    i = enc.info()
    print(f"encode: model {i.last_model_ms:.1f} ms rans {i.last_rans_ms:.1f} ms", flush=True)
    enc.close()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    # (lanes, mode, high rows, low rows, workgroups per CU); mode: u = unified, s = split, h = high only, n = none
    if args.geoms:
        geoms = [tuple(int(x) if x.lstrip("-").isdigit() else x for x in g.split(":")) for g in args.geoms.split(",")]
    elif args.config == "simple":
        geoms = [(16, "h", 32, 0, 7), (16, "h", 32, 0, 5), (16, "h", 64, 0, 4), (16, "h", 64, 0, 3), (16, "h", 128, 0, 2),
                 (16, "s", 32, 32, 4), (16, "s", 32, 32, 3), (16, "s", 32, 64, 3), (16, "s", 32, 64, 2), (16, "s", 64, 64, 2),
                 (16, "s", 64, 128, 1), (16, "s", 64, 64, 1), (16, "s", 32, 128, 1), (16, "s", 32, 128, 2) if False else (16, "s", 64, 128, 1),
                 (16, "u", 64, 0, 4), (16, "u", 128, 0, 2), (16, "u", 256, 0, 1), (16, "u", 128, 0, 1), (16, "n", 0, 0, 7), (16, "n", 0, 0, 8),
                 (8, "h", 32, 0, 4), (8, "h", 32, 0, 3), (8, "h", 32, 0, 2), (8, "h", 64, 0, 2), (8, "h", 64, 0, 1), (8, "h", 128, 0, 1)]
    else:
        geoms = [(16, "h", 32, 0, 7), (16, "h", 32, 0, 5), (16, "h", 64, 0, 4), (16, "h", 64, 0, 3), (16, "h", 128, 0, 2), (16, "s", 32, 32, 3),
                 (16, "s", 64, 64, 2), (16, "s", 64, 128, 1), (16, "u", 64, 0, 4), (16, "u", 128, 0, 2), (16, "u", 256, 0, 1), (16, "n", 0, 0, 7)]
    for lanes, mode, hi, lo, wpc in geoms:
        try:
            c = da.LiteralCodec(cfg, L)
            if lanes == 8:
                raise RuntimeError('the packed 8-lane layout was removed in round 3')
            if mode == "u":
                c.set_geometry(cache_rows=hi)
            elif mode == "n":
                c.set_geometry(cache_rows=0)
            else:
                c.set_split_cache(hi, lo)
            c.set_geometry(blocks=cus * wpc)
            ms = []
            for _ in range(2):
                d_back.zero_()
                c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
                torch.cuda.synchronize()
                ms.append(c.info().last_decode_ms)
            ok = bool(torch.equal(d_back, d_in))
            print(f"lanes {lanes:2d} cache {mode} high {hi:3d} low {lo:3d} wg/cu {wpc}: decode {min(ms):8.2f} ms  {N * L / 1e6 / min(ms):7.2f} GB/s  ok={ok}", flush=True)
            c.close()
        except Exception as e:  # noqa: BLE001
            print(f"lanes {lanes} cache {mode} high {hi} low {lo} wg/cu {wpc}: FAILED {e}", flush=True)


if __name__ == "__main__":
    main()
