#!/usr/bin/env python3
"""Tuning aid: decode time of the second-generation decoder (lit_decode2.hip) over cache geometries and persistent grids, next
to the first generation's default, one process, data built once.

    python scripts/decode2_sweep.py [--streams 65536] [--config simple|mixing] [--geoms hs:hc:ls:lc:sh_hs:sh_hc:sh_ls:sh_lc:wg[:gen],...]

gen 4 = lit_decode_t.hip (one lane per stream): wg = 64-stream workgroups per CU.
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
    ap.add_argument("--geoms", default="")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch
    import divans_amd as da
    import workload
    from bench import device_blocks
    dev = torch.device("cuda", 0)
    N, L = args.streams, 65536
    corpus = workload.load_corpus()
    d_in = device_blocks(torch, torch.from_numpy(corpus).to(dev), 0, N, L)
    cfg = da.config_simple() if args.config == "simple" else da.config_context_mixing()
    enc = da.LiteralCodec(cfg, L)
    outs = enc.alloc_encode_outputs(N, L)
    enc.encode_batch(d_in, N, L, outs)
    torch.cuda.synchronize()
    i = enc.info()
    print(f"{args.config} {N} streams; encode: model {i.last_model_ms:.1f} ms rans {i.last_rans_ms:.1f} ms", flush=True)
    enc.close()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)

    def run(label, setup):
        try:
            c = da.LiteralCodec(cfg, L)
            setup(c)
            ms = []
            for _ in range(args.reps):
                d_back.zero_()
                c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
                torch.cuda.synchronize()
                ms.append(c.info().last_decode_ms)
            ok = bool(torch.equal(d_back, d_in)) and c.status() == 0
            inf = c.info()
            print(f"{label}: decode {min(ms):8.2f} ms  {N * L / 1e6 / min(ms):7.2f} GB/s  resident {inf.resident_groups} ok={ok}", flush=True)
            c.close()
        except Exception as e:  # noqa: BLE001
            print(f"{label}: FAILED {e}", flush=True)

    if da.experimental_decoders():      # generations 1 and 4 exist only in DIVANS_WITH_EXPERIMENTAL_DECODERS=1 builds
        run("gen1 default", lambda c: c.set_decoder(1))
    run("gen2 default (direct mapped)", lambda c: c.set_decoder(2))
    run("gen3 default (2-way)", lambda c: c.set_decoder(3))
    if da.experimental_decoders():
        run("gen4 default (one lane per stream, no caches, one wave per SIMD)", lambda c: c.set_decoder(4))
    if args.geoms:
        geoms = [tuple(int(x) for x in g.split(":")) for g in args.geoms.split(",")]
    elif args.config == "simple":
        geoms = [(32, 0, 0, 0, 31, 5, 5, 5, w) for w in (5, 6, 7, 8)] + [(32, 0, 0, 0, 5, 5, 5, 5, 7), (16, 0, 0, 0, 31, 5, 5, 5, 8), (16, 0, 0, 0, 31, 5, 5, 5, 7),
                 (64, 0, 0, 0, 31, 5, 5, 5, 4), (32, 0, 16, 0, 31, 5, 5, 5, 6), (32, 0, 32, 0, 31, 5, 5, 5, 4), (32, 0, 64, 0, 31, 5, 5, 5, 3),
                 (64, 0, 128, 0, 31, 5, 5, 5, 1), (0, 0, 0, 0, 5, 5, 5, 5, 8), (0, 0, 0, 0, 5, 5, 5, 5, 7)]
    else:
        geoms = [(16, 16, 0, 0, 5, 5, 5, 5, w) for w in (5, 6, 7, 8)] + [(32, 16, 0, 0, 5, 5, 5, 5, 5), (32, 16, 0, 0, 5, 5, 5, 5, 4), (16, 8, 0, 0, 5, 5, 5, 5, 7),
                 (32, 0, 0, 0, 5, 5, 5, 5, 7), (16, 16, 0, 16, 5, 5, 5, 5, 5), (16, 16, 0, 32, 5, 5, 5, 5, 4), (32, 16, 0, 32, 5, 5, 5, 5, 3),
                 (16, 16, 16, 16, 5, 5, 5, 5, 4), (0, 0, 0, 0, 5, 5, 5, 5, 7)]
    for g in geoms:
        rows, shifts, wpc = g[0:4], g[4:8], g[8]
        gen = g[9] if len(g) > 9 else 2
        run(f"gen{gen} rows {rows} shifts {shifts} wg/cu {wpc}", lambda c: c.set_decoder(gen, rows, shifts, blocks=cus * wpc))


if __name__ == "__main__":
    main()
