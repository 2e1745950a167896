#!/usr/bin/env python3
"""Does the decode time depend on WHERE the CDF tables land in device memory?  The same decoder configuration, timed in one process
before and after other allocations of different sizes have come and gone (scripts/decode2_sweep.py showed 'gen2 default' 429-462 ms
next to an identical explicit geometry at 477 ms, the only difference being what had been allocated and freed before).

    python scripts/placement_probe.py [--streams 65536] [--config mixing]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--config", default="mixing")
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
    enc.close()
    torch.cuda.empty_cache()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    rows = (16, 16, 0, 0) if args.config == "mixing" else (32, 0, 0, 0)
    shifts = (5, 5, 5, 5) if args.config == "mixing" else (31, 5, 5, 5)

    def run(label, gen, explicit):
        c = da.LiteralCodec(cfg, L)
        if explicit == 2:
            c.set_decoder(gen, rows, shifts)
        elif explicit == 3:
            c.set_decoder(gen, None, None, blocks=cus * 7)
        elif explicit == 4:
            c.set_decoder(gen, rows, (5, 5, 0, 0), blocks=cus * 7)
        elif explicit:
            c.set_decoder(gen, rows, shifts, blocks=cus * 7)
        else:
            c.set_decoder(gen)
        ms = []
        for _ in range(3):
            c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
            torch.cuda.synchronize()
            ms.append(c.info().last_decode_ms)
        free, total = torch.cuda.mem_get_info()
        i = c.info()
        print(f"{label:58s} decode {' '.join('%7.2f' % m for m in ms)} ms   (free {free / 2**30:6.1f} GiB) {c.last_decode_kernel()[30:]} resident {i.resident_groups} tables {i.table_bytes}", flush=True)
        c.close()

    def churn(gib):
        t = torch.empty(int(gib * 2**30), dtype=torch.uint8, device=dev); t.fill_(1); torch.cuda.synchronize()
        del t; torch.cuda.empty_cache()
        print(f"-- allocated, filled and freed {gib} GiB", flush=True)

    # several codecs alive at once, decoding in turn: does a codec keep its speed (placement of ITS tables) or does the speed alternate with
    # the launch order (dispatch state)?
    def timed(c):
        c.decode_batch(outs["out"], outs["offsets"], outs["sizes"], N, L, d_back)
        torch.cuda.synchronize()
        return c.info().last_decode_ms
    codecs = []
    for k in range(6):
        c = da.LiteralCodec(cfg, L); c.set_decoder(2); codecs.append(c)
        print(f"codec {k} created: first decode {timed(c):7.2f} ms", flush=True)
    t = [timed(c) for c in codecs]
    print(f"six codecs alive: {' '.join('%7.2f' % x for x in t)}   min {min(t):.2f} mean {sum(t) / len(t):.2f} max {max(t):.2f}  rows/stream {codecs[0].info().rows_per_stream}", flush=True)
    for c in codecs:
        c.close()


if __name__ == "__main__":
    main()
