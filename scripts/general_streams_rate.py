#!/usr/bin/env python3
"""Rate of the segment entry points (row f3): the literals of a brotli-derived general stream -- Copy / Dict commands
between its Literal commands, literal block-type switches, clustered context map, per-context mixing values -- replicated
N times in HBM, coded and decoded through divans_gpu_lit_encode_segments_batch / _decode_segments_batch.
usage: general_streams_rate.py [copies] [ir name ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import divans_amd as da, irtext

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
names = sys.argv[2:] or ["alice29-q11", "alice29-priors"]
dev = torch.device("cuda", 0)
for name in names:
    for mixing in (0, 2):
        ir = da.CommandIR(irtext.load_ir_text(name))
        lit, segs = ir.literal_segments()
        cfg = ir.lit_config(dynamic_context_mixing=mixing, use_context_map=1)
        n, L = copies, int(lit.size)
        d_lit = torch.from_numpy(np.concatenate([lit, np.zeros(64, np.uint8)])).to(dev)
        d_all = torch.cat([d_lit[:L].repeat(n), torch.zeros(64, dtype=torch.uint8, device=dev)])
        d_off = torch.arange(n, dtype=torch.int64, device=dev) * L
        d_sz = torch.full((n,), L, dtype=torch.int32, device=dev)
        seg_begin = torch.arange(n + 1, dtype=torch.int32, device=dev) * int(segs.size)
        d_segs = torch.from_numpy(np.ascontiguousarray(np.tile(segs, n)).view(np.uint8)).to(dev)
        codec = da.LiteralCodec(cfg, max(L, 16))
        codec.set_block_types(ir.num_block_types)
        outs = codec.alloc_encode_outputs(n, max(L, 16))
        back = torch.zeros_like(d_all)
        best_e = best_d = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            codec.encode_segments_batch(d_all, d_off, d_sz, n, L, seg_begin, d_segs, outs)
            torch.cuda.synchronize(); best_e = min(best_e, time.perf_counter() - t)
            t = time.perf_counter()
            codec.decode_segments_batch(outs["out"], outs["offsets"], outs["sizes"], n, L, seg_begin, d_segs, back, d_off, d_sz)
            torch.cuda.synchronize(); best_d = min(best_d, time.perf_counter() - t)
        assert codec.status() == 0
        assert torch.equal(back[:n * L], d_all[:n * L])
        coded = int(outs["sizes"].sum().item())
        info = codec.info()
        print(json.dumps({"ir": name, "dynamic_context_mixing": mixing, "streams": n, "literal_bytes_per_stream": L, "segments_per_stream": int(segs.size),
                          "block_types": int(ir.num_block_types), "rows_per_stream": int(info.rows_per_stream), "ratio_of_literals": round(coded / (n * L), 4),
                          "encode_MBps": round(n * L / best_e / 1e6, 1), "decode_MBps": round(n * L / best_d / 1e6, 1),
                          "encode_decode_MBps": round(n * L / (best_e + best_d) / 1e6, 1), "round_trip_ok": True}))
        codec.close(); ir.close()
