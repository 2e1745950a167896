#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X literal coder.

Metric (BASELINE.json): MB/s encode+decode per GPU on 64 KiB metablocks, bit-exact vs the CPU path (= the in-repo
oracle, a restatement of the reference's CPU path; the reference itself is Rust and cannot be built here).
A "step" = one encode pass + one decode pass of the hot path over the whole batch of independent 64 KiB streams
(configs[1]: 65 536 streams, stride 1 / context map off = reference TestSimple), inputs resident in HBM.
value = N * 65536 bytes * n_gpus / (t_enc + t_dec) in MB/s (10^6 B/s), whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--config all|simple|mixing|decode_only]

One JSON line.  `value` is always BASELINE configs[1]; at N = 1 the line also carries `configs.mixing` (configs[2]:
context map + dynamic_context_mixing = 2) and `configs.decode_only` (configs[3]: pre-encoded random_then_unicode x 4096),
each with its own bit-exactness flag, kernel times and roofline.

Multi-GPU: `--gpus N` with no WORLD_SIZE in the environment re-launches itself as N ranks under torch.distributed.run
(one process per GPU, RCCL); under an external launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Rank 0 builds the
whole job's input and scatters contiguous stream ranges, every rank codes its shard (timed region: no collective in
it, weak scaling), then the coded sizes are exchanged and the coded bytes gathered to rank 0 and checked there
(divans_amd/sharding.py; reported as `multi_gpu`, never part of `value`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MASK64 = (1 << 64) - 1


def _i64(v):
    v &= MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


def device_blocks(torch, corpus_t, first, count, block_len, chunk=8192):
    """tests/workload.make_blocks on the GPU (same bytes): block i = corpus[o_i : o_i + L], o_i = (i * 4099) mod
    (len - L), then L // 100 bytes XORed with a non-zero value, positions / values from xorshift64* seeded
    0x9E3779B97F4A7C15 ^ i.  int64 arithmetic wraps like uint64; logical right shifts are masked."""
    dev = corpus_t.device
    L = int(block_len)
    span = corpus_t.numel() - L
    out = torch.empty((count, L), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev, dtype=torch.int64)
    lsr = lambda x, k: (x >> k) & ((1 << (64 - k)) - 1)
    for c0 in range(0, count, chunk):
        c1 = min(count, c0 + chunk)
        idx = torch.arange(first + c0, first + c1, device=dev, dtype=torch.int64)
        blk = corpus_t[((idx * 4099) % span)[:, None] + ar[None, :]]
        x = idx ^ _i64(0x9E3779B97F4A7C15)
        x = torch.where(x == 0, torch.ones_like(x), x)
        rows = torch.arange(c1 - c0, device=dev, dtype=torch.int64)
        for _ in range(L // 100):
            x = x ^ lsr(x, 12)
            x = x ^ (x << 25)
            x = x ^ lsr(x, 27)
            o = x * 2685821237909765
            pos = lsr(o, 20) % L
            val = (lsr(o, 8) & 0xFF).to(torch.uint8)
            val = torch.where(val == 0, torch.ones_like(val), val)
            blk[rows, pos] ^= val
        out[c0:c1] = blk
    return out


def usable_parallelism():
    """How many host threads can actually run at once: affinity mask and cgroup CPU quota, whichever is smaller."""
    info = {"nproc": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = info["nproc"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
        except Exception:
            pass
    if quota is None:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    info["cgroup_cpu_quota"] = quota
    usable = info["affinity"] if quota is None else max(1, min(info["affinity"], int(quota)))
    info["usable"] = usable
    return info


def cpu_baseline(cfg_name, workload, corpus, block_len, sample_blocks=None, decode_only=False, per_thread=None):
    """The C oracle ("port" of the reference CPU path) timed on this box's host cores on a bounded sample of the same
    workload: (1) one thread; (2) one independent stream per worker on every usable core.  Workers allocate their coder
    state and buffers before a start barrier; each direction is timed as wall clock from the barrier release to the last
    worker finishing (oracle/literal.c orc_lit_batch_bench).  sample_blocks(n) supplies n blocks of the workload (default:
    tests/workload.py blocks 0..n-1); decode_only reports the decode direction alone (BASELINE configs[3])."""
    import ctypes
    import pyoracle as po
    try:
        lib = po.lib(native=True)
    except Exception:
        lib = po.lib()
    cfg = po.config_simple() if cfg_name == "simple" else po.config_context_mixing()
    par = usable_parallelism()
    if sample_blocks is None:
        sample_blocks = lambda n: workload.make_blocks(corpus, 0, n, block_len=block_len)

    def run(n, threads):
        blocks = sample_blocks(n)
        enc = ctypes.c_double(0); dec = ctypes.c_double(0); coded = ctypes.c_uint64(0)
        rc = lib.orc_lit_batch_bench(ctypes.byref(cfg), blocks.ctypes.data, n, block_len, threads,
                                     ctypes.byref(enc), ctypes.byref(dec), ctypes.byref(coded))
        assert rc == 0, "oracle round trip failed"
        total = n * block_len
        both = enc.value + dec.value
        return {"streams": n, "threads": threads, "enc_s": enc.value, "dec_s": dec.value,
                "encode_MBps": round(total / 1e6 / enc.value, 2), "decode_MBps": round(total / 1e6 / dec.value, 2),
                "MBps": round(total / 1e6 / (dec.value if decode_only else both), 2)}

    scale = max(1, 65536 // max(block_len, 1))
    if per_thread is None:
        per_thread = 256 if cfg_name == "simple" else 96
    single = run(per_thread * scale, 1)
    threads = par["usable"]
    allc = run(threads * per_thread * scale, threads)
    factor = allc["MBps"] / single["MBps"]
    note = ""
    if factor < 0.5 * threads:
        note = (f"; all-core scaling {factor:.1f}x of {threads} threads: every worker walks its own 12.6 MB of prior tables "
                f"(2 x 3*256*256 rows as the reference lays them out, codec/priors.rs:35-37), so the cores share L3 / memory bandwidth")
    what = "decode only" if decode_only else "encode+decode"
    return {
        "value": allc["MBps"], "unit": f"MB/s {what}", "cores": threads, "kind": "port",
        "sample": (f"all cores: {allc['streams']} x {block_len} B streams of the same workload on {threads} threads (1 stream per thread at a time), "
                   f"wall enc {allc['enc_s']:.2f}s + dec {allc['dec_s']:.2f}s; single thread: {single['streams']} streams, "
                   f"enc {single['enc_s']:.2f}s + dec {single['dec_s']:.2f}s; buffers allocated before the start barrier" + note),
        "encode_MBps": allc["encode_MBps"], "decode_MBps": allc["decode_MBps"],
        "single_thread": {k: single[k] for k in ("streams", "MBps", "encode_MBps", "decode_MBps")},
        "all_cores": {k: allc[k] for k in ("streams", "threads", "MBps", "encode_MBps", "decode_MBps")},
        "scaling_factor": round(factor, 2), "parallelism": par,
    }


def load_traffic(cfg_name, n, block_len):
    """HBM bytes per launch REPLAYED from the committed rocprofv3 PMC passes of this same workload (profiles/traffic_latest.json;
    counters cannot be collected inside a plain bench run).  The returned dict carries the profile tag under "_source"."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        ent = tj if "configs" not in tj else tj["configs"].get(cfg_name, {})
        if ent.get("streams") == n and ent.get("block_bytes") == block_len and ent.get("config", cfg_name) == cfg_name:
            k = dict(ent["kernels"])
            k["_source"] = f"replayed from profiles/traffic_latest.json (rocprofv3 --pmc passes, tag {tj.get('tag', ent.get('tag', '?'))}), not measured in this run"
            return k
    except Exception:
        pass
    return {}


def _traffic_entry(traffic, kernel):
    """the PMC record of `kernel` (a rocprofv3 name): exact name, else the name without its template arguments / namespace"""
    if not traffic:
        return None
    base = kernel.split("<")[0].split("::")[-1]
    for key in (kernel, base):
        if key in traffic:
            return traffic[key]
    return None


def roofline_of(kern_ms, alg_bytes, traffic, replay_ms=None):
    """`frac` = algorithmic bytes of the dominant kernel over its time over 8 TB/s (the contract's roofline).  Beside it, what the kernel is
    actually up against -- the fabric's rate for randomly addressed 32-byte rows: `request_rate_Gps` = the L2 -> fabric requests of the
    launch (TCC_EA0_RDREQ + WRREQ of the committed PMC pass) over THIS run's kernel time, `request_ceiling_Gps` = the same requests over
    the time of divans_gpu_codec_row_replay measured in THIS run on THIS box (the same rows through the same caches, layout and grid with
    no decoding in between), `request_frac` = their ratio = replay time / decode time."""
    dom = max(kern_ms, key=kern_ms.get)
    achieved = alg_bytes[dom] / 1e9 / (kern_ms[dom] / 1e3) if kern_ms[dom] > 0 else 0.0
    ent = _traffic_entry(traffic, dom)
    t = ent.get("hbm_bytes_per_launch") if ent else None
    moved = (t / 1e9 / (kern_ms[dom] / 1e3)) if (t and kern_ms[dom] > 0) else None
    req = (ent.get("rdreq_per_launch", 0) + ent.get("wrreq_per_launch", 0)) if ent and ent.get("rdreq_per_launch") else None
    rp = _traffic_entry(traffic, "row_replay_kernel")
    rp_req = (rp.get("rdreq_per_launch", 0) + rp.get("wrreq_per_launch", 0)) if rp and rp.get("rdreq_per_launch") else None
    extra = {}
    if replay_ms and kern_ms[dom] > 0:
        frac = replay_ms / kern_ms[dom]
        extra = {"replay_ms": round(replay_ms, 3), "request_frac": round(frac, 4),
                 "request_rate_Gps": (round(req / 1e9 / (kern_ms[dom] / 1e3), 2) if req else None),
                 "request_ceiling_Gps": (round((rp_req or req) / 1e9 / (replay_ms / 1e3), 2) if (rp_req or req) else None),
                 "requests_per_launch": req, "replay_requests_per_launch": rp_req,
                 "request_note": "replay = divans_gpu_codec_row_replay on this box in this run: every CDF row this batch's decode touches, loaded / blended / stored through the same LDS caches, table "
                                 "layout, byte order and persistent grid, without entropy decoding and with the low-nibble rows requested a byte ahead (a decoder cannot); request counts replayed from the committed PMC pass"}
        bound = ("fabric requests: randomly addressed 32-byte CDF rows (the decode kernel runs at %d %% of the memory side's own time for its row traffic)" % round(100 * frac)) if frac >= 0.75 else \
                ("a stream's dependency chain + VALU issue (the row traffic alone would take %d %% of the kernel's time)" % round(100 * frac))
        extra["bound_detail"] = bound
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", **extra,
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": t,
            "traffic_ratio": (round(t / alg_bytes[dom], 1) if t else None),     # HBM bytes moved per algorithmic byte
            # what the launch actually moves over the fabric per second (2 x FETCH_SIZE + WRITE_SIZE of the replayed PMC pass over THIS run's kernel
            # time): randomly addressed 128-byte fills and 32-byte write-backs, the pattern HBM sustains worst -- the kernel's real operating point
            "traffic_GBps": (round(moved, 1) if moved else None), "traffic_frac_of_peak": (round(moved / HBM_PEAK_GBS, 4) if moved else None),
            "traffic_source": (traffic.get("_source") if t is not None else "no committed PMC pass matches this kernel / workload"),
            "algorithmic_bytes_per_launch": alg_bytes[dom]}


def spawn_ranks(args):
    """`python bench.py --gpus N` from a plain shell: become N ranks (one per GPU) under torch.distributed.run."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("DIVANS_BENCH_SHARE_GPU") != "1":
        sys.exit(f"bench: --gpus {args.gpus} needs {args.gpus} GPUs on this node, {have} visible (no CPU fallback, no silent single-rank run)")
    port = 29400 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def alloc_packed_outputs(torch, N, L, dev):
    """What divans_gpu_lit_encode_packed fills: the coded streams contiguous (room for 1.125 x the input: text codes to 0.46, random bytes
    to 1.003; DIVANS_GPU_STATUS_OUTPUT_FULL = status bit 8 would say it was not enough), their offsets, sizes and total."""
    return {"packed": torch.empty(N * (L + L // 8) + 4096, dtype=torch.uint8, device=dev),
            "packed_offsets": torch.empty(N, dtype=torch.int64, device=dev), "sizes": torch.empty(N, dtype=torch.int32, device=dev),
            "packed_total": torch.zeros(1, dtype=torch.int64, device=dev)}


def timed_steps(torch, codec, d_in, N, L, outs, d_back, steps, warmup, barrier):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    rec = {"enc": [], "dec": [], "model": [], "rans": [], "dkern": [], "pack": []}
    # the coded streams leave the encoder contiguous (the pack launches are part of the timed encode: that is what north_star's
    # "gather of coded streams" ships) and the decoder reads them from there
    packed, poff, ptotal = outs["packed"], outs["packed_offsets"], outs["packed_total"]

    def step(record):
        # the codec launches on torch's current stream, so these events bracket its kernels
        ev[0].record()
        codec.encode_packed(d_in, N, L, packed, poff, outs["sizes"], ptotal)
        ev[1].record()
        codec.decode_batch(packed, poff, outs["sizes"], N, L, d_back)
        ev[2].record()
        if record:
            torch.cuda.synchronize()
            rec["enc"].append(ev[0].elapsed_time(ev[1])); rec["dec"].append(ev[1].elapsed_time(ev[2]))
            inf = codec.info()   # hipEvent timings taken inside the C ABI around each kernel launch, on the launch stream
            rec["model"].append(inf.last_model_ms); rec["rans"].append(inf.last_rans_ms); rec["dkern"].append(inf.last_decode_ms); rec["pack"].append(inf.last_pack_ms)

    for _ in range(warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    return time.perf_counter() - t0, rec


def verify_first_pass(torch, po, codec, ocfg, d_in, N, L, outs, d_back, n_check):
    """Correctness on the codec's very first pass (its scratch holds nothing from an earlier pass that could stand in for
    skipped work), outside the timed region: exact round trip of every stream + coded bytes == oracle on a spread of
    n_check streams (the oracle encodes them on every usable host core, oracle/literal.c orc_lit_batch_check)."""
    import numpy as np
    codec.encode_packed(d_in, N, L, outs["packed"], outs["packed_offsets"], outs["sizes"], outs["packed_total"])
    codec.decode_batch(outs["packed"], outs["packed_offsets"], outs["sizes"], N, L, d_back)
    torch.cuda.synchronize()
    ok = bool(torch.equal(d_back, d_in)) and codec.status() == 0
    picks = sorted(set([0, N // 2, N - 1] + list(range(0, N, max(1, N // max(n_check, 1))))))
    idx = torch.tensor(picks, dtype=torch.int64, device=d_in.device)
    host_in = d_in[idx].cpu().numpy()
    offs = outs["packed_offsets"][idx].cpu().numpy().astype(np.int64); sz = outs["sizes"][idx].cpu().numpy().astype(np.int64)
    # gather the picked streams' coded bytes into one host blob
    starts = np.concatenate([[0], np.cumsum(sz[:-1])]).astype(np.int64)
    pos = torch.arange(int(sz.sum()), device=d_in.device, dtype=torch.int64)
    seg = torch.repeat_interleave(torch.arange(len(picks), device=d_in.device), torch.tensor(sz, device=d_in.device))
    src = torch.tensor(offs, device=d_in.device)[seg] + (pos - torch.tensor(starts, device=d_in.device)[seg])
    blob = outs["packed"][src].cpu().numpy()
    bad, first = po.lit_batch_check(ocfg, host_in, blob, starts.astype(np.uint64), sz.astype(np.uint32), threads=usable_parallelism()["usable"])
    if bad:
        sys.stderr.write(f"bench: {bad} of {len(picks)} checked streams differ from the oracle, first: stream {picks[first]}\n")
    return ok and bad == 0, len(picks)


def _order_record(codec):
    """the byte order the decoder's tables are laid out in (divans_gpu_codec_byte_order): mode, and the 12 most frequent byte values of a learned rank"""
    bo = codec.byte_order(with_rank=True)
    rec = {"mode": {0: "learned from the codec's first batch (64 streams x 2 KiB sampled on the device)", 1: "numeric", 2: "English-text hint"}[bo["mode"]], "in_use": bo["ready"]}
    if bo["ready"]:
        inv = sorted(range(256), key=lambda b: bo["rank"][b])
        rec["first_ranks"] = "".join(chr(b) if 32 < b < 127 else ("_" if b == 32 else "?") for b in inv[:16])
    return rec


def settle_placement(codec, step, limit=20):
    """untimed steps until the library's call-by-call placement search is over (one placement per decode call)"""
    extra = 0
    p = codec.table_placement()
    if p["policy_candidates"] > 1 and p["tried"] == 0 and not p["searching"]:
        step(); extra += 1       # the search starts with the first decode that fills half the grid once the byte order is settled
    while codec.table_placement()["searching"] and extra < limit:
        step(); extra += 1
    return extra


def run_pair_config(torch, da, po, name, d_in, N, L, args, dev, barrier, world_note="", traffic_name=None, byte_order=None, steps=None, warmup=None):
    """encode+decode config (simple or mixing): verify, time, report."""
    cfg = da.config_simple() if name == "simple" else da.config_context_mixing()
    ocfg = po.config_simple() if name == "simple" else po.config_context_mixing()
    codec = da.LiteralCodec(cfg, L, device=dev.index)
    if args.blocks_per_cu:
        cus = torch.cuda.get_device_properties(dev.index).multi_processor_count
        codec.set_geometry(blocks=max(1, int(cus * args.blocks_per_cu)))
    if args.cache_rows >= 0:
        codec.set_geometry(cache_rows=args.cache_rows)
    if args.encode_path:
        codec.set_encode_path(args.encode_path)
    if args.bucket_batch:
        codec.set_bucket_batch(args.bucket_batch)
    if args.decoder_generation:
        codec.set_decoder(args.decoder_generation)
    if args.split_cache:
        hi_rows, lo_rows = (int(x) for x in args.split_cache.split(","))
        codec.set_split_cache(hi_rows, lo_rows)
    if byte_order is not None:
        codec.set_byte_order(byte_order)
    if args.table_candidates:
        codec.tune_tables(args.table_candidates)      # 0 = what a plain divans_gpu_codec_create caller gets: the library's call-by-call search, which the
                                                      # untimed passes below complete (verification + warm-up + settle_placement); k >= 2 = eager, on the first decode
    outs = alloc_packed_outputs(torch, N, L, dev)
    d_back = torch.empty((N, L), dtype=torch.uint8, device=dev)
    ok, checked = True, 0
    if not args.no_verify:
        ok, checked = verify_first_pass(torch, po, codec, ocfg, d_in, N, L, outs, d_back, args.check_streams)
        first_sizes = outs["sizes"].clone(); d_back.zero_()
    steps, warm = (steps or args.steps), (args.warmup if warmup is None else warmup)
    if args.no_verify and args.table_candidates != 1:
        warm = max(warm, 1)          # the decode that tries the table placements is never a timed one

    def untimed_decode():      # (the coded streams are in `outs` since the verification pass; without one the warm-up's first step fills them)
        codec.decode_batch(outs["packed"], outs["packed_offsets"], outs["sizes"], N, L, d_back)
    if args.no_verify:
        codec.encode_packed(d_in, N, L, outs["packed"], outs["packed_offsets"], outs["sizes"], outs["packed_total"])
    search_steps = settle_placement(codec, untimed_decode)    # the placement search ends before the warm-up: no timed step runs on a candidate
    elapsed, rec = timed_steps(torch, codec, d_in, N, L, outs, d_back, steps, warm, barrier)
    if not args.no_verify:   # the timed passes must have produced the same thing
        ok = ok and bool(torch.equal(outs["sizes"], first_sizes)) and bool(torch.equal(d_back, d_in)) and codec.status() == 0
    coded_total = int(outs["sizes"].to(torch.int64).sum().item())
    avg = lambda xs: sum(xs) / max(len(xs), 1)
    dk = codec.last_decode_kernel() or "lit_decode_kernel"      # as rocprofv3 --kernel-trace names the instance that ran
    kern = {dk: avg(rec["dkern"]), "encode_model_pass": avg(rec["model"]), "encode_rans_pass": avg(rec["rans"]),
            "pack_streams": avg(rec["pack"])}
    raw = N * L
    replay_ms = None
    try:
        codec.row_replay(d_in, N, L)                 # (first launch: untimed)
        replay_ms = codec.row_replay(d_in, N, L)
    except Exception as e:       # no replay instance for this configuration / cache organisation
        sys.stderr.write(f"bench: row replay not available ({e})\n")
    # algorithmic bytes per launch: decode reads C + writes raw; the model pass reads raw and hands
    # 4 B per nibble to the rANS pass; the rANS pass reads that spill and writes C
    alg = {dk: raw + coded_total, "encode_model_pass": raw + 8 * raw, "encode_rans_pass": 8 * raw + coded_total,
           "pack_streams": 2 * coded_total}
    res = {
        "elapsed": elapsed, "steps": steps, "ok": ok, "checked_vs_oracle": checked, "coded_total": coded_total,
        "encode_MBps": round(raw / 1e6 / (avg(rec["enc"]) / 1e3), 2), "decode_MBps": round(raw / 1e6 / (avg(rec["dec"]) / 1e3), 2),
        "kernel_ms": {k: round(v, 3) for k, v in kern.items()},
        "roofline": roofline_of(kern, alg, load_traffic(traffic_name or name, N, L), replay_ms),
        "byte_order": _order_record(codec), "placement_search_steps": search_steps,
        # HBM the codec holds besides the caller's buffers: the encoder's work arrays (+ the decoder's CDF tables)
        "encoder_work_bytes_per_input_byte": round(codec.info().scratch_bytes / raw, 2),
        "table_bytes": int(codec.info().table_bytes),
        "table_placement": codec.table_placement(),
    }
    return res, codec, outs


def run_decode_only(torch, da, po, args, dev, copies=4096, steps=None):
    """BASELINE configs[3]: testdata/random_then_unicode (291 949 B) cut into 5 blocks of at most 64 KiB, each coded ONCE
    on the CPU by the oracle as an independent stream under TestContextMixing options, the coded streams replicated
    x4096 in HBM, all decoded by the GPU and every copy compared with the original."""
    import lzma
    import numpy as np
    with lzma.open(os.path.join(ROOT, "tests", "golden", "random_then_unicode.xz")) as f:
        data = np.frombuffer(f.read(), dtype=np.uint8).copy()
    L = 65536
    blocks = [data[i:i + L] for i in range(0, data.size, L)]
    ocfg = po.config_context_mixing()
    coded = [po.lit_encode(ocfg, b) for b in blocks]
    nb = len(blocks)
    al = [(c.size + 3) & ~3 for c in coded]
    one = np.zeros(sum(al), dtype=np.uint8)
    pos = 0
    for c, a in zip(coded, al):
        one[pos:pos + c.size] = c; pos += a
    N = nb * copies
    d_coded = torch.from_numpy(one).to(dev).repeat(copies)
    in_off = torch.tensor(np.cumsum([0] + al[:-1]), dtype=torch.int64, device=dev)
    d_in_off = (in_off[None, :] + torch.arange(copies, device=dev, dtype=torch.int64)[:, None] * int(sum(al))).reshape(-1).contiguous()
    d_in_sz = torch.tensor([c.size for c in coded], dtype=torch.int32, device=dev).repeat(copies).contiguous()
    out_off1 = torch.tensor(np.cumsum([0] + [b.size for b in blocks[:-1]]), dtype=torch.int64, device=dev)
    d_out_off = (out_off1[None, :] + torch.arange(copies, device=dev, dtype=torch.int64)[:, None] * int(data.size)).reshape(-1).contiguous()
    d_out_sz = torch.tensor([b.size for b in blocks], dtype=torch.int32, device=dev).repeat(copies).contiguous()
    d_out = torch.zeros(copies * data.size, dtype=torch.uint8, device=dev)
    codec = da.LiteralCodec(da.config_context_mixing(), L, device=dev.index)
    if args.table_candidates:
        codec.tune_tables(args.table_candidates)
    orig = torch.from_numpy(data).to(dev)

    def decode():
        codec.decode_batch(d_coded, d_in_off, d_in_sz, N, L, d_out, out_offsets=d_out_off, out_sizes=d_out_sz)

    decode(); torch.cuda.synchronize()
    ok = bool((d_out.view(copies, data.size) == orig[None, :]).all().item()) and codec.status() == 0
    search_steps = settle_placement(codec, decode)
    torch.cuda.synchronize()
    d_out.zero_()
    steps = max(1, steps or args.steps)
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    ms, kms = [], []
    for _ in range(steps):
        ev0.record(); decode(); ev1.record(); torch.cuda.synchronize()
        ms.append(ev0.elapsed_time(ev1)); kms.append(codec.info().last_decode_ms)
    ok = ok and bool((d_out.view(copies, data.size) == orig[None, :]).all().item())
    raw, ctot = copies * int(data.size), copies * int(sum(c.size for c in coded))
    dk = codec.last_decode_kernel() or "lit_decode_kernel"
    kern = {dk: sum(kms) / len(kms)}
    replay_ms = None
    try:
        codec.row_replay(d_out, N, L, offsets=d_out_off, sizes=d_out_sz)
        replay_ms = codec.row_replay(d_out, N, L, offsets=d_out_off, sizes=d_out_sz)
    except Exception as e:
        sys.stderr.write(f"bench: row replay not available ({e})\n")
    res = {
        "workload": f"BASELINE configs[3]: testdata/random_then_unicode ({data.size} B) as {nb} independent streams (4 x 65536 + {blocks[-1].size} B), coded once by the "
                    f"oracle under TestContextMixing options, x{copies} copies = {N} streams resident in HBM, decode only, every copy compared",
        "bit_exact": ok, "streams": N, "steps": steps, "ms_per_step": round(sum(ms) / len(ms), 3),
        "value": round(raw / 1e6 / (sum(ms) / len(ms) / 1e3), 2), "unit": "MB/s decode",
        "compressed_ratio": round(ctot / raw, 4), "kernel_ms": {k: round(v, 3) for k, v in kern.items()},
        "roofline": roofline_of(kern, {dk: raw + ctot}, load_traffic("decode_only", N, L), replay_ms),
        "table_placement": codec.table_placement(), "byte_order": _order_record(codec), "placement_search_steps": search_steps,
    }
    codec.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=65536, help="independent 64 KiB streams per GPU")
    ap.add_argument("--total-streams", type=int, default=0, help="strong scaling (BASELINE configs[4]): this many streams in total, split over the GPUs "
                    "(131072 = 8 GiB); 0 = weak scaling with --streams per GPU")
    ap.add_argument("--block-len", type=int, default=65536)
    ap.add_argument("--config", choices=["all", "simple", "mixing", "decode_only", "simple_binary"], default="all",
                    help="all = configs[1] as the headline + configs[2] and configs[3] as sub-records (N = 1); a single name runs only that one "
                         "(simple_binary = configs[1]'s options on non-text input, a sub-record of `all`: for profiler passes)")
    ap.add_argument("--check-streams", type=int, default=4096, help="streams whose coded bytes are compared with the oracle before timing")
    ap.add_argument("--blocks-per-cu", type=float, default=0, help="persistent-grid override (tuning)")
    ap.add_argument("--cache-rows", type=int, default=-1, help="per-stream LDS row cache override of the STREAMING ENCODER pass (tuning; in the default build the decoder keeps "
                    "its own caches: --decoder-generation / divans_gpu_codec_set_decoder shape those)")
    ap.add_argument("--decoder-generation", type=int, default=0, help="decode kernel: 1 = lit_kernels.hip, 2 / 3 = lit_decode2.hip direct-mapped / 2-way caches (tuning; 0 = the codec's default)")
    ap.add_argument("--table-candidates", type=int, default=0, help="placements of the CDF tables a codec tries before it keeps the fastest: 0 = the library's own policy, i.e. what "
                    "every divans_gpu_codec_create caller gets (tables of 2 GiB and more: 12 placements, ONE PER DECODE CALL -- the untimed passes before the warm-up complete the search); "
                    "1 = take the first allocation as it comes; k = the eager form, all k on the first decode (divans_gpu_codec_tune_tables)")
    ap.add_argument("--encode-path", type=int, default=0, help="encoder model pass: 0 automatic, 1 streaming, 2 bucketed (tuning)")
    ap.add_argument("--bucket-batch", type=int, default=0, help="streams per launch sequence of the two-model bucketed pass (tuning; default 32768)")
    ap.add_argument("--split-cache", default="", help="HIGH,LOW rows of the split LDS caches of the streaming encoder pass (tuning; see --cache-rows)")
    ap.add_argument("--host-data", action="store_true", help="build the input with tests/workload.py on the host instead of on the GPU (same bytes; "
                    "keeps the tens of thousands of small torch kernels of the GPU generator out of profiler runs)")
    ap.add_argument("--input-cache", default="", help="with --host-data: directory that keeps the generated blocks between runs (profiler passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--diag-data", choices=["corpus", "zeros", "random", "repeat1k"], default="corpus",
                    help="diagnostics only: 'zeros' touches two CDF rows per stream (cache-resident ceiling)")
    args = ap.parse_args()

    env_world = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus > 1 and env_world == 0:
        spawn_ranks(args)          # does not return
    world = max(env_world, 1)
    if env_world and args.gpus not in (1, world):
        sys.exit(f"bench: --gpus {args.gpus} contradicts WORLD_SIZE={world}")

    import numpy as np
    import torch
    import divans_amd as da
    import pyoracle as po
    import workload
    from divans_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DIVANS_BENCH_SHARE_GPU=1: verification of the N > 1 path on a one-GPU box -- every rank on cuda:0, gloo rendezvous (RCCL refuses two
    # ranks on one device), divans_amd.sharding staging device tensors through the host.  The line says so; its numbers mean nothing.
    share_gpu = world > 1 and os.environ.get("DIVANS_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench: rank {rank} needs GPU {local_rank}; {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = args.block_len
    strong = args.total_streams > 0
    if strong:
        total_streams = args.total_streams          # BASELINE configs[4]: one fixed job split over the GPUs
        first, last = sharding.shard_bounds(total_streams, rank, world)
        N = last - first
        if N == 0:
            sys.exit(f"bench: --total-streams {total_streams} leaves rank {rank} of {world} without work")
    else:
        N = args.streams
        total_streams = N * world      # weak scaling: every rank owns N streams of the job's N * world
        first, last = sharding.shard_bounds(total_streams, rank, world)
        assert last - first == N
    corpus = workload.load_corpus()
    mg = None
    full = None
    if args.diag_data == "zeros":
        d_in = torch.zeros((N, L), dtype=torch.uint8, device=dev)
    elif args.diag_data == "random":
        d_in = torch.randint(0, 256, (N, L), dtype=torch.uint8, device=dev)
    else:
        corpus_t = torch.from_numpy(corpus).to(dev)
        if world > 1:
            # rank 0 holds the job's input and scatters contiguous stream ranges (SURVEY.md 8e / north_star)
            full = device_blocks(torch, corpus_t, 0, total_streams, L) if rank == 0 else None
            barrier(); t0 = time.perf_counter()
            d_in = sharding.scatter_streams(full, total_streams, L, dev)
            barrier(); scatter_s = time.perf_counter() - t0
            mg = {"scatter_ms": round(sharding.max_over_ranks(scatter_s, dev) * 1e3, 3)}
        elif args.host_data:
            # profiler runs: no generator kernels on the GPU; the blocks are built once on the host and kept in a file
            cache = os.path.join(args.input_cache, f"divans_blocks_{first}_{N}_{L}.npy") if args.input_cache else None
            d_in = torch.empty((N, L), dtype=torch.uint8, device=dev)
            if cache and os.path.exists(cache):
                host = np.load(cache, mmap_mode="r")
                for c0 in range(0, N, 4096):
                    d_in[c0:min(N, c0 + 4096)].copy_(torch.from_numpy(np.ascontiguousarray(host[c0:min(N, c0 + 4096)])))
            else:
                host = np.lib.format.open_memmap(cache, mode="w+", dtype=np.uint8, shape=(N, L)) if cache else None
                for c0 in range(0, N, 2048):
                    c1 = min(N, c0 + 2048)
                    blk = workload.make_blocks(corpus, first + c0, c1 - c0, block_len=L)
                    if host is not None:
                        host[c0:c1] = blk
                    d_in[c0:c1].copy_(torch.from_numpy(blk))
                if host is not None:
                    host.flush()
        else:
            d_in = device_blocks(torch, corpus_t, first, N, L)
        # the GPU generator must be the committed workload (tests/workload.py), byte for byte
        probe = workload.make_blocks(corpus, first, min(N, 3), block_len=L)
        assert bool((d_in[:probe.shape[0]].cpu().numpy() == probe).all()), "device workload generator differs from tests/workload.py"
        if args.diag_data == "repeat1k":
            d_in = d_in[:, :1024].repeat(1, L // 1024).contiguous()

    scaling = "strong" if strong else "weak"
    shard_text = (f"{total_streams} independent {L} B streams in total, split into contiguous ranges over the GPUs ({N} on this rank)" if strong
                  else f"{N} independent {L} B streams per GPU")

    sub_steps = min(args.steps, 5)       # the sub-records time fewer steps than the headline's K and warm up once (their "steps" says how many;
    sub_warm = min(args.warmup, 1)       # the placement search's untimed decodes come before either)

    def pair_record(name, d_in=d_in, traffic_name=None, byte_order=None, steps=None, warmup=None):
        """One encode+decode configuration on every rank: verify, time (max over ranks), at world > 1 gather the coded streams
        to rank 0 and check them there.  Returns (record for rank 0, bit-exact on all ranks)."""
        res, codec, outs = run_pair_config(torch, da, po, name, d_in, N, L, args, dev, barrier, traffic_name=traffic_name, byte_order=byte_order, steps=steps, warmup=warmup)
        elapsed = sharding.max_over_ranks(res["elapsed"], dev)
        coded_all, ok_count = sharding.sum_over_ranks([res["coded_total"], int(res["ok"])], dev)
        ok_all = ok_count == world
        K = res["steps"]
        m = None
        if world > 1:
            # variable-length gather of the coded streams to rank 0 (outside the timed region), checked there
            barrier(); t0 = time.perf_counter()
            blob, goffs, gsizes = sharding.gather_coded(outs["packed"], outs["sizes"].to(torch.int64), total_streams)
            barrier(); gather_s = sharding.max_over_ranks(time.perf_counter() - t0, dev)
            g_ok = 1
            if rank == 0:
                ocfg = po.config_simple() if name == "simple" else po.config_context_mixing()
                g_ok = int(int(gsizes.sum().item()) == coded_all)
                for r in range(world):
                    rb, re = sharding.shard_bounds(total_streams, r, world)
                    for i in (rb, (rb + re) // 2, re - 1):
                        ref = po.lit_encode(ocfg, full[i].cpu().numpy())
                        got = blob[int(goffs[i]):int(goffs[i]) + int(gsizes[i])].cpu().numpy()
                        g_ok &= int(got.size == ref.size and bool((got == ref).all()))
            g_ok, = sharding.sum_over_ranks([g_ok if rank == 0 else 1], dev)
            ok_all = ok_all and g_ok == world
            step_s = elapsed / K
            m = dict(mg)
            m.update({"gather_ms": round(gather_s * 1e3, 3), "gathered_bytes": int(coded_all), "gather_checked_on_rank0": bool(g_ok == world),
                      "rccl_world_size": world, "transport": ("gloo, all ranks on one GPU (DIVANS_BENCH_SHARE_GPU: path verification only; RCCL itself has only run at world 1 -- init, all_reduce, all_gather, barrier, "
                                    "a batched isend / irecv pair, tests/test_gpu_sharding.py -- every lease of rounds 1-5 was one GPU: shards have never moved between two GPUs)") if share_gpu else "rccl",
                      "scatter_gather_inclusive_MBps": round(total_streams * L / 1e6 / (step_s + mg["scatter_ms"] / 1e3 + gather_s), 2)})
            t = torch.zeros((3, world), dtype=torch.float64, device="cpu" if share_gpu else dev)
            t[0, rank] = res["elapsed"] / K * 1e3; t[1, rank] = res["roofline"]["frac"]; t[2, rank] = res["roofline"]["achieved"]
            dist.all_reduce(t)
            m["code_ms"] = round(step_s * 1e3, 3)               # encode + pack + decode of every rank's shard, max over ranks (= ms_per_step)
            m["per_rank_ms_per_step"] = [round(float(x), 3) for x in t[0].tolist()]
            # north_star's "achieved-HBM-fraction curve": the dominant (decode) kernel's algorithmic GB/s over 8 TB/s, rank by rank
            m["per_rank_roofline_frac"] = [round(float(x), 6) for x in t[1].tolist()]
            m["per_rank_roofline_achieved_GBps"] = [round(float(x), 3) for x in t[2].tolist()]
            m["roofline_kernel"] = res["roofline"]["kernel"]
        codec.close()
        del outs
        torch.cuda.empty_cache()
        total_bytes = total_streams * L
        rec = {"value": round(total_bytes / 1e6 / (elapsed / K), 2), "steps": K, "ms_per_step": round(elapsed * 1e3 / K, 3),
               "bit_exact": bool(ok_all), "checked_vs_oracle": res["checked_vs_oracle"], "compressed_ratio": round(coded_all / float(total_bytes), 4),
               "encode_MBps": res["encode_MBps"], "decode_MBps": res["decode_MBps"], "kernel_ms": res["kernel_ms"], "roofline": res["roofline"],
               "encoder_work_bytes_per_input_byte": res["encoder_work_bytes_per_input_byte"], "table_bytes": res["table_bytes"],
               "table_placement": res["table_placement"], "byte_order": res["byte_order"], "placement_search_steps": res["placement_search_steps"]}
        if m:
            rec["multi_gpu"] = m
        return rec, ok_all

    head_name = "simple" if args.config in ("all", "simple") else ("mixing" if args.config == "mixing" else None)
    line = None
    if head_name:
        rec, ok_all = pair_record(head_name)
        if rank == 0:
            cfg_tag = "BASELINE configs[1]" if head_name == "simple" else "BASELINE configs[2]"
            if strong:
                cfg_tag = "BASELINE configs[4] share, options of " + cfg_tag
            cfg_text = ("TestSimple: stride 1, context map off" if head_name == "simple"
                        else "TestContextMixing: context map + dynamic_context_mixing=2")
            line = {
                "metric": "MB/s encode+decode per GPU, 64 KiB metablocks; bit-exact vs CPU",
                "value": rec["value"], "unit": "MB/s", "n_gpus": world, "steps": rec["steps"], "warmup": args.warmup,
                "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "dtype": "u16/u64 integer", "data": "synthetic",
                "config": {"workload": f"{cfg_tag}: {cfg_text}; {shard_text} cut from alice29||asyoulik (stride 4099, 1% xorshift64* perturbation)",
                           "streams_per_gpu": N, "total_streams": total_streams, "block_bytes": L,
                           "sharding": "contiguous stream ranges per rank; no collective inside the timed region"},
                "bit_exact": rec["bit_exact"], "bit_exact_against": "in-repo C oracle (restatement of the reference CPU path; its framing, CRC, CMD and LIT coders are pinned on the one compressed "
                                     "vector the reference tree holds, wasm/wasm.html:98-107, under that older build's wire variant; context maps, mixing, the "
                                     "65 536-symbol chunk seam and HEAD's two PredictionMode prior rows have NO reference-built bytes behind them -- "
                                     f"tests/golden/make_reference_vectors.rs needs cargo): coded bytes of {rec['checked_vs_oracle']} streams per rank + exact round trip of all",
                "compressed_ratio": rec["compressed_ratio"],
                "encode_MBps": rec["encode_MBps"], "decode_MBps": rec["decode_MBps"],
                "kernel_ms": rec["kernel_ms"], "roofline": rec["roofline"],
                "encoder_work_bytes_per_input_byte": rec["encoder_work_bytes_per_input_byte"], "table_bytes": rec["table_bytes"],
                "table_placement": rec["table_placement"], "byte_order": rec["byte_order"],
                "untimed_passes": {"verification": 0 if args.no_verify else 1, "placement_search": rec["placement_search_steps"], "warmup": args.warmup},
            }
            if "multi_gpu" in rec:
                line["multi_gpu"] = rec["multi_gpu"]
        if not ok_all:
            if rank == 0:
                print(json.dumps(line))
            sys.exit("bench: GPU output is NOT bit-exact / round-trip failed")

    if args.config in ("all", "mixing", "decode_only", "simple_binary"):
        sub = {}
        if args.config == "all":
            # configs[2] on the same streams; at world > 1 this is configs[4]'s second option set, with its own scatter/gather record
            r2, ok2 = pair_record("mixing", steps=sub_steps, warmup=sub_warm)
            r2 = dict(r2)
            r2.update({"workload": "BASELINE configs[2]: same streams, TestContextMixing: context map cm[i]=i&63, utf8, block type 1, dynamic_context_mixing=2 (BASELINE configs[2])",
                       "unit": "MB/s encode+decode"})
            sub["mixing"] = r2
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                sub["mixing"]["cpu_baseline"] = cpu_baseline("mixing", workload, corpus, L)
        if world == 1 and args.config in ("all", "simple_binary") and args.diag_data == "corpus":
            # configs[1]'s options on input that is not English text: the same cut (stride 4099, 1 % perturbation) out of testdata/
            # random_then_unicode (random bytes, then UTF-8 in several scripts); the decoder's table order is learned from the data here as in the headline
            import lzma
            with lzma.open(os.path.join(ROOT, "tests", "golden", "random_then_unicode.xz")) as f:
                rtu_t = torch.from_numpy(np.frombuffer(f.read(), dtype=np.uint8).copy()).to(dev)
            if args.host_data:      # profiler passes: no generator kernels on the GPU (tests/workload.py builds the same bytes on the host)
                rtu_h = rtu_t.cpu().numpy()
                d_bin = torch.empty((N, L), dtype=torch.uint8, device=dev)
                for c0 in range(0, N, 2048):
                    c1 = min(N, c0 + 2048)
                    d_bin[c0:c1].copy_(torch.from_numpy(workload.make_blocks(rtu_h, first + c0, c1 - c0, block_len=L)))
            else:
                d_bin = device_blocks(torch, rtu_t, first, N, L)
            rb, okb = pair_record("simple", d_in=d_bin, traffic_name="simple_binary", steps=sub_steps if args.config == "all" else None, warmup=sub_warm if args.config == "all" else None)
            rb = dict(rb)
            rb.update({"workload": f"BASELINE configs[1] options on non-text input: {N} x {L} B streams cut from testdata/random_then_unicode (stride 4099, 1% perturbation: random bytes and multi-script "
                                   "UTF-8), TestSimple options as the headline, the decoder's table order learned from this data like the headline's (byte_order)", "unit": "MB/s encode+decode"})
            sub["simple_binary"] = rb
            del d_bin
            torch.cuda.empty_cache()
        if world == 1 and args.config in ("all", "decode_only"):
            sub["decode_only"] = run_decode_only(torch, da, po, args, dev, steps=sub_steps)
            if not args.no_cpu_baseline:
                import lzma
                with lzma.open(os.path.join(ROOT, "tests", "golden", "random_then_unicode.xz")) as f:
                    rtu = np.frombuffer(f.read(), dtype=np.uint8).copy()
                whole = rtu[:(rtu.size // 65536) * 65536].reshape(-1, 65536)     # the four whole 64 KiB blocks of the file
                cb = cpu_baseline("mixing", workload, corpus, 65536, sample_blocks=lambda n: np.ascontiguousarray(whole[np.arange(n) % whole.shape[0]]),
                                  decode_only=True, per_thread=96)
                cb["sample"] = "decode direction of: " + cb["sample"].replace("streams of the same workload", "streams (the four whole 64 KiB blocks of testdata/random_then_unicode, repeated)")
                sub["decode_only"]["cpu_baseline"] = cb
        if rank == 0:
            if line is None:
                line = {"metric": "MB/s encode+decode per GPU, 64 KiB metablocks; bit-exact vs CPU", "value": None, "unit": "MB/s", "n_gpus": world,
                        "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "data": "synthetic", "scaling": scaling,
                        "config": {"workload": f"sub-config run: {args.config}"}}
            line["configs"] = sub
        bad = [k for k, v in sub.items() if not v["bit_exact"]]
        if bad:
            if rank == 0:
                print(json.dumps(line))
            sys.exit(f"bench: sub-config {bad} is NOT bit-exact")

    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and head_name:   # the host-core baseline is a single-GPU-run item
            line["cpu_baseline"] = cpu_baseline(head_name, workload, corpus, L)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
