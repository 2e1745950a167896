"""N>1 path on CPU: two gloo ranks shard a batch of streams, code their shards (oracle as the stand-in coder,
this is a test), and gather the per-stream sizes; the union must equal the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle as po
    import workload
    from divans_amd import sharding
    corpus = workload.load_corpus()
    b, e = sharding.shard_bounds(n_streams, rank, world)
    blocks = workload.make_blocks(corpus, b, e - b, block_len=2048)
    cfg = po.config_simple()
    sizes = torch.tensor([po.lit_encode(cfg, blk).size for blk in blocks], dtype=torch.int64)
    allsizes = sharding.gather_stream_sizes(sizes, n_streams)
    total, = sharding.sum_over_ranks([int(sizes.sum())], torch.device("cpu"))
    slowest = sharding.max_over_ranks(0.25 * (rank + 1), torch.device("cpu"))
    dist.barrier()
    if rank == 0:
        q.put((allsizes.tolist(), total, slowest))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    from divans_amd import sharding
    # bounds: contiguous, disjoint, complete, balanced
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            bounds = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(w - 1))
            lens = [e - b for b, e in bounds]
            assert max(lens) - min(lens) <= 1
    n_streams = 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_streams, q)) for r in range(2)]
    for p in procs:
        p.start()
    allsizes, total, slowest = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    import workload
    corpus = workload.load_corpus()
    blocks = workload.make_blocks(corpus, 0, n_streams, block_len=2048)
    ref = [int(po.lit_encode(po.config_simple(), blk).size) for blk in blocks]
    assert allsizes == ref and total == sum(ref) and abs(slowest - 0.5) < 1e-9
