"""N>1 data path on CPU: gloo ranks run exactly the functions bench.py runs on RCCL --
rank 0 scatters contiguous stream ranges, every rank codes its shard (the oracle is the stand-in coder: this is a
test), sizes are exchanged and the coded bytes gathered to rank 0 -- and the gathered blob must be byte-identical to
what one process produces.  World sizes 2 and 3, stream counts that do not divide evenly, both BASELINE configurations."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = 2048


def _code_and_pack(po, cfg, blocks):
    """what a rank does with its shard: code every stream, pack them back to back on 4-byte boundaries"""
    from divans_amd import sharding
    coded = [po.lit_encode(cfg, blk) for blk in blocks]
    sizes = torch.tensor([c.size for c in coded], dtype=torch.int64)
    al = sharding._aligned(sizes)
    packed = torch.zeros(int(al.sum()), dtype=torch.uint8)
    pos = 0
    for c, a in zip(coded, al.tolist()):
        packed[pos:pos + c.size] = torch.from_numpy(c)
        pos += a
    return packed, sizes


def _config(po, name):
    return po.config_simple() if name == "simple" else po.config_context_mixing()


def _worker(rank, world, port, n_streams, cfg_name, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle as po
    import workload
    from divans_amd import sharding
    sharding.MAX_MESSAGE_BYTES = 3001        # shards travel as several messages each (the 1 GiB limit, scaled to the test)
    cpu = torch.device("cpu")
    full = None
    if rank == 0:   # rank 0 holds the corpus
        full = torch.from_numpy(workload.make_blocks(workload.load_corpus(), 0, n_streams, block_len=BLOCK))
    mine = sharding.scatter_streams(full, n_streams, BLOCK, cpu)
    b, e = sharding.shard_bounds(n_streams, rank, world)
    assert tuple(mine.shape) == (e - b, BLOCK)
    packed, sizes = _code_and_pack(po, _config(po, cfg_name), mine.numpy())
    blob, offs, allsizes = sharding.gather_coded(packed, sizes, n_streams)
    total, = sharding.sum_over_ranks([int(sizes.sum())], cpu)
    slowest = sharding.max_over_ranks(0.25 * (rank + 1), cpu)
    dist.barrier()
    if rank == 0:
        q.put((blob.numpy().tobytes(), offs.tolist(), allsizes.tolist(), total, slowest))
    else:
        assert blob is None and offs is None
    dist.destroy_process_group()


def test_shard_bounds():
    from divans_amd import sharding
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            bounds = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(w - 1))
            lens = [e - b for b, e in bounds]
            assert max(lens) - min(lens) <= 1


@pytest.mark.parametrize("world,n_streams,cfg_name", [(2, 13, "simple"), (3, 10, "mixing"), (3, 2, "simple"), (2, 9, "mixing")])
def test_scatter_code_gather_matches_single_process(world, n_streams, cfg_name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 131 + n_streams) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, cfg_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob, offs, allsizes, total, slowest = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    import workload
    blocks = workload.make_blocks(workload.load_corpus(), 0, n_streams, block_len=BLOCK)
    ref_packed, ref_sizes = _code_and_pack(po, _config(po, cfg_name), blocks)
    assert allsizes == ref_sizes.tolist() and total == int(ref_sizes.sum())
    assert blob == ref_packed.numpy().tobytes()
    # every stream sits where the offsets say
    for i in (0, n_streams // 2, n_streams - 1):
        got = np.frombuffer(blob, dtype=np.uint8)[offs[i]:offs[i] + allsizes[i]]
        assert (got == po.lit_encode(_config(po, cfg_name), blocks[i])).all()
    assert abs(slowest - 0.25 * world) < 1e-9


def test_bench_refuses_more_gpus_than_present():
    """`python bench.py --gpus 2` without 2 GPUs must fail loudly, not run one rank and print n_gpus 1."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2 GPUs: the refusal path cannot be exercised")
    assert r.returncode != 0
    assert "needs 2 GPUs" in (r.stderr + r.stdout)
