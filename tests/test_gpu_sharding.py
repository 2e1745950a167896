"""The N > 1 data path with the HIP coder producing the shards: scatter of stream ranges from rank 0,
`divans_gpu_lit_encode_batch` + `divans_gpu_pack_streams` on every rank, size exchange and variable-length gather of the
coded bytes to rank 0 -- on CUDA(=HIP) tensors.  World 1 in-process; world 2 as two processes sharing the one GPU of the test
box, rendezvous over gloo (RCCL refuses two ranks on one device), which `divans_amd.sharding` serves by staging the device
tensors through the host.  Both configurations of BASELINE configs[4] (TestSimple and TestContextMixing)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = 4096

pytestmark = pytest.mark.gpu


def _rank_job(rank, world, n_streams, cfg_name):
    """what bench.py does per rank, reduced: returns on rank 0 (blob bytes, offsets, sizes) of the whole job"""
    import divans_amd as da
    import workload
    from divans_amd import sharding
    dev = torch.device("cuda", 0)
    full = None
    if rank == 0:
        full = torch.from_numpy(workload.make_blocks(workload.load_corpus(), 0, n_streams, block_len=BLOCK)).to(dev)
    mine = sharding.scatter_streams(full, n_streams, BLOCK, dev)
    b, e = sharding.shard_bounds(n_streams, rank, world)
    assert mine.is_cuda and tuple(mine.shape) == (e - b, BLOCK)
    cfg = da.config_simple() if cfg_name == "simple" else da.config_context_mixing()
    codec = da.LiteralCodec(cfg, BLOCK)
    n = e - b
    outs = codec.alloc_encode_outputs(n, BLOCK)
    codec.encode_batch(mine.contiguous(), n, BLOCK, outs)
    packed, poff, ptotal = codec.pack(outs, n)
    torch.cuda.synchronize()
    assert codec.status() == 0
    blob, offs, sizes = sharding.gather_coded(packed, outs["sizes"].to(torch.int64), n_streams)
    total, = sharding.sum_over_ranks([int(outs["sizes"].to(torch.int64).sum().item())], dev)
    codec.close()
    if rank == 0:
        assert blob.is_cuda
        return blob.cpu().numpy().tobytes(), offs.cpu().tolist(), sizes.cpu().tolist(), total
    assert blob is None
    return None


def _check_against_oracle(result, n_streams, cfg_name):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    import workload
    blob, offs, sizes, total = result
    ocfg = po.config_simple() if cfg_name == "simple" else po.config_context_mixing()
    blocks = workload.make_blocks(workload.load_corpus(), 0, n_streams, block_len=BLOCK)
    arr = np.frombuffer(blob, dtype=np.uint8)
    assert total == sum(sizes)
    pos = 0
    for i in range(n_streams):
        ref = po.lit_encode(ocfg, blocks[i])
        assert offs[i] == pos and sizes[i] == ref.size, i           # back to back on 4-byte boundaries, job order
        assert (arr[pos:pos + ref.size] == ref).all(), i
        pos += (ref.size + 3) & ~3


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_world1_scatter_code_gather_on_device(cfg_name):
    _check_against_oracle(_rank_job(0, 1, 37, cfg_name), 37, cfg_name)


def _worker(rank, world, port, n_streams, cfg_name, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from divans_amd import sharding
    sharding.MAX_MESSAGE_BYTES = 50000       # shards travel as several messages each
    res = _rank_job(rank, world, n_streams, cfg_name)
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg_name", ["simple", "mixing"])
def test_world2_hip_coder_shards_through_gloo_staging(cfg_name):
    world, n_streams = 2, 45
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() * 7 + len(cfg_name)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, cfg_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _check_against_oracle(res, n_streams, cfg_name)


def _rccl_world1(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as e:      # an image without a working RCCL: nothing to test here
        q.put(("skip", repr(e))); return
    dev = torch.device("cuda", 0)
    out = {}
    t = torch.tensor([3.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); out["max"] = float(t.item())
    v = torch.arange(5, dtype=torch.int64, device=dev); dist.all_reduce(v, op=dist.ReduceOp.SUM); out["sum"] = v.tolist()
    parts = [torch.empty(7, dtype=torch.int32, device=dev)]; dist.all_gather(parts, torch.arange(7, dtype=torch.int32, device=dev)); out["gather"] = parts[0].tolist()
    dist.barrier()
    try:    # the variable-length gather's transport: batched point-to-point, here to the only rank there is
        src = torch.arange(1 << 20, dtype=torch.int64, device=dev).to(torch.uint8); dst = torch.zeros_like(src)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
            w.wait()
        torch.cuda.synchronize()
        out["p2p_self"] = bool(torch.equal(src, dst))
    except Exception as e:
        out["p2p_self"] = "unsupported: " + repr(e)[:200]
    dist.destroy_process_group()
    q.put(("ok", out))


def test_rccl_backend_runs_on_this_box_at_world_1():
    """The "nccl" (= RCCL) branch of bench.py / sharding.py needs more than one GPU to move shards, and every lease of rounds 1-5 was one GPU.
    What one GPU can show: the backend initialises on this ROCm image, takes device tensors without staging, and the collective calls the
    N > 1 path makes (all_reduce MAX / SUM, all_gather, barrier, a batched isend / irecv pair) run through it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1, args=(32500 + os.getpid() % 1500, q))
    p.start()
    kind, res = q.get(timeout=300)
    p.join(timeout=120)
    if kind == "skip":
        pytest.skip("no working RCCL here: " + res)
    assert res["max"] == 3.5 and res["sum"] == [0, 1, 2, 3, 4] and res["gather"] == list(range(7)), res
    assert res["p2p_self"] is True or str(res["p2p_self"]).startswith("unsupported"), res
    print("RCCL at world 1:", res)
