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
