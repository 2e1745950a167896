"""Multi-GPU partitioning of a batch of independent literal streams.

Every 64 KiB stream is its own divans stream (own priors, own rANS states, own output): there is no exchange
step in the data path, so GPU g of G simply owns a contiguous range of streams.  torch.distributed (RCCL on
ROCm, gloo in the CPU tests) is used for the barrier, the max-over-ranks timing and for gathering the per-stream
coded sizes to rank 0 -- never for the payload of the timed region."""
import torch
import torch.distributed as dist


def shard_bounds(n_streams, rank, world):
    """[begin, end) of the streams rank owns: contiguous, sizes differ by at most one, covers everything once."""
    base, extra = divmod(int(n_streams), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, device):
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def gather_stream_sizes(local_sizes, n_streams):
    """Per-stream coded sizes of the whole job on every rank (variable shard lengths => padded all_gather)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return local_sizes.clone()
    longest = max(shard_bounds(n_streams, r, world)[1] - shard_bounds(n_streams, r, world)[0] for r in range(world))
    pad = torch.zeros(longest, dtype=local_sizes.dtype, device=local_sizes.device)
    pad[:local_sizes.numel()] = local_sizes
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        b, e = shard_bounds(n_streams, r, world)
        out.append(parts[r][:e - b])
    assert out[rank].numel() == local_sizes.numel()
    return torch.cat(out)
