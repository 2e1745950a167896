"""Multi-GPU partitioning of a batch of independent literal streams.

Every 64 KiB stream is its own divans stream (own priors, own rANS states, own output): there is no exchange
step inside the coding path, so GPU g of G simply owns a contiguous range of streams.  What does move between
GPUs is what BASELINE.json's north_star names: the input ranges go out from rank 0 (`scatter_streams`), the
per-stream coded sizes are exchanged (`gather_stream_sizes`) and the coded bytes come back to rank 0 as one
variable-length gather (`gather_coded`).  All of it is point-to-point `torch.distributed` send/recv batched into
one group per step (RCCL over xGMI on ROCm: rank 0 talks to every peer over its own link, nothing ring-shaped;
gloo in the CPU tests), plus all_reduce / all_gather for the scalars."""
import torch
import torch.distributed as dist


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _staged(t):
    """gloo moves host memory only: device tensors are staged through the host (the 2-ranks-on-one-GPU test and any CPU-only
    fabric); with RCCL the device tensors go out as they are."""
    return t.is_cuda and _world() > 1 and dist.get_backend() == "gloo"


def shard_bounds(n_streams, rank, world):
    """[begin, end) of the streams rank owns: contiguous, sizes differ by at most one, covers everything once."""
    base, extra = divmod(int(n_streams), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, device):
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if _world() > 1:
        if _staged(t):
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if _world() > 1:
        if _staged(t):
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def gather_stream_sizes(local_sizes, n_streams):
    """Per-stream coded sizes of the whole job on every rank (variable shard lengths => padded all_gather)."""
    world, rank = _world(), _rank()
    if world == 1:
        return local_sizes.clone()
    longest = max(shard_bounds(n_streams, r, world)[1] - shard_bounds(n_streams, r, world)[0] for r in range(world))
    dev = local_sizes.device
    wire = torch.device("cpu") if _staged(local_sizes) else dev
    pad = torch.zeros(longest, dtype=local_sizes.dtype, device=wire)
    pad[:local_sizes.numel()] = local_sizes.to(wire)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        b, e = shard_bounds(n_streams, r, world)
        out.append(parts[r][:e - b])
    assert out[rank].numel() == local_sizes.numel()
    return torch.cat(out).to(dev)


MAX_MESSAGE_BYTES = 1 << 30   # a shard (4 GiB at BASELINE configs[1]) travels as several messages: no 32-bit count anywhere on the way


def _run_p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def _pieces(t):
    """`t` (contiguous) as a list of flat views of at most MAX_MESSAGE_BYTES each; sender and receiver cut alike."""
    flat = t.reshape(-1)
    step = max(1, MAX_MESSAGE_BYTES // max(1, flat.element_size()))
    return [flat[i:i + step] for i in range(0, flat.numel(), step)]


def scatter_streams(all_streams, n_streams, stream_len, device, dtype=torch.uint8):
    """Rank 0 holds `all_streams` ([n_streams, stream_len], on `device`); every rank returns its contiguous shard
    [shard_bounds) as a [count, stream_len] tensor.  One send per peer, posted as a single batch."""
    world, rank = _world(), _rank()
    b, e = shard_bounds(n_streams, rank, world)
    if world == 1:
        return all_streams[b:e]
    if rank == 0:
        ops = []
        for r in range(1, world):
            rb, re = shard_bounds(n_streams, r, world)
            if re > rb:
                shard = all_streams[rb:re].contiguous()
                if _staged(shard):
                    shard = shard.cpu()
                ops.extend(dist.P2POp(dist.isend, piece, r) for piece in _pieces(shard))
        _run_p2p(ops)
        return all_streams[b:e]
    staged = torch.device(device).type == "cuda" and dist.get_backend() == "gloo"
    mine = torch.empty((e - b, stream_len), dtype=dtype, device="cpu" if staged else device)
    if e > b:
        _run_p2p([dist.P2POp(dist.irecv, piece, 0) for piece in _pieces(mine)])
    return mine.to(device) if staged else mine


def gather_coded(local_packed, local_sizes, n_streams):
    """Variable-length gather of the coded streams to rank 0.

    `local_packed`: this rank's coded streams back to back (stream i of the shard at the exclusive prefix sum of the
    sizes rounded up to `align`); `local_sizes`: their byte sizes (int64/int32).  Returns on rank 0
    (blob, offsets, sizes) covering all n_streams in job order, on the other ranks (None, None, sizes):
    first the size exchange (all_gather), then one recv per peer sized from it."""
    world, rank = _world(), _rank()
    sizes = gather_stream_sizes(local_sizes.to(torch.int64), n_streams)
    if world == 1:
        offs = torch.cumsum(_aligned(sizes), 0) - _aligned(sizes)
        return local_packed, offs, sizes
    al = _aligned(sizes)
    offs = torch.cumsum(al, 0) - al
    shard_bytes = []
    for r in range(world):
        rb, re = shard_bounds(n_streams, r, world)
        shard_bytes.append(int(al[rb:re].sum().item()))
    staged = _staged(local_packed)
    if rank == 0:
        blob = torch.empty(sum(shard_bytes), dtype=torch.uint8, device="cpu" if staged else local_packed.device)
        blob[:shard_bytes[0]] = local_packed[:shard_bytes[0]].to(blob.device)
        ops, pos = [], shard_bytes[0]
        for r in range(1, world):
            if shard_bytes[r]:
                ops.extend(dist.P2POp(dist.irecv, piece, r) for piece in _pieces(blob[pos:pos + shard_bytes[r]]))
            pos += shard_bytes[r]
        _run_p2p(ops)
        return (blob.to(local_packed.device) if staged else blob), offs, sizes
    if shard_bytes[rank]:
        mine = local_packed[:shard_bytes[rank]].contiguous()
        _run_p2p([dist.P2POp(dist.isend, piece, 0) for piece in _pieces(mine.cpu() if staged else mine)])
    return None, None, sizes


PACK_ALIGN = 4   # divans_gpu_pack_streams places every coded stream on a 4-byte boundary (coded streams are whole words)


def _aligned(sizes):
    return (sizes + (PACK_ALIGN - 1)) // PACK_ALIGN * PACK_ALIGN
