"""Utterance sharding over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The hot path shards by utterance and needs NO data-path collective: every rank runs the complete NMF / AuxIVA /
ILRMA loop on its own block of utterances.  RCCL is used only at the edges:
  * scatter_utterances : root -> ranks, the mixtures X (skipped when every rank loads / generates its own),
  * gather_utterances  : ranks -> root, the separated outputs Y,
  * max_over_ranks     : the MAX of a per-rank timing (bench.py contract).
Root <-> 7 peers is 7 concurrent point-to-point xGMI links, so scatter/gather are issued as one
`dist.scatter` / `dist.gather` (grouped send/recv inside RCCL), never as a ring.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Static contiguous block partition: item i belongs to exactly one rank; sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank: %r/%r" % (world_size, rank))
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world_size):
    return [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0]
            for r in range(world_size)]


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world_size, local_rank).  A single process (WORLD_SIZE unset or 1) needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def _as_real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def scatter_utterances(x_all, n_items, item_shape, dtype, device, src=0):
    """Root holds x_all (n_items, *item_shape); every rank returns its own block (n_local, *item_shape)."""
    rank, world = _world()
    lo, hi = shard_range(n_items, world, rank)
    if world == 1:
        return x_all[lo:hi].to(device)
    out = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=device)
    # equal-size requirement of dist.scatter: pad every block to the largest shard
    sizes = shard_sizes(n_items, world)
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(item_shape), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            a, b = shard_range(n_items, world, r)
            c = torch.zeros((pad,) + tuple(item_shape), dtype=dtype, device=device)
            c[: b - a] = x_all[a:b].to(device)
            chunks.append(_as_real(c).contiguous())
    dist.scatter(_as_real(buf), scatter_list=chunks, src=src)
    out.copy_(buf[: hi - lo])
    return out


def gather_utterances(y_local, n_items, dst=0):
    """Inverse of scatter_utterances: root returns (n_items, ...) in the original utterance order, others None."""
    rank, world = _world()
    if world == 1:
        return y_local
    sizes = shard_sizes(n_items, world)
    pad = max(sizes)
    item_shape = tuple(y_local.shape[1:])
    buf = torch.zeros((pad,) + item_shape, dtype=y_local.dtype, device=y_local.device)
    buf[: y_local.shape[0]] = y_local
    recv = None
    if rank == dst:
        recv = [torch.empty_like(_as_real(buf)) for _ in range(world)]
    dist.gather(_as_real(buf).contiguous(), gather_list=recv, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        t = recv[r]
        t = torch.view_as_complex(t) if y_local.is_complex() else t
        parts.append(t[: sizes[r]])
    return torch.cat(parts, dim=0)


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device="cpu"):
    """MAX of a Python float over all ranks (the per-rank elapsed time of a timed region)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_sharded(process_fn, x_all, n_items, item_shape, dtype, device, gather=True):
    """scatter -> process_fn(local block) -> gather.  process_fn maps (n_local, *item_shape) to (n_local, ...)."""
    x_local = scatter_utterances(x_all, n_items, item_shape, dtype, device)
    y_local = process_fn(x_local)
    return gather_utterances(y_local, n_items) if gather else y_local
