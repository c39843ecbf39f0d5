"""Utterance sharding over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The hot path shards by utterance and needs NO data-path collective: every rank runs the complete NMF / AuxIVA /
ILRMA loop on its own block of utterances (ref src/bss/ilrma.py:203-273 -- each `__call__`
owns all of its state).  RCCL is used only at the edges:
  * scatter_utterances : root -> ranks, the mixtures X (skipped when every rank loads / generates its own),
  * gather_utterances  : ranks -> root, the separated outputs Y,
  * max_over_ranks     : the MAX of a per-rank timing (bench.py contract).
The edges are point-to-point: root <-> 7 peers is 7 concurrent xGMI links, so both are ONE grouped batch of
send/recv operations on VIEWS of the root's array (`dist.batch_isend_irecv` = ncclGroupStart/End around
ncclSend/ncclRecv): no ring, no padded staging copies, ragged shards need no padding.

Communication deliberately lives on the host side (torch's RCCL binding), not in the C-ABI: include/assx.h is the
device boundary of ONE rank, and every entry point there is communication-free.

`all_gather_ordered_sum` is the deterministic reduction the F-sharded single-utterance mode
builds on.
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
            if local_rank >= torch.cuda.device_count():
                raise RuntimeError("LOCAL_RANK=%d but only %d GPU(s) are visible: one process per GPU is required"
                                   % (local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def _as_real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def _run_p2p(ops):
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def scatter_utterances(x_all, n_items, item_shape, dtype, device, src=0):
    """Root holds x_all (n_items, *item_shape) on `device`; every rank returns its own block (n_local, *item_shape).
    One grouped batch of sends of contiguous row-blocks (views of x_all); the root's own block is a view, not a copy."""
    rank, world = _world()
    lo, hi = shard_range(n_items, world, rank)
    if world == 1:
        return x_all[lo:hi].to(device=device, dtype=dtype)
    if rank == src:
        x_all = x_all.to(device=device, dtype=dtype).contiguous()
        ops = []
        for r in range(world):
            a, b = shard_range(n_items, world, r)
            if r != src and b > a:
                ops.append(dist.P2POp(dist.isend, _as_real(x_all[a:b]), r))
        _run_p2p(ops)
        return x_all[lo:hi]
    out = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=device)
    if hi > lo:
        _run_p2p([dist.P2POp(dist.irecv, _as_real(out), src)])
    return out


def gather_utterances(y_local, n_items, dst=0):
    """Inverse of scatter_utterances: root returns (n_items, ...) in the original utterance order, others None.
    Peers send their block once; the root receives every block straight into its slice of the result."""
    rank, world = _world()
    if world == 1:
        return y_local
    lo, hi = shard_range(n_items, world, rank)
    y_local = y_local.contiguous()
    if rank != dst:
        if hi > lo:
            _run_p2p([dist.P2POp(dist.isend, _as_real(y_local), dst)])
        return None
    out = torch.empty((n_items,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
    out[lo:hi] = y_local
    ops = []
    for r in range(world):
        a, b = shard_range(n_items, world, r)
        if r != dst and b > a:
            ops.append(dist.P2POp(dist.irecv, _as_real(out[a:b]), r))
    _run_p2p(ops)
    return out


def _is_nccl():
    return dist.is_initialized() and dist.get_backend() == "nccl"


def barrier(device=None):
    """Every rank's queued GPU work is finished, then every rank has arrived (timing bracket of bench.py)."""
    cuda = device is not None and torch.device(device).type == "cuda"
    if cuda:
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        if cuda and _is_nccl():
            dist.barrier(device_ids=[torch.device(device).index])
        else:
            dist.barrier()
    if cuda:
        torch.cuda.synchronize(device)


def max_over_ranks(value, device="cpu"):
    """MAX of a Python float over all ranks (the per-rank elapsed time of a timed region)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if _is_nccl() else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def warm_up_edges(device, src=0):
    """One tiny grouped send/recv root <-> every peer: RCCL builds its point-to-point channels on first use (seconds),
    which must not be billed to the first timed scatter."""
    rank, world = _world()
    if world == 1:
        return
    t = torch.zeros(8, dtype=torch.float32, device=device)
    for direction in (0, 1):
        ops = []
        if rank == src:
            for r in range(world):
                if r != src:
                    ops.append(dist.P2POp(dist.isend if direction == 0 else dist.irecv, t.clone(), r))
        else:
            ops.append(dist.P2POp(dist.irecv if direction == 0 else dist.isend, t.clone(), src))
        _run_p2p(ops)
    barrier(device if torch.device(device).type == "cuda" else None)


def run_sharded(process_fn, x_all, n_items, item_shape, dtype, device, gather=True):
    """scatter -> process_fn(local block) -> gather.  process_fn maps (n_local, *item_shape) to (n_local, ...)."""
    x_local = scatter_utterances(x_all, n_items, item_shape, dtype, device)
    y_local = process_fn(x_local)
    return gather_utterances(y_local, n_items) if gather else y_local


def separate_sharded(model_factory, x_all, n_items, item_shape, dtype, device, iteration=100, init_fn=None,
                     gather=True, comm_device=None):
    """Config 5: `n_items` independent utterances separated by the reference-surface classes, a contiguous block per
    rank as ONE batched launch sequence (leading utterance axis B = n_local).

    model_factory() -> a fresh model (GaussILRMA / AuxLaplaceIVA / ...).  init_fn(model, lo, hi) may assign the
    initial state of utterances lo..hi-1 (the reference draws it from the global NumPy RNG per call; a sharded run
    must draw per UTTERANCE so that the result does not depend on the partition).  comm_device: where the edge
    buffers live (default = `device`, i.e. RCCL on HBM buffers; "cpu" stages the edges through host memory for a
    gloo group).  Returns (Y on the root | None, model of this rank)."""
    rank, world = _world()
    lo, hi = shard_range(n_items, world, rank)
    comm_device = device if comm_device is None else comm_device
    x_local = scatter_utterances(x_all, n_items, item_shape, dtype, comm_device)
    model = model_factory()
    if hi > lo:
        if init_fn is not None:
            init_fn(model, lo, hi)
        y_local = model(x_local.to(device), iteration=iteration)
        if not isinstance(y_local, torch.Tensor):
            y_local = torch.from_numpy(y_local)
        y_local = y_local.to(comm_device)
    else:
        y_local = torch.empty((0,) + tuple(item_shape), dtype=dtype, device=comm_device)
    return (gather_utterances(y_local, n_items) if gather else y_local), model


# ---------------------------------------------------------------------------------------------------------------
# F-sharded single utterance (SURVEY.md 8e "within one utterance", 8 f2; ref src/bss/ilrma.py:421-428, 304-307)
# ---------------------------------------------------------------------------------------------------------------
def all_gather_ordered_sum(t):
    """Deterministic all-reduce(SUM): all-gather the per-rank partial, then every rank adds the `world` partials in
    RANK ORDER.  A ring/tree all-reduce may associate differently from run to run or rank to rank; this form gives
    every rank the same bits and the same bits as a single process that adds the same per-shard partials in shard
    order.  The payloads here are tiny (2.N.K.T reals and N scalars per iteration), so the extra bytes are free."""
    rank, world = _world()
    if world == 1:
        return t.clone()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    out = parts[0].clone()
    for p in parts[1:]:
        out += p
    return out
