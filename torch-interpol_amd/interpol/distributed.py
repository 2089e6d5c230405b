"""Multi-GPU use of the hot path on one MI355X node: one process per GPU,
`torch.distributed` with the "nccl" backend (= RCCL over xGMI on ROCm).

Every operator is independent per batch item (reference interpol/nd.py:95-106),
so pull / grad / count / push with per-item outputs shard over the batch axis
with NO communication: each rank simply calls the API on its shard
(`shard_range`).  The one exchange step of the path is splatting MANY sources
into ONE shared target volume (BASELINE config 4): in reference terms

    push  = grid_push(inp, grid, shape).sum(0)      count = grid_count(grid, shape).sum(0)

Here every rank accumulates its local sources straight into a single local
target (the kernels take a batch-stride-0 target, no per-item volumes), push
and count are stacked into one buffer, and ONE sum-reduce of that buffer runs
over RCCL.
"""
import torch

from . import ops
from .codes import bound_to_code, order_to_code, pad_codes


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced [start, stop) of the batch items owned by `rank`."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def slab_range(grid, rank, world_size):
    """[start, stop) of the slab of the OUTPUT-GRID axis 0 (`grid.shape[1]`) owned by `rank`: the
    split for a single huge volume (B = 1), where the batch axis has nothing to shard.  Every
    sample point is independent (reference interpol/nd.py:95-106), so a rank that pulls
    `grid[:, start:stop]` from the (replicated) image gets exactly the rows `start:stop` of the
    full result; for push / count a rank splats its slab of sources into a private full-size target
    (`push_count_shared` with the slabs as the local shard) and the targets are summed."""
    return shard_range(grid.shape[1], rank, world_size)


def grid_pull_slabs(input, grid, rank=None, world_size=None, group=None, gather=True, **kw):
    """grid_pull of ONE volume sharded over the output-grid axis 0: each rank samples its slab of
    `grid`; `gather=True` all-gathers the slabs (every rank returns the full result, bit-identical
    to the unsharded call), else the local slab is returned.  `kw` as for `interpol.grid_pull`."""
    from . import api
    dist = torch.distributed
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    sd = grid.dim() - 1 - grid.shape[-1]                 # axis of the first spatial dim of the grid
    lo, hi = shard_range(grid.shape[sd], rank, world_size)
    local = api.grid_pull(input, grid.narrow(sd, lo, hi - lo), **kw)
    if not gather or world_size == 1:
        return local
    od = local.dim() - grid.shape[-1]                    # the same axis in the output
    return _gather_slabs(local, od, grid.shape[sd], world_size, group)


def _gather_slabs(local, axis, n, world_size, group):
    """all_gather of slabs that may differ in length by one row (n not divisible by the world size): gloo, and older
    RCCL builds, need equal sizes, so every slab travels padded to ceil(n / world) rows and is trimmed on arrival."""
    dist = torch.distributed
    rows = -(-n // world_size)
    shape = list(local.shape)
    mine = shape[axis]
    if mine < rows:
        shape[axis] = rows - mine
        local = torch.cat([local, local.new_zeros(shape)], axis)
    local = local.contiguous()
    parts = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(parts, local, group=group)
    sizes = [shard_range(n, r, world_size) for r in range(world_size)]
    return torch.cat([p.narrow(axis, 0, b - a) for p, (a, b) in zip(parts, sizes)], axis)


def grid_push_slabs(input, grid, shape=None, rank=None, world_size=None, group=None, reduce='all', dst=0,
                    with_count=False, **kw):
    """grid_push (and, `with_count=True`, grid_count) of ONE volume (B = 1) sharded over axis 0 of the SOURCE lattice:
    each rank splats its slab of `input` / `grid` into a private full-size target and the targets are summed over the
    ranks (`reduce` as in `push_count_shared`).  Every source point is independent (reference interpol/nd.py:146-213),
    so the sum equals the unsharded `grid_push(input, grid, shape)` up to the order of the float additions.
    input : (1, C, *inshape) or (C, *inshape)     grid : (1, *inshape, D) or (*inshape, D)
    Returns push (C, *shape), or (push, count) with `with_count=True`."""
    dist = torch.distributed
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    dim = grid.shape[-1]
    if grid.dim() == dim + 1:
        grid = grid[None]
    if input.dim() == dim + 1:
        input = input[None]
    if grid.shape[0] != 1 or input.shape[0] != 1:
        raise ValueError('grid_push_slabs shards ONE volume; batches shard over the batch axis (shard_range)')
    if shape is None:
        shape = list(grid.shape[1:-1])
    lo, hi = shard_range(grid.shape[1], rank, world_size)
    push, count = push_count_shared(input[:, :, lo:hi], grid[:, lo:hi], shape, group=group, reduce=reduce, dst=dst,
                                    with_count=with_count, **kw)
    return (push, count) if with_count else push


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def _reduce_scatter_wanted():
    """INTERPOL_REDUCE=reduce_scatter: the shared-target sum as ONE reduce-scatter + ONE all-gather instead of the
    library's all_reduce (on the xGMI full mesh of an MI355X node every rank then exchanges 1/world of the
    1 GB buffer with every peer directly -- 7 links busy -- where a ring is bound by one link).  A switch, so that both
    can be timed on an 8-GPU node (`bench.py --config 4`)."""
    import os
    return os.environ.get('INTERPOL_REDUCE', 'all_reduce').lower() in ('reduce_scatter', 'rs')


def _all_reduce_by_reduce_scatter(buf, group):
    dist = torch.distributed
    if dist.get_backend(group) == 'gloo':                   # (gloo has no reduce_scatter: the CPU dry runs keep the all_reduce)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return
    world = dist.get_world_size(group)
    flat = buf.view(-1)
    n = flat.numel()
    per = -(-n // world)
    if per * world != n:                                    # equal chunks: pad the tail
        work = flat.new_zeros(per * world)
        work[:n] = flat
    else:
        work = flat
    mine = torch.empty(per, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(mine, work, op=dist.ReduceOp.SUM, group=group)
    dist.all_gather_into_tensor(work, mine, group=group)
    if work is not flat:
        flat.copy_(work[:n])


def push_count_shared(input, grid, shape, interpolation='linear', bound='zero', extrapolate=False,
                      group=None, reduce='all', dst=0, with_count=True):
    """Splat the LOCAL shard of sources into a shared target and sum over ranks.

    input : (B_local, C, *inshape) or None (count only)     grid : (B_local, *inshape, D)
    shape : target spatial shape
    reduce: 'all' (all_reduce: every rank gets the result), 'dst' (reduce to rank `dst`),
            'none' (local partial sums only)
    Returns (push (C, *shape) | None, count (1, *shape) | None).  Equivalent to
    `grid_push(all_inputs, all_grids, shape).sum(0)` / `grid_count(all_grids, shape).sum(0)`
    of the reference evaluated over the batches of ALL ranks.
    """
    dim = grid.shape[-1]
    shape = [int(s) for s in shape]
    b = pad_codes([bound_to_code(x) for x in _as_list(bound)], dim)
    o = pad_codes([order_to_code(x) for x in _as_list(interpolation)], dim)
    ex = int(extrapolate)
    C = 0 if input is None else input.shape[1]
    nch = C + (1 if with_count else 0)
    if nch == 0:
        raise ValueError('nothing to do: no input and with_count=False')
    # the accumulation dtype is the one the kernels promote to (an fp32 image with an fp64 grid
    # accumulates in fp64, as the per-item API does)
    if input is None:
        dtype = grid.dtype
    else:
        dtype = torch.promote_types(input.dtype, grid.dtype) if grid.dtype == torch.float64 else input.dtype
        input = input.to(dtype)
    if dtype in (torch.bfloat16, torch.float16):
        raise ValueError('push_count_shared accumulates in the target: use a float32 / float64 image (low-precision targets cannot accumulate)')
    # push and count share one buffer -> one collective message
    buf = torch.zeros([1, nch] + shape, dtype=dtype, device=grid.device)
    k = ops.kernels(input, grid, dim=dim)
    if grid.shape[0] > 0:
        if input is not None and with_count:
            k.push_shared_(buf, input, grid, b, o, ex, with_count=True)    # values and count in one pass over the grid
        elif input is not None:
            k.push_shared_(buf[:, :C], input, grid, b, o, ex)
        else:
            k.push_shared_(buf[:, C:], None, grid, b, o, ex)
    if reduce != 'none' and torch.distributed.is_available() and torch.distributed.is_initialized():
        if reduce == 'all' and _reduce_scatter_wanted():
            _all_reduce_by_reduce_scatter(buf, group)
        elif reduce == 'all':
            torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM, group=group)
        elif reduce == 'dst':
            torch.distributed.reduce(buf, dst=dst, op=torch.distributed.ReduceOp.SUM, group=group)
        else:
            raise ValueError("reduce must be 'all', 'dst' or 'none'")
    push = buf[0, :C] if input is not None else None
    count = buf[0, C:] if with_count else None
    return push, count
