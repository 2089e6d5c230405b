"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the batch sharding and
the one collective of the path (sum-reduce of the shared push/count target).
The ranks compute with the product's device-generic PyTorch kernel table (interpol/torch_kernels.py: what CPU tensors
get from the public API); the parent checks the results against the oracle.  On the GPU box the same code runs over
RCCL with the HIP kernels (test_hip_parity.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (import paths)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import interpol
    from interpol import ops
    from interpol.distributed import push_count_shared, shard_range

    g = torch.Generator().manual_seed(4321)          # same data on every rank, each takes its shard
    B, C, n, m = 5, 2, 6, 12
    inp = torch.randn([B, C, n, n, n], generator=g, dtype=torch.float64)
    ident = torch.stack(torch.meshgrid(*[torch.arange(float(n))] * 3, indexing="ij"), -1)
    grid = ident[None] * ((m - 1) / (n - 1)) + torch.randn([B, n, n, n, 3], generator=g, dtype=torch.float64)
    lo, hi = shard_range(B, rank, world)
    # (CPU tensors: the product's own device-generic kernel table, interpol/torch_kernels.py -- the oracle is the CHECKER, in
    #  the parent process, not the thing that computes here)
    assert ops.kernels(inp, grid, dim=3).__name__ == "TorchKernels"
    push, count = push_count_shared(inp[lo:hi], grid[lo:hi], [m, m, m], interpolation=3, bound="replicate",
                                    extrapolate=True, reduce="all")
    # sharded pull needs no communication: every rank computes its items
    pulled = interpol.grid_pull(inp[lo:hi], grid[lo:hi], interpolation=3, bound="dct2", extrapolate=True)
    p2, c2 = push_count_shared(inp[lo:hi], grid[lo:hi], [m, m, m], interpolation=3, bound="replicate",
                               extrapolate=True, reduce="dst", dst=1)
    # ONE volume (B = 1): the output-grid axis 0 is split into slabs instead
    from interpol.distributed import grid_pull_slabs, slab_range
    one_inp, one_grid = inp[:1], grid[:1, :, :, :5] * 0.4
    slab = grid_pull_slabs(one_inp, one_grid, interpolation=3, bound="dct2", extrapolate=True)
    a, b_ = slab_range(one_grid, rank, world)
    local = grid_pull_slabs(one_inp, one_grid, gather=False, interpolation=3, bound="dct2", extrapolate=True)
    assert local.shape[2] == b_ - a
    # slabs of unequal length (5 rows over 2 ranks): padded all_gather
    odd_grid = grid[:1, :5] * 0.4
    odd = grid_pull_slabs(one_inp, odd_grid, interpolation=3, bound="dct2", extrapolate=True)
    # grid_push of ONE volume over slabs of the source lattice (+ the count image), summed over the ranks;
    # the same with the reduce-scatter switch set (gloo keeps the all_reduce: control flow only)
    from interpol.distributed import grid_push_slabs
    pslab, cslab = grid_push_slabs(inp[:1, :, :5], grid[:1, :5], [m, m, m], interpolation=3, bound="replicate",
                                   extrapolate=True, with_count=True)
    os.environ["INTERPOL_REDUCE"] = "reduce_scatter"
    pslab2 = grid_push_slabs(inp[0, :, :5], grid[0, :5], [m, m, m], interpolation=3, bound="replicate", extrapolate=True)
    del os.environ["INTERPOL_REDUCE"]
    gathered = [None] * world
    dist.all_gather_object(gathered, pulled.numpy())
    if rank == 0:
        np.savez(tmp, push=push.numpy(), count=count.numpy(), pulled=np.concatenate(gathered, 0),
                 inp=inp.numpy(), grid=grid.numpy(), slab=slab.numpy(), one_grid=one_grid.numpy(),
                 odd=odd.numpy(), odd_grid=odd_grid.numpy(), pslab=pslab.numpy(), cslab=cslab.numpy(), pslab2=pslab2.numpy())
    if rank == 1:
        np.savez(tmp + ".dst", push=p2.numpy(), count=c2.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shared_target_dtype_rules():
    """mixed image / grid dtypes promote like the per-item API; low-precision targets are refused loudly."""
    sys.path.insert(0, HERE)
    from interpol import ops
    from interpol.distributed import push_count_shared
    from oracle_kernels import OracleKernels
    g = torch.Generator().manual_seed(1)
    inp = torch.randn([2, 1, 5, 5, 5], generator=g)
    grid = (torch.rand([2, 5, 5, 5, 3], generator=g, dtype=torch.float64) * 4)
    with ops.use_kernels(OracleKernels):
        push, count = push_count_shared(inp, grid, [6, 6, 6], interpolation=1, bound="zero", extrapolate=True, reduce="none")
        assert push.dtype == torch.float64 and count.dtype == torch.float64
        with pytest.raises(ValueError):
            push_count_shared(inp.bfloat16(), grid.float(), [6, 6, 6], reduce="none")


def test_shard_range():
    from interpol.distributed import shard_range
    for n in (0, 1, 5, 8, 64):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                a, b = shard_range(n, r, w)
                cover += list(range(a, b))
                assert 0 <= b - a <= n // w + 1
            assert cover == list(range(n))


@pytest.mark.timeout(300)
def test_push_count_shared_gloo_world2(tmp_path):
    from oracle import oracle
    port = 29500 + os.getpid() % 2000
    tmp = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    r = np.load(tmp)
    m = 12
    want_push = np.asarray(oracle.grid_push(r["inp"], r["grid"], [m, m, m], [1], [3], 1)).sum(0)
    want_count = np.asarray(oracle.grid_count(r["grid"], [m, m, m], [1], [3], 1)).sum(0)
    assert np.abs(r["push"] - want_push).max() < 1e-12 * np.abs(want_push).max()
    assert np.abs(r["count"] - want_count).max() < 1e-12 * np.abs(want_count).max()
    import interpol

    def unsharded(inp_, grid_):          # the same kernel table on the whole problem, in this process
        return interpol.grid_pull(torch.from_numpy(inp_), torch.from_numpy(grid_), interpolation=3, bound="dct2", extrapolate=True).numpy()

    def close(a, b):
        return np.abs(a - b).max() < 1e-12 * np.abs(b).max()
    want_pull = np.asarray(oracle.grid_pull(r["inp"], r["grid"], [3], [3], 1))
    assert close(r["pulled"], want_pull)                                   # right (the oracle is the checker) ...
    assert np.array_equal(r["pulled"], unsharded(r["inp"], r["grid"]))     # ... and batch sharding == full-batch result, bit for bit
    want_slab = np.asarray(oracle.grid_pull(r["inp"][:1], r["one_grid"], [3], [3], 1))
    assert close(r["slab"], want_slab)
    assert np.array_equal(r["slab"], unsharded(r["inp"][:1], r["one_grid"]))   # output-grid slabs == unsharded result, bit for bit
    want_odd = np.asarray(oracle.grid_pull(r["inp"][:1], r["odd_grid"], [3], [3], 1))
    assert close(r["odd"], want_odd)
    assert np.array_equal(r["odd"], unsharded(r["inp"][:1], r["odd_grid"]))    # unequal slabs (3 + 2 rows)
    want_ps = np.asarray(oracle.grid_push(r["inp"][:1, :, :5], r["grid"][:1, :5], [m, m, m], [1], [3], 1))[0]
    want_cs = np.asarray(oracle.grid_count(r["grid"][:1, :5], [m, m, m], [1], [3], 1))[0]
    assert np.abs(r["pslab"] - want_ps).max() < 1e-12 * np.abs(want_ps).max()      # source-lattice slabs of one volume
    assert np.abs(r["cslab"] - want_cs).max() < 1e-12 * np.abs(want_cs).max()
    assert np.abs(r["pslab2"] - want_ps).max() < 1e-12 * np.abs(want_ps).max()
    d = np.load(tmp + ".dst.npz")
    assert np.abs(d["push"] - want_push).max() < 1e-12 * np.abs(want_push).max()
    assert np.abs(d["count"] - want_count).max() < 1e-12 * np.abs(want_count).max()
