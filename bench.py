#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: BASELINE.json config 2
(grid_pull + grid_push, 3-D 4x2x256^3 fp32, cubic, bound=dct2, random deformation).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one grid_pull followed by one grid_push over the whole batch (both
through the drop-in API -> C-ABI -> HIP kernels), inputs resident in HBM.  With
N > 1 ranks the batch axis is sharded: every rank owns its own 4x2x256^3 batch
(weak scaling, no data-path collective: SURVEY 8e); the timed region is bracketed
by a barrier + synchronize and the MAX over ranks is reported.

Prints ONE JSON line (rank 0): metric/value = aggregate Mvox/s, where a voxel is
one sample location (B * prod(spatial)) and each op of the step processes all of
them; `roofline` is for the slower of the two operators (median of the HIP-event times
inside the timed region), `roofline_per_op` for both; `cpu_baseline` times the CPU oracle (a port of the
reference algorithm, all host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "torch-interpol_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def make_inputs(B, C, n, sigma, device, seed):
    """SURVEY 8d generator: `torch.manual_seed(seed)` on the CPU, inp = randn, grid = identity + sigma * randn (voxels),
    then moved to the device -- the inputs the CPU baseline of the reference was timed on."""
    import interpol
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn([B, C, n, n, n], generator=g, dtype=torch.float32).to(device)
    grid = torch.randn([B, n, n, n, 3], generator=g, dtype=torch.float32).mul_(sigma).to(device)
    grid += interpol.identity_grid([n, n, n], dtype=torch.float32, device=device)
    return inp, grid


def smooth_grid(B, n, sigma, device, seed):
    """Smooth random deformation (examples/interpolate.ipynb recipe): 12^3 control
    points ~ N(0, sigma^2), cubic-upsampled to the full lattice."""
    import interpol
    g = torch.Generator(device=device).manual_seed(seed)
    ctrl = torch.randn([B, 3, 12, 12, 12], generator=g, device=device).mul_(sigma)
    disp = interpol.resize(ctrl, shape=[n, n, n], interpolation=3, prefilter=False)
    return disp.permute(0, 2, 3, 4, 1).contiguous() + interpol.identity_grid([n, n, n], device=device)


def cpu_baseline(n_sample, C, sigma, order, bound):
    """Oracle (port of the reference algorithm, oracle/interpol_oracle.c) on the host cores."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)
    inp = torch.randn([1, C, n_sample, n_sample, n_sample], generator=g)
    ident = torch.stack(torch.meshgrid(*[torch.arange(float(n_sample))] * 3, indexing="ij"), -1)
    grid = ident[None] + sigma * torch.randn([1, n_sample, n_sample, n_sample, 3], generator=g)
    inp, grid = inp.numpy(), grid.numpy()
    oracle.grid_pull(inp[:, :, :8, :8, :8].copy(), grid[:, :8, :8, :8].copy(), [bound], [order], 1, threads=cores)  # load lib
    t0 = time.perf_counter()
    oracle.grid_pull(inp, grid, [bound], [order], 1, threads=cores)
    t1 = time.perf_counter()
    oracle.grid_push(inp, grid, None, [bound], [order], 1, threads=cores)
    t2 = time.perf_counter()
    vox = n_sample ** 3
    return {
        "value": round(2 * vox / (t2 - t0) / 1e6, 3), "unit": "Mvox/s", "cores": cores, "kind": "port",
        "sample": "1x%dx%d^3 fp32 cubic/dct2 pull+push, oracle (C port of nd.py) with %d OpenMP threads; "
                  "pull %.2f s, push %.2f s" % (C, n_sample, cores, t1 - t0, t2 - t1),
        "pull_mvox_s": round(vox / (t1 - t0) / 1e6, 3), "push_mvox_s": round(vox / (t2 - t1) / 1e6, 3),
    }


def config4(args, world, rank, device, dist):
    """BASELINE configs[3]: `--sources` volumes 1x128^3 pushed + counted into ONE shared 512^3 target
    (order 3, replicate), the sources sharded over the ranks; every rank accumulates its shard into a
    local target (push and count stacked: one buffer), then ONE sum-reduce over RCCL.  Kernel, zero-fill
    and reduce are timed separately; value = source voxels of ALL ranks per second."""
    import interpol
    from interpol.distributed import push_count_shared, shard_range
    n, m = 128, 512
    lo, hi = shard_range(args.sources, rank, world)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    nsrc = hi - lo
    x = torch.randn([nsrc, 1, n, n, n], generator=g, device=device)
    grid = torch.randn([nsrc, n, n, n, 3], generator=g, device=device).mul_(args.sigma)
    grid += interpol.identity_grid([n] * 3, device=device) * ((m - 1) / (n - 1))
    kw = dict(interpolation=3, bound="replicate", extrapolate=True)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        push, count = push_count_shared(x, grid, [m] * 3, reduce="none", **kw)     # zero-fill + kernels
        if ev is not None:
            ev[1].record()
        if dist is not None:
            dist.all_reduce(push._base)          # push and count are views of ONE buffer: one 1.07 GB message
        if ev is not None:
            ev[2].record()
        return push, count

    for _ in range(max(1, args.warmup)):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    fill_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    scratch = torch.empty([1, 2, m, m, m], device=device)
    torch.cuda.synchronize()
    fill_ev[0].record(); scratch.zero_(); fill_ev[1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    local_ms = sorted(e[0].elapsed_time(e[1]) for e in events)
    red_ms = sorted(e[1].elapsed_time(e[2]) for e in events)
    med = lambda v: v[len(v) // 2]
    fill_ms = fill_ev[0].elapsed_time(fill_ev[1])
    if rank == 0:
        src_vox = args.sources * n ** 3
        bytes_local = nsrc * n ** 3 * 16 + 2 * m ** 3 * 4                    # sources (12 B grid + 4 B value) + two targets written
        kern_ms = med(local_ms) - fill_ms
        print(json.dumps({
            "metric": "M source vox/s, grid_push + grid_count of %d sources 1x128^3 into one shared 512^3 target" % args.sources,
            "value": round(src_vox / (elapsed / args.steps) / 1e6, 1), "unit": "Mvox/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %d sources 1x128^3 -> shared 512^3, cubic, replicate, grid = identity * 511/127 + N(0,%g^2); "
                                   "sources batch-sharded over %d rank(s), one all_reduce of the stacked push + count buffer (1.07 GB)"
                                   % (args.sources, args.sigma, world),
                       "sources_per_rank": nsrc, "parallelism": "batch-sharded x%d + RCCL all_reduce" % world},
            "local_ms_median": round(med(local_ms), 4), "zero_fill_ms": round(fill_ms, 4), "kernels_ms_median": round(kern_ms, 4),
            "reduce_ms_median": round(med(red_ms), 4) if dist is not None else None,
            "roofline": {"bound": "hbm", "kernel": "push_bricks pipeline (one rank's share)", "achieved": round(bytes_local / (kern_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_local / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "algorithmic_bytes_per_launch": bytes_local, "avg_launch_ms": round(kern_ms, 4)},
        }))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--bound", default="dct2")
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--grid", default="random", choices=["random", "smooth", "identity"],
                    help="deformation: i.i.d. N(0,sigma^2) noise (headline), smooth, or identity")
    ap.add_argument("--cpu-sample", type=int, default=256, help="edge of the CPU-baseline sample volume (0 = skip)")
    ap.add_argument("--no-fastpath", action="store_true", help="force the generic kernels")
    ap.add_argument("--no-extras", action="store_true", help="skip the smooth / identity deformation timings")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="BASELINE.json configuration: 2 = pull + push of a batch (the headline), "
                         "4 = 64 sources 1x128^3 pushed + counted into ONE shared 512^3 target, batch-sharded, one RCCL reduce")
    ap.add_argument("--sources", type=int, default=64, help="config 4: number of source volumes (all ranks together)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, RCCL over xGMI)
        import socket
        import subprocess
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d: fewer ranks than asked for" % (args.gpus, world))
    # INTERPOL_BENCH_BACKEND=gloo: a dry run of the N > 1 control flow on a box with fewer GPUs (ranks share devices, the
    # collectives go through the host); never a measurement
    backend = os.environ.get("INTERPOL_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import interpol
    from interpol import _hip
    from interpol.codes import bound_to_code
    _hip.lib()                                   # the HIP extension must be there: no fallback

    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit("RCCL sees %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    if args.config == 4:
        return config4(args, world, rank, device, dist)

    B, C, n = args.batch, args.channels, args.size
    inp, grid = make_inputs(B, C, n, args.sigma, device, 1234 + rank)
    if args.grid == "smooth":
        grid = smooth_grid(B, n, args.sigma, device, 1234 + rank)
    elif args.grid == "identity":
        grid = interpol.identity_grid([n, n, n], device=device)[None].expand(B, n, n, n, 3).contiguous()
    kw = dict(interpolation=args.order, bound=args.bound, extrapolate=True)

    if args.no_fastpath:
        orig = _hip.make_problem

        def patched(*a, **k):
            p = orig(*a, **k)
            p.flags |= _hip.FLAG_NO_FASTPATH
            return p
        _hip.make_problem = patched

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        out = interpol.grid_pull(inp, grid, **kw)
        if ev is not None:
            ev[1].record()
        psh = interpol.grid_push(inp, grid, **kw)
        if ev is not None:
            ev[2].record()
        return out, psh

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # per-rank step times (a SCALE run describes itself: every rank's clock, and the number of ranks RCCL really spans)
    rank_ms = [1e3 * elapsed / args.steps]
    rccl_world = 1
    if dist is not None:
        rccl_world = dist.get_world_size()
        tl = torch.tensor([local_elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(tl) for _ in range(rccl_world)]
        dist.all_gather(allt, tl)
        rank_ms = [1e3 * float(x) / args.steps for x in allt]

    pull_ms = sorted(e[0].elapsed_time(e[1]) for e in events)
    push_ms = sorted(e[1].elapsed_time(e[2]) for e in events)
    pull_avg = sum(pull_ms) / len(pull_ms)
    push_avg = sum(push_ms) / len(push_ms)
    pull_med, push_med = pull_ms[len(pull_ms) // 2], push_ms[len(push_ms) // 2]

    vox_rank = B * n ** 3
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * 2 * vox_rank / (elapsed / args.steps) / 1e6

    # algorithmic bytes per launch: every input read once, every output written once
    bytes_pull = vox_rank * (3 * 4 + C * 4) + B * C * n ** 3 * 4
    bytes_push = vox_rank * (3 * 4 + C * 4) + B * C * n ** 3 * 4
    # SURVEY 8d protocol next to the API-level numbers: outputs allocated BEFORE the timed region, medians.  (pull writes
    # into `out=`; push accumulates into a target that was zeroed outside the events -- its zero-fill, 0.06 ms, is then
    # not in the number, which the API-level push_ms includes.)
    pre = {}
    if world == 1:
        bc, oc = [bound_to_code(args.bound)] * 3, [args.order] * 3
        pflags = _hip.FLAG_NO_FASTPATH if args.no_fastpath else 0
        o_pull = torch.empty_like(inp)
        o_push = torch.zeros_like(inp)
        ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        for k in range(-2, args.steps):
            e = ev2[max(k, 0)]
            e[0].record(); _hip.gather("pull", inp, grid, bc, oc, 1, flags=pflags, out=o_pull)
            e[1].record(); _hip.scatter("push", inp, grid, [n] * 3, bc, oc, 1, flags=pflags | _hip.FLAG_ACCUMULATE, out=o_push)
            e[2].record()
        torch.cuda.synchronize()
        a = sorted(e[0].elapsed_time(e[1]) for e in ev2); b_ = sorted(e[1].elapsed_time(e[2]) for e in ev2)
        pre = {"pull_ms_median": round(a[len(a) // 2], 4), "push_ms_median": round(b_[len(b_) // 2], 4),
               "note": "outputs allocated before the timed region (C-ABI level: interpol_pull into `out`, interpol_push with "
                       "INTERPOL_FLAG_ACCUMULATE into a target zeroed outside the events), medians of %d" % args.steps}
        del o_pull, o_push
    ops_roof = {}
    for name, nbytes, med_ms, avg_ms in (("grid_pull", bytes_pull, pull_med, pull_avg), ("grid_push", bytes_push, push_med, push_avg)):
        ops_roof[name] = {"bound": "hbm", "achieved": round(nbytes / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": nbytes,
                          "median_launch_ms": round(med_ms, 4), "avg_launch_ms": round(avg_ms, 4)}
    # the dominant operator's roofline from its AVERAGE launch duration over the timed steps (the median rides along)
    dom = "grid_push" if push_avg >= pull_avg else "grid_pull"
    dom_ms = max(push_avg, pull_avg)
    dom_med = push_med if dom == "grid_push" else pull_med
    dom_bytes = bytes_push if dom == "grid_push" else bytes_pull
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # HBM-side bytes of the same launch: NOT measured in this run -- read from the committed PMC summary
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_workload.py, see profiles/)
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get(dom)
            for name in ops_roof:
                ops_roof[name]["traffic"] = pj.get(name)
            traffic_source = "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, not this run)"
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "Mvox/s grid_pull & grid_push, 256^3 fp32 cubic/dct2",
            "value": round(value, 1), "unit": "Mvox/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: grid_pull + grid_push, %dx%dx%d^3 fp32 per GPU, order %d, "
                                   "bound %s, extrapolate=True, grid = identity + N(0,%g^2) (%s)"
                                   % (B, C, n, args.order, args.bound, args.sigma, args.grid),
                       "batch_per_gpu": B, "channels": C, "shape": [n, n, n], "parallelism": "batch-sharded x%d" % world},
            "rccl_world_size": rccl_world, "rank_ms_per_step": [round(x, 4) for x in rank_ms],
            "pull_ms": round(pull_avg, 4), "push_ms": round(push_avg, 4),
            "pull_ms_median": round(pull_med, 4), "push_ms_median": round(push_med, 4),
            "timing_note": "HIP events on the launch stream around each op inside the timed region; the op allocates its "
                           "output (caching allocator) and grid_push includes the zero-fill of its target",
            "pull_mvox_s": round(vox_rank / pull_avg / 1e3, 1), "push_mvox_s": round(vox_rank / push_avg / 1e3, 1),
            "pull_frac_hbm": round(bytes_pull / (pull_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "push_frac_hbm": round(bytes_push / (push_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 4), "median_launch_ms": round(dom_med, 4),
                         "timing": "average of %d launches (HIP events on the launch stream)" % args.steps,
                         "launch": "one interpol_push call = the kernels its probe routes to + zero-fill; on this workload's i.i.d. field the probe "
                                   "picks own_bin + 9 own_accumulate of csrc/push_owner.hip (a smooth field would take push_tiled: the "
                                   "other_deformations rows); per-kernel times: profiles/*_kernel_stats.txt"},
        }
        line["roofline_per_op"] = ops_roof
        if pre:
            line["preallocated_outputs"] = pre
        if world == 1 and args.grid == "random" and not args.no_extras:
            # same workload under the other deformation models of SURVEY 8d (not the headline)
            extras = {}
            for name in ("smooth", "identity"):
                g2 = smooth_grid(B, n, args.sigma, device, 1234) if name == "smooth" else \
                    interpol.identity_grid([n, n, n], device=device)[None].expand(B, n, n, n, 3).contiguous()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                tp, ts = [], []
                for it in range(6):
                    ev[0].record(); interpol.grid_pull(inp, g2, **kw)
                    ev[1].record(); interpol.grid_push(inp, g2, **kw)
                    ev[2].record(); torch.cuda.synchronize()
                    if it:
                        tp.append(ev[0].elapsed_time(ev[1])); ts.append(ev[1].elapsed_time(ev[2]))
                tp, ts = sum(tp) / len(tp), sum(ts) / len(ts)
                extras[name] = {"pull_ms": round(tp, 4), "push_ms": round(ts, 4),
                                "pull_frac_hbm": round(bytes_pull / (tp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "push_frac_hbm": round(bytes_push / (ts * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del g2
            line["other_deformations"] = extras
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, C, args.sigma, args.order, bound_to_code(args.bound))
            line["cpu_baseline"]["reference_measured_in_build_container"] = \
                "reference TorchScript CPU path, 8 cores: pull 0.823 / push 0.952 Mvox/s at 4x2x256^3 (BASELINE.md sec 2)"
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
