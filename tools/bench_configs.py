#!/usr/bin/env python
"""Timings of the other BASELINE.json configs (1, 3, 4, 5) on one MI355X: per-op ms, Mvox/s and
fraction of the HBM roofline (algorithmic bytes of BASELINE.md sec. 3 / 8 TB/s).  Config 2 is bench.py."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-interpol_amd")); sys.path.insert(0, ROOT)
import torch, interpol
from interpol.distributed import push_count_shared

dev = torch.device("cuda", 0)
PEAK = 8000e9


def timeit(fn, reps=5, inner=4):
    """median over `reps` of the time per call of `inner` back-to-back calls (a single call after a synchronisation runs
    at idle clocks: +15-20 % on this chip)"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def rec(out, name, ms, vox, nbytes):
    out[name] = {"ms": round(ms, 4), "Mvox_s": round(vox / ms / 1e3, 1), "frac_hbm": round(nbytes / (ms * 1e-3) / PEAK, 4)}


def ident(sp, B, sigma, g, dtype=torch.float32):
    grid = torch.randn([B, *sp, len(sp)], generator=g, device=dev, dtype=torch.float32).mul_(sigma)
    grid += interpol.identity_grid(sp, dtype=torch.float32, device=dev)
    return grid.to(dtype)


res = {}
g = torch.Generator(device=dev).manual_seed(1234)
which = sys.argv[1:] or ["1", "3", "4", "5", "f", "b"]

if "1" in which:    # cfg1: 1x1x128x128 linear zero identity
    x = torch.randn(1, 1, 128, 128, device=dev)
    gr = interpol.identity_grid([128, 128], device=dev)[None]
    ms = timeit(lambda: interpol.grid_pull(x, gr, interpolation=1, bound="zero", extrapolate=False), 20)
    rec(res, "cfg1_pull_linear_zero_128x128", ms, 128 * 128, 128 * 128 * (8 + 4 + 4))
    assert torch.equal(interpol.grid_pull(x, gr, interpolation=1, bound="zero", extrapolate=False), x)

if "3" in which:    # cfg3: 8x1x192^3 order 5 dft: grid_grad + backward of grid_pull
    B, C, n = 8, 1, 192
    x = torch.randn(B, C, n, n, n, generator=g, device=dev)
    gr = ident([n, n, n], B, 2.0, g)
    kw = dict(interpolation=5, bound="dft", extrapolate=True)
    vox = B * n ** 3
    rec(res, "cfg3_grid_grad_o5_dft", timeit(lambda: interpol.grid_grad(x, gr, **kw), 3), vox, vox * (12 + C * 12) + B * C * n ** 3 * 4)
    rec(res, "cfg3_grid_pull_o5_dft", timeit(lambda: interpol.grid_pull(x, gr, **kw), 3), vox, vox * (12 + C * 4) + B * C * n ** 3 * 4)
    xr, grr = x.clone().requires_grad_(True), gr.clone().requires_grad_(True)
    y = interpol.grid_pull(xr, grr, **kw)
    gy = torch.randn_like(y)

    def bwd():
        xr.grad = None; grr.grad = None
        y.backward(gy, retain_graph=True)
    rec(res, "cfg3_pull_backward_fused_o5_dft", timeit(bwd, 3), vox, vox * (24 + C * 4) + 2 * B * C * n ** 3 * 4)
    del xr, grr, y, gy

if "4" in which:    # cfg4: 64 sources 1x128^3 -> shared 512^3, order 3 replicate (one GPU's share at 8 GPUs = 8 sources; here all 64)
    for nsrc in (8, 64):
        n, m = 128, 512
        x = torch.randn(nsrc, 1, n, n, n, generator=g, device=dev)
        gr = torch.randn([nsrc, n, n, n, 3], generator=g, device=dev).mul_(2.0)
        gr += interpol.identity_grid([n, n, n], device=dev) * ((m - 1) / (n - 1))
        vox = nsrc * n ** 3
        ms = timeit(lambda: push_count_shared(x, gr, [m, m, m], interpolation=3, bound="replicate", extrapolate=True, reduce="none"), 3)
        rec(res, "cfg4_push+count_shared_%dsrc_o3_replicate" % nsrc, ms, vox, vox * 16 + vox * 12 + 2 * m ** 3 * 4)
        del x, gr

if "5" in which:    # cfg5: 2-D, orders [2,3,5]->[2,3], bounds [dct1,dst2,zero]->[dct1,dst2], bf16 storage, fp32 grid; 32 of 256 images (1/8: one GPU's share)
    B, C, n = 32, 3, 1024
    x = torch.randn(B, C, n, n, generator=g, device=dev).to(torch.bfloat16)
    gr = ident([n, n], B, 2.0, g)
    kw = dict(interpolation=[2, 3, 5], bound=["dct1", "dst2", "zero"], extrapolate=True)
    vox = B * n * n
    rec(res, "cfg5_pull_bf16_o23", timeit(lambda: interpol.grid_pull(x, gr, **kw), 3), vox, vox * (8 + C * 2) + B * C * n * n * 2)
    rec(res, "cfg5_push_bf16_o23", timeit(lambda: interpol.grid_push(x, gr, **kw), 3), vox, vox * (8 + C * 2) + B * C * n * n * 2)
    xf = x.float()
    rec(res, "cfg5_pull_f32_o23", timeit(lambda: interpol.grid_pull(xf, gr, **kw), 3), vox, vox * (8 + C * 4) + B * C * n * n * 4)
    rec(res, "cfg5_prefilter_bf16_o23_dct1dct2", timeit(lambda: interpol.spline_coeff_nd(x, [2, 3], ["dct1", "dct2"], 2), 3), vox, 2 * 2 * B * C * n * n * 2)
    from interpol import _hip
    b2, o2 = [2, 5], [2, 3]
    nb = vox * (16 + 2 * C * 2)
    rec(res, "cfg5_pull_backward_grid_only_bf16_o23", timeit(lambda: _hip.pull_backward(x, x, gr, b2, o2, 1, False, True), 3), vox, nb)
    rec(res, "cfg5_pull_backward_grid_only_bf16_o23_generic", timeit(lambda: _hip.pull_backward(x, x, gr, b2, o2, 1, False, True, flags=_hip.FLAG_NO_FASTPATH), 3), vox, nb)
    rec(res, "cfg5_pull_backward_both_bf16_o23", timeit(lambda: _hip.pull_backward(x, x, gr, b2, o2, 1, True, True), 3), vox, nb + vox * C * 4)
    rec(res, "cfg5_push_backward_both_bf16_o23", timeit(lambda: _hip.push_backward(x, x, gr, b2, o2, 1, True, True), 3), vox, nb + vox * C * 2)
    rec(res, "cfg5_prefilter_f32_o23_dct1dct2", timeit(lambda: interpol.spline_coeff_nd(xf, [2, 3], ["dct1", "dct2"], 2), 3), vox, 2 * 2 * B * C * n * n * 4)
    # round 5: the same calls on rougher fields -- the routed default (probe2d: lean tiles or the bricks of csrc/scatter2d.hip)
    for sg in (4.0, 8.0, 16.0):
        del gr
        gr = ident([n, n], B, sg, g)
        rec(res, "cfg5_pull_bf16_o23_sigma%d" % sg, timeit(lambda: interpol.grid_pull(x, gr, **kw), 3), vox, vox * (8 + C * 2) + B * C * n * n * 2)
        rec(res, "cfg5_push_bf16_o23_sigma%d" % sg, timeit(lambda: interpol.grid_push(x, gr, **kw), 3), vox, vox * (8 + C * 2) + B * C * n * n * 2)
        rec(res, "cfg5_pull_backward_both_bf16_o23_sigma%d" % sg, timeit(lambda: _hip.pull_backward(x, x, gr, b2, o2, 1, True, True), 3), vox, nb + vox * C * 4)
        rec(res, "cfg5_push_backward_both_bf16_o23_sigma%d" % sg, timeit(lambda: _hip.push_backward(x, x, gr, b2, o2, 1, True, True), 3), vox, nb + vox * C * 2)

if "f" in which:    # row f2: resize / restrict on a separable lattice vs the same call with a dense grid tensor
    B, C, n = 4, 2, 128
    x = torch.randn(B, C, n, n, n, generator=g, device=dev)
    kw = dict(factor=[2, 2, 2], anchor='e', interpolation=3, bound='dct2', prefilter=False)
    m = 2 * n
    vox = B * m ** 3
    nbytes = B * C * (n ** 3 + m ** 3) * 4
    rec(res, "f2_resize_2x_128to256_cubic_separable", timeit(lambda: interpol.resize(x, **kw), 3), vox, nbytes)
    lin = torch.arange(0., m, device=dev) * 0.5 + 0.5 * (0.5 - 1)

    def dense_path():
        grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing='ij'), -1)
        return interpol.grid_pull(x, grid, interpolation=3, bound='dct2', extrapolate=True)
    rec(res, "f2_resize_2x_128to256_cubic_dense_grid_incl_meshgrid", timeit(dense_path, 3), vox, nbytes)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing='ij'), -1)
    rec(res, "f2_resize_2x_128to256_cubic_dense_grid_kernel_only", timeit(lambda: interpol.grid_pull(x, grid, interpolation=3, bound='dct2', extrapolate=True), 3), vox, nbytes)
    del grid
    y = torch.randn(B, C, m, m, m, generator=g, device=dev)
    rec(res, "f2_restrict_2x_256to128_linear_separable", timeit(lambda: interpol.restrict(y, factor=[2, 2, 2], anchor='e', interpolation=1, bound='dct2'), 3), vox, nbytes)

if "f" in which:    # row f4: label map (50 labels), trilinear, 1x1x192^3, one-pass arg-max vs the reference's per-label loop
    from interpol import _hip
    n = 192
    lab = torch.randint(0, 50, [1, 1, n, n, n], generator=g, device=dev)
    gr = ident([n, n, n], 1, 2.0, g)
    vox = n ** 3
    rec(res, "f4_labels_50_linear_192_one_pass", timeit(lambda: interpol.grid_pull(lab, gr, interpolation=1, bound="dct2", extrapolate=True), 3), vox, vox * (12 + 8 + 8))

    def loop():
        out = torch.zeros_like(lab); pmax = torch.zeros(lab.shape, device=dev)
        for l in lab.unique():
            soft = _hip.gather("pull", (lab == l).float(), gr, [3] * 3, [1] * 3, 1)
            out[soft > pmax] = l
            pmax = torch.max(pmax, soft)
        return out
    rec(res, "f4_labels_50_linear_192_per_label_loop", timeit(loop, 3), vox, vox * (12 + 8 + 8))

    rec(res, "f4_labels_50_cubic_192_one_pass_64_taps", timeit(lambda: interpol.grid_pull(lab, gr, interpolation=3, bound="dct2", extrapolate=True), 3), vox, vox * (12 + 8 + 8))
    # the same with a piecewise-constant label map (8^3 blocks: what segmentations look like -- one or two labels under most stencils;
    # the i.i.d. map above is the adversarial case: ~36 distinct labels under every stencil)
    coarse = torch.randint(0, 50, [1, 1, n // 8, n // 8, n // 8], generator=g, device=dev)
    blocky = coarse.repeat_interleave(8, 2).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
    rec(res, "f4_labels_50_blocks_cubic_192_one_pass_64_taps", timeit(lambda: interpol.grid_pull(blocky, gr, interpolation=3, bound="dct2", extrapolate=True), 3), vox, vox * (12 + 8 + 8))
    rec(res, "f4_labels_50_blocks_linear_192_one_pass", timeit(lambda: interpol.grid_pull(blocky, gr, interpolation=1, bound="dct2", extrapolate=True), 3), vox, vox * (12 + 8 + 8))

if "f" in which:    # row f3: affine lattice evaluated in the kernel vs a dense affine grid tensor (4x2x256^3 cubic)
    B, C, n = 4, 2, 256
    x = torch.randn(B, C, n, n, n, generator=g, device=dev)
    mat = torch.tensor([[0.98, 0.05, -0.03, 1.5], [-0.04, 1.01, 0.02, -2.0], [0.03, -0.02, 0.99, 0.7]], device=dev)
    ag = interpol.AffineGrid(mat, [n, n, n])
    vox = B * n ** 3
    kw = dict(interpolation=3, bound="dct2", extrapolate=True)
    rec(res, "f3_affine_in_kernel_pull_256_cubic", timeit(lambda: interpol.grid_pull(x, ag, **kw), 3), vox, 2 * B * C * n ** 3 * 4)
    dense = ag.dense().reshape(1, n, n, n, 3).expand(B, n, n, n, 3).contiguous()
    rec(res, "f3_affine_dense_grid_pull_256_cubic", timeit(lambda: interpol.grid_pull(x, dense, **kw), 3), vox, 2 * B * C * n ** 3 * 4 + vox * 12)
    del dense, x

if "b" in which:    # backward of pull at config 2's shape (4x2x256^3 cubic dct2, sigma 2): the training step of a registration
    import bench
    from interpol import _hip
    inp, grid = bench.make_inputs(4, 2, 256, 2.0, dev, 1234)
    gout = torch.randn_like(inp)
    vox = 4 * 256 ** 3
    nb = vox * 12 + 3 * 4 * 2 * 256 ** 3 * 4
    B3, O3 = [3] * 3, [3] * 3
    rec(res, "cfg2shape_pull_backward_both_cubic", timeit(lambda: _hip.pull_backward(gout, inp, grid, B3, O3, 1, True, True), 3), vox, nb + vox * 12)
    rec(res, "cfg2shape_pull_backward_grid_only_cubic", timeit(lambda: _hip.pull_backward(gout, inp, grid, B3, O3, 1, False, True), 3), vox, nb)
    rec(res, "cfg2shape_pull_backward_grid_only_cubic_natural_tiles", timeit(lambda: _hip.pull_backward(gout, inp, grid, B3, O3, 1, False, True, flags=16 << 8), 3), vox, nb)
    rec(res, "cfg2shape_push_backward_both_cubic", timeit(lambda: _hip.push_backward(gout, inp, grid, B3, O3, 1, True, True), 3), vox, nb + vox * 12)
    rec(res, "cfg2shape_grid_grad_cubic", timeit(lambda: _hip.gather("grad", inp, grid, B3, O3, 1), 3), vox, vox * 12 + 4 * 4 * 2 * 256 ** 3 * 4)
    del inp, grid, gout

if "r" in which:    # config 2 vs the roughness of the deformation: identity + sigma * iid noise (voxels)
    import bench
    from interpol import backend
    for sigma in (0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0):
        inp, grid = bench.make_inputs(4, 2, 256, sigma, dev, 1234)
        kw = dict(interpolation=3, bound="dct2", extrapolate=True)
        vox = 4 * 256 ** 3
        nb = vox * 12 + 2 * 4 * 2 * 256 ** 3 * 4
        rec(res, "cfg2_pull_sigma_%g" % sigma, timeit(lambda: interpol.grid_pull(inp, grid, **kw), 3), vox, nb)
        rec(res, "cfg2_push_sigma_%g" % sigma, timeit(lambda: interpol.grid_push(inp, grid, **kw), 3), vox, nb)
        backend.rough_deformations = True
        try:
            rec(res, "cfg2_push_owner_sigma_%g" % sigma, timeit(lambda: interpol.grid_push(inp, grid, **kw), 3), vox, nb)
        finally:
            backend.rough_deformations = None
        if sigma in (0.0, 2.0, 4.0, 6.0):
            # round 5: the trilinear push behind the owner-computes probe (own_accumulate<1>) against its sample tiles alone
            kl = dict(interpolation=1, bound="dct2", extrapolate=True)
            rec(res, "cfg2shape_trilinear_push_sigma_%g" % sigma, timeit(lambda: interpol.grid_push(inp, grid, **kl), 3), vox, nb)
            backend.rough_deformations = False
            try:
                rec(res, "cfg2shape_trilinear_push_tiles_sigma_%g" % sigma, timeit(lambda: interpol.grid_push(inp, grid, **kl), 3), vox, nb)
            finally:
                backend.rough_deformations = None
        if sigma in (0.0, 1.0, 2.0, 4.0):
            # round 5: the reference's DEFAULT interpolation (trilinear) at the same shape -- routed: class-sorted tiles with K = 1 or the generic kernel
            from interpol import _hip
            kl = dict(interpolation=1, bound="dct2", extrapolate=True)
            rec(res, "cfg2shape_trilinear_pull_sigma_%g" % sigma, timeit(lambda: interpol.grid_pull(inp, grid, **kl), 3), vox, nb)
            rec(res, "cfg2shape_trilinear_pull_generic_sigma_%g" % sigma, timeit(lambda: _hip.gather("pull", inp, grid, [3] * 3, [1] * 3, 1, flags=_hip.FLAG_NO_FASTPATH), 3), vox, nb)
            rec(res, "cfg2shape_trilinear_backward_grid_only_sigma_%g" % sigma, timeit(lambda: _hip.pull_backward(inp, inp, grid, [3] * 3, [1] * 3, 1, False, True), 3), vox, nb + vox * 12)
        del inp, grid

if "r" in which:    # SURVEY 8(d)'s smooth variant: 12^3 control points, sigma = 2 / 8 voxels, cubic-upsampled displacement (a registration field)
    import bench
    for amp in (2.0, 8.0):
        inp, grid = bench.make_inputs(4, 2, 256, 0.0, dev, 1234)
        ctrl = torch.randn(4, 3, 12, 12, 12, generator=g, device=dev) * amp
        disp = interpol.resize(ctrl, shape=[256] * 3, anchor="e", interpolation=3, bound="dct2", prefilter=True)
        grid = grid + disp.permute(0, 2, 3, 4, 1)
        del ctrl, disp
        kw = dict(interpolation=3, bound="dct2", extrapolate=True)
        vox = 4 * 256 ** 3
        nb = vox * 12 + 2 * 4 * 2 * 256 ** 3 * 4
        rec(res, "cfg2_pull_smooth_field_amp_%g" % amp, timeit(lambda: interpol.grid_pull(inp, grid, **kw), 3), vox, nb)
        rec(res, "cfg2_push_smooth_field_amp_%g" % amp, timeit(lambda: interpol.grid_push(inp, grid, **kw), 3), vox, nb)
        gout = torch.randn_like(inp)
        from interpol import _hip
        rec(res, "cfg2_pull_backward_both_smooth_field_amp_%g" % amp, timeit(lambda: _hip.pull_backward(gout, inp, grid, [3] * 3, [3] * 3, 1, True, True), 3), vox, nb + vox * 12 + 4 * 2 * 256 ** 3 * 4)
        del inp, grid, gout

print(json.dumps(res, indent=1))
