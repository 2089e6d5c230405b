"""GPU property tests at BASELINE.json's FULL configuration sizes (the oracle cannot run there in
seconds): size-independent properties of the operators -- adjointness, partition of unity, mass
conservation, count == push(1), linearity, identity exactness, the interpolation property of the
prefilter -- plus agreement of the tiled and generic kernels on one batch item.  Tolerances are
stated per check; sums are accumulated in float64."""
import pytest
import torch

import interpol
from interpol import _hip
from interpol.distributed import push_count_shared

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


def _cfg2(seed=1234):
    g = torch.Generator(device=DEV).manual_seed(seed)
    inp = torch.randn([4, 2, 256, 256, 256], generator=g, device=DEV)
    grid = torch.randn([4, 256, 256, 256, 3], generator=g, device=DEV).mul_(2.0)
    grid += interpol.identity_grid([256] * 3, device=DEV)
    return inp, grid


def test_cfg2_full_size_pull_push_properties():
    """configs[1]: 4x2x256^3 fp32, cubic, dct2, i.i.d. sigma = 2 deformation."""
    inp, grid = _cfg2()
    kw = dict(interpolation=3, bound="dct2", extrapolate=True)
    out = interpol.grid_pull(inp, grid, **kw)
    assert out.shape == inp.shape and torch.isfinite(out).all()
    # adjointness <pull x, y> = <x, push y>  (float32 kernels: 1e-5 of the Cauchy-Schwarz scale)
    y = torch.randn_like(out)
    psh = interpol.grid_push(y, grid, **kw)
    lhs, rhs = _dot(out, y), _dot(inp, psh)
    scale = (float(out.double().norm()) * float(y.double().norm()))
    assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)
    # partition of unity: a constant image is reproduced (dct2 has no sign flips), extrapolate=True
    one = torch.ones_like(inp[:, :1])
    assert float((interpol.grid_pull(one, grid, **kw) - 1).abs().max()) < 1e-5
    # count == push(1), and the splatted mass is conserved: sum(count) = number of samples
    cnt = interpol.grid_count(grid, **kw)
    assert _rel(cnt, interpol.grid_push(one, grid, **kw)) < 2e-6
    n = grid[..., 0].numel()
    assert abs(float(cnt.double().sum()) - n) < 1e-6 * n
    assert abs(float(psh.double().sum()) - float(y.double().sum())) < 1e-5 * float(y.double().abs().sum())
    # linearity of pull
    x2 = torch.randn_like(inp)
    lin = interpol.grid_pull(2.5 * inp + x2, grid, **kw)
    assert _rel(lin, 2.5 * out + interpol.grid_pull(x2, grid, **kw)) < 1e-5
    # tiled and generic kernels agree (one batch item; the generic push takes ~0.1 s)
    b, o = [3] * 3, [3] * 3
    assert _rel(out[:1], _hip.gather("pull", inp[:1], grid[:1], b, o, 1, flags=_hip.FLAG_NO_FASTPATH)) < 5e-6
    assert _rel(psh[:1], _hip.scatter("push", y[:1], grid[:1], None, b, o, 1, flags=_hip.FLAG_NO_FASTPATH)) < 5e-6


@pytest.mark.timeout(900)
def test_cfg2_full_size_against_the_oracle():
    """Round 6: the headline workload itself -- configs[1] at FULL size, default routing (pull_sorted; own_bin + own_accumulate behind the
    probe) -- compared DIRECTLY with the C oracle on all host cores (float64 evaluation of the same float32 inputs): two of the four batch
    items (the oracle takes ~3 s per item and operator), pull and push + count, rtol 1e-5 + atol 1e-5 max|ref|."""
    import os
    import numpy as np
    from oracle import oracle
    inp, grid = _cfg2()
    b, o = [3] * 3, [3] * 3
    pull = _hip.gather("pull", inp, grid, b, o, 1)
    push = _hip.scatter("push", inp, grid, [256] * 3, b, o, 1, with_count=True)
    oracle.set_threads(os.cpu_count() or 8)
    try:
        for item in (0, 3):
            x64, g64 = inp[item:item + 1].cpu().double(), grid[item:item + 1].cpu().double()
            for name, got, ref in (("pull", pull[item:item + 1], oracle.grid_pull(x64, g64, b, o, 1)),
                                   ("push", push[item:item + 1, :2], oracle.grid_push(x64, g64, [256] * 3, b, o, 1)),
                                   ("count", push[item:item + 1, 2:], oracle.grid_count(g64, [256] * 3, b, o, 1))):
                ref = np.asarray(ref, dtype=np.float64)
                err = np.abs(got.cpu().double().numpy() - ref)
                tol = 1e-5 * np.abs(ref) + 1e-5 * np.abs(ref).max()
                assert (err <= tol).all(), (name, item, float(err.max()), float(np.abs(ref).max()))
    finally:
        oracle.set_threads(1)


def test_cfg2_full_size_identity_and_prefilter():
    inp, _ = _cfg2(7)
    inp = inp[:2]
    ident = interpol.identity_grid([256] * 3, device=DEV)[None].expand(2, -1, -1, -1, -1).contiguous()
    # trilinear sampling on the identity lattice is exact
    assert torch.equal(interpol.grid_pull(inp, ident, interpolation=1, bound="dct2", extrapolate=True), inp)
    # nearest: bit-exact copy
    assert torch.equal(interpol.grid_pull(inp, ident, interpolation=0, bound="dct2", extrapolate=True), inp)
    # cubic with prefilter interpolates: the samples come back (fp32 recursion: 2e-5)
    back = interpol.grid_pull(inp, ident, interpolation=3, bound="dct2", extrapolate=True, prefilter=True)
    assert _rel(back, inp) < 2e-5
    # displacement form of the same call
    zero = torch.zeros_like(ident)
    assert torch.equal(interpol.grid_pull(inp, zero, interpolation=1, bound="dct2", extrapolate=True, displacement=True), inp)


def test_cfg3_full_size_grad_and_backward():
    """configs[2]: 8x1x192^3 fp32, order 5, dft: grid_grad + autograd backward of grid_pull."""
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 192
    inp = torch.randn([8, 1, n, n, n], generator=g, device=DEV)
    grid = torch.randn([8, n, n, n, 3], generator=g, device=DEV).mul_(2.0) + interpol.identity_grid([n] * 3, device=DEV)
    kw = dict(interpolation=5, bound="dft", extrapolate=True)
    # the gradient of a constant image vanishes (derivative weights sum to zero)
    gc = interpol.grid_grad(torch.full_like(inp, 3.0), grid, **kw)
    assert float(gc.abs().max()) < 1e-5 * 3.0                                # (1e-5 of the image's amplitude)
    # backward of pull w.r.t. the image is its adjoint; w.r.t. the grid it is sum_c gout * grid_grad
    x = inp.clone().requires_grad_(True)
    gr = grid.clone().requires_grad_(True)
    out = interpol.grid_pull(x, gr, **kw)
    gout = torch.randn_like(out)
    out.backward(gout)
    lhs, rhs = _dot(out.detach(), gout), _dot(inp, x.grad)
    assert abs(lhs - rhs) <= 1e-5 * float(out.detach().double().norm()) * float(gout.double().norm())
    gg = interpol.grid_grad(inp, grid, **kw)                                   # (B, C, *out, 3)
    want = (gg * gout.unsqueeze(-1)).sum(1)
    assert _rel(gr.grad, want) < 1e-5


@pytest.mark.timeout(900)
def test_cfg3_and_cfg5_full_size_against_the_oracle():
    """Round 6: configs[2] (8x1x192^3, order 5, dft) and configs[4] (32x3x1024^2 bf16 storage, orders [2,3], bounds [dct1,dst2]) at FULL
    size against the C oracle on all host cores, one batch item each: cfg3 pull, grid_grad and both gradients of the pull's backward;
    cfg5 pull and push (bf16 tolerance 1e-2 of max|ref| on the rounded inputs)."""
    import os
    import numpy as np
    from oracle import oracle

    def close(name, got, ref, rtol, atol_rel):
        ref = np.asarray(ref, dtype=np.float64)
        err = np.abs(got.detach().cpu().double().numpy() - ref)
        assert (err <= rtol * np.abs(ref) + atol_rel * np.abs(ref).max()).all(), (name, float(err.max()), float(np.abs(ref).max()))

    oracle.set_threads(os.cpu_count() or 8)
    try:
        g = torch.Generator(device=DEV).manual_seed(3)
        n = 192
        inp = torch.randn([8, 1, n, n, n], generator=g, device=DEV)
        grid = torch.randn([8, n, n, n, 3], generator=g, device=DEV).mul_(2.0) + interpol.identity_grid([n] * 3, device=DEV)
        gout = torch.randn([8, 1, n, n, n], generator=g, device=DEV)
        b, o = [6] * 3, [5] * 3
        pull = _hip.gather("pull", inp, grid, b, o, 1)
        grad = _hip.gather("grad", inp, grid, b, o, 1)
        gi, gg = _hip.pull_backward(gout, inp, grid, b, o, 1, True, True)
        it = 5
        x64, g64, go64 = inp[it:it + 1].cpu().double(), grid[it:it + 1].cpu().double(), gout[it:it + 1].cpu().double()
        close("cfg3 pull", pull[it:it + 1], oracle.grid_pull(x64, g64, b, o, 1), 1e-5, 1e-5)
        close("cfg3 grad", grad[it:it + 1], oracle.grid_grad(x64, g64, b, o, 1), 2e-5, 2e-5)
        ri, rg = oracle.grid_pull_backward(go64, x64, g64, b, o, 1)
        close("cfg3 backward image", gi[it:it + 1], ri, 1e-5, 1e-5)
        close("cfg3 backward grid", gg[it:it + 1], rg, 2e-5, 2e-5)
        del inp, grid, gout, pull, grad, gi, gg
        g = torch.Generator(device=DEV).manual_seed(5)
        x = torch.randn(32, 3, 1024, 1024, generator=g, device=DEV).to(torch.bfloat16)
        g2 = torch.randn([32, 1024, 1024, 2], generator=g, device=DEV).mul_(2.0) + interpol.identity_grid([1024, 1024], device=DEV)
        b, o = [2, 5], [2, 3]
        pull = _hip.gather("pull", x, g2, b, o, 1)
        push = _hip.scatter("push", x, g2, [1024, 1024], b, o, 1)
        it = 17
        x64, g64 = x[it:it + 1].float().cpu().double(), g2[it:it + 1].cpu().double()
        close("cfg5 pull", pull[it:it + 1].float(), oracle.grid_pull(x64, g64, b, o, 1), 1e-2, 1e-2)
        close("cfg5 push", push[it:it + 1].float(), oracle.grid_push(x64, g64, [1024, 1024], b, o, 1), 1e-2, 1e-2)
    finally:
        oracle.set_threads(1)


def test_cfg4_full_size_shared_target_mass():
    """configs[3]: sources 1x128^3 -> shared 512^3, order 3, replicate (8 sources: one GPU's share)."""
    g = torch.Generator(device=DEV).manual_seed(4)
    nsrc, n, m = 8, 128, 512
    x = torch.randn(nsrc, 1, n, n, n, generator=g, device=DEV)
    grid = torch.randn([nsrc, n, n, n, 3], generator=g, device=DEV).mul_(2.0)
    grid += interpol.identity_grid([n] * 3, device=DEV) * ((m - 1) / (n - 1))
    push, count = push_count_shared(x, grid, [m] * 3, interpolation=3, bound="replicate", extrapolate=True, reduce="none")
    assert push.shape == (1, m, m, m) and count.shape == (1, m, m, m)
    # replicate folds the weights back into the lattice: every sample deposits exactly its value / 1
    nsamp = nsrc * n ** 3
    assert abs(float(count.double().sum()) - nsamp) < 1e-6 * nsamp
    assert abs(float(push.double().sum()) - float(x.double().sum())) < 1e-5 * float(x.double().abs().sum())
    assert float(count.min()) >= 0
    # the same result from the scatter kernels on one source (the brick path is the default here)
    ref = _hip.scatter("push", x[:1], grid[:1], [m] * 3, [1] * 3, [3] * 3, 1, with_count=True)
    one_p, one_c = push_count_shared(x[:1], grid[:1], [m] * 3, interpolation=3, bound="replicate", extrapolate=True, reduce="none")
    assert _rel(one_p[None], ref[:, :1]) < 5e-6 and _rel(one_c[None], ref[:, 1:]) < 5e-6
    # round 6: all 8 sources (one GPU's share of the 8-GPU run) against the C oracle, source by source, summed in float64
    import os
    import numpy as np
    from oracle import oracle
    oracle.set_threads(os.cpu_count() or 8)
    try:
        want_p, want_c = np.zeros([m] * 3), np.zeros([m] * 3)
        for i in range(nsrc):
            x64, g64 = x[i:i + 1].cpu().double(), grid[i:i + 1].cpu().double()
            want_p += np.asarray(oracle.grid_push(x64, g64, [m] * 3, [1] * 3, [3] * 3, 1), dtype=np.float64)[0, 0]
            want_c += np.asarray(oracle.grid_count(g64, [m] * 3, [1] * 3, [3] * 3, 1), dtype=np.float64)[0, 0]
    finally:
        oracle.set_threads(1)
    for name, got, want in (("push", push, want_p), ("count", count, want_c)):
        err = np.abs(got[0].cpu().double().numpy() - want)
        assert (err <= 1e-5 * np.abs(want) + 1e-5 * np.abs(want).max()).all(), (name, float(err.max()), float(np.abs(want).max()))


def test_cfg5_full_size_2d_bf16():
    """configs[4] (one GPU's share: 32 of 256 images): 2-D 32x3x1024^2 bf16, orders [2,3,5]->[2,3],
    bounds [dct1,dst2,zero]->[dct1,dst2], fp32 grid, spline_coeff_nd prefilter."""
    g = torch.Generator(device=DEV).manual_seed(5)
    B, C, n = 32, 3, 1024
    x = torch.randn(B, C, n, n, generator=g, device=DEV).to(torch.bfloat16)
    grid = torch.randn([B, n, n, 2], generator=g, device=DEV).mul_(2.0) + interpol.identity_grid([n, n], device=DEV)
    kw = dict(interpolation=[2, 3, 5], bound=["dct1", "dst2", "zero"], extrapolate=True)
    out = interpol.grid_pull(x, grid, **kw)
    assert out.dtype == torch.bfloat16 and out.shape == x.shape
    # bf16 storage, fp32 math: the result is the fp32 result rounded to bf16
    ref = interpol.grid_pull(x.float(), grid, **kw)
    assert _rel(out.float(), ref) < 1e-2
    # adjointness in fp32 on the same grid
    y = torch.randn(B, C, n, n, generator=g, device=DEV)
    lhs, rhs = _dot(ref, y), _dot(x.float(), interpol.grid_push(y, grid, **kw))
    assert abs(lhs - rhs) <= 1e-5 * float(ref.double().norm()) * float(y.double().norm())
    # bf16 PUSH at full size: bf16 in / out, fp32 accumulation, narrowed once
    pb = interpol.grid_push(x, grid, **kw)
    assert pb.dtype == torch.bfloat16 and pb.shape == x.shape
    pf = interpol.grid_push(x.float(), grid, **kw)
    assert _rel(pb.float(), pf) < 1e-2
    lhs, rhs = _dot(interpol.grid_pull(y, grid, **kw), x.float()), _dot(y, pf)          # <pull y, x> = <y, push x>
    assert abs(lhs - rhs) <= 1e-5 * float(y.double().norm()) * float(pf.double().norm())
    del pb, pf
    # prefilter + sampling on the identity lattice interpolates (dct1 / dct2 along the two dims)
    xf = x.float()
    coeff = interpol.spline_coeff_nd(xf, interpolation=[2, 3], bound=["dct1", "dct2"], dim=2)
    ident = interpol.identity_grid([n, n], device=DEV)[None].expand(B, -1, -1, -1).contiguous()
    back = interpol.grid_pull(coeff, ident, interpolation=[2, 3], bound=["dct1", "dct2"], extrapolate=True)
    assert _rel(back, xf) < 2e-5
    cb = interpol.spline_coeff_nd(x, interpolation=[2, 3], bound=["dct1", "dct2"], dim=2)
    assert cb.dtype == torch.bfloat16 and _rel(cb.float(), coeff) < 2e-2
