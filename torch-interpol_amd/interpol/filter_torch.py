"""`spline_coeff` along one dim in plain PyTorch: the device-generic counterpart of
csrc/prefilter.hip for tensors the HIP library does not serve (CPU tensors).
Reference: interpol/coeff.py:35-284 -- poles 35-65, gain 69-73, initial values 82-179
(dft / dct1 / dct2 classes; zero -> dct1, replicate -> dct2: 237-242), final values 183-227,
the two first-order recursions 258-284.  The recursions are linear, so they run as a loop over
the line with every other dim vectorised (the reference does the same, through TorchScript)."""
import math

import torch


def poles(order):
    if order in (0, 1):
        return []
    if order == 2:
        return [math.sqrt(8.) - 3.]
    if order == 3:
        return [math.sqrt(3.) - 2.]
    if order == 4:
        return [math.sqrt(664. - math.sqrt(438976.)) + math.sqrt(304.) - 19.,
                math.sqrt(664. + math.sqrt(438976.)) - math.sqrt(304.) - 19.]
    if order == 5:
        return [math.sqrt(67.5 - math.sqrt(4436.25)) + math.sqrt(26.25) - 6.5,
                math.sqrt(67.5 + math.sqrt(4436.25)) - math.sqrt(26.25) - 6.5]
    if order == 6:
        return [-0.488294589303044755130118038883789062112279161239377608394,
                -0.081679271076237512597937765737059080653379610398148178525368,
                -0.00141415180832581775108724397655859252786416905534669851652709]
    if order == 7:
        return [-0.5352804307964381655424037816816460718339231523426924148812,
                -0.122554615192326690515272264359357343605486549427295558490763,
                -0.0091486948096082769285930216516478534156925639545994482648003]
    raise NotImplementedError('spline order > 7')


def _bound_class(bound):
    # coeff.py:237-254: zero / dct1 -> 0, replicate / dct2 -> 1, dft -> 2; dst1 / dst2 are not implemented
    if bound in (0, 2):
        return 0
    if bound in (1, 3):
        return 1
    if bound == 6:
        return 2
    raise NotImplementedError('spline prefilter: boundary condition dst1 / dst2 is not implemented (reference coeff.py:243-244)')


def _f32(x):
    # the reference builds its pole powers from the pole rounded through float32 (TorchScript as_tensor)
    return float(torch.tensor(x, dtype=torch.float32))


def spline_filter_(data, bound, order, dim, src=None):
    """In-place prefilter of `data` along `dim` (reading `src` first when given)."""
    if src is not None and src is not data:
        data.copy_(src)
    ps = poles(order)
    n = data.shape[dim]
    cls = _bound_class(bound)
    if not ps or n == 1:                                              # coeff.py:264-265
        return data
    c = data.movedim(dim, 0)                                          # a view: the line runs along axis 0
    work = c.to(torch.float64) if c.dtype in (torch.float16, torch.bfloat16) else c
    gain = 1.
    for p in ps:
        gain *= (1. - p) * (1. - 1. / p)
    work = work * gain if work is not c else work.mul_(gain)
    idx = torch.arange(n, device=data.device, dtype=work.dtype).reshape([n] + [1] * (work.dim() - 1))
    for p in ps:
        pf = _f32(p)
        max_iter = int(math.ceil(-30. / math.log(abs(p))))
        # ---- initial value (coeff.py:82-179)
        if cls == 0:
            if max_iter < n:
                init = (work[:max_iter] * pf ** idx[:max_iter]).sum(0)
            else:
                pn = p ** (n - 1)
                w = pf ** idx + (pn * pn) / pf ** idx
                w[0] = 1.; w[n - 1] = pn
                init = (work * w).sum(0) / (1. - pn * pn)
        elif cls == 1:
            pn = p ** n
            w = pf ** idx + pn * pf ** (n - 1 - idx)
            init = (work * w).sum(0) * (p / (1. - pn * pn)) + work[0]
        else:
            m = min(max_iter, n)
            w = torch.zeros_like(idx)
            w[0] = 1.
            if m > 1:
                w[n - m + 1:] = pf ** (n - idx[n - m + 1:])
            init = (work * w).sum(0) / (1. - p ** m)
        # ---- causal pass
        work[0] = init
        for i in range(1, n):
            work[i] += p * work[i - 1]
        # ---- final value (coeff.py:183-227)
        if cls == 0:
            fin = (p * work[n - 2] + work[n - 1]) * (p / (p * p - 1.))
        elif cls == 1:
            fin = work[n - 1] * (p / (p - 1.))
        else:
            m = min(max_iter, n)
            dot = (work[:m - 1] * pf ** (idx[:m - 1] + 2)).sum(0) if m > 1 else 0.
            fin = (dot + p * work[n - 1]) / (p ** m - 1.)
        # ---- anticausal pass
        work[n - 1] = fin
        for i in range(n - 2, -1, -1):
            work[i] = (work[i + 1] - work[i]) * p
    if work is not c:
        c.copy_(work.to(c.dtype))
    return data
