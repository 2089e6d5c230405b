"""TEST-ONLY kernel table: routes the product's operator seam (interpol.ops) to
the CPU oracle so that the host logic (shape conventions, alias handling,
autograd wiring, sharding) can be exercised without a GPU.  Never used by the
product path."""
import torch

from oracle import oracle


def _g(grid, displacement=False):
    """dense coordinates of a grid argument: a SeparableGrid lattice is expanded, a displacement
    field gets the identity lattice added (interpol.add_identity_grid) -- for the oracle"""
    if hasattr(grid, 'dense'):
        return grid.dense()
    if displacement:
        import interpol
        return interpol.add_identity_grid(grid.detach())
    return grid.detach()


def _t(x, like):
    return torch.as_tensor(x).to(like.dtype)


class OracleKernels:
    @staticmethod
    def pull(inp, grid, bound, order, extrapolate, displacement=False):
        return oracle.grid_pull(inp.detach(), _g(grid, displacement), bound, order, extrapolate)

    @staticmethod
    def grad(inp, grid, bound, order, extrapolate, displacement=False):
        return oracle.grid_grad(inp.detach(), _g(grid, displacement), bound, order, extrapolate)

    @staticmethod
    def hess(inp, grid, bound, order, extrapolate, displacement=False):
        return oracle.grid_hess(inp.detach(), _g(grid, displacement), bound, order, extrapolate)

    @staticmethod
    def push(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return oracle.grid_push(inp.detach(), _g(grid, displacement), shape, bound, order, extrapolate)

    @staticmethod
    def count(grid, shape, bound, order, extrapolate, displacement=False):
        return oracle.grid_count(_g(grid, displacement), shape, bound, order, extrapolate)

    @staticmethod
    def pushgrad(inp, grid, shape, bound, order, extrapolate, displacement=False):
        return oracle.grid_pushgrad(inp.detach(), _g(grid, displacement), shape, bound, order, extrapolate)

    @staticmethod
    def push_shared_(out, inp, grid, bound, order, extrapolate, with_count=False):
        shape = list(out.shape[2:])
        if with_count:
            OracleKernels.push_shared_(out[:, :-1], inp, grid, bound, order, extrapolate)
            return OracleKernels.push_shared_(out[:, -1:], None, grid, bound, order, extrapolate)
        if inp is None:
            r = oracle.grid_count(_g(grid), shape, bound, order, extrapolate)
        else:
            r = oracle.grid_push(inp.detach(), _g(grid), shape, bound, order, extrapolate)
        out += torch.as_tensor(r).sum(0, keepdim=True).to(out.dtype)
        return out

    @staticmethod
    def pull_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        gi, gg = oracle.grid_pull_backward(grad.detach(), inp.detach(), _g(grid, displacement), bound, order, extrapolate)
        return (gi if need_inp else None), (gg if need_grid else None)

    @staticmethod
    def push_backward(grad, inp, grid, bound, order, extrapolate, need_inp, need_grid, displacement=False):
        gi, gg = oracle.grid_push_backward(grad.detach(), inp.detach(), _g(grid, displacement), bound, order, extrapolate)
        return (gi if need_inp else None), (gg if need_grid else None)

    @staticmethod
    def count_backward(grad, grid, bound, order, extrapolate, displacement=False):
        return oracle.grid_count_backward(grad.detach(), _g(grid, displacement), bound, order, extrapolate)

    @staticmethod
    def spline_filter_(data, bound, order, dim, src=None):
        data.copy_(oracle.spline_coeff((data if src is None else src).detach(), bound, order, dim=dim))
        return data

    @staticmethod
    def resample1d(src, lin, dim, order, bound, extrapolate, mode, adjoint, n_lattice):
        """1-D pass through the oracle's 1-D pull / push (batch = every other index)."""
        x = src.detach().movedim(dim, -1)
        lead = x.shape[:-1]
        x = x.reshape(-1, 1, x.shape[-1])
        grid = lin.detach().to(x.dtype).reshape(1, -1, 1).expand(x.shape[0], -1, 1)
        if adjoint:
            r = oracle.grid_push(x, grid, [int(n_lattice)], [bound], [order], extrapolate)
        else:
            r = oracle.grid_pull(x, grid, [bound], [order], extrapolate)
        r = torch.as_tensor(r).to(src.dtype)
        return r.reshape(*lead, r.shape[-1]).movedim(-1, dim).contiguous()

    @staticmethod
    def pull_labels(inp, grid, bound, order, extrapolate, displacement=False):
        """The reference's loop over the labels (api.py:194-205), with the oracle's pull."""
        g = _g(grid, displacement).float()
        out = torch.zeros([max(inp.shape[0], g.shape[0]), inp.shape[1], *g.shape[1:-1]], dtype=torch.int32)
        pmax = torch.zeros(out.shape, dtype=torch.float32)
        for label in inp.unique():
            soft = torch.as_tensor(oracle.grid_pull((inp == label).float(), g, bound, order, extrapolate))
            out[soft > pmax] = int(label)
            pmax = torch.max(pmax, soft)
        return out

    @staticmethod
    def push_count(inp, grid, shape, bound, order, extrapolate, displacement=False):
        a = torch.as_tensor(oracle.grid_push(inp.detach(), _g(grid, displacement), shape, bound, order, extrapolate))
        c = torch.as_tensor(oracle.grid_count(_g(grid, displacement), shape, bound, order, extrapolate))
        return torch.cat([a, c.expand(a.shape[0], 1, *c.shape[2:]).to(a.dtype)], 1)
