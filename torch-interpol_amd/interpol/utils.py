"""Small host-side helpers (shape broadcasting, list handling, grids).
Behavioural counterparts of reference interpol/utils.py:11-78 and
interpol/api.py:467-572."""
import torch


def make_list(x, n=None, **kwargs):
    """Listify; right-pad to length n with `default` (last element by default).
    reference interpol/utils.py:11-33"""
    x = list(x) if isinstance(x, (list, tuple)) else [x]
    if n and len(x) < n:
        x = x + [kwargs.get('default', x[-1])] * (n - len(x))
    return x


def expanded_shape(*shapes, side='left'):
    """Broadcast shapes (numpy rules); raises ValueError like the reference
    (interpol/utils.py:36-78)."""
    nb_dim = max([len(s) for s in shapes] + [0])
    out = [1] * nb_dim
    for shape1 in shapes:
        pad = [1] * (nb_dim - len(shape1))
        shape1 = pad + list(shape1) if side == 'left' else list(shape1) + pad
        new = []
        for s0, s1 in zip(out, shape1):
            if not (s0 == 1 or s1 == 1 or s0 == s1):
                raise ValueError('Incompatible shapes for broadcasting: {} and {}.'.format(s0, s1))
            new.append(max(s0, s1))
        out = new
    return tuple(out)


def identity_grid(shape, dtype=None, device=None):
    """(*shape, dim) tensor of voxel coordinates.  reference interpol/api.py:467-487"""
    axes = [torch.arange(float(s), dtype=dtype, device=device) for s in shape]
    return torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1)


def add_identity_grid_(disp):
    """Displacement -> transformation, in place.  reference interpol/api.py:490-513"""
    dim = disp.shape[-1]
    spatial = disp.shape[-dim - 1:-1]
    for d, n in enumerate(spatial):
        axis = torch.arange(n, dtype=disp.dtype, device=disp.device)
        view = [1] * (disp.dim() - 1)
        view[disp.dim() - 1 - dim + d] = n
        disp[..., d].add_(axis.reshape(view))
    return disp


def add_identity_grid(disp):
    """reference interpol/api.py:516-531"""
    return add_identity_grid_(disp.clone())


def affine_grid(mat, shape):
    """Dense grid from (..., D[+1], D+1) affine matrices.  reference interpol/api.py:534-572"""
    mat = torch.as_tensor(mat)
    shape = list(shape)
    D = len(shape)
    rows, cols = mat.shape[-2], mat.shape[-1]
    # same two refusals as the reference (ValueError both): the matrix maps D-dimensional homogeneous coordinates, and it
    # is either the full (D+1) x (D+1) matrix or its first D rows
    if cols != D + 1:
        raise ValueError("affine_grid: a %d-D lattice needs matrices with %d columns, got shape %s" % (D, D + 1, tuple(mat.shape)))
    if rows != D and rows != D + 1:
        raise ValueError("affine_grid: expected matrices of shape (..., %d, %d) or (..., %d, %d), got %s"
                         % (D, D + 1, D + 1, D + 1, tuple(mat.shape)))
    nb_dim = D
    grid = identity_grid(shape, mat.dtype, mat.device)
    lin = mat[..., :nb_dim, :nb_dim]
    off = mat[..., :nb_dim, -1]
    for _ in range(nb_dim):
        lin = lin.unsqueeze(-3)
        off = off.unsqueeze(-2)
    return torch.matmul(lin, grid.unsqueeze(-1)).squeeze(-1) + off
