"""Regular (tensor-product) sampling lattices: the argument conventions and the 1-D coordinate
vectors shared by `resize` and `restrict` (reference semantics: interpol/resize.py:70-110,
interpol/restrict.py:66-112).

Both operators relate a lattice of `n_pts` points to a reference lattice of `n_ref` voxels by a
ratio r (points per reference voxel):  `resize` places the OUTPUT points in the input's voxel
coordinates (n_pts = n_out, n_ref = n_in, r = n_out / n_in); `restrict`, its adjoint, places the
INPUT points in the output's (n_pts = n_in, n_ref = n_out, r = n_in / n_out).  One formula per
anchor serves both.
"""
import torch

from .utils import make_list

# anchor -> (offset of point 0, spacing) of the points in reference voxels, as functions of (r, n_pts, n_ref)
_ANCHORS = {
    'c': lambda r, n_pts, n_ref: None,                                             # corner centres aligned: linspace
    'e': lambda r, n_pts, n_ref: (0.5 * (n_ref / n_pts - 1), n_ref / n_pts),       # corner edges aligned
    'f': lambda r, n_pts, n_ref: (0.0, 1 / r),                                     # first voxel aligned, exact ratio
    'l': lambda r, n_pts, n_ref: ((n_ref - 1) - (n_pts - 1) / r, 1 / r),           # last voxel aligned, exact ratio
}


def plan(image, factor, shape, anchor, shrink):
    """Normalise (factor, shape, anchor) for an image whose last dims are spatial.
    `shrink`: the factor divides the extents (restrict) instead of multiplying them (resize).
    Returns (nb_dim, anchors as letters, ratios, input extents, output extents)."""
    factors = make_list(factor) if factor else []
    extents = make_list(shape) if shape else []
    anchors = make_list(anchor)
    nb_dim = max(len(factors), len(extents), len(anchors)) or (image.dim() - 2)
    if not factors and not extents:
        raise ValueError('One of `factor` or `shape` must be provided')
    letters = [str(a)[0].lower() for a in make_list(anchors, nb_dim)]
    src = list(image.shape[-nb_dim:])
    ratios = make_list(factors, nb_dim) if factors else None
    if extents:
        dst = make_list(extents, nb_dim)
    else:
        dst = [int(n / r) if shrink else int(n * r) for n, r in zip(src, ratios)]
    if ratios is None:
        ratios = [(n / m) if shrink else (m / n) for n, m in zip(src, dst)]
    return nb_dim, letters, ratios, src, dst


def positions(letter, ratio, n_pts, n_ref, **backend):
    """Coordinates (reference voxels) of the n_pts lattice points along one dim, and their spacing."""
    try:
        law = _ANCHORS[letter](ratio, n_pts, n_ref)
    except KeyError:
        raise ValueError('Unknown anchor {}'.format(letter)) from None
    if law is None:
        return torch.linspace(0, n_ref - 1, n_pts, **backend), (n_ref - 1) / (n_pts - 1) if n_pts > 1 else 1.0
    offset, spacing = law
    x = torch.arange(0., n_pts, **backend)
    if letter == 'e':
        return x * spacing + offset, spacing         # (same operation order as the reference: scale, then shift)
    return (x / ratio + offset if offset else x / ratio), spacing
