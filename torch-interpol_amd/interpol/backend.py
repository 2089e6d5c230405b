"""Backend switch kept for drop-in compatibility with the reference
(`interpol/backend.py:1`).  The external `jitfields` package is not part of this
build; setting the flag makes the API raise instead of silently ignoring it."""
jitfields = False

# Scatter operators (grid_push / grid_count and the scatter halves of the backward passes) accumulate
# tile by tile in LDS in FIXED POINT scaled by the tile's largest |source| (csrc/ops_tiled.hip): every
# contribution is rounded to about 2.5e-6 of that maximum at worst -- far below float32 rounding of the
# sums for ordinary images, but an ABSOLUTE error per tile: in a tile that holds both a spike and values
# many orders of magnitude smaller, the small voxels lose significant digits (the reference accumulates
# in float, relative to the local sum).  Set `exact_scatter = True` (or INTERPOL_EXACT_SCATTER=1 in the
# environment) to route every scatter through the generic kernels, which use float atomics like the
# reference's scatter_add_ -- two orders of magnitude slower at BASELINE config 2.
exact_scatter = False

# grid_push / grid_count: two organisations of the same scatter (same results within float32 rounding).  Times: 4x2x256^3 cubic
# dct2 on one MI355X, round 5 (HISTORY.md 4.2c, profiles/r05_rough_rows.txt).
#   * sample-stationary tiles (csrc/ops_tiled.hip): 16^3 tiles of samples accumulate in an LDS box that their
#     stencils must fit (33 x 33 x 32 lattice points); fastest for smooth deformations (2.2 ms at the identity,
#     3.5 ms under i.i.d. displacements of sigma = 2 voxels) and sharply slower beyond that (sigma = 3 / 4 / 6:
#     4.9 / 8.5 / 127 ms);
#   * owner-computes bricks (csrc/push_owner.hip): the samples are first sorted by target brick; cost nearly independent
#     of the deformation (2.8 - 3.6 ms from the identity to sigma = 6), about 22 bytes of workspace per sample point.
# `rough_deformations = None` (default): a probe kernel inside every call examines 128 tiles of the sample grid and
# gates the two organisations on the device (no host synchronisation, stateless, hipGraph-safe; ~50 us).
# Memory: under the default (and under True) every 3-D quadratic / cubic grid_push / grid_count -- and the image gradient of
# grid_pull's backward, which is such a push -- uses the bricks' workspace: about 22 bytes per sample point plus 1 KiB per 16^3 brick
# of the target (1.7 GB at 4x2x256^3), whichever organisation the probe then picks.  One buffer per (device, stream, host thread) is
# kept between calls and grown on demand, at most four of them (`release_workspaces()` frees them); it is only taken when it fits comfortably (at most half of
# the memory that is available), else the call falls back to the tiles, which need none (interpol/_hip.py: _optional_workspace).
# grid_pull (3-D quadratic / cubic, float32) is routed too, per TILE: the sample tiles of csrc/ops_sorted.hip leave the tiles whose
# LDS box cannot hold their stencils to bricks of the IMAGE (csrc/push_owner.hip: own_gather; 18 bytes of workspace per sample,
# allocated per call like the push's): 4x2x256^3 cubic under i.i.d. noise of sigma = 6 voxels 9.8 -> 3 ms; ~3 % on smooth fields.
# grid_grad, the grid gradient of grid_pull's backward and both gradients of grid_push's / grid_count's backward (float32, 3-D
# quadratic / cubic) take the same bricks: dense samplings altogether (a probe of the call), expanding ones stay with the tiles.
# True: always the bricks.  False: always the tiles (no workspace is allocated).
# Reproducibility: the organisation -- hence the summation order, hence the last bits of a result -- may differ between the
# settings and, under None, between calls whose probes decide differently; every organisation is held to the same tolerance
# (1e-5 of max|ref|).  Under None / True the backward of grid_pull is two passes (image gradient = grid_push of grad_out through
# this router, then the grid gradient), not the fused kernel.
rough_deformations = None


def release_workspaces():
    """The routed organisations keep ONE workspace per (device, stream, host thread) between calls (interpol/_hip.py:
    _optional_workspace; 1.7 GB at 4x2x256^3 once a push has run, shared by pull / push / backward; at most four are kept): this
    gives them back to torch's allocator -- call it next to torch.cuda.empty_cache()."""
    from . import _hip
    _hip.release_workspaces()


def want_exact_scatter():
    import os
    return bool(exact_scatter) or os.environ.get("INTERPOL_EXACT_SCATTER", "0") not in ("", "0")
