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

# grid_push / grid_count of VERY rough deformations (displacements that differ by more than ~8 voxels
# between neighbouring samples): the default scatter works on 16^3 tiles of samples whose stencils must
# fit a 32^3 box in LDS, and slows down sharply beyond that (4x2x256^3 cubic, i.i.d. displacements of
# sigma = 2 / 3 / 4 / 6 voxels: 3.5 / 5.0 / 8.6 / 126 ms).  `rough_deformations = True` selects the
# target-stationary organisation (csrc/push_binned.hip) whose cost does not depend on the deformation
# (6 - 7 ms in all those cases); it takes 24 bytes of workspace per sample point.
rough_deformations = False


def want_exact_scatter():
    import os
    return bool(exact_scatter) or os.environ.get("INTERPOL_EXACT_SCATTER", "0") not in ("", "0")
