/* ===========================================================================
 * interpol_hip.h -- C-ABI of libinterpol_hip.so, the MI355X (gfx950) kernels
 * behind the torch-interpol hot path.
 *
 * The reference (balbasty/torch-interpol @2024_10_08) has no FFI: its operator
 * seam is the Python module interpol/pushpull.py.  Each entry point below is
 * what a binding for that seam would call; the reference interface it
 * replaces is cited on every declaration.  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, element strides, int codes;
 *   - the caller owns every buffer (inputs, outputs, scratch); the library
 *     never retains a pointer to them.  State: the ONLY state of the library is
 *     the tile hand-back of csrc/defer.hip (see interpol_set_handback below):
 *     1 KiB of pinned host memory per device on first use, and up to 16 slots
 *     of 3 MiB of device memory per device, one per stream whose launches met
 *     stretched tiles -- recycled least-recently-used, released by
 *     interpol_release_stream(), freed at process exit;
 *   - kernels are enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); no internal synchronisation;
 *   - return value: 0 = ok, < 0 = INTERPOL_E_* (invalid argument, nothing
 *     launched), > 0 = a hipError_t raised by the launch;
 *   - boundary codes 0..6 = zero, replicate, dct1, dct2, dst1, dst2, dft
 *     (reference interpol/bounds.py:8-15); spline orders 0..7
 *     (interpol/splines.py:7-15); extrapolate 0 = no, 1 = yes, 2 = hist
 *     (interpol/bounds.py:18-21).
 * =========================================================================== */
#ifndef INTERPOL_HIP_H
#define INTERPOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INTERPOL_ABI_VERSION 1

/* storage types of image tensors / coordinate tensors */
enum {
    INTERPOL_F32  = 0,   /* float  storage, float math                        */
    INTERPOL_F64  = 1,   /* double storage, double math (coords must be F64)  */
    INTERPOL_BF16 = 2,   /* bf16 storage, float math   (coords must be F32)   */
    INTERPOL_F16  = 3    /* half storage, float math   (coords must be F32)   */
};

/* error codes (negative) */
enum {
    INTERPOL_OK          =  0,
    INTERPOL_E_DIM       = -1,   /* dim not in 1..3                            */
    INTERPOL_E_ORDER     = -2,   /* spline order not in 0..7 (splines.py:80)   */
    INTERPOL_E_BOUND     = -3,   /* bound code not in 0..6                     */
    INTERPOL_E_DTYPE     = -4,   /* unsupported dtype combination              */
    INTERPOL_E_SHAPE     = -5,   /* non-positive / too large extent            */
    INTERPOL_E_NULL      = -6,   /* required pointer is NULL                   */
    INTERPOL_E_EXTRAP    = -7,   /* extrapolate not in 0..2                    */
    INTERPOL_E_PREFILTER = -8,   /* prefilter bound dst1/dst2 (coeff.py:243)   */
    INTERPOL_E_SCRATCH   = -9,   /* scratch buffer too small                   */
    INTERPOL_E_STRIDE    = -10   /* stride pattern not supported (see below)   */
};

/* ---------------------------------------------------------------------------
 * Problem descriptor shared by all sampling operators.
 *
 *   "vol"  = the lattice that is INDEXED by the coordinates: the input image of
 *            pull/grad/hess, the target image of push/count/pushgrad.
 *            Layout (batch, channel, *vol_shape); any element strides.
 *   "grid" = the coordinates, layout (batch, *grid_shape, dim); component d
 *            addresses vol spatial dim d, in voxels (nd.py:44).
 *   "val"  = the per-sample values: the output of pull/grad/hess, the input of
 *            push/pushgrad.  Layout (batch, channel, *grid_shape[, dim[, dim]]).
 *
 * A batch stride of 0 broadcasts that tensor over the batch (api.py:122-126).
 * Strides are in ELEMENTS.  Unused trailing entries (dim < 3) are ignored.
 * Supported patterns (else INTERPOL_E_STRIDE): gather sources (vol of pull/grad/
 * hess) may have ANY non-negative strides; grid and val must be row-major
 * contiguous over their spatial (+component) dims with free batch/channel
 * strides; scatter targets (vol of push/count/pushgrad) must be dense, except
 * that a batch stride of 0 makes ONE shared (C, *shape) target into which every
 * batch item is accumulated (= grid_push(...).sum(0) of the reference).
 * One (batch, channel) image of vol must span < 4 GiB (32-bit byte offsets).
 * --------------------------------------------------------------------------- */
typedef struct interpol_problem {
    int32_t abi_version;        /* INTERPOL_ABI_VERSION                        */
    int32_t dim;                /* D in 1..3                                   */
    int32_t dtype;              /* INTERPOL_* of vol and val                   */
    int32_t grid_dtype;         /* INTERPOL_F32 or INTERPOL_F64                */
    int32_t extrapolate;        /* 0, 1, 2                                     */
    int32_t bound[3];           /* per spatial dim                             */
    int32_t order[3];           /* per spatial dim                             */
    int32_t flags;              /* INTERPOL_FLAG_*                             */
    int64_t batch;              /* B = max over tensors                        */
    int64_t channels;           /* C                                           */
    int64_t vol_shape[3];
    int64_t grid_shape[3];
    int64_t vol_stride[5];      /* batch, channel, s0, s1, s2                  */
    int64_t grid_stride[5];     /* batch, s0, s1, s2, component                */
    int64_t val_stride[7];      /* batch, channel, s0, s1, s2, d, e            */
} interpol_problem;

/* flags */
#define INTERPOL_FLAG_NO_FASTPATH   1   /* force the generic kernels (testing)           */
#define INTERPOL_FLAG_ACCUMULATE    2   /* push/count/pushgrad: do not zero the target (fp32 / fp64 targets only:
                                           INTERPOL_E_DTYPE for bf16 / f16, which are narrowed once) */
#define INTERPOL_FLAG_FORCE_TILED   4   /* take the LDS-tiled kernel wherever one exists (testing) */
/* Separable (tensor-product) coordinates, the grids of resize / restrict (resize.py:96-123,
 * restrict.py:88-117: `stack(meshgrid_ij(*lin), -1)`): `grid` points to the D coordinate vectors
 * lin_0 (grid_shape[0] values), lin_1, lin_2 stored back to back (grid_dtype); sample
 * (o_0, o_1, o_2) has coordinates (lin_0[o_0], lin_1[o_1], lin_2[o_2]) for every batch item.
 * grid_stride is ignored; no (B,*out,D) grid is read (-12 B/sample in 3-D).  The backward
 * entry points accept it only when grad_grid is not requested (else INTERPOL_E_STRIDE). */
#define INTERPOL_FLAG_SEPARABLE_GRID 8
/* `grid` holds DISPLACEMENTS in voxels, layout as a dense grid: the coordinates are
 * o_d + grid[b, o, d], the identity lattice being added in registers -- the fused form of
 * add_identity_grid_ (api.py:490-513) followed by the operator; same rounding (one float add of
 * the exactly representable index).  grad_grid of the backward entry points is then the
 * gradient w.r.t. the displacement (identical values).  Not combinable with SEPARABLE_GRID. */
#define INTERPOL_FLAG_DISPLACEMENT  16
/* interpol_push only: the target has channels + 1 channels, (B, C+1, *shape); channel C receives
 * the COUNT image (what interpol_count writes: the splatted ones, pushpull.py:106-142) in the same
 * pass over the grid -- push followed by count on the same grid is the usual pairing (normalised
 * splatting; SURVEY config 4).  vol_stride describe the (B, C+1, *shape) target. */
#define INTERPOL_FLAG_WITH_COUNT    32
/* interpol_push / interpol_count: organise the scatter OWNER-COMPUTES (push_owner.hip): the samples are
 * first sorted by the 16^3 brick of target lattice points their first tap falls into (one pass over the
 * inputs), then every brick is accumulated in LDS by one workgroup and added to the target with plain
 * loads and stores -- no global atomics: under replicate / dct1 / dct2 the stencils that leave the lattice
 * are folded back inside the LDS box of the brick at the end of the dim; what lies further out than 9
 * points, and the other boundary conditions, flush a thin shell of bricks with atomics.  The cost hardly
 * depends on the deformation (4x2x256^3 cubic: 2.8 - 3.6 ms from the identity to i.i.d. noise of sigma = 6
 * voxels), where the sample-stationary tiles need the stencils of a 16^3 tile of samples to fit a
 * 33 x 33 x 32 LDS box (2.2 ms at the identity, 3.5 at sigma = 2, 126 at sigma = 6).
 * 3-D, one order 1..3 or -- round 6 -- any mix of orders 1..3 on a dense grid (the cubic's bricks with every dim's own first tap and
 * weights: [1,2,3] at sigma = 6, 4x2x256^3: 3.9 ms where the tiles took 72; the gathers of interpol_pull_ws / interpol_grad_ws / the
 * backward passes likewise) (trilinear since round 5: 4x2x256^3, 2.3 - 3.1 ms from the identity to sigma = 6, where its tiles take
 * 1.3 - 16 ms; all orders 0, nearest neighbour, through the same bricks with the box held in FLOATS -- round 6: no fixed point at order
 * 0, one LDS float add per sample and channel, so a lattice point hit by one sample holds that sample's value bit for bit and a
 * non-finite source stays on its own lattice point, like iso0.py:65-118 -- 2.5 - 2.9 ms; under AUTO the probe chooses between the bricks
 * and the generic kernel's global atomics), float32 coordinates.  Needs the workspace announced by
 * interpol_scatter_workspace(); ignored (tiles / generic kernels) when it does not apply.
 *   INTERPOL_FLAG_BINNED_SCATTER: always;
 *   INTERPOL_FLAG_AUTO_SCATTER:   a probe kernel of the same call examines 128 tiles of the sample grid
 *     (owner-computes when samples fall outside the tiles' boxes, or -- two channels and more -- when
 *     roughness makes those boxes large) and writes a device-side gate; both organisations are enqueued, each kernel reads the gate on
 *     entry and one of the two returns at once (about 50 us of empty launches).  Stateless: the choice
 *     depends on the coordinates of this call alone; safe under hipGraph capture. */
#define INTERPOL_FLAG_BINNED_SCATTER 64
#define INTERPOL_FLAG_AUTO_SCATTER  (1 << 24)
/* (experimental, opt-in) interpol_pull_ws with the single-pass small-box tiles for smooth deformations in front of the class-sorted
 * tiles (pull_direct.hip): measured slower than the class-sorted tiles alone at config 2 (identity 1.16 against 1.05 ms, mixed
 * fields up to +26 %: profiles/r04_pull_steps.txt), so it is not the default. */
#define INTERPOL_FLAG_SMALL_TILES (1 << 25)
/* The sample coordinates are an AFFINE function of the sample index, x = A o + t -- the fused form of
 * affine_grid (api.py:534-572) followed by the operator: `grid` points to ONE D x (D+1) matrix [A | t]
 * (grid_dtype, row-major), evaluated in registers as ((A_d0 o_0) + A_d1 o_1 ...) + t_d with fused
 * multiply-adds; grid_stride is ignored and no (B,*out,D) grid is read (-4 D bytes per sample).  Not
 * combinable with SEPARABLE_GRID / DISPLACEMENT; no grad_grid (INTERPOL_E_STRIDE), like SEPARABLE_GRID. */
#define INTERPOL_FLAG_AFFINE_GRID   128

/* --- forward operators -------------------------------------------------------
 * interpol_pull      replaces pushpull.grid_pull      (interpol/pushpull.py:35-66;
 *                    nd.pull nd.py:80-143, iso1.pull*d, iso0.pull*d)
 *                    vol (B,C,*in) , grid (B,*out,D) -> val (B,C,*out)
 * interpol_push      replaces pushpull.grid_push      (pushpull.py:70-102; nd.push nd.py:146-213)
 *                    val (B,C,*in) , grid (B,*in,D)  -> vol (B,C,*shape), zero-filled here
 *                    unless INTERPOL_FLAG_ACCUMULATE
 * interpol_count     replaces pushpull.grid_count     (pushpull.py:106-142)
 *                    grid (B,*in,D) -> vol (B,1,*shape)     (p->channels must be 1)
 * interpol_grad      replaces pushpull.grid_grad      (pushpull.py:146-172; nd.grad nd.py:216-288)
 *                    vol , grid -> val (B,C,*out,D)
 * interpol_pushgrad  replaces pushpull.grid_pushgrad  (pushpull.py:176-203; nd.pushgrad nd.py:291-364)
 *                    val (B,C,*in,D) , grid -> vol (B,C,*shape)
 * interpol_hess      replaces pushpull.grid_hess      (pushpull.py:207-233; nd.hess nd.py:367-464)
 *                    vol , grid -> val (B,C,*out,D,D)
 *
 * Low-precision scatter: for dtype BF16/F16 the push-type operators need a
 * float accumulation buffer `scratch` of batch*channels*prod(vol_shape) floats
 * (scratch_bytes = that * 4); pass NULL/0 for F32/F64.
 *
 * Workspace of the scatters: with INTERPOL_FLAG_BINNED_SCATTER / INTERPOL_FLAG_AUTO_SCATTER, interpol_push /
 * interpol_count first sort the samples by target brick (push_owner.hip), which needs room for the sorted
 * records (18 B per sample + 4 B per sample and further channel, 2 KiB per brick):
 * interpol_scatter_workspace(p, count_only) returns the number of bytes `scratch` must then have
 * (it INCLUDES the fp32 accumulator of a BF16 / F16 target, which comes first), or 0 when the
 * organisation does not apply.  With a smaller (or no) scratch -- or one that is not aligned to 256 bytes (the
 * workspace holds 8- and 16-byte records) -- the operators fall back to the tiled / generic scatters: same results.
 *
 * Round 5: (a) a target SHARED by the batch items (batch stride 0) takes the same organisation -- the items' samples are sorted
 * into ONE brick grid (up to 512 runs per brick) and the bricks flush with plain loads and stores, no atomics; it applies once
 * all items together bring an eighth of a sample per target voxel (BASELINE config 4: 64 sources of 128^3 into 512^3, push + count
 * 16.3 -> 6.3 ms).  (b) Orders 4 and 5 (3-D, F32, private targets with at least a quarter of a sample per voxel):
 * interpol_scatter_workspace returns 16 B per sample + 1 KiB per brick for them and interpol_push / interpol_count -- and the image
 * gradient of interpol_pull_backward, which takes the workspace in its `scratch` -- go through bricks of the target (gather5.hip:
 * scatter5), behind a probe of the call under INTERPOL_FLAG_AUTO_SCATTER (smooth fields keep the LDS tiles): cost independent of
 * the deformation (8 x 1 x 192^3 order 5: 2.9 ms at sigma = 2, 3.1 ms at sigma = 6, where the tiles took 4.2 ms and fell off a cliff).
 * (c) 2-D, per-dim orders 1..3, F32 / BF16 / F16 sources, float32 coordinates (dense grids and displacement fields), private targets
 * with at least a quarter of a sample per pixel: interpol_scatter_workspace returns 16 B per sample + 1 KiB per 32 x 32 brick (plus
 * the accumulator of a 16-bit target) and interpol_push / interpol_count go through bricks of the target (scatter2d.hip) -- always
 * under INTERPOL_FLAG_BINNED_SCATTER, behind a probe of the call under INTERPOL_FLAG_AUTO_SCATTER (the bricks take the call when more than 50
 * pixels per million leave the lean tiles' 64 x 64 boxes AND the samples see a density of at least 0.6 per pixel; a sparser sampling -- a zoom --
 * stays with the tiles, below 0.22 it goes to the generic kernel; smooth fields keep the tiles at +3 %).  BASELINE config 5's shape (32 x 3 x 1024^2 bf16,
 * orders [2, 3]): 1.1 ms at every sigma, where the tiles take 1.04 ms at sigma = 2, 4.3 at 8 and 13.4 at 16.
 *
 * Accuracy of the LDS scatters (every fast path of interpol_push / interpol_count and of the scatter halves of the
 * backward operators; INTERPOL_FLAG_NO_FASTPATH selects the generic kernels, which add floats like the reference's
 * scatter_add_): contributions are summed in FIXED POINT scaled by the largest |source| of the tile / brick they belong
 * to, so the error of a lattice point is ABSOLUTE per tile, not relative to the point's own value.
 *   F32 / BF16 / F16, sample tiles (ops_tiled.hip, ops_tiled2d.hip): each addend rounded to 2^(h-29) of a power of two
 *     >= that maximum, h = the headroom bits the tile's sample density asks for (at most 7);
 *   F32 / BF16 / F16, bricks (push_owner.hip, "magic" format): each addend rounded to nearest at
 *     max|source| * wmax^3 * 2^-22 / 0.999 (wmax = 2/3 cubic, 3/4 quadratic: 2^-23.75 resp. 2^-23.25 of the brick's
 *     maximum), whatever the density; bricks whose stencil counts could overflow 32-bit sums use 64-bit sums of terms
 *     rounded at 2^-30 of a power of two >= the maximum; orders 4 - 5 (gather5.hip: scatter5): the same format in 32-bit
 *     sums, one channel per pass, wmax = 0.599 (order 4) / 0.55 (order 5): 2^-24.2 / 2^-24.6 of the brick's maximum per addend;
 *     2-D (scatter2d.hip): 32-bit sums, each addend rounded to nearest at M * wmax0 * wmax1 / U, M = the largest |source| of the sample
 *     tiles that reach the brick (rounded up to 8 bits), U = min(2^22 * 0.999, 2^31 * 0.99 / n) with n the records of the brick (its
 *     densest cell times the taps when n > 2048): 2^-22 .. 2^-23 of M at one sample per pixel, and never an overflow;
 *   F64 (push_f64.hip): each addend rounded at 2^-51 of a power of two > the tile's maximum -- about 2^-52 of the tile's
 *     largest source per term, again absolute per tile; tiles whose maximum is below 2^-970 use the generic arithmetic.
 * Sums inside a tile / brick are integers: order-free, bit-reproducible.
 * --------------------------------------------------------------------------- */
int64_t interpol_scatter_workspace(const interpol_problem *p, int32_t count_only);
int interpol_pull(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream);
/* interpol_pull with a workspace (the same seam, interpol/pushpull.py:69-104 -> nd.pull): deformation-independent cost for
 * 3-D quadratic / cubic F32 pulls.  With INTERPOL_FLAG_AUTO_SCATTER the sample tiles (ops_sorted.hip) run first and leave the
 * tiles whose LDS box cannot hold their stencils (more than 512 samples outside it: i.i.d. displacements beyond ~3.5 voxels,
 * folding fields) to a second organisation, which sorts those tiles' samples by the 16^3 brick of the image they read and gathers
 * brick by brick (push_owner.hip: own_bin in index mode + own_gather; 4x2x256^3 cubic, sigma = 6: 9.8 -> 3 ms) -- decided per
 * tile, on the device, no host synchronisation, hipGraph-safe.  INTERPOL_FLAG_BINNED_SCATTER: the bricks for every tile.
 * interpol_pull_workspace(p) returns the bytes `workspace` must have (18 B per sample + 2 KiB per brick + 4 B per tile; 0: the
 * organisation does not apply); with a smaller, missing or not 256-byte aligned workspace the call is interpol_pull.  The
 * workspace need not be cleared.  Same results within float32 rounding (sums in a different order). */
/* Orders 4 and 5 (3-D, float32; round 4, gather5.hip): the same two entry points, interpol_grad_ws and the grid gradient of
 * interpol_pull_backward serve them through bricks of the image of their own (16 B per sample + 1 KiB per brick of workspace).
 * AUTO: order 5 always (8 x 1 x 192^3, sigma = 2: pull 2.57 -> 1.77 ms, grad 2.89 -> 1.96; sigma = 6: 43 -> 1.9 ms), order 4 for
 * grid_grad and the grid gradient, its pull behind a probe of the call (smooth fields stay with the LDS tiles).
 * Orders 6 and 7 (round 6, gather7.hip: bricks of 14^3 cells, always under AUTO): 4 x 2 x 256^3 order 7 pull 14.1 -> 7.2 ms, grid_grad
 * 18.0 -> 8.0; sigma = 6: 167 / 436 -> 7.6 / 8.5 ms; their push / count likewise through scatter5's second compilation: 15.5 -> 11.2 ms,
 * sigma = 6: 250 -> 11.7). */
/* 2-D (round 5, scatter2d.hip: gather2d; per-dim orders 1..3, F32 / BF16 / F16 images, float32 coordinates): interpol_pull_ws, the grid
 * gradient of interpol_pull_backward (grad_vol == NULL for a 16-bit image, whose accumulator owns `scratch`) and both gradients of
 * interpol_push_backward_ws go through 32 x 32 bricks of the image -- 16 B per sample + 1 KiB per brick of workspace -- always under
 * INTERPOL_FLAG_BINNED_SCATTER, behind a probe of the call under INTERPOL_FLAG_AUTO_SCATTER (more than 400 pixels per million outside
 * the lean tiles' boxes and a density of at least 0.6 samples per pixel; sparser samplings -- a zoom beyond ~1.3 whose tiles leave their boxes --
 * go to the generic kernel, as the hand-back of rounds 3 - 4 sent them).  Config 5's shape: pull 0.88 ms at every sigma (tiles: 0.43 at sigma = 2, 2.8 at 8, 3.6 at 16). */
/* Trilinear pulls (round 5; 3-D, order 1 in every dim, F32 with dense grids and displacement fields, BF16 / F16 with dense grids, 32768
 * samples and more; the grid gradient of interpol_pull_backward: F32):
 * interpol_pull_workspace returns 256 -- the verdict of the call's probe (mean absolute second difference of the coordinates above one
 * voxel: rough) is all the workspace holds.  AUTO: rough fields take the class-sorted LDS tiles with K = 1 (4 x 2 x 256^3, i.i.d. noise
 * of sigma = 2: 2.5 -> 1.0 ms), smooth ones the generic kernel, which gathers at the HBM roofline there (0.47 ms); BINNED: the tiles. */
int64_t interpol_pull_workspace(const interpol_problem *p);
int interpol_pull_ws(const interpol_problem *p, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes, void *stream);
/* grid_grad (interpol_grad) with the same workspace (interpol_pull_workspace(p) bytes; for the grad problem the same number as for
 * the pull of its image): float32, 3-D quadratic / cubic.  INTERPOL_FLAG_BINNED_SCATTER: the bricks of the image always;
 * INTERPOL_FLAG_AUTO_SCATTER: when the probe of the call finds a dense or rough sampling (4 x 2 x 256^3 cubic, sigma = 2: the
 * natural-order tiles 3.5 ms, the bricks ~2.3), the tile / generic kernels otherwise; else, or without a workspace: interpol_grad.
 * Trilinear (float32, 3-D, dense or displacement grids; the 256-byte workspace of the trilinear pull): AUTO -- rough fields take the
 * LDS tiles (sigma = 2: 2.7 -> 1.8 ms), smooth ones the generic kernel (0.7 ms), on the verdict of the pull's probe; BINNED: the tiles. */
int interpol_grad_ws(const interpol_problem *p, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes, void *stream);
int interpol_push(const interpol_problem *p, const void *val, const void *grid, void *vol,
                  void *scratch, int64_t scratch_bytes, void *stream);
int interpol_count(const interpol_problem *p, const void *grid, void *vol,
                   void *scratch, int64_t scratch_bytes, void *stream);
int interpol_grad(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream);
int interpol_pushgrad(const interpol_problem *p, const void *val, const void *grid, void *vol,
                      void *scratch, int64_t scratch_bytes, void *stream);
int interpol_hess(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream);

/* --- fused backward operators -------------------------------------------------
 * interpol_pull_backward  replaces pushpull.grid_pull_backward (pushpull.py:237-258):
 *      grad_vol  (B,C,*in)    += push(grad_out)                 if grad_vol  != NULL
 *      grad_grid (B,*out,D)    = sum_c grad(vol)[c] * grad_out[c] if grad_grid != NULL
 *   no (B,C,N,D) temporary: with both outputs the call runs the push of grad_out and the channel-contracted grid gradient, two passes
 *   over the grid (round 5: the single fused LDS-tile pass is gone -- it was wrong under rough fields; the generic fused kernel remains
 *   for what the tiles decline).
 *   p->val_stride describes grad_out; grad_vol uses p->vol_stride's layout and is
 *   zero-filled here unless INTERPOL_FLAG_ACCUMULATE; grad_grid is contiguous (B,*out,D).
 *   `scratch`: INTERPOL_BF16 / F16 with grad_vol: the float32 accumulator (4 bytes per element of grad_vol).  INTERPOL_F32,
 *   grad_grid alone, 3-D quadratic / cubic, with INTERPOL_FLAG_AUTO_SCATTER or INTERPOL_FLAG_BINNED_SCATTER: optional workspace of
 *   interpol_pull_workspace(p) bytes (256-byte aligned, contents undefined on entry) for the deformation-independent
 *   organisation of the grid gradient -- the samples sorted by the 16^3 brick of the image their stencil starts in, every brick
 *   staged once in LDS (push_owner.hip: own_gather<K, true>).  AUTO: a probe of the call sends dense samplings there altogether
 *   (4 x 2 x 256^3 cubic: 2.0 against 2.4 ms at sigma = 2, 2.9 against 20 at sigma = 6) and expanding ones to the sample tiles,
 *   which then leave the tiles whose stencils do not fit their LDS box to the bricks; BINNED: the bricks always.  Without it
 *   (NULL, too small, misaligned): the sample tiles alone.
 * interpol_push_backward  replaces pushpull.grid_push_backward (pushpull.py:262-282):
 *      grad_val  (B,C,*in)     = pull(grad_vol_out)              if grad_val  != NULL
 *      grad_grid (B,*in,D)     = sum_c grad(grad_vol_out)[c] * val[c] if grad_grid != NULL
 *   p->vol_stride describes grad_vol_out (the incoming gradient, indexed), p->val_stride
 *   describes val; grad_val / grad_grid are contiguous.
 * interpol_count_backward replaces pushpull.grid_count_backward (pushpull.py:286-299):
 *      grad_grid (B,*in,D)     = sum_c grad(grad_vol_out)[c]
 * --------------------------------------------------------------------------- */
int interpol_pull_backward(const interpol_problem *p, const void *grad_out, const void *vol, const void *grid,
                           void *grad_vol, void *grad_grid, void *scratch, int64_t scratch_bytes, void *stream);
int interpol_push_backward(const interpol_problem *p, const void *grad_vol_out, const void *val, const void *grid,
                           void *grad_val, void *grad_grid, void *stream);
int interpol_count_backward(const interpol_problem *p, const void *grad_vol_out, const void *grid,
                            void *grad_grid, void *stream);
/* interpol_push_backward (val != NULL) / interpol_count_backward (val == NULL, grad_val == NULL) with the bricks workspace of the two
 * gathers they consist of -- interpol_pull_workspace(p) bytes, 256-byte aligned, contents undefined on entry; float32, 3-D quadratic /
 * cubic (2-D: see interpol_pull_workspace), INTERPOL_FLAG_AUTO_SCATTER or INTERPOL_FLAG_BINNED_SCATTER: grad_val is the routed pull of grad_vol_out (interpol_pull_ws),
 * grad_grid the routed grid gradient (as interpol_pull_backward's, the roles of the two images swapped: pushpull.py:276-281).
 * Otherwise, or without a workspace: exactly the two calls above. */
int interpol_push_backward_ws(const interpol_problem *p, const void *grad_vol_out, const void *val, const void *grid,
                              void *grad_val, void *grad_grid, void *workspace, int64_t workspace_bytes, void *stream);

/* --- target-stationary splatting -------------------------------------------------
 * interpol_push_bricks: the same operator as interpol_push (pushpull.py:70-102; with
 * INTERPOL_FLAG_WITH_COUNT also the count image, 106-142), organised for EXPANDING deformations
 * (e.g. 128^3 sources splatted into a shared 512^3 target): the samples are binned by 16^3 target
 * brick (count / scan / fill), every brick is accumulated in LDS by the one workgroup that owns
 * it and stored once, instead of one global atomic per tap.  3-D, INTERPOL_F32 data and grid,
 * channels (+1 with the count) <= 4, batch * samples < 2^32; otherwise INTERPOL_E_DTYPE /
 * INTERPOL_E_DIM / INTERPOL_E_SHAPE and the caller uses interpol_push.  Honours ACCUMULATE,
 * WITH_COUNT, SEPARABLE_GRID, DISPLACEMENT and the shared target (vol batch stride 0).
 * `workspace`: device scratch of interpol_push_bricks_workspace(p) bytes (32 B per sample + 12 B per brick). */
int64_t interpol_push_bricks_workspace(const interpol_problem *p);
int interpol_push_bricks(const interpol_problem *p, const void *val, const void *grid, void *vol,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* --- label maps ------------------------------------------------------------------
 * interpol_pull_labels replaces the per-label loop of api.grid_pull for integer inputs
 * (interpol/api.py:194-205, prefilter=False): vol and val hold int32 LABELS (describe them with
 * dtype = INTERPOL_F32, i.e. 4-byte elements; grid_dtype = INTERPOL_F32); for every sample the
 * label with the largest interpolated indicator value (> 0) under the stencil is returned, the
 * smallest such label on ties, 0 if none -- what the reference's loop over unique() computes,
 * in one pass.  Covered: all dims share one order <= 3 (stencils of up to 64 taps: beyond 27 taps a
 * second kernel visits the distinct labels under the stencil one by one); otherwise INTERPOL_E_ORDER
 * (the caller keeps the loop).  Dense / separable / displacement grids. */
int interpol_pull_labels(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream);

/* --- separable resampling ------------------------------------------------------
 * interpol_resample_1d: one pass of a tensor-product resampling -- what `resize` / `restrict`
 * (interpol/resize.py:13-119, restrict.py:9-121) compute through grid_pull / grid_push on
 * `stack(meshgrid_ij(*lin), -1)`, factorised into D one-dimensional passes (K+1 taps per
 * output instead of (K+1)^D; no grid tensor).  Contiguous (outer, n, inner) arrays.
 *   adjoint == 0 : src (outer, n_lattice, inner), lin[n_samples] -> dst (outer, n_samples, inner)
 *                  dst[b,s,c] = mask(lin[s]) * sum_j w_j sign_j src[b, wrap(i0+j), c]
 *   adjoint == 1 : src (outer, n_samples, inner) -> dst (outer, n_lattice, inner); the exact adjoint of the above (f32 / f64 only).
 *                  Round 5: with n_samples <= 4096 and inner == 1 or inner >= 64 it is a GATHER -- for a non-decreasing `lin` the samples
 *                  whose stencil covers a lattice point are a contiguous range (bisection), the samples that leave the lattice come back
 *                  through the boundary condition and are visited by every output; an unsorted `lin` is detected on the device and served by
 *                  visiting every sample: no atomics, every element of dst written once, bit-reproducible.  Otherwise dst is zero-filled
 *                  here and the taps are added with float atomics.
 * `mode`: 0 nd, 1 iso1, 2 iso0 semantics -- decided by ALL dims of the D-dimensional
 * operator (pushpull.py:48-66), so the caller passes it.  lin is float32 (float64 for F64
 * data).  n_lattice * inner * sizeof(element) must be < 4 GiB, n_samples * inner < 2^32. */
int interpol_resample_1d(int32_t dtype, int32_t lin_dtype, int32_t order, int32_t bound, int32_t extrapolate, int32_t mode,
                         int32_t adjoint, int64_t outer, int64_t n_samples, int64_t n_lattice, int64_t inner,
                         const void *src, const void *lin, void *dst, void *stream);
/* 1 when interpol_resample_1d(adjoint = 1) serves (dtype, n_samples, inner) with the gathering kernel (no atomics, no zero-fill), else 0:
 * the ONE statement of that rule -- the host layer (interpol/separable.py) asks instead of repeating it. */
int32_t interpol_resample_1d_gathers(int32_t dtype, int64_t n_samples, int64_t inner);

/* --- prefilter -----------------------------------------------------------------
 * interpol_spline_filter replaces coeff.spline_coeff (interpol/coeff.py:288-313,
 * filter coeff.py:258-284): in-place interpolating-coefficient IIR along the
 * middle axis of a contiguous (outer, n, inner) array.  order 0/1 = no-op.
 * bound codes as above; dst1/dst2 return INTERPOL_E_PREFILTER (coeff.py:243-244).
 * --------------------------------------------------------------------------- */
int interpol_spline_filter(void *data, int32_t dtype, int64_t outer, int64_t n, int64_t inner,
                           int32_t bound, int32_t order, void *stream);
/* The same filter reading `src` and writing `data` (same layout, no overlap unless src == data): the
 * first filtered dimension of an out-of-place spline_coeff_nd (coeff.py:317-347) then costs one read
 * and one write instead of a copy followed by an in-place pass. */
int interpol_spline_filter_to(const void *src, void *data, int32_t dtype, int64_t outer, int64_t n, int64_t inner,
                              int32_t bound, int32_t order, void *stream);

/* --- host-side helpers (no GPU needed) ------------------------------------------
 * The exact scalar primitives the kernels use, compiled for the host so they
 * can be checked without a device:
 *   interpol_host_bound_index / _sign  = Bound.index / Bound.transform (bounds.py:30-89);
 *                                        sign returns 2 where the reference returns None
 *   interpol_host_weight(order, x, which) which = 0 fastweight, 1 fastgrad, 2 fasthess
 *                                        (splines.py:30-195)
 * --------------------------------------------------------------------------- */
int32_t interpol_host_bound_index(int32_t bound, int32_t i, int32_t n);
int32_t interpol_host_bound_sign(int32_t bound, int32_t i, int32_t n);
double  interpol_host_weight(int32_t order, double x, int32_t which);
float   interpol_host_weight_f32(int32_t order, float x, int32_t which);

/* --- the tile hand-back (csrc/defer.hip): the library's only state -----------------
 * The LDS-tiled kernels hand tiles whose stencils do not fit their LDS box back to the generic kernel of the same
 * operator, launched right behind them on the same stream.  The two kernel families sum in different orders, so
 * WHETHER a launch hands back shows in the last bits of the result:
 *   INTERPOL_HANDBACK_ADAPTIVE (default): a stream hands back only after one of its recent launches met such a
 *       tile (a flag the kernels store into pinned host memory, read without synchronisation at the next launch):
 *       smooth workloads never pay the second kernel, but a result can depend on the history of the stream;
 *   INTERPOL_HANDBACK_ALWAYS / _NEVER: every launch / no launch hands back: each operator is then a deterministic
 *       function of its inputs, as the reference's gather is (nd.py:118-136).
 * Round 5: the hand-back only exists where no device-side router does.  Calls that carry a workspace for the bricks --
 * interpol_push / interpol_count with INTERPOL_FLAG_AUTO_SCATTER or _BINNED_SCATTER, interpol_pull_ws, interpol_grad_ws,
 * interpol_pull_backward / interpol_push_backward_ws with the bricks' workspace (3-D orders 2 - 7 and mixed 1 - 3, 2-D orders 1 - 3, as each
 * entry point documents) -- never hand back: their organisation is chosen by a probe of THIS call's coordinates, so the result is a
 * function of the inputs under every mode (tests: test_routed_operators_do_not_depend_on_the_streams_history).  The modes
 * below still govern 3-D order 1, orders 6 - 7, 2-D grid_grad (a generic kernel) and every call without a workspace.
 * interpol_set_handback(mode) returns the previous mode (process-wide; the environment variable
 * INTERPOL_HANDBACK = adaptive | always | never sets the initial one).
 * interpol_release_stream(stream): the hand-back slot (3 MiB of device memory) of `stream` on the current device goes
 * back to the pool; call it before destroying a stream that ran stretched workloads (optional: slots are recycled
 * least-recently-used).  Returns 1 when the stream held a slot.  Thread-safe, like every entry point: launches of
 * several host threads on one stream are serialised per stream. */
#define INTERPOL_HANDBACK_ADAPTIVE 0
#define INTERPOL_HANDBACK_ALWAYS   1
#define INTERPOL_HANDBACK_NEVER    2
int32_t interpol_set_handback(int32_t mode);
int32_t interpol_release_stream(void *stream);

int32_t     interpol_abi_version(void);
/* 1 for a library built with the measured-slower experimental organisations (torch-interpol_amd/experiments/, `make
 * experiments`: INTERPOL_FLAG_SMALL_TILES and debug bit 4096 select them); 0 for the product library, which ignores both. */
int32_t     interpol_has_experiments(void);
const char *interpol_error_string(int code);
/* name of the kernel family the dispatcher would pick for `p` and op
 * ("pull","push","count","grad","pushgrad","hess"); for tests and profiling. */
const char *interpol_kernel_name(const interpol_problem *p, const char *op);

#ifdef __cplusplus
}
#endif
#endif /* INTERPOL_HIP_H */
