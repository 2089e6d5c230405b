// ===========================================================================
// push_f64.hip -- grid_push / grid_count in float64 on LDS tiles: 3-D, spline orders 0..3 per dim (any mix),
// every boundary / extrapolation mode, every coordinate source of the generic kernels.
//   push, count: reference interpol/nd.py:146-213 (weights splines.py:30-80, indices bounds.py:30-89)
//
// The reference's own tests are float64 (tests/test_gradcheck_pushpull.py:8-18), and the generic float64 scatter is one
// global atomic per tap: under a rough deformation every lane of a wave instruction adds to its own cache line, and the
// memory side retires those at 0.02 G lane-atomics per ms (profiles/r03_micro_global_atomics.txt) -- 12 ms for
// 1 x 2 x 128^3 samples.  Here a workgroup owns a tile of 8^3 samples and accumulates their taps in an LDS box of 64-bit
// FIXED-POINT sums (ds_add_u64: LDS has no fast floating-point add), then adds every touched slot to the target with one
// COALESCED float64 atomic (0.3 G / ms): one channel at a time, scale 2^51 / 2^ceil(log2 max|source|) per tile and
// channel, so that every term is rounded at 2^-52 of the largest source -- finer than the float64 rounding of the terms
// themselves -- and the at most 512 terms of a slot (one per sample of the tile) cannot overflow 63 bits.  Integer sums are
// order-free: a tile's contribution is bit-reproducible; the global atomics of neighbouring tiles add in any order, as in
// the generic kernel.
//   * box: the bounding box of the tile's stencils, at most 24 lattice points per dim (110 KiB, one workgroup per CU);
//     samples whose stencil leaves the (clamped) box, and tiles whose sources are not finite, scatter tap by tap
//     (the generic arithmetic: stencil.hpp);
//   * boundary conditions: wrapped offset and sign per box plane / row / slice, applied at the flush.
// ===========================================================================
#include "../../include/interpol_hip.h"
#include "sorted_util.hpp"

namespace ip {
namespace f64tiles {

constexpr int TE = 8;                           // tile edge (samples)
constexpr int NT = 256;                         // threads per workgroup
constexpr int NS = TE * TE * TE;                // 512 samples
constexpr int VPT = NS / NT;                    // 2
constexpr int CAP = 24;                         // box capacity per dim (lattice points)
constexpr int KMAX = 3;

struct Smem {
    int   taboff[3][CAP];                       // wrapped lattice offset (BYTES) of box plane / row / slice
    int   tabsgn[3][CAP];                       // boundary sign: -1, 0, +1
    int   lo[3], hi[3];
    unsigned long long amax;                    // bits of max |masked source| of the tile and channel (non-negative doubles order like ints)
    int   nonfinite, pad;
    long long box[CAP * CAP * CAP];
};

__device__ __forceinline__ double atomic_add_f64(double *p, double v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool COUNT>
__global__ __launch_bounds__(NT) void push_f64_tiled(KParams p, const double *__restrict__ val, const double *__restrict__ grid,
                                                     double *__restrict__ vol, int gx, int gy, int gz, int nty, int ntz, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    const int tid = (int)threadIdx.x;
    const int work = (int)blockIdx.x;
    const int64_t b = work / ntiles;
    int tile = work % ntiles;
    const int tz = tile % ntz; tile /= ntz;
    const int ox0 = (tile / nty) * TE, oy0 = (tile % nty) * TE, oz0 = tz * TE;

    // ---- the thread's samples: coordinates, first taps, stencil coordinates, weights
    double w[VPT][3][KMAX + 1];
    int i0[VPT][3];
    double msk[VPT];
    int64_t so[VPT];
    bool valid[VPT];
    int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int id = tid + NT * v;
        const int ox = ox0 + (id >> 6), oy = oy0 + ((id >> 3) & 7), oz = oz0 + (id & 7);
        valid[v] = ox < gx && oy < gy && oz < gz;
        so[v] = ((int64_t)(valid[v] ? ox : 0) * gy + (valid[v] ? oy : 0)) * gz + (valid[v] ? oz : 0);
        double x[3];
        load_coords<double, double, 3>(p, grid, b, so[v], x);
        bool inb = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int k = p.order[d];
            if (p.extrapolate != 1) inb = inb && (x[d] > p.mask_lo) && (x[d] < p.mask_hi[d]);          // nd.py:10-27
            const double fl = floor(x[d] - 0.5 * (double)(k - 1));                                       // nd.py:45
            const double t = x[d] - fl;                                                                  // nd.py:46
            const double flc = fl < -1073741824. ? -1073741824. : (fl > 1073741824. ? 1073741824. : fl);
            i0[v][d] = (int)flc;
            if (!(fl == fl)) i0[v][d] = 0x40000000;                                                     // NaN coordinate: out of every box
#pragma unroll
            for (int j = 0; j <= KMAX; ++j) w[v][d][j] = j <= k ? bspline_w<double>(k, t - (double)j) : 0.;   // splines.py:30-80
            if (valid[v] && i0[v][d] != 0x40000000) { mn[d] = min(mn[d], i0[v][d]); mx[d] = max(mx[d], i0[v][d] + k); }
        }
        msk[v] = inb ? 1. : 0.;
    }
    if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
    for (int e = tid; e < CAP * CAP * CAP; e += NT) sm.box[e] = 0ll;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = tiled::wave_min(mn[d]), e = tiled::wave_max(mx[d]);
        if ((tid & 63) == 0) { atomicMin(&sm.lo[d], a); atomicMax(&sm.hi[d], e); }
    }
    __syncthreads();
    int lo[3], S[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int l = sm.lo[d], h = sm.hi[d];
        if (h < l) { l = 0; h = 0; }
        // (64-bit: first taps clamped to both -2^30 and +2^30 in one tile -- infinite or huge coordinates -- would overflow int)
        long long sz = (long long)h - (long long)l + 1;
        if (sz > CAP) { l = (int)((long long)l + (sz - CAP) / 2); sz = CAP; }     // keep the centre; the rest scatters directly
        lo[d] = l; S[d] = (int)sz;
    }
    if (tid < 3 * CAP) {
        const int d = tid / CAP, slot = tid - d * CAP;
        const int Sd = d == 0 ? S[0] : d == 1 ? S[1] : S[2];
        if (slot < Sd) {
            const int ld = d == 0 ? lo[0] : d == 1 ? lo[1] : lo[2];
            const long long pk = wrap_outofline(p.bound[d], ld + slot, p.vol_n[d]);
            sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * p.vol_ss[d];
            sm.tabsgn[d][slot] = (int)(pk >> 32);
        }
    }
    // in the box <=> lo <= i0 and i0 + k <= lo + S - 1 in every dim
    bool inbox[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        inbox[v] = valid[v];
#pragma unroll
        for (int d = 0; d < 3; ++d) inbox[v] = inbox[v] && i0[v][d] >= lo[d] && (long long)i0[v][d] + p.order[d] <= (long long)lo[d] + S[d] - 1;
    }

    for (int c = 0; c < p.C; ++c) {
        double *vc = vol + b * p.vol_sb + (int64_t)c * p.vol_sc;
        double src[VPT];
        unsigned long long am = 0ull;
        bool fin = true;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            src[v] = msk[v];
            if (!COUNT && valid[v]) src[v] = val[b * p.val_sb + (int64_t)c * p.val_sc + so[v]] * msk[v];
            if (!valid[v]) src[v] = 0.;
            const double a = fabs(src[v]);
            if (!(a <= 1.7e308)) fin = false;                         // inf / NaN
            else am = max(am, (unsigned long long)__double_as_longlong(a));
        }
        __syncthreads();                                              // tables written (first channel) / the previous channel's flush is done
        if (tid == 0) { sm.amax = 0ull; sm.nonfinite = 0; }
        __syncthreads();
        if (!fin) sm.nonfinite = 1;
        atomicMax(&sm.amax, am);
        __syncthreads();
        const double amaxd = __longlong_as_double((long long)sm.amax);
        // scale = 2^(51 - e) with 2^e > max |source| (weights <= 1: every term below 2^51)
        int ex = 0;
        (void)frexp(amaxd, &ex);
        // (block-uniform) no fixed-point scale -- non-finite sources, or sources so small that 2^(51 - e) is not a double
        // (max |source| < 2^-970): the generic arithmetic
        const bool direct_all = sm.nonfinite != 0 || ex < -960;
        const double scale = ldexp(1., 51 - ex), inv = ldexp(1., ex - 51);
        if (amaxd > 0. || direct_all) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!valid[v] || src[v] == 0.) continue;
                if (inbox[v] && !direct_all) {
                    const int bx = i0[v][0] - lo[0], by = i0[v][1] - lo[1], bz = i0[v][2] - lo[2];
                    for (int i = 0; i <= p.order[0]; ++i) {
                        const double si = src[v] * w[v][0][i] * scale;
                        for (int j = 0; j <= p.order[1]; ++j) {
                            const double sj = si * w[v][1][j];
                            long long *row = sm.box + ((bx + i) * S[1] + (by + j)) * S[2] + bz;
                            for (int k = 0; k <= p.order[2]; ++k) {
                                const long long q = __double2ll_rn(sj * w[v][2][k]);
                                atomicAdd(reinterpret_cast<unsigned long long *>(row + k), (unsigned long long)q);
                            }
                        }
                    }
                } else {
                    // tap by tap, like the generic kernel (bounds.py:30-89 per tap)
                    for (int i = 0; i <= p.order[0]; ++i) {
                        const long long p0 = wrap_outofline(p.bound[0], i0[v][0] == 0x40000000 ? 0 : i0[v][0] + i, p.vol_n[0]);
                        const double si = src[v] * w[v][0][i] * (double)(int)(p0 >> 32);
                        for (int j = 0; j <= p.order[1]; ++j) {
                            const long long p1 = wrap_outofline(p.bound[1], i0[v][1] == 0x40000000 ? 0 : i0[v][1] + j, p.vol_n[1]);
                            const double sj = si * w[v][1][j] * (double)(int)(p1 >> 32);
                            for (int k = 0; k <= p.order[2]; ++k) {
                                const long long p2 = wrap_outofline(p.bound[2], i0[v][2] == 0x40000000 ? 0 : i0[v][2] + k, p.vol_n[2]);
                                const double t = sj * w[v][2][k] * (double)(int)(p2 >> 32);
                                const unsigned off = (unsigned)(int)(p0 & 0xffffffffll) * (unsigned)p.vol_ss[0] + (unsigned)(int)(p1 & 0xffffffffll) * (unsigned)p.vol_ss[1]
                                                   + (unsigned)(int)(p2 & 0xffffffffll) * (unsigned)p.vol_ss[2];
                                atomic_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(vc) + off), t);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- flush: every touched slot -> one float64 atomic, consecutive threads along the unit-stride dim; the slot is cleared
        if (!direct_all && amaxd > 0.) {
            const int nslot = S[0] * S[1] * S[2];
            const float rS2 = 1.f / (float)S[2], rS1 = 1.f / (float)S[1];
            for (int e = tid; e < nslot; e += NT) {
                const long long q = sm.box[e];
                if (q == 0ll) continue;
                sm.box[e] = 0ll;
                const int r = (int)(((float)e + 0.5f) * rS2), z = e - r * S[2];          // e = (x * S1 + y) * S2 + z   (e < 13 824: exact)
                const int x = (int)(((float)r + 0.5f) * rS1), y = r - x * S[1];
                const int sg = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sm.tabsgn[2][z];
                if (sg == 0) continue;
                const unsigned off = (unsigned)(sm.taboff[0][x] + sm.taboff[1][y] + sm.taboff[2][z]);
                atomic_add_f64(reinterpret_cast<double *>(reinterpret_cast<char *>(vc) + off), (double)q * inv * (double)sg);
            }
        }
    }
}

} // namespace f64tiles

// `acc`: the zero-filled (or accumulating) float64 target; val == NULL: count.  1: took the problem, 0: declined.
int try_push_f64_tiles(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *acc, hipStream_t st)
{
    using namespace f64tiles;
    if (p->dim != 3 || p->dtype != INTERPOL_F64 || p->grid_dtype != INTERPOL_F64) return 0;
    if (k.mode == MODE_ISO0) return 0;                               // all-nearest: its own rounding rule (iso0.py:12), the generic kernel
    for (int d = 0; d < 3; ++d) if (k.order[d] > KMAX) return 0;
    if (k.gate) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    if ((int64_t)gx * gy * gz < 4096) return 0;                      // small problems: launch-bound either way
    const int ntx = (gx + TE - 1) / TE, nty = (gy + TE - 1) / TE, ntz = (gz + TE - 1) / TE;
    const int64_t total = (int64_t)ntx * nty * ntz * p->batch;
    if (total > 0x7fffffff) return 0;
    const int ntiles = ntx * nty * ntz;
    int rc;
    if (val) {
        rc = sorted::big_lds<push_f64_tiled<false>>(sizeof(Smem));
        if (rc) return rc;
        hipLaunchKernelGGL((push_f64_tiled<false>), dim3((unsigned)total), dim3(NT), sizeof(Smem), st, k, (const double *)val, (const double *)grid,
                           (double *)acc, gx, gy, gz, nty, ntz, ntiles);
    } else {
        rc = sorted::big_lds<push_f64_tiled<true>>(sizeof(Smem));
        if (rc) return rc;
        hipLaunchKernelGGL((push_f64_tiled<true>), dim3((unsigned)total), dim3(NT), sizeof(Smem), st, k, (const double *)nullptr, (const double *)grid,
                           (double *)acc, gx, gy, gz, nty, ntz, ntiles);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

} // namespace ip
