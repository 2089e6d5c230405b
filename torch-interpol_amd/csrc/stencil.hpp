// ===========================================================================
// stencil.hpp -- per-sample B-spline stencil held in registers.
//
// One thread owns one sample point (one row of the coordinate grid).  It
// computes, per spatial dim, the K+1 tap weights (and derivatives when needed),
// the wrapped lattice offsets and the boundary signs -- everything the
// reference materialises as (B,N) tensors in nd.get_weights
// (reference interpol/nd.py:30-77; iso1.py:10-20; iso0.py:10-15) -- and keeps
// them in VGPRs.  The (K+1)^D tap loop is fully unrolled over these arrays.
// ===========================================================================
#pragma once
#include "spline_math.hpp"
#include <stdint.h>

namespace ip {

// ---- storage <-> math conversions ----------------------------------------
struct bf16_t { unsigned short u; };
typedef _Float16 f16_t;

template <typename R, typename T> struct Cvt;
template <> struct Cvt<float, float>   { static IP_HD float ld(float v) { return v; }   static IP_HD float st(float v) { return v; } };
template <> struct Cvt<double, double> { static IP_HD double ld(double v) { return v; } static IP_HD double st(double v) { return v; } };
template <> struct Cvt<float, f16_t>   { static IP_HD float ld(f16_t v) { return (float)v; } static IP_HD f16_t st(float v) { return (f16_t)v; } };
template <> struct Cvt<float, bf16_t> {
    static IP_HD float ld(bf16_t v) { return __uint_as_float(((unsigned)v.u) << 16); }
    static IP_HD bf16_t st(float f) {            // round to nearest even, NaN kept quiet
        bf16_t r;
#if defined(__HIP_DEVICE_COMPILE__)
        const __bf16 h = (__bf16)f;              // v_cvt_pk_bf16_f32 on gfx950
        __builtin_memcpy(&r.u, &h, 2);
        return r;
#endif
        unsigned x = __float_as_uint(f);
        if ((x & 0x7fffffffu) > 0x7f800000u) { r.u = (unsigned short)((x >> 16) | 0x40); return r; }
        x += 0x7fffu + ((x >> 16) & 1u);
        r.u = (unsigned short)(x >> 16);
        return r;
    }
};

// ---- kernel-side problem description (device friendly copy of interpol_problem)
struct KParams {
    int dim;
    int extrapolate;
    int mode;               // 0 = nd, 1 = iso1 (all orders 1), 2 = iso0 (all orders 0)
    int bound[3];
    int order[3];
    int vol_n[3];
    int vol_ss[3];          // spatial strides of vol in BYTES (whole image < 2^32 bytes, checked on host)
    int C;
    int dbg;                // debug / ablation switches (interpol_problem.flags >> 8), 0 in production
    int cc;                 // push: 1 = the target has one more channel than val, which receives the count (INTERPOL_FLAG_WITH_COUNT)
    int sep;                // 1: INTERPOL_FLAG_SEPARABLE_GRID (grid = D coordinate vectors back to back); 2: INTERPOL_FLAG_DISPLACEMENT
    int gshape[3];          // sample-grid extents (problem dims), used to split a linear sample index when sep
    int64_t N;              // samples per batch item
    int64_t vol_sb, vol_sc;
    int64_t grid_sb;        // grid: spatial dims contiguous, component stride 1
    int64_t val_sb, val_sc; // val : spatial (+ trailing d,e) dims contiguous
    double mask_lo;         // -threshold
    double mask_hi[3];      // n-1+threshold
    float mask_lo_f, mask_hi_f[3];   // the same as floats (float kernels: scalar loads instead of a conversion hoisted into -- spilled -- VGPRs)
    int gate_n;             // interpol_pull_ws: ints in front of `gate` (header, brick counters, brick list) that pull_sorted zeroes for own_bin
    const int *verdict;     // interpol_pull_ws (round 5): the word the probe of THIS call writes (push_owner.hip: own_probe) -- 1: the bricks of the
                            // image take every tile, pull_sorted returns at once
    const int *gate;        // scatters under INTERPOL_FLAG_AUTO_SCATTER: a device word written by the roughness probe of this call
                            // (push_owner.hip); the tiled / generic scatter kernels return at once when it is non-zero
};

// Tiles that an LDS-tiled kernel hands back to the generic kernels (defer.hip): per (tile, batch item) work item of the
// launch a 64-bit descriptor -- batch item (20 bits), tile coordinates x, y, z (14 bits each) -- and the number of the
// launch that wrote it: an entry counts only if its stamp is the current launch's, so the tile kernels write nothing
// for the tiles they serve and nothing is ever reset.  desc == NULL: every sample (the plain generic launch).
struct TileList { const unsigned long long *desc; const unsigned *gen; unsigned cur; int nwork; int e[3]; };
// (flag: where a tile kernel reports -- by storing the launch number -- that it met a tile worth handing back, whether or not
//  this launch hands back: desc == NULL while the stream's recent launches met none, see defer.hip)
struct DeferArgs { unsigned long long *desc; unsigned *gen; unsigned *flag; unsigned cur; };
__host__ __device__ inline unsigned long long tile_desc(int64_t b, int cx, int cy, int cz)
{
    return ((unsigned long long)b << 42) | ((unsigned long long)cx << 28) | ((unsigned long long)cy << 14) | (unsigned long long)cz;
}
// one thread of the block, for a tile worth handing back; true: handed back (skip it), false: this launch serves everything itself
__device__ inline void defer_mark(const DeferArgs &d, int work, unsigned long long desc)
{
    *d.flag = d.cur;
    if (d.desc) { d.desc[work] = desc; d.gen[work] = d.cur; }
}

enum { MODE_ND = 0, MODE_ISO1 = 1, MODE_ISO0 = 2 };
enum { NEED_W = 0, NEED_G = 1, NEED_H = 2 };

// Number of taps the unrolled loops run along dim d.
template <int D, int KMAX> struct Taps {
    static constexpr int T0 = KMAX + 1;
    static constexpr int T1 = D > 1 ? KMAX + 1 : 1;
    static constexpr int T2 = D > 2 ? KMAX + 1 : 1;
};

// Border samples only: Bound.index and Bound.transform of one tap, kept out of
// line so that the (rarely executed) boundary switch exists once per kernel
// instead of once per tap.  Low word = wrapped index, high word = sign factor.
static __device__ __noinline__ long long wrap_outofline(int bound, int i, int n)
{
    const int idx = wrap_index(bound, i, n);                                       // bounds.py:30-60
    const int sgn = wrap_sign(bound, i, n);                                        // bounds.py:62-89
    return ((long long)sgn << 32) | (unsigned)idx;
}

// ---------------------------------------------------------------------------
// Stencil<R, D, KMAX, ISO, NEED>
//   ISO  : every dim has order == KMAX (compile-time order, no tap predicates)
//   !ISO : per-dim runtime orders <= KMAX; taps j > order[d] are predicated off
// ---------------------------------------------------------------------------
template <typename R, int D, int KMAX, bool ISO, int NEED>
struct Stencil {
    static constexpr int T = KMAX + 1;
    R   w[3][T];
    R   g[3][T];
    R   h[3][T];
    unsigned off[3][T];     // BYTE offsets (idx * stride * sizeof(element)), < 2^32 checked on host
    R   mask;               // 1 or 0 (extrapolate 0/2), always 1 for extrapolate == 1

    // x: the D coordinates of this sample
    __device__ __forceinline__ void setup(const KParams &p, const R *x)
    {
        bool inb = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d >= D) {
#pragma unroll
                for (int j = 0; j < T; ++j) { w[d][j] = R(1); g[d][j] = R(0); h[d][j] = R(0); off[d][j] = 0; }
                continue;
            }
            const int k  = ISO ? KMAX : p.order[d];
            const int n  = p.vol_n[d];
            const int bd = p.bound[d];
            const int ss = p.vol_ss[d];
            const R xd = x[d];
            if (p.extrapolate != 1)          // nd.py:10-27 / jit_utils.py:241-285
                inb = inb && (xd > (R)p.mask_lo) && (xd < (R)p.mask_hi[d]);
            R fl;
            if (KMAX == 0 && p.mode == MODE_ISO0) fl = rint_(xd);                 // iso0.py:12 round half to even
            else                                  fl = floor_(xd - R(0.5) * R(k - 1));  // nd.py:45 ; iso1.py:13 when k == 1
            const R t = xd - fl;                                                  // nd.py:46
            // clamp so that i0 + j cannot overflow; such coordinates are far outside any lattice
            const R flc = fl < R(-1073741824) ? R(-1073741824) : (fl > R(1073741824) ? R(1073741824) : fl);
            const int i0 = (int)flc;
            // ---- lattice indices & boundary signs --------------------------------
            // Fast path (one divergent test per dim, not per tap): when the whole
            // support [i0, i0+k] lies inside the lattice no wrapping is needed and
            // every sign is +1 (dst1 also needs i0 >= 1: its sign is 0 at index 0,
            // reference quirk B-3).  Only border samples take the out-of-line path.
            int idx[T], sg[T];
#pragma unroll
            for (int j = 0; j < T; ++j) { idx[j] = i0 + j; sg[j] = 1; }
            const bool inside = (i0 >= (bd == B_DST1 ? 1 : 0)) && (i0 + k < n);
            if (!inside) {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    if (ISO || j <= k) {
                        const long long pk = wrap_outofline(bd, i0 + j, n);
                        idx[j] = (int)(pk & 0xffffffffll);
                        sg[j]  = (int)(pk >> 32);
                    }
                }
            }
            // ---- weights ------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < T; ++j) {
                if (!ISO && j > k) { w[d][j] = R(0); g[d][j] = R(0); h[d][j] = R(0); off[d][j] = 0; continue; }
                off[d][j] = (unsigned)idx[j] * (unsigned)ss;
                const R tj = t - R(j);                                             // nd.py:56
                R wj, gj = R(0), hj = R(0);
                if (k == 1 && p.mode == MODE_ISO1) {                               // iso1.py:19-20, 311-313
                    wj = (j == 0) ? R(1) - t : t;
                    gj = (j == 0) ? R(-1) : R(1);
                } else {
                    wj = bspline_w<R>(k, tj);                                      // splines.py:30
                    if (NEED >= NEED_G) gj = bspline_g<R>(k, tj);                  // splines.py:90
                    if (NEED >= NEED_H) hj = bspline_h<R>(k, tj);                  // splines.py:149
                }
                // sign folded into the factors: (v*s)*w == v*(s*w) exactly for s in {-1,0,1}
                const R sf = R(sg[j]);
                w[d][j] = wj * sf;
                if (NEED >= NEED_G) g[d][j] = gj * sf;
                if (NEED >= NEED_H) h[d][j] = hj * sf;
            }
        }
        mask = inb ? R(1) : R(0);
    }

    static __device__ __forceinline__ float  floor_(float v)  { return floorf(v); }
    static __device__ __forceinline__ double floor_(double v) { return floor(v); }
    static __device__ __forceinline__ float  rint_(float v)   { return rintf(v); }
    static __device__ __forceinline__ double rint_(double v)  { return rint(v); }

    // is tap j of dim d part of the stencil?  (wave-uniform)
    __device__ __forceinline__ bool on(const KParams &p, int d, int j) const
    {
        return ISO || d >= D || j <= p.order[d];
    }
};

__device__ __forceinline__ float  coord_fma(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double coord_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Coordinates of sample `o` of batch item `b`.
template <typename R, typename G, int D>
__device__ __forceinline__ void load_coords(const KParams &p, const G *grid, int64_t b, int64_t o, R *x)
{
    if (p.sep) {
        unsigned r = (unsigned)o;
        int base = 0;
        unsigned od[3] = { 0u, 0u, 0u };
#pragma unroll
        for (int d = D - 1; d > 0; --d) { const unsigned q = r / (unsigned)p.gshape[d]; od[d] = r - q * (unsigned)p.gshape[d]; r = q; }
        od[0] = r;
        if (p.sep == 1) {
            // tensor-product coordinates (resize.py:96-123): x_d = lin_d[o_d]
#pragma unroll
            for (int d = 0; d < D; ++d) { x[d] = (R)grid[base + (int)od[d]]; base += p.gshape[d]; }
        } else if (p.sep == 3) {
            // affine lattice (affine_grid, api.py:534-572): x = A o + t, the D x (D+1) matrix [A | t] behind `grid`
#pragma unroll
            for (int d = 0; d < D; ++d) {
                G s = grid[d * (D + 1)] * (G)od[0];
#pragma unroll
                for (int e = 1; e < D; ++e) s = coord_fma(grid[d * (D + 1) + e], (G)od[e], s);
                x[d] = (R)(G)(s + grid[d * (D + 1) + D]);
            }
        } else {
            // displacement field: x_d = o_d + disp (add_identity_grid_, api.py:490-513, in the grid's dtype)
            const G *gp = grid + b * p.grid_sb + o * D;
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = (R)(G)(gp[d] + (G)od[d]);
        }
        return;
    }
    const G *gp = grid + b * p.grid_sb + o * D;
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = (R)gp[d];
}

} // namespace ip
