// ===========================================================================
// tile_common.hpp -- device helpers shared by the LDS-tile kernel families
// (ops_tiled.hip: natural-order tiles; ops_sorted.hip: class-sorted tiles):
// wave reductions, the floor / fraction split of a coordinate, per-dim weights,
// the rare out-of-line per-thread / tap-parallel paths through global memory, and
// the XCD-aware persistent work range.
// Numerical definition: reference interpol/nd.py:44-61 (stencil), splines.py:30-139
// (weights), bounds.py:30-89 (index wrapping), iso1.py:13-20 (all-linear weights).
// ===========================================================================
#pragma once
#include "stencil.hpp"
#include "defer.hpp"

namespace ip {
namespace tiled {

// Wave-wide min / max in registers: four row_shr steps inside the 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows (GFX9 DPP), result read from lane 63.  min and max
// are idempotent, so lanes that have no DPP source simply combine with themselves.  (The
// __shfl_xor butterfly costs six dependent ds_bpermute round trips per value.)
// (`old` = the identity of the operation: lanes without a DPP source keep their value, and
// the compiler can fuse the move into v_min_i32_dpp / v_max_i32_dpp.)
#define IP_DPP(id, v, ctrl) __builtin_amdgcn_update_dpp((int)(id), (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ int wave_min(int v)
{
    int t;
    t = IP_DPP(0x7fffffff, v, 0x111); v = t < v ? t : v;
    t = IP_DPP(0x7fffffff, v, 0x112); v = t < v ? t : v;
    t = IP_DPP(0x7fffffff, v, 0x114); v = t < v ? t : v;
    t = IP_DPP(0x7fffffff, v, 0x118); v = t < v ? t : v;
    t = IP_DPP(0x7fffffff, v, 0x142); v = t < v ? t : v;
    t = IP_DPP(0x7fffffff, v, 0x143); v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v)
{
    int t;
    t = IP_DPP(0x80000000, v, 0x111); v = t > v ? t : v;
    t = IP_DPP(0x80000000, v, 0x112); v = t > v ? t : v;
    t = IP_DPP(0x80000000, v, 0x114); v = t > v ? t : v;
    t = IP_DPP(0x80000000, v, 0x118); v = t > v ? t : v;
    t = IP_DPP(0x80000000, v, 0x142); v = t > v ? t : v;
    t = IP_DPP(0x80000000, v, 0x143); v = t > v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
#undef IP_DPP
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// floor index and stencil coordinate t of one coordinate (nd.py:45-47 / iso1.py:13-20)
__device__ __forceinline__ void split(int k, float x, int &i0, float &t)
{
    const float fl = floorf(x - 0.5f * (float)(k - 1));
    t = x - fl;
    const float flc = fl < -1073741824.f ? -1073741824.f : (fl > 1073741824.f ? 1073741824.f : fl);
    i0 = (int)flc;
}

// The scalars of KParams the kernels need per kernel dim (x, y, z), by value, so that
// out-of-line rare-path functions do not drag the whole KParams along.
struct Lattice {
    int bound[3], n[3], ss[3];     // boundary codes, extents, strides in elements
    int k[3];                      // spline order per kernel dim (0 for the degenerate x of 2-D)
    int lin;                       // iso1 weights (1-t, t) and gradients (-1, +1)
};

// k: the dim's order (a compile-time constant after inlining when the tile is ISO)
__device__ __forceinline__ float weight1(int lin, int k, float t, int j, int piece = -1)
{
    return lin ? (j == 0 ? 1.f - t : t) : bspline_w<float>(k, t - (float)j, piece);      // iso1.py:19-20 / splines.py:30-80
}
__device__ __forceinline__ float wgrad1(int lin, int k, float t, int j, int piece = -1)
{
    return lin ? (j == 0 ? -1.f : 1.f) : bspline_g<float>(k, t - (float)j, piece);       // iso1.py:311-313 / splines.py:90-139
}
// The polynomial piece of tap j of an order-K stencil: `split` puts the stencil coordinate t in [(K-1)/2, (K+1)/2), so
// |t - j| lies in ONE interval between the spline's breakpoints (at an interval's end the neighbouring pieces agree).
__host__ __device__ constexpr int tap_piece(int K, int j)
{
    return (K & 1) ? (j <= (K - 1) / 2 ? (K - 1) / 2 - j : j - (K - 1) / 2 - 1) : (j < K / 2 ? K / 2 - j : j - K / 2);
}
template <int KMAX>
__device__ __forceinline__ void weights(int lin, int k, float t, float *w)
{
    // (k == KMAX -- the isotropic tiles, a compile-time fact after inlining --: one polynomial piece per weight instead
    //  of all of them plus selects: 25 -> 9 VALU instructions per quintic weight)
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) w[j] = j <= k ? weight1(lin, k, t, j, k == KMAX ? tap_piece(KMAX, j) : -1) : 0.f;
}
template <int KMAX>
__device__ __forceinline__ void wgrads(int lin, int k, float t, float *g)
{
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) g[j] = j <= k ? wgrad1(lin, k, t, j, k == KMAX ? tap_piece(KMAX, j) : -1) : 0.f;
}

// ---------------------------------------------------------------------------
// Rare paths, out of line.
// ---------------------------------------------------------------------------
// One sample gathered tap by tap from global memory by ONE thread (rolled loops).
// which = -1: value, 0..2: derivative along kernel dim `which`.
template <typename T>
__device__ __noinline__ float gather_one_thread(Lattice L, const T *vc, int ix, int iy, int iz,
                                                float tx, float ty, float tz, int which)
{
    float acc = 0.f;
    for (int i = 0; i <= L.k[0]; ++i) {
        const long long pk0 = wrap_outofline(L.bound[0], ix + i, L.n[0]);
        const float fx = L.n[0] == 1 && L.ss[0] == 0 && L.k[0] == 0 ? 1.f
                       : (which == 0 ? wgrad1(L.lin, L.k[0], tx, i) : weight1(L.lin, L.k[0], tx, i));
        const float sx = fx * (float)(int)(pk0 >> 32);
        const int offx = (int)(pk0 & 0xffffffffll) * L.ss[0];
        float pl = 0.f;
        for (int j = 0; j <= L.k[1]; ++j) {
            const long long pk1 = wrap_outofline(L.bound[1], iy + j, L.n[1]);
            const float fy = which == 1 ? wgrad1(L.lin, L.k[1], ty, j) : weight1(L.lin, L.k[1], ty, j);
            const float sy = fy * (float)(int)(pk1 >> 32);
            const int offy = (int)(pk1 & 0xffffffffll) * L.ss[1];
            float r = 0.f;
            for (int k = 0; k <= L.k[2]; ++k) {
                const long long pk2 = wrap_outofline(L.bound[2], iz + k, L.n[2]);
                const float fz = which == 2 ? wgrad1(L.lin, L.k[2], tz, k) : weight1(L.lin, L.k[2], tz, k);
                r = __builtin_fmaf(fz * (float)(int)(pk2 >> 32),
                                   Cvt<float, T>::ld(vc[offx + offy + (int)(pk2 & 0xffffffffll) * L.ss[2]]), r);
            }
            pl = __builtin_fmaf(sy, r, pl);
        }
        acc = __builtin_fmaf(sx, pl, acc);
    }
    return acc;
}

// One sample scattered tap by tap to the (float) target by ONE thread (rolled loops).
__device__ __noinline__ void scatter_one_thread(Lattice L, float *vc, float src, int ix, int iy, int iz,
                                                float tx, float ty, float tz)
{
    for (int i = 0; i <= L.k[0]; ++i) {
        const long long pk0 = wrap_outofline(L.bound[0], ix + i, L.n[0]);
        const float sx = src * weight1(L.lin, L.k[0], tx, i) * (float)(int)(pk0 >> 32);
        const int offx = (int)(pk0 & 0xffffffffll) * L.ss[0];
        for (int j = 0; j <= L.k[1]; ++j) {
            const long long pk1 = wrap_outofline(L.bound[1], iy + j, L.n[1]);
            const float sy = sx * weight1(L.lin, L.k[1], ty, j) * (float)(int)(pk1 >> 32);
            const int offy = (int)(pk1 & 0xffffffffll) * L.ss[1];
            for (int k = 0; k <= L.k[2]; ++k) {
                const long long pk2 = wrap_outofline(L.bound[2], iz + k, L.n[2]);
                const float v = sy * weight1(L.lin, L.k[2], tz, k) * (float)(int)(pk2 >> 32);
                __hip_atomic_fetch_add(vc + offx + offy + (int)(pk2 & 0xffffffffll) * L.ss[2], v,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Tap-parallel: tap number `tap` (row-major over the per-dim taps) of the sample at
// coordinates (gx_, gy_, gz_): returns weight * sign (0 beyond the last tap), the lattice
// offset in *off_out and, when grads != nullptr, the three derivative weights (times the
// value weights of the other dims).  The degenerate x of a 2-D problem has order 0 and
// weight bspline_w(0, .) = 1.
// KX, KC: compile-time orders along x and along y / z for ISO tiles (folds the tap decode and
// the spline switch), -1 = runtime orders of L.
template <int KX, int KC>
__device__ __noinline__ float tap_weight_t(Lattice L, float gx_, float gy_, float gz_, int tap, int *off_out, float *grads)
{
    const int kk[3] = { KX >= 0 ? KX : L.k[0], KC >= 0 ? KC : L.k[1], KC >= 0 ? KC : L.k[2] };
    const int k1[3] = { kk[0] + 1, kk[1] + 1, kk[2] + 1 };
    const bool on = tap < k1[0] * k1[1] * k1[2];
    const int tp[3] = { on ? tap / (k1[1] * k1[2]) : 0, on ? (tap / k1[2]) % k1[1] : 0, on ? tap % k1[2] : 0 };
    const float g[3] = { gx_, gy_, gz_ };
    float w[3], dw[3];
    int off = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int i0; float t;
        split(kk[d], g[d], i0, t);
        if (L.n[d] == 1 && L.ss[d] == 0 && kk[d] == 0) i0 = 0;       // degenerate dim: coordinate is a dummy
        const int idx = i0 + tp[d];
        int widx = idx; float s = 1.f;
        if ((unsigned)idx >= (unsigned)L.n[d] || (L.bound[d] == B_DST1 && idx == 0)) {   // border taps only: Bound.index / transform (dst1: sign 0 at index 0, quirk B-3)
            const long long pk = wrap_outofline(L.bound[d], idx, L.n[d]);
            widx = (int)(pk & 0xffffffffll);
            s = (float)(int)(pk >> 32);
        }
        w[d] = weight1(L.lin, kk[d], t, tp[d]) * s;
        dw[d] = grads ? wgrad1(L.lin, kk[d], t, tp[d]) * s : 0.f;
        off += widx * L.ss[d];
    }
    *off_out = off;
    const float m = on ? 1.f : 0.f;
    if (grads) {
        grads[0] = m * dw[0] * w[1] * w[2];
        grads[1] = m * w[0] * dw[1] * w[2];
        grads[2] = m * w[0] * w[1] * dw[2];
    }
    return m * w[0] * w[1] * w[2];
}

// ---------------------------------------------------------------------------
// Persistent blocks: which tiles does this block process?  Workgroups are dealt round-robin
// to the 8 XCDs, each with its own L2.  Each XCD therefore takes one CONTIGUOUS eighth of the
// tile sequence (z-fastest tile order) and its blocks walk it side by side: the tiles in flight
// on one XCD are neighbours and share their halos in that XCD's L2, instead of every XCD
// fetching every halo.  (Measured on cfg2: the gathers gain on smooth deformations, 1.57 ->
// 1.49 ms; the scatters LOSE, 2.31 -> 2.51 ms -- neighbouring tiles then flush their halos
// into the same L2 lines at the same time -- so the scatter kernels keep the strided order.)
// ---------------------------------------------------------------------------
// Part of the hand-back decision (defer.hip), evaluated only for tiles with many samples outside the box: is the
// deformation SMOOTH over the tile (a zoom, a large but regular displacement)?  Then neighbouring samples share their
// lattice neighbourhood and the generic kernel serves the tile well.  Under rough (i.i.d.) coordinates it would thrash
// the caches and the tile keeps its samples: the in-box majority still runs from LDS.  Measure: mean absolute second
// difference of the coordinates along the last dim, summed over the dims, below one voxel.  Separable / affine lattices
// and displacement fields count as smooth.  Block-uniform; `acc`: two free ints of LDS.
__device__ __forceinline__ bool tile_smooth(const KParams &p, const float *__restrict__ grid, int64_t b, int D, int ox0, int oy0, int oz0,
                                            int ex, int ey, int ez, int gx, int gy, int gz, int *acc)
{
    if (p.sep) return true;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    float s = 0.f; int n = 0;
    for (int id = threadIdx.x; id < ex * ey * ez; id += 2 * (int)blockDim.x) {      // a sample of the tile is plenty
        int r = id;
        const int dz = r % ez; r /= ez;
        const int dy = r % ey, dx = r / ey;
        const int ox = ox0 + dx, oy = oy0 + dy, oz = oz0 + dz;
        if (ox >= gx || oy >= gy || oz + 2 >= gz || dz + 2 >= ez) continue;
        const float *q = grid + b * p.grid_sb + (((int64_t)ox * gy + oy) * gz + oz) * D;
        for (int d = 0; d < D; ++d) s += __builtin_fabsf(q[d] - 2.f * q[D + d] + q[2 * D + d]);
        ++n;
    }
    s = s < 1e4f ? s : 1e4f;                           // (also catches NaN)
    if (n) { atomicAdd(&acc[0], (int)(s * 16.f)); atomicAdd(&acc[1], n); }
    __syncthreads();
    const bool smooth = acc[0] < acc[1] * 16 || acc[1] == 0;
    __syncthreads();
    return smooth;
}

struct WorkRange {
    int first, end, step;
    __device__ __forceinline__ explicit WorkRange(int total, bool by_xcd = true)
    {
        const int G = (int)gridDim.x, bid = (int)blockIdx.x;
        if (by_xcd && (G & 7) == 0 && total >= 8) {
            const int per = (total + 7) >> 3, xcd = bid & 7;
            first = xcd * per + (bid >> 3);
            end = (xcd + 1) * per < total ? (xcd + 1) * per : total;
            step = G >> 3;
        } else {
            first = bid; end = total; step = G;
        }
    }
};

// floor(x + 0.5) in one instruction (v_cvt_i32_f32 truncates; rndne + cvt would be two)
__device__ __forceinline__ int cvt_rpi(float x)
{
    int q;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(x));
    return q;
}

// Headroom bits of the 32-bit fixed-point scatter (see scatter_channel), or -1 when 32 bits
// would not be precise enough for this tile (large sample density / many taps).
__device__ __forceinline__ int headroom32(const Lattice &L, int dmax)
{
    const float wsum[8] = { 1.f, 2.f, 1.75f, 1.6666667f, 1.5989584f, 1.55f, 1.5110244f, 1.4793651f };
    float cb = (float)(dmax > 0 ? dmax : 1);
    int ntap = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float ws = 1.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) ws = (L.k[d] == o) ? wsum[o] : ws;
        cb *= ws;
        ntap *= L.k[d] + 1;
    }
    const int hb = ((__float_as_int(cb * 1.0001f) >> 23) & 0xff) - 126;     // cb < 2^hb
    if (hb < 0 || hb > 20) return -1;
    return (float)(1 << hb) * sqrtf((float)ntap) <= 1032.f ? hb : -1;
}


} // namespace tiled
} // namespace ip
