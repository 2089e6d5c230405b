// ===========================================================================
// sorted_util.hpp -- helpers shared by the class-sorted tile kernels (ops_sorted.hip) and the
// owner-computes scatter (push_owner.hip): optimisation fences, packed float pairs, 32-lane scans,
// tile geometry of the sample grid, coordinate loads, 16-byte loads / stores, the extrapolation
// mask, the weights of a stencil in packed form, phase profiling, launch helpers.
// ===========================================================================
#pragma once
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include "tile_common.hpp"
#include <type_traits>

namespace ip {
namespace sorted {

using tiled::Lattice;
using tiled::split;
using tiled::wave_max;
using tiled::wave_min;
using tiled::wave_sum;
using tiled::WorkRange;

constexpr int TS = 16;                          // edge of a tile of the sample grid

#ifdef IP_PROF
static __device__ unsigned long long g_prof[16];
#endif
__device__ __forceinline__ void prof_mark(int i)
{
#ifdef IP_PROF
    __shared__ unsigned long long t0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long n = clock64();
        if (i >= 0) atomicAdd(&g_prof[i], n - t0);
        t0 = n;
    }
#endif
}

// A value the optimiser cannot see through.  The kernels are persistent loops (tiles > channel
// pairs > passes) whose phases all derive addresses from the thread index: left alone, LICM hoists
// every such value to the outermost level and the register allocator spills them by the hundred.
// Re-deriving them from an opaque copy at the top of a phase costs a few VALU operations.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

typedef float f2 __attribute__((ext_vector_type(2)));

// exclusive prefix sum over the 32 lanes of each half wave (both halves hold the same data)
__device__ __forceinline__ int half_excl_scan(int v, int &total)
{
    const int lane = __lane_id() & 31;
    int s = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(s, o, 32);
        if (lane >= o) s += t;
    }
    total = __shfl(s, 31, 32);
    return s - v;
}

struct TileGeom { int gx, gy, gz, ox0, oy0, oz0; };
__device__ __forceinline__ TileGeom tile_geom(int tile, int gx, int gy, int gz, int nty, int ntz)
{
    const int tzi = tile % ntz; tile /= ntz;
    return TileGeom{ gx, gy, gz, (tile / nty) * TS, (tile % nty) * TS, tzi * TS };
}

// natural order: sample `id` of the tile -> position in the sample grid
__device__ __forceinline__ void sample_pos(const TileGeom &g, int id, int &ox, int &oy, int &oz)
{
    ox = g.ox0 + (id >> 8); oy = g.oy0 + ((id >> 4) & 15); oz = g.oz0 + (id & 15);
}

// GM: 0 dense (B,*out,3) grid, 1 separable lattice (three coordinate vectors back to back),
// 2 displacement field (identity added in registers, api.py:490-513), 3 affine lattice (api.py:534-572)
template <int GM>
__device__ __forceinline__ void load_xyz(const KParams &p, const float *__restrict__ grid, int64_t b, const TileGeom &g,
                                         int ox, int oy, int oz, float *x)
{
    if (GM == 1) {
        x[0] = grid[ox]; x[1] = grid[g.gx + oy]; x[2] = grid[g.gx + g.gy + oz];
    } else if (GM == 3) {
        // affine lattice (affine_grid, api.py:534-572): the 3 x 4 matrix [A | t] behind `grid` (uniform: scalar loads);
        // same operation order as the generic kernels (stencil.hpp)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            x[d] = __builtin_fmaf(grid[4 * d + 2], (float)oz, __builtin_fmaf(grid[4 * d + 1], (float)oy, grid[4 * d] * (float)ox)) + grid[4 * d + 3];
    } else {
        const float *gp = grid + b * p.grid_sb + (((int64_t)ox * g.gy + oy) * g.gz + oz) * 3;
        x[0] = gp[0]; x[1] = gp[1]; x[2] = gp[2];
        if (GM == 2) { x[0] += (float)ox; x[1] += (float)oy; x[2] += (float)oz; }
    }
}

// four consecutive elements as floats (one 16-byte load for fp32, 8 bytes for the 16-bit types;
// 4-byte / 2-byte alignment is all that is asked for)
template <typename T>
__device__ __forceinline__ float4 ld4(const T *p)
{
    if constexpr (std::is_same<T, float>::value) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f4u v = *reinterpret_cast<const f4u *>(p);
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        typedef unsigned short h4u __attribute__((ext_vector_type(4), aligned(2)));
        const h4u v = *reinterpret_cast<const h4u *>(p);
        T e[4];
        __builtin_memcpy(e, &v, 8);
        return make_float4(Cvt<float, T>::ld(e[0]), Cvt<float, T>::ld(e[1]), Cvt<float, T>::ld(e[2]), Cvt<float, T>::ld(e[3]));
    }
}

template <typename T>
__device__ __forceinline__ void st4(T *p, float4 v)
{
    if constexpr (std::is_same<T, float>::value) {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        *reinterpret_cast<f4u *>(p) = f4u{ v.x, v.y, v.z, v.w };
    } else {
        typedef unsigned short h4u __attribute__((ext_vector_type(4), aligned(2)));
        if constexpr (std::is_same<T, bf16_t>::value) {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            typedef float f4 __attribute__((ext_vector_type(4)));
            const b4 h = __builtin_convertvector((f4{ v.x, v.y, v.z, v.w }), b4);
            h4u w;
            __builtin_memcpy(&w, &h, 8);
            *reinterpret_cast<h4u *>(p) = w;
            return;
        }
        const T e[4] = { Cvt<float, T>::st(v.x), Cvt<float, T>::st(v.y), Cvt<float, T>::st(v.z), Cvt<float, T>::st(v.w) };
        h4u w;
        __builtin_memcpy(&w, e, 8);
        *reinterpret_cast<h4u *>(p) = w;
    }
}

// extrapolation mask of a sample (nd.py:10-27): 1 or 0
__device__ __forceinline__ float inb_mask(const KParams &p, const float *x)
{
    if (p.extrapolate == 1) return 1.f;
    const bool inb = x[0] > p.mask_lo_f && x[0] < p.mask_hi_f[0] && x[1] > p.mask_lo_f
                  && x[1] < p.mask_hi_f[1] && x[2] > p.mask_lo_f && x[2] < p.mask_hi_f[2];
    return inb ? 1.f : 0.f;
}

// The K + 1 weights of the y- and z-stencils of a sample at once (packed math: component x of
// the vectors is y, component y is z), from the stencil coordinates t = x - i0 (nd.py:46,
// splines.py:30-80).  With t in [(K-1)/2, (K+1)/2) every tap sits on a known polynomial piece, so
// the |t - j| tests of the per-tap form fold away: 12 packed operations for both cubic stencils.
template <int K>
__device__ __forceinline__ void weights_yz(f2 t, f2 *w)
{
    if (K == 3) {
        // t in [1, 2): u = t - 1 in [0, 1), v = 1 - u;  taps at distances 1 + u, u, v, 1 + v
        const f2 u = t - 1.f, v = 2.f - t;
        const f2 u2 = u * u, v2 = v * v;
        w[0] = (v2 * v) * (1.f / 6.f);
        w[3] = (u2 * u) * (1.f / 6.f);
        w[1] = u2 * (u * 0.5f - 1.f) + 2.f / 3.f;
        w[2] = v2 * (v * 0.5f - 1.f) + 2.f / 3.f;
    } else if (K == 1) {
        // t in [0, 1): taps at distances t, 1 - t (iso1.py:19-20)
        w[0] = 1.f - t; w[1] = t; w[2] = f2{ 0.f, 0.f }; w[3] = f2{ 0.f, 0.f };
    } else {
        // K == 2, t in [0.5, 1.5): taps at distances t, |t - 1|, 2 - t
        const f2 a = 1.5f - t, c = t - 0.5f, m = t - 1.f;
        w[0] = (a * a) * 0.5f;
        w[1] = 0.75f - m * m;
        w[2] = (c * c) * 0.5f;
        w[3] = f2{ 0.f, 0.f };
    }
}
// one weight, tap i (branch-free form of splines.py:30-44)
template <int K>
__device__ __forceinline__ float weight_x(float t, int i)
{
    const float d = __builtin_fabsf(t - (float)i);
    if (K == 3) {
        const float e = 2.f - d;
        const float near = __builtin_fmaf(d * d, __builtin_fmaf(d, 0.5f, -1.f), 2.f / 3.f), far = (e * e * e) * (1.f / 6.f);
        return d < 1.f ? near : far;
    } else if (K == 1) {
        return i == 0 ? 1.f - t : (i == 1 ? t : 0.f);
    } else {
        const float e = 1.5f - d;
        const float near = 0.75f - d * d, far = 0.5f * (e * e);
        return i > 2 ? 0.f : (d < 0.5f ? near : far);
    }
}

// derivatives of the same weights with respect to the stencil coordinate (splines.py:90-139 on the interval of `split`)
template <int K>
__device__ __forceinline__ void wgrads_yz(f2 t, f2 *g)
{
    if (K == 3) {
        const f2 u = t - 1.f, v = 2.f - t;
        g[0] = (v * v) * -0.5f;
        g[3] = (u * u) * 0.5f;
        g[1] = u * (u * 1.5f - 2.f);
        g[2] = v * (v * -1.5f + 2.f);
    } else if (K == 1) {
        // all-linear stencils take the reference's iso1 path: -1, +1 (iso1.py:311-313)
        g[0] = f2{ -1.f, -1.f }; g[1] = f2{ 1.f, 1.f }; g[2] = f2{ 0.f, 0.f }; g[3] = f2{ 0.f, 0.f };
    } else {
        // K == 2: w0 = a^2 / 2 (a = 1.5 - t), w1 = 0.75 - m^2 (m = t - 1), w2 = c^2 / 2 (c = t - 0.5)
        g[0] = t - 1.5f;
        g[1] = (t - 1.f) * -2.f;
        g[2] = t - 0.5f;
        g[3] = f2{ 0.f, 0.f };
    }
}
template <int K>
__device__ __forceinline__ float wgrad_x(float t, int i)
{
    const float x = t - (float)i, d = __builtin_fabsf(x);
    const float s = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    if (K == 3) {
        const float e = 2.f - d;
        return (d < 1.f ? d * __builtin_fmaf(d, 1.5f, -2.f) : -0.5f * (e * e)) * s;
    } else if (K == 1) {
        return i == 0 ? -1.f : (i == 1 ? 1.f : 0.f);                 // iso1.py:311-313
    } else {
        return i > 2 ? 0.f : (d < 0.5f ? -2.f * d : d - 1.5f) * s;
    }
}

// One out-of-box sample gathered tap-parallel by a wave: lane = tap; returns this lane's products
// for the two channels (to be summed over the wave) and the sample's extrapolation mask.
// Mixed orders (1..3 per dim, kernel-uniform) in the cubic tiles: the four weights of one dim in the closed forms of weights_yz /
// wgrads_yz above, chosen by the dim's order (a compile-time constant after mix_dispatch); taps beyond the order 0
__device__ __forceinline__ void weights4(int k, float t, float *w)
{
    if (k == 3) {
        const float u = t - 1.f, v = 2.f - t, u2 = u * u, v2 = v * v;
        w[0] = (v2 * v) * (1.f / 6.f); w[3] = (u2 * u) * (1.f / 6.f);
        w[1] = u2 * (u * 0.5f - 1.f) + 2.f / 3.f; w[2] = v2 * (v * 0.5f - 1.f) + 2.f / 3.f;
    } else if (k == 2) {
        const float a = 1.5f - t, c = t - 0.5f, m = t - 1.f;
        w[0] = (a * a) * 0.5f; w[1] = 0.75f - m * m; w[2] = (c * c) * 0.5f; w[3] = 0.f;
    } else {
        w[0] = 1.f - t; w[1] = t; w[2] = 0.f; w[3] = 0.f;
    }
}
// (order 1 outside the all-linear mode: the reference's nd path, +sign(t - j) -- quirk B-4, spline_math.hpp: bspline_g)
__device__ __forceinline__ void wgrads4(int k, float t, float *g)
{
    if (k == 3) {
        const float u = t - 1.f, v = 2.f - t;
        g[0] = (v * v) * -0.5f; g[3] = (u * u) * 0.5f; g[1] = u * (u * 1.5f - 2.f); g[2] = v * (v * -1.5f + 2.f);
    } else if (k == 2) {
        g[0] = t - 1.5f; g[1] = (t - 1.f) * -2.f; g[2] = t - 0.5f; g[3] = 0.f;
    } else {
        g[0] = t > 0.f ? 1.f : 0.f; g[1] = -1.f; g[2] = 0.f; g[3] = 0.f;
    }
}
__device__ __forceinline__ void mixed_weights_yz(int ky, int kz, f2 t, f2 *w)
{
    float wy[4], wz[4];
    weights4(ky, t.x, wy);
    weights4(kz, t.y, wz);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = f2{ wy[j], wz[j] };
}
__device__ __forceinline__ void mixed_wgrads_yz(int ky, int kz, f2 t, f2 *g)
{
    float gy[4], gz[4];
    wgrads4(ky, t.x, gy);
    wgrads4(kz, t.y, gz);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = f2{ gy[j], gz[j] };
}
// the x-weight of tap i (per lane: i = the tap this pass serves) and its derivative
__device__ __forceinline__ float mixed_weight_x(int kx, float t, int i)
{
    return kx == 3 ? weight_x<3>(t, i) : (kx == 2 ? weight_x<2>(t, i) : weight_x<1>(t, i));
}
__device__ __forceinline__ float mixed_wgrad_x(int kx, float t, int i)
{
    return kx == 3 ? wgrad_x<3>(t, i) : (kx == 2 ? wgrad_x<2>(t, i) : (i == 0 ? (t > 0.f ? 1.f : 0.f) : (i == 1 ? -1.f : 0.f)));
}
// where the orders are runtime (push_owner.hip: own_gather<KMIX>): the slots of taps beyond a dim's order cleared after the reads
// (t2[4 * jy + k]; 0 * inf is not 0: a non-finite lattice point outside the true stencil must not reach the sums)
__device__ __forceinline__ void clear_unused_taps(int ky, int kz, f2 *t2)
{
    const f2 z = { 0.f, 0.f };
    if (kz < 3) { t2[3] = z; t2[7] = z; t2[11] = z; t2[15] = z; }
    if (kz < 2) { t2[2] = z; t2[6] = z; t2[10] = z; t2[14] = z; }
    if (ky < 3) { t2[12] = z; t2[13] = z; t2[14] = z; t2[15] = z; }
    if (ky < 2) { t2[8] = z; t2[9] = z; t2[10] = z; t2[11] = z; }
}
// f(kx, ky, kz) with the three orders (1..3 each, kernel-uniform) as integral constants
template <typename F>
__device__ __forceinline__ void mix_dispatch(int kx, int ky, int kz, F &&f)
{
    using std::integral_constant;
    auto dz = [&](auto a, auto b) {
        if (kz == 1) f(a, b, integral_constant<int, 1>{}); else if (kz == 2) f(a, b, integral_constant<int, 2>{}); else f(a, b, integral_constant<int, 3>{});
    };
    auto dy = [&](auto a) {
        if (ky == 1) dz(a, integral_constant<int, 1>{}); else if (ky == 2) dz(a, integral_constant<int, 2>{}); else dz(a, integral_constant<int, 3>{});
    };
    if (kx == 1) dy(integral_constant<int, 1>{}); else if (kx == 2) dy(integral_constant<int, 2>{}); else dy(integral_constant<int, 3>{});
}

template <int K, bool MIX>
__device__ __forceinline__ float mixed_or_weight_x(int kx, float t, int i)
{
    if constexpr (MIX) return mixed_weight_x(kx, t, i);
    else return weight_x<K>(t, i);
}

template <typename T, int K, int GM, bool MIX = false>
__device__ __forceinline__ void slow_taps(const KParams &p, const Lattice &L, const float *__restrict__ grid, int64_t b, const TileGeom &g,
                                          int id, int lane, const T *__restrict__ vc0, const T *__restrict__ vc1, float &a0, float &a1, float &m)
{
    int ox, oy, oz; float x[3];
    sample_pos(g, id, ox, oy, oz);
    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
    int off;
    const float w = tiled::tap_weight_t<MIX ? -1 : K, MIX ? -1 : K>(L, x[0], x[1], x[2], lane, &off, nullptr);
    a0 = 0.f; a1 = 0.f;
    if (lane < (MIX ? (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1) : (K + 1) * (K + 1) * (K + 1))) { a0 = w * Cvt<float, T>::ld(vc0[off]); a1 = w * Cvt<float, T>::ld(vc1[off]); }
    m = inb_mask(p, x);
}

// ---------------------------------------------------------------------------
// Launch helpers
// ---------------------------------------------------------------------------
static int cu_count()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// Kernels that need more than 64 KiB of dynamic LDS opt in, once per kernel and device.
template <auto Kernel>
static int big_lds(size_t bytes)
{
    static bool done[64] = { false };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute((const void *)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = true;
    return 0;
}

} // namespace sorted
} // namespace ip
