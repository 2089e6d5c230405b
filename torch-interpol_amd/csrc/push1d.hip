// ===========================================================================
// push1d.hip -- 1-D push / count on LDS tiles (round 6).
//
// Reference semantics: interpol/nd.py:146-213 (push), pushpull.py:106-142 (count = push of ones), weights splines.py:30-80,
// index wrapping bounds.py:30-89, mask nd.py:10-27, all-linear weights iso1.py:13-20.
//
// Until round 6 every 1-D scatter ran in the generic kernel: one float atomic per tap and channel straight to the target,
// 35 x the time of the 1-D pull of the same size (profiles/r05_api_sweep.txt).  Here a workgroup of 256 threads owns 1024
// consecutive samples of one batch item (four per thread: coordinates and sources move as 16-byte pieces) and accumulates
// them in an LDS box of <= 4096 lattice points in packed 32-bit fixed point, two channels per 64-bit slot (ds_add_u64 -- the
// recipe of the 2-D tiles, ops_tiled2d.hip: scale from the tile's max |source| per channel, headroom from the measured
// density of first taps, tiled::headroom32); the touched slots are then added to the target with coalesced float atomics
// through the boundary condition (slot -> wrapped index and sign).  Samples whose stencil leaves the box (zooms beyond 4 x,
// wild coordinates), tiles without a usable fixed point (non-finite sources, extreme densities) scatter per thread.
// Orders 1..7 (compile time), every bound, the three extrapolation modes, f32 / bf16 / f16 sources into the float target,
// values / count / values + count; dense coordinate vectors only (a separable 1-D "grid" is resample1d.hip's business).
// ===========================================================================
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include "tile_common.hpp"
#include "sorted_util.hpp"
#include <type_traits>

namespace ip {
namespace p1d {

using tiled::Lattice;
using tiled::wave_min;
using tiled::wave_max;
using sorted::ld4;
using sorted::f2;

constexpr int NT = 256, VPT = 4, TS = NT * VPT;
constexpr int CAP = 4096;                                          // box slots

struct Smem {
    unsigned long long box[CAP];
    int lo, hi, dmax, cmax[2];
};

// slot of the box -> lattice offset (elements) and sign; interior points inline, the border out of line
__device__ __forceinline__ void slot_target(const Lattice &L, int idx, int &off, float &sg)
{
    if ((unsigned)idx < (unsigned)L.n[2] && !(L.bound[2] == B_DST1 && idx == 0)) { off = idx * L.ss[2]; sg = 1.f; return; }
    const long long pk = wrap_outofline(L.bound[2], idx, L.n[2]);
    off = (int)(pk & 0xffffffffll) * L.ss[2];
    sg = (float)(int)(pk >> 32);
}

// MODE 0: values, 1: count, 2: values + count (the target has C + 1 channels)
template <typename T, int K, int MODE>
__global__ __launch_bounds__(NT) void push1d(KParams p, const T *__restrict__ val, const float *__restrict__ grid, float *__restrict__ vol,
                                             int n, int ntiles)
{
    __shared__ Smem sm;
    if (p.gate && *p.gate) return;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / ntiles;
    const int tile = blockIdx.x % ntiles;
    const int o0 = tile * TS + tid * VPT;                            // the thread's four consecutive samples
    const bool whole = tile * TS + TS <= n;                          // block-uniform
    Lattice L;                                                       // the helpers are 3-D: the dim sits in slot z, x and y are degenerate
    L.bound[0] = 1; L.n[0] = 1; L.ss[0] = 0; L.k[0] = 0;
    L.bound[1] = 1; L.n[1] = 1; L.ss[1] = 0; L.k[1] = 0;
    L.bound[2] = p.bound[0]; L.n[2] = p.vol_n[0]; L.ss[2] = p.vol_ss[0] / 4; L.k[2] = K;
    L.lin = K == 1 && p.mode == MODE_ISO1;
    for (int e = tid; e < CAP; e += NT) sm.box[e] = 0ull;
    if (tid == 0) { sm.lo = 0x7fffffff; sm.hi = -0x7fffffff; sm.dmax = 0; sm.cmax[0] = 0; sm.cmax[1] = 0; }
    const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);
    const T *sp = val + b * p.val_sb;
    auto sources = [&](int ch) -> float4 {
        if (MODE == 1 || ch >= p.C) return make_float4(1.f, 1.f, 1.f, 1.f);
        const T *q = sp + ch * p.val_sc;
        if (whole) return ld4<T>(q + o0);
        float r[4];
#pragma unroll
        for (int v = 0; v < VPT; ++v) r[v] = Cvt<float, T>::ld(q[o0 + v < n ? o0 + v : n - 1]);
        return make_float4(r[0], r[1], r[2], r[3]);
    };
    float4 sv0 = sources(0), sv1 = sources(1 < nch ? 1 : 0);
    // coordinates, first taps, stencil coordinates, masks
    float t[VPT]; int i0[VPT];
    unsigned valid = 0, inb = 0, in = 0;
    {
        const float *gp = grid + b * p.grid_sb;
        float c[VPT];
        if (whole) { const float4 g4 = ld4<float>(gp + o0); c[0] = g4.x; c[1] = g4.y; c[2] = g4.z; c[3] = g4.w; }
        else {
#pragma unroll
            for (int v = 0; v < VPT; ++v) c[v] = gp[o0 + v < n ? o0 + v : n - 1];
        }
        int mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (o0 + v < n) valid |= 1u << v;
            if (p.extrapolate == 1 || (c[v] > p.mask_lo_f && c[v] < p.mask_hi_f[0])) inb |= 1u << v;     // nd.py:10-27
            tiled::split(K, c[v], i0[v], t[v]);
            if ((valid >> v) & 1) { mn = i0[v] < mn ? i0[v] : mn; mx = i0[v] > mx ? i0[v] : mx; }
        }
        mn = wave_min(mn); mx = wave_max(mx);
        __syncthreads();                                             // (box cleared, lo / hi initialised)
        if ((tid & 63) == 0) { atomicMin(&sm.lo, mn); atomicMax(&sm.hi, mx); }
        __syncthreads();
    }
    int lo = 0, S = 0;                                               // box = lattice points lo ... lo + S - 1
    if (sm.hi >= sm.lo) {
        const long long span = (long long)sm.hi + K + 1 - sm.lo;      // (first taps are clamped to +- 2^30)
        lo = span > CAP ? sm.lo + (int)((span - CAP) / 2) : sm.lo;   // too wide: keep the centre, the rest scatters per thread
        S = span > CAP ? CAP : (int)span;
    }
    int cell[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        if (((valid >> v) & 1) && i0[v] >= lo && i0[v] <= lo + S - K - 1 && t[v] == t[v]) in |= 1u << v;     // (a NaN coordinate: per thread, like the generic kernel)
        cell[v] = ((in >> v) & 1) ? i0[v] - lo : 0;
    }
    // density: samples per first-tap cell (16-bit counters in the box, cleared again)
    {
        unsigned *cnt32 = reinterpret_cast<unsigned *>(sm.box);
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if ((in >> v) & 1) atomicAdd(&cnt32[cell[v] >> 1], 1u << (16 * (cell[v] & 1)));
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!((in >> v) & 1)) continue;
            const int cv = (int)((cnt32[cell[v] >> 1] >> (16 * (cell[v] & 1))) & 0xffffu);
            m = cv > m ? cv : m;
        }
        m = wave_max(m);
        if ((tid & 63) == 0 && m > 0) atomicMax(&sm.dmax, m);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VPT; ++v) if ((in >> v) & 1) cnt32[cell[v] >> 1] = 0u;
    }
    for (int cg = 0; cg < nch; cg += 2) {
        const bool two = cg + 1 < nch;
        float *vc0 = vol + b * p.vol_sb + cg * p.vol_sc;
        float *vc1 = two ? vc0 + p.vol_sc : vc0;
        f2 src[VPT];
        float am0 = 0.f, am1 = 0.f;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const float m = (float)((inb >> v) & 1);
            const float r0 = v == 0 ? sv0.x : (v == 1 ? sv0.y : (v == 2 ? sv0.z : sv0.w)), r1 = v == 0 ? sv1.x : (v == 1 ? sv1.y : (v == 2 ? sv1.z : sv1.w));
            src[v] = f2{ r0 * m, two ? r1 * m : 0.f };
            if ((in >> v) & 1) {
                const float a0 = __builtin_fabsf(src[v].x), a1 = __builtin_fabsf(src[v].y);
                am0 = (a0 > am0 || a0 != a0) ? a0 : am0; am1 = (a1 > am1 || a1 != a1) ? a1 : am1;
            }
        }
        if (cg + 2 < nch) { sv0 = sources(cg + 2); sv1 = sources(cg + 3 < nch ? cg + 3 : cg + 2); }   // next pair: in flight during the taps
        {
            const int m0 = wave_max(__float_as_int(am0)), m1 = wave_max(__float_as_int(am1));
            if ((tid & 63) == 0) { if (m0) atomicMax(&sm.cmax[0], m0); if (m1) atomicMax(&sm.cmax[1], m1); }
        }
        __syncthreads();
        const int hb = tiled::headroom32(L, sm.dmax);
        const int mb0 = sm.cmax[0], mb1 = sm.cmax[1];
        const bool fixedpt = hb >= 0 && (mb0 & 0x7f800000) != 0x7f800000 && (mb1 & 0x7f800000) != 0x7f800000;
        int ex0 = ((mb0 >> 23) & 0xff) - 127, ex1 = ((mb1 >> 23) & 0xff) - 127;
        ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
        const int hbc = hb < 0 ? 0 : hb;
        const f2 scale = { mb0 ? __int_as_float((127 + 29 - ex0 - hbc) << 23) : 0.f, mb1 ? __int_as_float((127 + 29 - ex1 - hbc) << 23) : 0.f };
        const float inv0 = __int_as_float((127 - 29 + ex0 + hbc) << 23), inv1 = __int_as_float((127 - 29 + ex1 + hbc) << 23);
        if (fixedpt) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!((in >> v) & 1)) continue;
                float w[K + 1];
                tiled::weights<K>(L.lin, K, t[v], w);
                unsigned long long *bp = sm.box + cell[v];
                const f2 ss = src[v] * scale;
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    const f2 pr = ss * f2{ w[j], w[j] };
                    const int q0 = tiled::cvt_rpi(pr.x), q1 = tiled::cvt_rpi(pr.y);
                    atomicAdd(bp + j, ((unsigned long long)(unsigned)(q1 + (q0 >> 31)) << 32) | (unsigned)q0);
                }
            }
        }
        // stencils outside the box, or no fixed point for this tile: float atomics straight to the target
        if (!fixedpt || in != valid) {
#pragma unroll 1
            for (int v = 0; v < VPT; ++v) {
                if (!((valid >> v) & 1) || (fixedpt && ((in >> v) & 1))) continue;
                tiled::scatter_one_thread(L, vc0, src[v].x, 0, 0, i0[v], 0.f, 0.f, t[v]);
                if (two) tiled::scatter_one_thread(L, vc1, src[v].y, 0, 0, i0[v], 0.f, 0.f, t[v]);
            }
        }
        __syncthreads();
        if (tid < 2) sm.cmax[tid] = 0;                               // (every thread read them before the barrier above)
        if (fixedpt) {
            // consecutive threads flush consecutive lattice points: coalesced atomics
            for (int s0 = tid; s0 < S; s0 += NT) {
                const long long a = (long long)sm.box[s0];
                if (a == 0) continue;
                sm.box[s0] = 0ull;
                const int lo_ = (int)(a & 0xffffffffll);
                const int hi_ = (int)((a - (long long)lo_) >> 32);
                int off; float sg;
                slot_target(L, lo + s0, off, sg);
                if (lo_ != 0) __hip_atomic_fetch_add(vc0 + off, (float)lo_ * (inv0 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hi_ != 0) __hip_atomic_fetch_add(vc1 + off, (float)hi_ * (inv1 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

template <typename T, int MODE>
static int launch_k(int K, const KParams &k, const void *val, const void *grid, void *vol, int n, int ntiles, int64_t B, hipStream_t st)
{
    const dim3 g((unsigned)(ntiles * B));
#define IP_P1D(KK) case KK: hipLaunchKernelGGL((push1d<T, KK, MODE>), g, dim3(NT), 0, st, k, (const T *)val, (const float *)grid, (float *)vol, n, ntiles); break;
    switch (K) { IP_P1D(1) IP_P1D(2) IP_P1D(3) IP_P1D(4) IP_P1D(5) IP_P1D(6) IP_P1D(7) default: return 0; }
#undef IP_P1D
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? (int)e : 1;
}

} // namespace p1d

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

// `vol` is the zero-filled (or accumulating) FLOAT target; val == NULL: count.  1: done, 0: declined, else an error.
int IP_SYM(try_push1d_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    using T = IP_TT;
    if (p->dim != 1 || k.sep != 0 || (k.dbg & 32)) return 0;
    if (k.order[0] < 1 || k.order[0] > 7) return 0;
    const int64_t n = p->grid_shape[0];
    if (n > 0x7fffffff - p1d::TS || n * p->batch < 4096) return 0;  // (small problems: the generic kernel's single launch)
    const int64_t ntiles = (n + p1d::TS - 1) / p1d::TS;
    if (ntiles * p->batch > 0x7fffffff) return 0;
    if (!val) return p1d::launch_k<T, 1>(k.order[0], k, val, grid, vol, (int)n, (int)ntiles, p->batch, st);
    if (k.cc) return p1d::launch_k<T, 2>(k.order[0], k, val, grid, vol, (int)n, (int)ntiles, p->batch, st);
    return p1d::launch_k<T, 0>(k.order[0], k, val, grid, vol, (int)n, (int)ntiles, p->batch, st);
}

} // namespace ip
