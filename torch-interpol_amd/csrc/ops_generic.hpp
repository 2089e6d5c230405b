// ===========================================================================
// ops_generic.hpp -- the generic fused sampling kernels (any D in 1..3, any
// order 0..7, any boundary, any extrapolation mode, any supported dtype).
//
// One thread = one sample point.  The whole (K+1)^D stencil is evaluated in
// registers and gathered from / scattered to global memory directly (served by
// L1/L2/Infinity Cache); there is no (B,N)-sized temporary of any kind.  This
// replaces the (K+1)^D full-tensor passes of reference interpol/nd.py:118-136
// (pull), 187-210 (push), 252-281 (grad), 329-361 (pushgrad), 405-446 (hess)
// and the hand-unrolled iso1.py / iso0.py special cases.
//
// The specialised LDS-tiled kernels (ops_tiled_*.hip) take over for the
// configurations they cover; these kernels are the reference-complete path.
// ===========================================================================
#pragma once
#include "stencil.hpp"

namespace ip {

constexpr int BLOCK = 256;

// Tap addressing: a wave-uniform 64-bit base (SGPR pair: batch item + channel)
// plus a 32-bit per-lane BYTE offset -- the `global_load_dword v, v_off, s[base]`
// form -- so a tap costs one integer add instead of a 64-bit address build.
template <typename T> __device__ __forceinline__ T ld_tap(const T *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

__device__ __forceinline__ float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename AccT> __device__ __forceinline__ void atomic_add(AccT *base, unsigned byte_off, AccT v)
{
    // relaxed, device scope, no return value -> global_atomic_add_f32 / _f64
    AccT *p = reinterpret_cast<AccT *>(reinterpret_cast<char *>(base) + byte_off);
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Tap loops.  The stencil arrays live in registers, so all three loops are
// fully unrolled (static indices).  Two things keep the register count sane:
//   * separable accumulation (row -> plane -> total): no (K+1)^D weight
//     products exist that loop-invariant code motion could hoist out of the
//     channel loop, and the inner loop is one FMA per tap;
//   * opaque(): the per-dim offsets are made opaque once per channel iteration
//     so the (K+1)^D offset sums are not hoisted either.
// ---------------------------------------------------------------------------
#define IP_FOR_I _Pragma("unroll") for (int i = 0; i < Taps<D, KMAX>::T0; ++i) if (s.on(p, 0, i))
#define IP_FOR_J _Pragma("unroll") for (int j = 0; j < Taps<D, KMAX>::T1; ++j) if (s.on(p, 1, j))
#define IP_FOR_K _Pragma("unroll") for (int k = 0; k < Taps<D, KMAX>::T2; ++k) if (s.on(p, 2, k))
// closes one innermost row: a scheduling fence for the big stencils, so that the
// compiler issues the gathers a row at a time instead of all (K+1)^D up front
#define IP_ROW_END do { if (Taps<D, KMAX>::T0 * Taps<D, KMAX>::T1 * Taps<D, KMAX>::T2 > 16) __builtin_amdgcn_sched_barrier(0); } while (0)

// ---------------------------------------------------------------------------
// Which samples a thread visits.
//   DEF == false : sample o = blockIdx.x * BLOCK + threadIdx.x of the batch items blockIdx.y, + gridDim.y, ...
//   DEF == true  : the samples of the tiles that an LDS-tiled kernel handed back (TileList, stencil.hpp: deformations
//                  too rough for its box).  Persistent blocks walk
//                  the descriptor list; the 256 threads of a block share a tile, z fastest.
// ---------------------------------------------------------------------------
template <bool DEF, int D> struct SampleIter;
template <int D> struct SampleIter<false, D> {
    int64_t b, o; bool ok;
    __device__ __forceinline__ SampleIter(const KParams &p, int B, const TileList &)
    {
        o = (int64_t)blockIdx.x * BLOCK + threadIdx.x; b = blockIdx.y; ok = o < p.N && b < B;
    }
    __device__ __forceinline__ void next(const KParams &, int B, const TileList &) { b += gridDim.y; ok = b < B; }
};
template <int D> struct SampleIter<true, D> {
    int64_t b, o; bool ok;
    int w, s, c[3]; bool have;
    __device__ __forceinline__ SampleIter(const KParams &p, int B, const TileList &tl)
    {
        // nothing handed back to this block (the usual case): one parallel look at its descriptors, not a serial walk
        int any = 0;
        for (int ww = blockIdx.x + threadIdx.x * gridDim.x; ww < tl.nwork; ww += BLOCK * gridDim.x) any |= (int)(tl.gen[ww] == tl.cur);
        if (!__syncthreads_or(any)) { ok = false; return; }
        w = blockIdx.x; have = false; next(p, B, tl);
    }
    __device__ __forceinline__ void next(const KParams &p, int, const TileList &tl)
    {
        const int ns = tl.e[0] * tl.e[1] * tl.e[2];
        for (;;) {
            if (w >= tl.nwork) { ok = false; return; }
            if (!have) {
                if (tl.gen[w] != tl.cur) { w += gridDim.x; continue; }
                const unsigned long long d = tl.desc[w];
                b = (int64_t)((d >> 42) & 0xfffffull);
                c[0] = (int)((d >> 28) & 0x3fffull) * tl.e[0]; c[1] = (int)((d >> 14) & 0x3fffull) * tl.e[1]; c[2] = (int)(d & 0x3fffull) * tl.e[2];
                have = true; s = threadIdx.x;
            } else {
                s += BLOCK;
            }
            if (s >= ns) { have = false; w += gridDim.x; continue; }
            // tile dims (x, y, z) are the LAST D dims of the kernel families: problem dim d <-> tile dim d + 3 - D
            int r = s, pos[3];
            pos[2] = c[2] + r % tl.e[2]; r /= tl.e[2];
            pos[1] = c[1] + r % tl.e[1]; r /= tl.e[1];
            pos[0] = c[0] + r;
            bool in = true; int64_t lin = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) { const int q = pos[d + 3 - D]; in = in && q < p.gshape[d]; lin = lin * p.gshape[d] + q; }
            if (!in) continue;
            o = lin; ok = true; return;
        }
    }
};
#define IP_FOR_SAMPLES for (SampleIter<DEF, D> it_(p, B, tl); it_.ok; it_.next(p, B, tl))

template <typename S> __device__ __forceinline__ void opaque(S &s)
{
#pragma unroll
    for (int i = 0; i < S::T; ++i) asm volatile("" : "+v"(s.off[0][i]));
}

// ---------------------------------------------------------------------------
// pull : val[b,c,o] = mask * sum_taps w * vol[b,c,tap]          (nd.py:80-143)
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, int D, int KMAX, bool ISO, bool DEF = false>
__global__ __launch_bounds__(BLOCK) void pull_generic(KParams p, const T *__restrict__ vol,
                                                      const G *__restrict__ grid, T *__restrict__ val, int B, TileList tl)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    if (p.gate_n == -2 && p.gate && *p.gate != 2) return;   // 2-D router (scatter2d.hip: probe2d): this kernel is the organisation of expanding fields only
    IP_FOR_SAMPLES {
        const int64_t b = it_.b, o = it_.o;
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_W> s;
        s.setup(p, x);
        const T *vb = vol + b * p.vol_sb;
        T *ob = val + b * p.val_sb + o;
        for (int c = 0; c < p.C; ++c) {
            const T *v0 = vb + c * p.vol_sc;
            opaque(s);
            R a0 = R(0);
            IP_FOR_I {
                R p0 = R(0);
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    R r0 = R(0);
                    IP_FOR_K {
                        r0 = fma_(s.w[2][k], Cvt<R, T>::ld(ld_tap(v0, oij + s.off[2][k])), r0);
                    } IP_ROW_END;
                    p0 = fma_(s.w[1][j], r0, p0);
                }
                a0 = fma_(s.w[0][i], p0, a0);
            }
            ob[c * p.val_sc] = Cvt<R, T>::st(a0 * s.mask);
        }
    }
}

// ---------------------------------------------------------------------------
// grad : val[b,c,o,d] = mask * sum_taps (g_d prod_{e!=d} w_e) * vol[b,c,tap]   (nd.py:216-288)
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, int D, int KMAX, bool ISO, bool DEF = false>
__global__ __launch_bounds__(BLOCK) void grad_generic(KParams p, const T *__restrict__ vol,
                                                      const G *__restrict__ grid, T *__restrict__ val, int B, TileList tl)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    IP_FOR_SAMPLES {
        const int64_t b = it_.b, o = it_.o;
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_G> s;
        s.setup(p, x);
        const T *vb = vol + b * p.vol_sb;
        T *ob = val + b * p.val_sb + o * D;
        for (int c = 0; c < p.C; ++c) {
            const T *v0 = vb + c * p.vol_sc;
            opaque(s);
            R a0 = R(0), a1 = R(0), a2 = R(0);
            IP_FOR_I {
                R pWW = R(0), pGW = R(0), pWG = R(0);
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    R rW = R(0), rG = R(0);
                    IP_FOR_K {
                        const R v = Cvt<R, T>::ld(ld_tap(v0, oij + s.off[2][k]));
                        rW = fma_(s.w[2][k], v, rW);
                        if (D > 2) rG = fma_(s.g[2][k], v, rG);
                    } IP_ROW_END;
                    pWW = fma_(s.w[1][j], rW, pWW);
                    if (D > 1) pGW = fma_(s.g[1][j], rW, pGW);
                    if (D > 2) pWG = fma_(s.w[1][j], rG, pWG);
                }
                a0 = fma_(s.g[0][i], pWW, a0);
                if (D > 1) a1 = fma_(s.w[0][i], pGW, a1);
                if (D > 2) a2 = fma_(s.w[0][i], pWG, a2);
            }
            T *oc = ob + c * p.val_sc;
            oc[0] = Cvt<R, T>::st(a0 * s.mask);
            if (D > 1) oc[1] = Cvt<R, T>::st(a1 * s.mask);
            if (D > 2) oc[2] = Cvt<R, T>::st(a2 * s.mask);
        }
    }
}

// ---------------------------------------------------------------------------
// hess : val[b,c,o,d,e] symmetric                                 (nd.py:367-464)
//   diagonal  : h_d prod_{f!=d} w_f       off-diagonal: g_d g_e prod_{f!=d,e} w_f
// (for extrapolate in {0,2} the reference raises -- its bug B-2; the intended
//  out*mask is implemented)
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, int D, int KMAX, bool ISO>
__global__ __launch_bounds__(BLOCK) void hess_generic(KParams p, const T *__restrict__ vol,
                                                      const G *__restrict__ grid, T *__restrict__ val, int B)
{
    const int64_t o = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= p.N) return;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_H> s;
        s.setup(p, x);
        const T *vb = vol + b * p.vol_sb;
        T *ob = val + b * p.val_sb + o * (D * D);
        for (int c = 0; c < p.C; ++c) {
            const T *v0 = vb + c * p.vol_sc;
            opaque(s);
            R hxx = 0, hyy = 0, hzz = 0, hxy = 0, hxz = 0, hyz = 0;
            IP_FOR_I {
                R pWW = 0, pGW = 0, pHW = 0, pWG = 0, pGG = 0, pWH = 0;
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    R rW = 0, rG = 0, rH = 0;
                    IP_FOR_K {
                        const R v = Cvt<R, T>::ld(ld_tap(v0, oij + s.off[2][k]));
                        rW = fma_(s.w[2][k], v, rW);
                        if (D > 2) { rG = fma_(s.g[2][k], v, rG); rH = fma_(s.h[2][k], v, rH); }
                    } IP_ROW_END;
                    pWW = fma_(s.w[1][j], rW, pWW);
                    if (D > 1) { pGW = fma_(s.g[1][j], rW, pGW); pHW = fma_(s.h[1][j], rW, pHW); }
                    if (D > 2) { pWG = fma_(s.w[1][j], rG, pWG); pGG = fma_(s.g[1][j], rG, pGG); pWH = fma_(s.w[1][j], rH, pWH); }
                }
                hxx = fma_(s.h[0][i], pWW, hxx);
                if (D > 1) { hyy = fma_(s.w[0][i], pHW, hyy); hxy = fma_(s.g[0][i], pGW, hxy); }
                if (D > 2) { hzz = fma_(s.w[0][i], pWH, hzz); hxz = fma_(s.g[0][i], pWG, hxz); hyz = fma_(s.w[0][i], pGG, hyz); }
            }
            T *oc = ob + c * p.val_sc;
            const R m = s.mask;
            if (D == 1) { oc[0] = Cvt<R, T>::st(hxx * m); }
            if (D == 2) {
                oc[0] = Cvt<R, T>::st(hxx * m); oc[1] = Cvt<R, T>::st(hxy * m);
                oc[2] = Cvt<R, T>::st(hxy * m); oc[3] = Cvt<R, T>::st(hyy * m);
            }
            if (D == 3) {
                oc[0] = Cvt<R, T>::st(hxx * m); oc[1] = Cvt<R, T>::st(hxy * m); oc[2] = Cvt<R, T>::st(hxz * m);
                oc[3] = Cvt<R, T>::st(hxy * m); oc[4] = Cvt<R, T>::st(hyy * m); oc[5] = Cvt<R, T>::st(hyz * m);
                oc[6] = Cvt<R, T>::st(hxz * m); oc[7] = Cvt<R, T>::st(hyz * m); oc[8] = Cvt<R, T>::st(hzz * m);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// push / count : vol[b,c,tap] += w * mask * val[b,c,o]             (nd.py:146-213)
//   COUNT: val == all ones, C == 1 (pushpull.py:106-142).
//   AccT is the accumulation type of the target buffer (float for f32/bf16/f16
//   storage, double for f64); the target is zero-filled by the caller side.
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, typename AccT, int D, int KMAX, bool ISO, bool COUNT, bool DEF = false>
__global__ __launch_bounds__(BLOCK) void push_generic(KParams p, const T *__restrict__ val,
                                                      const G *__restrict__ grid, AccT *__restrict__ vol, int B, TileList tl)
{
    // the owner-computes organisation took this call (push_owner.hip: own_probe); 2-D router (scatter2d.hip: probe2d), gate_n = -2: this
    // kernel takes the sparsely sampled targets only (verdict 2); gate_n = -4: the tiles of the 2-D router declined -- this kernel stands in
    // for them as well (every verdict but 1, the bricks)
    if (p.gate && (p.gate_n == -2 ? *p.gate != 2 : (p.gate_n == -4 ? *p.gate == 1 : *p.gate != 0))) return;
    IP_FOR_SAMPLES {
        const int64_t b = it_.b, o = it_.o;
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_W> s;
        s.setup(p, x);
        AccT *vb = vol + b * p.vol_sb;
        const T *ib = COUNT ? nullptr : val + b * p.val_sb + o;
        for (int c = 0; c < p.C; ++c) {
            AccT *v0 = vb + c * p.vol_sc;
            R s0 = R(1);
            if (!COUNT) s0 = Cvt<R, T>::ld(ib[c * p.val_sc]);
            s0 *= s.mask;
            opaque(s);
            IP_FOR_I {
                const R i0 = s0 * s.w[0][i];
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    const R j0 = i0 * s.w[1][j];
                    IP_FOR_K {
                        atomic_add<AccT>(v0, oij + s.off[2][k], (AccT)(j0 * s.w[2][k]));
                    } IP_ROW_END;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// pushgrad : vol[b,c,tap] += sum_d (g_d prod_{e!=d} w_e) * mask * val[b,c,o,d]   (nd.py:291-364)
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, typename AccT, int D, int KMAX, bool ISO>
__global__ __launch_bounds__(BLOCK) void pushgrad_generic(KParams p, const T *__restrict__ val,
                                                          const G *__restrict__ grid, AccT *__restrict__ vol, int B)
{
    const int64_t o = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= p.N) return;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_G> s;
        s.setup(p, x);
        AccT *vb = vol + b * p.vol_sb;
        const T *ib = val + b * p.val_sb + o * D;
        for (int c = 0; c < p.C; ++c) {
            AccT *v0 = vb + c * p.vol_sc;
            R sv0 = Cvt<R, T>::ld(ib[c * p.val_sc]) * s.mask, sv1 = R(0), sv2 = R(0);
            if (D > 1) sv1 = Cvt<R, T>::ld(ib[c * p.val_sc + 1]) * s.mask;
            if (D > 2) sv2 = Cvt<R, T>::ld(ib[c * p.val_sc + 2]) * s.mask;
            opaque(s);
            IP_FOR_I {
                const R ia = sv0 * s.g[0][i], ib1 = sv1 * s.w[0][i], ic = sv2 * s.w[0][i];
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    R jab = ia * s.w[1][j];
                    if (D > 1) jab += ib1 * s.g[1][j];
                    const R jc = ic * s.w[1][j];
                    IP_FOR_K {
                        R v = jab * s.w[2][k];
                        if (D > 2) v += jc * s.g[2][k];
                        atomic_add<AccT>(v0, oij + s.off[2][k], (AccT)v);
                    } IP_ROW_END;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Fused backward of pull (pushpull.py:237-258), one pass over the grid:
//   gvol[b,c,tap] += w * mask * gout[b,c,o]                       (= push(gout))
//   ggrid[b,o,d]   = mask * sum_c gout[b,c,o] * sum_taps (g_d prod w) vol[b,c,tap]
// replaces push + grad + a (B,C,N,D) temporary + a channel reduction.
// Either output may be NULL (requires_grad-driven skipping, pushpull.py:252-255).
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, typename AccT, int D, int KMAX, bool ISO, bool DEF = false>
__global__ __launch_bounds__(BLOCK) void pullbwd_generic(KParams p, const T *__restrict__ gout, const T *__restrict__ vol,
                                                         const G *__restrict__ grid, AccT *__restrict__ gvol,
                                                         G *__restrict__ ggrid, int B, int64_t gvol_sb, int64_t gvol_sc, TileList tl)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to another organisation (abi.hip: routed_gradc, trilinear)
    if (p.gate_n == -2 && p.gate && *p.gate != 2) return;   // 2-D router (scatter2d.hip: probe2d): the organisation of expanding fields only
    // gvol has vol's spatial layout (host guarantees both spatially contiguous) but AccT elements
    constexpr unsigned ACC_SCALE = sizeof(AccT) / sizeof(T);
    IP_FOR_SAMPLES {
        const int64_t b = it_.b, o = it_.o;
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_G> s;
        s.setup(p, x);
        const T *vb = vol + b * p.vol_sb;
        const T *gb = gout + b * p.val_sb + o;
        R gg0 = R(0), gg1 = R(0), gg2 = R(0);
        for (int c = 0; c < p.C; ++c) {
            const T *v0 = vb + c * p.vol_sc;
            AccT *q0 = gvol ? gvol + b * gvol_sb + c * gvol_sc : nullptr;
            const R go = Cvt<R, T>::ld(gb[c * p.val_sc]) * s.mask;
            opaque(s);
            R a0 = R(0), a1 = R(0), a2 = R(0);
            IP_FOR_I {
                R pWW = R(0), pGW = R(0), pWG = R(0);
                const R i0 = go * s.w[0][i];
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    const R j0 = i0 * s.w[1][j];
                    R rW = R(0), rG = R(0);
                    IP_FOR_K {
                        const unsigned off = oij + s.off[2][k];
                        if (ggrid) {
                            const R v = Cvt<R, T>::ld(ld_tap(v0, off));
                            rW = fma_(s.w[2][k], v, rW);
                            if (D > 2) rG = fma_(s.g[2][k], v, rG);
                        }
                        if (gvol) atomic_add<AccT>(q0, off * ACC_SCALE, (AccT)(j0 * s.w[2][k]));
                    } IP_ROW_END;
                    pWW = fma_(s.w[1][j], rW, pWW);
                    if (D > 1) pGW = fma_(s.g[1][j], rW, pGW);
                    if (D > 2) pWG = fma_(s.w[1][j], rG, pWG);
                }
                a0 = fma_(s.g[0][i], pWW, a0);
                if (D > 1) a1 = fma_(s.w[0][i], pGW, a1);
                if (D > 2) a2 = fma_(s.w[0][i], pWG, a2);
            }
            gg0 = fma_(a0, go, gg0); gg1 = fma_(a1, go, gg1); gg2 = fma_(a2, go, gg2);
        }
        if (ggrid) {
            G *gp = ggrid + (b * p.N + o) * D;
            gp[0] = (G)gg0;
            if (D > 1) gp[1] = (G)gg1;
            if (D > 2) gp[2] = (G)gg2;
        }
    }
}

// ---------------------------------------------------------------------------
// Fused backward of push / count (pushpull.py:262-299), one gather pass:
//   gval[b,c,o]  = mask * sum_taps w * gvol_out[b,c,tap]          (= pull(grad))
//   ggrid[b,o,d] = mask * sum_c val[b,c,o] * sum_taps (g_d prod w) gvol_out[b,c,tap]
// COUNT: val == all ones (grid_count_backward).  Either output may be NULL.
// ---------------------------------------------------------------------------
template <typename T, typename G, typename R, int D, int KMAX, bool ISO, bool COUNT, bool DEF = false>
__global__ __launch_bounds__(BLOCK) void pushbwd_generic(KParams p, const T *__restrict__ gvol_out, const T *__restrict__ val,
                                                         const G *__restrict__ grid, T *__restrict__ gval,
                                                         G *__restrict__ ggrid, int B, TileList tl)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to another organisation (abi.hip: routed_gradc, trilinear)
    if (p.gate_n == -2 && p.gate && *p.gate != 2) return;   // 2-D router (scatter2d.hip: probe2d): the organisation of expanding fields only
    IP_FOR_SAMPLES {
        const int64_t b = it_.b, o = it_.o;
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, KMAX, ISO, NEED_G> s;
        s.setup(p, x);
        const T *vb = gvol_out + b * p.vol_sb;
        R gg0 = R(0), gg1 = R(0), gg2 = R(0);
        for (int c = 0; c < p.C; ++c) {
            const T *v0 = vb + c * p.vol_sc;
            opaque(s);
            R aw = R(0), a0 = R(0), a1 = R(0), a2 = R(0);
            IP_FOR_I {
                R pWW = R(0), pGW = R(0), pWG = R(0);
                IP_FOR_J {
                    const unsigned oij = s.off[0][i] + s.off[1][j];
                    R rW = R(0), rG = R(0);
                    IP_FOR_K {
                        const R v = Cvt<R, T>::ld(ld_tap(v0, oij + s.off[2][k]));
                        rW = fma_(s.w[2][k], v, rW);
                        if (D > 2) rG = fma_(s.g[2][k], v, rG);
                    } IP_ROW_END;
                    pWW = fma_(s.w[1][j], rW, pWW);
                    if (D > 1) pGW = fma_(s.g[1][j], rW, pGW);
                    if (D > 2) pWG = fma_(s.w[1][j], rG, pWG);
                }
                aw = fma_(s.w[0][i], pWW, aw);
                a0 = fma_(s.g[0][i], pWW, a0);
                if (D > 1) a1 = fma_(s.w[0][i], pGW, a1);
                if (D > 2) a2 = fma_(s.w[0][i], pWG, a2);
            }
            if (gval) gval[b * p.val_sb + c * p.val_sc + o] = Cvt<R, T>::st(aw * s.mask);
            R sv = s.mask;
            if (!COUNT) sv *= Cvt<R, T>::ld(val[b * p.val_sb + c * p.val_sc + o]);
            gg0 = fma_(a0, sv, gg0); gg1 = fma_(a1, sv, gg1); gg2 = fma_(a2, sv, gg2);
        }
        if (ggrid) {
            G *gp = ggrid + (b * p.N + o) * D;
            gp[0] = (G)gg0;
            if (D > 1) gp[1] = (G)gg1;
            if (D > 2) gp[2] = (G)gg2;
        }
    }
}

// ---------------------------------------------------------------------------
// Elementwise helpers for the low-precision scatter path.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void narrow_kernel(const float *__restrict__ src, T *__restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        dst[i] = Cvt<float, T>::st(src[i]);
}

} // namespace ip
