// One-dimensional B-spline resampling along ONE dimension of a tensor viewed as
// (outer, n, inner): the building block of `resize` / `restrict` on tensor-product lattices
// (reference interpol/resize.py:96-117, restrict.py:88-118 build stack(meshgrid_ij(*lin), -1) and
// call the D-dimensional grid_pull / grid_push: (K+1)^D taps per voxel).  The stencil of a
// tensor-product lattice factorises, so D passes of K+1 taps give the same operator:
//     forward : dst[b, s, c]  = mask(x_s) * sum_j w_j(x_s) sign_j src[b, wrap(i0_s + j), c]
//     adjoint : dst[b, wrap(i0_s + j), c] += w_j(x_s) sign_j mask(x_s) src[b, s, c]
// with x_s = lin[s]; weights, indices, signs and mask are those of csrc/stencil.hpp
// (splines.py:30-80, bounds.py:30-89, nd.py:10-27).  One thread per (b, s, c): lanes run
// along c when inner > 1 (coalesced rows, wave-uniform stencil), along s for the last dim.
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace ip {
namespace {

template <typename T> __device__ __forceinline__ T ld_b(const T *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// grid: x = blocks over ns * inner, y = outer slices
template <typename T, typename G, typename R, int K, bool ADJ>
__global__ __launch_bounds__(256) void resample1d_kernel(KParams p, const T *__restrict__ src, const G *__restrict__ lin,
                                                         T *__restrict__ dst, unsigned ns, unsigned inner, int64_t nl, int64_t outer)
{
    const unsigned r = blockIdx.x * 256u + threadIdx.x;
    if (r >= ns * inner) return;
    const unsigned s = r / inner, c = r - s * inner;
    R x[1] = { (R)lin[s] };
    Stencil<R, 1, K, true, NEED_W> st;
    st.setup(p, x);
    for (int64_t b = blockIdx.y; b < outer; b += gridDim.y) {
        const T *sb = src + b * (ADJ ? (int64_t)ns : nl) * inner + c;
        T *db = dst + b * (ADJ ? nl : (int64_t)ns) * inner + c;
        if constexpr (!ADJ) {
            R acc = R(0);
#pragma unroll
            for (int j = 0; j <= K; ++j) acc = st.w[0][j] * Cvt<R, T>::ld(ld_b(sb, st.off[0][j])) + acc;
            db[(int64_t)s * inner] = Cvt<R, T>::st(acc * st.mask);
        } else {
            const R v = Cvt<R, T>::ld(sb[(int64_t)s * inner]) * st.mask;
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                T *q = reinterpret_cast<T *>(reinterpret_cast<char *>(db) + st.off[0][j]);
                __hip_atomic_fetch_add(q, (T)(st.w[0][j] * v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// fp32 forward pass, inner % 4 == 0: four consecutive c per thread, 16-byte loads and stores
template <int K>
__global__ __launch_bounds__(256) void resample1d_fwd_f32x4(KParams p, const float *__restrict__ src, const float *__restrict__ lin,
                                                            float *__restrict__ dst, unsigned ns, unsigned inner4, int64_t nl, int64_t outer)
{
    const unsigned r = blockIdx.x * 256u + threadIdx.x;
    if (r >= ns * inner4) return;
    const unsigned s = r / inner4, c = (r - s * inner4) * 4u;
    float x[1] = { lin[s] };
    Stencil<float, 1, K, true, NEED_W> st;
    st.setup(p, x);
    const int64_t inner = (int64_t)inner4 * 4;
    for (int64_t b = blockIdx.y; b < outer; b += gridDim.y) {
        const float *sb = src + b * nl * inner + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sb) + st.off[0][j]);
            const float w = st.w[0][j];
            acc.x = w * v.x + acc.x; acc.y = w * v.y + acc.y; acc.z = w * v.z + acc.z; acc.w = w * v.w + acc.w;
        }
        const float m = st.mask;
        *reinterpret_cast<float4 *>(dst + (b * ns + s) * inner + c) = make_float4(acc.x * m, acc.y * m, acc.z * m, acc.w * m);
    }
}

// ---------------------------------------------------------------------------
// The adjoint as a GATHER (round 5): dst[b, l, c] = sum over the samples s and taps j with wrap(i0_s + j) == l of
// w_j(x_s) sign_j mask(x_s) src[b, s, c].  `lin` of resize / restrict is non-decreasing, so the samples whose stencil covers lattice
// point l are a contiguous range found by bisection; the few samples whose stencil leaves the lattice (their taps come back through the
// boundary condition, anywhere) are visited by every output.  No atomics, no zero-fill, a deterministic sum in sample order; an
// unsorted `lin` is detected by every workgroup and served by visiting all samples (slow, correct).
// A workgroup owns LT lattice points x CT columns; the stencils of the samples it needs -- first tap, wrapped indices, signed and masked
// weights, exactly Stencil::setup's -- are built once into LDS and reused for every outer slice.
// restrict 4 x 2 x 256^3 -> 128^3 (tools/bench_configs.py f2): ONE 3-D push on the separable lattice took 0.92 (linear) / 2.27 ms (cubic);
// three scattering passes with atomics 3.0 ms (separable.py, round 2).
// ---------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ int lower_bound_lin(const R *lin, int ns, R val)          // first s with lin[s] >= val (lin: LDS copy)
{
    int lo = 0, hi = ns;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (lin[mid] < val) lo = mid + 1; else hi = mid; }
    return lo;
}

constexpr int ADJ_NS_MAX = 4096;                // samples along the dim: `lin` is copied to LDS (16 / 32 KiB)
template <typename T, typename G, typename R, int K, bool LASTDIM, int VEC>
__global__ __launch_bounds__(256) void resample1d_adj_gather(KParams p, const T *__restrict__ src, const G *__restrict__ lin_g, T *__restrict__ dst,
                                                             int ns, unsigned inner, int nl, int64_t outer, int cgroups, int cspan)
{
    constexpr int AW = LASTDIM ? 384 : 64;      // stencils resident per round (LASTDIM: 128 lattice points of a restriction by 2 need ~270)
    constexpr int LT = LASTDIM ? 128 : 4, CT = LASTDIM ? 1 : 64 * VEC;
    constexpr int MAXC = LASTDIM ? 24 : 64, LCOL = LASTDIM ? 128 : 4;
    static_assert(VEC == 1 || (!LASTDIM && sizeof(T) == 4), "quads of float columns");
    __shared__ int lcount[LCOL], lrest[LCOL];
    __shared__ R   lin[ADJ_NS_MAX];
    __shared__ int lst_s[MAXC][LCOL];           // the (sample, weight) pairs per lattice point of the tile
    __shared__ R   lst_w[MAXC][LCOL];
    __shared__ int e_s[AW], e_i0[AW];
    __shared__ int e_idx[AW][K + 1];
    __shared__ R   e_w[AW][K + 1];
    __shared__ int s_range[4];                  // main window [w0, w1), edge sets [0, elo) and [ehi, ns)
    const int tid = threadIdx.x;
    const int lt = blockIdx.x / cgroups, cg = blockIdx.x - lt * cgroups;
    const int l0 = lt * LT;
    const int l = LASTDIM ? l0 + (tid & (LT - 1)) : l0 + (tid >> 6);
    const bool lrow = l < nl;                   // (LASTDIM: the two halves of the workgroup hold the same lattice points, alternate slices)
    const R h = R(0.5) * R(K - 1);
    // ---- `lin` into LDS; is it non-decreasing?  (every workgroup looks)
    bool ok = true;
    for (int i = tid; i < ns; i += 256) lin[i] = (R)lin_g[i];
    __syncthreads();
    for (int i = tid; i + 1 < ns; i += 256) ok = ok && (lin[i] <= lin[i + 1]);
    const bool sorted = __syncthreads_and(ok ? 1 : 0) != 0;
    if (tid < 4) {
        int v = tid == 1 || tid == 3 ? ns : 0;
        if (sorted) {
            if (tid == 0) v = lower_bound_lin<R>(lin, ns, (R)(l0 - K - 1) + h);
            if (tid == 1) v = lower_bound_lin<R>(lin, ns, (R)(l0 + LT + 1) + h);
            if (tid == 2) v = lower_bound_lin<R>(lin, ns, h + (p.bound[0] == B_DST1 ? R(2) : R(1)));      // samples whose first tap may lie below the lattice (+1 slack)
            if (tid == 3) v = lower_bound_lin<R>(lin, ns, (R)(nl - K - 1) + h);                            // ... whose last tap may lie beyond it
        } else if (tid >= 2) v = tid == 2 ? 0 : ns;                  // (unsorted: one pass over [0, ns), every tap counted)
        s_range[tid] = v;
    }
    __syncthreads();
    const int w0 = s_range[0], w1 = s_range[1], elo = s_range[2], ehi = s_range[3] < s_range[2] ? s_range[2] : s_range[3];
    const int nm = w1 - w0, ne = sorted ? elo + (ns - ehi) : 0, total = nm + ne;
    // the lane's own samples inside the main window (LASTDIM: lanes hold different lattice points)
    int my0 = 0, my1 = nm;
    if (LASTDIM && sorted && lrow) {
        my0 = lower_bound_lin<R>(lin, ns, (R)(l - K - 1) + h) - w0;
        my1 = lower_bound_lin<R>(lin, ns, (R)(l + 2) + h) - w0;
        my0 = my0 > 0 ? my0 : 0; my1 = my1 < nm ? my1 : nm;
    }
    // virtual list of stencils: the main window (taps INSIDE the lattice count; unsorted: every tap), then the edge samples (taps OUTSIDE
    // count), in rounds of AW -- one round in practice (restrict by 2, cubic: 18 + 4 stencils)
    bool first = true;
    const int rounds = total > 0 ? (total + AW - 1) / AW : 1;
    for (int rd = 0; rd < rounds; ++rd) {
        const int v0 = rd * AW, cnt = total - v0 < AW ? total - v0 : AW;
        __syncthreads();                        // the previous round's readers are done
        for (int e = tid; e < cnt; e += 256) {
            const int v = v0 + e;
            const int sidx = v < nm ? w0 + v : (v - nm < elo ? v - nm : ehi + (v - nm - elo));
            R x[1] = { lin[sidx] };
            typedef Stencil<R, 1, K, true, NEED_W> St;
            St st;
            st.setup(p, x);
            R fl;
            if (K == 0 && p.mode == MODE_ISO0) fl = St::rint_(x[0]); else fl = St::floor_(x[0] - R(0.5) * R(K - 1));      // (Stencil::setup)
            fl = fl < R(-1073741824) ? R(-1073741824) : (fl > R(1073741824) ? R(1073741824) : fl);
            e_s[e] = sidx; e_i0[e] = (int)fl;
#pragma unroll
            for (int j = 0; j <= K; ++j) { e_idx[e][j] = (int)(st.off[0][j] / (unsigned)p.vol_ss[0]); e_w[e][j] = st.w[0][j] * st.mask; }
        }
        __syncthreads();
        // entries of this round: [0, nmr) main, [nmr, cnt) edge
        const int nmr = nm - v0 < 0 ? 0 : (nm - v0 < cnt ? nm - v0 : cnt);
        const int kmain = sorted ? 0 : 2;
        // weight of entry e on lattice point l (0 and false: none of its taps lands there in this entry's role)
        auto weigh = [&](int e, int kind, R &wsel) -> bool {
            bool any = false;
            wsel = R(0);
            const int i0 = e_i0[e];
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                const bool inside = (unsigned)(i0 + j) < (unsigned)nl;
                const bool m = e_idx[e][j] == l && (kind == 2 || (kind == 0 ? inside : !inside));
                wsel += m ? e_w[e][j] : R(0);
                any = any || m;
            }
            return any;
        };
        // ---- the (sample, weight) pairs of every lattice point of the tile, once per round: the loops over columns and slices below
        // are K + 1 loads and FMAs per output, like the forward pass
        int nlist = 0, rest = cnt;              // LASTDIM: pairs beyond the list's MAXC are recomputed per slice from entry `rest` on
        int a0 = 0, a1 = nmr;                   // the lane's part of the main entries
        if (LASTDIM && sorted) { a0 = my0 - v0 > 0 ? my0 - v0 : 0; a1 = my1 - v0 < nmr ? my1 - v0 : nmr; }
        if constexpr (!LASTDIM) {
            const int lane = tid & 63, wv = tid >> 6;
            static_assert(AW == 64, "one entry per lane");
            R wsel = R(0);
            const bool any = lane < cnt && lrow && weigh(lane, lane < nmr ? kmain : 1, wsel);
            const unsigned long long mask = __ballot(any);
            if (any) { const int pos = __popcll(mask & ((1ull << lane) - 1ull)); lst_s[pos][wv] = e_s[lane]; lst_w[pos][wv] = wsel; }
            nlist = __popcll(mask);
        } else if (lrow && tid < LT) {
            for (int part = 0; part < 2 && rest == cnt; ++part) {
                const int eb = part == 0 ? a0 : nmr, ee = part == 0 ? a1 : cnt;
                for (int e = eb; e < ee; ++e) {
                    R wsel;
                    if (!weigh(e, part == 0 ? kmain : 1, wsel)) continue;
                    if (nlist == MAXC) { rest = e; break; }
                    lst_s[nlist][tid] = e_s[e]; lst_w[nlist][tid] = wsel; ++nlist;
                }
            }
        }
        if (LASTDIM ? tid < LT : (tid & 63) == 0) { lcount[LASTDIM ? tid : (tid >> 6)] = nlist; lrest[LASTDIM ? tid : (tid >> 6)] = rest; }
        __syncthreads();
        const int col = LASTDIM ? (tid & (LT - 1)) : (tid >> 6);
        nlist = lcount[col]; rest = lrest[col];
        // slices: LASTDIM -- the halves of the workgroup take alternate slices; UB slices in flight per thread
        constexpr int UB = 4;
        // a workgroup takes a CONTIGUOUS run of slices (strided slices sent every access of the last-dim pass to another page: 0.30 ms
        // for 134 MB), the halves of a LASTDIM workgroup alternate inside it
        const int64_t chunk = (outer + gridDim.y - 1) / gridDim.y;
        const int64_t bend = ((int64_t)blockIdx.y + 1) * chunk < outer ? ((int64_t)blockIdx.y + 1) * chunk : outer;
        const int64_t bfirst = (int64_t)blockIdx.y * chunk + (LASTDIM ? (tid >> 7) : 0);
        const int64_t bstep = LASTDIM ? 2 : 1;
        for (int ci = 0; ci < cspan; ++ci) {
            const unsigned c = LASTDIM ? 0u : ((unsigned)cg * cspan + ci) * CT + (tid & 63) * VEC;
            if (!(lrow && c < inner)) continue;
            for (int64_t b0 = bfirst; b0 < bend; b0 += bstep * UB) {
                R acc[UB][VEC];
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[u][q] = R(0);
                for (int k = 0; k < nlist; ++k) {
                    const R w = lst_w[k][col];
                    const int64_t so = (int64_t)lst_s[k][col] * inner + c;
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int64_t b = b0 + u * bstep;
                        if (b >= bend) continue;
                        const T *sp = src + b * (int64_t)ns * inner + so;
                        if constexpr (VEC == 4) {
                            const float4 v = *reinterpret_cast<const float4 *>(sp);
                            acc[u][0] = w * v.x + acc[u][0]; acc[u][1] = w * v.y + acc[u][1]; acc[u][2] = w * v.z + acc[u][2]; acc[u][3] = w * v.w + acc[u][3];
                        } else acc[u][0] = w * Cvt<R, T>::ld(*sp) + acc[u][0];
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int64_t b = b0 + u * bstep;
                    if (b >= bend) continue;
                    if (LASTDIM) {
                        const T *sb = src + b * (int64_t)ns * inner + c;
                        for (int e = rest; e < cnt; ++e) {            // (a lattice point fed by more than MAXC samples: a restriction by 8 and more)
                            if (e < nmr && (e < a0 || e >= a1)) continue;
                            R wsel;
                            if (weigh(e, e < nmr ? kmain : 1, wsel)) acc[u][0] = wsel * Cvt<R, T>::ld(sb[(int64_t)e_s[e] * inner]) + acc[u][0];
                        }
                    }
                    T *q = dst + (b * (int64_t)nl + l) * inner + c;
                    if constexpr (VEC == 4) {
                        float4 o = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
                        if (!first) { const float4 old = *reinterpret_cast<const float4 *>(q); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                        *reinterpret_cast<float4 *>(q) = o;
                    } else *q = first ? Cvt<R, T>::st(acc[u][0]) : Cvt<R, T>::st(Cvt<R, T>::ld(*q) + acc[u][0]);
                }
            }
        }
        first = false;
    }
}

// the gather serves the adjoint when the lanes can run along a long inner dim or along the lattice itself
static inline bool adjoint_gathers(unsigned ns, unsigned inner) { return ns <= (unsigned)ADJ_NS_MAX && (inner == 1u || inner >= 64u); }

template <typename T, typename G, typename R>
int launch_adj_gather(int order, const KParams &p, const void *src, const void *lin, void *dst, unsigned ns, unsigned inner, int64_t nl,
                      int64_t outer, hipStream_t st)
{
    const bool last = inner == 1u;
    const bool vec = !last && std::is_same<T, float>::value && inner % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const int LT = last ? 128 : 4, CT = last ? 1 : (vec ? 256 : 64);
    const int ctiles = (int)((inner + CT - 1) / CT);
    const int cspan = ctiles < 8 ? ctiles : 8;                       // column tiles a workgroup walks with one set of stencils
    const int cgroups = (ctiles + cspan - 1) / cspan;
    const int64_t blocks = ((nl + LT - 1) / LT) * cgroups;
    if (blocks > 0x7fffffffll) return INTERPOL_E_SHAPE;
    const int64_t slices = last ? (outer + 7) / 8 : outer;           // (LASTDIM: a workgroup takes eight slices per step)
    // enough slices side by side to fill the chip, the rest in the loop (the stencils of a workgroup are built once: LASTDIM's per-lane
    // lists cost as much as ~30 slices)
    int64_t ny = (int64_t)(last ? 1u << 17 : 1u << 19) / (blocks * 256) + 1;
    ny = ny < slices ? ny : slices;
    ny = ny < 65535 ? ny : 65535;
    const dim3 grid((unsigned)blocks, (unsigned)ny);
#define IP_AGL(KK, LD, VV) hipLaunchKernelGGL((resample1d_adj_gather<T, G, R, KK, LD, VV>), grid, dim3(256), 0, st, p, (const T *)src, (const G *)lin, \
                                             (T *)dst, (int)ns, inner, (int)nl, outer, cgroups, cspan)
#define IP_AG(KK) case KK:                                                                                          \
        if (last) IP_AGL(KK, true, 1);                                                                              \
        else if constexpr (std::is_same<T, float>::value) { if (vec) IP_AGL(KK, false, 4); else IP_AGL(KK, false, 1); } \
        else IP_AGL(KK, false, 1);                                                                                  \
        break;
    switch (order) { IP_AG(0) IP_AG(1) IP_AG(2) IP_AG(3) IP_AG(4) IP_AG(5) IP_AG(6) IP_AG(7) default: return INTERPOL_E_ORDER; }
#undef IP_AG
#undef IP_AGL
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <typename T, typename G, typename R, bool ADJ>
int launch_k(int order, const KParams &p, const void *src, const void *lin, void *dst, unsigned ns, unsigned inner, int64_t nl,
             int64_t outer, hipStream_t st)
{
    constexpr bool VEC_OK = !ADJ && std::is_same<T, float>::value && std::is_same<G, float>::value;
    const bool vec = VEC_OK && inner % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const unsigned cols = vec ? inner / 4 : inner;
    const unsigned blocks = (unsigned)(((uint64_t)ns * cols + 255) / 256);
    // The stencil of a thread depends on s only: every thread walks several outer slices with
    // the same weights / offsets (its set-up costs ~100 instructions, a tap 2).  Enough slices
    // in parallel to fill the chip (~16 K threads per CU), the rest in the loop.
    int64_t ny = (int64_t)(4u << 20) / ((int64_t)blocks * 256) + 1;
    ny = ny < outer ? ny : outer;
    ny = ny < 65535 ? ny : 65535;
    const dim3 grid(blocks, (unsigned)ny);
    if constexpr (VEC_OK) {
        if (vec) {
#define IP_R4(KK) case KK: hipLaunchKernelGGL((resample1d_fwd_f32x4<KK>), grid, dim3(256), 0, st, p, (const float *)src, \
                                              (const float *)lin, (float *)dst, ns, cols, nl, outer); break;
            switch (order) { IP_R4(0) IP_R4(1) IP_R4(2) IP_R4(3) IP_R4(4) IP_R4(5) IP_R4(6) IP_R4(7) default: return INTERPOL_E_ORDER; }
#undef IP_R4
            const hipError_t e4 = hipGetLastError();
            return e4 == hipSuccess ? 0 : (int)e4;
        }
    }
#define IP_R1(KK) case KK: hipLaunchKernelGGL((resample1d_kernel<T, G, R, KK, ADJ>), grid, dim3(256), 0, st, p, (const T *)src, \
                                              (const G *)lin, (T *)dst, ns, inner, nl, outer); break;
    switch (order) { IP_R1(0) IP_R1(1) IP_R1(2) IP_R1(3) IP_R1(4) IP_R1(5) IP_R1(6) IP_R1(7) default: return INTERPOL_E_ORDER; }
#undef IP_R1
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace

// Called by interpol_resample_1d (abi.hip).  16-bit storage: forward only.
int launch_resample1d(int dtype, int lin_f64, int order, const KParams &p, int adjoint, const void *src, const void *lin, void *dst,
                      unsigned ns, unsigned inner, int64_t nl, int64_t outer, hipStream_t st)
{
    if (dtype == INTERPOL_F64) {
        if (!lin_f64) return INTERPOL_E_DTYPE;
        if (adjoint && adjoint_gathers(ns, inner)) return launch_adj_gather<double, double, double>(order, p, src, lin, dst, ns, inner, nl, outer, st);
        return adjoint ? launch_k<double, double, double, true>(order, p, src, lin, dst, ns, inner, nl, outer, st)
                       : launch_k<double, double, double, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    }
    if (lin_f64) return INTERPOL_E_DTYPE;
    if (dtype == INTERPOL_F32 && adjoint && adjoint_gathers(ns, inner))
        return launch_adj_gather<float, float, float>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    if (dtype == INTERPOL_F32)
        return adjoint ? launch_k<float, float, float, true>(order, p, src, lin, dst, ns, inner, nl, outer, st)
                       : launch_k<float, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    if (adjoint) return INTERPOL_E_DTYPE;
    if (dtype == INTERPOL_BF16) return launch_k<bf16_t, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    if (dtype == INTERPOL_F16) return launch_k<f16_t, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    return INTERPOL_E_DTYPE;
}

// (abi.hip: the scattering adjoint needs its target zero-filled, the gathering one writes every element)
bool resample1d_adjoint_gathers(int64_t n_samples, int64_t inner) { return n_samples <= ADJ_NS_MAX && (inner == 1 || inner >= 64); }

} // namespace ip
