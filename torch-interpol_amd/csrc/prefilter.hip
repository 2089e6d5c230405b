// ===========================================================================
// prefilter.hip -- interpolating-coefficient prefilter (recursive IIR), in place.
//
// Numerical definition = reference interpol/coeff.py:258-284 (filter), with the
// boundary-specific initial/final values of coeff.py:82-227 and the bound
// mapping of coeff.py:231-254 (zero -> dct1, replicate -> dct2).  The reference
// runs the recursion as a serial Python loop of n tiny tensor ops per pole.
//
// Layout: contiguous (outer, n, inner), filter along the middle axis.  Two kernels:
//
//  * inner == 1 (the lines are contiguous): ONE WAVE PER LINE, the whole line in
//    registers (lane l owns R consecutive samples), every pole processed on chip:
//    the first-order recurrences c[i] = a[i] + p c[i-1] (causal) and
//    d[i] = p (d[i+1] - c[i]) (anticausal) are linear, so each is a serial pass over
//    the lane's R samples + a 6-step wave scan of the lane carries with multipliers
//    p^(R 2^s) + a fix-up pass.  Traffic: one read and one write of the line
//    (the 2 n s algorithmic bytes), whatever the number of poles.  n <= 64 * 32.
//
//  * inner > 1 (lines interleaved, stride = inner): ONE THREAD PER LINE, consecutive
//    threads = consecutive lines, so every access is coalesced across the wave; the
//    line is streamed in register chunks of CH samples (all CH loads issued before
//    the serial recurrence) so that loads overlap the dependent arithmetic.
//    Traffic per pole: 2 reads + 2 writes of the line.
//
// Both share the boundary formulas below (one source of truth with the oracle).
// ===========================================================================
#include "stencil.hpp"
#include "filter_params.hpp"
#include <math.h>

namespace ip {

// pole^e for a small non-negative integer e (exact repeated squaring in R)
template <typename R>
__device__ __forceinline__ R powi(R base, int64_t e)
{
    R r = R(1);
    while (e > 0) { if (e & 1) r *= base; base *= base; e >>= 1; }
    return r;
}

// ---------------------------------------------------------------------------
// Initial value of the causal recursion as a weighted sum  init = A * sum_i w(i) c[i] + B * c[0]
// and final value of the anticausal one; `W` describes the weights so that both kernels can
// evaluate the sum in their own way (serial / wave-parallel).
//   dct1 (coeff.py:109-149):  n > max_iter:  w(i) = pf^i, i < max_iter
//                             else:          init = (c[0] + pn c[n-1] + sum_{0<i<n-1} (pf^i + pn^2/pf^i) c[i]) / (1 - pn^2),  pn = pole^(n-1)
//   dct2 (coeff.py:153-179):  init = pole/(1-pn^2) * sum_i (pf^i + pn pf^(n-1-i)) c[i] + c[0],  pn = pole^n
//   dft  (coeff.py:82-105):   init = (c[0] + sum_{j=1}^{m-1} pf^j c[n-j]) / (1 - pole^m),  m = min(max_iter, n)
// pf = float(pole): the reference builds its tensor of pole powers from the pole rounded
// through float32 (TorchScript as_tensor quirk, see oracle/interpol_oracle_body.inc).
// ---------------------------------------------------------------------------
template <typename R>
struct InitW {
    int kind;            // 0 dct1-truncated, 1 dct1-full, 2 dct2, 3 dft
    int64_t n, m;        // m: number of leading (kind 0, 2) or trailing (kind 3) terms that matter
    R pf, ipf, pn, pn2;
    R scale, c0w;        // init = scale * sum + c0w * c[0]

    // constants from the host (filter_params.hpp: make_pole_pre)
    __device__ __forceinline__ void load(const PolePre &q, int64_t n_)
    {
        n = n_; kind = q.kind; m = q.m;
        pf = (R)q.pf; ipf = R(1) / pf; pn = (R)q.pn; pn2 = (R)q.pn2;
        scale = (R)q.scale; c0w = (R)q.c0w;
    }
    // does index i contribute, and with which weight?  (serial callers walk i upwards)
    __device__ __forceinline__ bool on(int64_t i) const
    {
        if (kind == 3) return i == 0 || i > n - m;
        return i < m;
    }
    __device__ __forceinline__ R w(int64_t i) const
    {
        switch (kind) {
        case 0: return powi(pf, i);
        case 1: { if (i == 0) return R(1); if (i == n - 1) return pn; const R pw = powi(pf, i); return pw + pn2 / pw; }
        case 2: { R wv = powi(pf, i); if (m == n) wv += pn * powi(pf, n - 1 - i); return wv; }
        default: return i == 0 ? R(1) : powi(pf, n - i);
        }
    }
};

// final value (coeff.py:183-227) from the causal result: fin = fs * (sum of a few terms)
//   dct1: (p c[n-2] + c[n-1]) * pole/(pole^2-1);  dct2: c[n-1] * pole/(pole-1)
//   dft : (p c[n-1] + sum_{i<m-1} pf^(i+2) c[i]) / (pole^m - 1)

// ===========================================================================
// Kernel A: inner > 1, one thread per line, chunked streaming.
// ===========================================================================
template <typename T, typename R, int CH>
__global__ __launch_bounds__(256) void prefilter_strided(FilterParams fp, T *data)
{
    const int64_t line = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= fp.outer * fp.inner) return;
    const int64_t a0 = line / fp.inner, b0 = line - a0 * fp.inner;
    T *base = data + a0 * fp.n * fp.inner + b0;
    const int64_t st = fp.inner, n = fp.n;
    R gain = (R)fp.gain;                               // folded into the first pole's passes (coeff.py:268)
    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        const R p = (R)pole;
        InitW<R> iw; iw.load(fp.pre[ip], n);
        // ---- initial value ----
        R sum = R(0);
        if (iw.kind == 3) {
            for (int64_t j = 1; j < iw.m; ++j) sum += Cvt<R, T>::ld(base[(n - j) * st]) * powi(iw.pf, j);
            sum += Cvt<R, T>::ld(base[0]);
        } else {
            R pw = R(1), pwr = iw.kind == 2 && iw.m == n ? powi(iw.pf, n - 1) : R(0);
            for (int64_t i = 0; i < iw.m; ++i) {
                R wv;
                if (iw.kind == 0) wv = pw;
                else if (iw.kind == 1) wv = (i == 0) ? R(1) : (i == n - 1 ? iw.pn : pw + iw.pn2 / pw);
                else wv = pw + iw.pn * pwr;
                sum += Cvt<R, T>::ld(base[i * st]) * wv;
                pw *= iw.pf; pwr *= iw.ipf;
            }
        }
        const R c0 = Cvt<R, T>::ld(base[0]) * gain;
        R prev = iw.scale * (sum * gain) + iw.c0w * c0;
        // ---- causal pass: c[i] = gain*a[i] + p c[i-1]  (coeff.py:275-276), chunked ----
        base[0] = Cvt<R, T>::st(prev);
        R last2 = prev;                                 // c[n-2] for dct1_final
        for (int64_t i0 = 1; i0 < n; i0 += CH) {
            R v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = (i0 + u < n) ? Cvt<R, T>::ld(base[(i0 + u) * st]) : R(0);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (i0 + u < n) { last2 = prev; prev = v[u] * gain + p * prev; v[u] = prev; }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) if (i0 + u < n) base[(i0 + u) * st] = Cvt<R, T>::st(v[u]);
        }
        // ---- final value ----
        R fin;
        if (fp.bound == 0) fin = (p * last2 + prev) * (R)fp.pre[ip].fin_mul;
        else if (fp.bound == 1) fin = prev * (R)fp.pre[ip].fin_mul;
        else {
            R dot = R(0), pw = iw.pf * iw.pf;
            for (int64_t i = 0; i < iw.m - 1; ++i) { dot += Cvt<R, T>::ld(base[i * st]) * pw; pw *= iw.pf; }
            dot += p * prev;
            fin = dot * (R)fp.pre[ip].fin_mul;
        }
        // ---- anticausal pass: d[i] = p (d[i+1] - c[i])  (coeff.py:280-281), chunked ----
        base[(n - 1) * st] = Cvt<R, T>::st(fin);
        R next = fin;
        for (int64_t i0 = n - 2; i0 >= 0; i0 -= CH) {
            R v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = (i0 - u >= 0) ? Cvt<R, T>::ld(base[(i0 - u) * st]) : R(0);
#pragma unroll
            for (int u = 0; u < CH; ++u) if (i0 - u >= 0) { next = (next - v[u]) * p; v[u] = next; }
#pragma unroll
            for (int u = 0; u < CH; ++u) if (i0 - u >= 0) base[(i0 - u) * st] = Cvt<R, T>::st(v[u]);
        }
        gain = R(1);
    }
}

// ===========================================================================
// Kernel B: inner == 1, one wave per line, the line in registers.
// ===========================================================================
__device__ __forceinline__ float  shfl_up_(float v, int d)  { return __shfl_up(v, d); }
__device__ __forceinline__ double shfl_up_(double v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ float  shfl_dn_(float v, int d)  { return __shfl_down(v, d); }
__device__ __forceinline__ double shfl_dn_(double v, int d) { return __shfl_down(v, d); }
__device__ __forceinline__ float  shfl_(float v, int l)  { return __shfl(v, l); }
__device__ __forceinline__ double shfl_(double v, int l) { return __shfl(v, l); }
template <typename R> __device__ __forceinline__ R wave_sum_r(R v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_(v, (int)((threadIdx.x & 63) ^ o));
    return v;
}

template <typename T, typename R, int RPL>
__global__ __launch_bounds__(256) void prefilter_wave(FilterParams fp, T *data)
{
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= fp.outer) return;
    const int64_t n = fp.n;
    T *base = data + line * n;
    const int64_t i0 = (int64_t)lane * RPL;             // lane owns samples [i0, i0 + RPL)
    R c[RPL];
    // The lane-blocked ownership is what the recursion wants, but read that way every load
    // instruction would touch 64 different 64-byte segments.  Global accesses are therefore
    // coalesced (point r * 64 + lane) and the wave's line is turned round through LDS, rows padded
    // by one word (lane stride RPL + 1: conflict-free).
    constexpr bool VIA_LDS = sizeof(R) == 4;
    __shared__ float tr[VIA_LDS ? 4 * 64 * (RPL + 1) : 1];
    float *trw = tr + (VIA_LDS ? (threadIdx.x >> 6) * 64 * (RPL + 1) : 0);
    if constexpr (VIA_LDS) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = r * 64 + lane;
            trw[i + i / RPL] = (i < n) ? (float)(Cvt<R, T>::ld(base[i]) * (R)fp.gain) : 0.f;   // coeff.py:268
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) c[r] = (R)trw[lane * (RPL + 1) + r];
    } else {
#pragma unroll
        for (int r = 0; r < RPL; ++r) c[r] = (i0 + r < n) ? Cvt<R, T>::ld(base[i0 + r]) * (R)fp.gain : R(0);   // coeff.py:268
    }
    const int last_lane = (int)((n - 1) / RPL), last_r = (int)((n - 1) % RPL);

    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        const R p = (R)pole;
        InitW<R> iw; iw.load(fp.pre[ip], n);
        // ---- initial value: wave-parallel weighted sum ----
        R part = R(0);
        {
            bool any = false;
#pragma unroll
            for (int r = 0; r < RPL; ++r) any = any || (i0 + r < n && iw.on(i0 + r));
            if (any) {
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int64_t i = i0 + r;
                    if (i < n && iw.on(i)) part += c[r] * iw.w(i);
                }
            }
        }
        const R sum = wave_sum_r(part);
        const R c_first = shfl_(c[0], 0);
        const R init = iw.scale * sum + iw.c0w * c_first;
        if (lane == 0) c[0] = init;
        // ---- causal: serial inside the lane, scan of the lane carries, fix-up ----
        // local pass with zero carry-in (sample 0 of lane 0 is the given initial value)
        R run = R(0);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            if (lane == 0 && r == 0) run = c[0];
            else run = c[r] + p * run;
            c[r] = run;
        }
        // carries: carry[l] = value entering lane l = inclusive scan of lane totals with multiplier p^RPL
        R tot = c[RPL - 1];
        R mul = powi(p, RPL);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const R up = shfl_up_(tot, d);
            if (lane >= d) tot += mul * up;
            mul *= mul;
        }
        R carry = shfl_up_(tot, 1);                     // full causal value of the last sample of lane-1
        if (lane == 0) carry = R(0);
        {
            R pw = p;
#pragma unroll
            for (int r = 0; r < RPL; ++r) { c[r] += pw * carry; pw *= p; }
        }
        // ---- final value ----
        R c_last = R(0), c_last2 = R(0);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            if (r == last_r) c_last = c[r];
            if (r == (last_r + RPL - 1) % RPL) c_last2 = c[r];
        }
        c_last = shfl_(c_last, last_lane);
        c_last2 = shfl_(c_last2, last_r == 0 ? (last_lane > 0 ? last_lane - 1 : 0) : last_lane);
        R fin;
        if (fp.bound == 0) fin = (p * c_last2 + c_last) * (R)fp.pre[ip].fin_mul;
        else if (fp.bound == 1) fin = c_last * (R)fp.pre[ip].fin_mul;
        else {
            R dpart = R(0);
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int64_t i = i0 + r;
                if (i < iw.m - 1) dpart += c[r] * powi(iw.pf, i + 2);
            }
            const R dot = wave_sum_r(dpart) + p * c_last;
            fin = dot * (R)fp.pre[ip].fin_mul;
        }
        // ---- anticausal: d[i] = p (d[i+1] - c[i]);  d[n-1] = fin ----
        // write as d[i] = a[i] + p d[i+1] with a[i] = -p c[i]; samples >= n contribute nothing
        R runb = R(0);
#pragma unroll
        for (int r = RPL - 1; r >= 0; --r) {
            const int64_t i = i0 + r;
            if (i > n - 1) { c[r] = R(0); continue; }
            if (i == n - 1) runb = fin;
            else runb = -p * c[r] + p * runb;
            c[r] = runb;
        }
        R totb = c[0];                                  // value leaving the lane towards lane-1
        R mulb = powi(p, RPL);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const R dn = shfl_dn_(totb, d);
            if (lane + d < 64) totb += mulb * dn;
            mulb *= mulb;
        }
        R carryb = shfl_dn_(totb, 1);                   // full anticausal value of sample 0 of lane+1
        if (lane >= last_lane) carryb = R(0);
        {
            R pw = p;
#pragma unroll
            for (int r = RPL - 1; r >= 0; --r) {
                const int64_t i = i0 + r;
                if (i < n - 1) c[r] += pw * carryb;
                if (i <= n - 1) pw *= p;
            }
        }
    }
    if constexpr (VIA_LDS) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) trw[lane * (RPL + 1) + r] = (float)c[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = r * 64 + lane;
            if (i < n) base[i] = Cvt<R, T>::st((R)trw[i + i / RPL]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPL; ++r) if (i0 + r < n) base[i0 + r] = Cvt<R, T>::st(c[r]);
    }
}

// ===========================================================================
// Kernel B', the common case of kernel B without its generality: the line fills the wave
// exactly (n == 64 * RPL) and the initial value is a short leading sum (dct1 with n > max_iter,
// dct2) -- no per-element range tests, pole powers by running products.  Same arithmetic.
// ===========================================================================
// the line (64 * RPL points, RPL consecutive ones per lane) filtered in registers
template <typename R, int RPL>
__device__ __forceinline__ void line_filter_full(R (&c)[RPL], const FilterParams &fp, int lane)
{
    constexpr int n = 64 * RPL;
    const int i0 = lane * RPL;
    for (int ip = 0; ip < fp.npoles; ++ip) {
        const PolePre &q = fp.pre[ip];
        const R p = (R)fp.pole[ip], pf = (R)q.pf;
        // ---- initial value: leading sum over i < m (kind 0: pf^i; kind 2: pf^i + pn pf^(n-1-i) when m == n) ----
        R part = R(0);
        if (i0 < q.m) {
            R pw = powi(pf, (int64_t)i0);
            const bool mirror = q.kind == 2 && q.m == n;
            R pwr = mirror ? powi(pf, (int64_t)(n - 1 - i0)) : R(0);
            const R ipf = R(1) / pf, pn = (R)q.pn;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                if (i0 + r < q.m) part += c[r] * (mirror ? pw + pn * pwr : pw);
                pw *= pf; pwr *= ipf;
            }
        }
        const R sum = wave_sum_r(part);
        const R c_first = shfl_(c[0], 0);
        const R init = (R)q.scale * sum + (R)q.c0w * c_first;
        // ---- causal: serial inside the lane, scan of the lane carries, fix-up ----
        R run = lane == 0 ? init : c[0];
        c[0] = run;
#pragma unroll
        for (int r = 1; r < RPL; ++r) { run = c[r] + p * run; c[r] = run; }
        R tot = c[RPL - 1];
        R mul = powi(p, (int64_t)RPL);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const R up = shfl_up_(tot, d);
            if (lane >= d) tot += mul * up;
            mul *= mul;
        }
        R carry = shfl_up_(tot, 1);
        if (lane == 0) carry = R(0);
        {
            R pw = p;
#pragma unroll
            for (int r = 0; r < RPL; ++r) { c[r] += pw * carry; pw *= p; }
        }
        // ---- final value (coeff.py:183-227): dct1 (p c[n-2] + c[n-1]) f, dct2 c[n-1] f ----
        const R c_last = shfl_(c[RPL - 1], 63);
        const R c_last2 = shfl_(c[RPL > 1 ? RPL - 2 : 0], 63);
        const R fin = (fp.bound == 0 ? p * c_last2 + c_last : c_last) * (R)q.fin_mul;
        // ---- anticausal: d[i] = p (d[i+1] - c[i]);  d[n-1] = fin ----
        R runb = lane == 63 ? fin : -p * c[RPL - 1];
        c[RPL - 1] = runb;
#pragma unroll
        for (int r = RPL - 2; r >= 0; --r) { runb = -p * c[r] + p * runb; c[r] = runb; }
        R totb = c[0];
        R mulb = powi(p, (int64_t)RPL);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const R dn = shfl_dn_(totb, d);
            if (lane + d < 64) totb += mulb * dn;
            mulb *= mulb;
        }
        R carryb = shfl_dn_(totb, 1);
        if (lane == 63) carryb = R(0);
        {
            R pw = p;
#pragma unroll
            for (int r = RPL - 1; r >= 0; --r) {
                if (!(lane == 63 && r == RPL - 1)) c[r] += pw * carryb;
                pw *= p;
            }
        }
    }
}

template <typename T, typename R, int RPL>
__global__ __launch_bounds__(256) void prefilter_wave_full(FilterParams fp, const T *src, T *data)
{
    static_assert(sizeof(R) == 4, "float math only");
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= fp.outer) return;
    constexpr int n = 64 * RPL;
    T *base = data + line * n;
    const T *from = src + line * n;                 // == base for the in-place call
    R c[RPL];
    constexpr int BYTES = RPL * (int)sizeof(T);
    if constexpr (BYTES > 32) {
        // long lane runs: element-wise coalesced accesses, transposed through LDS (a lane reading 64 contiguous
        // bytes of its own touches a cache line per lane and instruction: measured 0.157 -> 0.185 ms at 1024 fp32)
        __shared__ float tr[4 * 64 * (RPL + 1)];
        float *trw = tr + (threadIdx.x >> 6) * 64 * (RPL + 1);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = r * 64 + lane;
            trw[i + i / RPL] = Cvt<R, T>::ld(from[i]) * (R)fp.gain;                     // coeff.py:268
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) c[r] = trw[lane * (RPL + 1) + r];
        line_filter_full<R, RPL>(c, fp, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) trw[lane * (RPL + 1) + r] = c[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = r * 64 + lane;
            base[i] = Cvt<R, T>::st(trw[i + i / RPL]);
        }
        return;
    }
    // lane l owns the RPL consecutive samples l * RPL ...: up to 32 bytes, they move as 8- / 16-byte pieces of
    // the lane's own contiguous run (wide per-lane accesses, no transposition through LDS: 0.133 -> 0.10 ms at
    // 1024 bf16)
    constexpr int PIECE = BYTES % 16 == 0 ? 16 : 8, NP = BYTES / PIECE, EPP = PIECE / (int)sizeof(T);
    static_assert(BYTES % PIECE == 0, "lane run");
    typedef unsigned piece_t __attribute__((ext_vector_type(PIECE / 4), aligned(sizeof(T) < 4 ? sizeof(T) : 4)));
    {
        const piece_t *fp_ = reinterpret_cast<const piece_t *>(from + lane * RPL);
        piece_t raw[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) raw[q] = fp_[q];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            T e[EPP];
            __builtin_memcpy(e, &raw[q], PIECE);
#pragma unroll
            for (int k = 0; k < EPP; ++k) c[q * EPP + k] = Cvt<R, T>::ld(e[k]) * (R)fp.gain;          // coeff.py:268
        }
    }
    line_filter_full<R, RPL>(c, fp, lane);
    {
        piece_t *tp = reinterpret_cast<piece_t *>(base + lane * RPL);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            T e[EPP];
#pragma unroll
            for (int k = 0; k < EPP; ++k) e[k] = Cvt<R, T>::st(c[q * EPP + k]);
            piece_t raw;
            __builtin_memcpy(&raw, e, PIECE);
            tp[q] = raw;
        }
    }
}

// ===========================================================================
// Kernel A', interleaved lines (inner > 1) of 64 * RPL points with a short leading initial sum:
// a workgroup stages a tile of TL neighbouring lines in LDS (the global accesses run along the
// lines' interleaving, TL * sizeof(T) contiguous bytes per point), every wave filters whole lines
// in registers as kernel B' does, and the tile goes back: ONE read and ONE write of the data for
// all poles, instead of a latency-bound thread per line making two passes per pole.
// LDS: TL rows of 64 * (RPL + 1) + 1 floats (odd pitch: the transposing accesses hit all banks).
// ===========================================================================
// LDS element: the storage type for 16-bit data (exact on the way in, and the rounding the store would apply
// anyway on the way out), float otherwise: twice the lines per tile.
template <typename T, typename R, int RPL, int TL, int NT>
__global__ __launch_bounds__(NT) void prefilter_tile(FilterParams fp, const T *src, T *data, int tiles_per_outer)
{
    static_assert(sizeof(R) == 4, "float math only");
    typedef typename std::conditional<sizeof(T) == 2, T, float>::type S;
    extern __shared__ float tile_raw[];
    S *tile = reinterpret_cast<S *>(tile_raw);
    constexpr int n = 64 * RPL, ROW = 64 * (RPL + 1) + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t a = blockIdx.x / tiles_per_outer;
    const int64_t l0 = (int64_t)(blockIdx.x % tiles_per_outer) * TL;
    T *base = data + a * n * fp.inner + l0;
    const T *from = src + a * n * fp.inner + l0;    // == base for the in-place call
    const int nl = fp.inner - l0 < TL ? (int)(fp.inner - l0) : TL;          // lines of this tile
    // (U loads in flight per thread before the first LDS write: a rolled load -> write loop pays one
    //  memory round trip per element -- 0.35 -> 0.20 ms at 32x3x1024^2)
    constexpr int IT = n * TL / NT, U = IT < 16 ? IT : 16;
    static_assert(n * TL % NT == 0 && IT % U == 0, "tile geometry");
#pragma unroll 1
    for (int k0 = 0; k0 < IT; k0 += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = tid + NT * (k0 + u), i = idx / TL, l = idx % TL;
            v[u] = from[(int64_t)i * fp.inner + (l < nl ? l : 0)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = tid + NT * (k0 + u), i = idx / TL, l = idx % TL;
            if (l < nl) {
                if constexpr (sizeof(T) == 2) tile[l * ROW + i + i / RPL] = v[u];
                else tile[l * ROW + i + i / RPL] = Cvt<R, T>::ld(v[u]);
            }
        }
    }
    __syncthreads();
    for (int l = wave; l < nl; l += NT / 64) {
        S *row = tile + l * ROW;
        R c[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) c[r] = Cvt<R, S>::ld(row[lane * (RPL + 1) + r]) * (R)fp.gain;   // coeff.py:268
        line_filter_full<R, RPL>(c, fp, lane);
#pragma unroll
        for (int r = 0; r < RPL; ++r) row[lane * (RPL + 1) + r] = Cvt<R, S>::st(c[r]);
    }
    __syncthreads();
#pragma unroll 4
    for (int idx = tid; idx < n * TL; idx += NT) {
        const int i = idx / TL, l = idx % TL;
        if (l < nl) {
            if constexpr (sizeof(T) == 2) base[(int64_t)i * fp.inner + l] = tile[l * ROW + i + i / RPL];
            else base[(int64_t)i * fp.inner + l] = Cvt<R, T>::st(tile[l * ROW + i + i / RPL]);
        }
    }
}

template <typename T, typename R, int RPL, int TL, int NT>
static int launch_tile(const FilterParams &fp, const void *src, void *data, hipStream_t st)
{
    const size_t lds = (sizeof(T) == 2 ? 2 : 4) * (size_t)TL * (64 * (RPL + 1) + 1);
    static bool done = false;
    if (!done && lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)prefilter_tile<T, R, RPL, TL, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    const int tiles = (int)((fp.inner + TL - 1) / TL);
    hipLaunchKernelGGL((prefilter_tile<T, R, RPL, TL, NT>), dim3((unsigned)(fp.outer * tiles)), dim3(NT), lds, st, fp, (const T *)src, (T *)data, tiles);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ===========================================================================
// Fallback: one thread per line, plain serial walk (any n, any inner).
// ===========================================================================
template <typename T, typename R>
__global__ __launch_bounds__(256) void prefilter_serial(FilterParams fp, T *data)
{
    const int64_t line = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= fp.outer * fp.inner) return;
    const int64_t a0 = line / fp.inner, b0 = line - a0 * fp.inner;
    T *base = data + a0 * fp.n * fp.inner + b0;
    const int64_t st = fp.inner, n = fp.n;
    for (int64_t i = 0; i < n; ++i) base[i * st] = Cvt<R, T>::st(Cvt<R, T>::ld(base[i * st]) * (R)fp.gain);
    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        const R p = (R)pole;
        InitW<R> iw; iw.load(fp.pre[ip], n);
        R sum = R(0);
        for (int64_t i = 0; i < n; ++i) if (iw.on(i)) sum += Cvt<R, T>::ld(base[i * st]) * iw.w(i);
        R prev = iw.scale * sum + iw.c0w * Cvt<R, T>::ld(base[0]);
        base[0] = Cvt<R, T>::st(prev);
        R last2 = prev;
        for (int64_t i = 1; i < n; ++i) { last2 = prev; prev = Cvt<R, T>::ld(base[i * st]) + p * prev; base[i * st] = Cvt<R, T>::st(prev); }
        R fin;
        if (fp.bound == 0) fin = (p * last2 + prev) * (R)fp.pre[ip].fin_mul;
        else if (fp.bound == 1) fin = prev * (R)fp.pre[ip].fin_mul;
        else {
            R dot = R(0), pw = iw.pf * iw.pf;
            for (int64_t i = 0; i < iw.m - 1; ++i) { dot += Cvt<R, T>::ld(base[i * st]) * pw; pw *= iw.pf; }
            fin = (dot + p * prev) * (R)fp.pre[ip].fin_mul;
        }
        base[(n - 1) * st] = Cvt<R, T>::st(fin);
        R next = fin;
        for (int64_t i = n - 2; i >= 0; --i) { next = (next - Cvt<R, T>::ld(base[i * st])) * p; base[i * st] = Cvt<R, T>::st(next); }
    }
}

template <typename T, typename R, int RPL>
static void launch_wave(const FilterParams &fp, void *data, hipStream_t st)
{
    hipLaunchKernelGGL((prefilter_wave<T, R, RPL>), dim3((unsigned)((fp.outer + 3) / 4)), dim3(256), 0, st, fp, (T *)data);
}

template <typename T, typename R>
static int launch_filter_t(const FilterParams &fp, const void *src, void *data, hipStream_t st)
{
    const int64_t lines = fp.outer * fp.inner;
    bool lead = sizeof(R) == 4 && fp.bound != 2;          // short leading initial sum (InitW kinds 0 and 2)
    for (int ip = 0; ip < fp.npoles; ++ip) lead = lead && (fp.pre[ip].kind == 0 || fp.pre[ip].kind == 2);
    if constexpr (sizeof(R) == 4) {
        if (fp.inner == 1 && lead && (fp.n == 64 * 4 || fp.n == 64 * 8 || fp.n == 64 * 16 || fp.n == 64 * 32)) {
            const dim3 g((unsigned)((fp.outer + 3) / 4));
            if (fp.n == 64 * 4) hipLaunchKernelGGL((prefilter_wave_full<T, R, 4>), g, dim3(256), 0, st, fp, (const T *)src, (T *)data);
            else if (fp.n == 64 * 8) hipLaunchKernelGGL((prefilter_wave_full<T, R, 8>), g, dim3(256), 0, st, fp, (const T *)src, (T *)data);
            else if (fp.n == 64 * 16) hipLaunchKernelGGL((prefilter_wave_full<T, R, 16>), g, dim3(256), 0, st, fp, (const T *)src, (T *)data);
            else hipLaunchKernelGGL((prefilter_wave_full<T, R, 32>), g, dim3(256), 0, st, fp, (const T *)src, (T *)data);
            const hipError_t e0 = hipGetLastError();
            return e0 == hipSuccess ? 0 : (int)e0;
        }
    }
    if constexpr (sizeof(R) == 4) {
        // interleaved lines: LDS tiles of TL lines (one pass over the data for all poles)
        if (fp.inner >= 16 && lead && fp.outer * ((fp.inner + 15) / 16) < 0x7fffffff) {
            // (wide tiles: TL * sizeof(T) contiguous bytes per point matter more than a second resident workgroup --
            //  measured at 32x3x1024^2: fp32 0.22 ms with TL = 32 / 1024 threads vs 0.27 with TL = 16 / 512 threads x 2)
            constexpr int W = sizeof(T) == 2 ? 2 : 1;               // 16-bit data: LDS holds the storage type, twice the lines
            if (fp.n == 64 * 4) return launch_tile<T, R, 4, 64 * W, 1024>(fp, src, data, st);
            if (fp.n == 64 * 8) return launch_tile<T, R, 8, 64 * W, 1024>(fp, src, data, st);
            if (fp.n == 64 * 16) return launch_tile<T, R, 16, 32 * W, 1024>(fp, src, data, st);
            if (fp.n == 64 * 32) return launch_tile<T, R, 32, 16 * W, 1024>(fp, src, data, st);
        }
    }
    if (src != data) {
        // the remaining kernels filter in place: copy first
        const hipError_t ec = hipMemcpyAsync(data, src, sizeof(T) * (size_t)(fp.outer * fp.n * fp.inner), hipMemcpyDeviceToDevice, st);
        if (ec != hipSuccess) return (int)ec;
    }
    if (fp.inner == 1 && fp.n <= 64 * 32 && fp.n >= 2) {
        if (fp.n <= 64 * 2) launch_wave<T, R, 2>(fp, data, st);
        else if (fp.n <= 64 * 4) launch_wave<T, R, 4>(fp, data, st);
        else if (fp.n <= 64 * 8) launch_wave<T, R, 8>(fp, data, st);
        else if (fp.n <= 64 * 16) launch_wave<T, R, 16>(fp, data, st);
        else launch_wave<T, R, 32>(fp, data, st);
    } else if (fp.inner >= 16) {
        hipLaunchKernelGGL((prefilter_strided<T, R, 8>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, st, fp, (T *)data);
    } else {
        hipLaunchKernelGGL((prefilter_serial<T, R>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, st, fp, (T *)data);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_filter(int dtype, const FilterParams &fp, const void *src, void *data, hipStream_t st)
{
    switch (dtype) {
    case 0: return launch_filter_t<float, float>(fp, src, data, st);
    case 1: return launch_filter_t<double, double>(fp, src, data, st);
    case 2: return launch_filter_t<bf16_t, float>(fp, src, data, st);
    case 3: return launch_filter_t<f16_t, float>(fp, src, data, st);
    default: return -4;
    }
}

} // namespace ip
