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

// Horizon beyond which |pole|^i is below one ulp of anything (fp64): terms further away
// are dropped from the initial-value sums that the reference extends over the whole line.
__device__ __forceinline__ int64_t horizon(double pole) { return (int64_t)ceil(-44. / log(fabs(pole))); }

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

    __device__ __forceinline__ void make(int bound, double pole, int64_t n_)
    {
        n = n_;
        pf = (R)(float)pole; ipf = R(1) / pf;
        const int64_t max_iter = (int64_t)ceil(-30. / log(fabs(pole)));
        if (bound == 0) {
            if (max_iter < n) { kind = 0; m = max_iter; pn = 0; pn2 = 0; scale = R(1); c0w = R(0); }
            else {
                kind = 1; m = n;
                const double polen = pow(pole, (double)(n - 1));
                pn = (R)polen; pn2 = (R)(polen * polen);
                scale = (R)(1. / (1. - polen * polen)); c0w = R(0);
            }
        } else if (bound == 1) {
            kind = 2;
            const double polen = pow(pole, (double)n);
            pn = (R)polen; pn2 = 0;
            const int64_t h = horizon(pole);
            m = n <= 2 * h ? n : h;               // long lines: the mirrored tail is < 1e-19 of the head
            scale = (R)(pole / (1. - polen * polen)); c0w = R(1);
        } else {
            kind = 3; m = max_iter < n ? max_iter : n; pn = 0; pn2 = 0;
            scale = (R)(1. / (1. - pow(pole, (double)m))); c0w = R(0);
        }
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
        InitW<R> iw; iw.make(fp.bound, pole, n);
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
        if (fp.bound == 0) fin = (p * last2 + prev) * (R)(pole / (pole * pole - 1.));
        else if (fp.bound == 1) fin = prev * (R)(pole / (pole - 1.));
        else {
            R dot = R(0), pw = iw.pf * iw.pf;
            for (int64_t i = 0; i < iw.m - 1; ++i) { dot += Cvt<R, T>::ld(base[i * st]) * pw; pw *= iw.pf; }
            dot += p * prev;
            fin = dot / (R)(pow(pole, (double)iw.m) - 1.);
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
#pragma unroll
    for (int r = 0; r < RPL; ++r) c[r] = (i0 + r < n) ? Cvt<R, T>::ld(base[i0 + r]) * (R)fp.gain : R(0);   // coeff.py:268
    const int last_lane = (int)((n - 1) / RPL), last_r = (int)((n - 1) % RPL);

    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        const R p = (R)pole;
        InitW<R> iw; iw.make(fp.bound, pole, n);
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
        if (fp.bound == 0) fin = (p * c_last2 + c_last) * (R)(pole / (pole * pole - 1.));
        else if (fp.bound == 1) fin = c_last * (R)(pole / (pole - 1.));
        else {
            R dpart = R(0);
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int64_t i = i0 + r;
                if (i < iw.m - 1) dpart += c[r] * powi(iw.pf, i + 2);
            }
            const R dot = wave_sum_r(dpart) + p * c_last;
            fin = dot / (R)(pow(pole, (double)iw.m) - 1.);
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
#pragma unroll
    for (int r = 0; r < RPL; ++r) if (i0 + r < n) base[i0 + r] = Cvt<R, T>::st(c[r]);
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
        InitW<R> iw; iw.make(fp.bound, pole, n);
        R sum = R(0);
        for (int64_t i = 0; i < n; ++i) if (iw.on(i)) sum += Cvt<R, T>::ld(base[i * st]) * iw.w(i);
        R prev = iw.scale * sum + iw.c0w * Cvt<R, T>::ld(base[0]);
        base[0] = Cvt<R, T>::st(prev);
        R last2 = prev;
        for (int64_t i = 1; i < n; ++i) { last2 = prev; prev = Cvt<R, T>::ld(base[i * st]) + p * prev; base[i * st] = Cvt<R, T>::st(prev); }
        R fin;
        if (fp.bound == 0) fin = (p * last2 + prev) * (R)(pole / (pole * pole - 1.));
        else if (fp.bound == 1) fin = prev * (R)(pole / (pole - 1.));
        else {
            R dot = R(0), pw = iw.pf * iw.pf;
            for (int64_t i = 0; i < iw.m - 1; ++i) { dot += Cvt<R, T>::ld(base[i * st]) * pw; pw *= iw.pf; }
            fin = (dot + p * prev) / (R)(pow(pole, (double)iw.m) - 1.);
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
static int launch_filter_t(const FilterParams &fp, void *data, hipStream_t st)
{
    const int64_t lines = fp.outer * fp.inner;
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

int launch_filter(int dtype, const FilterParams &fp, void *data, hipStream_t st)
{
    switch (dtype) {
    case 0: return launch_filter_t<float, float>(fp, data, st);
    case 1: return launch_filter_t<double, double>(fp, data, st);
    case 2: return launch_filter_t<bf16_t, float>(fp, data, st);
    case 3: return launch_filter_t<f16_t, float>(fp, data, st);
    default: return -4;
    }
}

} // namespace ip
