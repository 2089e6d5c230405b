// ===========================================================================
// prefilter.hip -- interpolating-coefficient prefilter (recursive IIR), in place.
//
// Numerical definition = reference interpol/coeff.py:258-284 (filter), with the
// boundary-specific initial/final values of coeff.py:82-227 and the bound
// mapping of coeff.py:231-254 (zero -> dct1, replicate -> dct2).  The reference
// runs the recursion as a serial Python loop of n tiny tensor ops per pole; here
// one thread owns one line and streams through it, all poles in sequence.
//
// Layout: contiguous (outer, n, inner), filter along the middle axis.
//   inner > 1 : thread = (outer index, inner index); consecutive lanes touch
//               consecutive addresses at every step of the recursion (coalesced).
//   inner == 1: lines are contiguous in memory; a block stages a tile of
//               LINES x n samples through LDS so that global traffic is coalesced.
// ===========================================================================
#include "stencil.hpp"
#include "filter_params.hpp"
#include <math.h>

namespace ip {

// One line, generic accessor A (A.get(i), A.set(i, v)); R = math type.
template <typename R, typename A>
__device__ __forceinline__ void filter_line(const FilterParams &fp, A &a)
{
    const int64_t n = fp.n;
    for (int64_t i = 0; i < n; ++i) a.set(i, a.get(i) * (R)fp.gain);              // coeff.py:268
    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        const R p = (R)pole;
        // the reference builds its tensor of pole powers from float(pole)
        // (TorchScript as_tensor quirk, see oracle/interpol_oracle_body.inc)
        const R pf = (R)(float)pole;
        int64_t max_iter = (int64_t)ceil(-30. / log(fabs(pole)));
        R init, fin;
        if (fp.bound == 0) {                                                       // dct1_initial, coeff.py:109-149
            if (max_iter < n) {
                R acc = R(0), pw = R(1);
                for (int64_t i = 0; i < max_iter; ++i) { acc += a.get(i) * pw; pw *= pf; }
                init = acc;
            } else {
                const double polen = pow(pole, (double)(n - 1));
                R acc = a.get(0) + (R)polen * a.get(n - 1);
                R dot = R(0), pw = pf;
                const R pn2 = (R)(polen * polen);
                for (int64_t i = 1; i < n - 1; ++i) { dot += a.get(i) * (pw + pn2 / pw); pw *= pf; }
                acc += dot;
                init = acc / (R)(1. - polen * polen);
            }
        } else if (fp.bound == 1) {                                                // dct2_initial, coeff.py:153-179
            const double polen = pow(pole, (double)n);
            // poles[i] + polen * poles[n-1-i]
            R dot = R(0), pw = R(1), pwr = (R)pow((double)(float)pole, (double)(n - 1));
            const R ipf = R(1) / pf;
            for (int64_t i = 0; i < n; ++i) { dot += a.get(i) * (pw + (R)polen * pwr); pw *= pf; pwr *= ipf; }
            init = dot * (R)(pole / (1. - polen * polen)) + a.get(0);
        } else {                                                                   // dft_initial, coeff.py:82-105
            const int64_t m = max_iter < n ? max_iter : n;
            R dot = R(0), pw = pf;
            for (int64_t j = 1; j < m; ++j) { dot += a.get(n - j) * pw; pw *= pf; }
            init = (dot + a.get(0)) / (R)(1. - pow(pole, (double)m));
        }
        a.set(0, init);
        R prev = init;
        for (int64_t i = 1; i < n; ++i) { prev = a.get(i) + p * prev; a.set(i, prev); }   // coeff.py:275-276
        if (fp.bound == 0) {                                                       // dct1_final, coeff.py:208-215
            fin = (p * a.get(n - 2) + a.get(n - 1)) * (R)(pole / (pole * pole - 1.));
        } else if (fp.bound == 1) {                                                // dct2_final, coeff.py:219-227
            fin = a.get(n - 1) * (R)(pole / (pole - 1.));
        } else {                                                                   // dft_final, coeff.py:183-204
            const int64_t m = max_iter < n ? max_iter : n;
            R dot = R(0), pw = pf * pf;
            for (int64_t i = 0; i < m - 1; ++i) { dot += a.get(i) * pw; pw *= pf; }
            dot += p * a.get(n - 1);
            fin = dot / (R)(pow(pole, (double)m) - 1.);
        }
        a.set(n - 1, fin);
        R next = fin;
        for (int64_t i = n - 2; i >= 0; --i) { next = (next - a.get(i)) * p; a.set(i, next); }   // coeff.py:280-281
    }
}

template <typename T, typename R>
struct StridedLine {
    T *base; int64_t stride;
    __device__ __forceinline__ R get(int64_t i) const { return Cvt<R, T>::ld(base[i * stride]); }
    __device__ __forceinline__ void set(int64_t i, R v) { base[i * stride] = Cvt<R, T>::st(v); }
};

// inner > 1 (or generic fallback): one thread per line, strided walk.
template <typename T, typename R>
__global__ __launch_bounds__(256) void prefilter_strided(FilterParams fp, T *data)
{
    const int64_t line = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (line >= fp.outer * fp.inner) return;
    const int64_t a = line / fp.inner, b = line - a * fp.inner;
    StridedLine<T, R> acc{ data + a * fp.n * fp.inner + b, fp.inner };
    filter_line<R>(fp, acc);
}

// inner == 1: LINES contiguous lines per block staged through LDS (row stride n+1
// words: conflict-free column walk), coalesced global load/store of the tile.
template <typename R>
struct LdsLine {
    R *row;
    __device__ __forceinline__ R get(int64_t i) const { return row[i]; }
    __device__ __forceinline__ void set(int64_t i, R v) { row[i] = v; }
};

template <typename T, typename R, int LINES>
__global__ __launch_bounds__(LINES) void prefilter_lds(FilterParams fp, T *data)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *tile = reinterpret_cast<R *>(smem_raw);
    const int64_t n = fp.n, ld = n + 1;
    const int64_t line0 = (int64_t)blockIdx.x * LINES;
    const int64_t nlines = (fp.outer - line0) < LINES ? (fp.outer - line0) : LINES;
    T *g = data + line0 * n;
    for (int64_t e = threadIdx.x; e < nlines * n; e += LINES) {
        const int64_t r = e / n, c = e - r * n;
        tile[r * ld + c] = Cvt<R, T>::ld(g[e]);
    }
    __syncthreads();
    if ((int64_t)threadIdx.x < nlines) {
        LdsLine<R> acc{ tile + threadIdx.x * ld };
        filter_line<R>(fp, acc);
    }
    __syncthreads();
    for (int64_t e = threadIdx.x; e < nlines * n; e += LINES) {
        const int64_t r = e / n, c = e - r * n;
        g[e] = Cvt<R, T>::st(tile[r * ld + c]);
    }
}

template <typename T, typename R>
static int launch_filter_t(const FilterParams &fp, void *data, hipStream_t st)
{
    constexpr int LINES = 64;
    const size_t lds = (size_t)LINES * (fp.n + 1) * sizeof(R);
    if (fp.inner == 1 && lds <= 64 * 1024) {
        const int64_t blocks = (fp.outer + LINES - 1) / LINES;
        hipLaunchKernelGGL((prefilter_lds<T, R, LINES>), dim3((unsigned)blocks), dim3(LINES), lds, st, fp, (T *)data);
    } else {
        const int64_t lines = fp.outer * fp.inner;
        hipLaunchKernelGGL((prefilter_strided<T, R>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, st, fp, (T *)data);
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
