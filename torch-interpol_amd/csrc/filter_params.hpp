// Parameters of the prefilter kernels (prefilter.hip), built by abi.hip.
#pragma once
#include <stdint.h>
#include <math.h>
namespace ip {

// Everything about one pole that does not depend on the data, evaluated ONCE on the host in
// double (pow / log are software routines on the device: hundreds of instructions per call, and
// the kernels used to call them per line and pole).  See InitW in prefilter.hip for the formulas
// (reference interpol/coeff.py:82-227).
struct PolePre {
    int kind;                  // 0 dct1-truncated, 1 dct1-full, 2 dct2, 3 dft
    int64_t m;                 // number of leading (kind 0, 2) or trailing (kind 3) terms that matter
    double pf;                 // float(pole): the reference's pole powers come from the float32-rounded pole
    double pn, pn2;            // pole^(n-1) | pole^n and its square (kinds 1, 2)
    double scale, c0w;         // init = scale * sum + c0w * c[0]
    double fin_mul;            // dct1: pole/(pole^2-1)   dct2: pole/(pole-1)   dft: 1/(pole^m - 1)
};

struct FilterParams {
    int64_t outer, n, inner;   // contiguous (outer, n, inner), filter along the middle axis
    int bound;                 // coeff bound class: 0 = dct1 (zero, dct1), 1 = dct2 (replicate, dct2), 2 = dft
    int npoles;
    double pole[3];            // reference interpol/coeff.py:35-65
    double gain;               // coeff.py:69-73
    PolePre pre[3];
};

// Horizon beyond which |pole|^i is below one ulp of anything (fp64): terms further away are
// dropped from the initial-value sums that the reference extends over the whole line.
inline int64_t filter_horizon(double pole) { return (int64_t)ceil(-44. / log(fabs(pole))); }

inline void make_pole_pre(FilterParams &fp)
{
    const int64_t n = fp.n;
    for (int ip = 0; ip < fp.npoles; ++ip) {
        const double pole = fp.pole[ip];
        PolePre &q = fp.pre[ip];
        q.pf = (double)(float)pole;
        q.pn = 0.; q.pn2 = 0.;
        const int64_t max_iter = (int64_t)ceil(-30. / log(fabs(pole)));        // coeff.py:112, 86
        if (fp.bound == 0) {
            if (max_iter < n) { q.kind = 0; q.m = max_iter; q.scale = 1.; q.c0w = 0.; }
            else {
                q.kind = 1; q.m = n;
                const double polen = pow(pole, (double)(n - 1));
                q.pn = polen; q.pn2 = polen * polen;
                q.scale = 1. / (1. - polen * polen); q.c0w = 0.;
            }
            q.fin_mul = pole / (pole * pole - 1.);
        } else if (fp.bound == 1) {
            q.kind = 2;
            const double polen = pow(pole, (double)n);
            q.pn = polen;
            const int64_t h = filter_horizon(pole);
            q.m = n <= 2 * h ? n : h;              // long lines: the mirrored tail is < 1e-19 of the head
            q.scale = pole / (1. - polen * polen); q.c0w = 1.;
            q.fin_mul = pole / (pole - 1.);
        } else {
            q.kind = 3; q.m = max_iter < n ? max_iter : n;
            q.scale = 1. / (1. - pow(pole, (double)q.m)); q.c0w = 0.;
            q.fin_mul = 1. / (pow(pole, (double)q.m) - 1.);
        }
    }
}

} // namespace ip
