// Parameters of the prefilter kernels (prefilter.hip), built by abi.hip.
#pragma once
#include <stdint.h>
namespace ip {
struct FilterParams {
    int64_t outer, n, inner;   // contiguous (outer, n, inner), filter along the middle axis
    int bound;                 // coeff bound class: 0 = dct1 (zero, dct1), 1 = dct2 (replicate, dct2), 2 = dft
    int npoles;
    double pole[3];            // reference interpol/coeff.py:35-65
    double gain;               // coeff.py:69-73
};
} // namespace ip
