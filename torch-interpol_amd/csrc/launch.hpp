// ===========================================================================
// launch.hpp -- host-side variant selection shared by the typed launchers.
// ===========================================================================
#pragma once
#include "stencil.hpp"
#include <type_traits>

namespace ip {

// launch geometry: x = ceil(N / BLOCK) blocks of one batch item, y = batch (strided)
inline dim3 sample_grid(const KParams &p, int B)
{
    const int64_t bx = (p.N + 255) / 256;
    // gate_n == -1 / -2: one of several organisations enqueued behind a router's verdict, usually not the one that runs: a few batch
    // items' worth of blocks, which stride over the batch -- 131 072 workgroups that return at once cost 32 us, 262 144 (config 2's
    // trilinear pull) 60.  (Striding over the samples as well was tried: the loop-variant sample index costs the generic kernels
    // 35 - 50 % on rough fields.)
    const bool gated = p.gate && (p.gate_n == -1 || p.gate_n == -2);      // (-4 stands in for the tiles: sized like an ungated launch)
    int by = B;
    if (gated) { by = (int)(32768 / (bx > 0 ? bx : 1)); by = by < 1 ? 1 : (by > B ? B : by); }
    return dim3((unsigned)bx, (unsigned)(by < 65535 ? by : 65535), 1);
}

// Which compiled variant serves this problem:
//   all dims share one order k      -> <D, k, ISO=true>
//   mixed orders, max <= 3          -> <D, 3, ISO=false>   (taps beyond a dim's order predicated off)
//   mixed orders, max <= 7          -> <D, 7, ISO=false>
struct Variant { int D, K; bool iso; };

inline Variant pick_variant(const KParams &p)
{
    bool same = true; int mx = 0;
    for (int d = 0; d < p.dim; ++d) { same = same && (p.order[d] == p.order[0]); mx = p.order[d] > mx ? p.order[d] : mx; }
    if (same) return { p.dim, p.order[0], true };
    return { p.dim, mx <= 3 ? 3 : 7, false };
}

template <int D, bool ISO, typename F>
int dispatch_k(int K, F &&f)
{
    using std::integral_constant;
    if constexpr (ISO) {
        switch (K) {
        case 0: f(integral_constant<int, D>{}, integral_constant<int, 0>{}, integral_constant<bool, ISO>{}); return 0;
        case 1: f(integral_constant<int, D>{}, integral_constant<int, 1>{}, integral_constant<bool, ISO>{}); return 0;
        case 2: f(integral_constant<int, D>{}, integral_constant<int, 2>{}, integral_constant<bool, ISO>{}); return 0;
        case 4: f(integral_constant<int, D>{}, integral_constant<int, 4>{}, integral_constant<bool, ISO>{}); return 0;
        case 5: f(integral_constant<int, D>{}, integral_constant<int, 5>{}, integral_constant<bool, ISO>{}); return 0;
        case 6: f(integral_constant<int, D>{}, integral_constant<int, 6>{}, integral_constant<bool, ISO>{}); return 0;
        default: break;
        }
    }
    switch (K) {
    case 3: f(integral_constant<int, D>{}, integral_constant<int, 3>{}, integral_constant<bool, ISO>{}); return 0;
    case 7: f(integral_constant<int, D>{}, integral_constant<int, 7>{}, integral_constant<bool, ISO>{}); return 0;
    default: return -2;
    }
}

template <typename F>
int dispatch_variant(const KParams &p, F &&f)
{
    const Variant v = pick_variant(p);
    int rc;
    if (v.iso) {
        switch (v.D) {
        case 1: rc = dispatch_k<1, true>(v.K, f); break;
        case 2: rc = dispatch_k<2, true>(v.K, f); break;
        case 3: rc = dispatch_k<3, true>(v.K, f); break;
        default: return -1;
        }
    } else {
        switch (v.D) {
        case 1: rc = dispatch_k<1, false>(v.K, f); break;
        case 2: rc = dispatch_k<2, false>(v.K, f); break;
        case 3: rc = dispatch_k<3, false>(v.K, f); break;
        default: return -1;
        }
    }
    if (rc != 0) return rc;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace ip
