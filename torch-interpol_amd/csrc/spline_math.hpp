// ===========================================================================
// spline_math.hpp -- scalar primitives of the sampling kernels, usable from
// device code (gfx950) and from the host (for GPU-less checks).
//
//   bspline_w / bspline_g / bspline_h : centred cardinal B-spline of order 0..7,
//       its first and second derivative; numerical definition =
//       reference interpol/splines.py:30-80, 90-139, 149-195 (piecewise Horner
//       forms; divisions by constants folded into multiplications).
//   wrap_index / wrap_sign : boundary handling, numerical definition =
//       reference interpol/bounds.py:30-60 (Bound.index), 62-89 (Bound.transform).
//
// All functions are force-inlined; when `order` / `bound` are compile-time or
// wave-uniform the switches fold into straight-line code / scalar branches.
// ===========================================================================
#pragma once
#include <hip/hip_runtime.h>

#define IP_HD __host__ __device__ __forceinline__

namespace ip {

enum Bound : int { B_ZERO = 0, B_REPLICATE = 1, B_DCT1 = 2, B_DCT2 = 3, B_DST1 = 4, B_DST2 = 5, B_DFT = 6 };

template <typename R> IP_HD R rabs(R x) { return x < R(0) ? -x : x; }

// --------------------------------------------------------------------------
// B-spline value (splines.py:30-80).  |t| <= (order+1)/2 is the caller's job.
// --------------------------------------------------------------------------
// `piece` >= 0 names the polynomial piece the caller KNOWS |t| to lie in (0 = innermost): the tiled kernels know it per
// tap from the interval of the stencil coordinate, and with a compile-time piece the other pieces are never evaluated.
template <typename R>
IP_HD R bspline_w(int order, R t, int piece = -1)
{
    R x = rabs(t);
    switch (order) {
    case 0: return R(1);
    case 1: return R(1) - x;
    case 2: {
        R u = R(1.5) - x;
        { const R lo = R(0.75) - x * x, up = R(0.5) * (u * u); return piece < 0 ? (x < R(0.5) ? lo : up) : (piece == 0 ? lo : up); }
    }
    case 3: {
        R u = R(2) - x;
        { const R lo = (x * x * (x - R(2)) * R(3) + R(4)) * R(1. / 6.), up = (u * u * u) * R(1. / 6.); return piece < 0 ? (x < R(1) ? lo : up) : (piece == 0 ? lo : up); }
    }
    case 4: {
        R x2 = x * x;
        R lo = x2 * (x2 * R(0.25) - R(0.625)) + R(115. / 192.);
        R v = R(1.5) - x;                                             // (the middle piece about its outer breakpoint, see order 6)
        R mid = (((R(-1. / 6.) * v + R(1. / 6.)) * v + R(0.25)) * v + R(1. / 6.)) * v + R(1. / 24.);
        R u = x - R(2.5); u = u * u;
        R up = (u * u) * R(1. / 24.);
        return piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
    }
    case 5: {
        R x2 = x * x;
        R lo = x2 * (x2 * (R(0.25) - x * R(1. / 12.)) - R(0.5)) + R(0.55);
        R v = R(2) - x;                                               // (the middle piece about its outer breakpoint, see order 6)
        R mid = ((((R(-1. / 24.) * v + R(1. / 24.)) * v + R(1. / 12.)) * v + R(1. / 12.)) * v + R(1. / 24.)) * v + R(1. / 120.);
        R u = R(3) - x; R u2 = u * u;
        R up = (u * u2 * u2) * R(1. / 120.);
        return piece < 0 ? (x < R(1) ? lo : (x < R(2) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
    }
    case 6: {
        R x2 = x * x;
        R lo = x2 * (x2 * (R(7. / 48.) - x2 * R(1. / 36.)) - R(77. / 192.)) + R(5887. / 11520.);
        // (the middle pieces are expanded about their OUTER breakpoint, v = 1.5 - x and 2.5 - x in (0, 1]: the same polynomials as
        //  splines.py:60-66, but Horner in x sums terms of order 1 to a value of order 1e-3 and loses five digits in fp32)
        R v = R(1.5) - x;
        R ml = (((((R(1. / 48.) * v - R(1. / 24.)) * v - R(1. / 16.)) * v + R(1. / 36.)) * v + R(3. / 16.)) * v + R(5. / 24.)) * v + R(19. / 240.);
        v = R(2.5) - x;
        R mu = (((((R(-1. / 120.) * v + R(1. / 120.)) * v + R(1. / 48.)) * v + R(1. / 36.)) * v + R(1. / 48.)) * v + R(1. / 120.)) * v + R(1. / 720.);
        R u = x - R(3.5); R u3 = u * u * u;
        R up = (u3 * u3) * R(1. / 720.);
        return piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
    }
    case 7: {
        R x2 = x * x;
        R lo = x2 * (x2 * (x2 * (x * R(1. / 144.) - R(1. / 36.)) + R(1. / 9.)) - R(1. / 3.)) + R(151. / 315.);
        // (middle pieces about their outer breakpoint, v = 2 - x and 3 - x in (0, 1]: splines.py:70-76 re-expanded, see order 6)
        R v = R(2) - x;
        R ml = ((((((R(1. / 240.) * v - R(1. / 120.)) * v - R(1. / 60.)) * v) * v + R(1. / 18.)) * v + R(1. / 10.)) * v + R(7. / 90.)) * v + R(1. / 42.);
        v = R(3) - x;
        R mu = ((((((R(-1. / 720.) * v + R(1. / 720.)) * v + R(1. / 240.)) * v + R(1. / 144.)) * v + R(1. / 144.)) * v + R(1. / 240.)) * v + R(1. / 720.)) * v + R(1. / 5040.);
        R u = R(4) - x; R u3 = u * u * u;
        R up = (u3 * u3 * u) * R(1. / 5040.);
        return piece < 0 ? (x < R(1) ? lo : (x < R(2) ? ml : (x < R(3) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
    }
    default: return R(0);
    }
}

// --------------------------------------------------------------------------
// First derivative (splines.py:90-139): _fastgrad(|t|) * sign(t).
// Order 1 in the generic (nd) path is +sign(t) in the reference (its quirk B-4,
// SURVEY Appendix B): reproduced here because parity is the contract.  The
// all-linear path (iso1.py) uses the correct -1/+1 and never calls this.
// --------------------------------------------------------------------------
template <typename R>
IP_HD R bspline_g(int order, R t, int piece = -1)
{
    R s = t > R(0) ? R(1) : (t < R(0) ? R(-1) : R(0));
    R x = rabs(t);
    R r;
    switch (order) {
    case 1: r = R(1); break;
    case 2: { const R lo = R(-2) * x, up = x - R(1.5); r = piece < 0 ? (x < R(0.5) ? lo : up) : (piece == 0 ? lo : up); break; }
    case 3: {
        R u = R(2) - x;
        { const R lo = x * (x * R(1.5) - R(2)), up = R(-0.5) * (u * u); r = piece < 0 ? (x < R(1) ? lo : up) : (piece == 0 ? lo : up); }
        break;
    }
    case 4: {
        R u = R(2) * x - R(5);
        R lo = x * (x * x - R(1.25));
        R v = R(1.5) - x;
        R mid = ((R(2. / 3.) * v - R(0.5)) * v - R(0.5)) * v - R(1. / 6.);
        R up = (u * u * u) * R(1. / 48.);
        r = piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
        break;
    }
    case 5: {
        R u = x - R(3); u = u * u;
        R lo = x * (x * (x * (x * R(-5. / 12.) + R(1))) - R(1));
        R v = R(2) - x;
        R mid = (((R(5. / 24.) * v - R(1. / 6.)) * v - R(0.25)) * v - R(1. / 6.)) * v - R(1. / 24.);
        R up = (u * u) * R(-1. / 24.);
        r = piece < 0 ? (x < R(1) ? lo : (x < R(2) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
        break;
    }
    case 6: {
        R x2 = x * x;
        R u = R(2) * x - R(7); R u2 = u * u;
        R lo = x * (x2 * R(7. / 12.) - (x2 * x2) * R(1. / 6.) - R(77. / 96.));
        R v = R(1.5) - x;                                             // (about the outer breakpoint, see bspline_w)
        R ml = ((((R(-1. / 8.) * v + R(5. / 24.)) * v + R(1. / 4.)) * v - R(1. / 12.)) * v - R(3. / 8.)) * v - R(5. / 24.);
        v = R(2.5) - x;
        R mu = ((((R(1. / 20.) * v - R(1. / 24.)) * v - R(1. / 12.)) * v - R(1. / 12.)) * v - R(1. / 24.)) * v - R(1. / 120.);
        R up = (u * u2 * u2) * R(1. / 3840.);
        r = piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
        break;
    }
    case 7: {
        R x2 = x * x;
        R u = x - R(4); R u3 = u * u * u;
        R lo = x * (x2 * (x2 * (x * R(7. / 144.) - R(1. / 6.)) + R(4. / 9.)) - R(2. / 3.));
        R v = R(2) - x;                                               // (about the outer breakpoint, see bspline_w)
        R ml = (((((R(-7. / 240.) * v + R(1. / 20.)) * v + R(1. / 12.)) * v) * v - R(1. / 6.)) * v - R(1. / 5.)) * v - R(7. / 90.);
        v = R(3) - x;
        R mu = (((((R(7. / 720.) * v - R(1. / 120.)) * v - R(1. / 48.)) * v - R(1. / 36.)) * v - R(1. / 48.)) * v - R(1. / 120.)) * v - R(1. / 720.);
        R up = (u3 * u3) * R(-1. / 720.);
        r = piece < 0 ? (x < R(1) ? lo : (x < R(2) ? ml : (x < R(3) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
        break;
    }
    default: return R(0);
    }
    return r * s;
}

// --------------------------------------------------------------------------
// Second derivative (splines.py:149-195).
// --------------------------------------------------------------------------
template <typename R>
IP_HD R bspline_h(int order, R t)
{
    R x = rabs(t);
    switch (order) {
    case 2: return x < R(0.5) ? R(-2) : R(1);
    case 3: return x < R(1) ? R(3) * x - R(2) : R(2) - x;
    case 4: {
        R u = R(2) * x - R(5);
        return x < R(0.5) ? R(3) * (x * x) - R(1.25)
             : (x < R(1.5) ? (R(-2) * (R(1.5) - x) + R(1)) * (R(1.5) - x) + R(0.5) : (u * u) * R(0.125));
    }
    case 5: {
        R lo = -(x * x) * (x * R(5. / 3.) - R(3)) - R(1);
        R v = R(2) - x;
        R mid = ((R(-5. / 6.) * v + R(0.5)) * v + R(0.5)) * v + R(1. / 6.);
        R u = R(3) - x;
        R up = (u * u * u) * R(1. / 6.);                              // (= 4.5 - x (x (x / 6 - 1.5) + 4.5), splines.py:173, without the cancellation)
        return x < R(1) ? lo : (x < R(2) ? mid : up);
    }
    case 6: {
        R x2 = x * x;
        R lo = -x2 * (x2 * R(5. / 6.) - R(1.75)) - R(77. / 96.);
        R v = R(1.5) - x;                                             // (about the outer breakpoint, see bspline_w)
        R ml = (((R(5. / 8.) * v - R(5. / 6.)) * v - R(3. / 4.)) * v + R(1. / 6.)) * v + R(3. / 8.);
        v = R(2.5) - x;
        R mu = (((R(-1. / 4.) * v + R(1. / 6.)) * v + R(1. / 4.)) * v + R(1. / 6.)) * v + R(1. / 24.);
        R u = x - R(3.5); u = u * u;
        R up = (u * u) * R(1. / 24.);                                 // (the outer piece as a power of the distance to the end of the support)
        return x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up));
    }
    case 7: {
        R x2 = x * x;
        R lo = x2 * (x2 * (x * R(7. / 24.) - R(5. / 6.)) + R(4. / 3.)) - R(2. / 3.);
        R v = R(2) - x;                                               // (about the outer breakpoint, see bspline_w)
        R ml = ((((R(7. / 40.) * v - R(1. / 4.)) * v - R(1. / 3.)) * v) * v + R(1. / 3.)) * v + R(1. / 5.);
        v = R(3) - x;
        R mu = ((((R(-7. / 120.) * v + R(1. / 24.)) * v + R(1. / 12.)) * v + R(1. / 12.)) * v + R(1. / 24.)) * v + R(1. / 120.);
        R u = R(4) - x; R u2 = u * u;
        R up = (u * u2 * u2) * R(1. / 120.);
        return x < R(1) ? lo : (x < R(2) ? ml : (x < R(3) ? mu : up));
    }
    default: return R(0);
    }
}

// --------------------------------------------------------------------------
// Boundary conditions on int32 lattice indices.
// Python-style remainder, m > 0.
// --------------------------------------------------------------------------
IP_HD int pymod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// Out-of-range branch of Bound.index (bounds.py:30-60).  `i` is NOT in [0, n).
// One period around the lattice is handled without an integer division (the
// common case for smooth deformations); anything further out takes the modulo.
IP_HD int wrap_index_outside(int bound, int i, int n)
{
    switch (bound) {
    case B_ZERO: case B_REPLICATE:
        return i < 0 ? 0 : n - 1;
    case B_DCT2: case B_DST2: {
        if (i < 0 && i >= -n) return -1 - i;
        if (i >= n && i < 2 * n) return 2 * n - 1 - i;
        int n2 = 2 * n;
        int j = i < 0 ? (n2 - 1) - pymod(-i - 1, n2) : pymod(i, n2);
        return j >= n ? n2 - 1 - j : j;
    }
    case B_DCT1: {
        if (n == 1) return 0;
        if (i < 0 && i > -n) return -i;
        if (i >= n && i < 2 * n - 1) return 2 * (n - 1) - i;
        int n2 = 2 * (n - 1);
        int j = pymod(i < 0 ? -i : i, n2);
        return j >= n ? n2 - j : j;
    }
    case B_DST1: {
        int n2 = 2 * (n + 1);
        int j = i < 0 ? -i - 2 : i;
        j = pymod(j, n2);
        if (j > n) j = n2 - 2 - j;
        if (j == -1) j = 0;
        if (j == n) j = n - 1;
        return j;
    }
    case B_DFT: {
        if (i < 0 && i >= -n) return i + n;
        if (i >= n && i < 2 * n) return i - n;
        return pymod(i, n);
    }
    default: return i;
    }
}

// Bound.index (bounds.py:30-60)
IP_HD int wrap_index(int bound, int i, int n)
{
    if ((unsigned)i < (unsigned)n) return i;
    return wrap_index_outside(bound, i, n);
}

// Bound.transform (bounds.py:62-89).  Returns 2 where the reference returns None.
IP_HD int wrap_sign_raw(int bound, int i, int n)
{
    switch (bound) {
    case B_ZERO:
        return (unsigned)i < (unsigned)n ? 1 : 0;
    case B_DST2: {
        if ((unsigned)i < (unsigned)n) return 1;
        int j = i < 0 ? n - 1 - i : i;
        return ((j / n) & 1) ? -1 : 1;
    }
    case B_DST1: {
        // NB: 0 at i == 0 and at every i = 0 mod 2(n+1): reference quirk B-3, kept.
        if (n == 1) return 2;
        int n2 = 2 * (n + 1);
        int j = i < 0 ? -i + (n - 1) : i;
        j = pymod(j, n2);
        int x = (j == 0) ? 0 : 1;
        if (pymod(j, n + 1) == n) x = 0;
        return ((j / (n + 1)) & 1) ? -x : x;
    }
    default: return 2;
    }
}

// Same, as a multiplicative factor (None -> 1).
IP_HD int wrap_sign(int bound, int i, int n)
{
    if (bound != B_ZERO && bound != B_DST1 && bound != B_DST2) return 1;
    int s = wrap_sign_raw(bound, i, n);
    return s == 2 ? 1 : s;
}

} // namespace ip
