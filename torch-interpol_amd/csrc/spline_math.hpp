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
        R mid = x * (x * (x * (R(5) - x) * R(1. / 6.) - R(1.25)) + R(5. / 24.)) + R(55. / 96.);
        R u = x - R(2.5); u = u * u;
        R up = (u * u) * R(1. / 24.);
        return piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
    }
    case 5: {
        R x2 = x * x;
        R lo = x2 * (x2 * (R(0.25) - x * R(1. / 12.)) - R(0.5)) + R(0.55);
        R mid = x * (x * (x * (x * (x * R(1. / 24.) - R(0.375)) + R(1.25)) - R(1.75)) + R(0.625)) + R(0.425);
        R u = R(3) - x; R u2 = u * u;
        R up = (u * u2 * u2) * R(1. / 120.);
        return piece < 0 ? (x < R(1) ? lo : (x < R(2) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
    }
    case 6: {
        R x2 = x * x;
        R lo = x2 * (x2 * (R(7. / 48.) - x2 * R(1. / 36.)) - R(77. / 192.)) + R(5887. / 11520.);
        R ml = x * (x * (x * (x * (x * (x * R(1. / 48.) - R(7. / 48.)) + R(0.328125)) - R(35. / 288.)) - R(91. / 256.)) - R(7. / 768.)) + R(7861. / 15360.);
        R mu = x * (x * (x * (x * (x * (R(7. / 60.) - x * R(1. / 120.)) - R(0.65625)) + R(133. / 72.)) - R(2.5703125)) + R(1267. / 960.)) + R(1379. / 7680.);
        R u = x - R(3.5); R u3 = u * u * u;
        R up = (u3 * u3) * R(1. / 720.);
        return piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
    }
    case 7: {
        R x2 = x * x;
        R lo = x2 * (x2 * (x2 * (x * R(1. / 144.) - R(1. / 36.)) + R(1. / 9.)) - R(1. / 3.)) + R(151. / 315.);
        R ml = x * (x * (x * (x * (x * (x * (R(0.05) - x * R(1. / 240.)) - R(7. / 30.)) + R(0.5)) - R(7. / 18.)) - R(0.1)) - R(7. / 90.)) + R(103. / 210.);
        R mu = x * (x * (x * (x * (x * (x * (x * R(1. / 720.) - R(1. / 36.)) + R(7. / 30.)) - R(19. / 18.)) + R(49. / 18.)) - R(23. / 6.)) + R(217. / 90.)) - R(139. / 630.);
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
        R mid = x * (x * (x * R(-2. / 3.) + R(2.5)) - R(2.5)) + R(5. / 24.);
        R up = (u * u * u) * R(1. / 48.);
        r = piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
        break;
    }
    case 5: {
        R u = x - R(3); u = u * u;
        R lo = x * (x * (x * (x * R(-5. / 12.) + R(1))) - R(1));
        R mid = x * (x * (x * (x * R(5. / 24.) - R(1.5)) + R(3.75)) - R(3.5)) + R(0.625);
        R up = (u * u) * R(-1. / 24.);
        r = piece < 0 ? (x < R(1) ? lo : (x < R(2) ? mid : up)) : (piece == 0 ? lo : (piece == 1 ? mid : up));
        break;
    }
    case 6: {
        R x2 = x * x;
        R u = R(2) * x - R(7); R u2 = u * u;
        R lo = x * (x2 * R(7. / 12.) - (x2 * x2) * R(1. / 6.) - R(77. / 96.));
        R ml = x * (x * (x * (x * (x * R(0.125) - R(35. / 48.)) + R(1.3125)) - R(35. / 96.)) - R(0.7109375)) - R(7. / 768.);
        R mu = x * (x * (x * (x * (x * R(-1. / 20.) + R(7. / 12.)) - R(2.625)) + R(133. / 24.)) - R(5.140625)) + R(1267. / 960.);
        R up = (u * u2 * u2) * R(1. / 3840.);
        r = piece < 0 ? (x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up))) : (piece == 0 ? lo : (piece == 1 ? ml : (piece == 2 ? mu : up)));
        break;
    }
    case 7: {
        R x2 = x * x;
        R u = x - R(4); R u3 = u * u * u;
        R lo = x * (x2 * (x2 * (x * R(7. / 144.) - R(1. / 6.)) + R(4. / 9.)) - R(2. / 3.));
        R ml = x * (x * (x * (x * (x * (x * R(-7. / 240.) + R(3. / 10.)) - R(7. / 6.)) + R(2)) - R(7. / 6.)) - R(1. / 5.)) - R(7. / 90.);
        R mu = x * (x * (x * (x * (x * (x * R(7. / 720.) - R(1. / 6.)) + R(7. / 6.)) - R(38. / 9.)) + R(49. / 6.)) - R(23. / 3.)) + R(217. / 90.);
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
             : (x < R(1.5) ? x * (R(-2) * x + R(5)) - R(2.5) : (u * u) * R(0.125));
    }
    case 5: {
        R lo = -(x * x) * (x * R(5. / 3.) - R(3)) - R(1);
        R mid = x * (x * (x * R(5. / 6.) - R(4.5)) + R(7.5)) - R(3.5);
        R up = R(4.5) - x * (x * (x * R(1. / 6.) - R(1.5)) + R(4.5));
        return x < R(1) ? lo : (x < R(2) ? mid : up);
    }
    case 6: {
        R x2 = x * x;
        R lo = -x2 * (x2 * R(5. / 6.) - R(1.75)) - R(77. / 96.);
        R ml = x * (x * (x * (x * R(0.625) - R(35. / 12.)) + R(63. / 16.)) - R(35. / 48.)) - R(91. / 128.);
        R mu = -(x * (x * (x * (x * R(0.25) - R(7. / 3.)) + R(63. / 8.)) - R(133. / 12.)) + R(329. / 64.));
        R up = x * (x * (x * (x * R(1. / 24.) - R(7. / 12.)) + R(49. / 16.)) - R(343. / 48.)) + R(2401. / 384.);
        return x < R(0.5) ? lo : (x < R(1.5) ? ml : (x < R(2.5) ? mu : up));
    }
    case 7: {
        R x2 = x * x;
        R lo = x2 * (x2 * (x * R(7. / 24.) - R(5. / 6.)) + R(4. / 3.)) - R(2. / 3.);
        R ml = -(x * (x * (x * (x * (x * R(7. / 40.) - R(1.5)) + R(14. / 3.)) - R(6)) + R(7. / 3.)) + R(0.2));
        R mu = x * (x * (x * (x * (x * R(7. / 120.) - R(5. / 6.)) + R(14. / 3.)) - R(38. / 3.)) + R(49. / 3.)) - R(23. / 3.);
        R up = -(x * (x * (x * (x * (x * R(1. / 120.) - R(1. / 6.)) + R(4. / 3.)) - R(16. / 3.)) + R(32. / 3.)) - R(128. / 15.));
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
