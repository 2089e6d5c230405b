// Not compiled (experiments/README.md).  Round 5: a windowed, out-of-place column prefilter -- measured slower than prefilter_tile.
// The kernel and its dispatch (inside launch_filter_t of csrc/prefilter.hip) as they were:

// ===========================================================================
// Kernel W (round 5), interleaved lines, OUT OF PLACE, one pole with |pole|^WH below 1e-9 (quadratic, cubic): a thread filters a WINDOW
// of one line in registers -- WS outputs, WH points of warm-up on either side: the causal recursion started from nothing WH points
// early and the anticausal one WH points late are exact to pole^WH (7e-10 for the cubic pole), the decay that lets the reference
// truncate its own initial sums (coeff.py:109-179).  Consecutive threads hold consecutive lines: every access is a coalesced row, there
// is no transposition through LDS and no second pass; the data is read 1.5 times (the halos, mostly from the Infinity Cache) and
// written once.  The first window of a line starts from the exact initial value, the last one ends on the exact final value.
// 32 x 3 x 1024^2 along dim -2: bf16 0.18 -> see profiles/r05_other_configs.json (prefilter_tile: 139 KiB of LDS, one workgroup per CU, ~1
// element per clock and CU whatever the element size).
// ===========================================================================
template <typename T, int WS, int WH>
__global__ __launch_bounds__(256) void prefilter_window(FilterParams fp, const T *__restrict__ src, T *__restrict__ dst, int nseg, int lblocks)
{
    typedef float R;
    constexpr int L = WS + 2 * WH;
    int r = blockIdx.x;
    const int lb = r % lblocks; r /= lblocks;
    const int q = r % nseg;
    const int64_t a_ = r / nseg;
    const int64_t line = (int64_t)lb * 256 + threadIdx.x;
    if (line >= fp.inner) return;
    const int n = (int)fp.n;
    const int o0 = q * WS;                                           // first output of the window
    const int a = o0 - WH > 0 ? o0 - WH : 0, b = o0 + WS + WH < n ? o0 + WS + WH : n;     // points [a, b) in registers
    const T *from = src + a_ * n * fp.inner + line;
    T *to = dst + a_ * n * fp.inner + line;
    const R p = (R)fp.pole[0], gain = (R)fp.gain;
    R c[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int idx = a + i < b ? a + i : b - 1;
        c[i] = Cvt<R, T>::ld(from[(int64_t)idx * fp.inner]) * gain;                          // coeff.py:268
    }
    // causal pass (coeff.py:270-276); a == 0: the exact initial value (leading terms only: the host checked m <= WS)
    if (a == 0) {
        InitW<R> iw; iw.load(fp.pre[0], n);
        R sum = R(0), pw = R(1);
#pragma unroll
        for (int i = 0; i < WS; ++i) { if (i < iw.m && i < n) sum += c[i] * pw; pw *= iw.pf; }
        c[0] = iw.scale * sum + iw.c0w * c[0];
    }
#pragma unroll
    for (int i = 1; i < L; ++i) c[i] = a + i < b ? c[i] + p * c[i - 1] : c[i];
    // anticausal pass (coeff.py:278-284); b == n: the exact final value
    const int last = b - 1 - a;                                      // register of point b - 1
    R next;
    if (b == n) {
        R prev = R(0), last2 = R(0);
#pragma unroll
        for (int i = 0; i < L; ++i) { if (i == last) prev = c[i]; if (i == last - 1) last2 = c[i]; }
        next = fp.bound == 0 ? (p * last2 + prev) * (R)fp.pre[0].fin_mul : prev * (R)fp.pre[0].fin_mul;
    } else {
        next = R(0);                                                 // (nothing beyond the window: pole^WH later it no longer matters)
    }
#pragma unroll
    for (int i = L - 1; i >= 0; --i) {
        if (i > last) continue;
        if (i == last) { if (b != n) next = (next - c[i]) * p; c[i] = next; }
        else { next = (next - c[i]) * p; c[i] = next; }
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int idx = a + i;
        if (idx >= o0 && idx < o0 + WS && idx < n) to[(int64_t)idx * fp.inner] = Cvt<R, T>::st(c[i]);
    }
}

template <typename T>
static int launch_window(const FilterParams &fp, const void *src, void *data, hipStream_t st)
{
    constexpr int WS = 64, WH = 16;
    const int nseg = (int)((fp.n + WS - 1) / WS), lblocks = (int)((fp.inner + 255) / 256);
    const int64_t blocks = (int64_t)nseg * lblocks * fp.outer;
    if (blocks > 0x7fffffffll) return -1;
    hipLaunchKernelGGL((prefilter_window<T, WS, WH>), dim3((unsigned)blocks), dim3(256), 0, st, fp, (const T *)src, (T *)data, nseg, lblocks);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


/* dispatch:
    if constexpr (sizeof(R) == 4) {
        // interleaved lines, out of place, one fast-decaying pole: windows of a line in registers (kernel W)
        if (src != data && fp.inner >= 64 && lead && fp.npoles == 1 && fabs(fp.pole[0]) < 0.2737 && fp.pre[0].m <= 64 && fp.pre[0].m < fp.n && fp.n >= 2
            && fp.n < 0x40000000) {
            const int rc = launch_window<T>(fp, src, data, st);
            if (rc >= 0) return rc;
        }
    }
*/
