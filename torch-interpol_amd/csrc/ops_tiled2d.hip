// ===========================================================================
// ops_tiled2d.hip -- lean LDS tiles for 2-D problems (BASELINE config 5: batches of 2-D images,
// mixed per-dim spline orders 1..3, bf16 / f16 / f32 storage, fp32 grid and math).
//   pull  : reference interpol/nd.py:80-143          push / count : nd.py:146-213, pushpull.py:106-142
//
// A 2-D stencil has at most 16 taps: the work per sample is small, so what decides the speed is the
// fixed cost per tile (the 3-D tile machinery of ops_tiled.hip loses to the generic kernel in 2-D for
// that reason) and the width of the memory accesses.  Hence:
//   * one workgroup of 256 threads per tile of 32 x 32 pixels, ~35 KiB of LDS, three (pull) or four (push)
//     workgroups per CU; nothing is sorted, nothing is done in passes;
//   * a thread owns FOUR NEIGHBOURING pixels of a row: coordinates, sources and results move as 16- / 32-
//     byte accesses (narrow accesses are what the vector memory path is slow at), and so do the rows of
//     the box when they are contiguous in memory -- as stored, without conversion: the boundary sign
//     (bounds.py:30-89) is a flip / clear of bits;
//   * the bounding box of the tile's stencils (<= 64 x 64 lattice points; beyond: the sample gathers /
//     scatters straight from / to global memory, out of line) is staged ONCE; an 8-byte LDS slot holds
//     ALL channels of a lattice point that fit -- four 16-bit values or two floats -- so one ds_read_b64
//     per tap feeds them all (config 5: 3 channels, packed FMAs on two of them);
//   * push / count accumulate pairs of channels in the LDS box in packed 32-bit fixed point (see
//     ops_sorted.hip) and add the box to the float target with coalesced global atomics.
// Measured at config 5 (32 x 3 x 1024^2 bf16, orders [2, 3], sigma = 2): the kernels are bound by the
// number of instructions issued per tile (SQ_INSTS_VALU ~ 940 / wave for pull, VALU active ~ 50 % with
// every phase -- staging 0.17, coordinates 0.12, taps 0.10, stores 0.08 ms, set-up 0.14: profiles/
// r02_phase_split.txt -- adding up), not by HBM or by latency: prefetching the next tile's coordinates in
// persistent workgroups changed nothing.  push: LDS atomics 0.55 ms + flush (global atomics) 0.45 ms of 1.05 ms.
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace t2d {

using namespace sorted;

constexpr int NT = 256;
constexpr int TY = 32, TZ = 32, NS = TY * TZ, VPT = NS / NT;
constexpr int CAP = 64;                          // box capacity per dim
constexpr int PZ = CAP + 2;                      // row pitch (slots): rows start in different banks, quads stay 16-byte aligned
constexpr int BOXSLOTS = CAP * PZ;

struct Smem {
    int   taboff[2][CAP];
    float tabsgn[2][CAP];
    int   lo[2], hi[2];
    int   cmax[2];
    int   dmax, nout;               // nout: valid pixels of the tile whose stencil leaves the box
    unsigned long long box[BOXSLOTS];
};

template <int GM>
__device__ __forceinline__ void load_yz(const KParams &p, const float *__restrict__ grid, int64_t b, int gy, int gz, int oy, int oz, float *x)
{
    if (GM == 1) { x[0] = grid[oy]; x[1] = grid[gy + oz]; }
    else if (GM == 3) {
        x[0] = __builtin_fmaf(grid[1], (float)oz, grid[0] * (float)oy) + grid[2];
        x[1] = __builtin_fmaf(grid[4], (float)oz, grid[3] * (float)oy) + grid[5];
    } else {
        const float2 v = *reinterpret_cast<const float2 *>(grid + b * p.grid_sb + ((int64_t)oy * gz + oz) * 2);
        x[0] = v.x; x[1] = v.y;
        if (GM == 2) { x[0] += (float)oy; x[1] += (float)oz; }
    }
}

template <int K>
__device__ __forceinline__ void weights1d(float t, float *w)
{
#pragma unroll
    for (int j = 0; j <= 3; ++j) w[j] = j <= K ? bspline_w<float>(K, t - (float)j) : 0.f;
}

// the K + 1 weights of a stencil coordinate t (closed forms of splines.py:30-44 on the interval `split` produces)
template <int K>
__device__ __forceinline__ void wts(float t, float *w)
{
    if (K == 1) { w[0] = 1.f - t; w[1] = t; w[2] = 0.f; w[3] = 0.f; }
    else if (K == 2) {
        // t in [0.5, 1.5): taps at distances t, |t - 1|, 2 - t
        const float a = 1.5f - t, c = t - 0.5f, m = t - 1.f;
        w[0] = (a * a) * 0.5f; w[1] = __builtin_fmaf(-m, m, 0.75f); w[2] = (c * c) * 0.5f; w[3] = 0.f;
    } else {
        // t in [1, 2): u = t - 1, v = 1 - u; taps at distances 1 + u, u, v, 1 + v
        const float u = t - 1.f, v = 2.f - t, u2 = u * u, v2 = v * v;
        w[0] = (v2 * v) * (1.f / 6.f); w[3] = (u2 * u) * (1.f / 6.f);
        w[1] = __builtin_fmaf(u2, __builtin_fmaf(u, 0.5f, -1.f), 2.f / 3.f);
        w[2] = __builtin_fmaf(v2, __builtin_fmaf(v, 0.5f, -1.f), 2.f / 3.f);
    }
}

// derivatives of the same weights with respect to t (splines.py:90-139; order 1 outside the all-linear mode keeps the
// reference's sign quirk B-4 through tiled::wgrad1)
template <int K>
__device__ __forceinline__ void dwts(int lin, float t, float *g)
{
    if (K == 3) {
        const float u = t - 1.f, v = 2.f - t;
        g[0] = -0.5f * (v * v); g[3] = 0.5f * (u * u);
        g[1] = u * __builtin_fmaf(u, 1.5f, -2.f);
        g[2] = v * __builtin_fmaf(v, -1.5f, 2.f);
    } else if (K == 2) {
        g[0] = t - 1.5f; g[1] = (t - 1.f) * -2.f; g[2] = t - 0.5f; g[3] = 0.f;
    } else {
        g[0] = tiled::wgrad1(lin, 1, t, 0); g[1] = tiled::wgrad1(lin, 1, t, 1); g[2] = 0.f; g[3] = 0.f;
    }
}

// `split` of tile_common.hpp with the order known and a one-instruction clamp
template <int K>
__device__ __forceinline__ void splitk(float x, int &i0, float &t)
{
    const float fl = floorf(x - 0.5f * (float)(K - 1));
    t = x - fl;
    i0 = (int)__builtin_amdgcn_fmed3f(fl, -1073741824.f, 1073741824.f);
}

// channels per 8-byte slot, pack / unpack
template <typename T> struct Slot;
template <> struct Slot<float> {
    static constexpr int NC = 2;
    static constexpr unsigned SIGN = 0x80000000u;
    typedef unsigned RawQ __attribute__((ext_vector_type(4)));                 // four consecutive values of one channel, as stored
    static __device__ __forceinline__ RawQ ldq(const float *p) { typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4))); return *reinterpret_cast<const u4u *>(p); }
    // the four slots of a quad from the raw quads of the channels: word 0 of slot k = channel 0, word 1 = channel 1
    static __device__ __forceinline__ void slots(const RawQ *a, unsigned (&w)[4][2])
    {
        w[0][0] = a[0].x; w[1][0] = a[0].y; w[2][0] = a[0].z; w[3][0] = a[0].w;
        w[0][1] = a[1].x; w[1][1] = a[1].y; w[2][1] = a[1].z; w[3][1] = a[1].w;
    }
    static __device__ __forceinline__ unsigned long long pack(const float *v) { return ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]); }
    static __device__ __forceinline__ void unpack(unsigned long long s, float *v) { v[0] = __uint_as_float((unsigned)s); v[1] = __uint_as_float((unsigned)(s >> 32)); }
    static __device__ __forceinline__ void unpack2(unsigned long long s, f2 &a, f2 &b) { a = f2{ __uint_as_float((unsigned)s), __uint_as_float((unsigned)(s >> 32)) }; b = f2{ 0.f, 0.f }; }
};
template <> struct Slot<bf16_t> {
    static constexpr int NC = 4;
    static constexpr unsigned SIGN = 0x80008000u;
    typedef unsigned RawQ __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ RawQ ldq(const bf16_t *p) { typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(2))); return *reinterpret_cast<const u2u *>(p); }
    // word 0 of slot k = channels 0 | 1 << 16, word 1 = channels 2 | 3 << 16
    static __device__ __forceinline__ void slots(const RawQ *a, unsigned (&w)[4][2])
    {
        w[0][0] = __builtin_amdgcn_perm(a[1].x, a[0].x, 0x05040100u); w[1][0] = __builtin_amdgcn_perm(a[1].x, a[0].x, 0x07060302u);
        w[2][0] = __builtin_amdgcn_perm(a[1].y, a[0].y, 0x05040100u); w[3][0] = __builtin_amdgcn_perm(a[1].y, a[0].y, 0x07060302u);
        w[0][1] = __builtin_amdgcn_perm(a[3].x, a[2].x, 0x05040100u); w[1][1] = __builtin_amdgcn_perm(a[3].x, a[2].x, 0x07060302u);
        w[2][1] = __builtin_amdgcn_perm(a[3].y, a[2].y, 0x05040100u); w[3][1] = __builtin_amdgcn_perm(a[3].y, a[2].y, 0x07060302u);
    }
    static __device__ __forceinline__ unsigned long long pack(const float *v)
    {
        unsigned long long s = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) s |= (unsigned long long)Cvt<float, bf16_t>::st(v[c]).u << (16 * c);
        return s;
    }
    static __device__ __forceinline__ void unpack(unsigned long long s, float *v)
    {
        const unsigned lo = (unsigned)s, hi = (unsigned)(s >> 32);
        v[0] = __uint_as_float(lo << 16); v[1] = __uint_as_float(lo & 0xffff0000u);
        v[2] = __uint_as_float(hi << 16); v[3] = __uint_as_float(hi & 0xffff0000u);
    }
    static __device__ __forceinline__ void unpack2(unsigned long long s, f2 &a, f2 &b)
    {
        const unsigned lo = (unsigned)s, hi = (unsigned)(s >> 32);
        a = f2{ __uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u) };
        b = f2{ __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u) };
    }
};
template <> struct Slot<f16_t> {
    static constexpr int NC = 4;
    static constexpr unsigned SIGN = 0x80008000u;
    typedef unsigned RawQ __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ RawQ ldq(const f16_t *p) { typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(2))); return *reinterpret_cast<const u2u *>(p); }
    // word 0 of slot k = channels 0 | 1 << 16, word 1 = channels 2 | 3 << 16
    static __device__ __forceinline__ void slots(const RawQ *a, unsigned (&w)[4][2])
    {
        w[0][0] = __builtin_amdgcn_perm(a[1].x, a[0].x, 0x05040100u); w[1][0] = __builtin_amdgcn_perm(a[1].x, a[0].x, 0x07060302u);
        w[2][0] = __builtin_amdgcn_perm(a[1].y, a[0].y, 0x05040100u); w[3][0] = __builtin_amdgcn_perm(a[1].y, a[0].y, 0x07060302u);
        w[0][1] = __builtin_amdgcn_perm(a[3].x, a[2].x, 0x05040100u); w[1][1] = __builtin_amdgcn_perm(a[3].x, a[2].x, 0x07060302u);
        w[2][1] = __builtin_amdgcn_perm(a[3].y, a[2].y, 0x05040100u); w[3][1] = __builtin_amdgcn_perm(a[3].y, a[2].y, 0x07060302u);
    }
    static __device__ __forceinline__ unsigned long long pack(const float *v)
    {
        unsigned long long s = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const f16_t h = (f16_t)v[c]; unsigned short u; __builtin_memcpy(&u, &h, 2); s |= (unsigned long long)u << (16 * c); }
        return s;
    }
    static __device__ __forceinline__ void unpack(unsigned long long s, float *v)
    {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const unsigned short u = (unsigned short)(s >> (16 * c)); f16_t h; __builtin_memcpy(&h, &u, 2); v[c] = (float)h; }
    }
    static __device__ __forceinline__ void unpack2(unsigned long long s, f2 &a, f2 &b) { float v[4]; unpack(s, v); a = f2{ v[0], v[1] }; b = f2{ v[2], v[3] }; }
};

// pixel v of thread tid: row tid >> 3, column 4 (tid & 7) + v
__device__ __forceinline__ void px_pos(int tid, int v, int oy0, int oz0, int &oy, int &oz) { oy = oy0 + (tid >> 3); oz = oz0 + 4 * (tid & 7) + v; }

// Tile set-up: coordinates, first taps / stencil coordinates, bounding box, boundary tables.
template <int K0, int K1, int GM>
struct Tile2 {
    int lo[2], S[2];
    float t0[VPT], t1[VPT];
    int   cell[VPT];                 // first tap's slot in the box (slot 0 unless the `in` bit is set)
    unsigned valid, in, inb;

    // the coordinates of the thread's four pixels (issued, not waited for: `build` consumes them)
    static __device__ __forceinline__ void load(const KParams &p, const float *__restrict__ grid, int64_t b,
                                                int gy, int gz, int oy0, int oz0, float (&c)[VPT][2])
    {
        const int tid = opaque((int)threadIdx.x);
        if (GM == 0 && oy0 + TY <= gy && oz0 + TZ <= gz) {
            // whole tile: the thread's four pixels are 32 contiguous bytes of the grid
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            int oy, oz;
            px_pos(tid, 0, oy0, oz0, oy, oz);
            const f4u *gp = reinterpret_cast<const f4u *>(grid + b * p.grid_sb + ((int64_t)oy * gz + oz) * 2);
            f4u a, e;
            if (p.dbg & 8) { a = f4u{ (float)oy, (float)oz, (float)oy, (float)oz + 1.f }; e = f4u{ (float)oy, (float)oz + 2.f, (float)oy, (float)oz + 3.f }; }
            else { a = gp[0]; e = gp[1]; }
            c[0][0] = a.x; c[0][1] = a.y; c[1][0] = a.z; c[1][1] = a.w; c[2][0] = e.x; c[2][1] = e.y; c[3][0] = e.z; c[3][1] = e.w;
        } else {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int oy, oz;
                px_pos(tid, v, oy0, oz0, oy, oz);
                oy = oy < gy ? oy : gy - 1; oz = oz < gz ? oz : gz - 1;
                load_yz<GM>(p, grid, b, gy, gz, oy, oz, c[v]);
            }
        }
    }

    __device__ __forceinline__ void build(const KParams &p, const Lattice &L, const float (&c)[VPT][2],
                                          int gy, int gz, int oy0, int oz0, Smem &sm)
    {
        const int tid = opaque((int)threadIdx.x);
        if (tid < 2) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        if (tid == 0) sm.nout = 0;
        valid = 0; inb = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            int oy, oz;
            px_pos(tid, v, oy0, oz0, oy, oz);
            if (oy < gy && oz < gz) valid |= 1u << v;
        }
        __syncthreads();
        int i0[VPT][2];
        int mn[2] = { 0x7fffffff, 0x7fffffff }, mx[2] = { -0x7fffffff, -0x7fffffff };
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (p.extrapolate == 1 || (c[v][0] > p.mask_lo_f && c[v][0] < p.mask_hi_f[0] && c[v][1] > p.mask_lo_f && c[v][1] < p.mask_hi_f[1]))
                inb |= 1u << v;
            splitk<K0>(c[v][0], i0[v][0], t0[v]);
            splitk<K1>(c[v][1], i0[v][1], t1[v]);
            if ((valid >> v) & 1) {
#pragma unroll
                for (int d = 0; d < 2; ++d) { mn[d] = i0[v][d] < mn[d] ? i0[v][d] : mn[d]; mx[d] = i0[v][d] > mx[d] ? i0[v][d] : mx[d]; }
            }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int a = wave_min(mn[d]), e = wave_max(mx[d]);
            if ((tid & 63) == 0) { atomicMin(&sm.lo[d], a); atomicMax(&sm.hi[d], e); }
        }
        __syncthreads();
        const int kd[2] = { K0, K1 };
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            int l = sm.lo[d], h = sm.hi[d] + kd[d];
            if (h < l) { l = 0; h = 0; }
            if (d == 1) l &= ~3;                       // rows of the box start on a quad of the lattice: aligned wide accesses
            int sz_ = h - l + 1;
            if (sz_ > CAP) { l += (sz_ - CAP) / 2; sz_ = CAP; }
            lo[d] = l; S[d] = sz_;
        }
        if (tid < 2 * CAP) {
            const int d = tid >> 6, slot = tid & 63;
            if (slot < (d ? S[1] : S[0])) {
                const int l = d ? lo[1] : lo[0], sz_ = d ? S[1] : S[0];
                if (l >= (L.bound[1 + d] == B_DST1 ? 1 : 0) && l + sz_ <= L.n[1 + d]) {
                    // the box is inside the lattice along this dim: no wrapping (and no integer division)
                    sm.taboff[d][slot] = (l + slot) * L.ss[1 + d];
                    sm.tabsgn[d][slot] = 1.f;
                } else {
                    const long long pk = wrap_outofline(L.bound[1 + d], l + slot, L.n[1 + d]);
                    sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * L.ss[1 + d];
                    sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
                }
            }
        }
        in = 0; inb &= valid;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int y0 = i0[v][0] - lo[0], z0 = i0[v][1] - lo[1];
            cell[v] = 0;
            if (((valid >> v) & 1) && y0 >= 0 && y0 + K0 < S[0] && z0 >= 0 && z0 + K1 < S[1]) { in |= 1u << v; cell[v] = y0 * PZ + z0; }
        }
        if (valid & ~in) atomicAdd(&sm.nout, __popc(valid & ~in));     // (none for smooth deformations)
        __syncthreads();
    }
};

__device__ __forceinline__ Lattice lattice2d(const KParams &p, int esz, int k0, int k1)
{
    Lattice L;
    L.bound[0] = 1; L.n[0] = 1; L.ss[0] = 0; L.k[0] = 0;               // degenerate x
    L.bound[1] = p.bound[0]; L.n[1] = p.vol_n[0]; L.ss[1] = p.vol_ss[0] / esz; L.k[1] = k0;
    L.bound[2] = p.bound[1]; L.n[2] = p.vol_n[1]; L.ss[2] = p.vol_ss[1] / esz; L.k[2] = k1;
    L.lin = (k0 == 1 && k1 == 1 && p.mode == MODE_ISO1);
    return L;
}

// ---------------------------------------------------------------------------
// pull
// ---------------------------------------------------------------------------
template <typename T, int K0, int K1, int GM>
__global__ __launch_bounds__(NT, 3) void pull2d(KParams p, const T *__restrict__ vol, const float *__restrict__ grid, T *__restrict__ val,
                                             int gy, int gz, int ntz, int ntiles, DeferArgs defer)
{
    __shared__ Smem sm;
    if (p.gate_n == -1 && p.gate && *p.gate != 0) return;   // a probe of the call gave it to the bricks of the image or, an expanding field, to the generic kernel (scatter2d.hip: probe2d)
    constexpr int NC = Slot<T>::NC;
    const Lattice L = lattice2d(p, (int)sizeof(T), K0, K1);
    const int64_t b = blockIdx.x / ntiles;
    const int tile = blockIdx.x % ntiles;
    const int oy0 = (tile / ntz) * TY, oz0 = (tile % ntz) * TZ;
    Tile2<K0, K1, GM> tl;
    prof_mark(-1);
    {
        float c[VPT][2];
        Tile2<K0, K1, GM>::load(p, grid, b, gy, gz, oy0, oz0, c);
        tl.build(p, L, c, gy, gz, oy0, oz0, sm);
    }
    if (defer.flag) {                                    // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
        // (a wave pays the per-thread fallback of its slowest lane: a tile with n pixels outside costs about 1 + n / 2 tiles, the generic
        //  kernel 3.3 -- measured at config 5's shape, sigma = 6 and 8)
        bool hand_back = sm.nout > ((NS / 256) << ((p.dbg >> 9) & 7));
        // (smooth or rough: unlike in 3-D the generic gather -- 16 taps per pixel -- beats the tile's per-thread fallbacks whatever the
        //  field; 16 pixels outside the box hit most of the tile's waves: 32 x 3 x 1024^2 bf16, sigma = 8: 5.1 -> 2.0 ms, 16: 7.0 -> 3.0)
        if (hand_back && threadIdx.x == 0) defer_mark(defer, (int)blockIdx.x, tile_desc(b, 0, oy0 / TY, oz0 / TZ));
        if (hand_back && defer.desc) return;
    }
    prof_mark(0);
    // the box's columns are contiguous runs of the unit-stride dim, sign +1 (dst1: sign 0 at index 0, quirk B-3)
    const bool zlin = L.ss[2] == 1 && tl.S[1] >= 4 && tl.lo[1] >= (L.bound[2] == B_DST1 ? 1 : 0) && tl.lo[1] + tl.S[1] <= L.n[2];
    for (int cg = 0; cg < p.C; cg += NC) {
        // (opaque copies: nothing below is to be hoisted out of this loop and kept -- spilled -- across it)
        const int tid = opaque((int)threadIdx.x);
        const int nc = p.C - cg < NC ? p.C - cg : NC;
        const T *vb = vol + b * p.vol_sb + cg * p.vol_sc;
        // stage the box: slot (y, z) = the nc channels of the wrapped lattice point, sign applied
        if (p.dbg & 1) { } else
        if (zlin) {
            // rows are contiguous runs of the lattice: quads of 4 slots, one wide load per channel; all the
            // loads of the tile are in flight together
            // (values travel as stored: the boundary sign is a flip / clear of bits, no conversion; channels
            //  the image does not have repeat channel 0 and are never stored)
            {
                const int nq = (tl.S[1] + 3) >> 2;                    // the last quad is shifted to END at S_z
                constexpr int QPR = CAP / 4, NU = (CAP * QPR + NT - 1) / NT;
                typename Slot<T>::RawQ a[NU][NC]; float sg[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int e = tid + NT * u, y = e / QPR, qd = e - y * QPR;
                    const bool on = y < tl.S[0] && qd < nq;
                    const int zs = 4 * qd + 4 <= tl.S[1] ? 4 * qd : tl.S[1] - 4;
                    const unsigned off = on ? (unsigned)(sm.taboff[0][y] + tl.lo[1] + zs) : 0u;
                    sg[u] = on ? sm.tabsgn[0][y] : 0.f;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) a[u][ch] = Slot<T>::ldq(vb + (ch < nc ? ch : 0) * p.vol_sc + off);
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int e = tid + NT * u, y = e / QPR, qd = e - y * QPR;
                    if (!(y < tl.S[0] && qd < nq)) continue;
                    const int zs = 4 * qd + 4 <= tl.S[1] ? 4 * qd : tl.S[1] - 4;
                    unsigned w[4][2];
                    Slot<T>::slots(a[u], w);
                    unsigned long long s4[4];
                    const unsigned flip = sg[u] < 0.f ? Slot<T>::SIGN : 0u, m = sg[u] != 0.f ? 0xffffffffu : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s4[k] = ((unsigned long long)((w[k][1] ^ flip) & m) << 32) | ((w[k][0] ^ flip) & m);
                    unsigned long long *d = sm.box + y * PZ + zs;
                    if (!(zs & 1)) { reinterpret_cast<ulonglong2 *>(d)[0] = ulonglong2{ s4[0], s4[1] }; reinterpret_cast<ulonglong2 *>(d)[1] = ulonglong2{ s4[2], s4[3] }; }
                    else { d[0] = s4[0]; d[1] = s4[1]; d[2] = s4[2]; d[3] = s4[3]; }
                }
            }
        } else {
            // general case (the box wraps, or strided columns): slot by slot through the tables, U at a time
            const int nslot = tl.S[0] * 64;
            constexpr int U = 4;
            for (int e0 = tid; e0 < nslot; e0 += NT * U) {
                float v[U][4]; float sg[U]; bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + NT * u;
                    const int y = e >> 6, z = e & 63;
                    on[u] = e < nslot && z < tl.S[1];
                    const int off = on[u] ? sm.taboff[0][y] + sm.taboff[1][z] : 0;
                    sg[u] = on[u] ? sm.tabsgn[0][y] * sm.tabsgn[1][z] : 0.f;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) v[u][ch] = Cvt<float, T>::ld(vb[(ch < nc ? ch : 0) * p.vol_sc + off]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + NT * u;
                    if (!on[u]) continue;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) v[u][ch] = ch < nc ? v[u][ch] * sg[u] : 0.f;
                    sm.box[(e >> 6) * PZ + (e & 63)] = Slot<T>::pack(v[u]);
                }
            }
        }
        __syncthreads();
        prof_mark(1);
        float res[VPT][4];
        // NCU = channels of the slot that are computed (the slot of a 16-bit type holds 4, config 5 has 3)
        auto taps = [&](auto ncu_) {
            constexpr int NCU = decltype(ncu_)::value;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                // (samples whose stencil is not in the box read slot 0 and are redone below: no branch here)
                float w0[4], w1[4];
                float t0 = tl.t0[v], t1 = tl.t1[v];
                asm volatile("" : "+v"(t0), "+v"(t1));
                if (L.lin) { w0[0] = 1.f - t0; w0[1] = t0; w1[0] = 1.f - t1; w1[1] = t1; w0[2] = w0[3] = w1[2] = w1[3] = 0.f; }
                else { wts<K0>(t0, w0); wts<K1>(t1, w1); }
                // (volatile LDS pointer: keeps single ds_read_b64 -- a merged ds_read2_b64 costs 3x more per byte on gfx950)
                const volatile __attribute__((address_space(3))) unsigned long long *bp =
                    (const volatile __attribute__((address_space(3))) unsigned long long *)(sm.box) + opaque(tl.cell[v]);
                f2 acc01 = { 0.f, 0.f }, acc23 = { 0.f, 0.f };
#pragma unroll
                for (int i = 0; i <= K0; ++i) {
                    f2 r01 = { 0.f, 0.f }, r23 = { 0.f, 0.f };
#pragma unroll
                    for (int j = 0; j <= K1; ++j) {
                        f2 u01, u23;
                        Slot<T>::unpack2(bp[i * PZ + j], u01, u23);
                        const f2 w = { w1[j], w1[j] };
                        r01 = __builtin_elementwise_fma(w, u01, r01);
                        if (NCU == 3) r23.x = __builtin_fmaf(w1[j], u23.x, r23.x);
                        if (NCU == 4) r23 = __builtin_elementwise_fma(w, u23, r23);
                    }
                    const f2 w = { w0[i], w0[i] };
                    acc01 = __builtin_elementwise_fma(w, r01, acc01);
                    if (NCU == 3) acc23.x = __builtin_fmaf(w0[i], r23.x, acc23.x);
                    if (NCU == 4) acc23 = __builtin_elementwise_fma(w, r23, acc23);
                }
                const float m = (float)((tl.inb >> v) & 1);
                res[v][0] = acc01.x * m; res[v][1] = acc01.y * m; res[v][2] = acc23.x * m; res[v][3] = acc23.y * m;
                __builtin_amdgcn_sched_barrier(0);          // one pixel's reads in flight at a time: 48 hoisted reads would not fit the registers
            }
        };
        if (p.dbg & 2) { for (int v = 0; v < VPT; ++v) for (int ch = 0; ch < 4; ++ch) res[v][ch] = tl.t0[v]; } else
        if (NC == 4 && nc == 3) taps(std::integral_constant<int, 3>{}); else taps(std::integral_constant<int, NC>{});
        prof_mark(2);
        {
            int oy, oz;
            px_pos(tid, 0, oy0, oz0, oy, oz);
            T *ob = val + b * p.val_sb + cg * p.val_sc + (int64_t)oy * gz + oz;
            if (p.dbg & 4) { if (res[0][0] == 123.f) ob[0] = Cvt<float, T>::st(res[1][1] + res[2][2] + res[3][0] + res[0][1] + res[1][2]); } else
            if (tl.valid == 0xf) {
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) if (ch < nc) st4<T>(ob + ch * p.val_sc, make_float4(res[0][ch], res[1][ch], res[2][ch], res[3][ch]));
            } else {
#pragma unroll
                for (int v = 0; v < VPT; ++v)
                    if ((tl.valid >> v) & 1)
#pragma unroll
                        for (int ch = 0; ch < NC; ++ch) if (ch < nc) ob[ch * p.val_sc + v] = Cvt<float, T>::st(res[v][ch]);
            }
            // stencils outside the (clamped) box: gathered from global memory, one by one (the coordinates are read
            // again: nothing of this rare path stays live across the hot one)
            if (tl.in != tl.valid) {
#pragma unroll 1
                for (int v = 0; v < VPT; ++v) {
                    if (!(((tl.valid & ~tl.in) >> v) & 1)) continue;
                    float x[2]; int iy, iz; float ty, tz;
                    load_yz<GM>(p, grid, b, gy, gz, oy, oz + v, x);
                    split(K0, x[0], iy, ty); split(K1, x[1], iz, tz);
                    const float m = (float)((tl.inb >> v) & 1);
                    for (int ch = 0; ch < nc; ++ch)
                        ob[ch * p.val_sc + v] = Cvt<float, T>::st(m * tiled::gather_one_thread<T>(L, vb + ch * p.vol_sc, 0, iy, iz, 0.f, ty, tz, -1));
                }
            }
        }
        __syncthreads();
        prof_mark(3);
    }
}

// ---------------------------------------------------------------------------
// grid gradient of pull (pushpull.py:256-257), 2-D: ggrid[b,o,d] = mask * sum_c gout[b,c,o] * d/dx_d pull(vol[b,c])(x_o);
// gout == NULL: ones (the backward of count).  The tile of pull2d with the channels contracted per tap.
// ---------------------------------------------------------------------------
template <typename T, int K0, int K1, int GM>
__global__ __launch_bounds__(NT, 3) void gradc2d(KParams p, const T *__restrict__ vol, const T *__restrict__ gout, const float *__restrict__ grid,
                                              float *__restrict__ ggrid, int gy, int gz, int ntz, int ntiles, DeferArgs defer)
{
    __shared__ Smem sm;
    if (p.gate_n == -1 && p.gate && *p.gate != 0) return;   // a probe of the call gave it to the bricks of the image or, an expanding field, to the generic kernel (scatter2d.hip: probe2d)
    constexpr int NC = Slot<T>::NC;
    const Lattice L = lattice2d(p, (int)sizeof(T), K0, K1);
    const int64_t b = blockIdx.x / ntiles;
    const int tile = blockIdx.x % ntiles;
    const int oy0 = (tile / ntz) * TY, oz0 = (tile % ntz) * TZ;
    Tile2<K0, K1, GM> tl;
    prof_mark(-1);
    {
        float c[VPT][2];
        Tile2<K0, K1, GM>::load(p, grid, b, gy, gz, oy0, oz0, c);
        tl.build(p, L, c, gy, gz, oy0, oz0, sm);
    }
    if (defer.flag) {                                    // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
        // (as pull2d: more than four pixels outside the box, smooth or rough -- the generic fused kernel serves such a tile faster than
        //  the per-thread fallbacks: config 5's shape, sigma = 8: grid gradient 7.6 ms against 2.3 generic)
        const bool hand_back = sm.nout > ((NS / 256) << ((p.dbg >> 9) & 7));
        if (hand_back && threadIdx.x == 0) defer_mark(defer, (int)blockIdx.x, tile_desc(b, 0, oy0 / TY, oz0 / TZ));
        if (hand_back && defer.desc) return;
    }
    prof_mark(0);
    // the box's columns are contiguous runs of the unit-stride dim, sign +1 (dst1: sign 0 at index 0, quirk B-3)
    const bool zlin = L.ss[2] == 1 && tl.S[1] >= 4 && tl.lo[1] >= (L.bound[2] == B_DST1 ? 1 : 0) && tl.lo[1] + tl.S[1] <= L.n[2];
    float gg[VPT][2];                                            // grid gradient of the thread's pixels, all channels
#pragma unroll
    for (int v = 0; v < VPT; ++v) { gg[v][0] = 0.f; gg[v][1] = 0.f; }
    for (int cg = 0; cg < p.C; cg += NC) {
        // (opaque copies: nothing below is to be hoisted out of this loop and kept -- spilled -- across it)
        const int tid = opaque((int)threadIdx.x);
        const int nc = p.C - cg < NC ? p.C - cg : NC;
        const T *vb = vol + b * p.vol_sb + cg * p.vol_sc;
        // stage the box: slot (y, z) = the nc channels of the wrapped lattice point, sign applied
        if (p.dbg & 1) { } else
        if (zlin) {
            // rows are contiguous runs of the lattice: quads of 4 slots, one wide load per channel; all the
            // loads of the tile are in flight together
            // (values travel as stored: the boundary sign is a flip / clear of bits, no conversion; channels
            //  the image does not have repeat channel 0 and are never stored)
            {
                const int nq = (tl.S[1] + 3) >> 2;                    // the last quad is shifted to END at S_z
                constexpr int QPR = CAP / 4, NU = (CAP * QPR + NT - 1) / NT;
                typename Slot<T>::RawQ a[NU][NC]; float sg[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int e = tid + NT * u, y = e / QPR, qd = e - y * QPR;
                    const bool on = y < tl.S[0] && qd < nq;
                    const int zs = 4 * qd + 4 <= tl.S[1] ? 4 * qd : tl.S[1] - 4;
                    const unsigned off = on ? (unsigned)(sm.taboff[0][y] + tl.lo[1] + zs) : 0u;
                    sg[u] = on ? sm.tabsgn[0][y] : 0.f;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) a[u][ch] = Slot<T>::ldq(vb + (ch < nc ? ch : 0) * p.vol_sc + off);
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int e = tid + NT * u, y = e / QPR, qd = e - y * QPR;
                    if (!(y < tl.S[0] && qd < nq)) continue;
                    const int zs = 4 * qd + 4 <= tl.S[1] ? 4 * qd : tl.S[1] - 4;
                    unsigned w[4][2];
                    Slot<T>::slots(a[u], w);
                    unsigned long long s4[4];
                    const unsigned flip = sg[u] < 0.f ? Slot<T>::SIGN : 0u, m = sg[u] != 0.f ? 0xffffffffu : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s4[k] = ((unsigned long long)((w[k][1] ^ flip) & m) << 32) | ((w[k][0] ^ flip) & m);
                    unsigned long long *d = sm.box + y * PZ + zs;
                    if (!(zs & 1)) { reinterpret_cast<ulonglong2 *>(d)[0] = ulonglong2{ s4[0], s4[1] }; reinterpret_cast<ulonglong2 *>(d)[1] = ulonglong2{ s4[2], s4[3] }; }
                    else { d[0] = s4[0]; d[1] = s4[1]; d[2] = s4[2]; d[3] = s4[3]; }
                }
            }
        } else {
            // general case (the box wraps, or strided columns): slot by slot through the tables, U at a time
            const int nslot = tl.S[0] * 64;
            constexpr int U = 4;
            for (int e0 = tid; e0 < nslot; e0 += NT * U) {
                float v[U][4]; float sg[U]; bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + NT * u;
                    const int y = e >> 6, z = e & 63;
                    on[u] = e < nslot && z < tl.S[1];
                    const int off = on[u] ? sm.taboff[0][y] + sm.taboff[1][z] : 0;
                    sg[u] = on[u] ? sm.tabsgn[0][y] * sm.tabsgn[1][z] : 0.f;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) v[u][ch] = Cvt<float, T>::ld(vb[(ch < nc ? ch : 0) * p.vol_sc + off]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + NT * u;
                    if (!on[u]) continue;
#pragma unroll
                    for (int ch = 0; ch < NC; ++ch) v[u][ch] = ch < nc ? v[u][ch] * sg[u] : 0.f;
                    sm.box[(e >> 6) * PZ + (e & 63)] = Slot<T>::pack(v[u]);
                }
            }
        }
        __syncthreads();
        prof_mark(1);
        // grad_out of the group's channels at the thread's four pixels (ones: the backward of count)
        float4 gv[NC];
        {
            int oy, oz;
            px_pos(tid, 0, oy0, oz0, oy, oz);
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) {
                gv[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ch >= nc) continue;
                if (!gout) { gv[ch] = make_float4(1.f, 1.f, 1.f, 1.f); continue; }
                const T *gp = gout + b * p.val_sb + (cg + ch) * p.val_sc + (int64_t)(oy < gy ? oy : gy - 1) * gz;
                if (tl.valid == 0xf) gv[ch] = ld4<T>(gp + oz);
                else {
                    float r[4];
#pragma unroll
                    for (int v = 0; v < VPT; ++v) r[v] = Cvt<float, T>::ld(gp[oz + v < gz ? oz + v : gz - 1]);
                    gv[ch] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
        // channels contracted with grad_out per tap, then the two derivative sums of that single image (pushpull.py:256-257)
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            float w0[4], w1[4], d0[4], d1[4];
            float t0 = tl.t0[v], t1 = tl.t1[v];
            asm volatile("" : "+v"(t0), "+v"(t1));
            if (L.lin) { w0[0] = 1.f - t0; w0[1] = t0; w1[0] = 1.f - t1; w1[1] = t1; w0[2] = w0[3] = w1[2] = w1[3] = 0.f; }
            else { wts<K0>(t0, w0); wts<K1>(t1, w1); }
            dwts<K0>(L.lin, t0, d0); dwts<K1>(L.lin, t1, d1);
            float g[4];
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) g[ch] = v == 0 ? gv[ch].x : (v == 1 ? gv[ch].y : (v == 2 ? gv[ch].z : gv[ch].w));
            const volatile __attribute__((address_space(3))) unsigned long long *bp =
                (const volatile __attribute__((address_space(3))) unsigned long long *)(sm.box) + opaque(tl.cell[v]);
            float ay = 0.f, az = 0.f;
#pragma unroll
            for (int i = 0; i <= K0; ++i) {
                float r = 0.f, rz = 0.f;
#pragma unroll
                for (int j = 0; j <= K1; ++j) {
                    float u[4] = { 0.f, 0.f, 0.f, 0.f };
                    Slot<T>::unpack(bp[i * PZ + j], u);
                    float sgl = g[0] * u[0];
#pragma unroll
                    for (int ch = 1; ch < NC; ++ch) sgl = __builtin_fmaf(g[ch], u[ch], sgl);
                    r = __builtin_fmaf(w1[j], sgl, r);
                    rz = __builtin_fmaf(d1[j], sgl, rz);
                }
                ay = __builtin_fmaf(d0[i], r, ay);
                az = __builtin_fmaf(w0[i], rz, az);
            }
            if ((tl.in >> v) & 1) { gg[v][0] += ay; gg[v][1] += az; }
            __builtin_amdgcn_sched_barrier(0);
        }
        prof_mark(2);
        __syncthreads();
        prof_mark(3);
    }
    {
        const int tid = opaque((int)threadIdx.x);
        int oy, oz;
        px_pos(tid, 0, oy0, oz0, oy, oz);
        // stencils outside the (clamped) box: gathered from global memory, one by one, all channels
        if (tl.in != tl.valid) {
#pragma unroll 1
            for (int v = 0; v < VPT; ++v) {
                if (!(((tl.valid & ~tl.in) >> v) & 1)) continue;
                float x[2]; int iy, iz; float ty, tz;
                load_yz<GM>(p, grid, b, gy, gz, oy, oz + v, x);
                split(K0, x[0], iy, ty); split(K1, x[1], iz, tz);
                float a0 = 0.f, a1 = 0.f;
                for (int ch = 0; ch < p.C; ++ch) {
                    const float gvs = gout ? Cvt<float, T>::ld(gout[b * p.val_sb + ch * p.val_sc + (int64_t)oy * gz + oz + v]) : 1.f;
                    a0 = __builtin_fmaf(gvs, tiled::gather_one_thread<T>(L, vol + b * p.vol_sb + ch * p.vol_sc, 0, iy, iz, 0.f, ty, tz, 1), a0);
                    a1 = __builtin_fmaf(gvs, tiled::gather_one_thread<T>(L, vol + b * p.vol_sb + ch * p.vol_sc, 0, iy, iz, 0.f, ty, tz, 2), a1);
                }
                gg[v][0] = a0; gg[v][1] = a1;
            }
        }
        float *ob = ggrid + b * p.N * 2 + ((int64_t)oy * gz + oz) * 2;      // dense (B, *out, 2), whatever the batch stride of the grid (0: broadcast)
        float m[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) m[v] = (float)((tl.inb >> v) & 1);
        if (tl.valid == 0xf) {
            st4<float>(ob, make_float4(gg[0][0] * m[0], gg[0][1] * m[0], gg[1][0] * m[1], gg[1][1] * m[1]));
            st4<float>(ob + 4, make_float4(gg[2][0] * m[2], gg[2][1] * m[2], gg[3][0] * m[3], gg[3][1] * m[3]));
        } else {
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if ((tl.valid >> v) & 1) { ob[2 * v] = gg[v][0] * m[v]; ob[2 * v + 1] = gg[v][1] * m[v]; }
        }
    }
}



// ---------------------------------------------------------------------------
// push / count: MODE 0 values, 1 count, 2 values + count (one more target channel)
// ---------------------------------------------------------------------------
template <typename T, int K0, int K1, int GM, int MODE>
__global__ __launch_bounds__(NT, 4) void push2d(KParams p, const T *__restrict__ val, const float *__restrict__ grid, float *__restrict__ vol,
                                             int gy, int gz, int ntz, int ntiles, DeferArgs defer)
{
    __shared__ Smem sm;
    if (p.gate && *p.gate) return;                     // the bricks of the target took this call (scatter2d.hip: probe2d)
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / ntiles;
    const int tile = blockIdx.x % ntiles;
    const int oy0 = (tile / ntz) * TY, oz0 = (tile % ntz) * TZ;
    const Lattice L = lattice2d(p, 4, K0, K1);
    Tile2<K0, K1, GM> tl;
    for (int e = tid; e < BOXSLOTS; e += NT) sm.box[e] = 0ull;
    if (tid == 0) sm.dmax = 0;
    if (tid < 2) sm.cmax[tid] = 0;
    const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);
    int oy, oz;
    px_pos(tid, 0, oy0, oz0, oy, oz);
    const bool whole = oy0 + TY <= gy && oz0 + TZ <= gz;                  // whole tile: sources move as quads
    const T *sp = val + b * p.val_sb + (int64_t)(oy < gy ? oy : gy - 1) * gz + (whole ? oz : 0);
    // sources of a channel (ones for the count channel): issued early, they travel with the coordinates
    auto sources = [&](int ch) -> float4 {
        if (MODE == 1 || ch >= p.C) return make_float4(1.f, 1.f, 1.f, 1.f);
        const T *q = sp + ch * p.val_sc;
        if (whole) return ld4<T>(q);
        float r[4];
#pragma unroll
        for (int v = 0; v < VPT; ++v) r[v] = Cvt<float, T>::ld(q[oz + v < gz ? oz + v : gz - 1]);
        return make_float4(r[0], r[1], r[2], r[3]);
    };
    float4 sv0 = sources(0), sv1 = sources(1 < nch ? 1 : 0);
    prof_mark(-1);
    {
        float c[VPT][2];
        Tile2<K0, K1, GM>::load(p, grid, b, gy, gz, oy0, oz0, c);
        tl.build(p, L, c, gy, gz, oy0, oz0, sm);
    }
    if (defer.flag) {                                    // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
        bool hand_back = sm.nout > ((NS / 64) << ((p.dbg >> 9) & 7));                  // (a wave pays the per-thread fallback of its slowest lane)
        if (hand_back) hand_back = tiled::tile_smooth(p, grid, b, 2, 0, oy0, oz0, 1, TY, TZ, 1, gy, gz, sm.hi);
        if (hand_back && threadIdx.x == 0) defer_mark(defer, (int)blockIdx.x, tile_desc(b, 0, oy0 / TY, oz0 / TZ));
        if (hand_back && defer.desc) return;
    }
    prof_mark(4);
    // density: samples per first-tap cell (16-bit counters in the box, cleared again)
    {
        unsigned *cnt32 = reinterpret_cast<unsigned *>(sm.box);
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if ((tl.in >> v) & 1) atomicAdd(&cnt32[tl.cell[v] >> 1], 1u << (16 * (tl.cell[v] & 1)));
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int cv = (int)((cnt32[tl.cell[v] >> 1] >> (16 * (tl.cell[v] & 1))) & 0xffffu);
            m = ((tl.in >> v) & 1) && cv > m ? cv : m;
        }
        m = wave_max(m);
        if ((tid & 63) == 0 && m > 0) atomicMax(&sm.dmax, m);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VPT; ++v) if ((tl.in >> v) & 1) cnt32[tl.cell[v] >> 1] = 0u;
    }
    prof_mark(5);
    for (int cg = 0; cg < nch; cg += 2) {
        const bool two = cg + 1 < nch;
        float *vc0 = vol + b * p.vol_sb + cg * p.vol_sc;
        float *vc1 = two ? vc0 + p.vol_sc : vc0;
        f2 src[VPT];
        float am0 = 0.f, am1 = 0.f;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const float m = (float)((tl.inb >> v) & 1);
            const float r0 = v == 0 ? sv0.x : (v == 1 ? sv0.y : (v == 2 ? sv0.z : sv0.w)), r1 = v == 0 ? sv1.x : (v == 1 ? sv1.y : (v == 2 ? sv1.z : sv1.w));
            src[v] = f2{ r0 * m, two ? r1 * m : 0.f };
            if ((tl.in >> v) & 1) {
                const float a0 = __builtin_fabsf(src[v].x), a1 = __builtin_fabsf(src[v].y);
                am0 = (a0 > am0 || a0 != a0) ? a0 : am0; am1 = (a1 > am1 || a1 != a1) ? a1 : am1;
            }
        }
        if (cg + 2 < nch) { sv0 = sources(cg + 2); sv1 = sources(cg + 3 < nch ? cg + 3 : cg + 2); }   // next pair: in flight during the taps
        {
            const int m0 = wave_max(__float_as_int(am0)), m1 = wave_max(__float_as_int(am1));
            if ((tid & 63) == 0) { if (m0) atomicMax(&sm.cmax[0], m0); if (m1) atomicMax(&sm.cmax[1], m1); }
        }
        __syncthreads();
        prof_mark(6);
        const int hb = tiled::headroom32(L, sm.dmax);
        const int mb0 = sm.cmax[0], mb1 = sm.cmax[1];
        const bool fixedpt = hb >= 0 && (mb0 & 0x7f800000) != 0x7f800000 && (mb1 & 0x7f800000) != 0x7f800000;
        int ex0 = ((mb0 >> 23) & 0xff) - 127, ex1 = ((mb1 >> 23) & 0xff) - 127;
        ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
        const int hbc = hb < 0 ? 0 : hb;
        const f2 scale = { mb0 ? __int_as_float((127 + 29 - ex0 - hbc) << 23) : 0.f, mb1 ? __int_as_float((127 + 29 - ex1 - hbc) << 23) : 0.f };
        const float inv0 = __int_as_float((127 - 29 + ex0 + hbc) << 23), inv1 = __int_as_float((127 - 29 + ex1 + hbc) << 23);
        if (fixedpt && !(p.dbg & 1)) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!((tl.in >> v) & 1)) continue;
                float w0[4], w1[4];
                float t0 = tl.t0[v], t1 = tl.t1[v];
                asm volatile("" : "+v"(t0), "+v"(t1));          // (not to be hoisted out of the channel loop)
                if (L.lin) { w0[0] = 1.f - t0; w0[1] = t0; w1[0] = 1.f - t1; w1[1] = t1; w0[2] = w0[3] = w1[2] = w1[3] = 0.f; }
                else { wts<K0>(t0, w0); wts<K1>(t1, w1); }
                unsigned long long *bp = sm.box + opaque(tl.cell[v]);
                const f2 ss = src[v] * scale;
#pragma unroll
                for (int i = 0; i <= K0; ++i) {
                    const f2 si = ss * f2{ w0[i], w0[i] };
#pragma unroll
                    for (int j = 0; j <= K1; ++j) {
                        const f2 pr = si * f2{ w1[j], w1[j] };
                        const int q0 = tiled::cvt_rpi(pr.x), q1 = tiled::cvt_rpi(pr.y);
                        atomicAdd(bp + i * PZ + j, ((unsigned long long)(unsigned)(q1 + (q0 >> 31)) << 32) | (unsigned)q0);
                    }
                }
            }
        }
        // stencils outside the (clamped) box, or no fixed point for this tile: float atomics straight to the target,
        // one sample at a time (coordinates and sources are read again: nothing of this rare path stays live above)
        if (!fixedpt || tl.in != tl.valid) {
#pragma unroll 1
            for (int v = 0; v < VPT; ++v) {
                if (!((tl.valid >> v) & 1) || (fixedpt && ((tl.in >> v) & 1))) continue;
                float x[2]; int iy, iz; float ty, tz;
                load_yz<GM>(p, grid, b, gy, gz, oy, oz + v, x);
                split(K0, x[0], iy, ty); split(K1, x[1], iz, tz);
                const float m = (float)((tl.inb >> v) & 1);
                const float s0 = (MODE == 1 || cg >= p.C) ? m : m * Cvt<float, T>::ld(val[b * p.val_sb + cg * p.val_sc + (int64_t)oy * gz + oz + v]);
                tiled::scatter_one_thread(L, vc0, s0, 0, iy, iz, 0.f, ty, tz);
                if (two) {
                    const float s1 = (MODE == 1 || cg + 1 >= p.C) ? m : m * Cvt<float, T>::ld(val[b * p.val_sb + (cg + 1) * p.val_sc + (int64_t)oy * gz + oz + v]);
                    tiled::scatter_one_thread(L, vc1, s1, 0, iy, iz, 0.f, ty, tz);
                }
            }
        }
        __syncthreads();
        prof_mark(7);
        if (fixedpt && !(p.dbg & 2)) {
            // (a thread keeps its column z = tid & 63 and walks the rows four at a time: the slots of a batch are read
            //  before the first is used, the column's table entries once)
            const int z = tid & 63;
            if (z < tl.S[1]) {
                const int offz = sm.taboff[1][z];
                const float sgz = sm.tabsgn[1][z];
                constexpr int RS = NT / 64, UF = 4;
                for (int y0 = tid >> 6; y0 < tl.S[0]; y0 += RS * UF) {
                    long long a[UF];
#pragma unroll
                    for (int u = 0; u < UF; ++u) a[u] = y0 + u * RS < tl.S[0] ? (long long)sm.box[(y0 + u * RS) * PZ + z] : 0ll;
#pragma unroll
                    for (int u = 0; u < UF; ++u) {
                        if (a[u] == 0) continue;
                        const int y = y0 + u * RS;
                        sm.box[y * PZ + z] = 0ull;
                        const int lo_ = (int)(a[u] & 0xffffffffll);
                        const int hi_ = (int)((a[u] - (long long)lo_) >> 32);
                        const int off = sm.taboff[0][y] + offz;
                        const float sg = sm.tabsgn[0][y] * sgz;
                        if (lo_ != 0) __hip_atomic_fetch_add(vc0 + off, (float)lo_ * (inv0 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (hi_ != 0) __hip_atomic_fetch_add(vc1 + off, (float)hi_ * (inv1 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        __syncthreads();
        prof_mark(8);
        if (tid < 2) sm.cmax[tid] = 0;
    }
}

template <typename T, int GM, typename F>
static int by_orders(int k0, int k1, F &&f)
{
    using std::integral_constant;
#define IP_O(A, B) if (k0 == A && k1 == B) { f(integral_constant<int, A>{}, integral_constant<int, B>{}); return 1; }
    IP_O(1, 1) IP_O(1, 2) IP_O(1, 3) IP_O(2, 1) IP_O(2, 2) IP_O(2, 3) IP_O(3, 1) IP_O(3, 2) IP_O(3, 3)
#undef IP_O
    return 0;
}

} // namespace t2d

static bool t2d_eligible(const interpol_problem *p, const KParams &k)
{
    if (p->dim != 2 || (k.dbg & 32)) return false;
    for (int d = 0; d < 2; ++d) if (k.order[d] < 1 || k.order[d] > 3 || p->grid_shape[d] > 0x3fffffff) return false;
    const int64_t n = p->grid_shape[0] * p->grid_shape[1];
    const int64_t nt = ((p->grid_shape[0] + t2d::TY - 1) / t2d::TY) * ((p->grid_shape[1] + t2d::TZ - 1) / t2d::TZ) * p->batch;
    return n >= 4096 && nt <= 0x7fffffff && (uint64_t)n * 8ull <= 0xffffffffull;
}

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

int IP_SYM(try_tiled2d_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using namespace t2d;
    using T = IP_TT;
    if (!t2d_eligible(p, k)) return 0;
    const int gy = (int)p->grid_shape[0], gz = (int)p->grid_shape[1];
    const int nty = (gy + TY - 1) / TY, ntz = (gz + TZ - 1) / TZ, ntiles = nty * ntz;
    const dim3 g((unsigned)(ntiles * (int)p->batch));
    const Defer df(k, st, ntiles, p->batch, 1, nty, ntz, 1, TY, TZ);
    int rc;
#define IP_PULL2D(GM) rc = by_orders<T, GM>(k.order[0], k.order[1], [&](auto k0, auto k1) {                             \
        hipLaunchKernelGGL((pull2d<T, decltype(k0)::value, decltype(k1)::value, GM>), g, dim3(NT), 0, st, k, (const T *)vol, \
                           (const float *)grid, (T *)val, gy, gz, ntz, ntiles, df.args); })
    if (k.sep == 0) IP_PULL2D(0); else if (k.sep == 1) IP_PULL2D(1); else if (k.sep == 2) IP_PULL2D(2); else IP_PULL2D(3);
#undef IP_PULL2D
    if (!rc) return 0;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rd = df.desc ? DeferOps<T>::pull(k, vol, grid, val, df.tl, st) : 0;
    return rd ? rd : 1;
}

// grid gradient of pull / push (roles swapped) / count (gout == NULL), dense grids and displacement fields
int IP_SYM(try_tiled2d_gradc_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st)
{
    using namespace t2d;
    using T = IP_TT;
    if (!t2d_eligible(p, k) || (k.sep != 0 && k.sep != 2) || (k.dbg & 16)) return 0;
    const int gy = (int)p->grid_shape[0], gz = (int)p->grid_shape[1];
    const int nty = (gy + TY - 1) / TY, ntz = (gz + TZ - 1) / TZ, ntiles = nty * ntz;
    const dim3 g((unsigned)(ntiles * (int)p->batch));
    const Defer df(k, st, ntiles, p->batch, 1, nty, ntz, 1, TY, TZ);
    int rc;
#define IP_GRADC2D(GM) rc = by_orders<T, GM>(k.order[0], k.order[1], [&](auto k0, auto k1) {                             \
        hipLaunchKernelGGL((gradc2d<T, decltype(k0)::value, decltype(k1)::value, GM>), g, dim3(NT), 0, st, k, (const T *)vol, \
                           (const T *)gout, (const float *)grid, (float *)ggrid, gy, gz, ntz, ntiles, df.args); })
    if (k.sep == 0) IP_GRADC2D(0); else IP_GRADC2D(2);
#undef IP_GRADC2D
    if (!rc) return 0;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rd = df.template gradc<T>(k, gout, vol, grid, ggrid, st);
    return rd ? rd : 1;
}

int IP_SYM(try_tiled2d_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    using namespace t2d;
    using T = IP_TT;
    if (!t2d_eligible(p, k)) return 0;
    const int gy = (int)p->grid_shape[0], gz = (int)p->grid_shape[1];
    const int nty = (gy + TY - 1) / TY, ntz = (gz + TZ - 1) / TZ, ntiles = nty * ntz;
    const dim3 g((unsigned)(ntiles * (int)p->batch));
    const int mode = !val ? 1 : (k.cc ? 2 : 0);
    const Defer df(k, st, ntiles, p->batch, 1, nty, ntz, 1, TY, TZ);
    int rc;
#define IP_PUSH2D(GM, MODE) rc = by_orders<T, GM>(k.order[0], k.order[1], [&](auto k0, auto k1) {                       \
        hipLaunchKernelGGL((push2d<T, decltype(k0)::value, decltype(k1)::value, GM, MODE>), g, dim3(NT), 0, st, k, (const T *)val, \
                           (const float *)grid, (float *)vol, gy, gz, ntz, ntiles, df.args); })
#define IP_PUSH2D_GM(MODE) { if (k.sep == 0) IP_PUSH2D(0, MODE); else if (k.sep == 1) IP_PUSH2D(1, MODE); else if (k.sep == 2) IP_PUSH2D(2, MODE); else IP_PUSH2D(3, MODE); }
    if (mode == 0) IP_PUSH2D_GM(0) else if (mode == 1) IP_PUSH2D_GM(1) else IP_PUSH2D_GM(2)
#undef IP_PUSH2D_GM
#undef IP_PUSH2D
    if (!rc) return 0;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rd = df.template push<T>(k, val, grid, vol, st);
    return rd ? rd : 1;
}

#ifdef IP_PROF
#define IP_PROF_NAME3(s) interpol_debug_prof_t2d_##s
#define IP_PROF_NAME2(s) IP_PROF_NAME3(s)
extern "C" __attribute__((visibility("default"))) int IP_PROF_NAME2(IP_TSFX)(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::sorted::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::sorted::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

} // namespace ip
