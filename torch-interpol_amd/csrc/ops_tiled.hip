// ===========================================================================
// ops_tiled.hip -- LDS-tiled fast paths for 3-D float32 pull and push.
//
// Why: the generic kernels gather/scatter every tap through the vector memory
// path; with an arbitrary deformation each lane of a wave touches its own cache
// line, so a 64-tap cubic stencil costs 64 x ~64 line accesses per wave
// (measured: 12 ms pull / 400 ms push at 4x2x256^3, ~1% of the HBM roofline).
//
// Here one workgroup owns a TILE of 16x16x16 sample points:
//   1. every thread computes (floor, fraction) of its 4 samples; a block-wide
//      min/max gives the bounding box of all stencil supports;
//   2. the box (clamped to what fits in LDS) is staged global -> LDS with the
//      boundary condition ALREADY APPLIED (wrapped index and sign per box row,
//      column, slice from three small tables), so the tap loop needs no index
//      wrapping at all;
//   3. pull: taps are read from LDS (ds_read), separable FMA accumulation;
//      push: taps are accumulated into the LDS box (ds_add_f32), then the box is
//      flushed with ONE coalesced global atomic per touched lattice point
//      instead of (K+1)^3 scattered atomics per sample;
//   4. samples whose support leaves the staged box (large local deformation)
//      are collected in a list and handled tap-parallel by whole waves (lane =
//      tap) straight from / to global memory.
// One channel is resident at a time (box = up to 33x33x32 floats = 139 KiB).
//
// Numerical definition: reference interpol/nd.py:80-143 (pull), 146-213 (push);
// weights splines.py:30-80; bounds bounds.py:30-89.  Parity with the generic
// kernels / oracle is tested in tests/test_hip_parity.py.
// ===========================================================================
#include "../../include/interpol_hip.h"
#include "stencil.hpp"

namespace ip {

namespace tiled {

constexpr int TX = 16, TY = 16, TZ = 16;       // samples per tile
constexpr int NT = 1024;                        // threads per block
constexpr int VPT = TX * TY * TZ / NT;          // samples per thread (4)
constexpr int PZ = 32;                          // LDS pitch along z (box extent along z <= 32)
constexpr int CAPX = 33, CAPY = 33, CAPZ = 32;  // box extents that fit: 33*33*32 floats = 139392 B
constexpr int BOX = CAPX * CAPY * PZ;
constexpr int SLOWCAP = 512;                    // out-of-box samples handled tap-parallel per tile

struct Smem {
    float box[BOX];
    int   taboff[3][40];       // wrapped lattice offset (elements) of box row / column / slice
    float tabsgn[3][40];       // boundary sign of the same
    int   lo[3], hi[3];        // block reduction of floor indices
    int   nslow;
    unsigned short slow[SLOWCAP];
};

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// floor index and stencil coordinate t of one coordinate (nd.py:45-47 / iso1.py:13-20)
template <int K>
__device__ __forceinline__ void split(float x, int &i0, float &t)
{
    const float fl = floorf(x - 0.5f * (float)(K - 1));
    t = x - fl;
    const float flc = fl < -1073741824.f ? -1073741824.f : (fl > 1073741824.f ? 1073741824.f : fl);
    i0 = (int)flc;
}

template <int K>
__device__ __forceinline__ void weights(const KParams &p, float t, float *w)
{
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        if (K == 1 && p.mode == MODE_ISO1) w[j] = (j == 0) ? 1.f - t : t;     // iso1.py:19-20
        else w[j] = bspline_w<float>(K, t - (float)j);                          // splines.py:30-80
    }
}

// The scalars of KParams the rare (out-of-box) paths need, passed by value to
// out-of-line functions so that their register needs stay out of the hot loops.
struct Lattice {
    int bound[3], n[3], ss[3];     // boundary codes, extents, strides in elements
    int lin;                       // iso1 weights (1-t, t)
};
__device__ __forceinline__ Lattice make_lattice(const KParams &p, int K)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] >> 2; }
    L.lin = (K == 1 && p.mode == MODE_ISO1);
    return L;
}

template <int K>
__device__ __forceinline__ float weight1(int lin, float t, int j)
{
    return lin ? (j == 0 ? 1.f - t : t) : bspline_w<float>(K, t - (float)j);
}

// One sample gathered tap by tap from global memory by ONE thread (rolled loops).
template <int K>
__device__ __noinline__ float gather_one_thread(Lattice L, const float *vc, int ix, int iy, int iz, float tx, float ty, float tz)
{
    float acc = 0.f;
    for (int i = 0; i <= K; ++i) {
        const long long pk0 = wrap_outofline(L.bound[0], ix + i, L.n[0]);
        const float sx = weight1<K>(L.lin, tx, i) * (float)(int)(pk0 >> 32);
        const int offx = (int)(pk0 & 0xffffffffll) * L.ss[0];
        float pl = 0.f;
        for (int j = 0; j <= K; ++j) {
            const long long pk1 = wrap_outofline(L.bound[1], iy + j, L.n[1]);
            const float sy = weight1<K>(L.lin, ty, j) * (float)(int)(pk1 >> 32);
            const int offy = (int)(pk1 & 0xffffffffll) * L.ss[1];
            float r = 0.f;
            for (int k = 0; k <= K; ++k) {
                const long long pk2 = wrap_outofline(L.bound[2], iz + k, L.n[2]);
                const float sz = weight1<K>(L.lin, tz, k) * (float)(int)(pk2 >> 32);
                r = __builtin_fmaf(sz, vc[offx + offy + (int)(pk2 & 0xffffffffll) * L.ss[2]], r);
            }
            pl = __builtin_fmaf(sy, r, pl);
        }
        acc = __builtin_fmaf(sx, pl, acc);
    }
    return acc;
}

// One sample scattered tap by tap to global memory by ONE thread (rolled loops).
template <int K>
__device__ __noinline__ void scatter_one_thread(Lattice L, float *vc, float src, int ix, int iy, int iz, float tx, float ty, float tz)
{
    for (int i = 0; i <= K; ++i) {
        const long long pk0 = wrap_outofline(L.bound[0], ix + i, L.n[0]);
        const float sx = src * weight1<K>(L.lin, tx, i) * (float)(int)(pk0 >> 32);
        const int offx = (int)(pk0 & 0xffffffffll) * L.ss[0];
        for (int j = 0; j <= K; ++j) {
            const long long pk1 = wrap_outofline(L.bound[1], iy + j, L.n[1]);
            const float sy = sx * weight1<K>(L.lin, ty, j) * (float)(int)(pk1 >> 32);
            const int offy = (int)(pk1 & 0xffffffffll) * L.ss[1];
            for (int k = 0; k <= K; ++k) {
                const long long pk2 = wrap_outofline(L.bound[2], iz + k, L.n[2]);
                const float v = sy * weight1<K>(L.lin, tz, k) * (float)(int)(pk2 >> 32);
                __hip_atomic_fetch_add(vc + offx + offy + (int)(pk2 & 0xffffffffll) * L.ss[2], v,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Tap-parallel: the calling WAVE handles one sample, lane = tap ((K+1)^3 <= 64).
// Returns this lane's (weight * sign) and lattice offset; lanes >= (K+1)^3 get weight 0.
template <int K>
__device__ __noinline__ float tap_of_lane(Lattice L, float gx_, float gy_, float gz_, int lane, int *off_out)
{
    constexpr int K1 = K + 1;
    const int tp[3] = { lane / (K1 * K1), (lane / K1) % K1, lane % K1 };
    const float g[3] = { gx_, gy_, gz_ };
    float w = lane < K1 * K1 * K1 ? 1.f : 0.f;
    int off = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int i0; float t;
        split<K>(g[d], i0, t);
        const int j = lane < K1 * K1 * K1 ? tp[d] : 0;
        const long long pk = wrap_outofline(L.bound[d], i0 + j, L.n[d]);
        w *= weight1<K>(L.lin, t, j) * (float)(int)(pk >> 32);
        off += (int)(pk & 0xffffffffll) * L.ss[d];
    }
    *off_out = off;
    return w;
}

// ---------------------------------------------------------------------------
// Per-sample quantities are recomputed from the coordinate grid whenever they
// are needed (3 L1/L2-resident loads + a few VALU ops) instead of being held in
// registers across phases: with 4 samples per thread the persistent state
// (24+ VGPRs) pushed the tap loops over the 128-VGPR budget of a 1024-thread
// block and into scratch.
// ---------------------------------------------------------------------------
template <int K>
struct Sample {
    bool  valid, inb;          // inside the sample grid / extrapolation mask (nd.py:10-27)
    int   i0[3];               // first tap (unwrapped lattice index)
    float t[3];                // stencil coordinate (nd.py:46)
    int64_t o;                 // linear index of the sample in its batch item
};

struct TileGeom {
    int gx, gy, gz;            // sample grid extents
    int ox0, oy0, oz0;         // tile origin
};

template <int K>
__device__ __forceinline__ Sample<K> load_sample(const KParams &p, const float *__restrict__ grid, int64_t b,
                                                 const TileGeom &g, int v)
{
    const int tid = threadIdx.x;
    const int ox = g.ox0 + (tid >> 8) + 4 * v, oy = g.oy0 + ((tid >> 4) & 15), oz = g.oz0 + (tid & 15);
    Sample<K> s;
    s.valid = ox < g.gx && oy < g.gy && oz < g.gz;
    s.inb = true;
    s.o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
#pragma unroll
    for (int d = 0; d < 3; ++d) { s.i0[d] = 0; s.t[d] = 0.f; }
    if (s.valid) {
        const float *gp = grid + b * p.grid_sb + s.o * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xd = gp[d];
            if (p.extrapolate != 1) s.inb = s.inb && xd > (float)p.mask_lo && xd < (float)p.mask_hi[d];
            split<K>(xd, s.i0[d], s.t[d]);
        }
    }
    return s;
}

// Bounding box of the tile's stencil supports, clamped to what fits in LDS, and the
// boundary tables of its rows / columns / slices.
template <int K>
struct Box {
    int lo[3], S[3];

    __device__ __forceinline__ void build(const KParams &p, const float *__restrict__ grid, int64_t b,
                                          const TileGeom &g, Smem &sm)
    {
        const int tid = threadIdx.x;
        if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        if (tid == 0) sm.nslow = 0;
        __syncthreads();
        int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const Sample<K> s = load_sample<K>(p, grid, b, g, v);
            if (s.valid) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    mn[d] = s.i0[d] < mn[d] ? s.i0[d] : mn[d];
                    mx[d] = s.i0[d] > mx[d] ? s.i0[d] : mx[d];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int a = wave_min(mn[d]), c = wave_max(mx[d]);
            if ((tid & 63) == 0) { atomicMin(&sm.lo[d], a); atomicMax(&sm.hi[d], c); }
        }
        __syncthreads();
        const int cap[3] = { CAPX, CAPY, CAPZ };
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int l = sm.lo[d], h = sm.hi[d] + K;          // supports span [l, h]
            if (h < l) { l = 0; h = 0; }                  // tile without valid samples
            int sz = h - l + 1;
            if (sz > cap[d]) { l += (sz - cap[d]) / 2; sz = cap[d]; }   // keep the centre; the rest goes to the slow list
            lo[d] = l; S[d] = sz;
        }
        // boundary tables: box slot -> wrapped lattice offset and sign (bounds.py:30-89)
        // (static d: a dynamic index into the by-value KParams would push it to scratch)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int sidx = tid - 64 * d;          // one wave per dim
            if (sidx >= 0 && sidx < S[d]) {
                const long long pk = wrap_outofline(p.bound[d], lo[d] + sidx, p.vol_n[d]);
                sm.taboff[d][sidx] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] >> 2);
                sm.tabsgn[d][sidx] = (float)(int)(pk >> 32);
            }
        }
        __syncthreads();
    }

    // is the whole support of the sample inside the staged box?
    __device__ __forceinline__ bool contains(const Sample<K> &s) const
    {
        bool in = s.valid;
#pragma unroll
        for (int d = 0; d < 3; ++d) in = in && (s.i0[d] >= lo[d]) && (s.i0[d] + K < lo[d] + S[d]);
        return in;
    }

    // Classify this thread's samples: bit v of the result = "fast" (in box); the others
    // that are valid go to the block's slow list.
    __device__ __forceinline__ unsigned classify(const KParams &p, const float *__restrict__ grid, int64_t b,
                                                 const TileGeom &g, Smem &sm) const
    {
        unsigned fastmask = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const Sample<K> s = load_sample<K>(p, grid, b, g, v);
            if (contains(s)) fastmask |= 1u << v;
            else if (s.valid) {
                const int slot = atomicAdd(&sm.nslow, 1);
                if (slot < SLOWCAP) sm.slow[slot] = (unsigned short)(threadIdx.x * VPT + v);
            }
        }
        __syncthreads();
        return fastmask;
    }
};

// Stage one channel of the box: LDS[x][y][z] = sign * vol[wrapped(x,y,z)]
__device__ __forceinline__ void stage_box(const float *__restrict__ vc, const int *S, Smem &sm)
{
    const int tid = threadIdx.x;
    const int z = tid & 31;
    const bool zin = z < S[2];
    const int oz = zin ? sm.taboff[2][z] : 0;
    const float sz = zin ? sm.tabsgn[2][z] : 0.f;
    const int rows = S[0] * S[1];
    int y = tid >> 5, x = 0;
    while (y >= S[1]) { y -= S[1]; ++x; }
    for (int r = tid >> 5; r < rows; r += NT / 32) {
        if (zin) {
            const float s = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sz;
            const float v = vc[sm.taboff[0][x] + sm.taboff[1][y] + oz];
            sm.box[r * PZ + z] = v * s;
        }
        y += NT / 32;
        while (y >= S[1]) { y -= S[1]; ++x; }
    }
}

// ---------------------------------------------------------------------------
// pull
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(NT) void pull_tiled(KParams p, const float *__restrict__ vol, const float *__restrict__ grid,
                                                 float *__restrict__ val, int gx, int gy, int gz, int ntx, int nty, int ntz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    int tile = blockIdx.x;
    const int tzi = tile % ntz; tile /= ntz;
    const int tyi = tile % nty; const int txi = tile / nty;
    const TileGeom g = { gx, gy, gz, txi * TX, tyi * TY, tzi * TZ };

    Box<K> box;
    box.build(p, grid, b, g, sm);
    const unsigned fastmask = box.classify(p, grid, b, g, sm);
    const int nslow = sm.nslow;
    const float thr_lo = (float)p.mask_lo;
    const float thr_hi[3] = { (float)p.mask_hi[0], (float)p.mask_hi[1], (float)p.mask_hi[2] };
    const Lattice L = make_lattice(p, K);

    for (int c = 0; c < p.C; ++c) {
        const float *vc = vol + b * p.vol_sb + c * p.vol_sc;
        float *oc = val + b * p.val_sb + c * p.val_sc;
        __syncthreads();                               // previous channel's readers are done
        if (!(p.dbg & 1)) stage_box(vc, box.S, sm);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (p.dbg & 2) continue;
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;   // invalid, or waiting in the slow list
            const Sample<K> s = load_sample<K>(p, grid, b, g, v);
            if (!s.valid) continue;
            float acc = 0.f;
            if (fast) {
                float wx[K + 1], wy[K + 1], wz[K + 1];
                weights<K>(p, s.t[0], wx); weights<K>(p, s.t[1], wy); weights<K>(p, s.t[2], wz);
                const float *bp = sm.box + ((s.i0[0] - box.lo[0]) * box.S[1] + (s.i0[1] - box.lo[1])) * PZ + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int i = 0; i <= K; ++i) {
                    float pl = 0.f;
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        const float *rp = bp + (i * box.S[1] + j) * PZ;
                        float r = 0.f;
#pragma unroll
                        for (int k = 0; k <= K; ++k) r = __builtin_fmaf(wz[k], rp[k], r);
                        pl = __builtin_fmaf(wy[j], r, pl);
                    }
                    acc = __builtin_fmaf(wx[i], pl, acc);
                }
            } else {
                // slow list overflowed (pathological deformation): per-thread global gather
                acc = gather_one_thread<K>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2]);
            }
            if (p.extrapolate != 1) acc *= s.inb ? 1.f : 0.f;      // nd.py:139-140
            oc[s.o] = acc;
        }
        // slow list: one wave per sample, lane = tap (K <= 3: (K+1)^3 <= 64 taps)
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                const int code = sm.slow[sidx];
                const int stid = code / VPT, sv = code % VPT;
                const int sx_ = g.ox0 + (stid >> 8) + 4 * sv, sy_ = g.oy0 + ((stid >> 4) & 15), sz_ = g.oz0 + (stid & 15);
                const int64_t o = ((int64_t)sx_ * gy + sy_) * gz + sz_;
                const float *gp = grid + b * p.grid_sb + o * 3;
                int off;
                const float w = tap_of_lane<K>(L, gp[0], gp[1], gp[2], lane, &off);
                float acc = wave_sum(w != 0.f ? w * vc[off] : 0.f);
                if (lane == 0) {
                    if (p.extrapolate != 1) {
                        const bool in = gp[0] > thr_lo && gp[0] < thr_hi[0] && gp[1] > thr_lo && gp[1] < thr_hi[1] &&
                                        gp[2] > thr_lo && gp[2] < thr_hi[2];
                        acc *= in ? 1.f : 0.f;
                    }
                    oc[o] = acc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// push / count: the adjoint.
//
// LDS float atomics are the wrong tool on gfx950: measured (tools/microbench/
// lds_atomics.hip) ds_add_f32 retires 0.33 lanes/clk/CU, ds_add_u32 5.7 and
// ds_add_u64 4.6 under the same random-address pattern.  So contributions are
// accumulated in the LDS box as 64-bit FIXED POINT:
//     q = rne(src * w * 2^e),  2^e * max|src| <= 2^30  (per tile and channel)
// i.e. every contribution is rounded with an absolute error <= 2^-31 max|src|
// (far below fp32 rounding of the sums) and a 64-bit slot cannot overflow.
// The box holds 8 bytes per slot, so it is filled in passes over slabs of box
// rows (x); each tap lands in exactly one pass.  After a pass every touched
// slot is converted back to float, given the boundary sign of the slot, and
// flushed with ONE coalesced global atomic instead of (K+1)^3 scattered ones.
// Non-finite sources (inf/nan) take the per-thread float path so that IEEE
// semantics survive.
// ---------------------------------------------------------------------------
constexpr int BOX64 = BOX / 2;                 // 64-bit slots that fit in the box area

template <int K, bool COUNT>
__global__ __launch_bounds__(NT) void push_tiled(KParams p, const float *__restrict__ val, const float *__restrict__ grid,
                                                 float *__restrict__ vol, int gx, int gy, int gz, int ntx, int nty, int ntz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    unsigned long long *box64 = reinterpret_cast<unsigned long long *>(sm.box);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    int tile = blockIdx.x;
    const int tzi = tile % ntz; tile /= ntz;
    const int tyi = tile % nty; const int txi = tile / nty;
    const TileGeom g = { gx, gy, gz, txi * TX, tyi * TY, tzi * TZ };

    Box<K> box;
    box.build(p, grid, b, g, sm);
    const unsigned fastmask = box.classify(p, grid, b, g, sm);
    const int nslow = sm.nslow;
    const float thr_lo = (float)p.mask_lo;
    const float thr_hi[3] = { (float)p.mask_hi[0], (float)p.mask_hi[1], (float)p.mask_hi[2] };
    const Lattice L = make_lattice(p, K);
    // rows of the box (x) per pass, so that rows * S_y * PZ 64-bit slots fit
    const int xrows = BOX64 / (box.S[1] * PZ);
    const int npass = (box.S[0] + xrows - 1) / xrows;

    for (int c = 0; c < p.C; ++c) {
        const float *ic = COUNT ? nullptr : val + b * p.val_sb + c * p.val_sc;
        float *vc = vol + b * p.vol_sb + c * p.vol_sc;
        // ---- block maximum of the sources of this channel -> fixed-point scale -----------
        float amax = 0.f;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const Sample<K> s = load_sample<K>(p, grid, b, g, v);
            if (s.valid) {
                float sv = COUNT ? 1.f : ic[s.o];
                if (p.extrapolate != 1) sv *= s.inb ? 1.f : 0.f;                   // nd.py:201-203
                const float a = __builtin_fabsf(sv);
                amax = (a > amax || a != a) ? a : amax;                            // NaN sticks
            }
        }
        __syncthreads();                               // previous channel is completely flushed
        if (tid == 0) sm.hi[0] = 0;                    // (lo/hi are free after Box::build)
        __syncthreads();
        {
            int bits = __float_as_int(amax);           // non-negative floats (and NaN) order like ints
            bits = wave_max(bits);
            if ((tid & 63) == 0) atomicMax(&sm.hi[0], bits);
        }
        __syncthreads();
        const int mbits = sm.hi[0];
        if (mbits == 0) continue;                      // nothing to splat in this tile / channel
        const bool finite = (mbits & 0x7f800000) != 0x7f800000;
        // 2^e * max <= 2^30 : e = 29 - exponent(max)
        int ex = ((mbits >> 23) & 0xff) - 127;
        ex = ex < -90 ? -90 : ex;                      // denormal / tiny maxima: keep 2^e finite
        const float scale = __int_as_float((127 + 29 - ex) << 23);
        const float inv_scale = __int_as_float((127 - 29 + ex) << 23);

        if (!finite || nslow > SLOWCAP) {
            // pathological tile (non-finite data, or the deformation does not fit the box):
            // per-thread float atomics straight to global memory
#pragma unroll 1
            for (int v = 0; v < VPT; ++v) {
                const Sample<K> s = load_sample<K>(p, grid, b, g, v);
                if (!s.valid || (p.dbg & 2)) continue;
                float sv = COUNT ? 1.f : ic[s.o];
                if (p.extrapolate != 1) sv *= s.inb ? 1.f : 0.f;
                scatter_one_thread<K>(L, vc, sv, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2]);
            }
            continue;
        }

        // ---- slow list: one wave per sample, lane = tap, one global atomic per lane ----
        if (nslow > 0) {
            const int wave = tid >> 6, lane = tid & 63;
            for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                const int code = sm.slow[sidx];
                const int stid = code / VPT, sv = code % VPT;
                const int sx_ = g.ox0 + (stid >> 8) + 4 * sv, sy_ = g.oy0 + ((stid >> 4) & 15), sz_ = g.oz0 + (stid & 15);
                const int64_t o = ((int64_t)sx_ * gy + sy_) * gz + sz_;
                const float *gp = grid + b * p.grid_sb + o * 3;
                float sv_ = COUNT ? 1.f : ic[o];
                if (p.extrapolate != 1) {
                    const bool in = gp[0] > thr_lo && gp[0] < thr_hi[0] && gp[1] > thr_lo && gp[1] < thr_hi[1] &&
                                    gp[2] > thr_lo && gp[2] < thr_hi[2];
                    sv_ *= in ? 1.f : 0.f;
                }
                int off;
                const float w = tap_of_lane<K>(L, gp[0], gp[1], gp[2], lane, &off);
                if (lane < (K + 1) * (K + 1) * (K + 1))
                    __hip_atomic_fetch_add(vc + off, w * sv_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // ---- passes over slabs of box rows ---------------------------------------------
        for (int ps = 0; ps < npass; ++ps) {
            const int x_lo = ps * xrows;
            const int x_n = (box.S[0] - x_lo) < xrows ? (box.S[0] - x_lo) : xrows;
            const int slab = x_n * box.S[1] * PZ;
            __syncthreads();                           // previous pass is flushed
            for (int e = tid; e < slab; e += NT) box64[e] = 0ull;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!((fastmask >> v) & 1) || (p.dbg & 2)) continue;
                const Sample<K> s = load_sample<K>(p, grid, b, g, v);
                const int bx = s.i0[0] - box.lo[0] - x_lo;              // row of tap i = 0 inside this slab
                if (bx + K < 0 || bx >= x_n) continue;
                float sv = COUNT ? 1.f : ic[s.o];
                if (p.extrapolate != 1) sv *= s.inb ? 1.f : 0.f;
                float wx[K + 1], wy[K + 1], wz[K + 1];
                weights<K>(p, s.t[0], wx); weights<K>(p, s.t[1], wy); weights<K>(p, s.t[2], wz);
                const float ss = sv * scale;
                unsigned long long *bp = box64 + (bx * box.S[1] + (s.i0[1] - box.lo[1])) * PZ + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int i = 0; i <= K; ++i) {
                    if (bx + i < 0 || bx + i >= x_n) continue;
                    const float si = ss * wx[i];
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        unsigned long long *rp = bp + (i * box.S[1] + j) * PZ;
                        const float sj = si * wy[j];
#pragma unroll
                        for (int k = 0; k <= K; ++k) {
                            const int q = __float2int_rn(sj * wz[k]);
                            atomicAdd(rp + k, (unsigned long long)(long long)q);      // ds_add_u64
                        }
                    }
                }
            }
            __syncthreads();
            // flush the slab: fixed point -> float, slot sign, one coalesced global atomic per touched slot
            if (!(p.dbg & 1)) {
                const int z = tid & 31;
                const bool zin = z < box.S[2];
                const int oz_ = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                const int rows = x_n * box.S[1];
                int y = tid >> 5, x = x_lo;
                while (y >= box.S[1]) { y -= box.S[1]; ++x; }
                for (int r = tid >> 5; r < rows; r += NT / 32) {
                    if (zin) {
                        const long long a = (long long)box64[r * PZ + z];
                        if (a != 0) {
                            const float sgn = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sz;
                            const float f = (float)((double)a * (double)inv_scale);
                            __hip_atomic_fetch_add(vc + sm.taboff[0][x] + sm.taboff[1][y] + oz_, f * sgn,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    y += NT / 32;
                    while (y >= box.S[1]) { y -= box.S[1]; ++x; }
                }
            }
        }
    }
}

template <int K>
static int launch_push_tiled(const KParams &k, const void *val, const void *grid, void *vol, int B,
                             const int64_t *gshape, hipStream_t st)
{
    const int gx = (int)gshape[0], gy = (int)gshape[1], gz = (int)gshape[2];
    const int ntx = (gx + TX - 1) / TX, nty = (gy + TY - 1) / TY, ntz = (gz + TZ - 1) / TZ;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)push_tiled<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)push_tiled<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 g((unsigned)(ntx * nty * ntz), (unsigned)B);
    if (val)
        hipLaunchKernelGGL((push_tiled<K, false>), g, dim3(NT), sizeof(Smem), st,
                           k, (const float *)val, (const float *)grid, (float *)vol, gx, gy, gz, ntx, nty, ntz);
    else
        hipLaunchKernelGGL((push_tiled<K, true>), g, dim3(NT), sizeof(Smem), st,
                           k, (const float *)nullptr, (const float *)grid, (float *)vol, gx, gy, gz, ntx, nty, ntz);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

template <int K>
static int launch_pull_tiled(const KParams &k, const void *vol, const void *grid, void *val, int B,
                             const int64_t *gshape, hipStream_t st)
{
    const int gx = (int)gshape[0], gy = (int)gshape[1], gz = (int)gshape[2];
    const int ntx = (gx + TX - 1) / TX, nty = (gy + TY - 1) / TY, ntz = (gz + TZ - 1) / TZ;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)pull_tiled<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((pull_tiled<K>), dim3((unsigned)(ntx * nty * ntz), (unsigned)B), dim3(NT), sizeof(Smem), st,
                       k, (const float *)vol, (const float *)grid, (float *)val, gx, gy, gz, ntx, nty, ntz);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

} // namespace tiled

// The tiled kernels cover: 3-D, float32, isotropic order 1..3, any bound / extrapolate,
// batch <= 65535, lattice spatially addressable with 4-byte-aligned strides.
static bool tiled_eligible(const interpol_problem *p, const KParams &k)
{
    if (p->dim != 3 || p->dtype != INTERPOL_F32) return false;
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2]) return false;
    if (k.order[0] < 1 || k.order[0] > 3) return false;
    if (p->batch > 65535) return false;
    for (int d = 0; d < 3; ++d) if (p->grid_shape[d] > 0x7fffffff / 4) return false;
    // tiny problems: the generic kernel has less fixed cost
    if (p->grid_shape[0] * p->grid_shape[1] * p->grid_shape[2] < 4096) return false;
    return true;
}

int try_fast_pull(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    if (!tiled_eligible(p, k)) return 0;
    switch (k.order[0]) {
    case 1: return tiled::launch_pull_tiled<1>(k, vol, grid, val, (int)p->batch, p->grid_shape, st);
    case 2: return tiled::launch_pull_tiled<2>(k, vol, grid, val, (int)p->batch, p->grid_shape, st);
    case 3: return tiled::launch_pull_tiled<3>(k, vol, grid, val, (int)p->batch, p->grid_shape, st);
    default: return 0;
    }
}

// `vol` is the (already zero-filled or accumulating) float target; `val` == NULL means count.
int try_fast_push(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    if (!tiled_eligible(p, k)) return 0;
    switch (k.order[0]) {
    case 1: return tiled::launch_push_tiled<1>(k, val, grid, vol, (int)p->batch, p->grid_shape, st);
    case 2: return tiled::launch_push_tiled<2>(k, val, grid, vol, (int)p->batch, p->grid_shape, st);
    case 3: return tiled::launch_push_tiled<3>(k, val, grid, vol, (int)p->batch, p->grid_shape, st);
    default: return 0;
    }
}

} // namespace ip
