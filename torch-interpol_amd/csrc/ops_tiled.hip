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

// ---------------------------------------------------------------------------
// Common prologue: coordinates, bounding box, boundary tables.
// On exit: lo[d] = first lattice index of the box, S[d] = extents.
// ---------------------------------------------------------------------------
template <int K>
struct Tile {
    int   i0[VPT][3];
    float t[VPT][3];
    bool  valid[VPT];
    bool  inb[VPT];            // extrapolation mask of the sample (nd.py:10-27)
    int   lo[3], S[3];

    __device__ __forceinline__ void prologue(const KParams &p, const float *__restrict__ grid, int64_t b,
                                             int gx, int gy, int gz, int ox0, int oy0, int oz0, Smem &sm)
    {
        const int tid = threadIdx.x;
        if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        if (tid == 0) sm.nslow = 0;
        __syncthreads();
        const int tz = tid & 15, ty = (tid >> 4) & 15, tx0 = tid >> 8;
        int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int ox = ox0 + tx0 + 4 * v, oy = oy0 + ty, oz = oz0 + tz;
            valid[v] = ox < gx && oy < gy && oz < gz;
            inb[v] = true;
            if (valid[v]) {
                const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
                const float *gp = grid + b * p.grid_sb + o * 3;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float xd = gp[d];
                    if (p.extrapolate != 1) inb[v] = inb[v] && xd > (float)p.mask_lo && xd < (float)p.mask_hi[d];
                    split<K>(xd, i0[v][d], t[v][d]);
                    mn[d] = i0[v][d] < mn[d] ? i0[v][d] : mn[d];
                    mx[d] = i0[v][d] > mx[d] ? i0[v][d] : mx[d];
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
            int s = h - l + 1;
            if (s > cap[d]) { l += (s - cap[d]) / 2; s = cap[d]; }   // keep the centre; the rest goes to the slow list
            lo[d] = l; S[d] = s;
        }
        // boundary tables: box slot -> wrapped lattice offset and sign (bounds.py:30-89)
        // (static d: a dynamic index into the by-value KParams would push it to scratch)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int sidx = tid - 64 * d;          // one wave per dim
            if (sidx >= 0 && sidx < S[d]) {
                const int i = lo[d] + sidx;
                const long long pk = wrap_outofline(p.bound[d], i, p.vol_n[d]);
                sm.taboff[d][sidx] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] >> 2);
                sm.tabsgn[d][sidx] = (float)(int)(pk >> 32);
            }
        }
        __syncthreads();
    }

    // is the whole support of sample v inside the staged box?
    __device__ __forceinline__ bool inbox(int v) const
    {
        bool in = valid[v];
#pragma unroll
        for (int d = 0; d < 3; ++d) in = in && (i0[v][d] >= lo[d]) && (i0[v][d] + K < lo[d] + S[d]);
        return in;
    }
    __device__ __forceinline__ int base(int v) const
    {
        return ((i0[v][0] - lo[0]) * S[1] + (i0[v][1] - lo[1])) * PZ + (i0[v][2] - lo[2]);
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
    const int ox0 = txi * TX, oy0 = tyi * TY, oz0 = tzi * TZ;

    Tile<K> T;
    T.prologue(p, grid, b, gx, gy, gz, ox0, oy0, oz0, sm);

    // classify: fast (in box) or slow (list)
    bool fast[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        fast[v] = T.inbox(v);
        if (T.valid[v] && !fast[v]) {
            const int slot = atomicAdd(&sm.nslow, 1);
            if (slot < SLOWCAP) sm.slow[slot] = (unsigned short)(tid * VPT + v);
        }
    }
    const int tz = tid & 15, ty = (tid >> 4) & 15, tx0 = tid >> 8;
    const float thr_lo = (float)p.mask_lo;
    const float thr_hi[3] = { (float)p.mask_hi[0], (float)p.mask_hi[1], (float)p.mask_hi[2] };

    for (int c = 0; c < p.C; ++c) {
        const float *vc = vol + b * p.vol_sb + c * p.vol_sc;
        float *oc = val + b * p.val_sb + c * p.val_sc;
        __syncthreads();                               // previous channel's readers are done
        stage_box(vc, T.S, sm);
        __syncthreads();
        const int nslow = sm.nslow;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!T.valid[v]) continue;
            const int ox = ox0 + tx0 + 4 * v, oy = oy0 + ty, oz = oz0 + tz;
            const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
            if (fast[v] || nslow > SLOWCAP) {
                float acc = 0.f;
                if (fast[v]) {
                    float wx[K + 1], wy[K + 1], wz[K + 1];
                    weights<K>(p, T.t[v][0], wx); weights<K>(p, T.t[v][1], wy); weights<K>(p, T.t[v][2], wz);
                    const float *bp = sm.box + T.base(v);
#pragma unroll
                    for (int i = 0; i <= K; ++i) {
                        float pl = 0.f;
#pragma unroll
                        for (int j = 0; j <= K; ++j) {
                            const float *rp = bp + (i * T.S[1] + j) * PZ;
                            float r = 0.f;
#pragma unroll
                            for (int k = 0; k <= K; ++k) r = __builtin_fmaf(wz[k], rp[k], r);
                            pl = __builtin_fmaf(wy[j], r, pl);
                        }
                        acc = __builtin_fmaf(wx[i], pl, acc);
                    }
                } else {
                    // slow list overflowed (pathological deformation): per-thread global gather.
                    // Rolled loops, weights recomputed on the fly (no register arrays indexed dynamically).
                    const bool lin = (K == 1 && p.mode == MODE_ISO1);
                    for (int i = 0; i <= K; ++i) {
                        const int ix = T.i0[v][0] + i;
                        const long long pk0 = wrap_outofline(p.bound[0], ix, p.vol_n[0]);
                        const int offx = (int)(pk0 & 0xffffffffll) * (p.vol_ss[0] >> 2);
                        const float wi = lin ? (i == 0 ? 1.f - T.t[v][0] : T.t[v][0]) : bspline_w<float>(K, T.t[v][0] - (float)i);
                        const float sx = wi * (float)(int)(pk0 >> 32);
                        float pl = 0.f;
                        for (int j = 0; j <= K; ++j) {
                            const int iy = T.i0[v][1] + j;
                            const long long pk1 = wrap_outofline(p.bound[1], iy, p.vol_n[1]);
                            const int offy = (int)(pk1 & 0xffffffffll) * (p.vol_ss[1] >> 2);
                            const float wj = lin ? (j == 0 ? 1.f - T.t[v][1] : T.t[v][1]) : bspline_w<float>(K, T.t[v][1] - (float)j);
                            const float sy = wj * (float)(int)(pk1 >> 32);
                            float r = 0.f;
                            for (int k = 0; k <= K; ++k) {
                                const int iz = T.i0[v][2] + k;
                                const long long pk2 = wrap_outofline(p.bound[2], iz, p.vol_n[2]);
                                const int offz = (int)(pk2 & 0xffffffffll) * (p.vol_ss[2] >> 2);
                                const float wk = lin ? (k == 0 ? 1.f - T.t[v][2] : T.t[v][2]) : bspline_w<float>(K, T.t[v][2] - (float)k);
                                const float sz = wk * (float)(int)(pk2 >> 32);
                                r = __builtin_fmaf(sz, vc[offx + offy + offz], r);
                            }
                            pl = __builtin_fmaf(sy, r, pl);
                        }
                        acc = __builtin_fmaf(sx, pl, acc);
                    }
                }
                if (p.extrapolate != 1) acc *= T.inb[v] ? 1.f : 0.f;      // nd.py:139-140
                oc[o] = acc;
            }
        }
        // slow list: one wave per sample, lane = tap (K <= 3: (K+1)^3 <= 64 taps)
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            constexpr int K1 = K + 1;
            const int li = lane / (K1 * K1), lj = (lane / K1) % K1, lk = lane % K1;
            const bool tap = lane < K1 * K1 * K1;
            for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                const int code = sm.slow[sidx];
                const int stid = code / VPT, sv = code % VPT;
                const int sx_ = ox0 + (stid >> 8) + 4 * sv, sy_ = oy0 + ((stid >> 4) & 15), sz_ = oz0 + (stid & 15);
                const int64_t o = ((int64_t)sx_ * gy + sy_) * gz + sz_;
                const float *gp = grid + b * p.grid_sb + o * 3;
                int i0[3]; float t[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) split<K>(gp[d], i0[d], t[d]);
                float contrib = 0.f;
                if (tap) {
                    const int tp[3] = { li, lj, lk };
                    float w = 1.f; int off = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const int i = i0[d] + tp[d];
                        float wd = (K == 1 && p.mode == MODE_ISO1) ? (tp[d] == 0 ? 1.f - t[d] : t[d])
                                                                   : bspline_w<float>(K, t[d] - (float)tp[d]);
                        const long long pk = wrap_outofline(p.bound[d], i, p.vol_n[d]);
                        w *= wd * (float)(int)(pk >> 32);
                        off += (int)(pk & 0xffffffffll) * (p.vol_ss[d] >> 2);
                    }
                    contrib = w * vc[off];
                }
                float acc = wave_sum(contrib);
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
// push / count: the adjoint.  Contributions are accumulated in the LDS box with
// ds_add_f32 (unsigned by the boundary: the sign belongs to the box slot and is
// applied once at flush time), then every touched slot is flushed with one
// global atomic, consecutive lanes -> consecutive addresses.
// ---------------------------------------------------------------------------
template <int K, bool COUNT>
__global__ __launch_bounds__(NT) void push_tiled(KParams p, const float *__restrict__ val, const float *__restrict__ grid,
                                                 float *__restrict__ vol, int gx, int gy, int gz, int ntx, int nty, int ntz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    int tile = blockIdx.x;
    const int tzi = tile % ntz; tile /= ntz;
    const int tyi = tile % nty; const int txi = tile / nty;
    const int ox0 = txi * TX, oy0 = tyi * TY, oz0 = tzi * TZ;

    Tile<K> T;
    T.prologue(p, grid, b, gx, gy, gz, ox0, oy0, oz0, sm);

    bool fast[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        fast[v] = T.inbox(v);
        if (T.valid[v] && !fast[v]) {
            const int slot = atomicAdd(&sm.nslow, 1);
            if (slot < SLOWCAP) sm.slow[slot] = (unsigned short)(tid * VPT + v);
        }
    }
    const int tz = tid & 15, ty = (tid >> 4) & 15, tx0 = tid >> 8;
    const float thr_lo = (float)p.mask_lo;
    const float thr_hi[3] = { (float)p.mask_hi[0], (float)p.mask_hi[1], (float)p.mask_hi[2] };
    const int boxn = T.S[0] * T.S[1] * PZ;

    for (int c = 0; c < p.C; ++c) {
        const float *ic = COUNT ? nullptr : val + b * p.val_sb + c * p.val_sc;
        float *vc = vol + b * p.vol_sb + c * p.vol_sc;
        __syncthreads();                               // previous channel's flush is done
        for (int e = tid; e < boxn; e += NT) sm.box[e] = 0.f;
        __syncthreads();
        const int nslow = sm.nslow;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!T.valid[v]) continue;
            const int ox = ox0 + tx0 + 4 * v, oy = oy0 + ty, oz = oz0 + tz;
            const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
            if (fast[v] || nslow > SLOWCAP) {
                float src = COUNT ? 1.f : ic[o];
                if (p.extrapolate != 1) src *= T.inb[v] ? 1.f : 0.f;              // nd.py:201-203
                if (fast[v]) {
                    float wx[K + 1], wy[K + 1], wz[K + 1];
                    weights<K>(p, T.t[v][0], wx); weights<K>(p, T.t[v][1], wy); weights<K>(p, T.t[v][2], wz);
                    float *bp = sm.box + T.base(v);
#pragma unroll
                    for (int i = 0; i <= K; ++i) {
                        const float si = src * wx[i];
#pragma unroll
                        for (int j = 0; j <= K; ++j) {
                            float *rp = bp + (i * T.S[1] + j) * PZ;
                            const float sj = si * wy[j];
#pragma unroll
                            for (int k = 0; k <= K; ++k) atomicAdd(rp + k, sj * wz[k]);      // ds_add_f32
                        }
                    }
                } else {
                    // slow list overflowed: per-thread global scatter
                    const bool lin = (K == 1 && p.mode == MODE_ISO1);
                    for (int i = 0; i <= K; ++i) {
                        const long long pk0 = wrap_outofline(p.bound[0], T.i0[v][0] + i, p.vol_n[0]);
                        const float wi = lin ? (i == 0 ? 1.f - T.t[v][0] : T.t[v][0]) : bspline_w<float>(K, T.t[v][0] - (float)i);
                        const float sx = src * wi * (float)(int)(pk0 >> 32);
                        const int offx = (int)(pk0 & 0xffffffffll) * (p.vol_ss[0] >> 2);
                        for (int j = 0; j <= K; ++j) {
                            const long long pk1 = wrap_outofline(p.bound[1], T.i0[v][1] + j, p.vol_n[1]);
                            const float wj = lin ? (j == 0 ? 1.f - T.t[v][1] : T.t[v][1]) : bspline_w<float>(K, T.t[v][1] - (float)j);
                            const float sy = sx * wj * (float)(int)(pk1 >> 32);
                            const int offy = (int)(pk1 & 0xffffffffll) * (p.vol_ss[1] >> 2);
                            for (int k = 0; k <= K; ++k) {
                                const long long pk2 = wrap_outofline(p.bound[2], T.i0[v][2] + k, p.vol_n[2]);
                                const float wk = lin ? (k == 0 ? 1.f - T.t[v][2] : T.t[v][2]) : bspline_w<float>(K, T.t[v][2] - (float)k);
                                const int offz = (int)(pk2 & 0xffffffffll) * (p.vol_ss[2] >> 2);
                                __hip_atomic_fetch_add(vc + offx + offy + offz, sy * wk * (float)(int)(pk2 >> 32),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
            }
        }
        // slow list: one wave per sample, lane = tap, one global atomic per lane
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            constexpr int K1 = K + 1;
            const int li = lane / (K1 * K1), lj = (lane / K1) % K1, lk = lane % K1;
            const bool tap = lane < K1 * K1 * K1;
            for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                const int code = sm.slow[sidx];
                const int stid = code / VPT, sv = code % VPT;
                const int sx_ = ox0 + (stid >> 8) + 4 * sv, sy_ = oy0 + ((stid >> 4) & 15), sz_ = oz0 + (stid & 15);
                const int64_t o = ((int64_t)sx_ * gy + sy_) * gz + sz_;
                const float *gp = grid + b * p.grid_sb + o * 3;
                float src = COUNT ? 1.f : ic[o];
                if (p.extrapolate != 1) {
                    const bool in = gp[0] > thr_lo && gp[0] < thr_hi[0] && gp[1] > thr_lo && gp[1] < thr_hi[1] &&
                                    gp[2] > thr_lo && gp[2] < thr_hi[2];
                    src *= in ? 1.f : 0.f;
                }
                if (tap) {
                    const int tp[3] = { li, lj, lk };
                    float w = src; int off = 0;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        int i0; float t;
                        split<K>(gp[d], i0, t);
                        const float wd = (K == 1 && p.mode == MODE_ISO1) ? (tp[d] == 0 ? 1.f - t : t)
                                                                         : bspline_w<float>(K, t - (float)tp[d]);
                        const long long pk = wrap_outofline(p.bound[d], i0 + tp[d], p.vol_n[d]);
                        w *= wd * (float)(int)(pk >> 32);
                        off += (int)(pk & 0xffffffffll) * (p.vol_ss[d] >> 2);
                    }
                    __hip_atomic_fetch_add(vc + off, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        // flush: LDS box -> global, sign of the slot applied here
        {
            const int z = tid & 31;
            const bool zin = z < T.S[2];
            const int oz_ = zin ? sm.taboff[2][z] : 0;
            const float sz = zin ? sm.tabsgn[2][z] : 0.f;
            const int rows = T.S[0] * T.S[1];
            int y = tid >> 5, x = 0;
            while (y >= T.S[1]) { y -= T.S[1]; ++x; }
            for (int r = tid >> 5; r < rows; r += NT / 32) {
                if (zin) {
                    const float a = sm.box[r * PZ + z];
                    if (a != 0.f) {
                        const float s = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sz;
                        __hip_atomic_fetch_add(vc + sm.taboff[0][x] + sm.taboff[1][y] + oz_, a * s,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                y += NT / 32;
                while (y >= T.S[1]) { y -= T.S[1]; ++x; }
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
