// Target-stationary splatting ("bricks") for EXPANDING deformations (3-D, fp32).
//
// push / count scatter every sample to the (K+1)^3 lattice points of its stencil (nd.py:146-213).
// When the deformation expands -- splatting 128^3 sources into a 512^3 target, SURVEY config 4 --
// neighbouring samples own disjoint target voxels: a sample-stationary tile has nothing to merge
// in LDS and every contribution becomes a global atomic (2.1 G of them for 8 sources, 32 ms).
// Here the TARGET is cut into 16^3 bricks and each brick is owned by one workgroup:
//   1. count : every sample adds 1 to the counter of each brick its stencil overlaps (<= 8);
//              border samples (stencil leaving the lattice: boundary wrap) are left to a
//              separate walk that splats them tap-parallel with float atomics; per-channel
//              max|src| comes from a two-stage reduction
//   2. scan  : exclusive prefix of the counters -> list offsets
//   3. fill  : the same walk writes the sample ids into the bricks' lists
//   4. brick : a workgroup accumulates its brick in LDS -- 64-bit fixed point, ds_add_u64, all
//              sources of the batch -- from the samples of its list (one THREAD per sample: its
//              weights are computed once, then the taps that fall inside the brick are walked)
//              and adds the brick to the target with plain coalesced read-modify-writes: it is
//              the only writer.
// Weights, first-tap indices and the extrapolation mask are those of csrc/stencil.hpp.
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include <hip/hip_runtime.h>

namespace ip {
namespace {

constexpr int BS = 16, BSLOTS = BS * BS * BS;
constexpr int HDR = 16;                       // workspace header (ints): [0..7] max|src| bits per channel

struct Bricks {
    int nb[3];                                // bricks per dim
    int per_target;                           // nb[0] * nb[1] * nb[2]
    int shared;                               // batch stride 0: one target for the whole batch
    unsigned N;                               // samples per batch item
    int *hdr, *counts, *offsets, *cursor;     // workspace
    unsigned *list;                           // per-brick sample lists
    int *partial;                             // 8 x MAXPART partial maxima
};

// first tap and stencil coordinate of one dim (nd.py:45-46; iso0.py:12 rounds half to even)
__device__ __forceinline__ void split1(const KParams &p, int d, float x, int &i0, float &t)
{
    const int k = p.order[d];
    const float fl = (k == 0 && p.mode == MODE_ISO0) ? rintf(x) : floorf(x - 0.5f * (float)(k - 1));
    t = x - fl;
    const float flc = fl < -1073741824.f ? -1073741824.f : (fl > 1073741824.f ? 1073741824.f : fl);
    i0 = (int)flc;
}

__device__ __forceinline__ float weight_of(const KParams &p, int d, float t, int j)
{
    const int k = p.order[d];
    if (k == 1 && p.mode == MODE_ISO1) return j == 0 ? 1.f - t : t;            // iso1.py:19-20
    return bspline_w<float>(k, t - (float)j);                                   // splines.py:30-80
}

// the K + 1 weights of a stencil coordinate: closed forms of splines.py:30-44 on the interval of `split1` for the
// quadratic and cubic splines, the generic evaluation otherwise
template <int K>
__device__ __forceinline__ void stencil_weights(const KParams &p, int d, float t, float *w)
{
    if (K == 3) {
        const float u = t - 1.f, v = 2.f - t, u2 = u * u, v2 = v * v;          // t in [1, 2)
        w[0] = (v2 * v) * (1.f / 6.f); w[3] = (u2 * u) * (1.f / 6.f);
        w[1] = __builtin_fmaf(u2, __builtin_fmaf(u, 0.5f, -1.f), 2.f / 3.f);
        w[2] = __builtin_fmaf(v2, __builtin_fmaf(v, 0.5f, -1.f), 2.f / 3.f);
    } else if (K == 2) {
        const float a = 1.5f - t, c = t - 0.5f, m = t - 1.f;                     // t in [0.5, 1.5)
        w[0] = (a * a) * 0.5f; w[1] = __builtin_fmaf(-m, m, 0.75f); w[2] = (c * c) * 0.5f;
    } else {
#pragma unroll
        for (int j = 0; j <= K; ++j) w[j] = weight_of(p, d, t, j);
    }
}

// -> 0: masked out (contributes nothing), 1: interior (whole stencil inside the lattice, signs +1), 2: border
__device__ __forceinline__ int classify(const KParams &p, const float *x, int *i0, float *t)
{
    bool inb = true, inside = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (p.extrapolate != 1) inb = inb && x[d] > p.mask_lo_f && x[d] < p.mask_hi_f[d];   // nd.py:10-27
        split1(p, d, x[d], i0[d], t[d]);
        inside = inside && i0[d] >= (p.bound[d] == B_DST1 ? 1 : 0) && i0[d] + p.order[d] < p.vol_n[d];
    }
    return !inb ? 0 : (inside ? 1 : 2);
}

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

// The (at most two) bricks that the taps of one dim fall into, after the boundary condition
// (bounds.py:30-89) for border samples.  Taps with sign 0 go nowhere.  Returns the number of
// distinct bricks (0, 1, 2) or 3 when there are more than two (tiny lattices): such samples are
// left to the atomic walk (bricks_border).
__device__ __noinline__ int dim_bricks(int bound, int k, int n, int i0, bool border, int *b0, int *b1)
{
    if (!border) { *b0 = i0 / BS; *b1 = (i0 + k) / BS; return *b1 > *b0 ? 2 : 1; }
    int cnt = 0;
    for (int j = 0; j <= k; ++j) {
        const long long pk = wrap_outofline(bound, i0 + j, n);
        if ((int)(pk >> 32) == 0) continue;
        const int br = (int)(pk & 0xffffffffll) / BS;
        if (cnt == 0) { *b0 = br; cnt = 1; }
        else if (br != *b0 && (cnt == 1 || br != *b1)) { if (cnt == 2) return 3; *b1 = br; cnt = 2; }
    }
    return cnt;
}

// max |src| per value channel, stage 1 of 2 (stage 2 in bricks_scan): blockIdx.y = channel,
// each block reduces a strided share of the (B, N) values and writes one partial.
constexpr int MAXPART = 256;
__global__ __launch_bounds__(256) void bricks_max(KParams p, Bricks bk, const float *__restrict__ val, int B)
{
    __shared__ int red[4];
    const int c = blockIdx.y;
    int bits = 0;
    for (int b = 0; b < B; ++b) {
        const float *vc = val + (int64_t)b * p.val_sb + (int64_t)c * p.val_sc;
        for (unsigned o = blockIdx.x * 256u + threadIdx.x; o < bk.N; o += MAXPART * 256u) {
            const float a = __builtin_fabsf(vc[o]);
            const int ab = (a != a) ? 0x7fc00000 : __float_as_int(a);      // NaN sticks
            bits = ab > bits ? ab : bits;
        }
    }
    bits = wave_max_i(bits);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) bits = red[i] > bits ? red[i] : bits;
        bk.partial[c * MAXPART + blockIdx.x] = bits;
    }
}

// Steps 1 and 3: one thread per sample.  A workgroup's 256 samples are neighbours: they hit a few dozen
// distinct bricks.  The (brick -> samples of this workgroup) counts are merged in an LDS hash table first and
// ONE global atomic per distinct brick and workgroup reserves the list positions; same-address global atomics
// serialise, and a per-wave merge loop pays one atomic round trip per distinct brick, one after the other
// (measured at config 4, 8 sources: count + fill 2.55 ms that way).
// hdr[8] is set when a sample spans more than two bricks along a dim (tiny lattices): only then has
// bricks_border anything to do.
constexpr int HS = 2048;                        // hash slots = the most insertions of a chunk (256 samples x 8 bricks): probing ends
template <bool FILL>
__global__ __launch_bounds__(256) void bricks_walk(KParams p, Bricks bk, const float *__restrict__ val, const float *__restrict__ grid,
                                                   float *__restrict__ vol, int B, int nch)
{
    __shared__ int hkey[HS], hcnt[HS], used[HS], nused;
    // persistent workgroups over (batch item, chunk of 256 samples); the next chunk's coordinates are fetched
    // while this one is processed
    const unsigned cpb = (bk.N + 255u) / 256u;                      // chunks per batch item
    const unsigned total = cpb * (unsigned)B;
    for (int e = threadIdx.x; e < HS; e += 256) { hkey[e] = -1; hcnt[e] = 0; }
    if (threadIdx.x == 0) nused = 0;
    float xn[3] = { 0.f, 0.f, 0.f };
    if (blockIdx.x < total) {
        const unsigned b = blockIdx.x / cpb, o = (blockIdx.x - b * cpb) * 256u + threadIdx.x;
        if (o < bk.N) load_coords<float, float, 3>(p, grid, (int64_t)b, (int64_t)o, xn);
    }
    __syncthreads();
    for (unsigned w = blockIdx.x; w < total; w += gridDim.x) {
        const int b = (int)(w / cpb);
        const unsigned o = (w - (unsigned)b * cpb) * 256u + threadIdx.x;
        const bool live = o < bk.N;
        const float x[3] = { xn[0], xn[1], xn[2] };
        {
            const unsigned wn = w + gridDim.x;
            if (wn < total) {
                const unsigned bn = wn / cpb, on = (wn - bn * cpb) * 256u + threadIdx.x;
                if (on < bk.N) load_coords<float, float, 3>(p, grid, (int64_t)bn, (int64_t)on, xn);
            }
        }
        int i0[3]; float t[3];
        int cls = 0;
        if (live) cls = classify(p, x, i0, t);
        const int tb = bk.shared ? 0 : b;
        int bb[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, nb_[3] = { 0, 0, 0 };
        bool in = cls != 0;
        if (cls != 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (cls == 1) {            // interior: no boundary condition to apply (the common case, inline)
                    bb[d][0] = i0[d] / BS; bb[d][1] = (i0[d] + p.order[d]) / BS;
                    nb_[d] = bb[d][1] > bb[d][0] ? 2 : 1;
                } else if (p.bound[d] == B_REPLICATE || p.bound[d] == B_ZERO) {
                    // clamping boundaries inline (the out-of-line table walk below cost 0.3 ms of each 0.6 ms pass at
                    // config 4, where 4.6 % of the samples touch the border): the taps land on [lo, hi], <= 2 bricks
                    const int n1 = p.vol_n[d] - 1, a = i0[d], e = i0[d] + p.order[d];
                    const int lo_ = a < 0 ? 0 : (a > n1 ? n1 : a), hi_ = e < 0 ? 0 : (e > n1 ? n1 : e);
                    const bool none = p.bound[d] == B_ZERO && (e < 0 || a > n1);          // every tap has sign 0
                    bb[d][0] = lo_ / BS; bb[d][1] = hi_ / BS;
                    nb_[d] = none ? 0 : (bb[d][1] > bb[d][0] ? 2 : 1);
                } else {
                    nb_[d] = dim_bricks(p.bound[d], p.order[d], p.vol_n[d], i0[d], true, &bb[d][0], &bb[d][1]);
                    if (!FILL && nb_[d] == 3) bk.hdr[8] = 1;
                }
                in = in && nb_[d] >= 1 && nb_[d] <= 2;          // 0: nothing to splat; 3: left to bricks_border
            }
        }
        int slot[8], rank[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int ex = corner >> 2, ey = (corner >> 1) & 1, ez = corner & 1;
            const bool on = in && (!ex || nb_[0] == 2) && (!ey || nb_[1] == 2) && (!ez || nb_[2] == 2);
            slot[corner] = -1; rank[corner] = 0;
            if (on) {
                const int br = tb * bk.per_target + (bb[0][ex] * bk.nb[1] + bb[1][ey]) * bk.nb[2] + bb[2][ez];
                int sl = (int)(((unsigned)br * 2654435761u) >> 21) & (HS - 1);
                while (true) {
                    const int prev = atomicCAS(&hkey[sl], -1, br);
                    if (prev == -1) used[atomicAdd(&nused, 1)] = sl;      // a new key of this chunk
                    if (prev == -1 || prev == br) break;
                    sl = (sl + 1) & (HS - 1);
                }
                slot[corner] = sl;
                rank[corner] = atomicAdd(&hcnt[sl], 1);
            }
        }
        __syncthreads();
        const int nu = nused;
        for (int e = threadIdx.x; e < nu; e += 256) {
            const int sl = used[e], key = hkey[sl];
            if (FILL) hcnt[sl] = atomicAdd(&bk.cursor[key], hcnt[sl]);
            else __hip_atomic_fetch_add(&bk.counts[key], hcnt[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (FILL) {
#pragma unroll
            for (int corner = 0; corner < 8; ++corner)
                if (slot[corner] >= 0) bk.list[hcnt[slot[corner]] + rank[corner]] = (unsigned)b * bk.N + o;
            __syncthreads();
        }
        for (int e = threadIdx.x; e < nu; e += 256) { const int sl = used[e]; hkey[sl] = -1; hcnt[sl] = 0; }   // only what was touched
        if (threadIdx.x == 0) nused = 0;
        __syncthreads();
    }
}

// Border samples (stencil leaving the lattice).  Same walk as above -- every lane classifies its
// own sample from a coalesced read -- then the wave takes its border samples one at a time, lanes =
// taps, boundary wrap per tap (bounds.py:30-89), float atomics straight to the target.
__global__ __launch_bounds__(256) void bricks_border(KParams p, Bricks bk, const float *__restrict__ val, const float *__restrict__ grid,
                                                     float *__restrict__ vol, int B, int nch)
{
    if (bk.hdr[8] == 0) return;                            // no sample spans more than two bricks along a dim (bricks_walk)
    const unsigned o = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int k1[3] = { p.order[0] + 1, p.order[1] + 1, p.order[2] + 1 };
    const int ntap = k1[0] * k1[1] * k1[2];
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        float x[3] = { 0.f, 0.f, 0.f };
        int i0[3] = { 0, 0, 0 }; float t[3] = { 0.f, 0.f, 0.f };
        int cls = 0;
        if (o < bk.N) { load_coords<float, float, 3>(p, grid, b, o, x); cls = classify(p, x, i0, t); }
        bool far = false;                                  // more than two bricks along a dim: not in the lists
        if (cls == 2) {
            int u0, u1;
#pragma unroll
            for (int d = 0; d < 3; ++d) far = far || dim_bricks(p.bound[d], p.order[d], p.vol_n[d], i0[d], true, &u0, &u1) == 3;
        }
        unsigned long long todo = __ballot(far);
        if (!todo) continue;
        float sv[4] = { 1.f, 1.f, 1.f, 1.f };
        if (far)
            for (int c = 0; c < p.C; ++c) sv[c] = val[(int64_t)b * p.val_sb + c * p.val_sc + o];
        float *vb = vol + (bk.shared ? 0 : (int64_t)b * p.vol_sb);
        while (todo) {
            const int j = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int q0[3] = { __shfl(i0[0], j), __shfl(i0[1], j), __shfl(i0[2], j) };
            const float tt[3] = { __shfl(t[0], j), __shfl(t[1], j), __shfl(t[2], j) };
            const float s4[4] = { __shfl(sv[0], j), __shfl(sv[1], j), __shfl(sv[2], j), __shfl(sv[3], j) };
            for (int t0 = 0; t0 < ntap; t0 += 64) {
                const int tap = t0 + lane;
                if (tap >= ntap) continue;
                const int tp[3] = { tap / (k1[1] * k1[2]), (tap / k1[2]) % k1[1], tap % k1[2] };
                float w = 1.f; int64_t off = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const long long pk = wrap_outofline(p.bound[d], q0[d] + tp[d], p.vol_n[d]);
                    w *= weight_of(p, d, tt[d], tp[d]) * (float)(int)(pk >> 32);
                    off += (int64_t)(int)(pk & 0xffffffffll) * (p.vol_ss[d] / 4);
                }
                for (int c = 0; c < nch; ++c)
                    __hip_atomic_fetch_add(vb + c * p.vol_sc + off, w * (c < p.C ? s4[c < 4 ? c : 3] : 1.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Step 2: exclusive scan of the counters (one workgroup; n is a few 10^4 .. 10^6)
__global__ __launch_bounds__(1024) void bricks_scan(Bricks bk, int n, int nval)
{
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    if ((int)threadIdx.x < nval) {                         // stage 2 of the channel maxima
        int m = 0;
        for (int i = 0; i < 256; ++i) { const int v = bk.partial[threadIdx.x * 256 + i]; m = v > m ? v : m; }
        bk.hdr[threadIdx.x] = m;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? bk.counts[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            const int u = threadIdx.x >= s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        const int excl = carry + part[threadIdx.x] - v;
        if (i < n) { bk.offsets[i] = excl; bk.cursor[i] = excl; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) bk.offsets[n] = carry;
}

// Step 4: one workgroup per brick (persistent), one thread per listed sample.  ACC_NT = 512: two workgroups fit a CU
// (73 VGPRs, 64 KiB of LDS for two channels) and overlap each other's list / flush phases -- better for short lists
// (config 4, 8 sources: 2.37 -> 2.0 ms); long lists (64 sources: ~6800 samples per brick) do better with one
// workgroup of 1024 threads (23.9 vs 29.0 ms).
template <int NCH, int K, int ACC_NT>
__global__ __launch_bounds__(ACC_NT) void bricks_accumulate(KParams p, Bricks bk, const float *__restrict__ val, const float *__restrict__ grid,
                                                          float *__restrict__ vol, int nbricks)
{
    extern __shared__ unsigned long long acc[];            // NCH x BSLOTS
    const int tid = threadIdx.x;
    // fixed-point scales: 2^e max|src| <= 2^30 (the count channel's source is 1)
    float scale[NCH], inv[NCH]; bool finite[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int bits = c < p.C ? bk.hdr[c] : 0x3f800000;
        finite[c] = (bits & 0x7f800000) != 0x7f800000;
        int ex = ((bits >> 23) & 0xff) - 127;
        ex = ex < -90 ? -90 : ex;
        scale[c] = __int_as_float((127 + 29 - ex) << 23);
        inv[c] = __int_as_float((127 - 29 + ex) << 23);
    }
    for (int br = blockIdx.x; br < nbricks; br += gridDim.x) {
        const int beg = bk.offsets[br], end = bk.offsets[br + 1];
        if (beg == end) continue;                           // block-uniform
        const int tb = br / bk.per_target, r = br - tb * bk.per_target;
        const int bx = r / (bk.nb[1] * bk.nb[2]), by = (r / bk.nb[2]) % bk.nb[1], bz = r % bk.nb[2];
        __syncthreads();
        for (int e = tid; e < NCH * BSLOTS; e += ACC_NT) acc[e] = 0ull;
        __syncthreads();
        float *vt = vol + (int64_t)tb * p.vol_sb;
        for (int e = beg + tid; e < end; e += ACC_NT) {
            const unsigned id = bk.list[e];
            const unsigned b = id / bk.N, o = id - b * bk.N;
            float x[3]; int i0[3]; float t[3];
            load_coords<float, float, 3>(p, grid, (int64_t)b, (int64_t)o, x);
            const int cls = classify(p, x, i0, t);
            float sv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) sv[c] = (c < p.C ? val[(int64_t)b * p.val_sb + c * p.val_sc + o] : 1.f) * scale[c];
            float wx[K + 1], wy[K + 1], wz[K + 1];
            stencil_weights<K>(p, 0, t[0], wx); stencil_weights<K>(p, 1, t[1], wy); stencil_weights<K>(p, 2, t[2], wz);
            // lattice index (relative to the brick) and sign of every tap; border samples wrap
            int jx[K + 1], jy[K + 1], jz[K + 1];
#pragma unroll
            for (int j = 0; j <= K; ++j) { jx[j] = i0[0] + j - bx * BS; jy[j] = i0[1] + j - by * BS; jz[j] = i0[2] + j - bz * BS; }
            if (cls == 2) {
                // (clamping boundaries inline, the others through the out-of-line boundary switch)
                auto wrap1 = [&](int d, int i, int &idx, float &sgn) {
                    const int bd = p.bound[d], n = p.vol_n[d];
                    if (bd == B_REPLICATE || bd == B_ZERO) {
                        const bool in = (unsigned)i < (unsigned)n;
                        idx = i < 0 ? 0 : (i >= n ? n - 1 : i);
                        sgn = (bd == B_ZERO && !in) ? 0.f : 1.f;
                    } else {
                        const long long pk = wrap_outofline(bd, i, n);
                        idx = (int)(pk & 0xffffffffll); sgn = (float)(int)(pk >> 32);
                    }
                };
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    int ix_, iy_, iz_; float sx_, sy_, sz_;
                    wrap1(0, i0[0] + j, ix_, sx_); wrap1(1, i0[1] + j, iy_, sy_); wrap1(2, i0[2] + j, iz_, sz_);
                    jx[j] = ix_ - bx * BS; wx[j] *= sx_;
                    jy[j] = iy_ - by * BS; wy[j] *= sy_;
                    jz[j] = iz_ - bz * BS; wz[j] *= sz_;
                }
            }
#pragma unroll
            for (int i = 0; i <= K; ++i) {
                if ((unsigned)jx[i] >= (unsigned)BS) continue;                     // another brick's taps
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    if ((unsigned)jy[j] >= (unsigned)BS) continue;
                    const float wij = wx[i] * wy[j];
                    const int row = (jx[i] * BS + jy[j]) * BS;
#pragma unroll
                    for (int k = 0; k <= K; ++k) {
                        if ((unsigned)jz[k] >= (unsigned)BS) continue;
                        const float w = wij * wz[k];
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            if (finite[c]) {
                                atomicAdd(&acc[c * BSLOTS + row + jz[k]], (unsigned long long)(long long)__float2int_rn(sv[c] * w));   // ds_add_u64
                            } else {
                                const int64_t off = (int64_t)(bx * BS + jx[i]) * (p.vol_ss[0] / 4) + (int64_t)(by * BS + jy[j]) * (p.vol_ss[1] / 4)
                                                  + (int64_t)(bz * BS + jz[k]) * (p.vol_ss[2] / 4);
                                __hip_atomic_fetch_add(vt + c * p.vol_sc + off, w * (sv[c] * inv[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // the brick goes to the target: this workgroup is its only writer (border samples are
        // splatted by another kernel of the same stream)
        // (all the reads of the thread's voxels first, then the writes: one memory round trip, not one per voxel)
        constexpr int NV = BSLOTS / ACC_NT;
        float cur[NV][NCH]; int64_t offs[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = tid + ACC_NT * u;
            const int sx = e / (BS * BS), sy = (e / BS) % BS, sz = e % BS;
            const int gx = bx * BS + sx, gy = by * BS + sy, gz = bz * BS + sz;
            offs[u] = (gx >= p.vol_n[0] || gy >= p.vol_n[1] || gz >= p.vol_n[2]) ? -1
                    : (int64_t)gx * (p.vol_ss[0] / 4) + (int64_t)gy * (p.vol_ss[1] / 4) + (int64_t)gz * (p.vol_ss[2] / 4);
#pragma unroll
            for (int c = 0; c < NCH; ++c) cur[u][c] = offs[u] >= 0 ? vt[c * p.vol_sc + offs[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (offs[u] < 0) continue;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const long long a = (long long)acc[c * BSLOTS + tid + ACC_NT * u];
                if (a != 0) vt[c * p.vol_sc + offs[u]] = cur[u][c] + (float)a * inv[c];
            }
        }
    }
}

} // namespace

int64_t bricks_workspace_bytes(const KParams &p, int B, int shared)
{
    int64_t nbt = shared ? 1 : B;
    for (int d = 0; d < 3; ++d) nbt *= (p.vol_n[d] + BS - 1) / BS;
    return 4 * ((int64_t)HDR + 8 * 256 + nbt + (nbt + 1) + nbt) + 4 * 8 * (int64_t)B * p.N;      // per sample: <= 8 brick entries
}

// k.C value channels (+ k.cc: count channel).  1 = done, 0 = declined, < 0 error.
int launch_push_bricks(const KParams &p, int B, int shared, const void *val, const void *grid, void *vol,
                       void *workspace, int64_t workspace_bytes, hipStream_t st)
{
    const int nch = p.C + p.cc;
    if (p.dim != 3 || nch < 1 || nch > 4 || (uint64_t)B * (uint64_t)p.N > 0xffffffffull) return 0;
    const int K = p.order[0];
    if (p.order[1] != K || p.order[2] != K || K > 3) return 0;                        // one order <= 3 for all dims
    if (workspace_bytes < bricks_workspace_bytes(p, B, shared)) return INTERPOL_E_SCRATCH;
    Bricks bk;
    int64_t per = 1;
    for (int d = 0; d < 3; ++d) { bk.nb[d] = (p.vol_n[d] + BS - 1) / BS; per *= bk.nb[d]; }
    const int64_t nbt = per * (shared ? 1 : B);
    if (nbt > 0x3fffffff) return 0;
    bk.per_target = (int)per; bk.shared = shared; bk.N = (unsigned)p.N;
    int *w = (int *)workspace;
    bk.hdr = w; bk.partial = w + HDR; bk.counts = bk.partial + 8 * 256; bk.offsets = bk.counts + nbt; bk.cursor = bk.offsets + nbt + 1;
    bk.list = (unsigned *)(bk.cursor + nbt);
    hipError_t e = hipMemsetAsync(workspace, 0, 4 * (size_t)(HDR + 8 * 256 + nbt), st);       // header + partial maxima + counters
    if (e != hipSuccess) return (int)e;
    const dim3 grid1((unsigned)((p.N + 255) / 256), (unsigned)(B < 65535 ? B : 65535));
    const uint64_t chunks = (uint64_t)((p.N + 255) / 256) * (uint64_t)B;
    if (chunks > 0xffffffffull) return 0;
    const dim3 gridw((unsigned)(chunks < 4096 ? chunks : 4096));          // persistent walkers (4096 vs exactly-resident 1536: 0.59 vs 0.63 ms)
    if (p.C > 0) hipLaunchKernelGGL((bricks_max), dim3(256, (unsigned)p.C), dim3(256), 0, st, p, bk, (const float *)val, B);
    hipLaunchKernelGGL((bricks_walk<false>), gridw, dim3(256), 0, st, p, bk, (const float *)val, (const float *)grid, (float *)vol, B, nch);
    hipLaunchKernelGGL((bricks_scan), dim3(1), dim3(1024), 0, st, bk, (int)nbt, p.C);
    hipLaunchKernelGGL((bricks_walk<true>), gridw, dim3(256), 0, st, p, bk, (const float *)val, (const float *)grid, (float *)vol, B, nch);
    hipLaunchKernelGGL((bricks_border), grid1, dim3(256), 0, st, p, bk, (const float *)val, (const float *)grid, (float *)vol, B, nch);
    const size_t lds = (size_t)nch * BSLOTS * 8;
    const unsigned blocks = (unsigned)(nbt < 2048 ? nbt : 2048);
    const bool long_lists = (double)B * (double)p.N * 2. > 3000. * (double)nbt;       // estimated samples per brick list
#define IP_BR3(NC, KK, NTH) { \
        if (lds > 64 * 1024) { e = hipFuncSetAttribute((const void *)bricks_accumulate<NC, KK, NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                               if (e != hipSuccess) return (int)e; } \
        hipLaunchKernelGGL((bricks_accumulate<NC, KK, NTH>), dim3(blocks), dim3(NTH), lds, st, p, bk, (const float *)val, (const float *)grid, (float *)vol, (int)nbt); }
#define IP_BR2(NC, KK) { if (long_lists) IP_BR3(NC, KK, 1024) else IP_BR3(NC, KK, 512) }
#define IP_BR(NC) case NC: switch (K) { case 0: IP_BR2(NC, 0) break; case 1: IP_BR2(NC, 1) break; case 2: IP_BR2(NC, 2) break; default: IP_BR2(NC, 3) break; } break;
    switch (nch) { IP_BR(1) IP_BR(2) IP_BR(3) IP_BR(4) default: return 0; }
#undef IP_BR
#undef IP_BR2
#undef IP_BR3
    e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

} // namespace ip
