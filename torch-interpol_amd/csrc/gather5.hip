// ===========================================================================
// gather5.hip -- grid_pull / grid_grad (reference interpol/nd.py:80-143, 216-288) for spline orders 4 and 5 in 3-D, float32,
// through BRICKS OF THE IMAGE: the organisation of push_owner.hip's own_gather (HISTORY.md 4.2e) for the 125 / 216-tap stencils of
// BASELINE config 3 (8 x 1 x 192^3, order 5).
//
// The LDS tiles of ops_tiled.hip stage, per 16^3-sample tile, the bounding box of the tile's stencils -- (16 + 5 + 4 sigma)^3
// lattice points under i.i.d. noise, 147 KiB: ONE workgroup per CU, every phase of a tile exposed.  Here the samples are first
// sorted by the 16^3 brick of the image their stencil STARTS in (bin5: one pass over the sample grid, tile-local counting sort,
// runs of records (x, y, z, sample index) published per (tile, brick)); then a workgroup draws a non-empty brick, stages the
// brick's 21^3 lattice points once per channel (1.7 lattice points per sample whatever the deformation; 69 KiB: two workgroups
// per CU) and gathers the brick's records from it (gather5).  A box slot holds the PAIR (v[z], v[z + 1]): one ds_read_b64 feeds
// two z-taps of a single channel at any z -- 108 reads for the 216 taps of a quintic stencil.
// Samples whose stencil starts more than 160 points outside the lattice, tiles that spread over more than 6 bricks per dim and
// runs beyond a brick's 128 descriptors are gathered by their own thread from global memory.
// Every boundary condition (a box slot is a lattice point through the tables), the three extrapolation modes, the four coordinate
// sources.  Workspace: 16 B per sample + 1 KiB per brick (interpol_pull_workspace).
// ===========================================================================
//
// Orders 6 and 7 (round 6): the same file compiled a second time (gather7.hip: -DIP_G5_HIGH) -- bricks of 14^3 first-tap cells so that the
// 343 / 512-tap stencils of a brick still touch 21^3 lattice points (the same LDS box, two workgroups per CU), rows of four pair slots
// (the eighth z-slot of an order-6 stencil is cleared after the read; scatter5: rows of 7 / 8 adds, first-tap cells in rows of 16); namespace g7,
// entry points try_gather7 / try_scatter7 / *_workspace_bytes, reached through try_gather5 / try_scatter5 / their *_workspace_bytes.  Until then these orders had the 8^3-sample round-1
// tiles alone (4 x 2 x 256^3: pull 14 ms, grid_grad 18).
#include "sorted_util.hpp"

#ifdef IP_G5_HIGH
#define IP_G5_CP 16                             /* pitch of the first-tap cell arrays of scatter5 (bricks of 14 cells in rows of 16) */
#define IP_G5_NS g7
#define IP_G5_KLO 6
#define IP_G5_KHI 7
#define IP_G5_TRY try_gather7
#define IP_G5_WSB gather7_workspace_bytes
#define IP_S5_TRY try_scatter7
#define IP_S5_WSB scatter7_workspace_bytes
#define IP_B5_TRY try_backward7
#define IP_PB5_TRY try_pushbwd7
#else
#define IP_G5_CP BR
#define IP_G5_NS g5
#define IP_G5_KLO 4
#define IP_G5_KHI 5
#define IP_G5_TRY try_gather5
#define IP_G5_WSB gather5_workspace_bytes
#define IP_S5_TRY try_scatter5
#define IP_S5_WSB scatter5_workspace_bytes
#define IP_B5_TRY try_backward5
#define IP_PB5_TRY try_pushbwd5
#endif

namespace ip {
namespace IP_G5_NS {

using namespace sorted;

#ifdef IP_G5_HIGH
constexpr int BR = 14;                          // brick edge, in first-tap cells
constexpr int OFFB = 160;                       // first taps in [-OFFB, n + OFFB) are binned
constexpr int BOX = BR + 7;                     // lattice points a brick's stencils touch per dim (K <= 7)
#else
constexpr int BR = 16;                          // brick edge, in first-tap cells
constexpr int OFFB = 160;                       // first taps in [-OFFB, n + OFFB) are binned
constexpr int BOX = BR + 5;                     // lattice points a brick's stencils touch per dim (K <= 5)
#endif
constexpr int PZ = BOX - 1;                     // pair slots per row: slot z = (v[z], v[z + 1])
constexpr int PLANE = BOX * PZ;
constexpr int NT = 512;                         // gather5: threads (two workgroups per CU)
constexpr int NS = TS * TS * TS, NT1 = 512, VPT1 = NS / NT1;
constexpr int LB = 6, NBIN = LB * LB * LB;      // bricks around a tile that are sorted locally
constexpr int CAPD = 128;                       // runs per brick

struct Grid5 { int nb[3]; int per_item; };
static Grid5 brick_grid(const KParams &k)
{
    Grid5 g;
    for (int d = 0; d < 3; ++d) g.nb[d] = (k.vol_n[d] + 2 * OFFB + BR - 1) / BR;
    g.per_item = g.nb[0] * g.nb[1] * g.nb[2];
    return g;
}

struct Workspace { int *hdr; int *ndesc; int *list; uint2 *desc; float4 *rec; int64_t nbricks, nrec; };
static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int64_t layout(const Grid5 &bg, int B, int64_t ntiles, void *base, Workspace *w)
{
    const int64_t nbricks = (int64_t)bg.per_item * B, nrec = ntiles * NS * B;
    unsigned char *p = (unsigned char *)base;
    int64_t o = 0;
    const int64_t o_hdr = o; o += 256;                               // header (64 ints), brick counters, brick list: ONE zero-fill
    const int64_t o_nd = o; o += nbricks * 4;
    const int64_t o_li = o; o += (nbricks + 1) * 4; o = align256(o);
    const int64_t o_desc = o; o += align256(nbricks * CAPD * 8);
    const int64_t o_rec = o; o += align256(nrec * 16);
    if (w) { w->hdr = (int *)(p + o_hdr); w->ndesc = (int *)(p + o_nd); w->list = (int *)(p + o_li); w->desc = (uint2 *)(p + o_desc);
             w->rec = (float4 *)(p + o_rec); w->nbricks = nbricks; w->nrec = nrec; }
    return o;
}

__global__ void zero5(int *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// MODE 0: pull, val[b,c,o]; MODE 2: grid_grad, val[b,c,o,:]; MODE 1: the grid gradient of the pull's backward (pushpull.py:256-257),
// out[b,o,:] = mask * sum_c gout[b,c,o] * grad pull(img[b,c])(x_o) (gout == NULL: ones; p.val_* describe gout)
template <int K, int GM, int MODE>
__device__ __forceinline__ void direct5(const KParams &p, const float *__restrict__ img, const float *__restrict__ grid, float *__restrict__ out,
                                        int64_t b, TileGeom g, int tid, unsigned mask, const float *__restrict__ gout)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = K; }
    L.lin = 0;
#pragma unroll 1
    for (int v = 0; v < VPT1; ++v) {
        if (!((mask >> v) & 1)) continue;
        int ox, oy, oz; float x[3];
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
        const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
        const float m = inb_mask(p, x);
        int ii[3]; float tt[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) split(K, x[d], ii[d], tt[d]);
        if (MODE == 3) {
            // push / count (scatter5): `img` is the SOURCE image (NULL: count), `out` the target; channel C (p.cc) receives the count
#pragma unroll 1
            for (int ch = 0; ch < p.C + p.cc; ++ch) {
                const float sv = (img && ch < p.C) ? img[b * p.val_sb + ch * p.val_sc + o] * m : m;
                if (sv != 0.f) tiled::scatter_one_thread(L, out + b * p.vol_sb + ch * p.vol_sc, sv, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2]);
            }
            continue;
        }
        if (MODE == 1) {
            float a[3] = { 0.f, 0.f, 0.f };
#pragma unroll 1
            for (int ch = 0; ch < p.C; ++ch) {
                const float gv = gout ? gout[b * p.val_sb + ch * p.val_sc + o] : 1.f;
#pragma unroll 1
                for (int d = 0; d < 3; ++d)
                    a[d] = __builtin_fmaf(gv, tiled::gather_one_thread<float>(L, img + b * p.vol_sb + ch * p.vol_sc, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], d), a[d]);
            }
            float *dst = out + (b * p.N + o) * 3;
            dst[0] = a[0] * m; dst[1] = a[1] * m; dst[2] = a[2] * m;
            continue;
        }
#pragma unroll 1
        for (int ch = 0; ch < p.C; ++ch) {
            const float *ic = img + b * p.vol_sb + ch * p.vol_sc;
            if (MODE == 0) out[b * p.val_sb + ch * p.val_sc + o] = m * tiled::gather_one_thread<float>(L, ic, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1);
            else {
#pragma unroll 1
                for (int d = 0; d < 3; ++d)
                    out[b * p.val_sb + ch * p.val_sc + 3 * o + d] = m * tiled::gather_one_thread<float>(L, ic, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], d);
            }
        }
    }
}

struct BinSmem { int lo[3], pad; int cnt[NBIN], base[NBIN]; };

// One workgroup per 16^3-sample tile: the tile's samples sorted by the brick of their first tap (nd.py:45: i0 = floor(x - (K-1)/2)).
template <int K, int GM, int MODE>
__global__ __launch_bounds__(NT1, 4) void bin5(KParams p, Grid5 bg, const float *__restrict__ img, const float *__restrict__ grid, float *__restrict__ out,
                                               int *__restrict__ ndesc, int *__restrict__ list, uint2 *__restrict__ desc, float4 *__restrict__ rec,
                                               int gx, int gy, int gz, int nty, int ntz, int ntiles, const int *__restrict__ gate,
                                               const float *__restrict__ gout, float *__restrict__ acc2)
{
    // MODE 4 (round 6): ONE binning for both halves of the pull's backward -- the records serve scatter5 (image gradient) and gather5<K, 1>
    // (grid gradient); img / out / gout as in MODE 1, acc2 = the dense float image gradient, `gate` = the SCATTER's verdict: the binning
    // itself always runs (the grid gradient needs it), the direct scatter of the unbinned samples only when the bricks have the scatter
    if (MODE < 4 && gate && *gate != 1) return;                      // INTERPOL_FLAG_AUTO_SCATTER: the probe chose the tiles
    __shared__ BinSmem sm;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / ntiles;
    const TileGeom g = tile_geom(blockIdx.x % ntiles, gx, gy, gz, nty, ntz);
    for (int i = tid; i < NBIN; i += NT1) sm.cnt[i] = 0;
    if (tid < 3) sm.lo[tid] = 0x7fffffff;
    float c[VPT1][3];
    int idx[VPT1];
    unsigned valid = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        int ox, oy, oz;
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        if (ox < gx && oy < gy && oz < gz) valid |= 1u << v;
        ox = ox < gx ? ox : gx - 1; oy = oy < gy ? oy : gy - 1; oz = oz < gz ? oz : gz - 1;
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, c[v]);
        idx[v] = (int)(((int64_t)ox * gy + oy) * gz + oz);
    }
    int bx[VPT1][3];
    unsigned ok = 0;
    int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff };
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        bool in = (valid >> v) & 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float fl = floorf(c[v][d] - 0.5f * (float)(K - 1));
            in = in && fl >= (float)(-OFFB) && fl < (float)(bg.nb[d] * BR - OFFB);     // (false for NaN)
#ifdef IP_G5_HIGH
            bx[v][d] = in ? (__float2int_rz(fl) + OFFB) / BR : 0;
#else
            bx[v][d] = in ? (__float2int_rz(fl) + OFFB) >> 4 : 0;
#endif
        }
        if (in) {
            ok |= 1u << v;
#pragma unroll
            for (int d = 0; d < 3; ++d) mn[d] = bx[v][d] < mn[d] ? bx[v][d] : mn[d];
        }
    }
    __syncthreads();                                                 // counters zero
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = wave_min(mn[d]);
        if ((tid & 63) == 0) atomicMin(&sm.lo[d], a);
    }
    __syncthreads();
    const int lo[3] = { sm.lo[0], sm.lo[1], sm.lo[2] };
    int lbin[VPT1];                                                  // local brick (8 bits), rank inside it (above)
    unsigned local = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        const int r0 = bx[v][0] - lo[0], r1 = bx[v][1] - lo[1], r2 = bx[v][2] - lo[2];
        const bool l = ((ok >> v) & 1) && (unsigned)r0 < (unsigned)LB && (unsigned)r1 < (unsigned)LB && (unsigned)r2 < (unsigned)LB;
        lbin[v] = l ? (r0 * LB + r1) * LB + r2 : 0;
        if (l) { local |= 1u << v; lbin[v] |= atomicAdd(&sm.cnt[lbin[v]], 1) << 8; }
    }
    __syncthreads();
    const int64_t tilebase = (int64_t)blockIdx.x * NS;
    if (tid < 64) {
        constexpr int PER = (NBIN + 63) / 64;                        // 4
        int cn[PER], s = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int e = tid * PER + i; cn[i] = e < NBIN ? sm.cnt[e] : 0; s += cn[i]; }
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
        int run = incl - s;
        int bk[PER], slot[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {                              // (all slots drawn before the first is used: one round trip)
            const int e = tid * PER + i;
            const int r0 = e / (LB * LB), r1 = (e / LB) % LB, r2 = e % LB;
            bk[i] = (int)b * bg.per_item + ((lo[0] + r0) * bg.nb[1] + (lo[1] + r1)) * bg.nb[2] + (lo[2] + r2);
            slot[i] = 0;
            if (e < NBIN && cn[i] > 0) slot[i] = atomicAdd(&ndesc[bk[i]], 1);
            if (e < NBIN && cn[i] > 0 && slot[i] == 0) list[1 + atomicAdd(&list[0], 1)] = bk[i];          // first run of the brick
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid * PER + i;
            if (e < NBIN) {
                sm.base[e] = run;
                if (cn[i] > 0) {
                    if (slot[i] < CAPD) desc[(int64_t)bk[i] * CAPD + slot[i]] = make_uint2((unsigned)(tilebase + run), (unsigned)cn[i]);
                    else sm.cnt[e] = -1;                             // the brick's list is full: gathered directly, below
                }
                run += cn[i];
            }
        }
    }
    __syncthreads();
    unsigned direct = valid & ~local;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (!((local >> v) & 1)) continue;
        const int e = lbin[v] & 255;
        if (sm.cnt[e] < 0) { direct |= 1u << v; continue; }
        rec[tilebase + sm.base[e] + (lbin[v] >> 8)] = make_float4(c[v][0], c[v][1], c[v][2], __int_as_float(idx[v]));
    }
    if (MODE == 5) {
        // ONE binning for both gradients of the PUSH's backward (pushpull.py:262-282): the records serve gather5<K, 0> (out: the gradient of the
        // values, a pull of grad_vol_out = img) and gather5<K, 1> (acc2: the grid gradient, contracted with the values = gout)
        if (direct) {
            direct5<K, GM, 0>(p, img, grid, out, b, g, tid, direct, nullptr);
            direct5<K, GM, 1>(p, img, grid, acc2, b, g, tid, direct, gout);
        }
        return;
    }
    if (MODE == 4) {
        if (direct) {
            direct5<K, GM, 1>(p, img, grid, out, b, g, tid, direct, gout);
            if (!gate || *gate == 1) {
                KParams q = p;                                       // the scatter's view: sources = grad_out (val_*), target = the dense image gradient
                int64_t dense = 1;
#pragma unroll
                for (int d = 2; d >= 0; --d) { q.vol_ss[d] = (int)(dense * 4); dense *= p.vol_n[d]; }
                q.vol_sc = dense; q.vol_sb = dense * p.C; q.cc = 0;
                direct5<K, GM, 3>(q, gout, grid, acc2, b, g, tid, direct, nullptr);
            }
        }
        return;
    }
    if (direct) direct5<K, GM, MODE>(p, img, grid, out, b, g, tid, direct, gout);
}

// INTERPOL_FLAG_AUTO_SCATTER: NPROBE tiles of the sample grid are examined the way the LDS tiles of ops_tiled.hip would cut them --
// the box of a 16^3-sample tile of orders 4 / 5 holds 33 x 33 x 32 lattice points, centred on the tile's stencils; a sample whose
// stencil leaves it costs a WAVE there (64 times a sample inside).  Used for the order-4 pull only (try_gather5): hdr[0] = 1 (the
// bricks) when more than 1 / 300 of the probed samples do, or when the field is ROUGH (8 x 1 x 192^3, i.i.d. noise: sigma = 1 1.54
// against 1.78 ms, sigma = 2 1.61 / 2.02); smooth fields stay with the tiles (identity 1.19 / 1.27, zoom 1.2 1.23 / 1.51).
// hdr[1..5]: counters.
constexpr int NPROBE = 128;
template <int K, int GM>
__global__ __launch_bounds__(NT1) void probe5(KParams p, const float *__restrict__ grid, int *__restrict__ hdr, int gx, int gy, int gz, int nty, int ntz,
                                              int ntiles, int nbatch)
{
    __shared__ int lo[3], hi[3], cnt[4];
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)ntiles * nbatch;
    const int64_t work = (int64_t)blockIdx.x * total / gridDim.x;
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom((int)(work % ntiles), gx, gy, gz, nty, ntz);
    if (tid < 3) { lo[tid] = 0x7fffffff; hi[tid] = -0x7fffffff; }
    if (tid < 4) cnt[tid] = 0;
    float fl[VPT1][3];
    unsigned valid = 0;
    float rough = 0.f;                                               // sum over the thread's samples and the dims of |second difference along z|
    int nrough = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        int ox, oy, oz; float c[3];
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        if (ox < gx && oy < gy && oz < gz) valid |= 1u << v;
        load_xyz<GM>(p, grid, b, g, ox < gx ? ox : gx - 1, oy < gy ? oy : gy - 1, oz < gz ? oz : gz - 1, c);
        // (a row of 16 z-neighbours lies in 16 consecutive lanes: sample id = tid + 512 v)
        const bool mid = (tid & 15) >= 1 && (tid & 15) <= 14 && oz + 1 < gz && ox < gx && oy < gy;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float f = floorf(c[d] - 0.5f * (float)(K - 1));
            fl[v][d] = f == f ? __builtin_fmaxf(__builtin_fminf(f, 1073741824.f), -1073741824.f) : 0.f;
            const float d2 = __shfl_up(c[d], 1) - 2.f * c[d] + __shfl_down(c[d], 1);
            if (mid && d2 == d2) rough += __builtin_fminf(__builtin_fabsf(d2), 64.f);
        }
        if (mid) ++nrough;
    }
    __syncthreads();
    int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (!((valid >> v) & 1)) continue;
#pragma unroll
        for (int d = 0; d < 3; ++d) { const int i = __float2int_rz(fl[v][d]); mn[d] = i < mn[d] ? i : mn[d]; mx[d] = i > mx[d] ? i : mx[d]; }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = wave_min(mn[d]), e = wave_max(mx[d]);
        if ((tid & 63) == 0) { atomicMin(&lo[d], a); atomicMax(&hi[d], e); }
    }
    __syncthreads();
    const int cap[3] = { 33, 33, 32 };
    int l[3], h[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int a = lo[d], sz = hi[d] + K - a + 1;
        if (sz > cap[d]) { a += (sz - cap[d]) / 2; sz = cap[d]; }
        l[d] = a; h[d] = a + sz - K - 1;
    }
    int slow = 0, nv = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (!((valid >> v) & 1)) continue;
        ++nv;
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) { const int i = __float2int_rz(fl[v][d]); in = in && i >= l[d] && i <= h[d]; }
        if (!in) ++slow;
    }
    slow = wave_sum(slow); nv = wave_sum(nv);
    const int r256 = wave_sum((int)(rough * 16.f)), nr = wave_sum(nrough);
    if ((tid & 63) == 0) { atomicAdd(&cnt[0], slow); atomicAdd(&cnt[1], nv); atomicAdd(&cnt[2], r256); atomicAdd(&cnt[3], nr); }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&hdr[1], cnt[0]); atomicAdd(&hdr[2], cnt[1]); atomicAdd(&hdr[4], cnt[2] >> 4); atomicAdd(&hdr[5], cnt[3]);
        __threadfence();
        if (atomicAdd(&hdr[3], 1) == (int)gridDim.x - 1) {
            const int ns = atomicAdd(&hdr[1], 0), nn = atomicAdd(&hdr[2], 0);
            // rough: the mean |second difference| of the coordinates along z, summed over the dims, exceeds two voxels (i.i.d. noise of
            // sigma voxels: ~ 6 sigma -- sigma = 0.25: tiles 1.23 against 1.32 ms, sigma = 1: 1.78 / 1.54; registration fields, zooms, the identity: ~ 0)
            const int r2 = atomicAdd(&hdr[4], 0), n2 = atomicAdd(&hdr[5], 0);
            hdr[0] = ((int64_t)ns * 300 > nn || (int64_t)r2 > 2 * (int64_t)n2) ? 1 : 0;
        }
    }
}

struct GatSmem {
    int   taboff[3][BOX + 3];
    float tabsgn[3][BOX + 3];
    unsigned start[CAPD];
    int   pref[CAPD + 2];                      // records in front of each run of the brick; [nd ...]: all of them
    int   brick, qmax;                         // qmax: the fullest bank class (QUEUE)
    float2 box[BOX * PLANE + 64];              // 21 x 21 x 20 pair slots = 70 560 B (+ the quartic stencil's unused sixth row / plane)
    // gather5<K, MODE, QUEUE = true> only (the list-order kernels are launched with the bytes up to here):
    int   qcnt[32], qoff[33];                  // records per bank class of the brick, their offsets in the queue
    unsigned short queue[4480];                // the brick's records (index in its list) sorted by the class of their first pair slot
};
constexpr int QCAPG = 4480;                     // records of a brick that are walked in class order (a brick holds 4096 at unit density; more: in list order)
static_assert(sizeof(GatSmem) <= 80 * 1024, "two workgroups per CU");

#define IP_RD(o, off) "ds_read_b64 %" #o ", %18 offset:" #off "\n\t"
// the 36 taps of one x-plane of a stencil: six rows 160 bytes apart, three pairs per row
__device__ __forceinline__ void plane_reads(unsigned addr, f2 (&v)[18])
{
    static_assert(PZ * 8 == 160, "the immediate offsets are (row * PZ + 2 k) * 8");
    asm volatile(IP_RD(0, 0) IP_RD(1, 16) IP_RD(2, 32) IP_RD(3, 160) IP_RD(4, 176) IP_RD(5, 192)
                 IP_RD(6, 320) IP_RD(7, 336) IP_RD(8, 352) IP_RD(9, 480) IP_RD(10, 496) IP_RD(11, 512)
                 IP_RD(12, 640) IP_RD(13, 656) IP_RD(14, 672) IP_RD(15, 800) IP_RD(16, 816) IP_RD(17, 832)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
                   "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15]), "=&v"(v[16]), "=&v"(v[17])
                 : "v"(addr) : "memory");
}
#undef IP_RD
#ifdef IP_G5_HIGH
// orders 6 / 7: the 64 taps of one x-plane of a stencil -- eight rows 160 bytes apart, four pairs per row (two blocks: an asm statement takes
// at most 30 operands)
#define IP_RD(o, off) "ds_read_b64 %" #o ", %16 offset:" #off "\n\t"
__device__ __forceinline__ void plane_reads8(unsigned addr, f2 (&v)[32])
{
    static_assert(PZ * 8 == 160, "the immediate offsets are (row * PZ + 2 k) * 8");
    asm volatile(IP_RD(0, 0) IP_RD(1, 16) IP_RD(2, 32) IP_RD(3, 48) IP_RD(4, 160) IP_RD(5, 176) IP_RD(6, 192) IP_RD(7, 208)
                 IP_RD(8, 320) IP_RD(9, 336) IP_RD(10, 352) IP_RD(11, 368) IP_RD(12, 480) IP_RD(13, 496) IP_RD(14, 512) IP_RD(15, 528)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : "v"(addr) : "memory");
    asm volatile(IP_RD(0, 640) IP_RD(1, 656) IP_RD(2, 672) IP_RD(3, 688) IP_RD(4, 800) IP_RD(5, 816) IP_RD(6, 832) IP_RD(7, 848)
                 IP_RD(8, 960) IP_RD(9, 976) IP_RD(10, 992) IP_RD(11, 1008) IP_RD(12, 1120) IP_RD(13, 1136) IP_RD(14, 1152) IP_RD(15, 1168)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[16]), "=&v"(v[17]), "=&v"(v[18]), "=&v"(v[19]), "=&v"(v[20]), "=&v"(v[21]), "=&v"(v[22]), "=&v"(v[23]),
                   "=&v"(v[24]), "=&v"(v[25]), "=&v"(v[26]), "=&v"(v[27]), "=&v"(v[28]), "=&v"(v[29]), "=&v"(v[30]), "=&v"(v[31])
                 : "v"(addr) : "memory");
}
#undef IP_RD
#endif

// QUEUE (round 6, the pull of >= 400 taps per sample): the brick's records walked through a class-sorted queue, as scatter5's -- a ds_read_b64
// is served per half wave and is conflict-free when its 32 lanes read 32 different pair slots mod 32; all reads of a stencil add the same
// offsets in every lane, so lane q of every half wave walks the records whose FIRST slot has class q.  The price: in class order a wave's
// record reads and result stores are scattered where list order (a tile's samples side by side) half coalesces them.  Measured
// (tools/r6/gather5_queue.py, profiles/r06_orders_6_7.txt): order 7, two channels: 7.2 -> 5.7 ms; order 5, two channels: 3.68 -> 3.32;
// config 3 (one channel of 216 taps): 1.76 -> 1.85; the gradient modes lose (order 5 grid_grad 4.1 -> 5.0): they keep the list order,
// and their kernels (QUEUE = false) the code of round 5 -- the tap code is gather5_taps.inc, included by both walks.
template <int K, int MODE, bool QUEUE = false>
__global__ __launch_bounds__(NT, 4) void gather5(KParams p, Grid5 bg, const int *__restrict__ ndesc, const uint2 *__restrict__ desc,
                                                 const float4 *__restrict__ rec, const int *__restrict__ list, int *__restrict__ draw,
                                                 const float *__restrict__ img, float *__restrict__ out, const int *__restrict__ gate,
                                                 const float *__restrict__ gout)
{
    if (gate && *gate != 1) return;                                  // INTERPOL_FLAG_AUTO_SCATTER: the probe chose the tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    GatSmem &sm = *reinterpret_cast<GatSmem *>(smem_raw);
    const int nlist = list[0];
    for (;;) {
        const int tid = opaque((int)threadIdx.x);
        __syncthreads();                                             // the previous brick's readers are done
        if (tid == 0) { const int i = atomicAdd(draw, 1); sm.brick = i < nlist ? list[1 + i] : -1; }
        __syncthreads();
        const int bk = sm.brick;
        if (bk < 0) break;
        const int64_t b = bk / bg.per_item;
        int r = bk - (int)b * bg.per_item;
        const int bz = r % bg.nb[2]; r /= bg.nb[2];
        const int by = r % bg.nb[1], bx = r / bg.nb[1];
        const int b0[3] = { bx * BR - OFFB, by * BR - OFFB, bz * BR - OFFB };       // lattice index of box slot 0
        const int nd = min(ndesc[bk], CAPD);
        if (tid < 64) {
            // the runs of the brick and the exclusive prefix of their lengths: the threads walk the brick's records as ONE list
            static_assert(CAPD == 128, "two runs per lane");
            const int e0 = 2 * tid, e1 = e0 + 1;
            const uint2 d0 = e0 < nd ? desc[(int64_t)bk * CAPD + e0] : make_uint2(0u, 0u), d1 = e1 < nd ? desc[(int64_t)bk * CAPD + e1] : make_uint2(0u, 0u);
            const int sum = (int)d0.y + (int)d1.y;
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
            sm.start[e0] = d0.x; sm.start[e1] = d1.x;
            sm.pref[e0] = incl - sum; sm.pref[e1] = incl - (int)d1.y;
            if (tid == 63) { sm.pref[CAPD] = incl; sm.pref[CAPD + 1] = 0x7fffffff; }
        }
        if (tid >= 128 && tid < 128 + 3 * 64) {                      // box slot -> wrapped lattice offset and sign (bounds.py:30-89)
            const int d = (tid - 128) >> 6, slot = tid & 63;
            if (slot < BOX) {
                const long long pk = wrap_outofline(p.bound[d], (d == 0 ? b0[0] : d == 1 ? b0[1] : b0[2]) + slot, p.vol_n[d]);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] / 4);
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
        [[maybe_unused]] bool sorted_ = false;
        if constexpr (QUEUE) {
            if (tid < 32) sm.qcnt[tid] = 0;
            __syncthreads();                                         // prefix of the runs written, counters zero
            const int nq = sm.pref[CAPD];
            sorted_ = nq <= QCAPG && !(p.dbg & 64);                  // (block-uniform; debug bit 64: list order, the A/B)
            if (sorted_) {
                int rq = 0;
                for (int j = tid; j < nq; j += NT) {
                    while (j >= sm.pref[rq + 1]) ++rq;
                    const float4 rc = rec[sm.start[rq] + (unsigned)(j - sm.pref[rq])];
                    int cx = __float2int_rz(floorf(rc.x - 0.5f * (float)(K - 1))) - b0[0], cy = __float2int_rz(floorf(rc.y - 0.5f * (float)(K - 1))) - b0[1],
                        cz = __float2int_rz(floorf(rc.z - 0.5f * (float)(K - 1))) - b0[2];
                    cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                    atomicAdd(&sm.qcnt[((cx * BOX + cy) * PZ + cz) & 31], 1);
                }
                __syncthreads();
                if (tid < 32) {
                    const int cq = sm.qcnt[tid];
                    int tot;
                    const int off = half_excl_scan(cq, tot);
                    sm.qoff[tid] = off;
                    int mx = cq;
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { const int t = __shfl_xor(mx, o, 32); mx = t > mx ? t : mx; }
                    if (tid == 0) { sm.qoff[32] = tot; sm.qmax = mx; }
                    sm.qcnt[tid] = 0;                                // (becomes the fill position)
                }
                __syncthreads();
                rq = 0;
                for (int j = tid; j < nq; j += NT) {
                    while (j >= sm.pref[rq + 1]) ++rq;
                    const float4 rc = rec[sm.start[rq] + (unsigned)(j - sm.pref[rq])];
                    int cx = __float2int_rz(floorf(rc.x - 0.5f * (float)(K - 1))) - b0[0], cy = __float2int_rz(floorf(rc.y - 0.5f * (float)(K - 1))) - b0[1],
                        cz = __float2int_rz(floorf(rc.z - 0.5f * (float)(K - 1))) - b0[2];
                    cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                    const int q = ((cx * BOX + cy) * PZ + cz) & 31;
                    sm.queue[sm.qoff[q] + atomicAdd(&sm.qcnt[q], 1)] = (unsigned short)j;
                }
            }
        }
        for (int c = 0; c < p.C; ++c) {
            const float *vc = img + b * p.vol_sb + (int64_t)c * p.vol_sc;
            float *oc = out + b * p.val_sb + (int64_t)c * p.val_sc;
            __syncthreads();                                         // tables (queue) written / the previous channel's readers are done
            // rows that are contiguous runs of the image's unit-stride dim with sign +1: five quads of pair slots per row from a
            // 16-byte load and the value behind it; else slot by slot through the z table
            const bool zlin = p.vol_ss[2] == 4 && b0[2] >= (p.bound[2] == B_DST1 ? 1 : 0) && b0[2] + BOX <= p.vol_n[2];
            if (zlin) {
                for (int e = tid; e < BOX * BOX * 5; e += NT) {
                    const int row = e / 5, q = e - row * 5;
                    const int x = row / BOX, y = row - x * BOX;
                    const int off = sm.taboff[0][x] + sm.taboff[1][y] + b0[2] + 4 * q;
                    const float sg = sm.tabsgn[0][x] * sm.tabsgn[1][y];
                    const float4 a = ld4<float>(vc + off);
                    const float n = vc[off + 4] * sg;
                    float4 *dst = reinterpret_cast<float4 *>(sm.box + row * PZ + 4 * q);
                    dst[0] = make_float4(a.x * sg, a.y * sg, a.y * sg, a.z * sg);
                    dst[1] = make_float4(a.z * sg, a.w * sg, a.w * sg, n);
                }
            } else {
                for (int e = tid; e < BOX * BOX * PZ; e += NT) {
                    const int x = e / (BOX * PZ), y = (e / PZ) % BOX, z = e % PZ;
                    const int o0 = sm.taboff[0][x] + sm.taboff[1][y];
                    const float sg = sm.tabsgn[0][x] * sm.tabsgn[1][y];
                    sm.box[e] = make_float2(vc[o0 + sm.taboff[2][z]] * (sg * sm.tabsgn[2][z]), vc[o0 + sm.taboff[2][z + 1]] * (sg * sm.tabsgn[2][z + 1]));
                }
            }
            __syncthreads();
            const int ntot = sm.pref[CAPD];
            int rr = 0;
            if constexpr (!QUEUE) {
            for (int j = tid; j < ntot; j += NT) {
                {
                    while (j >= sm.pref[rr + 1]) ++rr;               // (runs beyond the last hold nothing: their prefix is the total)
#include "gather5_taps.inc"
                }
            }
            } else {
            const int q_ = tid & 31, hw = tid >> 5;
            const int qbeg = sorted_ ? sm.qoff[q_] : 0, qn = sorted_ ? sm.qoff[q_ + 1] - qbeg : 0;
            const int nwalk = sorted_ ? (sm.qmax + NT / 32 - 1) / (NT / 32) : (ntot + NT - 1) / NT;     // (block-uniform)
            for (int it = 0; it < nwalk; ++it) {
                int j;
                if (sorted_) {
                    const int i = hw + it * (NT / 32);
                    if (i >= qn) continue;
                    j = sm.queue[qbeg + i];
                    rr = 0;                                          // the run of record j: the last one whose prefix is <= j
#pragma unroll
                    for (int st = CAPD / 2; st > 0; st >>= 1) rr += sm.pref[rr + st] <= j ? st : 0;
                } else {                                             // (a brick beyond the queue's capacity: list order)
                    j = tid + it * NT;
                    if (j >= ntot) continue;
                    while (j >= sm.pref[rr + 1]) ++rr;
                }
                {
#include "gather5_taps.inc"
                }
            }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// scatter5 (round 5) -- grid_push / grid_count (nd.py:146-213, pushpull.py:106-142) of orders 4 and 5 through bricks of the TARGET:
// the adjoint of gather5 and the deformation-independent scatter of these orders (the LDS tiles of ops_tiled.hip leave samples
// outside their box to a slow list, then to per-thread atomics: BASELINE config 3's image gradient 4.2 ms at sigma = 2 and an order
// of magnitude more at sigma = 6).  bin5 sorts the samples as for the gathers (records x, y, z, sample index); a workgroup draws a
// non-empty brick, counts the density of first-tap cells (stencil counts per slot: the box filter of it), and adds every record's
// (K + 1)^3 taps into the brick's 21^3 box of 32-bit sums in the MAGIC format of push_owner.hip (the float  t + 1.5 * 2^23  IS the
// fixed-point addend: one v_fma_f32 + one ds_add_u32 per tap, one channel per pass); the box is flushed with float atomics through the
// boundary tables (boxes of neighbouring bricks overlap by K points: 2.3 atomics per lattice point and channel -- the tiles' flush
// issues more).  Bricks too dense for 32-bit sums and non-finite sources are scattered tap by tap (always correct).
// ---------------------------------------------------------------------------
constexpr int NZ5 = BOX + 1;                    // row pitch of the stencil counts (16 bits each, two to a word)
constexpr int NCELL5 = IP_G5_CP * IP_G5_CP * IP_G5_CP;
constexpr int QCAP5 = 6144;                     // records of a brick that are walked in class order (more: in list order)
constexpr float MAGIC5 = 12582912.f;            // 1.5 * 2^23: bits 0x4B400000
constexpr unsigned MAGIC5_BITS = 0x4B400000u;
struct ScatSmem {
    int   taboff[3][BOX + 3];
    float tabsgn[3][BOX + 3];
    unsigned start[CAPD];
    int   pref[CAPD + 2];
    int   brick, dmax, amax, nonfinite;
    int   qcnt[32], qoff[33], qmax;            // records per bank class of the brick, their offsets in the queue, the fullest class
    unsigned short queue[QCAP5];               // the brick's records (index in its list) sorted by the class of their first slot
    union alignas(16) {
        unsigned cells[NCELL5 / 2];             // density of first-tap cells: 16-bit counters
        unsigned nreg[BOX * BOX * NZ5 / 2];     // stencils per slot, 16 bits each: slot (x, y, z) at (x * BOX + y) * NZ5 + z
    };
    unsigned box[BOX * BOX * BOX];              // 37 044 B
};
static_assert(sizeof(ScatSmem) <= 80 * 1024, "two workgroups per CU");

#ifdef IP_G5_HIGH
// wmax = the spline at 0: 151 / 315 (order 7), 5887 / 11520 (order 6)
template <int K> __device__ __forceinline__ float units5() { return K == 7 ? 4194304.f * 0.999f / (0.4793651f * 0.4793651f * 0.4793651f) : 4194304.f * 0.999f / (0.5110244f * 0.5110244f * 0.5110244f); }
#else
template <int K> __device__ __forceinline__ float units5() { return K == 5 ? 4194304.f * 0.999f / (0.55f * 0.55f * 0.55f) : 4194304.f * 0.999f / (0.5989584f * 0.5989584f * 0.5989584f); }
#endif
template <int K> __device__ __forceinline__ float cbmax5() { return 2147483648.f * 0.99f / units5<K>(); }

typedef unsigned short us2_5 __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ void slide5(const unsigned *in, unsigned *out)
{
    us2_5 s = { 0, 0 };
#pragma unroll
    for (int j = 0; j < BOX; ++j) {
        if (j < IP_G5_CP) s += __builtin_bit_cast(us2_5, in[j]);
        if (j >= W) s -= __builtin_bit_cast(us2_5, in[j - W]);
        out[j] = __builtin_bit_cast(unsigned, s);
    }
}
// stencils per slot = the (K + 1)^3 box filter of the cell density, dim after dim, in place (push_owner.hip: stencil_counts)
template <int K>
__device__ __forceinline__ void counts5(ScatSmem &sm, int tid)
{
    constexpr int W = K + 1, HZ = NZ5 / 2;
    unsigned *rg = sm.nreg;
    unsigned in[IP_G5_CP], out[BOX];
    if (tid < IP_G5_CP * IP_G5_CP) {
        const uint4 lo = reinterpret_cast<const uint4 *>(sm.cells)[2 * tid], hi = reinterpret_cast<const uint4 *>(sm.cells)[2 * tid + 1];
        in[0] = lo.x; in[1] = lo.y; in[2] = lo.z; in[3] = lo.w; in[4] = hi.x; in[5] = hi.y; in[6] = hi.z; in[7] = hi.w;
    }
    __syncthreads();
    if (tid < IP_G5_CP * IP_G5_CP) {
        int v[IP_G5_CP], o[NZ5], s = 0;
#pragma unroll
        for (int i = 0; i < IP_G5_CP / 2; ++i) { v[2 * i] = (int)(in[i] & 0xffffu); v[2 * i + 1] = (int)(in[i] >> 16); }
#pragma unroll
        for (int j = 0; j < BOX; ++j) { if (j < IP_G5_CP) s += v[j]; if (j >= W) s -= v[j - W]; o[j] = s; }
        o[BOX] = 0;
#pragma unroll
        for (int i = 0; i < HZ; ++i) rg[tid * HZ + i] = (unsigned)o[2 * i] | ((unsigned)o[2 * i + 1] << 16);
    }
    __syncthreads();
    {
        const int x = tid / HZ, zp = tid - x * HZ;
        if (tid < IP_G5_CP * HZ) {
#pragma unroll
            for (int k = 0; k < IP_G5_CP; ++k) in[k] = rg[(x * IP_G5_CP + k) * HZ + zp];
        }
        __syncthreads();
        if (tid < IP_G5_CP * HZ) {
            slide5<W>(in, out);
#pragma unroll
            for (int j = 0; j < BOX; ++j) rg[(x * BOX + j) * HZ + zp] = out[j];
        }
        __syncthreads();
    }
    {
        const int y = tid / HZ, zp = tid - y * HZ;
        if (tid < BOX * HZ) {
#pragma unroll
            for (int k = 0; k < IP_G5_CP; ++k) in[k] = rg[(k * BOX + y) * HZ + zp];
        }
        __syncthreads();
        if (tid < BOX * HZ) {
            slide5<W>(in, out);
#pragma unroll
            for (int j = 0; j < BOX; ++j) rg[(j * BOX + y) * HZ + zp] = out[j];
        }
        __syncthreads();
    }
}

// the K + 1 adds of one row of the stencil at immediate offsets
template <int K, int I, int J>
__device__ __forceinline__ void row_adds5(unsigned addr, const float *v)
{
    constexpr int o = ((I * BOX + J) * BOX) * 4;
#ifdef IP_G5_HIGH
    asm volatile("ds_add_u32 %0, %1 offset:%8\n\tds_add_u32 %0, %2 offset:%9\n\tds_add_u32 %0, %3 offset:%10\n\tds_add_u32 %0, %4 offset:%11\n\t"
                 "ds_add_u32 %0, %5 offset:%12\n\tds_add_u32 %0, %6 offset:%13\n\tds_add_u32 %0, %7 offset:%14"
                 :: "v"(addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]),
                    "n"(o), "n"(o + 4), "n"(o + 8), "n"(o + 12), "n"(o + 16), "n"(o + 20), "n"(o + 24) : "memory");
    if (K == 7) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(addr), "v"(v[7]), "n"(o + 28) : "memory");
    return;
#endif
    if (K == 5)
        asm volatile("ds_add_u32 %0, %1 offset:%7\n\tds_add_u32 %0, %2 offset:%8\n\tds_add_u32 %0, %3 offset:%9\n\tds_add_u32 %0, %4 offset:%10\n\t"
                     "ds_add_u32 %0, %5 offset:%11\n\tds_add_u32 %0, %6 offset:%12"
                     :: "v"(addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]),
                        "n"(o), "n"(o + 4), "n"(o + 8), "n"(o + 12), "n"(o + 16), "n"(o + 20) : "memory");
    else
        asm volatile("ds_add_u32 %0, %1 offset:%6\n\tds_add_u32 %0, %2 offset:%7\n\tds_add_u32 %0, %3 offset:%8\n\tds_add_u32 %0, %4 offset:%9\n\t"
                     "ds_add_u32 %0, %5 offset:%10"
                     :: "v"(addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]),
                        "n"(o), "n"(o + 4), "n"(o + 8), "n"(o + 12), "n"(o + 16) : "memory");
}
template <int K, int I, int J>
__device__ __forceinline__ void scatter_row5(unsigned addr, float sx, const float *wy, const float *wz)
{
    const float sy = sx * wy[J];
#ifdef IP_G5_HIGH
    float v[8];
#else
    float v[6];
#endif
#pragma unroll
    for (int k = 0; k <= K; ++k) v[k] = __builtin_fmaf(sy, wz[k], MAGIC5);
    row_adds5<K, I, J>(addr, v);
}
template <int K, int I>
__device__ __forceinline__ void scatter_plane5(unsigned addr, float s, float wxi, const float *wy, const float *wz)
{
    const float sx = s * wxi;
    scatter_row5<K, I, 0>(addr, sx, wy, wz); scatter_row5<K, I, 1>(addr, sx, wy, wz); scatter_row5<K, I, 2>(addr, sx, wy, wz);
    scatter_row5<K, I, 3>(addr, sx, wy, wz); scatter_row5<K, I, 4>(addr, sx, wy, wz);
    if (K >= 5) scatter_row5<K, I, 5>(addr, sx, wy, wz);
#ifdef IP_G5_HIGH
    scatter_row5<K, I, 6>(addr, sx, wy, wz);
    if (K == 7) scatter_row5<K, I, 7>(addr, sx, wy, wz);
#endif
}

template <int K>
__global__ __launch_bounds__(NT, 4) void scatter5(KParams p, Grid5 bg, const int *__restrict__ ndesc, const uint2 *__restrict__ desc,
                                                  const float4 *__restrict__ rec, const int *__restrict__ list, int *__restrict__ draw,
                                                  const float *__restrict__ src, float *__restrict__ vol, const int *__restrict__ gate)
{
    if (gate && *gate != 1) return;                                  // INTERPOL_FLAG_AUTO_SCATTER: the probe chose the tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ScatSmem &sm = *reinterpret_cast<ScatSmem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = K; }
    L.lin = 0;
    const int nlist = list[0];
    const int nch = p.C + p.cc;
    for (int e = threadIdx.x; e < BOX * BOX * BOX; e += NT) sm.box[e] = 0u;
    for (;;) {
        const int tid = opaque((int)threadIdx.x);
        __syncthreads();                                             // the previous brick is flushed
        if (tid == 0) { const int i = atomicAdd(draw, 1); sm.brick = i < nlist ? list[1 + i] : -1; sm.dmax = 0; }
        __syncthreads();
        const int bk = sm.brick;
        if (bk < 0) break;
        const int64_t b = bk / bg.per_item;
        int r = bk - (int)b * bg.per_item;
        const int bz = r % bg.nb[2]; r /= bg.nb[2];
        const int by = r % bg.nb[1], bx = r / bg.nb[1];
        const int b0[3] = { bx * BR - OFFB, by * BR - OFFB, bz * BR - OFFB };
        const int nd = min(ndesc[bk], CAPD);
        if (tid < 64) {
            static_assert(CAPD == 128, "two runs per lane");
            const int e0 = 2 * tid, e1 = e0 + 1;
            const uint2 d0 = e0 < nd ? desc[(int64_t)bk * CAPD + e0] : make_uint2(0u, 0u), d1 = e1 < nd ? desc[(int64_t)bk * CAPD + e1] : make_uint2(0u, 0u);
            const int sum = (int)d0.y + (int)d1.y;
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
            sm.start[e0] = d0.x; sm.start[e1] = d1.x;
            sm.pref[e0] = incl - sum; sm.pref[e1] = incl - (int)d1.y;
            if (tid == 63) { sm.pref[CAPD] = incl; sm.pref[CAPD + 1] = 0x7fffffff; }
        }
        if (tid >= 128 && tid < 128 + 3 * 64) {                      // box slot -> wrapped lattice offset and sign (bounds.py:30-89)
            const int d = (tid - 128) >> 6, slot = tid & 63;
            if (slot < BOX) {
                const long long pk = wrap_outofline(p.bound[d], (d == 0 ? b0[0] : d == 1 ? b0[1] : b0[2]) + slot, p.vol_n[d]);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] / 4);
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        for (int e = tid; e < NCELL5 / 2; e += NT) sm.cells[e] = 0u;
        if (tid < 32) sm.qcnt[tid] = 0;
        __syncthreads();
        const int ntot = sm.pref[CAPD];
        const bool sorted_ = ntot <= QCAP5;                          // (block-uniform)
        // ---- density of the first-tap cells
        {
            int rr = 0;
            for (int j = tid; j < ntot; j += NT) {
                while (j >= sm.pref[rr + 1]) ++rr;
                const float4 rc = rec[sm.start[rr] + (unsigned)(j - sm.pref[rr])];
                int cx = __float2int_rz(floorf(rc.x - 0.5f * (float)(K - 1))) - b0[0], cy = __float2int_rz(floorf(rc.y - 0.5f * (float)(K - 1))) - b0[1],
                    cz = __float2int_rz(floorf(rc.z - 0.5f * (float)(K - 1))) - b0[2];
                cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                const int cell = (cx * IP_G5_CP + cy) * IP_G5_CP + cz;
                atomicAdd(&sm.cells[cell >> 1], 1u << (16 * (cell & 1)));
                if (sorted_) atomicAdd(&sm.qcnt[((cx * BOX + cy) * BOX + cz) & 31], 1);
            }
        }
        __syncthreads();
        // ---- class-sorted queue of the brick's records: lane q of every half wave walks class q (first slot mod 32: the 32 lanes of a
        // half wave then hit 32 different banks and every tap adds the same offset -- ds_add_u32 at 4.2 clk instead of 7.5, push_owner.hip)
        if (sorted_) {
            if (tid < 32) {
                const int cq = sm.qcnt[tid];
                int tot;
                const int off = half_excl_scan(cq, tot);
                sm.qoff[tid] = off;
                int mx = cq;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { const int t = __shfl_xor(mx, o, 32); mx = t > mx ? t : mx; }
                if (tid == 0) { sm.qoff[32] = tot; sm.qmax = mx; }
                sm.qcnt[tid] = 0;                                    // (becomes the fill position)
            }
            __syncthreads();
            int rr = 0;
            for (int j = tid; j < ntot; j += NT) {
                while (j >= sm.pref[rr + 1]) ++rr;
                const float4 rc = rec[sm.start[rr] + (unsigned)(j - sm.pref[rr])];
                int cx = __float2int_rz(floorf(rc.x - 0.5f * (float)(K - 1))) - b0[0], cy = __float2int_rz(floorf(rc.y - 0.5f * (float)(K - 1))) - b0[1],
                    cz = __float2int_rz(floorf(rc.z - 0.5f * (float)(K - 1))) - b0[2];
                cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                const int q = ((cx * BOX + cy) * BOX + cz) & 31;
                sm.queue[sm.qoff[q] + atomicAdd(&sm.qcnt[q], 1)] = (unsigned short)j;
            }
            __syncthreads();
        }
        {
            int dm = 0;
            for (int e = tid; e < NCELL5 / 2; e += NT) {
                const unsigned w2 = sm.cells[e];
                const int a = (int)(w2 & 0xffffu), c2 = (int)(w2 >> 16);
                dm = a > dm ? a : dm; dm = c2 > dm ? c2 : dm;
            }
            dm = wave_max(dm);
            if ((tid & 63) == 0 && dm > 0) atomicMax(&sm.dmax, dm);
        }
        __syncthreads();
        // 32-bit sums hold while density * prod_d sum_j max_t w_j(t) units of max |source| fit (tile_common.hpp: headroom32)
#ifdef IP_G5_HIGH
        const float wsum = K == 7 ? 1.4793651f : 1.5110244f;
#else
        const float wsum = K == 5 ? 1.55f : 1.5989584f;
#endif
        const bool dense = ntot >= 60000 || (float)sm.dmax * (wsum * wsum * wsum) > cbmax5<K>();           // (block-uniform)
        if (!dense) counts5<K>(sm, tid);
        const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
        for (int c = 0; c < nch; ++c) {
            const float *sc = (src && c < p.C) ? src + b * p.val_sb + (int64_t)c * p.val_sc : nullptr;      // NULL: the count (source = mask)
            float *vc = vol + b * p.vol_sb + (int64_t)c * p.vol_sc;
            if (tid == 0) { sm.amax = 0; sm.nonfinite = 0; }
            __syncthreads();
            // ---- max |masked source| of the brick's records
            {
                int rr = 0, am = 0; bool fin = true;
                for (int j = tid; j < ntot; j += NT) {
                    while (j >= sm.pref[rr + 1]) ++rr;
                    const float4 rc = rec[sm.start[rr] + (unsigned)(j - sm.pref[rr])];
                    const float xyz[3] = { rc.x, rc.y, rc.z };
                    const float v = (sc ? sc[__float_as_int(rc.w)] : 1.f) * inb_mask(p, xyz);
                    const int a = __float_as_int(__builtin_fabsf(v));
                    if ((a & 0x7f800000) == 0x7f800000) fin = false; else am = a > am ? a : am;
                }
                am = wave_max(am);
                if ((tid & 63) == 0 && am) atomicMax(&sm.amax, am);
                if (!fin) sm.nonfinite = 1;
            }
            __syncthreads();
            const int mb = sm.amax;
            const bool direct = dense || sm.nonfinite != 0;          // (block-uniform)
            if (mb == 0 && !direct) continue;                        // nothing but zeros
            const float a0 = fmaxf(__int_as_float(mb), 1e-27f);
            const float scale = units5<K>() / a0, inv = a0 * (1.f / units5<K>());
            // ---- the taps
            {
                int rr = 0;
                const int q = tid & 31, hw = tid >> 5;
                const int qbeg = sorted_ ? sm.qoff[q] : 0, qn = sorted_ ? sm.qoff[q + 1] - qbeg : 0;
                const int nwalk = sorted_ ? (sm.qmax + NT / 32 - 1) / (NT / 32) : (ntot + NT - 1) / NT;     // (block-uniform)
                for (int it = 0; it < nwalk; ++it) {
                    int j;
                    if (sorted_) {
                        const int i = hw + it * (NT / 32);
                        if (i >= qn) continue;
                        j = sm.queue[qbeg + i];
                        rr = 0;                                      // the run of record j: the last one whose prefix is <= j
#pragma unroll
                        for (int st = CAPD / 2; st > 0; st >>= 1) rr += sm.pref[rr + st] <= j ? st : 0;
                    } else {
                        j = tid + it * NT;
                        if (j >= ntot) continue;
                        while (j >= sm.pref[rr + 1]) ++rr;
                    }
                    const float4 rc = rec[sm.start[rr] + (unsigned)(j - sm.pref[rr])];
                    const float fx = floorf(rc.x - 0.5f * (float)(K - 1)), fy = floorf(rc.y - 0.5f * (float)(K - 1)), fz = floorf(rc.z - 0.5f * (float)(K - 1));
                    const float tx = rc.x - fx, ty = rc.y - fy, tz = rc.z - fz;
                    const float xyz[3] = { rc.x, rc.y, rc.z };
                    const float sv = (sc ? sc[__float_as_int(rc.w)] : 1.f) * inb_mask(p, xyz);
                    // (a zero source adds MAGIC to every slot of its stencil like any other record: the stencil counts include it)
                    if (direct) {
                        if (sv != 0.f) tiled::scatter_one_thread(L, vc, sv, __float2int_rz(fx), __float2int_rz(fy), __float2int_rz(fz), tx, ty, tz);
                        continue;
                    }
                    int cx = __float2int_rz(fx) - b0[0], cy = __float2int_rz(fy) - b0[1], cz = __float2int_rz(fz) - b0[2];
                    cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                    const unsigned addr = boxaddr + (unsigned)((cx * BOX + cy) * BOX + cz) * 4u;
#ifdef IP_G5_HIGH
                    float wy[8], wz[8];
                    wy[7] = 0.f; wz[7] = 0.f;
#else
                    float wy[6], wz[6];
                    wy[5] = 0.f; wz[5] = 0.f;
#endif
                    tiled::weights<K>(0, K, ty, wy);
                    tiled::weights<K>(0, K, tz, wz);
                    const float ss = sv * scale;
                    scatter_plane5<K, 0>(addr, ss, tiled::weight1(0, K, tx, 0, tiled::tap_piece(K, 0)), wy, wz);
                    scatter_plane5<K, 1>(addr, ss, tiled::weight1(0, K, tx, 1, tiled::tap_piece(K, 1)), wy, wz);
                    scatter_plane5<K, 2>(addr, ss, tiled::weight1(0, K, tx, 2, tiled::tap_piece(K, 2)), wy, wz);
                    scatter_plane5<K, 3>(addr, ss, tiled::weight1(0, K, tx, 3, tiled::tap_piece(K, 3)), wy, wz);
                    scatter_plane5<K, 4>(addr, ss, tiled::weight1(0, K, tx, 4, tiled::tap_piece(K, 4)), wy, wz);
                    if (K >= 5) scatter_plane5<K, 5>(addr, ss, tiled::weight1(0, K, tx, 5, tiled::tap_piece(K, 5)), wy, wz);
#ifdef IP_G5_HIGH
                    scatter_plane5<K, 6>(addr, ss, tiled::weight1(0, K, tx, 6, tiled::tap_piece(K, 6)), wy, wz);
                    if (K == 7) scatter_plane5<K, 7>(addr, ss, tiled::weight1(0, K, tx, 7, tiled::tap_piece(K, 7)), wy, wz);
#endif
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- flush: a slot that n stencils cover holds n * MAGIC_BITS + the sum of its addends (mod 2^32)
            if (!direct) {
                const unsigned short *nn = reinterpret_cast<const unsigned short *>(sm.nreg);
                for (int e = tid; e < BOX * BOX * BOX; e += NT) {
                    const int x = e / (BOX * BOX), rem = e - x * (BOX * BOX), y = rem / BOX, z = rem - y * BOX;
                    const unsigned n = nn[(x * BOX + y) * NZ5 + z];
                    if (n == 0u) continue;
                    const unsigned w = sm.box[e];
                    sm.box[e] = 0u;
                    const int sq = (int)(w - n * MAGIC5_BITS);
                    if (sq == 0) continue;
                    const float sg = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sm.tabsgn[2][z];
                    __hip_atomic_fetch_add(vc + (sm.taboff[0][x] + sm.taboff[1][y] + sm.taboff[2][z]), (float)sq * (inv * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

static bool eligible(const interpol_problem *p, const KParams &k)
{
    if (p->dim != 3 || p->dtype != INTERPOL_F32 || p->grid_dtype != INTERPOL_F32 || p->batch > 4096) return false;
    if (!(p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER))) return false;
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2] || k.order[0] < IP_G5_KLO || k.order[0] > IP_G5_KHI) return false;
    if (p->val_stride[0] < 0) return false;
    int64_t n = 1, nt = p->batch, nb = p->batch;
    for (int d = 0; d < 3; ++d) {
        if (p->grid_shape[d] > 0x7fffffff / 4) return false;
        n *= p->grid_shape[d];
        nt *= (p->grid_shape[d] + TS - 1) / TS;
        nb *= (p->vol_shape[d] + 2 * OFFB + BR - 1) / BR;
    }
    if (n < 4096 || nt * NS > 0x7fffffffll || nb > 0x7fffffffll / CAPD) return false;
    if ((uint64_t)n * 12ull > 0xffffffffull) return false;
    return true;
}

} // namespace g5 / g7

int64_t gather7_workspace_bytes(const interpol_problem *p, const KParams &k);
int try_gather7(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes,
                int mode, const void *gout, hipStream_t st, const int **gate_out);

int64_t IP_G5_WSB(const interpol_problem *p, const KParams &k)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return gather7_workspace_bytes(p, k);      // orders 6 - 7: this file's second compilation (gather7.hip)
#endif
    if (!IP_G5_NS::eligible(p, k)) return 0;
    int64_t nt = 1;
    for (int d = 0; d < 3; ++d) nt *= (p->grid_shape[d] + sorted::TS - 1) / sorted::TS;
    return IP_G5_NS::layout(IP_G5_NS::brick_grid(k), (int)p->batch, nt, nullptr, nullptr);
}

// grid_pull (grad == false) / grid_grad through the bricks: 1 = done, 0 = declined, else an error
// INTERPOL_FLAG_AUTO_SCATTER: 2 = launched behind the probe's verdict -- the caller launches the tile / generic kernels as well, with
// KParams::gate = *gate_out and gate_n = -1 (they return at once when the verdict is 1).
// mode 0: grid_pull, 2: grid_grad, 1: the grid gradient of the pull's backward (val := the dense (B, *out, 3) gradient, gout := grad_out)
int IP_G5_TRY(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes,
              int mode, const void *gout, hipStream_t st, const int **gate_out)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return try_gather7(p, k, vol, grid, val, workspace, workspace_bytes, mode, gout, st, gate_out);
#endif
    const bool grad = mode != 0;
    using namespace IP_G5_NS;
    if (!workspace || ((uintptr_t)workspace & 255u) != 0 || !eligible(p, k)) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = ((gx + TS - 1) / TS) * nty * ntz;
    const Grid5 bg = brick_grid(k);
    Workspace w;
    if (layout(bg, (int)p->batch, ntiles, workspace, &w) > workspace_bytes) return 0;
    const int64_t nz = 64 + 2 * w.nbricks + 1;
    if (nz > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(zero5, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, w.hdr, (int)nz);
    // order 5: the bricks beat the LDS tiles on every field measured (8 x 1 x 192^3: identity 1.37 against 1.59 ms, sigma = 2 1.77 / 2.58,
    // zoom 2 3.5 / 9.3; grad 1.70 / 2.02, 1.95 / 2.89 -- a tie at zoom 1.5): no probe, no tiles.  Order 4: the tiles keep smooth fields
    // (identity 1.19 against 1.27 ms, zoom 1.2 1.23 / 1.51), the probe gives rough ones to the bricks.
    // grid_grad of order 4: the bricks as well (identity 1.50 against 1.54, sigma = 2 1.72 / 2.31).
#ifdef IP_G5_HIGH
    const bool gated = false && grad;                              // orders 6 - 7: the bricks always (the tiles behind them are the 8^3 ones of round 1)
#else
    const bool gated = !(p->flags & INTERPOL_FLAG_BINNED_SCATTER) && k.order[0] == 4 && !grad;
#endif
    const int *gate = gated ? w.hdr : nullptr;
#ifndef IP_G5_HIGH
    if (gated) {
        const long long total = (long long)ntiles * p->batch;
        const dim3 pgrid((unsigned)(total < NPROBE ? total : NPROBE));
#define IP_P5(KK, GM) hipLaunchKernelGGL((probe5<KK, GM>), pgrid, dim3(NT1), 0, st, k, (const float *)grid, w.hdr, gx, gy, gz, nty, ntz, ntiles, (int)p->batch);
#define IP_P5_GM(KK) { if (k.sep == 0) IP_P5(KK, 0) else if (k.sep == 1) IP_P5(KK, 1) else if (k.sep == 2) IP_P5(KK, 2) else IP_P5(KK, 3) }
        if (k.order[0] == 5) IP_P5_GM(5) else IP_P5_GM(4)
#undef IP_P5_GM
#undef IP_P5
    }
#endif
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    const long long want = 2ll * cu_count();
    const dim3 ggrid((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_G5(KK, GM, MD)                                                                                               \
    {                                                                                                                   \
        hipLaunchKernelGGL((bin5<KK, GM, MD>), tgrid, dim3(NT1), 0, st, k, bg, (const float *)vol, (const float *)grid, (float *)val, \
                           w.ndesc, w.list, w.desc, w.rec, gx, gy, gz, nty, ntz, ntiles, gate, (const float *)gout, (float *)nullptr);   \
        if (MD == 0 && k.C * (KK + 1) * (KK + 1) * (KK + 1) >= 400) {   /* the class-sorted walk pays from there on (see the kernel) */ \
            const int attr = big_lds<gather5<KK, 0, true>>(sizeof(GatSmem));                                            \
            if (attr) return attr;                                                                                      \
            hipLaunchKernelGGL((gather5<KK, 0, true>), ggrid, dim3(NT), sizeof(GatSmem), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                               (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)vol, (float *)val, gate, (const float *)gout); \
        } else {                                                                                                        \
            const int attr = big_lds<gather5<KK, MD>>(offsetof(GatSmem, qcnt));                                         \
            if (attr) return attr;                                                                                      \
            hipLaunchKernelGGL((gather5<KK, MD>), ggrid, dim3(NT), offsetof(GatSmem, qcnt), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                               (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)vol, (float *)val, gate, (const float *)gout); \
        }                                                                                                               \
    }
#define IP_G5_GM(KK, MD) { if (k.sep == 0) IP_G5(KK, 0, MD) else if (k.sep == 1) IP_G5(KK, 1, MD) else if (k.sep == 2) IP_G5(KK, 2, MD) else IP_G5(KK, 3, MD) }
    if (k.order[0] == IP_G5_KHI) { if (mode == 2) IP_G5_GM(IP_G5_KHI, 2) else if (mode == 1) IP_G5_GM(IP_G5_KHI, 1) else IP_G5_GM(IP_G5_KHI, 0) }
    else { if (mode == 2) IP_G5_GM(IP_G5_KLO, 2) else if (mode == 1) IP_G5_GM(IP_G5_KLO, 1) else IP_G5_GM(IP_G5_KLO, 0) }
#undef IP_G5_GM
#undef IP_G5
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (gate_out) *gate_out = gate;
    return gated ? 2 : 1;
}

// grid_push (val != NULL) / grid_count of orders 4 and 5 through bricks of the target (scatter5): 1 = done, 0 = declined, else an error.
// The target `vol` (float, zeroed or accumulated into by the caller) takes p->channels (+ 1 with k.cc) channels.
int64_t scatter7_workspace_bytes(const interpol_problem *p, const KParams &k);
int try_scatter7(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, void *workspace, int64_t workspace_bytes,
                 hipStream_t st, const int **gate_out);

int64_t IP_S5_WSB(const interpol_problem *p, const KParams &k)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return scatter7_workspace_bytes(p, k);     // orders 6 - 7: this file's second compilation (gather7.hip)
#endif
    interpol_problem q = *p;
    q.val_stride[0] = 0;                                             // (the gathers' test of the output strides does not apply)
    if (!IP_G5_NS::eligible(&q, k)) return 0;
    // the bricks when there is at least a quarter of a sample per target voxel (sparser: the tiles / the target-stationary splatting)
    int64_t n = 1, nv = 1, nt = 1;
    for (int d = 0; d < 3; ++d) { n *= p->grid_shape[d]; nv *= p->vol_shape[d]; nt *= (p->grid_shape[d] + sorted::TS - 1) / sorted::TS; }
    if (4 * n < nv || p->vol_stride[0] == 0) return 0;               // (a shared target: not here)
    return IP_G5_NS::layout(IP_G5_NS::brick_grid(k), (int)p->batch, nt, nullptr, nullptr);
}
// INTERPOL_FLAG_AUTO_SCATTER: 2 = launched behind the verdict of probe5 (smooth fields stay with the LDS tiles: 8 x 1 x 192^3 order 5 at the
// identity 2.22 against 2.37 ms; rough ones go to the bricks: sigma = 2 4.24 / 3.68) -- the caller launches the tiles as well, with
// KParams::gate = *gate_out (they return at once when the verdict is 1).
int IP_S5_TRY(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, void *workspace, int64_t workspace_bytes,
              hipStream_t st, const int **gate_out)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return try_scatter7(p, k, val, grid, vol, workspace, workspace_bytes, st, gate_out);
#endif
    using namespace IP_G5_NS;
    if (!workspace || ((uintptr_t)workspace & 255u) != 0) return 0;
    const int64_t need = IP_S5_WSB(p, k);
    if (need <= 0 || need > workspace_bytes) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = ((gx + TS - 1) / TS) * nty * ntz;
    const Grid5 bg = brick_grid(k);
    Workspace w;
    if (layout(bg, (int)p->batch, ntiles, workspace, &w) > workspace_bytes) return 0;
    const int64_t nz = 64 + 2 * w.nbricks + 1;
    if (nz > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(zero5, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, w.hdr, (int)nz);
#ifdef IP_G5_HIGH
    const bool gated = false;                                        // orders 6 - 7: the bricks always
#else
    const bool gated = !(p->flags & INTERPOL_FLAG_BINNED_SCATTER);
#endif
    const int *gate = gated ? w.hdr : nullptr;
#ifndef IP_G5_HIGH
    if (gated) {
        const long long total = (long long)ntiles * p->batch;
        const dim3 pgrid((unsigned)(total < NPROBE ? total : NPROBE));
#define IP_P5(KK, GM) hipLaunchKernelGGL((probe5<KK, GM>), pgrid, dim3(NT1), 0, st, k, (const float *)grid, w.hdr, gx, gy, gz, nty, ntz, ntiles, (int)p->batch);
#define IP_P5_GM(KK) { if (k.sep == 0) IP_P5(KK, 0) else if (k.sep == 1) IP_P5(KK, 1) else if (k.sep == 2) IP_P5(KK, 2) else IP_P5(KK, 3) }
        if (k.order[0] == 5) IP_P5_GM(5) else IP_P5_GM(4)
#undef IP_P5_GM
#undef IP_P5
    }
#endif
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    const long long want = 2ll * cu_count();
    const dim3 ggrid((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_S5(KK, GM)                                                                                                   \
    {                                                                                                                   \
        hipLaunchKernelGGL((bin5<KK, GM, 3>), tgrid, dim3(NT1), 0, st, k, bg, (const float *)val, (const float *)grid, (float *)vol, \
                           w.ndesc, w.list, w.desc, w.rec, gx, gy, gz, nty, ntz, ntiles, gate, (const float *)nullptr, (float *)nullptr); \
        const int attr = big_lds<scatter5<KK>>(sizeof(ScatSmem));                                                       \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((scatter5<KK>), ggrid, dim3(NT), sizeof(ScatSmem), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)val, (float *)vol, gate);   \
    }
#define IP_S5_GM(KK) { if (k.sep == 0) IP_S5(KK, 0) else if (k.sep == 1) IP_S5(KK, 1) else if (k.sep == 2) IP_S5(KK, 2) else IP_S5(KK, 3) }
    if (k.order[0] == IP_G5_KHI) IP_S5_GM(IP_G5_KHI) else IP_S5_GM(IP_G5_KLO)
#undef IP_S5_GM
#undef IP_S5
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (gate_out) *gate_out = gate;
    return gated ? 2 : 1;
}

// Both gradients of the pull's backward (pushpull.py:237-258) with ONE binning of the samples (round 6; 8 x 1 x 192^3 order 5: 5.09 -> 4.7 ms): the records
// of bin5<K, GM, 4> serve scatter5 (image gradient: push of grad_out into `acc`, the dense float (B, C, *vol) gradient) and gather5<K, 1> (grid gradient).
// k: the gather's parameters (vol_* the image, val_* grad_out), kp: the scatter's (vol_* = acc).  1 = both done; 2 = the grid gradient done, the image
// gradient launched behind the probe's verdict (the caller launches the tiles / generic push as well, with KParams::gate = *gate_out); 0 = declined.
int try_backward7(const interpol_problem *p, const KParams &k, const KParams &kp, const void *gout, const void *vol, const void *grid, void *acc, void *ggrid,
                  void *workspace, int64_t workspace_bytes, hipStream_t st, const int **gate_out);
int IP_B5_TRY(const interpol_problem *p, const KParams &k, const KParams &kp, const void *gout, const void *vol, const void *grid, void *acc, void *ggrid,
              void *workspace, int64_t workspace_bytes, hipStream_t st, const int **gate_out)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return try_backward7(p, k, kp, gout, vol, grid, acc, ggrid, workspace, workspace_bytes, st, gate_out);
#endif
    using namespace IP_G5_NS;
    if (!workspace || ((uintptr_t)workspace & 255u) != 0 || !eligible(p, k) || kp.cc) return 0;
    const int64_t need = IP_S5_WSB(p, kp);                            // (the scatter's own conditions: density, no shared target)
    if (need <= 0 || need > workspace_bytes) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = ((gx + TS - 1) / TS) * nty * ntz;
    const Grid5 bg = brick_grid(k);
    Workspace w;
    if (layout(bg, (int)p->batch, ntiles, workspace, &w) > workspace_bytes) return 0;
    const int64_t nz = 64 + 2 * w.nbricks + 1;
    if (nz > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(zero5, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, w.hdr, (int)nz);
#ifdef IP_G5_HIGH
    const bool gated = false;
#else
    const bool gated = !(p->flags & INTERPOL_FLAG_BINNED_SCATTER);   // the scatter alone: smooth fields keep the LDS tiles (try_scatter5)
#endif
    const int *gate = gated ? w.hdr : nullptr;
#ifndef IP_G5_HIGH
    if (gated) {
        const long long total = (long long)ntiles * p->batch;
        const dim3 pgrid((unsigned)(total < NPROBE ? total : NPROBE));
#define IP_P5(KK, GM) hipLaunchKernelGGL((probe5<KK, GM>), pgrid, dim3(NT1), 0, st, k, (const float *)grid, w.hdr, gx, gy, gz, nty, ntz, ntiles, (int)p->batch);
#define IP_P5_GM(KK) { if (k.sep == 0) IP_P5(KK, 0) else if (k.sep == 1) IP_P5(KK, 1) else if (k.sep == 2) IP_P5(KK, 2) else IP_P5(KK, 3) }
        if (k.order[0] == 5) IP_P5_GM(5) else IP_P5_GM(4)
#undef IP_P5_GM
#undef IP_P5
    }
#endif
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    const long long want = 2ll * cu_count();
    const dim3 ggrid_((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_B5(KK, GM)                                                                                                   \
    {                                                                                                                   \
        hipLaunchKernelGGL((bin5<KK, GM, 4>), tgrid, dim3(NT1), 0, st, k, bg, (const float *)vol, (const float *)grid, (float *)ggrid, \
                           w.ndesc, w.list, w.desc, w.rec, gx, gy, gz, nty, ntz, ntiles, gate, (const float *)gout, (float *)acc); \
        int attr = big_lds<scatter5<KK>>(sizeof(ScatSmem));                                                             \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((scatter5<KK>), ggrid_, dim3(NT), sizeof(ScatSmem), st, kp, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)gout, (float *)acc, gate); \
        attr = big_lds<gather5<KK, 1>>(offsetof(GatSmem, qcnt));                                                        \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((gather5<KK, 1>), ggrid_, dim3(NT), offsetof(GatSmem, qcnt), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.list, w.hdr + 41, (const float *)vol, (float *)ggrid, (const int *)nullptr, (const float *)gout); \
    }
#define IP_B5_GM(KK) { if (k.sep == 0) IP_B5(KK, 0) else if (k.sep == 1) IP_B5(KK, 1) else if (k.sep == 2) IP_B5(KK, 2) else IP_B5(KK, 3) }
    if (k.order[0] == IP_G5_KHI) IP_B5_GM(IP_G5_KHI) else IP_B5_GM(IP_G5_KLO)
#undef IP_B5_GM
#undef IP_B5
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (gate_out) *gate_out = gate;
    return gated ? 2 : 1;
}

// Both gradients of the push's backward (pushpull.py:262-282) with ONE binning (round 6): gval = pull of grad_vol_out, ggrid = the grid gradient of that
// pull contracted with the values.  k: vol_* grad_vol_out, val_* the values / their gradient (same strides).  1 = both done, 0 = declined.
int try_pushbwd7(const interpol_problem *p, const KParams &k, const void *gvol_out, const void *val, const void *grid, void *gval, void *ggrid,
                 void *workspace, int64_t workspace_bytes, hipStream_t st);
int IP_PB5_TRY(const interpol_problem *p, const KParams &k, const void *gvol_out, const void *val, const void *grid, void *gval, void *ggrid,
               void *workspace, int64_t workspace_bytes, hipStream_t st)
{
#ifndef IP_G5_HIGH
    if (k.order[0] >= 6) return try_pushbwd7(p, k, gvol_out, val, grid, gval, ggrid, workspace, workspace_bytes, st);
#endif
    using namespace IP_G5_NS;
    if (!workspace || ((uintptr_t)workspace & 255u) != 0 || !eligible(p, k) || !val || !gval || !ggrid) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = ((gx + TS - 1) / TS) * nty * ntz;
    const Grid5 bg = brick_grid(k);
    Workspace w;
    if (layout(bg, (int)p->batch, ntiles, workspace, &w) > workspace_bytes) return 0;
    const int64_t nz = 64 + 2 * w.nbricks + 1;
    if (nz > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(zero5, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, w.hdr, (int)nz);
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    const long long want = 2ll * cu_count();
    const dim3 ggrid_((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_PB5(KK, GM)                                                                                                  \
    {                                                                                                                   \
        hipLaunchKernelGGL((bin5<KK, GM, 5>), tgrid, dim3(NT1), 0, st, k, bg, (const float *)gvol_out, (const float *)grid, (float *)gval, \
                           w.ndesc, w.list, w.desc, w.rec, gx, gy, gz, nty, ntz, ntiles, (const int *)nullptr, (const float *)val, (float *)ggrid); \
        if (k.C * (KK + 1) * (KK + 1) * (KK + 1) >= 400) {                                                              \
            const int attr = big_lds<gather5<KK, 0, true>>(sizeof(GatSmem));                                            \
            if (attr) return attr;                                                                                      \
            hipLaunchKernelGGL((gather5<KK, 0, true>), ggrid_, dim3(NT), sizeof(GatSmem), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                               (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)gvol_out, (float *)gval, (const int *)nullptr, (const float *)nullptr); \
        } else {                                                                                                        \
            const int attr = big_lds<gather5<KK, 0>>(offsetof(GatSmem, qcnt));                                          \
            if (attr) return attr;                                                                                      \
            hipLaunchKernelGGL((gather5<KK, 0>), ggrid_, dim3(NT), offsetof(GatSmem, qcnt), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                               (const float4 *)w.rec, (const int *)w.list, w.hdr + 40, (const float *)gvol_out, (float *)gval, (const int *)nullptr, (const float *)nullptr); \
        }                                                                                                               \
        const int attr1 = big_lds<gather5<KK, 1>>(offsetof(GatSmem, qcnt));                                             \
        if (attr1) return attr1;                                                                                        \
        hipLaunchKernelGGL((gather5<KK, 1>), ggrid_, dim3(NT), offsetof(GatSmem, qcnt), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.list, w.hdr + 41, (const float *)gvol_out, (float *)ggrid, (const int *)nullptr, (const float *)val); \
    }
#define IP_PB5_GM(KK) { if (k.sep == 0) IP_PB5(KK, 0) else if (k.sep == 2) IP_PB5(KK, 2) else return 0; }
    if (k.order[0] == IP_G5_KHI) IP_PB5_GM(IP_G5_KHI) else IP_PB5_GM(IP_G5_KLO)
#undef IP_PB5_GM
#undef IP_PB5
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? (int)e : 1;
}

} // namespace ip
