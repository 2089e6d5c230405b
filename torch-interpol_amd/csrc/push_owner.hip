// ===========================================================================
// push_owner.hip -- grid_push / grid_count, OWNER-COMPUTES: every brick of the target lattice is
// accumulated by exactly one workgroup and written with plain loads / stores -- no global atomics on
// the main path.  3-D, one spline order 2..3, f32 / bf16 / f16 sources, every boundary / extrapolation
// mode, every coordinate source (dense grid, separable lattice, displacement field, affine lattice).
// Reference semantics: interpol/nd.py:146-213 (push), pushpull.py:106-142 (count).
//
// Why.  Measured on this chip (tools/microbench/global_atomics.hip, profiles/r03_micro_global_atomics.txt):
// the memory side retires 0.31 G float lane-atomics per ms however they are coalesced, scoped (the
// compiler emits the SAME instruction for wavefront / workgroup / agent scope) or packed (u64 pairs:
// 0.18 G/ms, +15 % payload), while plain coalesced read-modify-write runs at 0.5-0.85 G floats per ms and
// plain stores at 1.3-1.5 G.  A sample-stationary tile under sigma = 2 voxel noise flushes 9 lattice
// points per sample: 1.6 of the 3.5 ms of push_tiled / push_sorted are those atomics.  Here:
//
//   own_bin        : one workgroup per tile of 16^3 SAMPLE points: coordinates -> brick of the first tap
//                    (16^3 first-tap cells), counting sort of the tile's samples by brick in LDS, the
//                    sorted records (coordinates + masked source values as floats) leave as coalesced
//                    16-byte stores into the tile's own segment of the workspace, each with the cell of its
//                    first tap inside the brick (2 bytes); one descriptor (first record, count) per non-empty
//                    (tile, brick) pair is appended to the brick's descriptor list (one returning atomic per
//                    pair: ~30 per tile) and the tile's max |source| is folded into the brick's.  No count
//                    pass, no scan: ONE pass over the inputs.
//   own_accumulate : one workgroup per TARGET brick walks the runs its descriptors name: density of the
//                    first-tap cells (with the brick's max |source| -> fixed-point scale, tile_common.hpp
//                    headroom32), then every record adds its (K+1)^3 taps into the brick's LDS box (16 + K
//                    lattice points per dim) with packed 32-bit fixed-point ds_add_u64 (two channels per
//                    atomic), and the box is ADDED to the target with plain loads and stores.  Boxes of bricks
//                    two apart are disjoint, so the bricks are launched in 8 colours (parity of the brick
//                    coordinates): within a launch no two workgroups touch the same lattice point, and
//                    the launches are ordered by the stream.  Stencils that leave the lattice: under
//                    replicate / dct1 / dct2 the bricks at the ends of the dim hold them (up to 9 points out)
//                    and fold that part of the box back inside it in LDS before the flush (BrickGrid below);
//                    what lies further out, and the other boundary conditions, go last, in a launch of shell
//                    bricks that flush through the boundary tables with global atomics.  A target shared by
//                    the batch items (batch stride 0) is flushed with atomics throughout.
// Samples whose first tap lies outside [-160, n + 160), tiles spread over more than 6 bricks per dim and
// runs beyond a brick's 128 descriptors are scattered directly with float atomics by own_bin (always
// correct; the target is zeroed before own_bin and the brick launches come after it on the stream).
//
// Sums inside a brick are integer (exact, order-free); the float additions of up to 8 boxes per lattice
// point happen in the fixed order of the colours: where no atomics are involved (folding dims, samples
// within 9 points of the lattice) the result is bit-reproducible.
//
// Workspace (caller's, interpol_scatter_workspace()): 18 B per sample + 4 B per sample and further
// channel, 12 + 1024 B per brick.
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace owner {

using namespace sorted;          // helpers of sorted_util.hpp

constexpr int BR = 16;                          // brick edge, in first-tap cells
constexpr int OFF = 160;                        // first taps in [-OFF, n + OFF) are binned (a zoom by 2 about the centre of a 256^3 lattice reaches 128 voxels
                                                // beyond it); beyond: scattered directly
constexpr int BOX = BR + 3;                     // lattice points a brick's stencils touch per dim (K <= 3)
constexpr int PZ = BOX;                         // row pitch of the LDS box (8-byte slots): rows back to back
constexpr int PLANE = BOX * PZ;                 // 361
constexpr int BOXSLOTS = BOX * PLANE;           // 6859 slots = 54 872 B
constexpr int NCELL = BR * BR * BR;
constexpr int CAPD = 128;                       // descriptors (runs) per brick
constexpr int CAPX = 512;                       // ... of a target shared by the batch items (round 5: up to eight tiles of every item reach a brick)
constexpr int NT = 512;                         // accumulate: threads per workgroup (two workgroups per CU)
constexpr int NS = TS * TS * TS;                // samples per tile (own_bin)
constexpr int NT1 = 512, VPT1 = NS / NT1;       // own_bin: threads, samples per thread
constexpr int LB = 6, NBIN = LB * LB * LB;      // bricks around a tile that are sorted locally
constexpr int HDR_SHELL = 34;                   // word of the workspace header own_bin sets when a shell brick received a run
// Mixed per-dim orders 1..3 (round 6): the kernels instantiated with K = KMIX run the CUBIC organisation -- bricks, 19^3 boxes, all 64 taps
// (the adds of taps beyond a dim's order carry weight 0: exactly MAGIC, so the stencil counts stay the cubic's), colours, flush -- with the
// first tap floor(x - (k_d - 1)/2) and the weights of each dim's own order (KParams::order).  KSV<K>: the order the organisation sees.
constexpr int KMIX = 13;
template <int K> constexpr int KSV = K == KMIX ? 3 : K;
template <int K> __device__ __forceinline__ int kd(const KParams &p, int d) { return K == KMIX ? p.order[d] : K; }
// own_accumulate and its helpers are written through these: in the isotropic module they expand to the expressions of rounds 3 - 5 token
// for token (with the template forms above, equivalent as they are, the compiler allocated own_accumulate<3>'s registers differently)
#ifdef IP_OWNER_MIX_TU
#define IP_KS KSV<K>
#define IP_KD(d) kd<K>(p, d)
#define IP_WX(i) own_weight_x<K>(p, tx, i)
#define IP_MU magic_units_p<K>(p)
#else
#define IP_KS K
#define IP_KD(d) K
#define IP_WX(i) weight_x<K>(tx, i)
#define IP_MU magic_units<K>()
#endif

// Bricks of first-tap cells, per dim.  INTERIOR cells [lo, top) lie in nin bricks of BR cells: the bricks up to index
// `split` are aligned to lo, the ones above it to top (brick `split` is the short one in between).  Two cases:
//   * the boundary condition of the dim is a single reflection or a clamp without a change of sign (replicate, dct1, dct2) and
//     the lattice is not tiny: lo = -9, top = n + 6 -- the boxes of the bricks at the two ends leave the lattice by at most 9
//     points: their contents are FOLDED back inside the box in LDS before the flush, and that far out the target of every fold
//     still lies in the same box (low end: box [-9, 10), point -9 -> 8; high end: box [n - 10, n + 9), point n + 8 -> n - 9);
//   * else: lo = 0, top = n - K -- the stencils that lie inside the lattice (split = nin - 1: the last brick holds the remainder).
// [lo - OFF, lo) lies in NLO bricks, [top, top + NHI * BR) in NHI bricks: the SHELL (stencils far outside the lattice, wrapping and
// sign-changing boundary conditions), whose bricks flush through the boundary tables with global atomics -- their boxes alias.
constexpr int NLO = OFF / BR, NHI = (OFF + 3 + BR - 1) / BR;
struct BrickGrid {
    int nb[3];                                  // bricks per dim
    int lo[3], top[3], nin[3], split[3];        // interior first-tap cells [lo, top), bricks that hold them, the short brick
    int per_item;                               // nb[0] * nb[1] * nb[2]
    int item;                                   // brick-index stride of a batch item: per_item, or 0 when the items share ONE target (their bricks are the same)
    int capd;                                   // descriptors (runs) per brick: CAPD, CAPX for a shared target (every item's tiles feed the brick)
};
__host__ __device__ __forceinline__ bool folds(int bound, int n) { return (bound == B_REPLICATE || bound == B_DCT1 || bound == B_DCT2) && n >= 2 * BR; }
static BrickGrid brick_grid(const KParams &k)
{
    BrickGrid g;
    const bool mixed = k.order[0] != k.order[1] || k.order[0] != k.order[2];   // mixed orders (K = KMIX below): the cubic's bricks
    for (int d = 0; d < 3; ++d) {
        const int K = mixed ? 3 : k.order[d], n = k.vol_n[d];
        const bool f = folds(k.bound[d], n);
        // folding dims: as far out as the mirror image of every point of the end bricks' boxes stays inside the box
        g.lo[d] = f ? -(BOX - 1) / 2 : 0;
        g.top[d] = f ? n - (BOX + 1) / 2 + BR : (n - K > 0 ? n - K : 0);
        g.nin[d] = (g.top[d] - g.lo[d] + BR - 1) / BR;
        if (f) {
            // no short brick of less than K cells: the box of the brick below it (K points beyond its own cells) must not
            // reach the brick above it, which has the same colour
            const int c = g.top[d] - g.lo[d] - BR * (g.nin[d] - 1);
            if (c < K) { g.top[d] -= c; g.nin[d] -= 1; }
        }
        g.split[d] = f ? g.nin[d] / 2 : (g.nin[d] > 0 ? g.nin[d] - 1 : 0);
        g.nb[d] = NLO + g.nin[d] + NHI;
    }
    g.per_item = g.nb[0] * g.nb[1] * g.nb[2];
    g.item = g.per_item; g.capd = 128;
    return g;
}
// brick of a first-tap cell (inside [lo - OFF, top + NHI * BR)) and the cell's index inside its brick (<< 16)
__host__ __device__ __forceinline__ int brick_and_cell(int ft, int lo, int top, int nin, int split)
{
    if (ft < lo) return ((ft - lo + OFF) >> 4) | (((ft - lo + OFF) & (BR - 1)) << 16);
    if (ft >= top) return (NLO + nin + ((ft - top) >> 4)) | (((ft - top) & (BR - 1)) << 16);
    const int u = ft - lo, v = top - 1 - ft;
    const bool hi = u >= BR * split && (v >> 4) < nin - 1 - split;
    return hi ? (NLO + nin - 1 - (v >> 4)) | ((BR - 1 - (v & (BR - 1))) << 16) : (NLO + (u >> 4)) | ((u & (BR - 1)) << 16);
}
// first cell of a brick
__host__ __device__ __forceinline__ int brick_origin(int bk, int lo, int top, int nin, int split)
{
    if (bk < NLO) return lo + bk * BR - OFF;
    if (bk >= NLO + nin) return top + (bk - NLO - nin) * BR;
    const int j = bk - NLO;
    return j <= split ? lo + j * BR : top - (nin - j) * BR;
}

// ---------------------------------------------------------------------------
// own_bin
// ---------------------------------------------------------------------------
struct BinSmem {
    int lo[3], total;
    int cnt[NBIN];             // samples of the tile per local brick; after the scan: -1 marks an orphan run
    int base[NBIN];            // first sorted position of the local brick
    int bmx[2];                // max |masked source| over the tile's binned samples, first two channels (float bits)
    int orph, pad_;            // the tile has orphan runs (a brick's descriptor list was full)
    int gbk[NBIN];             // global brick of the local brick when its run was published, else -1
};

// value of target channel ch for a sample: masked source (nd.py:201-203), or the mask itself for the count channel
template <typename T>
__device__ __forceinline__ float src_value(const KParams &p, const T *__restrict__ val, int64_t b, int64_t o, int ch, float m)
{
    if (val == nullptr || ch >= p.C) return m;
    return m * Cvt<float, T>::ld(val[b * p.val_sb + ch * p.val_sc + o]);
}

// Direct scatter of a thread's unbinned samples (bit v of `mask`: sample tid + NT1 v of the tile): float atomics, one tap at a time.
template <typename T, int K, int GM>
__device__ __forceinline__ void scatter_direct(const KParams &p, const T *__restrict__ val, const float *__restrict__ grid, float *__restrict__ vol,
                                            int64_t b, TileGeom g, int tid, unsigned mask, int nch)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = kd<K>(p, d); }
    L.lin = 0;
#pragma unroll 1
    for (int v = 0; v < VPT1; ++v) {
        if (!((mask >> v) & 1)) continue;
        int ox, oy, oz; float x[3];
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
        const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
        const float m = inb_mask(p, x);
        if (KSV<K> == 1 && p.mode == MODE_ISO0) {
            // nearest neighbour: ONE lattice point, weight 1 (iso0.py:12, 65-118) -- see own_accumulate
            const long long pk0 = wrap_outofline(L.bound[0], __float2int_rn(rintf(x[0])), L.n[0]);
            const long long pk1 = wrap_outofline(L.bound[1], __float2int_rn(rintf(x[1])), L.n[1]);
            const long long pk2 = wrap_outofline(L.bound[2], __float2int_rn(rintf(x[2])), L.n[2]);
            const float sgn = (float)((int)(pk0 >> 32) * (int)(pk1 >> 32) * (int)(pk2 >> 32));
            float *q = vol + b * p.vol_sb + (int)(pk0 & 0xffffffffll) * L.ss[0] + (int)(pk1 & 0xffffffffll) * L.ss[1] + (int)(pk2 & 0xffffffffll) * L.ss[2];
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch)
                __hip_atomic_fetch_add(q + (int64_t)ch * p.vol_sc, src_value<T>(p, val, b, o, ch, m) * sgn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        int ii[3]; float tt[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) split(kd<K>(p, d), x[d], ii[d], tt[d]);
#pragma unroll 1
        for (int ch = 0; ch < nch; ++ch)
            tiled::scatter_one_thread(L, vol + b * p.vol_sb + ch * p.vol_sc, src_value<T>(p, val, b, o, ch, m), ii[0], ii[1], ii[2], tt[0], tt[1], tt[2]);
    }
}

// Direct gather of a thread's unbinned samples (own_bin in index mode, the owner-computes PULL): one thread per sample, taps from
// global memory.  `img`: the image, `out`: the output (p: the gather's parameters -- vol_* the image, val_* the output).
template <typename T, int K, int GM>
__device__ __forceinline__ void gather_direct(const KParams &p, const T *__restrict__ img, const float *__restrict__ grid, T *__restrict__ out,
                                              int64_t b, TileGeom g, int tid, unsigned mask)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = kd<K>(p, d); }
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
        for (int d = 0; d < 3; ++d) split(kd<K>(p, d), x[d], ii[d], tt[d]);
#pragma unroll 1
        for (int ch = 0; ch < p.C; ++ch)
            out[b * p.val_sb + ch * p.val_sc + o] = Cvt<float, T>::st(m * tiled::gather_one_thread<T>(L, img + b * p.vol_sb + ch * p.vol_sc, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1));
    }
}

// The same for the grid gradient of the pull (pushpull.py:256-257): ggrid[b,o,:] = mask * sum_c gout[b,c,o] * grad pull(img[b,c])(x_o);
// p: val_* describe grad_out, the grid gradient is dense (B, *out, 3).
template <typename T, int K, int GM>
__device__ __forceinline__ void gradc_direct(const KParams &p, const T *__restrict__ img, const T *__restrict__ gout, const float *__restrict__ grid,
                                             float *__restrict__ ggrid, int64_t b, TileGeom g, int tid, unsigned mask)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = kd<K>(p, d); }
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
        for (int d = 0; d < 3; ++d) split(kd<K>(p, d), x[d], ii[d], tt[d]);
        float a[3] = { 0.f, 0.f, 0.f };
#pragma unroll 1
        for (int ch = 0; ch < p.C; ++ch) {
            const float gv = gout ? Cvt<float, T>::ld(gout[b * p.val_sb + ch * p.val_sc + o]) : 1.f;
#pragma unroll 1
            for (int d = 0; d < 3; ++d)
                a[d] = __builtin_fmaf(gv, tiled::gather_one_thread<T>(L, img + b * p.vol_sb + ch * p.vol_sc, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], d), a[d]);
        }
        float *dst = ggrid + (b * p.N + o) * 3;
        dst[0] = a[0] * m; dst[1] = a[1] * m; dst[2] = a[2] * m;
    }
}

// ... and for grid_grad (nd.py:216-288): out[b,c,o,:] = mask * the three derivative sums (own_bin index mode 3)
template <typename T, int K, int GM>
__device__ __forceinline__ void grad_direct(const KParams &p, const T *__restrict__ img, const float *__restrict__ grid, T *__restrict__ out,
                                            int64_t b, TileGeom g, int tid, unsigned mask)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = kd<K>(p, d); }
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
        for (int d = 0; d < 3; ++d) split(kd<K>(p, d), x[d], ii[d], tt[d]);
#pragma unroll 1
        for (int ch = 0; ch < p.C; ++ch)
#pragma unroll 1
            for (int d = 0; d < 3; ++d)
                out[b * p.val_sb + ch * p.val_sc + 3 * o + d] =
                    Cvt<float, T>::st(m * tiled::gather_one_thread<T>(L, img + b * p.vol_sb + ch * p.vol_sc, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], d));
    }
}

// IDX (the owner-computes pull, own_gather below): the records carry the sample's linear index instead of a source value; `val` is
// then the IMAGE and `vol` the OUTPUT of the gather (for the samples gathered directly), `bmax` the list of non-empty bricks
// (entry 0: their number).  IDX == 2: the grid gradient of the pull (own_gather<K, true>): `aux` is grad_out, `vol` the grid gradient.
template <typename T, int K, int GM, int IDX = 0>
__global__ __launch_bounds__(NT1, 4) void own_bin(KParams p, BrickGrid bg, const T *__restrict__ val, const float *__restrict__ grid,
                                               float *__restrict__ vol, int *__restrict__ ndesc, uint2 *__restrict__ desc,
                                               float4 *__restrict__ rec, float *__restrict__ vals, unsigned short *__restrict__ meta,
                                               int *__restrict__ bmax, int64_t nrec,
                                               int gx, int gy, int gz, int nty, int ntz, int ntiles, const int *__restrict__ gate,
                                               const T *__restrict__ aux, const int *__restrict__ all)
{
    // AUTO: the probe chose the tiles / (pull, grid gradient) the sample tiles served this tile -- unless the probe gave every tile to the bricks (*all == 1)
    // (index mode without tile flags: `all` alone decides -- every tile or none)
    if (IDX) { const bool every = all && *all == 1; if (gate ? (gate[blockIdx.x] == 0 && !every) : (all && !every)) return; }
    else if (gate && *gate != 1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    BinSmem &sm = *reinterpret_cast<BinSmem *>(smem_raw);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / ntiles;
    const TileGeom g = tile_geom(blockIdx.x % ntiles, gx, gy, gz, nty, ntz);
    const int nch = (IDX || val == nullptr) ? 1 : p.C + p.cc;
    for (int i = tid; i < NBIN; i += NT1) { sm.cnt[i] = 0; sm.gbk[i] = -1; }
    if (tid < 2) sm.bmx[tid] = 0;
    if (tid == 2) sm.orph = 0;
    if (tid < 3) sm.lo[tid] = 0x7fffffff;
    float c[VPT1][3], v0[VPT1], v1[VPT1];
    unsigned valid = 0;
    prof_mark(-1);
    // sources of the first two channels: loaded unconditionally (a branch per load would serialise the round trips), from the
    // coordinates themselves where there is no such channel (count: the mask is the source)
    const bool has0 = !IDX && val != nullptr && p.C > 0, has1 = !IDX && val != nullptr && p.C > 1;
    const T *vp0 = has0 ? val + b * p.val_sb : reinterpret_cast<const T *>(grid);
    const T *vp1 = has1 ? val + b * p.val_sb + p.val_sc : reinterpret_cast<const T *>(grid);
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        int ox, oy, oz;
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        if (ox < gx && oy < gy && oz < gz) valid |= 1u << v;
        ox = ox < gx ? ox : gx - 1; oy = oy < gy ? oy : gy - 1; oz = oz < gz ? oz : gz - 1;
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, c[v]);
        const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
        if (IDX) { v0[v] = __int_as_float((int)o); v1[v] = 0.f; continue; }
        v0[v] = Cvt<float, T>::ld(vp0[has0 ? o : 0]);
        v1[v] = Cvt<float, T>::ld(vp1[has1 ? o : 0]);
    }
    unsigned inbits = 0;
#pragma unroll
    for (int v = 0; v < VPT1 && !IDX; ++v) {                          // masked sources (nd.py:201-203); count: the mask itself
        const float m = inb_mask(p, c[v]);
        v0[v] = has0 ? v0[v] * m : m; v1[v] = has1 ? v1[v] * m : m;
        if (KSV<K> == 1 && m != 0.f) inbits |= 1u << v;                   // (kept for the further channels: the nearest-neighbour mode rounds c below)
    }
    // nearest-neighbour scatters (all orders 0; the host passes them as trilinear ones, KParams::mode still MODE_ISO0): the coordinates
    // rounded half to even (iso0.py:12) AFTER the mask saw the real ones -- a 2 x 2 x 2 stencil at t = 0: weight 1 on the first tap, +0 elsewhere
    if (KSV<K> == 1 && !IDX && p.mode == MODE_ISO0) {
#pragma unroll
        for (int v = 0; v < VPT1; ++v) { c[v][0] = rintf(c[v][0]); c[v][1] = rintf(c[v][1]); c[v][2] = rintf(c[v][2]); }
    }
    // ---- brick of the first tap (nd.py:45: i0 = floor(x - (K-1)/2)), block minimum of the brick coordinates
    int bx[VPT1][3];
    unsigned ok = 0;
    int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff };
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        bool in = (valid >> v) & 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float fl = floorf(c[v][d] - 0.5f * (float)(kd<K>(p, d) - 1));
            in = in && fl >= (float)(bg.lo[d] - OFF) && fl < (float)(bg.top[d] + NHI * BR);        // (false for NaN)
            // brick, and above it the cell inside the brick
            bx[v][d] = in ? brick_and_cell(__float2int_rz(fl), bg.lo[d], bg.top[d], bg.nin[d], bg.split[d]) : 0;
        }
        if (in) {
            ok |= 1u << v;
#pragma unroll
            for (int d = 0; d < 3; ++d) mn[d] = (bx[v][d] & 0xffff) < mn[d] ? (bx[v][d] & 0xffff) : mn[d];
        }
    }
    __syncthreads();                                                 // counters zero
    prof_mark(0);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = wave_min(mn[d]);
        if ((tid & 63) == 0) atomicMin(&sm.lo[d], a);
    }
    __syncthreads();
    const int lo[3] = { sm.lo[0], sm.lo[1], sm.lo[2] };
    int lbin[VPT1];                     // local brick (8 bits), cell inside the brick (12), rank inside the local brick (12)
    unsigned local = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        const int r0 = (bx[v][0] & 0xffff) - lo[0], r1 = (bx[v][1] & 0xffff) - lo[1], r2 = (bx[v][2] & 0xffff) - lo[2];
        const bool l = ((ok >> v) & 1) && (unsigned)r0 < (unsigned)LB && (unsigned)r1 < (unsigned)LB && (unsigned)r2 < (unsigned)LB;
        // local brick, and above it the first-tap cell inside the brick: x0 << 8 | y0 << 4 | z0
        lbin[v] = l ? ((r0 * LB + r1) * LB + r2) | ((bx[v][0] >> 16) << 16 | (bx[v][1] >> 16) << 12 | (bx[v][2] >> 16) << 8) : 0;
        if (l) { local |= 1u << v; lbin[v] |= atomicAdd(&sm.cnt[lbin[v] & 255], 1) << 20; }
    }
    __syncthreads();
    prof_mark(1);
    // ---- exclusive scan of the local brick counts (one wave); one descriptor per non-empty local brick -- its slot in the
    // brick's list is DRAWN here (one returning atomic per run) and USED only after the tile's records have been stored: the
    // round trip to the L2 is off the tile's critical path (round 5; rounds 3-4 waited for the slots before the first store)
    const int64_t tilebase = (int64_t)blockIdx.x * NS;
    constexpr int PER = (NBIN + 63) / 64;                            // 4
    int slot[PER];
    if (tid < 64) {
        int cn[PER], s = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int e = tid * PER + i; cn[i] = e < NBIN ? sm.cnt[e] : 0; s += cn[i]; }
        int incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
        int run = incl - s;
        if (tid == 63) sm.total = incl;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid * PER + i;
            const int r0 = e / (LB * LB), r1 = (e / LB) % LB, r2 = e % LB;
            const int bk = (int)b * bg.item + ((lo[0] + r0) * bg.nb[1] + (lo[1] + r1)) * bg.nb[2] + (lo[2] + r2);
            slot[i] = 0;
            if (e < NBIN) sm.base[e] = run;
            if (e < NBIN && cn[i] > 0) slot[i] = atomicAdd(&ndesc[bk], 1);
            run += cn[i];
        }
    }
    __syncthreads();
    prof_mark(2);
    // ---- sorted position of every binned sample (above it: the first-tap cell inside the brick)
    int pos[VPT1];
#pragma unroll
    for (int v = 0; v < VPT1; ++v)
        pos[v] = ((local >> v) & 1) ? (sm.base[lbin[v] & 255] + (int)((unsigned)lbin[v] >> 20)) | (((lbin[v] >> 8) & 0xfff) << 16) : -1;
    // ---- the sorted records are stored straight to their places (16 + 2 + 4 bytes per lane: the lines are completed by this
    // workgroup within microseconds and merge in the L2).  Index mode: the record (x, y, z, sample index) is all the gather needs
    const bool two = nch > 1;
    int amx0 = 0, amx1 = 0;             // max |source| of what this thread stores (non-negative floats, and NaN, order like ints)
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (pos[v] < 0) continue;
        const int64_t at = tilebase + (pos[v] & 0xffff);
        rec[at] = make_float4(c[v][0], c[v][1], c[v][2], v0[v]);
        if (IDX) continue;
        meta[at] = (unsigned short)(pos[v] >> 16);
        const int a0 = __float_as_int(__builtin_fabsf(v0[v]));
        amx0 = a0 > amx0 ? a0 : amx0;
        if (two) {
            vals[at] = v1[v];
            const int a1 = __float_as_int(__builtin_fabsf(v1[v]));
            amx1 = a1 > amx1 ? a1 : amx1;
        }
    }
    for (int ch = 2; ch < nch && !IDX; ++ch) {                       // further channels
#pragma unroll
        for (int v = 0; v < VPT1; ++v) {
            if (pos[v] < 0) continue;
            int ox, oy, oz;
            sample_pos(g, tid + NT1 * v, ox, oy, oz);
            vals[(int64_t)(ch - 1) * nrec + tilebase + (pos[v] & 0xffff)] = src_value<T>(p, val, b, ((int64_t)ox * gy + oy) * gz + oz, ch, KSV<K> == 1 ? (float)((inbits >> v) & 1u) : inb_mask(p, c[v]));
        }
    }
    // ---- the descriptors, now that the slots have arrived.  A run whose brick's list was full is an ORPHAN: its records stay
    // where they are (nobody reads them) and its samples are scattered / gathered directly, below
    if (tid < 64) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid * PER + i;
            const int cn = e < NBIN ? sm.cnt[e] : 0;
            if (cn <= 0) continue;
            const int r0 = e / (LB * LB), r1 = (e / LB) % LB, r2 = e % LB;
            const int bk = (int)b * bg.item + ((lo[0] + r0) * bg.nb[1] + (lo[1] + r1)) * bg.nb[2] + (lo[2] + r2);
            if (slot[i] < bg.capd) {
                desc[(int64_t)bk * bg.capd + slot[i]] = make_uint2((unsigned)(tilebase + sm.base[e]), (unsigned)cn);
                sm.gbk[e] = bk;
                // a run in a brick of the SHELL (outside the interior bricks of some dim): the shell launch of own_accumulate has work (round 6:
                // else it returns at once -- 36 us of candidate scanning at config 2 for nothing)
                const int bc[3] = { lo[0] + r0, lo[1] + r1, lo[2] + r2 };
                bool shell = false;
#pragma unroll
                for (int d = 0; d < 3; ++d) shell = shell || bc[d] < NLO + (p.bound[d] == B_DST1 ? 1 : 0) || bc[d] >= NLO + bg.nin[d];
                if (!IDX && shell) ndesc[HDR_SHELL - 64] = 1;        // (the header lies 64 words in front of the brick counters, layout())
                if (IDX && slot[i] == 0) bmax[1 + atomicAdd(&bmax[0], 1)] = bk;      // first run of the brick: onto the list
            } else { sm.cnt[e] = -1; sm.orph = 1; }
        }
    }
    // max |source| of the first channel pair, per brick: the tile's maximum goes to every brick it published a run for (the
    // fixed-point scale of own_accumulate is as local as the tiled scatter's)
    if (!IDX) {
        amx0 = wave_max(amx0); amx1 = wave_max(amx1);
        if ((tid & 63) == 0) { if (amx0) atomicMax(&sm.bmx[0], amx0); if (amx1) atomicMax(&sm.bmx[1], amx1); }
    }
    __syncthreads();
    for (int e = tid; e < NBIN && !IDX; e += NT1) {                  // (index mode: `bmax` is the brick list)
        const int bk = sm.gbk[e];
        if (bk < 0) continue;
        if (sm.bmx[0]) atomicMax(&bmax[2 * (int64_t)bk], sm.bmx[0]);
        if (sm.bmx[1]) atomicMax(&bmax[2 * (int64_t)bk + 1], sm.bmx[1]);
    }
    // ---- samples that are not binned (first tap far outside the lattice, tile spread over more than LB bricks, orphan runs)
    // are handled directly at the END of the kernel, from re-read coordinates: the out-of-line scatter would otherwise force
    // every live register of the hot path through scratch around its call
    unsigned direct = valid & ~local;
    if (sm.orph) {
#pragma unroll
        for (int v = 0; v < VPT1; ++v)
            if (((local >> v) & 1) && sm.cnt[lbin[v] & 255] < 0) direct |= 1u << v;
    }
    if (IDX == 3) { if (direct) grad_direct<T, K, GM>(p, val, grid, reinterpret_cast<T *>(vol), b, g, tid, direct); return; }
    if (IDX == 2) { if (direct) gradc_direct<T, K, GM>(p, val, aux, grid, vol, b, g, tid, direct); return; }
    if (IDX) { if (direct) gather_direct<T, K, GM>(p, val, grid, reinterpret_cast<T *>(vol), b, g, tid, direct); return; }
    if (direct) scatter_direct<T, K, GM>(p, val, grid, vol, b, g, tid, direct, nch);
}

// ---------------------------------------------------------------------------
// own_accumulate
// ---------------------------------------------------------------------------
constexpr int VPT = 12;                         // pieces per wave and batch
constexpr int NPIECE = VPT * (NT / 64);         // pieces (<= 64 consecutive records of one run) per batch: 96
constexpr int BATCH = NPIECE * 64;              // records per class-sorted batch, at most (a brick holds 4096 on average)
constexpr int NCLS = 16;                        // classes = (8-byte slot) mod 16: ds_add_u64 is served in groups of 16 contiguous lanes (round 6, tools/microbench/lds_add_classes.hip)
constexpr int NHW = NT / NCLS;                  // groups of 16 lanes per workgroup
constexpr unsigned PMASK = 0x1ffffffu;          // AccSmem::ppref: the prefix field
static_assert(NPIECE <= 128, "a queue entry is piece << 6 | lane in 16 bits");

struct AccSmem {
    int   taboff[3][BOX + 1];
    float tabsgn[3][BOX + 1];
    unsigned ppref[CAPX];                      // per run: exclusive prefix of the runs' piece counts (bits 0-24) and the records of its last piece
                                               // (bits 25-31: 1 .. 64); entries beyond the last run: all ones
    unsigned start[CAPX];                      // first record of each run
    uint2 piece[NPIECE];                       // pieces of the current batch: first record, records (0: none)
    int   cmax[2];
    int   dmax, n, npieces;
    int   rlo[3], rhi[3];                      // launches that flush with atomics: occupied first-tap cells of the brick, per dim
    int   qcnt[NCLS];                          // records per class of the batch
    int   qsur[NCLS + 1], qhol[NCLS + 1];      // exclusive prefixes of the classes' surplus records / free queue slots
    int   qeff[NCLS], qcap;                    // occupied slots per class once the holes are filled; slots per class
    union alignas(16) {
        struct {
            unsigned cells[NCELL / 2];         // density: 16-bit counters per first-tap cell
            unsigned short queue[BATCH];       // records of the batch (piece << 6 | lane), sorted by class
        };
        // once the taps of a channel pair are done: the number of stencils that cover each slot of the box (16 bits,
        // slot (x, y, z) at (x * BOX + y) * NZ + z) -- the box filter of `cells`, built in place (stencil_counts)
        unsigned nreg[BOX * BOX * (BOX + 1) / 2];
    };
    unsigned long long box[BOXSLOTS];
};
constexpr int NZ = BOX + 1;                     // row pitch of the stencil counts (even: two slots per 32-bit word)
static_assert(sizeof(AccSmem) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(unsigned) * (BOX * BOX * NZ / 2) <= sizeof(unsigned) * (NCELL / 2) + sizeof(unsigned short) * BATCH, "the stencil counts fit in cells + queue");

// ---------------------------------------------------------------------------
// Accumulation format (round 5).  A tap's contribution t = source * scale * w_x w_y w_z is bounded by 2^22 in magnitude and
// leaves the packed FMA as the float  t + 1.5 * 2^23 : in [2^23, 2^24) the spacing of floats is 1, so the BIT PATTERN of the
// result is MAGIC_BITS + round(t) -- the integer the LDS box sums, already in place, for both channels of the pair (the 64-bit
// operand of ds_add_u64 IS the register pair of the packed FMA: no conversion, no packing, no borrow).  A slot that n stencils
// cover then holds  sum_i (MAGIC_BITS + q1_i) * 2^32 + sum_i (MAGIC_BITS + q0_i)  (mod 2^64); n is the box filter of the density of
// first-tap cells (which pass 1 builds anyway), so both sums are recovered exactly at the flush (magic_decode).  Rounds 3-4 spent
// 397 of the kernel's 711 VALU instructions per sample on v_cvt_rpi / shift / add per tap and channel; this is one v_pk_fma_f32
// per tap pair.  Resolution: one unit = max |source| (of the tiles that feed the brick) * wmax^3 * 2^-22 / 0.999, wmax = 2/3 (cubic),
// 3/4 (quadratic): 2^-23.75 resp. 2^-23.25 of max |source| per addend, rounded to nearest (fused: one rounding).
// ---------------------------------------------------------------------------
constexpr float MAGIC = 12582912.f;             // 1.5 * 2^23
constexpr unsigned MAGIC_ODD = 301u;            // bits(MAGIC) = 0x4B400000 = 301 << 22
template <int K> __device__ __forceinline__ float magic_units() { return K == 3 ? 4194304.f * 0.999f * 3.375f : (K == 2 ? 4194304.f * 0.999f * (64.f / 27.f) : 4194304.f * 0.999f); }    // 2^22 * 0.999 / wmax^3 (wmax = 1 for K = 1)
// the largest  density * prod_d sum_j max_t w_j(t)  (tile_common.hpp: headroom32) whose slot sums stay inside 32 bits
template <int K> __device__ __forceinline__ float magic_cbmax() { return 2147483648.f * 0.99f / magic_units<K>(); }
// ... and for mixed orders (K = KMIX): wmax per dim, 1 / (3/4) / (2/3) for orders 1 / 2 / 3
template <int K> __device__ __forceinline__ float magic_units_p(const KParams &p)
{
    if (K != KMIX) return magic_units<K>();
    float u = 4194304.f * 0.999f;
#pragma unroll
    for (int d = 0; d < 3; ++d) u *= p.order[d] == 3 ? 1.5f : (p.order[d] == 2 ? 4.f / 3.f : 1.f);
    return u;
}

__device__ __forceinline__ void magic_decode(unsigned long long W, unsigned n, int &s0, int &s1)
{
    const unsigned pr = n * MAGIC_ODD;                               // n < 65536;  n * MAGIC_BITS = pr << 22
    const unsigned mlo = pr << 22, mhi = pr >> 10;
    s0 = (int)((unsigned)W - mlo);                                   // |sum q0| < 2^31
    // the low fields summed to n * MAGIC_BITS + s0 exactly: what lies above bit 31 was carried into the high field
    const unsigned long long L = (((unsigned long long)mhi << 32) | mlo) + (unsigned long long)(long long)s0;
    s1 = (int)((unsigned)(W >> 32) - mlo - (unsigned)(L >> 32));
}

// the K + 1 LDS adds of one row of the stencil, at immediate offsets (i, jy compile-time)
template <int I, int J, int NZT = 4>
__device__ __forceinline__ void row_adds(unsigned addr, unsigned long long v0, unsigned long long v1, unsigned long long v2, unsigned long long v3)
{
    constexpr int o = (I * PLANE + J * PZ) * 8;
    if (NZT == 4)
        asm volatile("ds_add_u64 %0, %1 offset:%5\n\tds_add_u64 %0, %2 offset:%6\n\tds_add_u64 %0, %3 offset:%7\n\tds_add_u64 %0, %4 offset:%8"
                     :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(o), "n"(o + 8), "n"(o + 16), "n"(o + 24) : "memory");
    else if (NZT == 3)
        asm volatile("ds_add_u64 %0, %1 offset:%4\n\tds_add_u64 %0, %2 offset:%5\n\tds_add_u64 %0, %3 offset:%6"
                     :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "n"(o), "n"(o + 8), "n"(o + 16) : "memory");
    else
        asm volatile("ds_add_u64 %0, %1 offset:%3\n\tds_add_u64 %0, %2 offset:%4"
                     :: "v"(addr), "v"(v0), "v"(v1), "n"(o), "n"(o + 8) : "memory");
}
// WIDE: one channel, 64-bit sums of 31-bit terms (dense bricks); else the channel pair in the magic format (above)
template <int K, int I, int J, bool WIDE>
__device__ __forceinline__ void scatter_row(unsigned addr, f2 sx, const f2 *w, int dbg)
{
    const f2 sy = sx * f2{ w[J].x, w[J].x };
    unsigned long long v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (WIDE) {
            v[k] = (unsigned long long)(long long)tiled::cvt_rpi(sy.x * w[k].y);       // cannot overflow
        } else {
            const f2 pr = __builtin_elementwise_fma(sy, f2{ w[k].y, w[k].y }, f2{ MAGIC, MAGIC });
            v[k] = __builtin_bit_cast(unsigned long long, pr);
        }
    }
#ifdef IP_ABLATE
    if (dbg & 4) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); return; }     // (ablation: the arithmetic without the LDS adds)
#endif
    row_adds<I, J, WIDE ? 4 : IP_KS + 1>(addr, v[0], v[1], v[2], v[3]);
}
template <int K, int I, bool WIDE>
__device__ __forceinline__ void scatter_plane(unsigned addr, f2 s, float wxi, const f2 *w, int dbg)
{
#ifdef IP_ABLATE
    if (dbg & 16) {                                                  // (ablation: the LDS adds without the arithmetic)
        const unsigned long long c = 0x100000001ull;
        row_adds<I, 0>(addr, c, c, c, c); row_adds<I, 1>(addr, c, c, c, c); row_adds<I, 2>(addr, c, c, c, c);
        if (IP_KS == 3) row_adds<I, 3>(addr, c, c, c, c);
        return;
    }
#endif
    const f2 sx = s * f2{ wxi, wxi };
    scatter_row<K, I, 0, WIDE>(addr, sx, w, dbg); scatter_row<K, I, 1, WIDE>(addr, sx, w, dbg);
    if (IP_KS >= 2) scatter_row<K, I, 2, WIDE>(addr, sx, w, dbg);
    if (IP_KS == 3) scatter_row<K, I, 3, WIDE>(addr, sx, w, dbg);
}

// Stencils per slot of the box: the (K + 1)^3 box filter of the density of first-tap cells, dim after dim, in place (every thread
// reads its line, barrier, writes it: the three arrays overlap).  In: sm.cells (16-bit counters, cell (x0 * BR + y0) * BR + z0).
// Out: sm.nreg, 16 bits per slot, slot (x, y, z) at (x * BOX + y) * NZ + z.  Destroys cells and queue.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ void slide(const unsigned *in, unsigned *out)
{
    us2 s = { 0, 0 };
#pragma unroll
    for (int j = 0; j < BOX; ++j) {
        if (j < BR) s += __builtin_bit_cast(us2, in[j]);
        if (j >= W && j - W < BR) s -= __builtin_bit_cast(us2, in[j - W]);      // (K = 1: the window leaves the 16 cells before the box ends)
        out[j] = __builtin_bit_cast(unsigned, s);
    }
}
template <int K, typename SM>
__device__ __forceinline__ void stencil_counts(SM &sm, int tid)
{
    constexpr int W = IP_KS + 1, HZ = NZ / 2;
    unsigned *rg = sm.nreg;
    unsigned in[BR], out[BOX];
    // z: thread = row (x0, y0) of 16 cells -> 19 slots (+ one of padding)
    if (tid < BR * BR) {
        const uint4 lo = reinterpret_cast<const uint4 *>(sm.cells)[2 * tid], hi = reinterpret_cast<const uint4 *>(sm.cells)[2 * tid + 1];
        in[0] = lo.x; in[1] = lo.y; in[2] = lo.z; in[3] = lo.w; in[4] = hi.x; in[5] = hi.y; in[6] = hi.z; in[7] = hi.w;
    }
    __syncthreads();
    if (tid < BR * BR) {
        int v[BR], o[NZ], s = 0;
#pragma unroll
        for (int i = 0; i < BR / 2; ++i) { v[2 * i] = (int)(in[i] & 0xffffu); v[2 * i + 1] = (int)(in[i] >> 16); }
#pragma unroll
        for (int j = 0; j < BOX; ++j) { if (j < BR) s += v[j]; if (j >= W && j - W < BR) s -= v[j - W]; o[j] = s; }
        o[BOX] = 0;
#pragma unroll
        for (int i = 0; i < HZ; ++i) rg[tid * HZ + i] = (unsigned)o[2 * i] | ((unsigned)o[2 * i + 1] << 16);
    }
    __syncthreads();
    // y: thread = (x0, pair of slots along z): 16 rows -> 19
    {
        const int x = tid / HZ, zp = tid - x * HZ;
        if (tid < BR * HZ) {
#pragma unroll
            for (int k = 0; k < BR; ++k) in[k] = rg[(x * BR + k) * HZ + zp];
        }
        __syncthreads();
        if (tid < BR * HZ) {
            slide<W>(in, out);
#pragma unroll
            for (int j = 0; j < BOX; ++j) rg[(x * BOX + j) * HZ + zp] = out[j];
        }
        __syncthreads();
    }
    // x: thread = (row y, pair of slots along z): 16 planes -> 19
    {
        const int y = tid / HZ, zp = tid - y * HZ;
        if (tid < BOX * HZ) {
#pragma unroll
            for (int k = 0; k < BR; ++k) in[k] = rg[(k * BOX + y) * HZ + zp];
        }
        __syncthreads();
        if (tid < BOX * HZ) {
            slide<W>(in, out);
#pragma unroll
            for (int j = 0; j < BOX; ++j) rg[(j * BOX + y) * HZ + zp] = out[j];
        }
        __syncthreads();
    }
}
__device__ __forceinline__ unsigned stencils_at(const AccSmem &sm, int xr, int yr, int zr)
{
    return (unsigned)reinterpret_cast<const unsigned short *>(sm.nreg)[(xr * BOX + yr) * NZ + zr];
}

// the x-weight of tap i (mixed orders: by the dim's own order; 0 beyond it)
template <int K> __device__ __forceinline__ float own_weight_x(const KParams &p, float t, int i)
{
    if constexpr (K == KMIX) return mixed_weight_x(p.order[0], t, i);
    else return weight_x<K>(t, i);
}
template <int K> __device__ __forceinline__ float own_wgrad_x(const KParams &p, float t, int i)
{
    if constexpr (K == KMIX) return mixed_wgrad_x(p.order[0], t, i);
    else return wgrad_x<K>(t, i);
}
// first-tap cell and stencil coordinates of a record (same arithmetic as own_bin: nd.py:45-46)
#ifdef IP_OWNER_MIX_TU
#define IP_RECORD_CELL(...) record_cell<K>(p, __VA_ARGS__)
template <int K>
__device__ __forceinline__ void record_cell(const KParams &p, const float4 &r, const int *b0, int &x0, int &y0, int &z0, float &tx, float &ty, float &tz)
{
    const float fx = floorf(r.x - 0.5f * (float)(kd<K>(p, 0) - 1)), fy = floorf(r.y - 0.5f * (float)(kd<K>(p, 1) - 1)), fz = floorf(r.z - 0.5f * (float)(kd<K>(p, 2) - 1));
#else
#define IP_RECORD_CELL(...) record_cell<K>(__VA_ARGS__)
template <int K>
__device__ __forceinline__ void record_cell(const float4 &r, const int *b0, int &x0, int &y0, int &z0, float &tx, float &ty, float &tz)
{
    const float fx = floorf(r.x - 0.5f * (float)(K - 1)), fy = floorf(r.y - 0.5f * (float)(K - 1)), fz = floorf(r.z - 0.5f * (float)(K - 1));
#endif
    tx = r.x - fx; ty = r.y - fy; tz = r.z - fz;
    x0 = (__float2int_rz(fx) - b0[0]) & (BR - 1); y0 = (__float2int_rz(fy) - b0[1]) & (BR - 1); z0 = (__float2int_rz(fz) - b0[2]) & (BR - 1);
}

// first brick index and number of bricks a launch enumerates along dim d (host and device)
__host__ __device__ __forceinline__ int color_first(int color, int d)
{
    if (color >= 8) return 0;
    const int par = (color >> (2 - d)) & 1;
    return NLO + ((par - NLO) & 1);                                  // first interior brick index of that parity
}
__host__ __device__ __forceinline__ int color_count(int color, int d, const BrickGrid &bg)
{
    if (color >= 8) return bg.nb[d];
    const int j0 = color_first(color, d) - NLO;
    return bg.nin[d] > j0 ? (bg.nin[d] - j0 + 1) >> 1 : 0;
}

// COLOR 0..7: the interior bricks of that parity, flushed with plain loads / stores; COLOR 8: the bricks whose box
// leaves the lattice, flushed with atomics through the boundary tables; COLOR 9: every brick, with atomics (shared target)
//
// The taps.  The LDS box is 19^3 slots of 8 bytes, row-major.  A ds_add_u64 of a wave is served per 32-lane half; it is
// conflict-free when the 32 lanes hit 32 different 8-byte bank pairs, i.e. different (slot mod 32) -- and all taps of a
// stencil add the same offset in every lane (tools/microbench/lds_gather.hip: 6.3 clk against 12.6 for random lanes;
// measured here, unsorted: 21 clk).  So the records of a batch are counting-sorted by the CLASS (first slot mod 32) into
// an index queue in LDS, and lane q of every half wave walks class q: half wave h takes entries h, h + 16, ... of its class.
// NEAR (K == 1 only): the nearest-neighbour scatter (all orders 0) -- the float box below.  A template parameter since round 6's last evidence
// pass: inside the trilinear kernel that code cost it 144 B of scratch per lane (240 against 96) and 13 - 15 % of its time (4 x 2 x 256^3, sigma = 2:
// 2.43 -> 2.76 ms, profiles/r06_other_configs.json against r05's).
template <int K, bool NEAR = false>
__global__ __launch_bounds__(NT, 4) void own_accumulate(KParams p, BrickGrid bg, const int *__restrict__ ndesc, const uint2 *__restrict__ desc,
                                                        const float4 *__restrict__ rec, const float *__restrict__ vals,
                                                        const unsigned short *__restrict__ meta, const int *__restrict__ bmax, int64_t nrec,
                                                        float *__restrict__ vol, int nch, int color, int nbatch, const int *__restrict__ gate,
                                                        int *__restrict__ ctr)
{
    if (gate && *gate != 1) return;                                  // INTERPOL_FLAG_AUTO_SCATTER: the probe chose the tiles
    if (color == 8 && ctr[HDR_SHELL - 16 - 8] == 0) return;          // the shell launch: own_bin published no run outside the interior bricks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    AccSmem &sm = *reinterpret_cast<AccSmem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = IP_KD(d); }
    L.lin = 0;
    // bricks of this launch: a colour enumerates the INTERIOR bricks of its parity only (every work item then carries a
    // brick's worth of samples: the workgroups stay balanced); the other launches enumerate all bricks
    const int step = color < 8 ? 2 : 1;
    const int c0[3] = { color_first(color, 0), color_first(color, 1), color_first(color, 2) };
    const int m0 = color_count(color, 0, bg), m1 = color_count(color, 1, bg), m2 = color_count(color, 2, bg);
    const int nwork = m0 * m1 * m2 * nbatch;
    for (int e = threadIdx.x; e < BOXSLOTS; e += NT) sm.box[e] = 0ull;
    // Colour launches: every candidate is an interior brick; they are drawn one at a time from a counter of the launch
    // (balanced whatever the brick count).  Other launches: most candidates hold nothing; a workgroup examines 64 at once (one per
    // lane: candidates blockIdx + (64 k + lane) gridDim, interleaved over the workgroups -- the bricks that hold records cluster
    // along the faces of the lattice) and visits the ones that hold records.
    const bool dynamic = color < 8;
    __shared__ int next_chunk;
    for (int round = 0; ; ++round) {
    int chunk, stride;
    if (dynamic) {
        __syncthreads();
        if (threadIdx.x == 0) next_chunk = atomicAdd(ctr, 1);
        __syncthreads();
        chunk = next_chunk; stride = nwork;                          // (one candidate: lane 0)
    } else {
        chunk = (int)blockIdx.x + round * 64 * (int)gridDim.x; stride = (int)gridDim.x;
    }
    if (chunk >= nwork) break;
    unsigned long long pending;
    {
        const long long wl = (long long)chunk + (long long)(threadIdx.x & 63) * stride;
        const int w = wl < nwork ? (int)wl : nwork;
        bool take = false;
        if (w < nwork) {
            int r = w;
            const int iz = r % m2; r /= m2;
            const int iy = r % m1; r /= m1;
            const int ix = r % m0;
            const int bb = r / m0;
            const int bx_[3] = { ix * step + c0[0], iy * step + c0[1], iz * step + c0[2] };
            bool interior = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) interior = interior && bx_[d] >= NLO + (L.bound[d] == B_DST1 ? 1 : 0) && bx_[d] < NLO + bg.nin[d];
            if (color < 8 ? interior : (color == 9 || !interior))
                take = ndesc[bb * bg.item + (bx_[0] * bg.nb[1] + bx_[1]) * bg.nb[2] + bx_[2]] != 0;
        }
        pending = __ballot(take);
    }
    while (pending) {
        const int work = chunk + (__ffsll((long long)pending) - 1) * stride;
        pending &= pending - 1;
        const int tid = opaque((int)threadIdx.x);
        int r = work;
        const int iz = r % m2; r /= m2;
        const int iy = r % m1; r /= m1;
        const int ix = r % m0;
        const int64_t b = r / m0;
        const int bxyz[3] = { ix * step + c0[0], iy * step + c0[1], iz * step + c0[2] };
        const int b0[3] = { brick_origin(bxyz[0], bg.lo[0], bg.top[0], bg.nin[0], bg.split[0]), brick_origin(bxyz[1], bg.lo[1], bg.top[1], bg.nin[1], bg.split[1]),
                            brick_origin(bxyz[2], bg.lo[2], bg.top[2], bg.nin[2], bg.split[2]) };                       // lattice index of box slot 0
        // interior: every stencil of the brick lies inside the lattice (dst1: index 0 carries the sign 0, bounds.py:62-89 -- tables)
        bool interior = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) interior = interior && bxyz[d] >= NLO + (L.bound[d] == B_DST1 ? 1 : 0) && bxyz[d] < NLO + bg.nin[d];
        if (color < 8 ? !interior : (color == 8 && interior)) continue;       // (block-uniform)
        const bool atomic = color >= 8;
        // bricks at the ends of a folding dim (BrickGrid): part of the box lies outside the lattice
        bool edge = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) edge = edge || b0[d] < 0 || b0[d] + BOX > L.n[d];
        edge = edge && !atomic;
        // ... and a lattice point of the box then collects the sums of `foldmul` points: its mirror image per dim (dct1, dct2), all
        // the points beyond the end (replicate) -- the fixed-point headroom counts them in (many: the 64-bit sums)
        int foldmul = 1;
        if (edge) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int nout = b0[d] < 0 ? -b0[d] : (b0[d] + BOX > L.n[d] ? b0[d] + BOX - L.n[d] : 0);
                foldmul *= nout == 0 ? 1 : (L.bound[d] == B_REPLICATE ? nout + 1 : 2);
            }
        }
        const int brick = (int)b * bg.item + (bxyz[0] * bg.nb[1] + bxyz[1]) * bg.nb[2] + bxyz[2];
        int nd = ndesc[brick];
        if (nd == 0) continue;                                       // (block-uniform)
        nd = nd < bg.capd ? nd : bg.capd;
        __syncthreads();                                             // the previous brick's flush is done with the tables / the box
        prof_mark(-1);
        if (tid < 64) {
            // runs of the brick: records, pieces (<= 64 consecutive records) and the exclusive prefix of the piece counts
            constexpr int PER = CAPX / 64;
            int cn[PER], np[PER], s = 0, sp = 0; unsigned st_[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = tid * PER + i;
                const uint2 dsc = e < nd ? desc[(int64_t)brick * bg.capd + e] : make_uint2(0u, 0u);
                cn[i] = (int)dsc.y; st_[i] = dsc.x; np[i] = (cn[i] + 63) >> 6; s += cn[i]; sp += np[i];
            }
            int incl = s, inclp = sp;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o), tp = __shfl_up(inclp, o);
                if (tid >= o) { incl += t; inclp += tp; }
            }
            int runp = inclp - sp;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = tid * PER + i;
                const unsigned last = cn[i] > 0 ? (unsigned)(cn[i] - 64 * (np[i] - 1)) : 0u;
                sm.ppref[e] = e < nd ? ((unsigned)runp & PMASK) | (last << 25) : 0xffffffffu;
                sm.start[e] = st_[i];
                runp += np[i];
            }
            if (tid == 63) { sm.n = incl; sm.npieces = inclp; sm.dmax = 0; sm.cmax[0] = 0; sm.cmax[1] = 0; }
            if (tid < 3) { sm.rlo[tid] = BR; sm.rhi[tid] = -1; }
        } else if (atomic && tid < 64 + 3 * 32) {
            const int d = (tid - 64) >> 5, slot = tid & 31;
            if (slot < BOX) {
                const long long pk = wrap_outofline(L.bound[d], b0[d] + slot, L.n[d]);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * L.ss[d];
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        for (int e = tid; e < NCELL / 2; e += NT) sm.cells[e] = 0u;
        if (tid < NCLS) sm.qcnt[tid] = 0;
        __syncthreads();
        const int n = sm.n, npieces = sm.npieces;
        const int wave = tid >> 6, lane = tid & 63;
        // piece table of batch `pb0 / NPIECE`: piece g -> run (search over <= 256 runs, once per 64 records), first record, length
        auto build_pieces = [&](int pb0) {
            if (tid < NPIECE) {
                const int g = pb0 + tid;
                uint2 pc = make_uint2(0u, 0u);
                if (g < npieces) {
                    int j = 0;
#pragma unroll
                    for (int st = CAPX / 2; st > 0; st >>= 1) j += (int)(sm.ppref[j + st] & PMASK) <= g ? st : 0;
                    const unsigned pj = sm.ppref[j];
                    const int q = g - (int)(pj & PMASK);
                    // (the run's last piece -- the next run starts behind it -- holds what is left of the run, every other one 64 records)
                    const int nextp = j + 1 < CAPX ? (int)(sm.ppref[j + 1] & PMASK) : (int)PMASK;
                    const bool lastp = g + 1 == (nextp < npieces ? nextp : npieces);
                    pc = make_uint2(sm.start[j] + 64u * (unsigned)q, lastp ? (pj >> 25) : 64u);
                }
                sm.piece[tid] = pc;
            }
            __syncthreads();
        };
        prof_mark(8);
        if (NEAR) {
            // Nearest neighbour (all orders 0; own_bin stored the ROUNDED coordinates): a sample touches ONE lattice point with weight 1
            // (iso0.py:65-118: inp * sign * mask, scatter_add_ in no particular order).  Round 6: no fixed point here -- the box holds the
            // channel pair as two FLOATS and every record adds its sources with one ds_add_f32 each (slow LDS atomics, ~190 clk per wave
            // instruction, but 2 per record instead of the 64 packed ones of a cubic stencil), the flush adds the non-zero slots to the
            // target.  A lattice point hit by a single sample holds that sample's value bit for bit, a non-finite source stays on its own
            // lattice point (round 5's 2 x 2 x 2 stencil at t = 0 quantised the value to the brick's fixed point and turned the seven
            // zero-weight neighbours of an inf into NaN), sums of several hits round like the reference's.  Stencils of one point need
            // no folding pass: a record outside the lattice goes to its image under the boundary condition, which lies in the same box
            // (BrickGrid).  Shell bricks (colour 8: few records) add straight to the target, one float atomic per record and channel
            // (all bricks that way: 5.8 ms at config-2 size -- scattered global atomics, profiles/r06_nearest_push.txt).
            float *boxf = reinterpret_cast<float *>(sm.box);
            for (int c = 0; c < nch; c += 2) {
                const bool two = c + 1 < nch;
                float *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
                float *vc1 = two ? vc0 + p.vol_sc : vc0;
                const float *va = c == 0 ? nullptr : vals + (int64_t)(c - 1) * nrec;
                const float *vb = two ? vals + (int64_t)c * nrec : nullptr;
                for (int pb0 = 0; pb0 < npieces; pb0 += NPIECE) {
                    if (pb0 > 0 || c > 0) __syncthreads();
                    build_pieces(pb0);
#pragma unroll 1
                    for (int k = 0; k < VPT; ++k) {
                        const uint2 pc = sm.piece[wave + k * (NT / 64)];
                        if (lane >= (int)pc.y) continue;
                        const unsigned ri = pc.x + (unsigned)lane;
                        const float4 rc = rec[ri];
                        const float s0 = va ? va[ri] : rc.w, s1 = vb ? vb[ri] : 0.f;
                        int ii[3] = { __float2int_rn(rc.x), __float2int_rn(rc.y), __float2int_rn(rc.z) };
                        float sgn = 1.f;
                        if (atomic || ii[0] < 0 || ii[0] >= L.n[0] || ii[1] < 0 || ii[1] >= L.n[1] || ii[2] < 0 || ii[2] >= L.n[2]) {
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                const long long pk = wrap_outofline(L.bound[d], ii[d], L.n[d]);
                                ii[d] = (int)(pk & 0xffffffffll); sgn *= (float)(int)(pk >> 32);
                            }
                        }
                        const int xr = ii[0] - b0[0], yr = ii[1] - b0[1], zr = ii[2] - b0[2];
                        if (atomic || (unsigned)xr >= (unsigned)BOX || (unsigned)yr >= (unsigned)BOX || (unsigned)zr >= (unsigned)BOX) {
                            // shell bricks; (never for interior ones: the image of a point of an end brick's box lies in the box)
                            const int off = ii[0] * L.ss[0] + ii[1] * L.ss[1] + ii[2] * L.ss[2];
                            __hip_atomic_fetch_add(vc0 + off, s0 * sgn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (two) __hip_atomic_fetch_add(vc1 + off, s1 * sgn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            continue;
                        }
                        float *slot = boxf + 2 * (xr * PLANE + yr * PZ + zr);
                        __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)slot, s0 * sgn, 0, 0, false);
                        if (two) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(slot + 1), s1 * sgn, 0, 0, false);
                    }
                }
                __syncthreads();
                if (!atomic) {
                    constexpr int UF0 = 7;
#pragma unroll 1
                    for (int e0 = tid; e0 < BOXSLOTS; e0 += UF0 * NT) {
                        int off[UF0]; float a0[UF0], a1[UF0], t0[UF0], t1[UF0];
                        unsigned nz = 0;
#pragma unroll
                        for (int u = 0; u < UF0; ++u) {
                            const int e = e0 + u * NT;
                            off[u] = 0; a0[u] = 0.f; a1[u] = 0.f;
                            if (e < BOXSLOTS) {
                                a0[u] = boxf[2 * e]; a1[u] = boxf[2 * e + 1];
                                if (a0[u] != 0.f || a1[u] != 0.f) {           // (true for NaN)
                                    const int xr = e / PLANE, rem = e - xr * PLANE, yr = rem / PZ, zr = rem - yr * PZ;
                                    off[u] = (b0[0] + xr) * L.ss[0] + (b0[1] + yr) * L.ss[1] + (b0[2] + zr) * L.ss[2];
                                    nz |= 1u << u;
                                    sm.box[e] = 0ull;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < UF0; ++u) {
                            t0[u] = ((nz >> u) & 1) ? vc0[off[u]] : 0.f;
                            t1[u] = (((nz >> u) & 1) && two) ? vc1[off[u]] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < UF0; ++u) {
                            if (!((nz >> u) & 1)) continue;
                            vc0[off[u]] = t0[u] + a0[u];
                            if (two) vc1[off[u]] = t1[u] + a1[u];
                        }
                    }
                }
                __syncthreads();
            }
            continue;                                                // (block-uniform) next brick
        }
        // ---- pass 1 over all records: density of the first-tap cells (own_bin left every record's cell in `meta`, 2 bytes, and the
        // brick's max |source| of the first channel pair in `bmax`); the records of the first batch are also counted into their
        // classes (rank kept in registers: qr = class | rank << 5)
        int qr[VPT];
        if (tid == 0) { sm.cmax[0] = bmax[2 * (int64_t)brick]; sm.cmax[1] = nch > 1 ? bmax[2 * (int64_t)brick + 1] : 0; }
        for (int pb0 = 0; pb0 < npieces; pb0 += NPIECE) {
            if (pb0 > 0) __syncthreads();                            // (every wave is done with the previous piece table)
            build_pieces(pb0);
            unsigned mk[VPT];
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                const uint2 pc = sm.piece[wave + k * (NT / 64)];
                mk[k] = lane < (int)pc.y ? (unsigned)meta[pc.x + (unsigned)lane] : 0xffffffffu;
            }
#pragma unroll
            for (int k = 0; k < VPT; ++k) {
                if (pb0 == 0) qr[k] = -1;
                if (mk[k] != 0xffffffffu) {
                    const int x0 = (mk[k] >> 8) & 15, y0 = (mk[k] >> 4) & 15, z0 = mk[k] & 15;
                    const int cell = (int)(mk[k] & 4095u);
                    atomicAdd(&sm.cells[cell >> 1], 1u << (16 * (cell & 1)));
                    if (pb0 == 0) {
                        const int q = (x0 * PLANE + y0 * PZ + z0) & (NCLS - 1);
                        qr[k] = q | (atomicAdd(&sm.qcnt[q], 1) << 4);
                    }
                }
            }
        }
        __syncthreads();
        prof_mark(9);
        {
            int dm = 0;
            int mn[3] = { BR, BR, BR }, mx[3] = { -1, -1, -1 };
            for (int e = tid; e < NCELL / 2; e += NT) {
                const unsigned w2 = sm.cells[e];
                const int a = (int)(w2 & 0xffffu), c2 = (int)(w2 >> 16);
                dm = a > dm ? a : dm; dm = c2 > dm ? c2 : dm;
                if (atomic && w2) {                                  // cells 2e, 2e + 1 = (x0 * BR + y0) * BR + z0
                    const int x0 = e >> 7, y0 = (e >> 3) & 15, z0 = (2 * e) & 15;
                    mn[0] = x0 < mn[0] ? x0 : mn[0]; mx[0] = x0 > mx[0] ? x0 : mx[0];
                    mn[1] = y0 < mn[1] ? y0 : mn[1]; mx[1] = y0 > mx[1] ? y0 : mx[1];
                    mn[2] = z0 < mn[2] ? z0 : mn[2]; mx[2] = z0 + 1 > mx[2] ? z0 + 1 : mx[2];
                }
            }
            dm = wave_max(dm);
            if (lane == 0 && dm > 0) atomicMax(&sm.dmax, dm);
            if (atomic) {
                // a brick of the shell holds a few layers of samples: only the part of the box they reach is flushed
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int a = wave_min(mn[d]), c2 = wave_max(mx[d]);
                    if (lane == 0) { atomicMin(&sm.rlo[d], a); atomicMax(&sm.rhi[d], c2); }
                }
            }
        }
        __syncthreads();
        prof_mark(10);
        // 32-bit channel pairs while no slot's sums can leave 32 bits: density * prod_d sum_j max_t w_j(t) units of max |source|
        // (tile_common.hpp: headroom32) -- 32 samples per first-tap cell for cubic stencils; beyond: 64-bit sums, one channel per pass
#ifdef IP_OWNER_MIX_TU
        float wsum3 = 1.f;                                           // prod_d sum_j max_t w_j(t): 2 / 1.75 / 1.6667 for orders 1 / 2 / 3
#pragma unroll
        for (int d = 0; d < 3; ++d) wsum3 *= kd<K>(p, d) == 3 ? 1.6666667f : (kd<K>(p, d) == 2 ? 1.75f : 2.f);
        const bool dense = n >= 60000 || (float)(sm.dmax * foldmul) * wsum3 > 2147483648.f * 0.99f / magic_units_p<K>(p);    // (16-bit density counters)
#else
        const float wsum = K == 3 ? 1.6666667f : (K == 2 ? 1.75f : 2.f);
        const bool dense = n >= 60000 || (float)(sm.dmax * foldmul) * (wsum * wsum * wsum) > magic_cbmax<K>();    // (16-bit density counters)
#endif
        const bool one_batch = npieces <= NPIECE;
        for (int c = 0; c < nch; c += 2) {
            const bool two = c + 1 < nch;
            float *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
            float *vc1 = two ? vc0 + p.vol_sc : vc0;
            const float *va = c == 0 ? nullptr : vals + (int64_t)(c - 1) * nrec;       // channel c (channel 0 travels in the record)
            const float *vb = two ? vals + (int64_t)c * nrec : nullptr;                 // channel c + 1
            if (c > 0) {
                // max |source| of this channel pair; the density of the first-tap cells again (the stencil counts of the previous
                // pair took its place)
                float am0 = 0.f, am1 = 0.f;
                for (int e = tid; e < NCELL / 2; e += NT) sm.cells[e] = 0u;
                __syncthreads();
                for (int pb0 = 0; pb0 < npieces; pb0 += NPIECE) {
                    if (!one_batch) { __syncthreads(); build_pieces(pb0); }
                    for (int k = 0; k < VPT; ++k) {
                        const uint2 pc = sm.piece[wave + k * (NT / 64)];
                        if (lane >= (int)pc.y) continue;
                        const unsigned ri = pc.x + (unsigned)lane;
                        const int cell = (int)(meta[ri] & 4095u);
                        atomicAdd(&sm.cells[cell >> 1], 1u << (16 * (cell & 1)));
                        const float a0 = __builtin_fabsf(va[ri]), a1 = two ? __builtin_fabsf(vb[ri]) : 0.f;
                        am0 = (a0 > am0 || a0 != a0) ? a0 : am0;
                        am1 = (a1 > am1 || a1 != a1) ? a1 : am1;
                    }
                }
                const int w0 = wave_max(__float_as_int(am0)), w1 = wave_max(__float_as_int(am1));
                if (lane == 0) { if (w0) atomicMax(&sm.cmax[0], w0); if (w1) atomicMax(&sm.cmax[1], w1); }
                __syncthreads();
            }
            const int mb0 = sm.cmax[0], mb1 = sm.cmax[1];
            const bool fin = (mb0 & 0x7f800000) != 0x7f800000 && (mb1 & 0x7f800000) != 0x7f800000;
            // 32-bit channel pairs (magic format) when the density allows; else WIDE: one channel per pass, 64-bit sums of 31-bit
            // terms (strongly contracting deformations: folds, strides); float atomics only for non-finite sources
            const bool wide = dense && fin;
            const bool fixedpt = fin && !(p.dbg & 8);
            int ex0 = ((mb0 >> 23) & 0xff) - 127, ex1 = ((mb1 >> 23) & 0xff) - 127;
            ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
            // magic format: |source| * scale * w_x w_y w_z < 2^22 -- scale = magic_units / max |source| (not a power of two: the
            // products carry float roundings like the reference's own)
            const float a0 = fmaxf(__int_as_float(mb0), 1e-27f), a1 = fmaxf(__int_as_float(mb1), 1e-27f);
            const f2 scale = { mb0 ? IP_MU / a0 : 0.f, mb1 ? IP_MU / a1 : 0.f };
            const float inv0 = a0 * (1.f / IP_MU), inv1 = a1 * (1.f / IP_MU);
            const f2 scalew = { mb0 ? __int_as_float((127 + 30 - ex0) << 23) : 0.f, mb1 ? __int_as_float((127 + 30 - ex1) << 23) : 0.f };
            const float invw0 = __int_as_float((127 - 30 + ex0) << 23), invw1 = __int_as_float((127 - 30 + ex1) << 23);
            const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
            for (int sub = 0; sub < (wide && two ? 2 : 1); ++sub) {
            for (int pb0 = 0; pb0 < npieces; pb0 += NPIECE) {
                const int tid = opaque((int)threadIdx.x);
                const int wave = tid >> 6, lane = tid & 63;
                if (pb0 > 0 || c > 0 || sub > 0) {
                    // classes of a later batch / of the same records for a later channel (pair): count again
                    __syncthreads();
                    if (tid < NCLS) sm.qcnt[tid] = 0;
                    if (!one_batch) build_pieces(pb0); else __syncthreads();
#pragma unroll
                    for (int k = 0; k < VPT; ++k) {
                        const uint2 pc = sm.piece[wave + k * (NT / 64)];
                        qr[k] = -1;
                        if (lane < (int)pc.y) {
                            const unsigned mk = meta[pc.x + (unsigned)lane];
                            const int x0 = (mk >> 8) & 15, y0 = (mk >> 4) & 15, z0 = mk & 15;
                            const int q = (x0 * PLANE + y0 * PZ + z0) & (NCLS - 1);
                            qr[k] = q | (atomicAdd(&sm.qcnt[q], 1) << 4);
                        }
                    }
                    __syncthreads();
                } else if (!one_batch) {
                    __syncthreads();
                    build_pieces(0);                                 // (pass 1 left the table of the LAST batch)
                }
                // Every class owns qcap = ceil(records / 32) slots of the queue (rounded to the NHW half waves that walk it): the
                // records a class holds beyond that fill the free slots of the others (those lanes may collide on a bank pair),
                // so that every lane walks the same number of entries -- the classes of an i.i.d. field are Poisson-filled, the
                // fullest of 32 holds 20 % more than the mean.
                if (tid < 32) {
                    const bool cls = tid < NCLS;
                    const int cq = cls ? sm.qcnt[tid] : 0;
                    int tot, nsur, nhol;
                    half_excl_scan(cq, tot);
                    const int cap = (((tot + NCLS - 1) / NCLS) + NHW - 1) / NHW * NHW;
                    const int sur = cls && cq > cap ? cq - cap : 0, hol = cls && cq < cap ? cap - cq : 0;
                    const int so = half_excl_scan(sur, nsur), ho = half_excl_scan(hol, nhol);
                    if (cls) { sm.qsur[tid] = so; sm.qhol[tid] = ho; }
                    if (tid == 31) { sm.qsur[NCLS] = nsur; sm.qhol[NCLS] = nhol; sm.qcap = cap; }
                    const int fill = nsur - ho < 0 ? 0 : (nsur - ho > hol ? hol : nsur - ho);
                    if (cls) sm.qeff[tid] = (cq < cap ? cq : cap) + fill;
                }
                __syncthreads();
                const int qcap = sm.qcap;
#pragma unroll
                for (int k = 0; k < VPT; ++k) {
                    if (qr[k] < 0) continue;
                    const int qq = qr[k] & (NCLS - 1), r = qr[k] >> 4;
                    int slot = qq * qcap + r;
                    if (r >= qcap) {                                 // surplus record o of the batch: into the o-th free slot
                        const int o = sm.qsur[qq] + r - qcap;
                        int c2 = 0;
#pragma unroll
                        for (int st = NCLS / 2; st > 0; st >>= 1) if (sm.qhol[c2 + st] <= o) c2 += st;      // last class with qhol <= o
                        slot = c2 * qcap + sm.qcnt[c2] + (o - sm.qhol[c2]);
                    }
                    sm.queue[slot] = (unsigned short)(((wave + k * (NT / 64)) << 6) | lane);
                }
                __syncthreads();
                // ---- the taps: lane q of half wave hw walks the slots of class q
                const int q = tid & (NCLS - 1), hw = tid >> 4;
                const int ebeg = q * qcap + hw, eend = q * qcap + sm.qeff[q];
                const int nit = qcap / NHW;                          // (block-uniform)
                auto fetch = [&](int e, float4 &rc, float &s0, float &s1) {
                    const unsigned qe = sm.queue[e];
                    const unsigned ri = sm.piece[qe >> 6].x + (qe & 63u);
                    rc = rec[ri]; s0 = va ? va[ri] : rc.w; s1 = vb ? vb[ri] : 0.f;
                };
                float4 rc = make_float4(0.f, 0.f, 0.f, 0.f); float s0 = 0.f, s1 = 0.f;
                if (ebeg < eend) fetch(ebeg, rc, s0, s1);
#pragma unroll 1
                for (int it = 0; it < nit; ++it) {
                    const int e = ebeg + it * NHW;
                    const float4 cur = rc; const float cs0 = s0, cs1 = s1;
                    if (e + NHW < eend) fetch(e + NHW, rc, s0, s1);
                    if (e >= eend) continue;
                    int x0, y0, z0; float tx, ty, tz;
                    IP_RECORD_CELL(cur, b0, x0, y0, z0, tx, ty, tz);
                    if (fixedpt) {
#ifdef IP_ABLATE
                        if (p.dbg & 2) continue;                     // (ablation: no taps)
#endif
                        unsigned addr = boxaddr + 8u * (unsigned)(x0 * PLANE + y0 * PZ + z0);
                        f2 w[4];
#ifdef IP_OWNER_MIX_TU
                        if constexpr (K == KMIX) mixed_weights_yz(p.order[1], p.order[2], f2{ ty, tz }, w);
                        else
#endif
                        weights_yz<K>(f2{ ty, tz }, w);
                        if (wide) {
                            const f2 ss = sub ? f2{ cs1 * scalew.y, 0.f } : f2{ cs0 * scalew.x, 0.f };
                            scatter_plane<K, 0, true>(addr, ss, IP_WX(0), w, p.dbg);
                            scatter_plane<K, 1, true>(addr, ss, IP_WX(1), w, p.dbg);
                            if (IP_KS >= 2) scatter_plane<K, 2, true>(addr, ss, IP_WX(2), w, p.dbg);
                            if (IP_KS == 3) scatter_plane<K, 3, true>(addr, ss, IP_WX(3), w, p.dbg);
                        } else {
                            const f2 ss = f2{ cs0, cs1 } * scale;
                            scatter_plane<K, 0, false>(addr, ss, IP_WX(0), w, p.dbg);
                            scatter_plane<K, 1, false>(addr, ss, IP_WX(1), w, p.dbg);
                            if (IP_KS >= 2) scatter_plane<K, 2, false>(addr, ss, IP_WX(2), w, p.dbg);
                            if (IP_KS == 3) scatter_plane<K, 3, false>(addr, ss, IP_WX(3), w, p.dbg);
                        }
                    } else {
                        // no fixed point for this brick (density beyond the precision rule, non-finite sources): float atomics,
                        // straight to global memory.  Safe inside a colour: the taps stay inside this brick's own box.
                        tiled::scatter_one_thread(L, vc0, cs0, b0[0] + x0, b0[1] + y0, b0[2] + z0, tx, ty, tz);
                        if (two) tiled::scatter_one_thread(L, vc1, cs1, b0[0] + x0, b0[1] + y0, b0[2] + z0, tx, ty, tz);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            prof_mark(11);
            const bool magic = fixedpt && !wide;
            if (magic) stencil_counts<K>(sm, tid);                   // (block-uniform)
            prof_mark(13);
            // ---- bricks at the ends of a folding dim: the planes / rows / slices of the box that lie outside the lattice are added
            // onto their mirror images (replicate: the end point; dct1 / dct2: bounds.py:30-61) -- inside the same box, dim after dim
            if (edge && fixedpt) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int nlo = b0[d] < 0 ? -b0[d] : 0, nhi = b0[d] + BOX > L.n[d] ? b0[d] + BOX - L.n[d] : 0;     // (block-uniform)
                    if (nlo + nhi == 0) continue;
                    for (int i = tid; i < (nlo + nhi) * (BOX * BOX); i += NT) {
                        const int o = i / (BOX * BOX), rest = i - o * (BOX * BOX), r1 = rest / BOX, r2 = rest - r1 * BOX;
                        const int rd = o < nlo ? o : BOX - nhi + (o - nlo);
                        const int idx = b0[d] + rd;
                        int j;
                        if (L.bound[d] == B_REPLICATE) j = idx < 0 ? 0 : L.n[d] - 1;
                        else if (L.bound[d] == B_DCT1) j = idx < 0 ? -idx : 2 * L.n[d] - 2 - idx;
                        else j = idx < 0 ? -1 - idx : 2 * L.n[d] - 1 - idx;
                        const int rt = j - b0[d];
                        const int es = d == 0 ? rd * PLANE + r1 * PZ + r2 : d == 1 ? r1 * PLANE + rd * PZ + r2 : r1 * PLANE + r2 * PZ + rd;
                        const int et = d == 0 ? rt * PLANE + r1 * PZ + r2 : d == 1 ? r1 * PLANE + rt * PZ + r2 : r1 * PLANE + r2 * PZ + rt;
                        const unsigned long long v = sm.box[es];
                        if (v) { sm.box[es] = 0ull; atomicAdd(&sm.box[et], v); }
                        if (magic) {
                            // ... and the stencil counts with them (16-bit halves of a word: no sum reaches 2^16)
                            const int ns = d == 0 ? (rd * BOX + r1) * NZ + r2 : d == 1 ? (r1 * BOX + rd) * NZ + r2 : (r1 * BOX + r2) * NZ + rd;
                            const int nt = d == 0 ? (rt * BOX + r1) * NZ + r2 : d == 1 ? (r1 * BOX + rt) * NZ + r2 : (r1 * BOX + r2) * NZ + rt;
                            unsigned short *nn = reinterpret_cast<unsigned short *>(sm.nreg);
                            const unsigned c2 = nn[ns];
                            if (c2) { nn[ns] = 0; atomicAdd(&sm.nreg[nt >> 1], c2 << (16 * (nt & 1))); }
                        }
                    }
                    __syncthreads();
                }
            }
            // ---- flush: fixed point -> float; the box is re-zeroed on the way
            if (fixedpt && !atomic && !wide && !(p.dbg & 1)) {
                // colour launches, channel pairs (the common case): the loads of seven of the thread's lattice points are in flight
                // at once -- two exposed round trips to the target instead of four; the sums stay in LDS meanwhile
                constexpr int UF2 = 7;
#pragma unroll 1
                for (int e0 = tid; e0 < BOXSLOTS; e0 += UF2 * NT) {
                    int off[UF2]; float t0[UF2], t1[UF2];
                    unsigned nz = 0, nst[UF2];
#pragma unroll
                    for (int u = 0; u < UF2; ++u) {
                        const int e = e0 + u * NT;
                        off[u] = 0;
                        if (e < BOXSLOTS) {
                            const int xr = e / PLANE, rem = e - xr * PLANE, yr = rem / PZ, zr = rem - yr * PZ;
                            off[u] = (b0[0] + xr) * L.ss[0] + (b0[1] + yr) * L.ss[1] + (b0[2] + zr) * L.ss[2];
                            nst[u] = stencils_at(sm, xr, yr, zr);
                            if (nst[u] != 0u) nz |= 1u << u;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UF2; ++u) {
                        t0[u] = ((nz >> u) & 1) ? vc0[off[u]] : 0.f;
                        t1[u] = (((nz >> u) & 1) && two) ? vc1[off[u]] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < UF2; ++u) {
                        if (!((nz >> u) & 1)) continue;
                        const int e = e0 + u * NT;
                        const unsigned long long a = sm.box[e];
                        sm.box[e] = 0ull;
                        int lo_, hi_;
                        magic_decode(a, nst[u], lo_, hi_);
                        vc0[off[u]] = t0[u] + (float)lo_ * inv0;
                        if (two) vc1[off[u]] = t1[u] + (float)hi_ * inv1;
                    }
                }
            } else if (fixedpt && !(p.dbg & 1)) {
                constexpr int UF = 4;
                // (atomic launches: the slots [rlo, rhi + K] per dim, enumerated with a float reciprocal -- exact: the quotients
                // stay half a unit away from the integers)
                const int fl0 = atomic ? sm.rlo[0] : 0, fl1 = atomic ? sm.rlo[1] : 0, fl2 = atomic ? sm.rlo[2] : 0;
                const int fe1 = atomic ? sm.rhi[1] + IP_KS - fl1 + 1 : BOX, fe2 = atomic ? sm.rhi[2] + IP_KS - fl2 + 1 : BOX;
                const int nflush = atomic ? (sm.rhi[0] + IP_KS - fl0 + 1) * fe1 * fe2 : BOXSLOTS;
                const float r2 = 1.f / (float)fe2, r1 = 1.f / (float)fe1;
                for (int e0 = tid; e0 < nflush; e0 += UF * NT) {
                    long long a[UF]; int off[UF]; float sg[UF]; unsigned nst[UF];
#pragma unroll
                    for (int u = 0; u < UF; ++u) {
                        int e = e0 + u * NT;
                        a[u] = 0; off[u] = 0; sg[u] = 1.f; nst[u] = 0u;
                        if (e < nflush) {
                            int xr, yr, zr;
                            if (atomic) {
                                const int t = (int)(((float)e + 0.5f) * r2);
                                zr = e - t * fe2 + fl2;
                                xr = (int)(((float)t + 0.5f) * r1);
                                yr = t - xr * fe1 + fl1; xr += fl0;
                                e = xr * PLANE + yr * PZ + zr;
                            } else {
                                xr = e / PLANE; const int rem = e - xr * PLANE; yr = rem / PZ; zr = rem - yr * PZ;
                            }
                            a[u] = (long long)sm.box[e];
                            sm.box[e] = 0ull;
                            if (!wide) nst[u] = stencils_at(sm, xr, yr, zr);
                            if (atomic) {
                                off[u] = sm.taboff[0][xr] + sm.taboff[1][yr] + sm.taboff[2][zr];
                                sg[u] = sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sm.tabsgn[2][zr];
                            } else {
                                off[u] = (b0[0] + xr) * L.ss[0] + (b0[1] + yr) * L.ss[1] + (b0[2] + zr) * L.ss[2];
                            }
                        }
                    }
                    if (wide) {
                        // one channel: the slot is a 64-bit sum
                        float *vcs = sub ? vc1 : vc0;
                        const float invs = sub ? invw1 : invw0;
#pragma unroll
                        for (int u = 0; u < UF; ++u) {
                            if (a[u] == 0) continue;
                            const float val = (float)((double)a[u] * (double)(invs * sg[u]));
                            if (atomic) __hip_atomic_fetch_add(vcs + off[u], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else vcs[off[u]] += val;
                        }
                    } else if (atomic) {
#pragma unroll
                        for (int u = 0; u < UF; ++u) {
                            if (nst[u] == 0u) continue;
                            int lo_, hi_;
                            magic_decode((unsigned long long)a[u], nst[u], lo_, hi_);
                            if (lo_ != 0) __hip_atomic_fetch_add(vc0 + off[u], (float)lo_ * (inv0 * sg[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (hi_ != 0 && two) __hip_atomic_fetch_add(vc1 + off[u], (float)hi_ * (inv1 * sg[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        float t0[UF], t1[UF];
#pragma unroll
                        for (int u = 0; u < UF; ++u) {
                            t0[u] = nst[u] != 0u ? vc0[off[u]] : 0.f;
                            t1[u] = (nst[u] != 0u && two) ? vc1[off[u]] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < UF; ++u) {
                            if (nst[u] == 0u) continue;
                            int lo_, hi_;
                            magic_decode((unsigned long long)a[u], nst[u], lo_, hi_);
                            vc0[off[u]] = t0[u] + (float)lo_ * inv0;
                            if (two) vc1[off[u]] = t1[u] + (float)hi_ * inv1;
                        }
                    }
                }
            }
            __syncthreads();
            }                                                        // sub
            prof_mark(12);
            if (tid == 0) { sm.cmax[0] = 0; sm.cmax[1] = 0; }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// own_probe (INTERPOL_FLAG_AUTO_SCATTER): which organisation serves this call?
// The sample-stationary tiles (ops_tiled.hip) accumulate a tile of 16^3 samples in an LDS box of at most 33 x 33 x 32
// lattice points centred on the tile's stencils; samples outside it take a slow path, and beyond a few per cent of them the
// kernel degrades by one to two orders of magnitude (config 2, i.i.d. noise: 3.6 / 5.0 / 8.6 / 126 ms at sigma 2 / 3 / 4 / 6,
// where this file needs 4.1 - 4.4 ms throughout).  128 tiles spread over the sample grid are examined exactly as
// Box::build does it: the workgroup that finishes last writes gate = 1 (owner-computes) when more than 0.4 % of the
// probed samples fall outside their tile's box -- and hardly any outside the binned range, where this file would fall
// back to per-sample atomics -- else 0 (tiles).  Every scatter kernel of the call reads the word on entry: one of the two
// organisations returns at once.  Stateless and deterministic: the choice depends on the coordinates of this call alone.
// ---------------------------------------------------------------------------
static __global__ void own_zero(int *__restrict__ p, int n)      // (static: this file is two translation units, see IP_OWNER_MIX_TU)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) p[i] = 0;
}


// ---------------------------------------------------------------------------
// own_gather -- the owner-computes PULL (grid_pull, reference interpol/nd.py:80-143) for deformations too rough for the sample
// tiles of ops_sorted.hip, whose LDS box (32 x 32 x 36 lattice points around 16^3 samples) holds the stencils of i.i.d.
// displacements up to sigma ~ 3 voxels only (sigma = 6: 11 ms at config 2, as slow as the generic kernel).  own_bin (index
// mode) sorts the samples by the brick of their first tap; here a workgroup draws a non-empty brick from the list, stages
// the brick's 19^3 lattice points ONCE per channel pair (1.7 lattice points per sample, whatever the deformation) and gathers
// the brick's records from it: weights once, 64 LDS reads at immediate offsets, 84 packed FMAs, two scattered 4-byte stores.
// Cost independent of the deformation.  Every boundary condition: a box slot is a lattice point through the tables.
// ---------------------------------------------------------------------------
constexpr int GPZ = 20;                         // row pitch of the gather's box (8-byte slots)
constexpr int GPLANE = BOX * GPZ;
struct GatSmem {
    int   taboff[3][BOX + 1];
    float tabsgn[3][BOX + 1];
    unsigned start[CAPD];
    int   pref[CAPD + 2];                      // records in front of each run of the brick; [nd ...]: all of them
    int   brick, pad;
    float2 box[BOX * GPLANE];                  // 57 760 B
};
#define IP_RD(o, off) "ds_read_b64 %" #o ", %16 offset:" #off "\n\t"
__device__ __forceinline__ void gather_reads(unsigned addr, f2 (&v)[16])
{
    static_assert(GPZ * 8 == 160, "the immediate offsets are (row * GPZ + k) * 8");
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16) IP_RD(3, 24)
                 IP_RD(4, 160) IP_RD(5, 168) IP_RD(6, 176) IP_RD(7, 184)
                 IP_RD(8, 320) IP_RD(9, 328) IP_RD(10, 336) IP_RD(11, 344)
                 IP_RD(12, 480) IP_RD(13, 488) IP_RD(14, 496) IP_RD(15, 504)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : "v"(addr) : "memory");
}
#undef IP_RD

// GRAD: the grid gradient of the pull instead (pushpull.py:256-257) -- per tap the channel pair contracted with grad_out (`gout`, two
// scattered 4-byte loads per sample and pair), three derivative sums, `out` = the dense (B, *out, 3) grid gradient: written by the
// first pair, accumulated by the following ones (the same thread owns the sample in every pair).
// GRAD == 2: grid_grad (nd.py:216-288) -- val[b,c,o,:] = mask * the three derivative sums of each channel, the pair packed.
template <int K, int GRAD = 0>
__global__ __launch_bounds__(NT, 4) void own_gather(KParams p, BrickGrid bg, const int *__restrict__ ndesc, const uint2 *__restrict__ desc,
                                                    const float4 *__restrict__ rec, const int *__restrict__ list, int *__restrict__ draw,
                                                    const float *__restrict__ img, float *__restrict__ out, const int *__restrict__ gate,
                                                    const float *__restrict__ gout)
{
    if (gate && *gate != 1) return;                                  // the probe chose the sample tiles
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
        const int b0[3] = { brick_origin(bx, bg.lo[0], bg.top[0], bg.nin[0], bg.split[0]), brick_origin(by, bg.lo[1], bg.top[1], bg.nin[1], bg.split[1]),
                            brick_origin(bz, bg.lo[2], bg.top[2], bg.nin[2], bg.split[2]) };      // lattice index of box slot 0
        const int nd = min(ndesc[bk], CAPD);
        if (tid >= 256 && tid < 320) {
            // the runs of the brick and the exclusive prefix of their lengths: the threads walk the brick's records as ONE list (a run
            // holds ~150 records at sigma = 2, ~30 at sigma = 6: a wave per run left half of its lanes idle)
            static_assert(CAPD == 128, "two runs per lane");
            const int ln = tid - 256, e0 = 2 * ln, e1 = e0 + 1;
            const uint2 d0 = e0 < nd ? desc[(int64_t)bk * CAPD + e0] : make_uint2(0u, 0u), d1 = e1 < nd ? desc[(int64_t)bk * CAPD + e1] : make_uint2(0u, 0u);
            const int sum = (int)d0.y + (int)d1.y;
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (ln >= o) incl += t; }
            sm.start[e0] = d0.x; sm.start[e1] = d1.x;
            sm.pref[e0] = incl - sum; sm.pref[e1] = incl - (int)d1.y;
            if (ln == 63) { sm.pref[CAPD] = incl; sm.pref[CAPD + 1] = 0x7fffffff; }
        }
        if (tid < 3 * 64) {                                          // box slot -> wrapped lattice offset and sign (bounds.py:30-89)
            const int d = tid >> 6, slot = tid & 63;
            if (slot < BOX) {
                const long long pk = wrap_outofline(p.bound[d], (d == 0 ? b0[0] : d == 1 ? b0[1] : b0[2]) + slot, p.vol_n[d]);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] / 4);
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        for (int c = 0; c < p.C; c += 2) {
            const bool two = c + 1 < p.C;
            const float *vc0 = img + b * p.vol_sb + (int64_t)c * p.vol_sc;
            const float *vc1 = two ? vc0 + p.vol_sc : vc0;
            float *oc0 = out + b * p.val_sb + (int64_t)c * p.val_sc;
            __syncthreads();                                         // tables written / the previous pair's readers are done
            // rows of the box that are contiguous runs of the image's unit-stride dim with sign +1 (the box's z-range inside the
            // lattice; dst1: sign 0 at index 0) move as QUADS -- five per row, the last one shifted to end at slot 18 -- of 16-byte
            // loads (narrow loads are what the vector L1 is slow at: 0.5 -> 0.2 ms at config 2); else slot by slot through the z table
            const bool zlin = p.vol_ss[2] == 4 && b0[2] >= (p.bound[2] == B_DST1 ? 1 : 0) && b0[2] + BOX <= p.vol_n[2];
            if (p.dbg & 1) {                                         // (ablation: no staging)
            } else if (zlin) {
                for (int e = tid; e < BOX * BOX * 5; e += NT) {
                    const int row = e / 5, q = e - row * 5;
                    const int x = row / BOX, y = row - x * BOX;
                    const int zs = q < 4 ? 4 * q : BOX - 4;
                    const int off = sm.taboff[0][x] + sm.taboff[1][y] + b0[2] + zs;
                    const float sg = sm.tabsgn[0][x] * sm.tabsgn[1][y];
                    const float4 a0 = ld4<float>(vc0 + off), a1 = ld4<float>(vc1 + off);
                    float2 *dst = sm.box + row * GPZ + zs;
                    if (q < 4) {
                        reinterpret_cast<float4 *>(dst)[0] = make_float4(a0.x * sg, a1.x * sg, a0.y * sg, a1.y * sg);
                        reinterpret_cast<float4 *>(dst)[1] = make_float4(a0.z * sg, a1.z * sg, a0.w * sg, a1.w * sg);
                    } else {                                         // slots 15 .. 18: 8-byte aligned only
                        dst[0] = make_float2(a0.x * sg, a1.x * sg); dst[1] = make_float2(a0.y * sg, a1.y * sg);
                        dst[2] = make_float2(a0.z * sg, a1.z * sg); dst[3] = make_float2(a0.w * sg, a1.w * sg);
                    }
                }
            } else {
                for (int e = tid; e < BOX * BOX * BOX; e += NT) {
                    const int x = e / (BOX * BOX), y = (e / BOX) % BOX, z = e % BOX;
                    const int off = sm.taboff[0][x] + sm.taboff[1][y] + sm.taboff[2][z];
                    const float sg = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sm.tabsgn[2][z];
                    sm.box[(x * BOX + y) * GPZ + z] = make_float2(vc0[off] * sg, vc1[off] * sg);
                }
            }
            __syncthreads();
            const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
            const int ntot = sm.pref[CAPD];
            int rr = 0;
            for (int j = tid; j < ntot; j += NT) {
                {
                    while (j >= sm.pref[rr + 1]) ++rr;               // (runs beyond the last hold nothing: their prefix is the total)
                    const float4 rc = rec[sm.start[rr] + (unsigned)(j - sm.pref[rr])];
                    const float fx = floorf(rc.x - 0.5f * (float)(kd<K>(p, 0) - 1)), fy = floorf(rc.y - 0.5f * (float)(kd<K>(p, 1) - 1)), fz = floorf(rc.z - 0.5f * (float)(kd<K>(p, 2) - 1));
                    const float tx = rc.x - fx; const f2 tyz = f2{ rc.y - fy, rc.z - fz };
                    // first-tap cell inside the brick: 0 .. 15 by construction of the bins (own_bin); clamped, should a coordinate be off
                    int cx = __float2int_rz(fx) - b0[0], cy = __float2int_rz(fy) - b0[1], cz = __float2int_rz(fz) - b0[2];
                    cx = max(0, min(cx, BR - 1)); cy = max(0, min(cy, BR - 1)); cz = max(0, min(cz, BR - 1));
                    const unsigned addr = boxaddr + (unsigned)((cx * BOX + cy) * GPZ + cz) * 8u;
                    float wx[4];
                    { // the K + 1 weights of the x-stencil (scalar form of weights_yz; splines.py:30-80)
                        if constexpr (K == KMIX) weights4(p.order[0], tx, wx);
                        else if (K == 3) { const float u = tx - 1.f, v = 2.f - tx, u2 = u * u, v2 = v * v;
                                      wx[0] = (v2 * v) * (1.f / 6.f); wx[3] = (u2 * u) * (1.f / 6.f); wx[1] = u2 * (u * 0.5f - 1.f) + 2.f / 3.f; wx[2] = v2 * (v * 0.5f - 1.f) + 2.f / 3.f; }
                        else { const float a = 1.5f - tx, cc = tx - 0.5f, m = tx - 1.f; wx[0] = (a * a) * 0.5f; wx[1] = 0.75f - m * m; wx[2] = (cc * cc) * 0.5f; wx[3] = 0.f; }
                    }
                    f2 w[4];
                    if constexpr (K == KMIX) mixed_weights_yz(p.order[1], p.order[2], tyz, w);
                    else weights_yz<K>(tyz, w);
                    if (GRAD == 2) {
                        f2 dq[4];
                        if constexpr (K == KMIX) mixed_wgrads_yz(p.order[1], p.order[2], tyz, dq);
                        else wgrads_yz<K>(tyz, dq);
                        f2 g0 = { 0.f, 0.f }, g1 = { 0.f, 0.f }, g2 = { 0.f, 0.f };
#pragma unroll
                        for (int ii = 0; ii <= KSV<K>; ++ii) {
                            if (K == KMIX && ii > p.order[0]) continue;                  // (mixed orders: no plane, no slot beyond a dim's own stencil)
                            f2 t2[16];
                            gather_reads(addr + (unsigned)(ii * GPLANE * 8), t2);
                            if constexpr (K == KMIX) clear_unused_taps(p.order[1], p.order[2], t2);
                            f2 pp = { 0.f, 0.f }, ppy = { 0.f, 0.f }, ppz = { 0.f, 0.f };
#pragma unroll
                            for (int jy = 0; jy <= KSV<K>; ++jy) {
                                f2 q = { 0.f, 0.f }, qz = { 0.f, 0.f };
#pragma unroll
                                for (int k = 0; k <= KSV<K>; ++k) {
                                    q = f2{ w[k].y, w[k].y } * t2[4 * jy + k] + q;
                                    qz = f2{ dq[k].y, dq[k].y } * t2[4 * jy + k] + qz;
                                }
                                pp = f2{ w[jy].x, w[jy].x } * q + pp;
                                ppy = f2{ dq[jy].x, dq[jy].x } * q + ppy;
                                ppz = f2{ w[jy].x, w[jy].x } * qz + ppz;
                            }
                            const float wxi = own_weight_x<K>(p, tx, ii), gxi = own_wgrad_x<K>(p, tx, ii);
                            g0 = f2{ gxi, gxi } * pp + g0;
                            g1 = f2{ wxi, wxi } * ppy + g1;
                            g2 = f2{ wxi, wxi } * ppz + g2;
                            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2));      // (one x-plane at a time)
                        }
                        const float xyz[3] = { rc.x, rc.y, rc.z };
                        const float m = inb_mask(p, xyz);            // nd.py:286-287
                        float *dst = oc0 + 3 * (int64_t)__float_as_int(rc.w);
                        dst[0] = g0.x * m; dst[1] = g1.x * m; dst[2] = g2.x * m;
                        if (two) { dst[p.val_sc] = g0.y * m; dst[p.val_sc + 1] = g1.y * m; dst[p.val_sc + 2] = g2.y * m; }
                        continue;
                    }
                    if (GRAD == 1) {
                        const int64_t o = (int64_t)__float_as_int(rc.w);
                        const float *gc0 = gout ? gout + b * p.val_sb + (int64_t)c * p.val_sc : nullptr;
                        const float go0 = gc0 ? gc0[o] : 1.f, go1 = two ? (gc0 ? gc0[p.val_sc + o] : 1.f) : 0.f;   // (no grad_out: ones -- the backward of count)
                        f2 dq[4];
                        if constexpr (K == KMIX) mixed_wgrads_yz(p.order[1], p.order[2], tyz, dq);
                        else wgrads_yz<K>(tyz, dq);
                        float ag0 = 0.f, ag1 = 0.f, ag2 = 0.f;
#pragma unroll
                        for (int ii = 0; ii <= KSV<K>; ++ii) {
                            if (K == KMIX && ii > p.order[0]) continue;                  // (mixed orders: no plane, no slot beyond a dim's own stencil)
                            f2 t2[16];
                            gather_reads(addr + (unsigned)(ii * GPLANE * 8), t2);
                            if constexpr (K == KMIX) clear_unused_taps(p.order[1], p.order[2], t2);
                            float pp = 0.f, ppy = 0.f, ppz = 0.f;
#pragma unroll
                            for (int jy = 0; jy <= KSV<K>; ++jy) {
                                float q = 0.f, qz = 0.f;
#pragma unroll
                                for (int k = 0; k <= KSV<K>; ++k) {
                                    const float sgl = __builtin_fmaf(go1, t2[4 * jy + k].y, go0 * t2[4 * jy + k].x);
                                    q = __builtin_fmaf(w[k].y, sgl, q);
                                    qz = __builtin_fmaf(dq[k].y, sgl, qz);
                                }
                                pp = __builtin_fmaf(w[jy].x, q, pp);
                                ppy = __builtin_fmaf(dq[jy].x, q, ppy);
                                ppz = __builtin_fmaf(w[jy].x, qz, ppz);
                            }
                            const float wxi = own_weight_x<K>(p, tx, ii);
                            ag0 = __builtin_fmaf(own_wgrad_x<K>(p, tx, ii), pp, ag0);
                            ag1 = __builtin_fmaf(wxi, ppy, ag1);
                            ag2 = __builtin_fmaf(wxi, ppz, ag2);
                            asm volatile("" : "+v"(ag0), "+v"(ag1), "+v"(ag2));      // (one x-plane at a time: the planes' 16 reads each interleaved spill)
                        }
                        const float xyz[3] = { rc.x, rc.y, rc.z };
                        const float m = inb_mask(p, xyz);            // nd.py:139-140
                        float *dst = out + (b * p.N + o) * 3;
                        if (c == 0) { dst[0] = ag0 * m; dst[1] = ag1 * m; dst[2] = ag2 * m; }
                        else { dst[0] += ag0 * m; dst[1] += ag1 * m; dst[2] += ag2 * m; }
                        continue;
                    }
                    f2 a = { 0.f, 0.f };
#pragma unroll
                    for (int ii = 0; ii <= KSV<K>; ++ii) {
                        if (p.dbg & 2) { a = f2{ w[0].x + wx[ii], w[1].y }; break; }      // (ablation: no taps)
                        if (K == KMIX && ii > p.order[0]) continue;                      // (mixed orders: no plane, no slot beyond a dim's own stencil)
                        f2 t2[16];
                        gather_reads(addr + (unsigned)(ii * GPLANE * 8), t2);
                        if constexpr (K == KMIX) clear_unused_taps(p.order[1], p.order[2], t2);
                        f2 pp = { 0.f, 0.f };
#pragma unroll
                        for (int jy = 0; jy <= KSV<K>; ++jy) {
                            f2 q = { 0.f, 0.f };
#pragma unroll
                            for (int k = 0; k <= KSV<K>; ++k) q = f2{ w[k].y, w[k].y } * t2[4 * jy + k] + q;
                            pp = f2{ w[jy].x, w[jy].x } * q + pp;
                        }
                        a = f2{ wx[ii], wx[ii] } * pp + a;
                    }
                    const float xyz[3] = { rc.x, rc.y, rc.z };
                    const float m = inb_mask(p, xyz);                // nd.py:139-140
                    const int64_t o = (int64_t)__float_as_int(rc.w);
                    if ((p.dbg & 4) && a.x != 12345.f) continue;     // (ablation: no stores)
                    oc0[o] = a.x * m;
                    if (two) oc0[p.val_sc + o] = a.y * m;
                }
            }
        }
    }
}

constexpr int NPROBE = 128;
constexpr int BOXVOL = 14500;
constexpr int DENSEVOL = 13000;
constexpr int PULLOUT = 15;                     // own_probe, pull: per mille of the probed samples outside their tile's box beyond which the bricks take the call
struct ProbeHdr { int gate, done, nslow, nfar, nvalid, nbox, nfull, ncorner; };

template <int K, int GM>
__global__ __launch_bounds__(NT1) void own_probe(KParams p, BrickGrid bg, const float *__restrict__ grid, ProbeHdr *__restrict__ hdr,
                                                 int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch, int nch)
{
    __shared__ int lo[3], hi[3], cnt[3], clo[3], chi[3];
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)ntiles * nbatch;
    const int64_t work = (int64_t)blockIdx.x * total / gridDim.x;
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom((int)(work % ntiles), gx, gy, gz, nty, ntz);
    if (tid < 3) { lo[tid] = 0x7fffffff; hi[tid] = -0x7fffffff; cnt[tid] = 0; clo[tid] = 0x7fffffff; chi[tid] = -0x7fffffff; }
    float fl[VPT1][3];
    unsigned valid = 0, corner = 0;
    int far = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        int ox, oy, oz; float c[3];
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        if (ox < gx && oy < gy && oz < gz) valid |= 1u << v;
        // the eight corner samples of the tile: their bounding box is what a smooth deformation makes of the tile
        if (((ox - g.ox0) % (TS - 1)) == 0 && ((oy - g.oy0) % (TS - 1)) == 0 && ((oz - g.oz0) % (TS - 1)) == 0) corner |= 1u << v;
        ox = ox < gx ? ox : gx - 1; oy = oy < gy ? oy : gy - 1; oz = oz < gz ? oz : gz - 1;
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, c);
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            fl[v][d] = floorf(c[d] - 0.5f * (float)(K - 1));
            in = in && fl[v][d] >= (float)(bg.lo[d] - OFF) && fl[v][d] < (float)(bg.top[d] + NHI * BR);
            fl[v][d] = __builtin_fmaxf(__builtin_fminf(fl[v][d], 1073741824.f), -1073741824.f);
        }
        if (((valid >> v) & 1) && !in) ++far;
    }
    __syncthreads();
    int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (!((valid >> v) & 1)) continue;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int i = fl[v][d] == fl[v][d] ? __float2int_rz(fl[v][d]) : 0;
            mn[d] = i < mn[d] ? i : mn[d]; mx[d] = i > mx[d] ? i : mx[d];
            if ((corner >> v) & 1) { atomicMin(&clo[d], i); atomicMax(&chi[d], i); }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = wave_min(mn[d]), e = wave_max(mx[d]);
        if ((tid & 63) == 0) { atomicMin(&lo[d], a); atomicMax(&hi[d], e); }
    }
    __syncthreads();
    // the tile's box as the tiles cut it (ops_tiled.hip: Box::build): centred, at most 33 x 33 x 32 lattice points
    const int cap[3] = { 33, 33, 32 };
    int l[3], h[3], boxvol = 1, cornervol = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int a = lo[d], sz = hi[d] + K - a + 1;
        if (sz > cap[d]) { a += (sz - cap[d]) / 2; sz = cap[d]; }
        l[d] = a; h[d] = a + sz - K - 1;
        boxvol *= sz > 0 ? sz : 0;
        const int cs = chi[d] + K - clo[d] + 1;
        cornervol *= cs > 0 ? (cs < cap[d] ? cs : cap[d]) : 0;
    }
    int slow = 0, nv = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (!((valid >> v) & 1)) continue;
        ++nv;
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int i = fl[v][d] == fl[v][d] ? __float2int_rz(fl[v][d]) : 0x7fffffff;
            in = in && i >= l[d] && i <= h[d];
        }
        if (!in) ++slow;
    }
    // (one LDS add per wave: 3 x 512 adds to the same three words took a third of the kernel)
    slow = wave_sum(slow); far = wave_sum(far); nv = wave_sum(nv);
    if ((tid & 63) == 0) { atomicAdd(&cnt[0], slow); atomicAdd(&cnt[1], far); atomicAdd(&cnt[2], nv); }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&hdr->nslow, cnt[0]); atomicAdd(&hdr->nfar, cnt[1]); atomicAdd(&hdr->nvalid, cnt[2]);
        if (cnt[2] == NS) { atomicAdd(&hdr->nbox, boxvol); atomicAdd(&hdr->nfull, 1); atomicAdd(&hdr->ncorner, cornervol); }   // (whole tiles: their boxes are comparable)
        __threadfence();
        if (atomicAdd(&hdr->done, 1) == (int)gridDim.x - 1) {
            const int ns = atomicAdd(&hdr->nslow, 0), nf = atomicAdd(&hdr->nfar, 0), nn = atomicAdd(&hdr->nvalid, 0);
            const int nb = atomicAdd(&hdr->nbox, 0), nt = atomicAdd(&hdr->nfull, 0), nc = atomicAdd(&hdr->ncorner, 0);
            // owner-computes when the tiles would leave samples outside their boxes, or -- two channels and more -- when the boxes
            // they flush with global atomics are large (mean above BOXVOL lattice points: i.i.d. noise of sigma ~ 0.9 voxels on a
            // 16^3 tile; the tiles then need 2.9 ms and more at config 2, this file 2.9 - 3.0; a single channel flushes half as
            // much and stays with the tiles: count 2.1 against 2.5 ms) AND it is roughness that makes them large -- they hold 1.6
            // times the boxes of the tiles' eight corner samples and more; an expanding smooth field has large boxes that hardly
            // overlap, the tiles flush little more than the target then (zoom 1.5: tiles 4.0, this file 5.5 ms) -- unless samples
            // lie outside the binned range
            const bool rough = nch > 1 && (int64_t)nb > (int64_t)BOXVOL * nt && (int64_t)nb * 10 > (int64_t)nc * 16;
            // nch < 0, the grid gradient of the pull (own_gather<K, true>): the bricks of the image beat the sample tiles on DENSE
            // samplings whatever their roughness (config 2: identity 1.9 against 2.2 ms, sigma = 2 2.06 / 2.47, sigma = 6 2.8 / 20) and
            // lose on expanding ones, whose samples spread over more bricks (zoom 1.5: 4.1 / 2.3 ms) -- dense: the eight corner samples
            // of a tile span at most DENSEVOL lattice points on average (19^3 at the identity, 27^3 at zoom 1.5), or the tiles leave
            // samples outside their boxes
            const bool dense = (int64_t)nc <= (int64_t)DENSEVOL * nt;
            // nch == -2, the pull itself (round 5): the sample tiles win while their LDS boxes hold the stencils (identity 1.02 against
            // 1.39 ms, i.i.d. sigma = 3 -- 0.6 % of the probed samples outside their tile's box -- 1.58 / 1.75) and fall behind quickly
            // beyond (every sample outside costs a wave: 3.1 % outside -- sigma = 4, a folding smooth field of amplitude 8 -- 2.4 / 1.8
            // and 5.2 / 1.6 ms; sigma = 6, 15 %: 9.7 / 1.9; zoom 2, 23 %: 7.5 / 3.6): the bricks when more than PULLOUT per mille of the
            // probed samples lie outside -- BEFORE the tiles pay their coordinate loads and sort (the per-tile hand-over of round 4,
            // which remains behind it, came after: 2.8 ms at sigma = 6, 2.4 on the folding field)
            if (nch == -2) hdr->gate = ((int64_t)ns * 1000 > (int64_t)nn * PULLOUT && (int64_t)nf * 64 <= nn) ? 1 : 0;
            else if (nch < 0) hdr->gate = (((int64_t)ns * 250 > nn || dense) && (int64_t)nf * 64 <= nn) ? 1 : 0;
            else hdr->gate = (((int64_t)ns * 250 > nn || rough) && (int64_t)nf * 64 <= nn) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct Workspace {
    int *flags;
    ProbeHdr *hdr; int *ndesc; int *bmax; uint2 *desc; float4 *rec; float *vals; unsigned short *meta;
    int64_t nrec; int nbricks;
};
static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// shared: the batch items scatter into ONE target -- they share its bricks (BrickGrid::item = 0) and a brick takes CAPX runs
static int64_t layout(const KParams &k, int B, int ntiles, int nch, void *base, Workspace *w, int64_t nflags = 0, bool shared = false)
{
    const BrickGrid bg = brick_grid(k);
    const int64_t nbricks = (int64_t)bg.per_item * (shared ? 1 : B);
    const int64_t nrec = (int64_t)ntiles * NS * B;
    int64_t o = 0;
    unsigned char *p = (unsigned char *)base;
    const int64_t o_hdr = o; o += 256;                               // header and brick counters are zeroed by ONE memset
    const int64_t o_nd = o; o += nbricks * 4;                        // (header, brick counters and brick maxima: ONE zero-fill)
    const int64_t o_bm = o; o += nbricks * 8;
    const int64_t o_fl = o; o += nflags * 4; o = align256(o);        // (pull: one flag per sample tile; what follows holds 8- and 16-byte elements: aligned)
    const int64_t o_desc = o; o += align256(nbricks * (shared ? CAPX : CAPD) * 8);
    const int64_t o_rec = o; o += align256(nrec * 16);
    const int64_t o_val = o; o += align256(nrec * 4 * (nch > 1 ? nch - 1 : 0));
    const int64_t o_meta = o; o += align256(nrec * 2);
    if (w) {
        w->hdr = (ProbeHdr *)(p + o_hdr); w->ndesc = (int *)(p + o_nd); w->desc = (uint2 *)(p + o_desc);
        w->rec = (float4 *)(p + o_rec); w->vals = (float *)(p + o_val); w->meta = (unsigned short *)(p + o_meta);
        w->bmax = (int *)(p + o_bm); w->flags = (int *)(p + o_fl);
        w->nrec = nrec; w->nbricks = (int)nbricks;
    }
    return o;
}

static int tile_count(const interpol_problem *p)
{
    int64_t n = 1;
    for (int d = 0; d < 3; ++d) n *= (p->grid_shape[d] + TS - 1) / TS;
    return n > 0x7fffffff ? 0 : (int)n;
}

// Mixed orders 1..3: the K = KMIX instantiations and their launchers live in a translation unit of their own (this file compiled with
// -DIP_OWNER_MIX_TU, as ops_sorted.hip's mixed kernels: the isotropic kernels of this module keep their code to the byte).
// dtype: of the sources (own_bin, idx == 0); idx 1 / 2 / 3: the pull / the grid gradient of its backward / grid_grad (float).
int mix_launch_bin(int dtype, int idx, const KParams &k, const BrickGrid &bg, const Workspace &w, const void *val, const void *grid, void *vol,
                   const int *gate, const void *aux, const int *all, int gx, int gy, int gz, int ntiles, int B, hipStream_t st);
int mix_launch_acc(const KParams &k, const BrickGrid &bg, const Workspace &w, void *vol, int nch, int color, int Bw, const int *gate,
                   unsigned nblocks, hipStream_t st);
int mix_launch_gather(int grad, const KParams &k, const BrickGrid &bg, const Workspace &w, const void *vol, void *val, const int *gate,
                      const void *gout, unsigned nblocks, hipStream_t st);

#ifdef IP_OWNER_MIX_TU
int mix_launch_bin(int dtype, int idx, const KParams &k, const BrickGrid &bg, const Workspace &w, const void *val, const void *grid, void *vol,
                   const int *gate, const void *aux, const int *all, int gx, int gy, int gz, int ntiles, int B, hipStream_t st)
{
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS;
    const dim3 tgrid((unsigned)(ntiles * B));
#define IP_MIX_BIN(T, IDX)                                                                                              \
    {                                                                                                                   \
        const int attr = big_lds<own_bin<T, KMIX, 0, IDX>>(sizeof(BinSmem));                                            \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((own_bin<T, KMIX, 0, IDX>), tgrid, dim3(NT1), sizeof(BinSmem), st, k, bg, (const T *)val, (const float *)grid, \
                           (float *)vol, w.ndesc, w.desc, w.rec, w.vals, w.meta, w.bmax, w.nrec, gx, gy, gz, nty, ntz, ntiles, gate, (const T *)aux, all); \
    }
    if (idx == 0) { if (dtype == INTERPOL_F32) IP_MIX_BIN(float, 0) else if (dtype == INTERPOL_BF16) IP_MIX_BIN(bf16_t, 0) else if (dtype == INTERPOL_F16) IP_MIX_BIN(f16_t, 0) else return INTERPOL_E_DTYPE; }
    else if (idx == 1) IP_MIX_BIN(float, 1) else if (idx == 2) IP_MIX_BIN(float, 2) else IP_MIX_BIN(float, 3)
#undef IP_MIX_BIN
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int mix_launch_acc(const KParams &k, const BrickGrid &bg, const Workspace &w, void *vol, int nch, int color, int Bw, const int *gate,
                   unsigned nblocks, hipStream_t st)
{
    const int attr = big_lds<own_accumulate<KMIX>>(sizeof(AccSmem));
    if (attr) return attr;
    hipLaunchKernelGGL((own_accumulate<KMIX>), dim3(nblocks), dim3(NT), sizeof(AccSmem), st, k, bg, (const int *)w.ndesc,
                       (const uint2 *)w.desc, (const float4 *)w.rec, (const float *)w.vals, (const unsigned short *)w.meta,
                       (const int *)w.bmax, w.nrec, (float *)vol, nch, color, Bw, gate, (int *)w.hdr + 16 + color);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
int mix_launch_gather(int grad, const KParams &k, const BrickGrid &bg, const Workspace &w, const void *vol, void *val, const int *gate,
                      const void *gout, unsigned nblocks, hipStream_t st)
{
#define IP_MIX_GAT(GR)                                                                                                  \
    {                                                                                                                   \
        const int attr = big_lds<own_gather<KMIX, GR>>(sizeof(GatSmem));                                                \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((own_gather<KMIX, GR>), dim3(nblocks), dim3(NT), sizeof(GatSmem), st, k, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.bmax, (int *)w.hdr + 40, (const float *)vol, (float *)val, gate, (const float *)gout); \
    }
    if (grad == 1) IP_MIX_GAT(1) else if (grad == 2) IP_MIX_GAT(2) else IP_MIX_GAT(0)
#undef IP_MIX_GAT
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
#endif

} // namespace owner

#ifndef IP_OWNER_MIX_TU

// Eligible: 3-D, one order 2..3, sample grid about as dense as the target (else the tiled / brick
// scatters are the better organisation), sizes within 32-bit record counts.
static bool shared_target(const interpol_problem *p) { return p->vol_stride[0] == 0 && p->batch > 1; }
static bool mixed_orders(const KParams &k) { return k.order[0] != k.order[1] || k.order[0] != k.order[2]; }
static bool owner_eligible(const interpol_problem *p, const KParams &k, bool scatter = false)
{
    const bool shared = scatter && shared_target(p);
    if (p->dim != 3 || p->batch > 4096) return false;
    if (!(p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER))) return false;   // see interpol_hip.h
    // (round 5: the trilinear push / count take the organisation as well; the gathers' bricks stay with orders 2 - 3)
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2]) {
        // mixed orders 1 .. 3 (round 6): the cubic's organisation with the dims' own first taps and weights (K = KMIX), dense grids
        for (int d = 0; d < 3; ++d) if (k.order[d] < 1 || k.order[d] > 3) return false;
        if (k.sep != 0) return false;
    } else if (k.order[0] < (scatter ? 1 : 2) || k.order[0] > 3) return false;
    int64_t n = 1, nv = 1, nb = shared ? 1 : p->batch;
    for (int d = 0; d < 3; ++d) {
        n *= p->grid_shape[d]; nv *= p->vol_shape[d];
        nb *= owner::NLO + owner::NHI + (p->vol_shape[d] + 15 + owner::BR - 1) / owner::BR;   // (an upper bound of BrickGrid::nb)
        if (p->grid_shape[d] > 0x7fffffff / 4) return false;
    }
    const int64_t nt = owner::tile_count(p);
    if (n < 4096 || nt == 0 || nt * owner::NS * p->batch > 0x7fffffffll || nb > 0x7fffffffll / (shared ? owner::CAPX : owner::CAPD)) return false;
    if ((uint64_t)n * 12ull > 0xffffffffull) return false;
    // at least a quarter of a sample per target voxel -- all the items' samples count when they share the target (round 5:
    // BASELINE config 4, 64 sources of 128^3 into 512^3, is as dense as config 2 once the sources share the bricks)
    return (shared ? 8 * n * p->batch : 4 * n) >= nv;
}

// Nearest-neighbour push / count (3-D, all orders 0; round 5): the generic kernel's one global atomic per sample and channel costs 5.7 ms
// for 4 x 2 x 256^3 whatever the field; as a trilinear scatter of the ROUNDED coordinates (own_bin) the bricks take 2.3 - 3 ms.  The
// organisation sees orders 1 (lattice bricks, stencil counts, flush), KParams::mode stays MODE_ISO0 and tells own_bin to round.
int linear_pull_probe(const interpol_problem *p, const KParams &k, const void *grid, void *workspace, hipStream_t st, const int **gate_out, int thr16);
constexpr int NEAREST_THR16 = 24;
static bool nearest_scatter(const KParams &k) { return k.dim == 3 && k.mode == MODE_ISO0 && k.order[0] == 0 && k.order[1] == 0 && k.order[2] == 0; }
static KParams nearest_as_trilinear(const KParams &k)
{
    KParams q = k;
    if (nearest_scatter(k)) { q.order[0] = 1; q.order[1] = 1; q.order[2] = 1; }
    return q;
}

// bytes of workspace the owner-computes organisation needs for this problem (0: not applicable)
int64_t owner_workspace_bytes(const interpol_problem *p, const KParams &k_in, bool count_only)
{
    const KParams k = nearest_as_trilinear(k_in);
    if (!owner_eligible(p, k, true)) return 0;
    const int nch = count_only ? 1 : k.C + (k.cc ? 1 : 0);
    return owner::layout(k, (int)p->batch, owner::tile_count(p), nch, nullptr, nullptr, 0, shared_target(p));
}

namespace owner {
template <typename T, int IDX = 0>
static int launch_bin(const interpol_problem *p, const KParams &k, const BrickGrid &bg, const Workspace &w, const void *val, const void *grid,
                      void *vol, const int *gate, hipStream_t st, const void *aux = nullptr, const int *all = nullptr)
{
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS;
    const int ntiles = tile_count(p);
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2])
        return mix_launch_bin(std::is_same<T, float>::value ? INTERPOL_F32 : (std::is_same<T, bf16_t>::value ? INTERPOL_BF16 : INTERPOL_F16), IDX, k, bg, w,
                              val, grid, vol, gate, aux, all, gx, gy, gz, ntiles, (int)p->batch, st);
#define IP_OWN_BIN(KK, GM)                                                                                              \
    {                                                                                                                   \
        const int attr = big_lds<own_bin<T, KK, GM, IDX>>(sizeof(BinSmem));                                               \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((own_bin<T, KK, GM, IDX>), tgrid, dim3(NT1), sizeof(BinSmem), st, k, bg, (const T *)val, (const float *)grid, \
                           (float *)vol, w.ndesc, w.desc, w.rec, w.vals, w.meta, w.bmax, w.nrec, gx, gy, gz, nty, ntz, ntiles, gate, (const T *)aux, all); \
    }
#define IP_OWN_BY_GM(KK)                                                                                                \
    { if (k.sep == 0) IP_OWN_BIN(KK, 0) else if (k.sep == 1) IP_OWN_BIN(KK, 1) else if (k.sep == 2) IP_OWN_BIN(KK, 2) else IP_OWN_BIN(KK, 3) }
    if (k.order[0] == 3) IP_OWN_BY_GM(3) else if (k.order[0] == 2) IP_OWN_BY_GM(2) else IP_OWN_BY_GM(1)
#undef IP_OWN_BY_GM
#undef IP_OWN_BIN
    return 0;
}
} // namespace owner

// returns 1 when it took the problem, 2 when it launched itself GATED behind the roughness probe (INTERPOL_FLAG_AUTO_SCATTER:
// the caller launches the tiled / generic scatter as well, with KParams::gate = *gate_out), 0 to decline (workspace missing /
// not eligible), else an error
int try_owner_push(const interpol_problem *p, const KParams &k_in, const void *val, const void *grid, void *vol,
                   void *workspace, int64_t workspace_bytes, hipStream_t st, const int **gate_out)
{
    using namespace owner;
    const bool nearest = nearest_scatter(k_in);
    const KParams k = nearest_as_trilinear(k_in);
    if (!workspace || !owner_eligible(p, k, true)) return 0;
    const bool count_only = val == nullptr;
    const int nch = count_only ? 1 : k.C + (k.cc ? 1 : 0);
    Workspace w;
    if (((uintptr_t)workspace & 255u) != 0) return 0;                // (interpol_hip.h: 256-byte aligned, or the other scatters run)
    const bool shared = shared_target(p);
    if (layout(k, (int)p->batch, tile_count(p), nch, workspace, &w, 0, shared) > workspace_bytes) return 0;
    BrickGrid bg = brick_grid(k);
    if (shared) { bg.item = 0; bg.capd = CAPX; }                     // the items' samples meet in the same bricks
    const bool gated = !(p->flags & INTERPOL_FLAG_BINNED_SCATTER);
    if (nearest && gated && k.sep != 0 && k.sep != 2) return 0;     // (the nearest-neighbour probe reads dense grids and displacement fields)
    // (a kernel, not hipMemsetAsync: under hipGraph capture the memset node of ROCm 7.2 was observed not to re-run on replays)
    hipLaunchKernelGGL(own_zero, dim3((unsigned)((64 + 3ll * w.nbricks + 1023) / 1024)), dim3(1024), 0, st, (int *)w.hdr, 64 + 3 * w.nbricks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int *gate = nullptr;
    const int B = (int)p->batch;
    if (gated && nearest) {
        // the other organisation is the generic kernel, one global atomic per sample and channel: at the HBM roofline on the identity
        // (4 x 2 x 256^3: 0.59 ms, the bricks 2.3), 4.5 ms at sigma = 1 (2.4).  The trilinear pull's probe decides (lin_probe: mean |second
        // difference| of the coordinates; word 0 of the header is the gate, its ticket sits in words 2 - 3, the colour counters behind)
        const int rp = linear_pull_probe(p, k, grid, w.hdr, st, &gate, NEAREST_THR16);
        if (rp) return rp;
    } else if (gated) {
        const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
        const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = tile_count(p);
        const long long total = (long long)ntiles * B;
        const dim3 pgrid((unsigned)(total < NPROBE ? total : NPROBE));
#define IP_OWN_PROBE(KK, GM) hipLaunchKernelGGL((own_probe<KK, GM>), pgrid, dim3(NT1), 0, st, k, bg, (const float *)grid, w.hdr, gx, gy, gz, nty, ntz, ntiles, B, nch);
#define IP_OWN_PROBE_GM(KK) { if (k.sep == 0) IP_OWN_PROBE(KK, 0) else if (k.sep == 1) IP_OWN_PROBE(KK, 1) else if (k.sep == 2) IP_OWN_PROBE(KK, 2) else IP_OWN_PROBE(KK, 3) }
        if (k.order[0] == 3 || mixed_orders(k)) IP_OWN_PROBE_GM(3) else if (k.order[0] == 2) IP_OWN_PROBE_GM(2) else IP_OWN_PROBE_GM(1)
#undef IP_OWN_PROBE_GM
#undef IP_OWN_PROBE
        gate = &w.hdr->gate;
    }
    int rc;
    switch (count_only ? INTERPOL_F32 : p->dtype) {
    case INTERPOL_F32: rc = launch_bin<float>(p, k, bg, w, val, grid, vol, gate, st); break;
    case INTERPOL_BF16: rc = launch_bin<bf16_t>(p, k, bg, w, val, grid, vol, gate, st); break;
    case INTERPOL_F16: rc = launch_bin<f16_t>(p, k, bg, w, val, grid, vol, gate, st); break;
    default: return 0;
    }
    if (rc) return rc;
    // (a shared target used to take ONE launch over the bricks of every item with an atomic flush, colour 9; since the items share
    //  the bricks -- round 5 -- it takes the colour launches like any other target: plain loads and stores, one batch item)
    const long long want = 2ll * cu_count();
    const int Bw = shared ? 1 : B;
    for (int color = 0; color < 9; ++color) {
        long long nwork = Bw;
        for (int d = 0; d < 3; ++d) nwork *= color_count(color, d, bg);
        if (nwork <= 0) continue;
        const dim3 agrid((unsigned)(nwork < want ? nwork : want));
#define IP_OWN_ACC(KK)                                                                                                  \
        {                                                                                                               \
            const int attr = big_lds<own_accumulate<KK>>(sizeof(AccSmem));                                              \
            if (attr) return attr;                                                                                      \
            hipLaunchKernelGGL((own_accumulate<KK>), agrid, dim3(NT), sizeof(AccSmem), st, k, bg, (const int *)w.ndesc,  \
                               (const uint2 *)w.desc, (const float4 *)w.rec, (const float *)w.vals, (const unsigned short *)w.meta,  \
                               (const int *)w.bmax, w.nrec, (float *)vol, nch, color, Bw, gate, \
                               (int *)w.hdr + 16 + color);                                                              \
        }
        if (mixed_orders(k)) { const int rm = mix_launch_acc(k, bg, w, vol, nch, color, Bw, gate, agrid.x, st); if (rm) return rm; }
        else if (k.order[0] == 3) IP_OWN_ACC(3) else if (k.order[0] == 2) IP_OWN_ACC(2)
        else if (nearest) {
            const int attr = big_lds<own_accumulate<1, true>>(sizeof(AccSmem));
            if (attr) return attr;
            hipLaunchKernelGGL((own_accumulate<1, true>), agrid, dim3(NT), sizeof(AccSmem), st, k, bg, (const int *)w.ndesc,
                               (const uint2 *)w.desc, (const float4 *)w.rec, (const float *)w.vals, (const unsigned short *)w.meta,
                               (const int *)w.bmax, w.nrec, (float *)vol, nch, color, Bw, gate, (int *)w.hdr + 16 + color);
        }
        else IP_OWN_ACC(1)
#undef IP_OWN_ACC
    }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (gate_out) *gate_out = gate;
    return gated ? 2 : 1;
}

// ---------------------------------------------------------------------------
// The owner-computes PULL: interpol_pull_ws (abi.hip).  Same workspace layout as the push with one channel; returns 1 when it
// took the problem, 2 when it launched itself GATED behind the roughness probe (INTERPOL_FLAG_AUTO_SCATTER: the caller launches the
// sample tiles as well, with KParams::gate = *gate_out), 0 to decline.
// ---------------------------------------------------------------------------
static bool owner_pull_eligible(const interpol_problem *p, const KParams &k)
{
    if (p->dtype != INTERPOL_F32 || p->grid_dtype != INTERPOL_F32) return false;
    if (p->val_stride[0] < 0) return false;
    return owner_eligible(p, k);
}
int64_t owner_pull_workspace_bytes(const interpol_problem *p, const KParams &k)
{
    if (!owner_pull_eligible(p, k)) return 0;
    const int nt = owner::tile_count(p);
    return owner::layout(k, (int)p->batch, nt, 1, nullptr, nullptr, (int64_t)nt * p->batch);
}
// Step 1 (before the sample tiles): zero the header, the brick counters and the tile flags.  Returns 1 and the flag array (one int
// per (batch item, tile) in pull_sorted's order: the tiles it leaves to the bricks), 0 to decline.
int owner_pull_prepare(const interpol_problem *p, const KParams &k, void *workspace, int64_t workspace_bytes, hipStream_t st, int **flags_out, int *nzero_out)
{
    using namespace owner;
    if (!workspace || !owner_pull_eligible(p, k)) return 0;
    if (((uintptr_t)workspace & 255u) != 0) return 0;
    const int nt = tile_count(p);
    const int64_t nflags = (int64_t)nt * p->batch;
    Workspace w;
    if (layout(k, (int)p->batch, nt, 1, workspace, &w, nflags) > workspace_bytes) return 0;
    const int64_t nz = 64 + 3ll * w.nbricks;                         // header, brick counters, brick list: contiguous, right in front of the flags
    if (nz + nflags > 0x7fffffffll || (int *)w.hdr + nz != w.flags) return 0;
    if (p->flags & INTERPOL_FLAG_BINNED_SCATTER) {
        // the bricks alone: no tile kernel in front that could clear the counters on its way
        hipLaunchKernelGGL(own_zero, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, (int *)w.hdr, (int)nz);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    *flags_out = w.flags;
    *nzero_out = (int)nz;
    return 1;
}
// Step 1b (grid gradient, INTERPOL_FLAG_AUTO_SCATTER): clear the counters and let the probe decide whether EVERY tile goes to the
// bricks (hdr->gate = 1: gradc_sorted returns at once, own_bin takes every tile) or the sample tiles run and flag what they leave
int owner_grad_probe(const interpol_problem *p, const KParams &k, const void *grid, void *workspace, int64_t workspace_bytes, hipStream_t st, int mode)
{
    using namespace owner;
    const int ntiles = tile_count(p);
    Workspace w;
    if (layout(k, (int)p->batch, ntiles, 1, workspace, &w, (int64_t)ntiles * p->batch) > workspace_bytes) return INTERPOL_E_SCRATCH;
    const BrickGrid bg = brick_grid(k);
    const int64_t nz = 64 + 3ll * w.nbricks;
    hipLaunchKernelGGL(own_zero, dim3((unsigned)((nz + 1023) / 1024)), dim3(1024), 0, st, (int *)w.hdr, (int)nz);
    const int B = (int)p->batch;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS;
    const long long total = (long long)ntiles * B;
    const dim3 pgrid((unsigned)(total < NPROBE ? total : NPROBE));
#define IP_OWN_PROBE(KK, GM) hipLaunchKernelGGL((own_probe<KK, GM>), pgrid, dim3(NT1), 0, st, k, bg, (const float *)grid, w.hdr, gx, gy, gz, nty, ntz, ntiles, B, mode);
#define IP_OWN_PROBE_GM(KK) { if (k.sep == 0) IP_OWN_PROBE(KK, 0) else if (k.sep == 1) IP_OWN_PROBE(KK, 1) else if (k.sep == 2) IP_OWN_PROBE(KK, 2) else IP_OWN_PROBE(KK, 3) }
    if (k.order[0] == 3 || mixed_orders(k)) IP_OWN_PROBE_GM(3) else IP_OWN_PROBE_GM(2)
#undef IP_OWN_PROBE_GM
#undef IP_OWN_PROBE
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
// ---------------------------------------------------------------------------
// The router of the TRILINEAR pull (round 5).  Eight taps leave an LDS box nothing to amortise on a smooth field -- the generic kernel
// gathers at the HBM roofline there (4 x 2 x 256^3: 0.56 ms; the class-sorted tiles 0.78) -- but under roughness its gathers miss the
// caches: i.i.d. noise of sigma = 1 / 2 / 4 voxels 1.27 / 2.49 / 4.5 ms against 0.89 / 1.02 / 1.7 for the tiles.  lin_probe looks at up to
// 512 sample tiles: mean absolute second difference of the coordinates along the last dim, summed over the dims (tile_common.hpp:
// tile_smooth's measure), above thr16 / 16 voxels -> word 0 of the 256-byte workspace = 1, the tiles; else 0, the generic kernel (the pull:
// one voxel, sigma ~ 0.17; the grid gradient, whose tile kernel costs 2.4 ms whatever the field: four voxels, sigma ~ 0.7).  Both are
// enqueued behind it; one relaxed 64-bit add per workgroup carries sum, count and ticket (no fence).
// ---------------------------------------------------------------------------
namespace owner {
template <int GM>
__global__ __launch_bounds__(256) void lin_probe(KParams p, const float *__restrict__ grid, int *__restrict__ hdr, int gx, int gy, int gz, int nty, int ntz,
                                                 int ntiles, long long total, int stride, int thr16)
{
    __shared__ int acc[2];
    const int tid = threadIdx.x;
    if (tid < 2) acc[tid] = 0;
    __syncthreads();
    const long long work = (long long)blockIdx.x * stride;
    const long long wk = work < total ? work : total - 1;
    const int64_t b = wk / ntiles;
    const TileGeom g = tile_geom((int)(wk % ntiles), gx, gy, gz, nty, ntz);
    float s = 0.f; int n = 0;
    for (int id = tid; id < TS * TS * TS; id += 512) {               // a sample of the tile is plenty
        int ox, oy, oz;
        sample_pos(g, id, ox, oy, oz);
        if (ox >= gx || oy >= gy || oz + 2 >= gz || (id & 15) + 2 >= TS) continue;
        float q0[3], q1[3], q2[3];
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, q0);
        load_xyz<GM>(p, grid, b, g, ox, oy, oz + 1, q1);
        load_xyz<GM>(p, grid, b, g, ox, oy, oz + 2, q2);
        for (int d = 0; d < 3; ++d) s += __builtin_fabsf(q0[d] - 2.f * q1[d] + q2[d]);
        ++n;
    }
    s = s < 1e4f ? s : 1e4f;                                         // (also catches NaN)
    if (n) { atomicAdd(&acc[0], (int)(s * 16.f)); atomicAdd(&acc[1], n); }
    __syncthreads();
    if (tid == 0) {
        // sum (bits 34..63: <= 512 tiles x 2^20), count (12..33), ticket (0..11)
        const unsigned long long a0 = (unsigned long long)(acc[0] < (1 << 20) ? acc[0] : (1 << 20)), a1 = (unsigned long long)acc[1];
        const unsigned long long add = work < total ? (a0 << 34) | (a1 << 12) | 1ull : 1ull;
        const unsigned long long tot = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(hdr + 2), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
        if ((tot & 0xfffull) == (unsigned long long)gridDim.x) {
            const unsigned long long sum16 = tot >> 34, cnt = (tot >> 12) & 0x3fffffull;
            __hip_atomic_store(&hdr[0], (cnt > 0 && sum16 > cnt * (unsigned long long)thr16) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace owner

// 0 = enqueued (*gate_out: the verdict word), else an error
int linear_pull_probe(const interpol_problem *p, const KParams &k, const void *grid, void *workspace, hipStream_t st, const int **gate_out, int thr16)
{
    using namespace owner;
    int *hdr = (int *)workspace;
    hipLaunchKernelGGL(own_zero, dim3(1), dim3(64), 0, st, hdr, 8);
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int ntx = (gx + TS - 1) / TS, nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS, ntiles = ntx * nty * ntz;
    const long long total = (long long)ntiles * p->batch;
    int stride = (int)((total + 511) / 512);
    stride = stride < 1 ? 1 : (stride | 1);                          // (odd: the probed tiles do not line up along an axis)
    const unsigned nb = (unsigned)((total + stride - 1) / stride);
    if (k.sep == 0) hipLaunchKernelGGL((lin_probe<0>), dim3(nb), dim3(256), 0, st, k, (const float *)grid, hdr, gx, gy, gz, nty, ntz, ntiles, total, stride, thr16);
    else hipLaunchKernelGGL((lin_probe<2>), dim3(nb), dim3(256), 0, st, k, (const float *)grid, hdr, gx, gy, gz, nty, ntz, ntiles, total, stride, thr16);
    *gate_out = hdr;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Step 2 (behind the sample tiles, or alone: all == true, INTERPOL_FLAG_BINNED_SCATTER): the flagged tiles' samples sorted by the
// brick of the image they read (own_bin, index mode; unbinned samples are gathered on the spot), then the bricks (own_gather).
// grad == true: the grid gradient of the pull (val := the dense grid gradient, gout := grad_out or NULL for ones).
int owner_pull_finish(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val,
                      void *workspace, int64_t workspace_bytes, bool all, hipStream_t st, bool grad, const void *gout, bool probed, bool spatial)
{
    using namespace owner;
    const int nt = tile_count(p);
    Workspace w;
    if (layout(k, (int)p->batch, nt, 1, workspace, &w, (int64_t)nt * p->batch) > workspace_bytes) return INTERPOL_E_SCRATCH;
    const BrickGrid bg = brick_grid(k);
    KParams kk = k;
    kk.gate = nullptr;
    // (spatial: grid_grad through the bricks -- the records of the pull; with the probe every tile or none, own_gather is gated)
    int rc = grad ? launch_bin<float, 2>(p, kk, bg, w, vol, grid, val, all ? nullptr : w.flags, st, gout, probed ? &w.hdr->gate : nullptr)
           : spatial ? launch_bin<float, 3>(p, kk, bg, w, vol, grid, val, nullptr, st, nullptr, probed ? &w.hdr->gate : nullptr)
                  : launch_bin<float, 1>(p, kk, bg, w, vol, grid, val, all ? nullptr : w.flags, st, nullptr, probed ? &w.hdr->gate : nullptr);
    if (rc) return rc;
    const long long want = 2ll * cu_count();
    const dim3 ggrid((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_OWN_GAT(KK, GR)                                                                                              \
    {                                                                                                                   \
        const int attr = big_lds<own_gather<KK, GR>>(sizeof(GatSmem));                                                  \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((own_gather<KK, GR>), ggrid, dim3(NT), sizeof(GatSmem), st, kk, bg, (const int *)w.ndesc, (const uint2 *)w.desc, \
                           (const float4 *)w.rec, (const int *)w.bmax, (int *)w.hdr + 40, (const float *)vol, (float *)val,            \
                           (spatial && probed) ? (const int *)&w.hdr->gate : (const int *)nullptr, (const float *)gout);  \
    }
    if (mixed_orders(k)) {
        const int rm = mix_launch_gather(grad ? 1 : (spatial ? 2 : 0), kk, bg, w, vol, val, (spatial && probed) ? (const int *)&w.hdr->gate : (const int *)nullptr, gout, ggrid.x, st);
        if (rm) return rm;
    }
    else if (grad) { if (k.order[0] == 3) IP_OWN_GAT(3, 1) else IP_OWN_GAT(2, 1) }
    else if (spatial) { if (k.order[0] == 3) IP_OWN_GAT(3, 2) else IP_OWN_GAT(2, 2) }
    else { if (k.order[0] == 3) IP_OWN_GAT(3, 0) else IP_OWN_GAT(2, 0) }
#undef IP_OWN_GAT
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace ip

#ifdef IP_PROF
extern "C" __attribute__((visibility("default"))) int interpol_debug_prof_owner(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::sorted::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::sorted::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
#else // IP_OWNER_MIX_TU
} // namespace ip
#endif
#undef IP_KS
#undef IP_KD
#undef IP_WX
#undef IP_MU
#undef IP_RECORD_CELL
