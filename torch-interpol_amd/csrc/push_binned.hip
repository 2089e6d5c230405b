// ===========================================================================
// push_binned.hip -- grid_push / grid_count for deformations at the sampling density of the
// target ("same resolution" splatting, BASELINE config 2), TARGET-STATIONARY after a binning pass:
// 3-D, one spline order 2..3, f32 / bf16 / f16 sources, any boundary / extrapolation mode.
// Reference semantics: interpol/nd.py:146-213 (push), pushpull.py:106-142 (count).
//
// Why.  A tile of SOURCE samples scatters into the bounding box of its stencils: tile + K + twice
// the largest displacement -- 9 lattice points per sample for 16^3 tiles under sigma = 2 voxel
// noise.  Every one of them costs a global atomic per channel when the tile is flushed, and the
// memory side retires only ~0.6 G float atomics per millisecond: 2 of the 3.5 ms of the tiled push
// (measured: tools/ablate_sorted.py) whatever the tap loop does.  Binned by the lattice cell of their
// FIRST TAP instead, the samples of a brick of 12^3 cells touch brick + K = 15^3 lattice points
// whatever the deformation: 1.95 points per sample, 4.6 times fewer global atomics, a 30 KiB LDS box,
// one pass over the whole stencil (weights evaluated once), no out-of-box samples.
//
//   bin_count : per tile of 16^3 sample points, how many land in which brick (LDS histogram over
//               the 6^3 bricks around the tile, then one global add per non-empty bin)
//   bin_scan  : exclusive scan of the brick counts (one workgroup) -> list offsets
//   bin_fill  : the same walk again; each sample's record (stencil coordinates t, cell inside the
//               brick) and its source values (masked, converted to float) go to the brick's list
//   bin_accumulate : one workgroup per brick (persistent, two per CU): batches of <= 2048 records are
//               class-sorted onto the lanes through LDS (conflict-free ds_add_u64: ops_sorted.hip),
//               accumulated in the LDS box in packed 32-bit fixed point (two channels per atomic),
//               and the box is added to the target with coalesced global atomics.
// The workspace (records: 16 B per sample, values: 8 B per sample and channel pair, three ints per
// brick) is the caller's: interpol_push_workspace() sizes it.
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace binned {

using namespace sorted;          // helpers of sorted_util.hpp

constexpr int BR = 12;                          // brick edge, in first-tap cells
constexpr int OFF = 24;                         // first taps in [-OFF, n + OFF) are binned; beyond: scattered directly
constexpr int BOX = BR + 3;                     // lattice points a brick's stencils touch per dim (K <= 3)
constexpr int PZ = 17;                          // row pitch of the LDS box (odd: spreads the bank classes)
constexpr int PLANE = BOX * PZ;                 // 255
constexpr int BOXSLOTS = BOX * PLANE;           // 3825 slots of 8 bytes
constexpr int NT = 512, VPT = 4, CAP = NT * VPT;   // accumulate: threads, records per thread, records per batch
constexpr int NCLS = 32, NSLOT = CAP / NCLS;
constexpr int LB = 6, NBIN = LB * LB * LB;      // bricks around a tile that are counted in LDS
constexpr int NT1 = 256, VPT1 = TS * TS * TS / NT1;   // count / fill: threads, samples per thread

struct BrickGrid {
    int nb[3];                                  // bricks per dim
    int per_item;                               // nb[0] * nb[1] * nb[2]
};
static BrickGrid brick_grid(const KParams &k)
{
    BrickGrid g;
    for (int d = 0; d < 3; ++d) g.nb[d] = (k.vol_n[d] + 2 * OFF + BR - 1) / BR;
    g.per_item = g.nb[0] * g.nb[1] * g.nb[2];
    return g;
}

// first-tap cell -> brick; false when the cell lies outside the binned range
__device__ __forceinline__ bool brick_of(const float *fl, const BrickGrid &bg, int *bxyz, int *cell)
{
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        ok = ok && fl[d] >= (float)(-OFF) && fl[d] < (float)(bg.nb[d] * BR - OFF);
        const int i = ok ? __float2int_rz(fl[d]) + OFF : 0;
        bxyz[d] = i / BR;
        cell[d] = i - bxyz[d] * BR;
    }
    return ok;
}

// the 6^3 bricks counted locally around the tile at (ox0, oy0, oz0): first taps from tile origin - 25
__device__ __forceinline__ void local_base(const TileGeom &g, const float *scale, int *lb)
{
    const int o[3] = { g.ox0, g.oy0, g.oz0 };
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int lo = (int)((float)o[d] * scale[d]) - 25 + OFF;
        lb[d] = lo < 0 ? 0 : lo / BR;
    }
}

// ---------------------------------------------------------------------------
// bin_count / bin_fill
// ---------------------------------------------------------------------------
struct BinSmem {
    int cnt[NBIN];             // samples of the tile per local brick
    int base[NBIN];            // fill: position of the bin's first record in its brick's list
};

// FILL = false: count;  FILL = true: write records and values (MODE: 0 values, 1 count only, 2 values + count)
template <typename T, int K, int GM, bool FILL, int MODE>
__global__ __launch_bounds__(NT1) void bin_tiles(KParams p, BrickGrid bg, const T *__restrict__ val, const float *__restrict__ grid,
                                                 float *__restrict__ vol, int *__restrict__ cnt, const unsigned *__restrict__ offs,
                                                 int *__restrict__ cursor, float4 *__restrict__ rec, float2 *__restrict__ vals, int64_t nrec,
                                                 int gx, int gy, int gz, int nty, int ntz, int ntiles)
{
    __shared__ BinSmem sm;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / ntiles;
    const TileGeom g = tile_geom(blockIdx.x % ntiles, gx, gy, gz, nty, ntz);
    for (int i = tid; i < NBIN; i += NT1) sm.cnt[i] = 0;
    // scale of the sample grid onto the lattice (only to centre the local brick window)
    const float scale[3] = { (float)p.vol_n[0] / (float)gx, (float)p.vol_n[1] / (float)gy, (float)p.vol_n[2] / (float)gz };
    int lb[3];
    local_base(g, scale, lb);
    float c[VPT1][3];
    unsigned valid = 0;
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        int ox, oy, oz;
        sample_pos(g, tid + NT1 * v, ox, oy, oz);
        if (ox < gx && oy < gy && oz < gz) valid |= 1u << v;
        ox = ox < gx ? ox : gx - 1; oy = oy < gy ? oy : gy - 1; oz = oz < gz ? oz : gz - 1;
        load_xyz<GM>(p, grid, b, g, ox, oy, oz, c[v]);
    }
    __syncthreads();
    int lbin[VPT1], rank[VPT1], brick[VPT1], key[VPT1];
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        float fl[3];
        const float m = inb_mask(p, c[v]);
#pragma unroll
        for (int d = 0; d < 3; ++d) { fl[d] = floorf(c[v][d] - 0.5f * (float)(K - 1)); c[v][d] -= fl[d]; }
        int bx[3], cell[3];
        const bool ok = brick_of(fl, bg, bx, cell) && ((valid >> v) & 1);
        brick[v] = ok ? (int)b * bg.per_item + (bx[0] * bg.nb[1] + bx[1]) * bg.nb[2] + bx[2] : -1;
        key[v] = cell[0] | (cell[1] << 4) | (cell[2] << 8) | ((m != 0.f ? 1 : 0) << 12);
        const int r0 = bx[0] - lb[0], r1 = bx[1] - lb[1], r2 = bx[2] - lb[2];
        const bool local = ok && (unsigned)r0 < LB && (unsigned)r1 < LB && (unsigned)r2 < LB;
        lbin[v] = local ? (r0 * LB + r1) * LB + r2 : -1;
        rank[v] = 0;
        if (local) rank[v] = atomicAdd(&sm.cnt[lbin[v]], 1);
        else if (ok) {
            // outside the local window (rare): straight to the brick's global counter / cursor
            if (!FILL) atomicAdd(&cnt[brick[v]], 1);
            else rank[v] = atomicAdd(&cursor[brick[v]], 1);
        } else if (FILL && ((valid >> v) & 1)) {
            // first tap outside the binned range (far outside the field of view): scatter it here
            Lattice L;
#pragma unroll
            for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = K; }
            L.lin = 0;
            int ox, oy, oz;
            sample_pos(g, tid + NT1 * v, ox, oy, oz);
            const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
            const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);
            for (int ch = 0; ch < nch; ++ch) {
                const float s = (MODE == 1 || ch >= p.C) ? m : m * Cvt<float, T>::ld(val[b * p.val_sb + ch * p.val_sc + o]);
                const float f0 = fl[0] < -1073741824.f ? -1073741824.f : (fl[0] > 1073741824.f ? 1073741824.f : fl[0]);
                const float f1 = fl[1] < -1073741824.f ? -1073741824.f : (fl[1] > 1073741824.f ? 1073741824.f : fl[1]);
                const float f2_ = fl[2] < -1073741824.f ? -1073741824.f : (fl[2] > 1073741824.f ? 1073741824.f : fl[2]);
                tiled::scatter_one_thread(L, vol + b * p.vol_sb + ch * p.vol_sc, s, (int)f0, (int)f1, (int)f2_, c[v][0], c[v][1], c[v][2]);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < NBIN; i += NT1) {
        const int n = sm.cnt[i];
        if (n == 0) continue;
        const int r0 = i / (LB * LB), r1 = (i / LB) % LB, r2 = i % LB;
        const int bk = (int)b * bg.per_item + ((lb[0] + r0) * bg.nb[1] + (lb[1] + r1)) * bg.nb[2] + (lb[2] + r2);
        if (!FILL) atomicAdd(&cnt[bk], n);
        else sm.base[i] = atomicAdd(&cursor[bk], n);
    }
    if (!FILL) return;
    __syncthreads();
    const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);
#pragma unroll
    for (int v = 0; v < VPT1; ++v) {
        if (brick[v] < 0) continue;
        const int64_t pos = (int64_t)offs[brick[v]] + (lbin[v] >= 0 ? sm.base[lbin[v]] : 0) + rank[v];
        rec[pos] = make_float4(c[v][0], c[v][1], c[v][2], __int_as_float(key[v]));
        if (MODE != 1) {
            int ox, oy, oz;
            sample_pos(g, tid + NT1 * v, ox, oy, oz);
            const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
            const float m = (float)((key[v] >> 12) & 1);
            const T *ib = val + b * p.val_sb + o;
            for (int ch = 0; ch < p.C; ch += 2) {
                const float s0 = Cvt<float, T>::ld(ib[ch * p.val_sc]) * m;
                const float s1 = ch + 1 < p.C ? Cvt<float, T>::ld(ib[(ch + 1) * p.val_sc]) * m : (MODE == 2 ? m : 0.f);
                vals[(ch >> 1) * nrec + pos] = make_float2(s0, s1);
            }
            if (MODE == 2 && (p.C & 1) == 0) {
                // even channel count + count: the count gets a pair of its own (its second half is unused)
                // (no store needed: the accumulate kernel builds the ones from the mask bit)
            }
        }
        (void)nch;
    }
}

// exclusive scan of the brick counts (one workgroup; the counts are then reused as fill cursors)
template <int DUMMY>            // (a template: the file is compiled once per storage type and this kernel is the same in each)
__global__ __launch_bounds__(1024) void bin_scan(int *__restrict__ cnt, unsigned *__restrict__ offs, int *__restrict__ cursor, int nbricks)
{
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    const int per = (nbricks + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < nbricks ? lo + per : nbricks;
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += (unsigned)cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned t = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    unsigned run = part[tid] - s;
    for (int i = lo; i < hi; ++i) { offs[i] = run; run += (unsigned)cnt[i]; cursor[i] = 0; }
    if (tid == 1023) offs[nbricks] = part[1023];
}

// ---------------------------------------------------------------------------
// bin_accumulate
// ---------------------------------------------------------------------------
struct AccSmem {
    int   taboff[3][16];
    float tabsgn[3][16];
    int   cnt[NCLS + 4];
    int   ooff[NCLS];
    int   cmax[2];
    int   dmax, pad;
    unsigned cells[BR * BR * BR / 2];          // density: 16-bit counters per first-tap cell
    unsigned short holes[CAP];                 // surplus placement (see ops_sorted.hip)
    // the record / value exchange of the sort and the accumulation box share their memory (the box is
    // all-zero between batches: it is re-zeroed once the sorted records have been read)
    union {
        struct { float4 xch[CAP]; float2 xv[CAP]; } x;
        unsigned long long box[BOXSLOTS];
    } u;
};

// the 4 LDS adds of one row of the stencil, at immediate offsets (i, jy compile-time)
template <int I, int J>
__device__ __forceinline__ void row_adds(unsigned addr, unsigned long long v0, unsigned long long v1, unsigned long long v2, unsigned long long v3)
{
    constexpr int o = (I * PLANE + J * PZ) * 8;
    asm volatile("ds_add_u64 %0, %1 offset:%5\n\tds_add_u64 %0, %2 offset:%6\n\tds_add_u64 %0, %3 offset:%7\n\tds_add_u64 %0, %4 offset:%8"
                 :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(o), "n"(o + 8), "n"(o + 16), "n"(o + 24) : "memory");
}

template <int K, int I, int J>
__device__ __forceinline__ void scatter_row(unsigned addr, f2 sx, const f2 *w)
{
    const f2 sy = sx * f2{ w[J].x, w[J].x };
    unsigned long long v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f2 pr = sy * f2{ w[k].y, w[k].y };
        const int q0 = tiled::cvt_rpi(pr.x), q1 = tiled::cvt_rpi(pr.y);
        // (q1 << 32) + sext(q0): low word q0, high word q1 + (q0 < 0 ? -1 : 0)
        v[k] = ((unsigned long long)(unsigned)(q1 + (q0 >> 31)) << 32) | (unsigned)q0;
    }
    row_adds<I, J>(addr, v[0], v[1], v[2], v[3]);
}
template <int K, int I>
__device__ __forceinline__ void scatter_plane(unsigned addr, f2 s, float wxi, const f2 *w)
{
    const f2 sx = s * f2{ wxi, wxi };
    scatter_row<K, I, 0>(addr, sx, w); scatter_row<K, I, 1>(addr, sx, w); scatter_row<K, I, 2>(addr, sx, w);
    if (K == 3) scatter_row<K, I, 3>(addr, sx, w);
}

// MODE as above; the values are float (bin_fill converted them)
template <typename TAG, int K, int MODE>     // TAG: one copy per translation unit (storage type); the kernel itself reads floats
__global__ __launch_bounds__(NT, 4) void bin_accumulate(KParams p, BrickGrid bg, const int *__restrict__ cnt, const unsigned *__restrict__ offs,
                                                        const float4 *__restrict__ rec, const float2 *__restrict__ vals, int64_t nrec,
                                                        float *__restrict__ vol, int nbricks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    AccSmem &sm = *reinterpret_cast<AccSmem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = K; }
    L.lin = 0;
    const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);
    for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
        const int n = cnt[brick];
        if (n == 0) continue;                                        // block-uniform
        const int tid = opaque((int)threadIdx.x);
        const int64_t b = brick / bg.per_item;
        int r = brick % bg.per_item;
        const int bz = r % bg.nb[2]; r /= bg.nb[2];
        const int b0[3] = { (r / bg.nb[1]) * BR - OFF, (r % bg.nb[1]) * BR - OFF, bz * BR - OFF };   // lattice index of box slot 0
        __syncthreads();                                             // the previous brick's flush read the tables
        if (tid < 3 * 16) {
            const int d = tid >> 4, slot = tid & 15;
            if (slot < BOX) {
                const long long pk = wrap_outofline(L.bound[d], b0[d] + slot, L.n[d]);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * L.ss[d];
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        const int64_t off = (int64_t)offs[brick];
        prof_mark(-1);
        for (int s0 = 0; s0 < n; s0 += CAP) {
            const int m = n - s0 < CAP ? n - s0 : CAP;
            // ---- records of the batch, list order; class = LDS slot of the first tap mod 32
            if (tid <= NCLS) sm.cnt[tid] = 0;
            if (tid == 0) { sm.dmax = 0; sm.cmax[0] = 0; sm.cmax[1] = 0; }
            for (int e = tid; e < BR * BR * BR / 2; e += NT) sm.cells[e] = 0u;
            float4 rc[VPT]; float2 rv[VPT];
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int li = tid + NT * v;
                rc[v] = rec[off + s0 + (li < m ? li : 0)];
                rv[v] = MODE == 1 ? make_float2(0.f, 0.f) : vals[off + s0 + (li < m ? li : 0)];     // values of the first channel pair
            }
            __syncthreads();
            prof_mark(0);
            int kq[VPT], rk[VPT];
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int li = tid + NT * v;
                const bool on = li < m;
                const int key = __float_as_int(rc[v].w);
                const int x0 = key & 15, y0 = (key >> 4) & 15, z0 = (key >> 8) & 15;
                const int slot = x0 * PLANE + y0 * PZ + z0;
                // sorted key: bits 0-11 LDS slot of the first tap, 12 mask, 13-23 list index in the batch, 24 record present
                kq[v] = slot | (key & (1 << 12)) | (li << 13) | ((on ? 1 : 0) << 24);
                rk[v] = atomicAdd(&sm.cnt[on ? (slot & (NCLS - 1)) : NCLS], 1);
                if (on) {
                    const int cell = (x0 * BR + y0) * BR + z0;
                    atomicAdd(&sm.cells[cell >> 1], 1u << (16 * (cell & 1)));
                }
            }
            __syncthreads();
            prof_mark(1);
            // holes and surplus (ops_sorted.hip: Tile::build)
            const int q_l = tid & 31;
            const int cq = sm.cnt[q_l];
            const int holes = cq < NSLOT ? NSLOT - cq : 0, surplus = cq > NSLOT ? cq - NSLOT : 0;
            int nholes, nsurplus;
            const int hoff = half_excl_scan(holes, nholes);
            const int soff = half_excl_scan(surplus, nsurplus);
            if (tid < NCLS) sm.ooff[tid] = soff;
            for (int mm = tid >> 5; mm < holes && hoff + mm < nsurplus; mm += NT / 32)
                sm.holes[hoff + mm] = (unsigned short)((cq + mm) * 32 + q_l);
            const int filled = nsurplus - hoff < 0 ? 0 : (nsurplus - hoff > holes ? holes : nsurplus - hoff);
            const int cnteff = (cq < NSLOT ? cq : NSLOT) + filled;
            {   // density of the batch: the largest number of records sharing a first-tap cell
                int dm = 0;
                for (int e = tid; e < BR * BR * BR / 2; e += NT) {
                    const unsigned w2 = sm.cells[e];
                    const int a = (int)(w2 & 0xffffu), c2 = (int)(w2 >> 16);
                    dm = a > dm ? a : dm; dm = c2 > dm ? c2 : dm;
                }
                dm = wave_max(dm);
                if ((tid & 63) == 0 && dm > 0) atomicMax(&sm.dmax, dm);
            }
            __syncthreads();
            prof_mark(2);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!((kq[v] >> 24) & 1)) continue;
                const int q = kq[v] & 31, rr = rk[v];
                int pos = rr * 32 + q;
                if (rr >= NSLOT) pos = (int)sm.holes[sm.ooff[q] + rr - NSLOT];
                sm.u.x.xch[pos] = make_float4(rc[v].x, rc[v].y, rc[v].z, __int_as_float(kq[v]));
                sm.u.x.xv[pos] = rv[v];
            }
            __syncthreads();
            float tx[VPT]; f2 tyz[VPT]; int key[VPT]; float2 sv0[VPT];
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const float4 x4 = sm.u.x.xch[tid + NT * j];
                sv0[j] = sm.u.x.xv[tid + NT * j];
                const bool on = (tid >> 5) + (NT / 32) * j < cnteff;
                tx[j] = on ? x4.x : 1.25f; tyz[j] = on ? f2{ x4.y, x4.z } : f2{ 1.25f, 1.25f };
                key[j] = on ? __float_as_int(x4.w) : 0;
            }
            __syncthreads();                                         // the exchange buffers become the (zero) box
            for (int e = tid; e < BOXSLOTS; e += NT) sm.u.box[e] = 0ull;
            const int hb = tiled::headroom32(L, sm.dmax);
            prof_mark(3);
            for (int c = 0; c < nch; c += 2) {
                const bool two = c + 1 < nch;
                float *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
                float *vc1 = two ? vc0 + p.vol_sc : vc0;
                const bool ones0 = MODE == 1 || (MODE == 2 && c >= p.C), ones1 = MODE == 1 || (MODE == 2 && c + 1 >= p.C);
                f2 src[VPT];
                float am0 = 0.f, am1 = 0.f;
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const bool on = (key[j] >> 24) & 1;
                    const float mk = (float)((key[j] >> 12) & 1);
                    float2 sv = sv0[j];
                    if (c > 0 && !ones0) sv = vals[(int64_t)(c >> 1) * nrec + off + s0 + ((key[j] >> 13) & 2047)];   // (further pairs: gathered; list index 0 for an empty slot)
                    const float s0_ = on ? (ones0 ? mk : sv.x) : 0.f;
                    const float s1_ = (on && two) ? (ones1 ? mk : sv.y) : 0.f;
                    src[j] = f2{ s0_, s1_ };
                    const float a0 = __builtin_fabsf(s0_), a1 = __builtin_fabsf(s1_);
                    am0 = (a0 > am0 || a0 != a0) ? a0 : am0;
                    am1 = (a1 > am1 || a1 != a1) ? a1 : am1;
                }
                {
                    const int m0 = wave_max(__float_as_int(am0)), m1 = wave_max(__float_as_int(am1));
                    if ((tid & 63) == 0) { if (m0) atomicMax(&sm.cmax[0], m0); if (m1) atomicMax(&sm.cmax[1], m1); }
                }
                __syncthreads();
                const int mb0 = sm.cmax[0], mb1 = sm.cmax[1];
                const bool fin = (mb0 & 0x7f800000) != 0x7f800000 && (mb1 & 0x7f800000) != 0x7f800000;
                int ex0 = ((mb0 >> 23) & 0xff) - 127, ex1 = ((mb1 >> 23) & 0xff) - 127;
                ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
                const int hbc = hb < 0 ? 0 : hb;
                const f2 scale = { mb0 ? __int_as_float((127 + 29 - ex0 - hbc) << 23) : 0.f, mb1 ? __int_as_float((127 + 29 - ex1 - hbc) << 23) : 0.f };
                const float inv0 = __int_as_float((127 - 29 + ex0 + hbc) << 23), inv1 = __int_as_float((127 - 29 + ex1 + hbc) << 23);
                prof_mark(4);
                if (hb >= 0 && fin) {
                    const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.u.box);
#pragma unroll
                    for (int j = 0; j < VPT; ++j) {
                        // empty slots sit out: they would all add zeros to the SAME slots, and same-address LDS
                        // atomics of one instruction serialise
                        if (!((key[j] >> 24) & 1)) continue;
                        const unsigned addr = boxaddr + 8u * (unsigned)(key[j] & 4095);
                        f2 w[4];
                        weights_yz<K>(tyz[j], w);
                        const f2 ss = src[j] * scale;
                        scatter_plane<K, 0>(addr, ss, weight_x<K>(tx[j], 0), w);
                        scatter_plane<K, 1>(addr, ss, weight_x<K>(tx[j], 1), w);
                        scatter_plane<K, 2>(addr, ss, weight_x<K>(tx[j], 2), w);
                        if (K == 3) scatter_plane<K, 3>(addr, ss, weight_x<K>(tx[j], 3), w);
                        asm volatile("" :: "v"(addr));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                    prof_mark(5);
                    // flush: fixed point -> float, slot sign, one global atomic per touched slot and channel; re-zeroed on the way
                    for (int e = tid; e < BOX * BOX * 16; e += NT) {
                        const int z = e & 15, row = e >> 4;
                        if (z >= BOX) continue;
                        const int xr = row / BOX, yr = row - xr * BOX;
                        unsigned long long *sp = sm.u.box + xr * PLANE + yr * PZ + z;
                        const long long a = (long long)*sp;
                        if (a == 0) continue;
                        if (c + 2 < nch) *sp = 0ull;                 // (another channel pair accumulates next)
                        const int lo_ = (int)(a & 0xffffffffll);
                        const int hi_ = (int)((a - (long long)lo_) >> 32);
                        const int o3 = sm.taboff[0][xr] + sm.taboff[1][yr] + sm.taboff[2][z];
                        const float sg = sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sm.tabsgn[2][z];
                        if (lo_ != 0) __hip_atomic_fetch_add(vc0 + o3, (float)lo_ * (inv0 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (hi_ != 0) __hip_atomic_fetch_add(vc1 + o3, (float)hi_ * (inv1 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    // no fixed point for this batch (density beyond the precision rule, non-finite sources):
                    // every record tap-parallel, one wave per record, float atomics straight to global memory
                    const int lane = tid & 63;
                    for (int j = 0; j < VPT; ++j) {
                        for (int l = 0; l < 64; ++l) {
                            const int kk = __shfl(key[j], l);
                            if (!((kk >> 24) & 1)) continue;         // wave-uniform
                            const int slot = kk & 4095;
                            const int x0 = slot / PLANE, y0 = (slot - x0 * PLANE) / PZ, z0 = slot - x0 * PLANE - y0 * PZ;
                            const float x = (float)(b0[0] + x0) + __shfl(tx[j], l), y = (float)(b0[1] + y0) + __shfl(tyz[j].x, l),
                                        z = (float)(b0[2] + z0) + __shfl(tyz[j].y, l);
                            const float s0_ = __shfl(src[j].x, l), s1_ = __shfl(src[j].y, l);
                            int o3;
                            const float wt = tiled::tap_weight_t<K, K>(L, x, y, z, lane, &o3, nullptr);
                            if (lane < (K + 1) * (K + 1) * (K + 1)) {
                                __hip_atomic_fetch_add(vc0 + o3, wt * s0_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (two) __hip_atomic_fetch_add(vc1 + o3, wt * s1_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                __syncthreads();
                prof_mark(6);
                if (tid == 0) { sm.cmax[0] = 0; sm.cmax[1] = 0; }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct Workspace {
    int *cnt; unsigned *offs; int *cursor; float4 *rec; float2 *vals;
    int64_t nrec; int nbricks; int npairs;
};
static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static int64_t layout(const KParams &k, int B, int nch, bool with_vals, void *base, Workspace *w)
{
    const BrickGrid bg = brick_grid(k);
    const int64_t nbricks = (int64_t)bg.per_item * B;
    const int64_t nrec = k.N * B;
    const int npairs = with_vals ? (nch + 1) / 2 : 0;
    int64_t o = 0;
    unsigned char *p = (unsigned char *)base;
    const int64_t o_cnt = o; o += align256(nbricks * 4);
    const int64_t o_off = o; o += align256((nbricks + 1) * 4);
    const int64_t o_cur = o; o += align256(nbricks * 4);
    const int64_t o_rec = o; o += align256(nrec * 16);
    const int64_t o_val = o; o += align256(nrec * 8 * npairs);
    if (w) {
        w->cnt = (int *)(p + o_cnt); w->offs = (unsigned *)(p + o_off); w->cursor = (int *)(p + o_cur);
        w->rec = (float4 *)(p + o_rec); w->vals = (float2 *)(p + o_val);
        w->nrec = nrec; w->nbricks = (int)nbricks; w->npairs = npairs;
    }
    return o;
}

} // namespace binned

// Eligible: 3-D, one order 2..3, sample grid about as dense as the target (else the tiled / brick
// scatters are the better organisation), sizes within 32-bit record counts.
static bool binned_eligible(const interpol_problem *p, const KParams &k)
{
    if (p->dim != 3 || p->batch > 4096) return false;
    if (!(p->flags & INTERPOL_FLAG_BINNED_SCATTER)) return false;   // opt-in: see interpol_hip.h
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2] || k.order[0] < 2 || k.order[0] > 3) return false;
    int64_t n = 1, nv = 1, nb = p->batch;
    for (int d = 0; d < 3; ++d) {
        n *= p->grid_shape[d]; nv *= p->vol_shape[d];
        nb *= (p->vol_shape[d] + 2 * binned::OFF + binned::BR - 1) / binned::BR;
        if (p->grid_shape[d] > 0x7fffffff / 4) return false;
    }
    if (n < 4096 || n * p->batch > 0x7fffffffll || nb > 0x7fffffffll) return false;
    if ((uint64_t)n * 12ull > 0xffffffffull) return false;
    return 4 * n >= nv;                                              // at least a quarter of a sample per target voxel
}

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

// bytes of workspace the binned organisation needs for this problem (0: not applicable)
int64_t IP_SYM(binned_workspace_bytes_, IP_TSFX)(const interpol_problem *p, const KParams &k, bool count_only)
{
    if (!binned_eligible(p, k)) return 0;
    const int nch = count_only ? 1 : k.C + (k.cc ? 1 : 0);
    return binned::layout(k, (int)p->batch, nch, !count_only, nullptr, nullptr);
}

// returns 1 when it took the problem, 0 to decline (workspace missing / not eligible), else an error
int IP_SYM(try_binned_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol,
                                      void *workspace, int64_t workspace_bytes, hipStream_t st)
{
    using namespace binned;
    using T = IP_TT;
    if (!workspace || k.sep == 3 || !binned_eligible(p, k)) return 0;
    const bool count_only = val == nullptr;
    const int nch = count_only ? 1 : k.C + (k.cc ? 1 : 0);
    Workspace w;
    if (layout(k, (int)p->batch, nch, !count_only, workspace, &w) > workspace_bytes) return 0;
    const BrickGrid bg = brick_grid(k);
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int ntx = (gx + TS - 1) / TS, nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS;
    const int ntiles = ntx * nty * ntz;
    const dim3 tgrid((unsigned)(ntiles * (int)p->batch));
    hipError_t e = hipMemsetAsync(w.cnt, 0, (size_t)w.nbricks * 4, st);
    if (e != hipSuccess) return (int)e;
    const int K = k.order[0];
    const int mode = count_only ? 1 : (k.cc ? 2 : 0);
#define IP_BIN_TILES(KK, GM, FILL, MODE)                                                                               \
    hipLaunchKernelGGL((bin_tiles<T, KK, GM, FILL, MODE>), tgrid, dim3(NT1), 0, st, k, bg, (const T *)val, (const float *)grid, \
                       (float *)vol, w.cnt, (const unsigned *)w.offs, w.cursor, w.rec, w.vals, w.nrec, gx, gy, gz, nty, ntz, ntiles)
#define IP_BIN_BY_MODE(KK, GM, FILL)                                                                                   \
    { if (mode == 0) IP_BIN_TILES(KK, GM, FILL, 0); else if (mode == 1) IP_BIN_TILES(KK, GM, FILL, 1); else IP_BIN_TILES(KK, GM, FILL, 2); }
#define IP_BIN_BY_GM(KK, FILL)                                                                                         \
    { if (k.sep == 0) IP_BIN_BY_MODE(KK, 0, FILL) else if (k.sep == 1) IP_BIN_BY_MODE(KK, 1, FILL) else IP_BIN_BY_MODE(KK, 2, FILL) }
    if (K == 3) IP_BIN_BY_GM(3, false) else IP_BIN_BY_GM(2, false)
    hipLaunchKernelGGL(bin_scan<0>, dim3(1), dim3(1024), 0, st, w.cnt, w.offs, w.cursor, w.nbricks);
    if (K == 3) IP_BIN_BY_GM(3, true) else IP_BIN_BY_GM(2, true)
    const long long want = 2ll * cu_count();
    const dim3 agrid((unsigned)(w.nbricks < want ? w.nbricks : want));
#define IP_BIN_ACC(KK, MODE)                                                                                           \
    {                                                                                                                 \
        const int attr = big_lds<bin_accumulate<T, KK, MODE>>(sizeof(AccSmem));                                          \
        if (attr) return attr;                                                                                        \
        hipLaunchKernelGGL((bin_accumulate<T, KK, MODE>), agrid, dim3(NT), sizeof(AccSmem), st, k, bg, (const int *)w.cnt, \
                           (const unsigned *)w.offs, (const float4 *)w.rec, (const float2 *)w.vals, w.nrec, (float *)vol, w.nbricks); \
    }
    if (K == 3) { if (mode == 0) IP_BIN_ACC(3, 0) else if (mode == 1) IP_BIN_ACC(3, 1) else IP_BIN_ACC(3, 2) }
    else        { if (mode == 0) IP_BIN_ACC(2, 0) else if (mode == 1) IP_BIN_ACC(2, 1) else IP_BIN_ACC(2, 2) }
    e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

} // namespace ip

#ifdef IP_PROF
#define IP_PROF_NAME3(s) interpol_debug_prof_binned_##s
#define IP_PROF_NAME2(s) IP_PROF_NAME3(s)
extern "C" __attribute__((visibility("default"))) int IP_PROF_NAME2(IP_TSFX)(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::sorted::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::sorted::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
