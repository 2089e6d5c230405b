// ===========================================================================
// ops_tiled.hip -- LDS-tiled fast paths: 2-D / 3-D, f32 / bf16 / f16 storage (fp32
// math), per-dim spline orders 0..7, any boundary / extrapolation mode.
//   pull, grad            (gather)
//   push, count           (scatter)
//   pull / push backward  (fused: scatter or gather of the incoming gradient + grid gradient)
//
// Why: the generic kernels move every tap through the vector memory path; with
// an arbitrary deformation each lane of a wave touches its own cache line, and a
// scattered device-scope float atomic costs one fabric transaction (measured:
// ~21 G atomics/s).  At BASELINE config 2 that is 12 ms pull / 400 ms push
// (~1 % / 0.06 % of the HBM roofline).
//
// Here one (persistent) workgroup owns a TILE of TX x TY x TZ sample points at a time:
//   1. Box::build: a block-wide min/max of the first-tap indices gives the bounding box of
//      all stencil supports, clamped to what fits in LDS; boundary tables (wrapped offset and
//      sign per box row / column / slice); classification of the samples;
//   2. gather: the box is staged global -> LDS with the boundary condition ALREADY APPLIED:
//      the tap loop needs no index wrapping at all and reads LDS, separable FMA accumulation.
//      pull2_tiled keeps TWO channels per 8-byte slot (one ds_read2_b64 per two z-taps of both);
//   3. scatter: contributions are accumulated in the LDS box in FIXED POINT (ds_add_f32 retires
//      0.33 lanes/clk/CU on gfx950, ds_add_u32 5.7, ds_add_u64 4.6 -- tools/microbench/
//      lds_atomics.hip): two channels packed in one ds_add_u64 when the sample density bounds
//      the sums (scatter_pair), else 32-bit or 64-bit single-channel modes (scatter_channel);
//      each touched slot is then flushed with ONE coalesced global atomic instead of (K+1)^D
//      scattered ones;
//   4. an 8-byte box does not fit LDS for rough deformations: two passes split by the PARITY of
//      the box row x (every sample has taps in both passes, no lane idles); one pass when it fits;
//   5. samples whose support leaves the staged box (large local deformation) go to a per-tile
//      list handled tap-parallel by whole waves (lane = tap); tiles whose list overflows
//      (expanding deformations) are processed the same way entirely.
// Per-sample quantities (floor index, fraction) are recomputed from the coordinate grid in
// every phase instead of being held in registers (register budget of a 1024-thread block: the
// kernels sit at 128 VGPRs, and variants -- coordinate modes Cfg::GM, push with count -- are
// compile-time copies for that reason).
//
// Numerical definition: reference interpol/nd.py:80-143 (pull), 146-213 (push),
// 216-288 (grad), pushpull.py:237-258 (pull backward); iso1.py for all-linear;
// weights splines.py:30-139; bounds bounds.py:30-89.  Parity with the generic
// kernels / oracle is tested in tests/test_hip_parity.py.
// ===========================================================================
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include "tile_common.hpp"
#include <type_traits>

namespace ip {
namespace tiled {

constexpr int SLOWCAP = 512;                   // out-of-box samples handled tap-parallel per tile
constexpr int ROWN = 17 * 33;                  // box rows (x, y) resident in one 8-byte pass
constexpr int TABN = 72;                       // max box extent along one dim

// Tile configuration.  Kernel dims are always (x, y, z); a 2-D problem (D = 2) uses
// a degenerate x (one row, one tap, weight 1) and maps (y, z) to problem dims (0, 1).
// T = storage type of images (float, bf16_t, f16_t; math is always fp32),
// K = spline order (ISO: every dim has order K; !ISO: per-dim runtime orders <= K, taps
// beyond a dim's order are predicated off with wave-uniform tests).
// Phase profiling aid (tools/phase_prof.py): build this TU with -DIP_PROF and every
// prof_mark(i) adds the cycles since the previous mark (thread 0, after a block barrier) to
// g_prof[i].  Compiled out otherwise.
#ifdef IP_PROF
__device__ unsigned long long g_prof[16];
#endif
__device__ __forceinline__ void prof_mark(int i)
{
#ifdef IP_PROF
    __shared__ unsigned long long t0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long n = clock64();
        if (i >= 0) atomicAdd(&g_prof[i], n - t0);
        t0 = n;
    }
#endif
}

#ifndef IP_LDS_PAD
#define IP_LDS_PAD 0
#endif
template <typename T_, int K_, bool ISO_, int D_, int TX_, int TY_, int TZ_, int NT_, int PZ_, int GM_ = 0>
struct Cfg {
    using T = T_;
    static constexpr bool ISO = ISO_;
    static constexpr int K = K_, D = D_, TX = TX_, TY = TY_, TZ = TZ_, NT = NT_, PZ = PZ_;
    // Coordinate mode.  0: dense (B,*out,D) grid -- the hot kernels, nothing else in their way
    // (they sit at the 128-VGPR limit: folding the other modes in at run time cost 0.3 ms of 2.5);
    // 1: separable lattice (INTERPOL_FLAG_SEPARABLE_GRID), 2: displacement field
    // (INTERPOL_FLAG_DISPLACEMENT) -- own instantiations of the forward kernels, see launch_*.
    static constexpr int GM = GM_;
    template <int G> using Mode = Cfg<T_, K_, ISO_, D_, TX_, TY_, TZ_, NT_, PZ_, G>;
    // LDS row stride in slots: odd, so that rows (x, y) of the box start in different banks
    static constexpr int PS = PZ_ + IP_LDS_PAD;
    static constexpr int NS = TX * TY * TZ;            // samples per tile
    static constexpr int VPT = NS / NT;                 // samples per thread
    static constexpr int XSTEP = NT / (TY * TZ);        // x distance between a thread's samples
    static constexpr int KX = D == 3 ? K : 0;           // taps - 1 along x
    static constexpr int CAPZ = PZ;
    static constexpr int CAPX = D == 3 ? 33 : 1;
    static constexpr int CAPY = D == 3 ? 33 : TABN;
    // floats of LDS for the box: 3-D 34*33*32 (143616 B of the 160 KiB); 2-D twice the
    // gather box so that the 64-bit scatter needs a single pass
    static constexpr int BOXF = D == 3 ? (CAPX + 1) * CAPY * PS : 2 * CAPY * PS;   // CAPX + 1: two half boxes of 8-byte slots
    static_assert(NS % NT == 0 && NT % (TY * TZ) == 0, "tile / thread mismatch");
    static_assert(NT % PZ == 0, "staging maps z to tid % PZ");
    // problem dim of kernel dim d (-1: degenerate)
    __host__ __device__ static constexpr int pd(int d) { return d - (3 - D); }
};

// Fixed part of the LDS image; the box (C::BOXF floats) follows it.
struct Smem {
    int   taboff[3][TABN];     // wrapped lattice offset (elements) of box row / column / slice
    float tabsgn[3][TABN];     // boundary sign of the same
    int   lo[3], hi[3];        // block reductions
    int   nslow;
    int   dmax;                // max number of fast samples sharing one first-tap cell (scatter kernels)
    int   cmax[8];             // per-channel max |source| of the tile (float bits), first 8 channels
    unsigned short slow[SLOWCAP];
    int2  rowtab[ROWN];        // per resident box row (x, y): { taboff_x + taboff_y, bits of tabsgn_x * tabsgn_y }
    float box[1];              // really C::BOXF floats (dynamic LDS)
};
template <typename C> constexpr size_t smem_bytes() { return sizeof(Smem) + sizeof(float) * (C::BOXF - 1); }

// esz: size in bytes of one element of the indexed lattice as the kernel addresses it
template <typename C>
__device__ __forceinline__ Lattice make_lattice(const KParams &p, int esz)
{
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        constexpr int dummy = 0;
        const int q = C::pd(d) < 0 ? dummy : C::pd(d);
        if (C::pd(d) >= 0) { L.bound[d] = p.bound[q]; L.n[d] = p.vol_n[q]; L.ss[d] = p.vol_ss[q] / esz; L.k[d] = C::ISO ? C::K : p.order[q]; }
        else               { L.bound[d] = 1; L.n[d] = 1; L.ss[d] = 0; L.k[d] = 0; }
    }
    L.lin = (C::ISO && C::K == 1 && p.mode == MODE_ISO1);
    return L;
}

// (call as tap_weight<C>(...) inside the kernels)
template <typename C>
__device__ __forceinline__ float tap_weight(const Lattice &L, float gx_, float gy_, float gz_, int tap, int *off_out, float *grads)
{
    return tap_weight_t<C::ISO ? C::KX : -1, C::ISO ? C::K : -1>(L, gx_, gy_, gz_, tap, off_out, grads);
}


// ---------------------------------------------------------------------------
// Samples, tile geometry, bounding box.
// ---------------------------------------------------------------------------
template <typename C>
struct Sample {
    bool  valid, inb;          // inside the sample grid / extrapolation mask (nd.py:10-27)
    int   i0[3];               // first tap (unwrapped lattice index) per kernel dim
    float t[3];                // stencil coordinate (nd.py:46)
    int64_t o;                 // linear index of the sample in its batch item
};

struct TileGeom {
    int gx, gy, gz;            // sample grid extents (gx = 1 for 2-D)
    int ox0, oy0, oz0;         // tile origin
};

// Stretched beyond the box (more than 1/8 of the tile's samples outside it, smooth coordinates): the tile is handed back to the generic kernel
// of the operator (defer.hip) instead of crawling through the per-thread fallback.  Block-uniform.  Call after Box::build.
template <typename C>
__device__ __forceinline__ bool hand_back(const DeferArgs &defer, int work, int64_t b, const TileGeom &g, int nslow, const KParams &p,
                                          const float *__restrict__ grid, Smem &sm, bool scatter = false)
{
    if (!defer.flag) return false;
    // gathers: 1/8 of the samples outside the box; scatters: 3/16 -- the generic scatter pays 64 global atomics per sample where
    // the tile pays ~4; a stride of 2 (23 % outside) is still worth handing back (measured: tools/handback_sweep.py, profiles/r02_handback.txt)
    bool hb = nslow > (((scatter ? 3 * C::NS : 2 * C::NS) / 16) << ((p.dbg >> 9) & 7));
    if (hb) hb = tile_smooth(p, grid, b, C::D, g.ox0, g.oy0, g.oz0, C::TX, C::TY, C::TZ, g.gx, g.gy, g.gz, sm.hi);
    if (hb && threadIdx.x == 0) defer_mark(defer, work, tile_desc(b, g.ox0 / C::TX, g.oy0 / C::TY, g.oz0 / C::TZ));
    hb = hb && defer.desc != nullptr;
    if (hb) __syncthreads();                           // everyone has read sm.nslow before the next tile's build resets it
    return hb;
}

template <typename C>
__device__ __forceinline__ void sample_pos(const TileGeom &g, int tid, int v, int &ox, int &oy, int &oz)
{
    ox = g.ox0 + tid / (C::TZ * C::TY) + C::XSTEP * v;
    oy = g.oy0 + (tid / C::TZ) % C::TY;
    oz = g.oz0 + tid % C::TZ;
}

// Coordinates of sample v of this thread.  The loads are UNCONDITIONAL, from a position
// clamped into the sample grid: a branch around them would keep the compiler from batching
// the loads of a thread's samples (one exposed HBM round trip per sample instead of one per
// tile; measured 6.7 us -> of Box::build per 16^3 tile).  Returns whether the sample exists;
// `o` is the linear index of the (clamped) position in its batch item.
template <typename C>
__device__ __forceinline__ bool load_coords(const KParams &p, const float *__restrict__ grid, int64_t b, const TileGeom &g,
                                            int tid, int v, float *x, int64_t &o)
{
    int ox, oy, oz;
    sample_pos<C>(g, tid, v, ox, oy, oz);
    const bool valid = ox < g.gx && oy < g.gy && oz < g.gz;
    ox = ox < g.gx ? ox : g.gx - 1; oy = oy < g.gy ? oy : g.gy - 1; oz = oz < g.gz ? oz : g.gz - 1;
    o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
    if constexpr (C::GM == 1) {
        // tensor-product coordinates: the vectors lin_x | lin_y | lin_z back to back
        x[0] = C::D == 3 ? grid[ox] : 0.f;
        x[1] = grid[(C::D == 3 ? g.gx : 0) + oy];
        x[2] = grid[(C::D == 3 ? g.gx : 0) + g.gy + oz];
    } else {
        const float *gp = grid + b * p.grid_sb + o * C::D;
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = C::pd(d) >= 0 ? gp[C::pd(d) < 0 ? 0 : C::pd(d)] : 0.f;
        if constexpr (C::GM == 2) {                        // displacement field: add the identity lattice
            if (C::D == 3) x[0] += (float)ox;
            x[1] += (float)oy;
            x[2] += (float)oz;
        }
    }
    return valid;
}

template <typename C>
__device__ __forceinline__ Sample<C> load_sample(const KParams &p, const float *__restrict__ grid, int64_t b,
                                                 const TileGeom &g, int tid, int v)
{
    // per-dim order (compile-time for ISO tiles)
    const int kd[3] = { C::pd(0) < 0 ? 0 : (C::ISO ? C::K : p.order[C::pd(0) < 0 ? 0 : C::pd(0)]),
                        C::ISO ? C::K : p.order[C::pd(1) < 0 ? 0 : C::pd(1)],
                        C::ISO ? C::K : p.order[C::pd(2) < 0 ? 0 : C::pd(2)] };
    float x[3];
    Sample<C> s;
    s.valid = load_coords<C>(p, grid, b, g, tid, v, x, s.o);
    s.inb = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) { s.i0[d] = 0; s.t[d] = 0.f; }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (C::pd(d) < 0) continue;
        if (p.extrapolate != 1) s.inb = s.inb && x[d] > p.mask_lo_f && x[d] < p.mask_hi_f[C::pd(d) < 0 ? 0 : C::pd(d)];
        int i0; float t;
        split(kd[d], x[d], i0, t);
        s.i0[d] = s.valid ? i0 : 0;
        s.t[d] = s.valid ? t : 0.f;
    }
    return s;
}

template <typename C>
struct Box {
    int lo[3], S[3];

    // max |src_at(c, .)| over the tile's samples for channels c0 and c0 + 1 (< nc) -> sm.cmax.
    // Unconditional loads at clamped positions (see load_coords); NaN sticks.
    template <typename SrcAt>
    __device__ __forceinline__ static void channel_maxima(const TileGeom &g, Smem &sm, unsigned validmask, SrcAt src_at, int c0, int nc)
    {
        const int tid = threadIdx.x;
        const int c1 = c0 + 1 < nc ? c0 + 1 : c0;
        float a[2][C::VPT];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            int ox, oy, oz;
            sample_pos<C>(g, tid, v, ox, oy, oz);
            ox = ox < g.gx ? ox : g.gx - 1; oy = oy < g.gy ? oy : g.gy - 1; oz = oz < g.gz ? oz : g.gz - 1;
            const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
            a[0][v] = src_at(c0, o);
            a[1][v] = src_at(c1, o);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float amax = 0.f;
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                const float av = ((validmask >> v) & 1) ? __builtin_fabsf(a[h][v]) : 0.f;
                amax = (av > amax || av != av) ? av : amax;
            }
            const int bits = wave_max(__float_as_int(amax));   // non-negative floats (and NaN) order like ints
            if ((tid & 63) == 0 && bits != 0) atomicMax(&sm.cmax[h ? c1 : c0], bits);
        }
    }

    // Bounding box of the tile + classification of this thread's samples, from ONE read of
    // the coordinates: bit v of the result = sample v is "fast" (support inside the box);
    // the other valid samples are appended to the block's slow list.
    // DENSITY (scatter kernels): also leaves sm.dmax, the per-channel source maxima sm.cmax[c]
    // (c < min(C, 8), from `src_at(c, o)`) and an all-zero box behind.
    template <bool DENSITY = false, typename SrcAt = int>
    __device__ __forceinline__ unsigned build(const KParams &p, const Lattice &L, const float *__restrict__ grid, int64_t b,
                                              const TileGeom &g, Smem &sm, SrcAt src_at = 0, bool box_clean = false, int extra_channels = 0)
    {
        const int tid = threadIdx.x;
        if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        if (tid == 0) { sm.nslow = 0; sm.dmax = 0; }
        if (tid < 8) sm.cmax[tid] = 0;
        __syncthreads();
        int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
        int i0[C::VPT][3];
        unsigned validmask = 0;
        Sample<C> smp[C::VPT];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            smp[v] = load_sample<C>(p, grid, b, g, tid, v);
            if (smp[v].valid) validmask |= 1u << v;
        }
        // per-channel maxima of the (unmasked) sources -> fixed-point scales.  The loads of the
        // first two channels are issued here, right behind the coordinate loads, so that the tile
        // pays ONE exposed HBM round trip for both.
        if constexpr (DENSITY && !std::is_same<SrcAt, int>::value)
            channel_maxima(g, sm, validmask, src_at, 0, p.C + extra_channels < 8 ? p.C + extra_channels : 8);
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const Sample<C> &s = smp[v];
#pragma unroll
            for (int d = 0; d < 3; ++d) i0[v][d] = s.i0[d];
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
        if (DENSITY) prof_mark(9);
        const int cap[3] = { C::CAPX, C::CAPY, C::CAPZ };
        const int kd[3] = { L.k[0], L.k[1], L.k[2] };
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int l = sm.lo[d], h = sm.hi[d] + kd[d];      // supports span [l, h]
            if (h < l) { l = 0; h = 0; }                  // tile without valid samples
            int sz = h - l + 1;
            if (sz > cap[d]) { l += (sz - cap[d]) / 2; sz = cap[d]; }   // keep the centre; the rest goes to the slow list
            lo[d] = l; S[d] = sz;
        }
        // boundary tables: box slot -> wrapped lattice offset and sign (bounds.py:30-89)
        // (one pair of waves per dim, so that the three out-of-line wraps run side by side)
        {
            const int d = tid >> 7, slot = tid & 127;
            const int Sd = d == 0 ? S[0] : d == 1 ? S[1] : S[2];
            if (d < 3 && slot < Sd) {
                const int bd = d == 0 ? L.bound[0] : d == 1 ? L.bound[1] : L.bound[2];
                const int ld = d == 0 ? lo[0] : d == 1 ? lo[1] : lo[2];
                const int nd = d == 0 ? L.n[0] : d == 1 ? L.n[1] : L.n[2];
                const int sd = d == 0 ? L.ss[0] : d == 1 ? L.ss[1] : L.ss[2];
                const long long pk = wrap_outofline(bd, ld + slot, nd);
                sm.taboff[d][slot] = (int)(pk & 0xffffffffll) * sd;
                sm.tabsgn[d][slot] = (float)(int)(pk >> 32);
            }
        }
        // classification
        unsigned fastmask = 0;
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            if (!((validmask >> v) & 1)) continue;
            bool in = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) in = in && (i0[v][d] >= lo[d]) && (i0[v][d] + kd[d] < lo[d] + S[d]);
            if (in) fastmask |= 1u << v;
            else {
                const int slot = atomicAdd(&sm.nslow, 1);
                if (slot < SLOWCAP) sm.slow[slot] = (unsigned short)(tid * C::VPT + v);
            }
        }
        __syncthreads();
        if (DENSITY) prof_mark(10);
        if (DENSITY) {
            // Sample density: how many fast samples share one first-tap cell.  It bounds how
            // many contributions any lattice point of the box can receive, which is what lets
            // the scatter accumulate in 32-bit fixed point (see scatter_channel).  Only the
            // cells that hold samples are touched (count, read back, re-zero): the box is
            // all-zero on entry when the previous tile left it so (box_clean), and on exit.
            unsigned *cnt = reinterpret_cast<unsigned *>(sm.box);
            if (!box_clean) {
                const int nslots = C::BOXF;
                for (int e = tid; e < nslots; e += C::NT) cnt[e] = 0u;
                __syncthreads();
            }
            int cell[C::VPT];
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                cell[v] = ((i0[v][0] - lo[0]) * S[1] + (i0[v][1] - lo[1])) * C::PS + (i0[v][2] - lo[2]);
                if ((fastmask >> v) & 1) atomicAdd(&cnt[cell[v]], 1u);
            }
            __syncthreads();
            int m = 0;
#pragma unroll
            for (int v = 0; v < C::VPT; ++v)
                if ((fastmask >> v) & 1) { const int cv = (int)cnt[cell[v]]; m = cv > m ? cv : m; }
            m = wave_max(m);
            if ((tid & 63) == 0 && m > 0) atomicMax(&sm.dmax, m);
            __syncthreads();
#pragma unroll
            for (int v = 0; v < C::VPT; ++v)
                if ((fastmask >> v) & 1) cnt[cell[v]] = 0u;
            prof_mark(11);
            // per-channel maxima of the channels beyond the first two (those were reduced above)
            if constexpr (!std::is_same<SrcAt, int>::value) {
                const int nc = p.C + extra_channels < 8 ? p.C + extra_channels : 8;
                for (int c = 2; c < nc; c += 2) channel_maxima(g, sm, validmask, src_at, c, nc);
            }
            __syncthreads();
        }
        return fastmask;
    }

    __device__ __forceinline__ int base(const Sample<C> &s) const
    {
        return ((s.i0[0] - lo[0]) * S[1] + (s.i0[1] - lo[1])) * C::PS + (s.i0[2] - lo[2]);
    }
};

// Decode an entry of the slow list into the sample's linear index / coordinates.
template <typename C>
__device__ __forceinline__ int64_t slow_sample(const TileGeom &g, int code, const KParams &p, const float *__restrict__ grid,
                                               int64_t b, float *x)
{
    const int stid = code / C::VPT, sv = code % C::VPT;
    int ox, oy, oz;
    sample_pos<C>(g, stid, sv, ox, oy, oz);
    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
    if (C::GM == 1) {
        x[0] = C::D == 3 ? grid[ox] : 0.f;
        x[1] = grid[(C::D == 3 ? g.gx : 0) + oy];
        x[2] = grid[(C::D == 3 ? g.gx : 0) + g.gy + oz];
        return o;
    }
    const float *gp = grid + b * p.grid_sb + o * C::D;
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = C::pd(d) >= 0 ? gp[C::pd(d) < 0 ? 0 : C::pd(d)] : 0.f;
    if (C::GM == 2) {
        if (C::D == 3) x[0] += (float)ox;
        x[1] += (float)oy;
        x[2] += (float)oz;
    }
    return o;
}

template <typename C>
__device__ __forceinline__ bool coords_inb(const KParams &p, const float *x)
{
    bool in = true;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (C::pd(d) >= 0) in = in && x[d] > p.mask_lo_f && x[d] < p.mask_hi_f[C::pd(d) < 0 ? 0 : C::pd(d)];
    return in;
}

// Stage one channel of the box: LDS[x][y][z] = sign * vol[wrapped(x,y,z)].
// Memory-level parallelism matters here: with one block per CU a rolled loop keeps a
// single 4-byte load in flight per thread (4 KB per CU: ~4 GB/s per CU, measured 6 us
// for a 27 KB box); the loop is unrolled U-fold with all loads issued before the first
// LDS write.  (x, y) of a flattened row come from an exact float reciprocal.
template <typename C>
__device__ __forceinline__ void stage_box(const typename C::T *__restrict__ vc, const int *S, Smem &sm)
{
    constexpr int U = 8;
    constexpr int RSTEP = C::NT / C::PZ;
    const int tid = threadIdx.x;
    const int z = tid % C::PZ;
    const bool zin = z < S[2];
    const int oz = zin ? sm.taboff[2][z] : 0;
    const float sz = zin ? sm.tabsgn[2][z] : 0.f;
    const int rows = S[0] * S[1];
    const float inv_sy = 1.f / (float)S[1];
    for (int r0 = tid / C::PZ; r0 < rows; r0 += RSTEP * U) {
        float v[U], sg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * RSTEP;
            const bool on = zin && r < rows;
            const int x = (int)(((float)r + 0.5f) * inv_sy);
            const int y = r - x * S[1];
            sg[u] = on ? sm.tabsgn[0][on ? x : 0] * sm.tabsgn[1][on ? y : 0] * sz : 0.f;
            v[u] = on ? Cvt<float, typename C::T>::ld(vc[sm.taboff[0][x] + sm.taboff[1][y] + oz]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * RSTEP;
            if (zin && r < rows) sm.box[r * C::PS + z] = v[u] * sg[u];
        }
    }
}

template <typename C>
__device__ __forceinline__ TileGeom tile_geom(int tile, int gx, int gy, int gz, int nty, int ntz)
{
    const int tzi = tile % ntz; tile /= ntz;
    const int tyi = tile % nty; const int txi = tile / nty;
    return TileGeom{ gx, gy, gz, txi * C::TX, tyi * C::TY, tzi * C::TZ };
}

// ---------------------------------------------------------------------------
// Gather from the staged box: value (GRAD = false) or the three partial
// derivatives (GRAD = true).  out[0] = value, out[1..3] = d/dx, d/dy, d/dz.
// ---------------------------------------------------------------------------
template <typename C, bool GRAD>
__device__ __forceinline__ void gather_box(const Smem &sm, const Box<C> &box, const Sample<C> &s, const Lattice &L, float *out)
{
    constexpr int K = C::K, KX = C::KX;
    const int lin = L.lin;
    float wx[KX + 1], wy[K + 1], wz[K + 1], gx_[KX + 1], gy_[K + 1], gz_[K + 1];
    if (KX > 0) weights<KX>(lin, L.k[0], s.t[0], wx); else wx[0] = 1.f;
    weights<K>(lin, L.k[1], s.t[1], wy); weights<K>(lin, L.k[2], s.t[2], wz);
    if (GRAD) {
        if (KX > 0) wgrads<KX>(lin, L.k[0], s.t[0], gx_); else gx_[0] = 0.f;
        wgrads<K>(lin, L.k[1], s.t[1], gy_); wgrads<K>(lin, L.k[2], s.t[2], gz_);
    }
    const float *bp = sm.box + box.base(s);
    float a = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int i = 0; i <= KX; ++i) {
        if (!C::ISO && i > L.k[0]) continue;          // wave-uniform
        float pW = 0.f, pGy = 0.f, pGz = 0.f;
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            if (!C::ISO && j > L.k[1]) continue;
            const float *rp = bp + (i * box.S[1] + j) * C::PS;
            float rW = 0.f, rG = 0.f;
#pragma unroll
            for (int k = 0; k <= K; ++k) {
                if (!C::ISO && k > L.k[2]) continue;
                const float v = rp[k];
                rW = __builtin_fmaf(wz[k], v, rW);
                if (GRAD) rG = __builtin_fmaf(gz_[k], v, rG);
            }
            pW = __builtin_fmaf(wy[j], rW, pW);
            if (GRAD) { pGy = __builtin_fmaf(gy_[j], rW, pGy); pGz = __builtin_fmaf(wy[j], rG, pGz); }
        }
        a = __builtin_fmaf(wx[i], pW, a);
        if (GRAD) {
            ax = __builtin_fmaf(gx_[i], pW, ax);
            ay = __builtin_fmaf(wx[i], pGy, ay);
            az = __builtin_fmaf(wx[i], pGz, az);
        }
    }
    out[0] = a;
    if (GRAD) { out[1] = ax; out[2] = ay; out[3] = az; }
}

// ---------------------------------------------------------------------------
// pull (GRAD = false): val[b,c,o]     = mask * sum w vol
// grad (GRAD = true) : val[b,c,o,d]   = mask * sum (g_d prod w) vol
// ---------------------------------------------------------------------------
template <typename C, bool GRAD>
__global__ __launch_bounds__(C::NT) void gather_tiled(KParams p, const typename C::T *__restrict__ vol, const float *__restrict__ grid,
                                                      typename C::T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch, DeferArgs defer)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    if (p.gate_n == -3 && p.verdict && *p.verdict != 1) return;   // trilinear grid_grad (abi.hip: interpol_grad_ws): the tiles run on the probe's verdict 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    constexpr int D = C::D;
    using T = typename C::T;
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    // persistent blocks: one per CU, striding over the (tile, batch item) work list
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);

    Box<C> box;
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;

    for (int c = 0; c < p.C; ++c) {
        const T *vc = vol + b * p.vol_sb + c * p.vol_sc;
        T *oc = val + b * p.val_sb + c * p.val_sc;
        __syncthreads();                               // previous channel's readers are done
        if (!(p.dbg & 1)) stage_box<C>(vc, box.S, sm);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            if (p.dbg & 2) continue;
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;   // invalid, or waiting in the slow list
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r[4] = { 0.f, 0.f, 0.f, 0.f };
            if (fast) {
                gather_box<C, GRAD>(sm, box, s, L, r);
            } else {
                // slow list overflowed (pathological deformation): per-thread global gather
                if (!GRAD) r[0] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], -1);
                else {
                    if (D == 3) r[1] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 0);
                    r[2] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 1);
                    r[3] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 2);
                }
            }
            const float m = (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f;      // nd.py:139-140, 284-285
            if (!GRAD) oc[s.o] = Cvt<float, T>::st(r[0] * m);
            else {
#pragma unroll
                for (int d = 0; d < D; ++d) oc[s.o * D + d] = Cvt<float, T>::st(r[1 + (3 - D) + d] * m);
            }
        }
        // slow list: one wave per sample, lanes = taps
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a[4] = { 0.f, 0.f, 0.f, 0.f };
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off; float gr[3];
                    const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, GRAD ? gr : nullptr);
                    const float v = (t0 + lane < NTAP) ? Cvt<float, T>::ld(vc[off]) : 0.f;
                    if (!GRAD) a[0] += w * v;
                    else { a[1] += gr[0] * v; a[2] += gr[1] * v; a[3] += gr[2] * v; }
                }
                const float m = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : 1.f;
                if (!GRAD) { const float r = wave_sum(a[0]); if (lane == 0) oc[o] = Cvt<float, T>::st(r * m); }
                else {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float r = wave_sum(a[1 + (3 - D) + d]);
                        if (lane == 0) oc[o * D + d] = Cvt<float, T>::st(r * m);
                    }
                }
            }
        }
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// pull, two channels at a time (3-D, even channel count): the box holds channel PAIRS
// interleaved (8 bytes per slot) so that one ds_read_b64 feeds both channels -- the tap
// loop is LDS-issue bound, this halves its LDS instructions.  8-byte slots do not fit the
// whole box: it is staged in passes over slabs of x-rows and the per-sample partial sums
// stay in registers across the passes.
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void pull2_tiled(KParams p, const typename C::T *__restrict__ vol, const float *__restrict__ grid,
                                                     typename C::T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz,
                                                     int ntiles, int nbatch, DeferArgs defer)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    static_assert(C::D == 3, "pair mode is 3-D only");
    constexpr int K = C::K;
    constexpr int BOX64 = C::BOXF / 2;
    using T = typename C::T;
    float2 *box2 = reinterpret_cast<float2 *>(sm.box);
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    prof_mark(-1);
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;
    prof_mark(0);
    // Two passes per channel pair, split by the PARITY of the box row x (see scatter_pair):
    // LDS row r = xh * S_y + y of pass ps holds box row x = 2 xh + ps.  Every sample reads taps
    // in both passes (i = par, par + 2, ... with par = (x0 ^ ps) & 1): no lane idles.
    // A box small enough for one pass (smooth deformations: S ~ tile + K) is staged whole.
    static_assert(((C::CAPX + 1) / 2) * C::CAPY * C::PS <= BOX64, "half box must fit the 8-byte slots");
    const int xmul = box.S[0] * box.S[1] * C::PS <= BOX64 ? 1 : 2;

    for (int c = 0; c < p.C; c += 2) {
        const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
        const T *vc1 = vc0 + p.vol_sc;
        T *oc0 = val + b * p.val_sb + c * p.val_sc;
        T *oc1 = oc0 + p.val_sc;
        float acc[C::VPT][2];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) { acc[v][0] = 0.f; acc[v][1] = 0.f; }
        for (int ps = 0; ps < xmul; ++ps) {
            const int nxh = xmul == 1 ? box.S[0] : (box.S[0] - ps + 1) >> 1;
            const int r_n = nxh * box.S[1];
            __syncthreads();                           // previous pass's readers are done
            for (int r = tid; r < r_n; r += C::NT) {
                const int xh = r / box.S[1], y = r - xh * box.S[1];
                const int x = xmul * xh + ps;
                sm.rowtab[r] = make_int2(sm.taboff[0][x] + sm.taboff[1][y], __float_as_int(sm.tabsgn[0][x] * sm.tabsgn[1][y]));
            }
            __syncthreads();
            {   // stage the rows of this parity, both channels (unrolled for memory-level parallelism)
                constexpr int U = 4, RSTEP = C::NT / C::PZ;
                const int z = tid % C::PZ;
                const bool zin = z < box.S[2];
                const int oz = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                for (int r0 = tid / C::PZ; r0 < r_n; r0 += RSTEP * U) {
                    float v0[U], v1[U], sg[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const bool on = zin && r < r_n;
                        const int2 rt = sm.rowtab[r < r_n ? r : 0];
                        const int off = on ? rt.x + oz : 0;
                        sg[u] = on ? __int_as_float(rt.y) * sz : 0.f;
                        v0[u] = on ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                        v1[u] = on ? Cvt<float, T>::ld(vc1[off]) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        if (zin && r < r_n) box2[r * C::PS + z] = make_float2(v0[u] * sg[u], v1[u] * sg[u]);
                    }
                }
            }
            __syncthreads();
            prof_mark(5);
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                if (!((fastmask >> v) & 1)) continue;
                const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
                const int x0 = s.i0[0] - box.lo[0];
                const int par = xmul == 1 ? 0 : (x0 ^ ps) & 1;
                float wx[K + 2], wy[K + 1], wz[K + 1];
                weights<K>(L.lin, L.k[0], s.t[0], wx); weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
                wx[K + 1] = 0.f;
                const int r0 = (xmul == 1 ? x0 : (x0 + par) >> 1) * box.S[1] + (s.i0[1] - box.lo[1]);
                const float2 *bp = box2 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int ii = 0; ii <= K; ++ii) {
                    // tap i = par + xmul * ii of this sample lives in LDS row r0 + ii * S_y
                    if (par + xmul * ii > (C::ISO ? K : L.k[0])) continue;     // uniform but for the last i of even tap counts
                    const float w2 = par ? wx[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : wx[2 * ii <= K ? 2 * ii : K + 1];
                    const float wxi = xmul == 1 ? wx[ii] : w2;
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        if (!C::ISO && j > L.k[1]) continue;
                        const volatile __attribute__((address_space(3))) unsigned long long *rp =       // (single ds_read_b64: see pull1s_tiled)
                            (const volatile __attribute__((address_space(3))) unsigned long long *)(bp + (ii * box.S[1] + j) * C::PS);
                        float q0 = 0.f, q1 = 0.f;
#pragma unroll
                        for (int k = 0; k <= K; ++k) {
                            if (!C::ISO && k > L.k[2]) continue;
                            const unsigned long long raw = rp[k];          // ds_read_b64: both channels
                            const float2 t2 = make_float2(__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32)));
                            q0 = __builtin_fmaf(wz[k], t2.x, q0);
                            q1 = __builtin_fmaf(wz[k], t2.y, q1);
                        }
                        p0 = __builtin_fmaf(wy[j], q0, p0);
                        p1 = __builtin_fmaf(wy[j], q1, p1);
                    }
                    acc[v][0] = __builtin_fmaf(wxi, p0, acc[v][0]);
                    acc[v][1] = __builtin_fmaf(wxi, p1, acc[v][1]);
                }
            }
            prof_mark(6);
        }
        // outputs of the fast samples
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r0v = acc[v][0], r1v = acc[v][1];
            if (!fast) {   // slow list overflowed: per-thread global gather
                r0v = gather_one_thread<T>(L, vc0, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], -1);
                r1v = gather_one_thread<T>(L, vc1, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], -1);
            }
            const float m = (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f;
            oc0[s.o] = Cvt<float, T>::st(r0v * m);
            oc1[s.o] = Cvt<float, T>::st(r1v * m);
        }
        // slow list: one wave per sample, lanes = taps
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a0 = 0.f, a1 = 0.f;
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off;
                    const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, nullptr);
                    if (t0 + lane < NTAP) { a0 += w * Cvt<float, T>::ld(vc0[off]); a1 += w * Cvt<float, T>::ld(vc1[off]); }
                }
                const float m = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : 1.f;
                a0 = wave_sum(a0); a1 = wave_sum(a1);
                if (lane == 0) { oc0[o] = Cvt<float, T>::st(a0 * m); oc1[o] = Cvt<float, T>::st(a1 * m); }
            }
        }
        prof_mark(7);
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Single-channel pull with SHIFTED PAIRS: slot z of the LDS box holds (v[z], v[z + 1]), so one
// ds_read2_b64 (slots z0, z0 + 2) delivers four consecutive z-taps of a row -- half the LDS
// instructions of the 4-byte box (the cost of a random LDS access is per instruction, not per
// byte: tools/microbench/lds_banks.hip).  Same structure as pull2_tiled otherwise (8-byte slots,
// parity passes).  Used for a single channel and for the last channel of an odd count.
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void pull1s_tiled(KParams p, const typename C::T *__restrict__ vol, const float *__restrict__ grid,
                                                     typename C::T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz,
                                                     int ntiles, int nbatch, DeferArgs defer)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    static_assert(C::D == 3 && C::ISO, "shifted-pair mode: 3-D, one compile-time order");
    constexpr int K = C::K;
    constexpr int BOX64 = C::BOXF / 2;
    using T = typename C::T;
    float2 *box2 = reinterpret_cast<float2 *>(sm.box);
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    prof_mark(-1);
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;
    prof_mark(0);
    // Two passes per channel pair, split by the PARITY of the box row x (see scatter_pair):
    // LDS row r = xh * S_y + y of pass ps holds box row x = 2 xh + ps.  Every sample reads taps
    // in both passes (i = par, par + 2, ... with par = (x0 ^ ps) & 1): no lane idles.
    // A box small enough for one pass (smooth deformations: S ~ tile + K) is staged whole.
    static_assert(((C::CAPX + 1) / 2) * C::CAPY * C::PS <= BOX64, "half box must fit the 8-byte slots");
    const int xmul = box.S[0] * box.S[1] * C::PS <= BOX64 ? 1 : 2;

    for (int c = 0; c < p.C; ++c) {
        const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
        T *oc0 = val + b * p.val_sb + c * p.val_sc;
        float acc[C::VPT];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) acc[v] = 0.f;
        for (int ps = 0; ps < xmul; ++ps) {
            const int nxh = xmul == 1 ? box.S[0] : (box.S[0] - ps + 1) >> 1;
            const int r_n = nxh * box.S[1];
            __syncthreads();                           // previous pass's readers are done
            for (int r = tid; r < r_n; r += C::NT) {
                const int xh = r / box.S[1], y = r - xh * box.S[1];
                const int x = xmul * xh + ps;
                sm.rowtab[r] = make_int2(sm.taboff[0][x] + sm.taboff[1][y], __float_as_int(sm.tabsgn[0][x] * sm.tabsgn[1][y]));
            }
            __syncthreads();
            {   // stage the rows of this parity: slot z = (v[z], v[z+1]), the neighbour from lane + 1
                constexpr int U = 4, RSTEP = C::NT / C::PZ;
                const int z = tid % C::PZ;
                const bool zin = z < box.S[2];
                const int oz = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                for (int r0 = tid / C::PZ; r0 < r_n; r0 += RSTEP * U) {
                    float v0[U], sg[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const bool on = zin && r < r_n;
                        const int2 rt = sm.rowtab[r < r_n ? r : 0];
                        const int off = on ? rt.x + oz : 0;
                        sg[u] = on ? __int_as_float(rt.y) * sz : 0.f;
                        v0[u] = on ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const float sv = v0[u] * sg[u];
                        const float nb = __shfl_down(sv, 1);           // z + 1 (same row: PZ divides 64); unused for the last z
                        if (zin && r < r_n) box2[r * C::PS + z] = make_float2(sv, nb);
                    }
                }
            }
            __syncthreads();
            prof_mark(5);
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                if (!((fastmask >> v) & 1)) continue;
                const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
                const int x0 = s.i0[0] - box.lo[0];
                const int par = xmul == 1 ? 0 : (x0 ^ ps) & 1;
                float wx[K + 2], wy[K + 1], wz[K + 1];
                weights<K>(L.lin, L.k[0], s.t[0], wx); weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
                wx[K + 1] = 0.f;
                const int r0 = (xmul == 1 ? x0 : (x0 + par) >> 1) * box.S[1] + (s.i0[1] - box.lo[1]);
                const float2 *bp = box2 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int ii = 0; ii <= K; ++ii) {
                    // tap i = par + xmul * ii of this sample lives in LDS row r0 + ii * S_y
                    if (par + xmul * ii > (C::ISO ? K : L.k[0])) continue;     // uniform but for the last i of even tap counts
                    const float w2 = par ? wx[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : wx[2 * ii <= K ? 2 * ii : K + 1];
                    const float wxi = xmul == 1 ? wx[ii] : w2;
                    float p0 = 0.f;
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        // (volatile LDS pointer: keeps single ds_read_b64 -- a compiler-merged ds_read2_b64 costs 25 clk per
                        //  wave instruction against 2 x 7.6 with random lane bases, tools/microbench/lds_gather.hip)
                        const volatile __attribute__((address_space(3))) unsigned long long *rp =
                            (const volatile __attribute__((address_space(3))) unsigned long long *)(bp + (ii * box.S[1] + j) * C::PS);
                        float q0 = 0.f;
#pragma unroll
                        for (int k = 0; k <= K; k += 2) {
                            const unsigned long long raw = rp[k];          // ds_read_b64: taps k and k + 1
                            const float2 t2 = make_float2(__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32)));
                            q0 = __builtin_fmaf(wz[k], t2.x, q0);
                            if (k + 1 <= K) q0 = __builtin_fmaf(wz[k + 1 <= K ? k + 1 : K], t2.y, q0);
                        }
                        p0 = __builtin_fmaf(wy[j], q0, p0);
                    }
                    acc[v] = __builtin_fmaf(wxi, p0, acc[v]);
                }
            }
            prof_mark(6);
        }
        // outputs of the fast samples
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r0v = acc[v];
            if (!fast)     // slow list overflowed: per-thread global gather
                r0v = gather_one_thread<T>(L, vc0, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], -1);
            const float m = (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f;
            oc0[s.o] = Cvt<float, T>::st(r0v * m);
        }
        // slow list: one wave per sample, lanes = taps
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a0 = 0.f;
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off;
                    const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, nullptr);
                    if (t0 + lane < NTAP) a0 += w * Cvt<float, T>::ld(vc0[off]);
                }
                const float m = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : 1.f;
                a0 = wave_sum(a0);
                if (lane == 0) oc0[o] = Cvt<float, T>::st(a0 * m);
            }
        }
        prof_mark(7);
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// grad with shifted pairs (see pull1s_tiled): val[b,c,o,d] = mask * sum (g_d prod w) vol.
// Used for the higher orders (one sample per thread: the extra accumulators fit).
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void grad1s_tiled(KParams p, const typename C::T *__restrict__ vol, const float *__restrict__ grid,
                                                     typename C::T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz,
                                                     int ntiles, int nbatch, DeferArgs defer)
{
    if (p.gate_n == -1 && p.gate && *p.gate == 1) return;   // a probe of the call gave it to the bricks of the image (interpol_pull_ws / interpol_grad_ws)
    if (p.gate_n == -3 && p.verdict && *p.verdict != 1) return;   // trilinear grid_grad (abi.hip: interpol_grad_ws): the tiles run on the probe's verdict 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    static_assert(C::D == 3 && C::ISO, "shifted-pair mode: 3-D, one compile-time order");
    constexpr int K = C::K;
    constexpr int BOX64 = C::BOXF / 2;
    using T = typename C::T;
    float2 *box2 = reinterpret_cast<float2 *>(sm.box);
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    prof_mark(-1);
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;
    prof_mark(0);
    // Two passes per channel pair, split by the PARITY of the box row x (see scatter_pair):
    // LDS row r = xh * S_y + y of pass ps holds box row x = 2 xh + ps.  Every sample reads taps
    // in both passes (i = par, par + 2, ... with par = (x0 ^ ps) & 1): no lane idles.
    // A box small enough for one pass (smooth deformations: S ~ tile + K) is staged whole.
    static_assert(((C::CAPX + 1) / 2) * C::CAPY * C::PS <= BOX64, "half box must fit the 8-byte slots");
    const int xmul = box.S[0] * box.S[1] * C::PS <= BOX64 ? 1 : 2;

    for (int c = 0; c < p.C; ++c) {
        const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
        T *oc0 = val + b * p.val_sb + c * p.val_sc;
        float acc[C::VPT][3];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) { acc[v][0] = 0.f; acc[v][1] = 0.f; acc[v][2] = 0.f; }
        for (int ps = 0; ps < xmul; ++ps) {
            const int nxh = xmul == 1 ? box.S[0] : (box.S[0] - ps + 1) >> 1;
            const int r_n = nxh * box.S[1];
            __syncthreads();                           // previous pass's readers are done
            for (int r = tid; r < r_n; r += C::NT) {
                const int xh = r / box.S[1], y = r - xh * box.S[1];
                const int x = xmul * xh + ps;
                sm.rowtab[r] = make_int2(sm.taboff[0][x] + sm.taboff[1][y], __float_as_int(sm.tabsgn[0][x] * sm.tabsgn[1][y]));
            }
            __syncthreads();
            {   // stage the rows of this parity: slot z = (v[z], v[z+1]), the neighbour from lane + 1
                constexpr int U = 4, RSTEP = C::NT / C::PZ;
                const int z = tid % C::PZ;
                const bool zin = z < box.S[2];
                const int oz = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                for (int r0 = tid / C::PZ; r0 < r_n; r0 += RSTEP * U) {
                    float v0[U], sg[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const bool on = zin && r < r_n;
                        const int2 rt = sm.rowtab[r < r_n ? r : 0];
                        const int off = on ? rt.x + oz : 0;
                        sg[u] = on ? __int_as_float(rt.y) * sz : 0.f;
                        v0[u] = on ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const float sv = v0[u] * sg[u];
                        const float nb = __shfl_down(sv, 1);           // z + 1 (same row: PZ divides 64); unused for the last z
                        if (zin && r < r_n) box2[r * C::PS + z] = make_float2(sv, nb);
                    }
                }
            }
            __syncthreads();
            prof_mark(5);
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                if (!((fastmask >> v) & 1)) continue;
                const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
                const int x0 = s.i0[0] - box.lo[0];
                const int par = xmul == 1 ? 0 : (x0 ^ ps) & 1;
                float wx[K + 2], wy[K + 1], wz[K + 1], gx_[K + 2], gy_[K + 1], gz_[K + 1];
                weights<K>(L.lin, L.k[0], s.t[0], wx); weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
                wgrads<K>(L.lin, L.k[0], s.t[0], gx_); wgrads<K>(L.lin, L.k[1], s.t[1], gy_); wgrads<K>(L.lin, L.k[2], s.t[2], gz_);
                wx[K + 1] = 0.f; gx_[K + 1] = 0.f;
                const int r0 = (xmul == 1 ? x0 : (x0 + par) >> 1) * box.S[1] + (s.i0[1] - box.lo[1]);
                const float2 *bp = box2 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int ii = 0; ii <= K; ++ii) {
                    // tap i = par + xmul * ii of this sample lives in LDS row r0 + ii * S_y
                    if (par + xmul * ii > (C::ISO ? K : L.k[0])) continue;     // uniform but for the last i of even tap counts
                    const float w2 = par ? wx[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : wx[2 * ii <= K ? 2 * ii : K + 1];
                    const float wxi = xmul == 1 ? wx[ii] : w2;
                    const float g2 = par ? gx_[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : gx_[2 * ii <= K ? 2 * ii : K + 1];
                    const float gxi = xmul == 1 ? gx_[ii] : g2;
                    float pW = 0.f, pGy = 0.f, pGz = 0.f;
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        const volatile __attribute__((address_space(3))) unsigned long long *rp =       // (single ds_read_b64: see pull1s_tiled)
                            (const volatile __attribute__((address_space(3))) unsigned long long *)(bp + (ii * box.S[1] + j) * C::PS);
                        float rW = 0.f, rG = 0.f;
#pragma unroll
                        for (int k = 0; k <= K; k += 2) {
                            const unsigned long long raw = rp[k];          // ds_read_b64: taps k and k + 1
                            const float2 t2 = make_float2(__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32)));
                            rW = __builtin_fmaf(wz[k], t2.x, rW);
                            rG = __builtin_fmaf(gz_[k], t2.x, rG);
                            if (k + 1 <= K) {
                                rW = __builtin_fmaf(wz[k + 1 <= K ? k + 1 : K], t2.y, rW);
                                rG = __builtin_fmaf(gz_[k + 1 <= K ? k + 1 : K], t2.y, rG);
                            }
                        }
                        pW = __builtin_fmaf(wy[j], rW, pW);
                        pGy = __builtin_fmaf(gy_[j], rW, pGy);
                        pGz = __builtin_fmaf(wy[j], rG, pGz);
                    }
                    acc[v][0] = __builtin_fmaf(gxi, pW, acc[v][0]);
                    acc[v][1] = __builtin_fmaf(wxi, pGy, acc[v][1]);
                    acc[v][2] = __builtin_fmaf(wxi, pGz, acc[v][2]);
                }
            }
            prof_mark(6);
        }
        // outputs of the fast samples
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r[3] = { acc[v][0], acc[v][1], acc[v][2] };
            if (!fast) {   // slow list overflowed: per-thread global gather
#pragma unroll
                for (int d = 0; d < 3; ++d) r[d] = gather_one_thread<T>(L, vc0, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], d);
            }
            const float m = (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) oc0[s.o * 3 + d] = Cvt<float, T>::st(r[d] * m);
        }
        // slow list: one wave per sample, lanes = taps
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a3[3] = { 0.f, 0.f, 0.f };
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off; float gr[3];
                    tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, gr);
                    const float vv = (t0 + lane < NTAP) ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                    a3[0] += gr[0] * vv; a3[1] += gr[1] * vv; a3[2] += gr[2] * vv;
                }
                const float m = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float rr = wave_sum(a3[d]);
                    if (lane == 0) oc0[o * 3 + d] = Cvt<float, T>::st(rr * m);
                }
            }
        }
        prof_mark(7);
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Grid gradient of pull (pushpull.py:256-257) with shifted pairs: grad1s_tiled with the
// contraction  ggrid[b,o,d] = mask * sum_c gout[b,c,o] * d/dx_d pull(vol)[b,c,o]  folded in.
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void gradc1s_tiled(KParams p, const typename C::T *__restrict__ gout, const typename C::T *__restrict__ vol,
                                                       const float *__restrict__ grid, float *__restrict__ ggrid,
                                                       int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch, DeferArgs defer)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    static_assert(C::D == 3 && C::ISO, "shifted-pair mode: 3-D, one compile-time order");
    constexpr int K = C::K;
    constexpr int BOX64 = C::BOXF / 2;
    using T = typename C::T;
    float2 *box2 = reinterpret_cast<float2 *>(sm.box);
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    prof_mark(-1);
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;
    prof_mark(0);
    // Two passes per channel pair, split by the PARITY of the box row x (see scatter_pair):
    // LDS row r = xh * S_y + y of pass ps holds box row x = 2 xh + ps.  Every sample reads taps
    // in both passes (i = par, par + 2, ... with par = (x0 ^ ps) & 1): no lane idles.
    // A box small enough for one pass (smooth deformations: S ~ tile + K) is staged whole.
    static_assert(((C::CAPX + 1) / 2) * C::CAPY * C::PS <= BOX64, "half box must fit the 8-byte slots");
    const int xmul = box.S[0] * box.S[1] * C::PS <= BOX64 ? 1 : 2;

    float gg[C::VPT][3];
#pragma unroll
    for (int v = 0; v < C::VPT; ++v) { gg[v][0] = 0.f; gg[v][1] = 0.f; gg[v][2] = 0.f; }
    for (int c = 0; c < p.C; ++c) {
        const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
        const T *gc = gout + b * p.val_sb + c * p.val_sc;
        float acc[C::VPT][3];
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) { acc[v][0] = 0.f; acc[v][1] = 0.f; acc[v][2] = 0.f; }
        for (int ps = 0; ps < xmul; ++ps) {
            const int nxh = xmul == 1 ? box.S[0] : (box.S[0] - ps + 1) >> 1;
            const int r_n = nxh * box.S[1];
            __syncthreads();                           // previous pass's readers are done
            for (int r = tid; r < r_n; r += C::NT) {
                const int xh = r / box.S[1], y = r - xh * box.S[1];
                const int x = xmul * xh + ps;
                sm.rowtab[r] = make_int2(sm.taboff[0][x] + sm.taboff[1][y], __float_as_int(sm.tabsgn[0][x] * sm.tabsgn[1][y]));
            }
            __syncthreads();
            {   // stage the rows of this parity: slot z = (v[z], v[z+1]), the neighbour from lane + 1
                constexpr int U = 4, RSTEP = C::NT / C::PZ;
                const int z = tid % C::PZ;
                const bool zin = z < box.S[2];
                const int oz = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                for (int r0 = tid / C::PZ; r0 < r_n; r0 += RSTEP * U) {
                    float v0[U], sg[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const bool on = zin && r < r_n;
                        const int2 rt = sm.rowtab[r < r_n ? r : 0];
                        const int off = on ? rt.x + oz : 0;
                        sg[u] = on ? __int_as_float(rt.y) * sz : 0.f;
                        v0[u] = on ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = r0 + u * RSTEP;
                        const float sv = v0[u] * sg[u];
                        const float nb = __shfl_down(sv, 1);           // z + 1 (same row: PZ divides 64); unused for the last z
                        if (zin && r < r_n) box2[r * C::PS + z] = make_float2(sv, nb);
                    }
                }
            }
            __syncthreads();
            prof_mark(5);
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                if (!((fastmask >> v) & 1)) continue;
                const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
                const int x0 = s.i0[0] - box.lo[0];
                const int par = xmul == 1 ? 0 : (x0 ^ ps) & 1;
                float wx[K + 2], wy[K + 1], wz[K + 1], gx_[K + 2], gy_[K + 1], gz_[K + 1];
                weights<K>(L.lin, L.k[0], s.t[0], wx); weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
                wgrads<K>(L.lin, L.k[0], s.t[0], gx_); wgrads<K>(L.lin, L.k[1], s.t[1], gy_); wgrads<K>(L.lin, L.k[2], s.t[2], gz_);
                wx[K + 1] = 0.f; gx_[K + 1] = 0.f;
                const int r0 = (xmul == 1 ? x0 : (x0 + par) >> 1) * box.S[1] + (s.i0[1] - box.lo[1]);
                const float2 *bp = box2 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
                for (int ii = 0; ii <= K; ++ii) {
                    // tap i = par + xmul * ii of this sample lives in LDS row r0 + ii * S_y
                    if (par + xmul * ii > (C::ISO ? K : L.k[0])) continue;     // uniform but for the last i of even tap counts
                    const float w2 = par ? wx[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : wx[2 * ii <= K ? 2 * ii : K + 1];
                    const float wxi = xmul == 1 ? wx[ii] : w2;
                    const float g2 = par ? gx_[2 * ii + 1 <= K ? 2 * ii + 1 : K + 1] : gx_[2 * ii <= K ? 2 * ii : K + 1];
                    const float gxi = xmul == 1 ? gx_[ii] : g2;
                    float pW = 0.f, pGy = 0.f, pGz = 0.f;
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        const volatile __attribute__((address_space(3))) unsigned long long *rp =       // (single ds_read_b64: see pull1s_tiled)
                            (const volatile __attribute__((address_space(3))) unsigned long long *)(bp + (ii * box.S[1] + j) * C::PS);
                        float rW = 0.f, rG = 0.f;
#pragma unroll
                        for (int k = 0; k <= K; k += 2) {
                            const unsigned long long raw = rp[k];          // ds_read_b64: taps k and k + 1
                            const float2 t2 = make_float2(__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32)));
                            rW = __builtin_fmaf(wz[k], t2.x, rW);
                            rG = __builtin_fmaf(gz_[k], t2.x, rG);
                            if (k + 1 <= K) {
                                rW = __builtin_fmaf(wz[k + 1 <= K ? k + 1 : K], t2.y, rW);
                                rG = __builtin_fmaf(gz_[k + 1 <= K ? k + 1 : K], t2.y, rG);
                            }
                        }
                        pW = __builtin_fmaf(wy[j], rW, pW);
                        pGy = __builtin_fmaf(gy_[j], rW, pGy);
                        pGz = __builtin_fmaf(wy[j], rG, pGz);
                    }
                    acc[v][0] = __builtin_fmaf(gxi, pW, acc[v][0]);
                    acc[v][1] = __builtin_fmaf(wxi, pGy, acc[v][1]);
                    acc[v][2] = __builtin_fmaf(wxi, pGz, acc[v][2]);
                }
            }
            prof_mark(6);
        }
        // outputs of the fast samples
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r[3] = { acc[v][0], acc[v][1], acc[v][2] };
            if (!fast) {   // slow list overflowed: per-thread global gather
#pragma unroll
                for (int d = 0; d < 3; ++d) r[d] = gather_one_thread<T>(L, vc0, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], d);
            }
            const float gv = Cvt<float, T>::ld(gc[s.o]);
            const float go = (p.extrapolate != 1 && !s.inb) ? 0.f * gv : gv;
#pragma unroll
            for (int d = 0; d < 3; ++d) gg[v][d] = __builtin_fmaf(r[d], go, gg[v][d]);
        }
        // slow list: one wave per sample, lanes = taps
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a3[3] = { 0.f, 0.f, 0.f };
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off; float gr[3];
                    tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, gr);
                    const float vv = (t0 + lane < NTAP) ? Cvt<float, T>::ld(vc0[off]) : 0.f;
                    a3[0] += gr[0] * vv; a3[1] += gr[1] * vv; a3[2] += gr[2] * vv;
                }
                const float go = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : Cvt<float, T>::ld(gc[o]);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float rr = wave_sum(a3[d]);
                    if (lane == 0) {            // the same lane of the same wave owns a slow sample for every channel
                        float *q = ggrid + (b * p.N + o) * 3 + d;
                        *q = (c == 0 ? 0.f : *q) + rr * go;
                    }
                }
            }
        }
        prof_mark(7);
    }
    // sum over the channels of grad_out * d pull / d grid (pushpull.py:256-257), fast samples
#pragma unroll
    for (int v = 0; v < C::VPT; ++v) {
        const bool fast = (fastmask >> v) & 1;
        if (!fast && nslow <= SLOWCAP) continue;       // slow-list samples were written above
        int ox, oy, oz;
        sample_pos<C>(g, tid, v, ox, oy, oz);
        if (!(ox < gx && oy < gy && oz < gz)) continue;
        const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
        float *q = ggrid + (b * p.N + o) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = gg[v][d];
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Scatter of one channel: src(sample) * weights -> target, through the LDS box in
// 64-bit fixed point.  Shared by push / count and by the fused pull backward.
// `src_of(sample)` returns the (masked) source value of a fast sample,
// `src_slow(o)` the unmasked source of a slow-list sample.
//
//     q = trunc(src * w * 2^e),  2^e * max|src| <= 2^30   (per tile and channel)
// every contribution is rounded with absolute error < 2^-30 max|src| (far below
// fp32 rounding of the sums) and a 64-bit slot cannot overflow.  The box holds 8
// bytes per slot, so it is filled in passes over slabs of (flattened) box rows;
// each tap lands in exactly one pass.  Non-finite sources (inf / nan) take the
// per-thread float path so that IEEE semantics survive.
// ---------------------------------------------------------------------------
template <typename C, typename SrcFn, typename SrcSlowFn>
__device__ __forceinline__ bool scatter_channel(const KParams &p, const Lattice &L, const float *__restrict__ grid, int64_t b,
                                                const TileGeom &g, const Box<C> &box, unsigned fastmask, int nslow, int dmax,
                                                int cmax_bits, bool box_is_zero,
                                                float *__restrict__ vc, Smem &sm, SrcFn src_of, SrcSlowFn src_slow)
{
    constexpr int K = C::K, KX = C::KX;
    constexpr int BOX64 = C::BOXF / 2;             // 64-bit slots that fit in the box area
    unsigned long long *box64 = reinterpret_cast<unsigned long long *>(sm.box);
    const int tid = threadIdx.x;
    // ---- block maximum of the sources -> fixed-point scale ---------------------------
    // cmax_bits >= 0: already reduced by Box::build (first 8 channels); else reduce here.
    // (the unmasked sources bound the masked ones: no need to re-read the coordinates)
    int mbits = cmax_bits;
    if (cmax_bits < 0) {
        float amax = 0.f;
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            int ox, oy, oz;
            sample_pos<C>(g, tid, v, ox, oy, oz);
            if (ox < g.gx && oy < g.gy && oz < g.gz) {
                const float a = __builtin_fabsf(src_slow(((int64_t)ox * g.gy + oy) * g.gz + oz));
                amax = (a > amax || a != a) ? a : amax;                            // NaN sticks
            }
        }
        __syncthreads();                               // whoever used sm.hi before is done
        if (tid == 0) sm.hi[0] = 0;
        __syncthreads();
        {
            int bits = __float_as_int(amax);
            bits = wave_max(bits);
            if ((tid & 63) == 0) atomicMax(&sm.hi[0], bits);
        }
        __syncthreads();
        mbits = sm.hi[0];
    }
    if (mbits == 0) return box_is_zero;            // nothing to splat in this tile / channel
    const bool finite = (mbits & 0x7f800000) != 0x7f800000;
    // 2^e * max <= 2^30 : e = 29 - exponent(max)
    int ex = ((mbits >> 23) & 0xff) - 127;
    ex = ex < -90 ? -90 : ex;                      // denormal / tiny maxima: keep 2^e finite
    const float scale = __int_as_float((127 + 29 - ex) << 23);
    const float inv_scale = __int_as_float((127 - 29 + ex) << 23);

    if (!finite || nslow > SLOWCAP) {
        // The deformation does not fit the box (expanding / very rough fields, config 4) or the
        // data are non-finite: every sample of the tile goes the slow-list way -- one wave per
        // sample, lanes = taps, float atomics straight to global memory.  A wave instruction then
        // touches the (K+1)^2 rows of ONE stencil (contiguous runs of K+1 floats) instead of 64
        // unrelated cache lines as with one sample per lane.
        const int wave = tid >> 6, lane = tid & 63;
        const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
        for (int code = wave; code < C::NS; code += C::NT / 64) {
            int ox, oy, oz;
            sample_pos<C>(g, code / C::VPT, code % C::VPT, ox, oy, oz);
            if (!(ox < g.gx && oy < g.gy && oz < g.gz) || (p.dbg & 2)) continue;     // wave-uniform
            float x[3];
            const int64_t o = slow_sample<C>(g, code, p, grid, b, x);
            float sv = src_slow(o);
            if (p.extrapolate != 1 && !coords_inb<C>(p, x)) sv *= 0.f;
            for (int t0 = 0; t0 < NTAP; t0 += 64) {
                int off;
                const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, nullptr);
                if (t0 + lane < NTAP)
                    __hip_atomic_fetch_add(vc + off, w * sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return box_is_zero;
    }

    // ---- slow list: one wave per sample, lanes = taps, one global atomic per lane --------
    if (nslow > 0) {
        const int wave = tid >> 6, lane = tid & 63;
        const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
        for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
            float x[3];
            const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
            float sv = src_slow(o);
            if (p.extrapolate != 1 && !coords_inb<C>(p, x)) sv *= 0.f;
            for (int t0 = 0; t0 < NTAP; t0 += 64) {
                int off;
                const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, nullptr);
                if (t0 + lane < NTAP)
                    __hip_atomic_fetch_add(vc + off, w * sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- 32-bit mode -----------------------------------------------------------------------
    // A lattice point x of the box receives, in units of max|src|, at most
    //     cb = dmax * prod_d  sum_j max_t w_j(t)
    // (group the samples by first-tap cell c: at most dmax per cell, and a sample of cell c
    // reaches x with weight prod_d w_{x_d - c_d}(t_d) <= prod_d max_t w_j(t); the per-order sums
    // of the tap maxima are 1, 2, 1.75, 5/3, 1.599, 1.55, 1.511, 1.479).  With hb = ceil(log2 cb)
    // bits of headroom the sum of q = rne(src * w * 2^(e - hb)) cannot leave int32, and the
    // whole box fits LDS at 4 bytes per slot: ONE pass with ds_add_u32 (5.7 lanes/clk/CU vs 4.6
    // for ds_add_u64) instead of slab passes.  Each contribution is rounded to +-2^(hb-30)
    // max|src|; with n taps the worst slot error is ~1.3 sqrt(n) 2^(hb-29) max|src|, kept below
    // 2.5e-6 by requiring 2^hb <= 1032 / sqrt(n).  Strongly contracting deformations (large
    // dmax) and high orders keep the 64-bit path below.
    {
        const int hb = headroom32(L, dmax);
        const bool precise = hb >= 0;
        if (hb >= 0 && precise && !(p.dbg & 8)) {
            unsigned *box32 = reinterpret_cast<unsigned *>(sm.box);
            const float scale32 = __int_as_float((127 + 29 - ex - hb) << 23);
            const float inv32 = __int_as_float((127 - 29 + ex + hb) << 23);
            const int nslots = box.S[0] * box.S[1] * C::PS;
            // the box is all-zero on entry when the caller says so (left so by Box::build or by
            // the previous channel's flush): no zeroing pass, one barrier less
            if (!box_is_zero) {
                __syncthreads();
                for (int e = tid; e < nslots; e += C::NT) box32[e] = 0u;
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < C::VPT; ++v) {
                if (!((fastmask >> v) & 1) || (p.dbg & 2)) continue;
                const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
                float wx[KX + 1], wy[K + 1], wz[K + 1];
                if (KX > 0) weights<KX>(L.lin, L.k[0], s.t[0], wx); else wx[0] = 1.f;
                weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
                const float ss = src_of(s) * scale32;
                unsigned *bp = box32 + box.base(s);
#pragma unroll
                for (int i = 0; i <= KX; ++i) {
                    if (!C::ISO && i > L.k[0]) continue;      // wave-uniform
                    const float si = ss * wx[i];
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        if (!C::ISO && j > L.k[1]) continue;
                        unsigned *rp = bp + (i * box.S[1] + j) * C::PS;
                        const float sj = si * wy[j];
#pragma unroll
                        for (int k = 0; k <= K; ++k) {
                            if (!C::ISO && k > L.k[2]) continue;
                            // round to nearest: truncation would bias every contribution the same way
                            atomicAdd(rp + k, (unsigned)cvt_rpi(sj * wz[k]));           // ds_add_u32
                        }
                    }
                }
            }
            __syncthreads();
            {
                const int z = tid % C::PZ;
                const bool zin = z < box.S[2];
                const int oz_ = zin ? sm.taboff[2][z] : 0;
                const float sz = zin ? sm.tabsgn[2][z] : 0.f;
                constexpr int RSTEP = C::NT / C::PZ;
                const float inv_sy = 1.f / (float)box.S[1];
                const int rows = box.S[0] * box.S[1];
                for (int r = tid / C::PZ; r < rows; r += RSTEP) {
                    if (zin) {
                        const int a = (int)box32[r * C::PS + z];
                        if (a != 0) {
                            box32[r * C::PS + z] = 0u;                 // leave the box zeroed for the next channel
                            if (p.dbg & 1) continue;
                            const int x = (int)(((float)r + 0.5f) * inv_sy);
                            const int y = r - x * box.S[1];
                            const float sgn = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sz;
                            __hip_atomic_fetch_add(vc + sm.taboff[0][x] + sm.taboff[1][y] + oz_, (float)a * inv32 * sgn,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            return true;                                   // every touched slot was reset by the flush
        }
    }

    // ---- passes over slabs of flattened box rows r = x * S_y + y (8 bytes per slot) -------
    const int rows_all = box.S[0] * box.S[1];
    int rows_pp = BOX64 / C::PS;                   // rows per pass
    rows_pp -= (C::D == 3) ? rows_pp % box.S[1] : 0;   // 3-D: whole x-rows per pass (BOX64 / PZ >= CAPY rows always)
    const int npass = (rows_all + rows_pp - 1) / rows_pp;
    for (int ps = 0; ps < npass; ++ps) {
        const int r_lo = ps * rows_pp;
        const int r_n = (rows_all - r_lo) < rows_pp ? (rows_all - r_lo) : rows_pp;
        __syncthreads();                           // previous pass is flushed
        for (int e = tid; e < r_n * C::PS; e += C::NT) box64[e] = 0ull;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            if (!((fastmask >> v) & 1) || (p.dbg & 2)) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            const int r0 = (s.i0[0] - box.lo[0]) * box.S[1] + (s.i0[1] - box.lo[1]) - r_lo;   // row of tap (0,0) in this slab
            if (r0 + KX * box.S[1] + K < 0 || r0 >= r_n) continue;
            float wx[KX + 1], wy[K + 1], wz[K + 1];
            if (KX > 0) weights<KX>(L.lin, L.k[0], s.t[0], wx); else wx[0] = 1.f;
            weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
            const float ss = src_of(s) * scale;
            unsigned long long *bp = box64 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
            for (int i = 0; i <= KX; ++i) {
                if (!C::ISO && i > L.k[0]) continue;      // wave-uniform
                // 3-D slabs hold whole x-rows: one in-slab test per i instead of per (i, j)
                const int ri = r0 + i * box.S[1];
                if (C::D == 3 && (ri < 0 || ri >= r_n)) continue;
                const float si = ss * wx[i];
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    if (!C::ISO && j > L.k[1]) continue;
                    if (C::D != 3 && (ri + j < 0 || ri + j >= r_n)) continue;
                    unsigned long long *rp = bp + (i * box.S[1] + j) * C::PS;
                    const float sj = si * wy[j];
#pragma unroll
                    for (int k = 0; k <= K; ++k) {
                        if (!C::ISO && k > L.k[2]) continue;
                        // truncation (v_cvt_i32_f32) instead of rne: |error| < 1 unit = 2^-30 max|src|
                        const int q = (int)(sj * wz[k]);
                        atomicAdd(rp + k, (unsigned long long)(long long)q);      // ds_add_u64
                    }
                }
            }
        }
        __syncthreads();
        // flush the slab: fixed point -> float, slot sign, one coalesced global atomic per touched slot
        if (!(p.dbg & 1)) {
            const int z = tid % C::PZ;
            const bool zin = z < box.S[2];
            const int oz_ = zin ? sm.taboff[2][z] : 0;
            const float sz = zin ? sm.tabsgn[2][z] : 0.f;
            constexpr int RSTEP = C::NT / C::PZ;
            const float inv_sy = 1.f / (float)box.S[1];
            for (int r = tid / C::PZ; r < r_n; r += RSTEP) {
                if (zin) {
                    const long long a = (long long)box64[r * C::PS + z];
                    if (a != 0) {
                        const int rg = r_lo + r;
                        const int x = (int)(((float)rg + 0.5f) * inv_sy);
                        const int y = rg - x * box.S[1];
                        const float sgn = sm.tabsgn[0][x] * sm.tabsgn[1][y] * sz;
                        const float f = (float)((double)a * (double)inv_scale);
                        __hip_atomic_fetch_add(vc + sm.taboff[0][x] + sm.taboff[1][y] + oz_, f * sgn,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
    return false;                                      // the 64-bit slabs leave the box dirty
}

// ---------------------------------------------------------------------------
// Scatter of a channel PAIR in the 32-bit regime: both channels' contributions to a lattice
// point travel in ONE ds_add_u64,  W += (q1 << 32) + q0  (two's complement, exact integer
// arithmetic: as long as each channel's true sum fits int32 -- which the headroom bound
// guarantees -- the fields are recovered as lo = (int32) W, hi = (W - lo) >> 32).  Halves the
// LDS atomics of the tap loop; 8-byte slots again mean slab passes.  Returns false when the
// pair is not eligible (caller then scatters the two channels one by one).
// ---------------------------------------------------------------------------
template <typename C, typename SrcFn, typename SlowFn>
__device__ __forceinline__ bool scatter_pair(const KParams &p, const Lattice &L, const float *__restrict__ grid, int64_t b,
                                             const TileGeom &g, const Box<C> &box, unsigned fastmask, int nslow, int dmax,
                                             int mbits0, int mbits1, bool box_is_zero, float *__restrict__ vc0, float *__restrict__ vc1,
                                             Smem &sm, SrcFn src_of, SlowFn src_slow)
{
    constexpr int K = C::K, KX = C::KX;
    constexpr int BOX64 = C::BOXF / 2;
    const int hb = headroom32(L, dmax);
    const bool fin0 = (mbits0 & 0x7f800000) != 0x7f800000, fin1 = (mbits1 & 0x7f800000) != 0x7f800000;
    if (hb < 0 || !fin0 || !fin1 || nslow > SLOWCAP || (p.dbg & 8) || mbits0 == 0 || mbits1 == 0) return false;
    unsigned long long *box64 = reinterpret_cast<unsigned long long *>(sm.box);
    const int tid = threadIdx.x;
    // slow list: one wave per sample, lanes = taps, one global atomic per lane and channel
    if (nslow > 0) {
        const int wave = tid >> 6, lane = tid & 63;
        const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
        for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
            float x[3];
            const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
            float sv0 = src_slow(0, o), sv1 = src_slow(1, o);
            if (p.extrapolate != 1 && !coords_inb<C>(p, x)) { sv0 *= 0.f; sv1 *= 0.f; }
            for (int t0 = 0; t0 < NTAP; t0 += 64) {
                int off;
                const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, nullptr);
                if (t0 + lane < NTAP) {
                    __hip_atomic_fetch_add(vc0 + off, w * sv0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(vc1 + off, w * sv1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    int ex0 = ((mbits0 >> 23) & 0xff) - 127, ex1 = ((mbits1 >> 23) & 0xff) - 127;
    ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
    const float sc0 = __int_as_float((127 + 29 - ex0 - hb) << 23), sc1 = __int_as_float((127 + 29 - ex1 - hb) << 23);
    const float inv0 = __int_as_float((127 - 29 + ex0 + hb) << 23), inv1 = __int_as_float((127 - 29 + ex1 + hb) << 23);
    prof_mark(1);
    // 3-D: an 8-byte box does not fit LDS, so two passes -- split by the PARITY of the box row
    // x, not into contiguous slabs: row x lives at LDS row (x >> 1) * S_y + y of pass (x & 1).
    // Every sample then has taps in both passes (i = par, par + 2, ... with par = (x0 ^ pass) & 1),
    // all lanes stay active and no tap needs an in-slab test.  2-D: one pass (x is degenerate).
    // A box small enough for one pass (smooth deformations: S ~ tile + K) is scattered whole.
    static_assert(C::D != 3 || ((C::CAPX + 1) / 2) * C::CAPY * C::PS <= BOX64, "half box must fit the 8-byte slots");
    const int xmul = box.S[0] * box.S[1] * C::PS <= BOX64 ? 1 : 2;
    for (int ps = 0; ps < xmul; ++ps) {
        const int nxh = xmul == 1 ? box.S[0] : (box.S[0] - ps + 1) >> 1;  // box rows x of this pass
        const int r_n = nxh * box.S[1];
        __syncthreads();                           // previous pass flushed (and its slots re-zeroed)
        if (ps == 0 && !box_is_zero) {
            for (int e = tid; e < BOX64; e += C::NT) box64[e] = 0ull;
        }
        // row table of this pass: LDS row r = xh * S_y + y  <->  box row x = 2 xh + ps
        for (int r = tid; r < r_n; r += C::NT) {
            const int xh = r / box.S[1], y = r - xh * box.S[1];
            const int x = xmul * xh + ps;
            sm.rowtab[r] = make_int2(sm.taboff[0][x] + sm.taboff[1][y], __float_as_int(sm.tabsgn[0][x] * sm.tabsgn[1][y]));
        }
        if (ps == 0 && !box_is_zero) __syncthreads();
        prof_mark(2);
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            if (!((fastmask >> v) & 1) || (p.dbg & 2)) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            const int x0 = s.i0[0] - box.lo[0];
            const int par = xmul == 1 ? 0 : (x0 ^ ps) & 1;
            float wx[KX + 2], wy[K + 1], wz[K + 1];
            if (KX > 0) weights<KX>(L.lin, L.k[0], s.t[0], wx); else wx[0] = 1.f;
            wx[KX + 1] = 0.f;
            weights<K>(L.lin, L.k[1], s.t[1], wy); weights<K>(L.lin, L.k[2], s.t[2], wz);
            const float s0 = src_of(0, s) * sc0, s1 = src_of(1, s) * sc1;
            const int r0 = (xmul == 1 ? x0 : (x0 + par) >> 1) * box.S[1] + (s.i0[1] - box.lo[1]);
            unsigned long long *bp = box64 + r0 * C::PS + (s.i0[2] - box.lo[2]);
#pragma unroll
            for (int ii = 0; ii <= KX; ++ii) {
                // tap i = par + xmul * ii of this sample lives in LDS row r0 + ii * S_y
                if (par + xmul * ii > (C::ISO ? KX : L.k[0])) continue;     // uniform but for the last i of even tap counts
                const float w2 = par ? wx[2 * ii + 1 <= KX ? 2 * ii + 1 : KX + 1] : wx[2 * ii <= KX ? 2 * ii : KX + 1];
                const float wxi = xmul == 1 ? wx[ii] : w2;
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    if (!C::ISO && j > L.k[1]) continue;
                    unsigned long long *rp = bp + (ii * box.S[1] + j) * C::PS;
                    const float wij = wxi * wy[j];
                    const float a0 = s0 * wij, a1 = s1 * wij;
#pragma unroll
                    for (int k = 0; k <= K; ++k) {
                        if (!C::ISO && k > L.k[2]) continue;
                        const int q0 = cvt_rpi(a0 * wz[k]), q1 = cvt_rpi(a1 * wz[k]);
                        // (q1 << 32) + sext(q0): low word q0, high word q1 + (q0 < 0 ? -1 : 0)
                        const unsigned hi = (unsigned)(q1 + (q0 >> 31));
                        atomicAdd(rp + k, ((unsigned long long)hi << 32) | (unsigned)q0);   // ds_add_u64, both channels
                    }
                }
            }
        }
        __syncthreads();
        prof_mark(3);
        {
            // flush: fixed point -> float, slot sign, one coalesced global atomic per touched slot
            // and channel; touched slots are re-zeroed on the way (the box stays clean)
            const int z = tid % C::PZ;
            const bool zin = z < box.S[2];
            const int oz_ = zin ? sm.taboff[2][z] : 0;
            const float sz = zin ? sm.tabsgn[2][z] : 0.f;
            const float f0 = inv0 * sz, f1 = inv1 * sz;
            constexpr int RSTEP = C::NT / C::PZ;
            if (zin) {
                for (int r = tid / C::PZ; r < r_n; r += RSTEP) {
                    const long long a = (long long)box64[r * C::PS + z];
                    if (a != 0) {
                        box64[r * C::PS + z] = 0ull;
                        if (p.dbg & 1) continue;
                        const int2 rt = sm.rowtab[r];
                        const int lo = (int)(a & 0xffffffffll);
                        const int hi = (int)((a - (long long)lo) >> 32);
                        const float sg = __int_as_float(rt.y);
                        const int off = rt.x + oz_;
                        if (lo != 0) __hip_atomic_fetch_add(vc0 + off, (float)lo * (f0 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (hi != 0) __hip_atomic_fetch_add(vc1 + off, (float)hi * (f1 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        prof_mark(4);
    }
    return true;
}

// ---------------------------------------------------------------------------
// push / count : vol[b,c,tap] += w * mask * val[b,c,o]      (nd.py:146-213, pushpull.py:106-142)
// ---------------------------------------------------------------------------
// WC (INTERPOL_FLAG_WITH_COUNT): the target has one more channel than `val`; it receives the
// count image in the same pass.  A compile-time variant: the plain kernels stay exactly as they
// are (they sit at the register limit; any code added to them costs 3-25 % even when not executed).
template <typename C, bool COUNT, bool WC = false>
__global__ __launch_bounds__(C::NT) void push_tiled(KParams p, const typename C::T *__restrict__ val, const float *__restrict__ grid,
                                                    float *__restrict__ vol, int gx, int gy, int gz, int nty, int ntz,
                                                    int ntiles, int nbatch, DeferArgs defer)
{
    if (p.gate && *p.gate) return;                     // the owner-computes organisation took this call (push_owner.hip: own_probe)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    using T = typename C::T;
    const Lattice L = make_lattice<C>(p, 4);         // the target (or its fp32 scratch) is float
    bool clean = false;                                // is the LDS box all-zero?  (block-uniform)
    const WorkRange wr(ntiles * nbatch, false);
    for (int work = wr.first; work < wr.end; work += wr.step) {
        const int64_t b = work / ntiles;
        const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
        Box<C> box;
        const T *ib = COUNT ? nullptr : val + b * p.val_sb;
        prof_mark(-1);
        const unsigned fastmask = box.template build<true>(p, L, grid, b, g, sm,
            [&](int c, int64_t o) { return (COUNT || (WC && c >= p.C)) ? 1.f : Cvt<float, T>::ld(ib[c * p.val_sc + o]); }, clean, WC ? 1 : 0);
        const int nslow = sm.nslow, dmax = sm.dmax;
        prof_mark(0);
        clean = true;                                  // Box::build leaves the box zeroed
        if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm, true)) continue;
        const int nch = p.C + (WC ? 1 : 0);
        for (int c = 0; c < nch; ++c) {
            float *vc = vol + b * p.vol_sb + c * p.vol_sc;
            if constexpr (WC) {
                if (c >= p.C) {
                    // the count channel on its own (no partner left): ones
                    clean = scatter_channel<C>(p, L, grid, b, g, box, fastmask, nslow, dmax, c < 8 ? sm.cmax[c] : -1, clean, vc, sm,
                        [&](const Sample<C> &s) { return (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f; },
                        [&](int64_t) { return 1.f; });
                    continue;
                }
            }
            const T *ic = COUNT ? nullptr : val + b * p.val_sb + c * p.val_sc;
            if (!COUNT && c + 1 < p.C && c < 7) {
                // two channels per LDS atomic when the 32-bit regime applies
                const T *ic1 = ic + p.val_sc;
                const bool done = scatter_pair<C>(p, L, grid, b, g, box, fastmask, nslow, dmax, sm.cmax[c], sm.cmax[c + 1], clean,
                    vc, vc + p.vol_sc, sm,
                    [&](int which, const Sample<C> &s) {
                        const float v = Cvt<float, T>::ld((which ? ic1 : ic)[s.o]);
                        return (p.extrapolate != 1 && !s.inb) ? 0.f * v : v; },
                    [&](int which, int64_t o) { return Cvt<float, T>::ld((which ? ic1 : ic)[o]); });
                if (done) { clean = true; ++c; continue; }          // the pair flush leaves the box zeroed
            }
            if constexpr (WC) {
                if (c + 1 == p.C && c < 7) {
                    // last value channel paired with the count channel
                    const bool done = scatter_pair<C>(p, L, grid, b, g, box, fastmask, nslow, dmax, sm.cmax[c], sm.cmax[c + 1], clean,
                        vc, vc + p.vol_sc, sm,
                        [&](int which, const Sample<C> &s) {
                            const float v = Cvt<float, T>::ld(ic[s.o]);
                            const float r = which ? 1.f : v;
                            return (p.extrapolate != 1 && !s.inb) ? 0.f * r : r; },
                        [&](int which, int64_t o) { const float v = Cvt<float, T>::ld(ic[o]); return which ? 1.f : v; });
                    if (done) { clean = true; ++c; continue; }
                }
            }
            clean = scatter_channel<C>(p, L, grid, b, g, box, fastmask, nslow, dmax, c < 8 ? sm.cmax[c] : -1, clean, vc, sm,
                [&](const Sample<C> &s) { const float v = COUNT ? 1.f : Cvt<float, T>::ld(ic[s.o]); return (p.extrapolate != 1 && !s.inb) ? 0.f * v : v; },   // nd.py:201-203
                [&](int64_t o) { return COUNT ? 1.f : Cvt<float, T>::ld(ic[o]); });
            prof_mark(8);
        }
        __syncthreads();                               // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Grid gradient of the backward of pull (pushpull.py:256-257), the channels contracted per sample:
//   ggrid[b,o,d] = mask * sum_c gout[b,c,o] * d/dx_d pull(vol[b,c])(x_o)
// Structured like gather_tiled<C, true> (one staged box per channel, the three derivative sums of a sample kept in registers over
// the channels and stored behind the last one).
//
// Round 6: this kernel used to take gvol as well (the image gradient scattered from the same tile).  That fused mode, and the grid
// gradient of THIS kernel under it, returned wrong or unwritten values for the isotropic 16^3 tiles of orders 1 - 3 once a workgroup
// served several tiles of a rough field (tools/r6/repro_ggrid.py reproduces it on the round-5 source, commit 0f8bfa2: 2 x C x 96^3,
// sigma = 4, FORCE_TILED + debug bit 16; profiles/r06_pullbwd_repro.txt).  What the reproducer showed: the grid gradient alone fails the same way (so the scatter half is not
// the cause), the wrong entries are entries the kernel never wrote (they change with the allocation, zeros in fresh memory), the
// failure needs neither the density pass of Box::build nor the tile hand-back, and it disappears under EITHER of two changes that do
// not touch the arithmetic -- the XCD-wise work range of the other gather kernels instead of a plain stride, or storing a sample's
// sums inside the channel loop instead of in a second loop over the samples behind it -- and even under a recompilation with those
// two alternatives present but switched off.  No missing barrier or stale LDS state was found by reading; the cause is not isolated
// (a code-generation or timing sensitivity of the old loop nest is what the evidence points to).  The scatter half is gone (every
// backward is split since round 5: push of grad_out + this gradient), the gather half has the loop nest of gather_tiled, which the
// many-tiles sweeps have exercised since round 5, and tests/test_hip_parity.py::test_tile_grid_gradient_many_tiles pins the regime.
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void pullbwd_tiled(KParams p, const typename C::T *__restrict__ gout, const typename C::T *__restrict__ vol,
                                                       const float *__restrict__ grid, float *__restrict__ ggrid,
                                                       int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch, DeferArgs defer)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    constexpr int D = C::D;
    using T = typename C::T;
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;

    float gg[C::VPT][3];
#pragma unroll
    for (int v = 0; v < C::VPT; ++v) { gg[v][0] = 0.f; gg[v][1] = 0.f; gg[v][2] = 0.f; }

    for (int c = 0; c < p.C; ++c) {
        const T *vc = vol + b * p.vol_sb + c * p.vol_sc;
        const T *gc = gout + b * p.val_sb + c * p.val_sc;
        __syncthreads();                               // previous channel's readers are done
        stage_box<C>(vc, box.S, sm);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;   // invalid, or waiting in the slow list
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r[4] = { 0.f, 0.f, 0.f, 0.f };
            if (fast) gather_box<C, true>(sm, box, s, L, r);
            else {
                // slow list overflowed (pathological deformation): per-thread global gather
                if (D == 3) r[1] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 0);
                r[2] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 1);
                r[3] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 2);
            }
            const float gv = Cvt<float, T>::ld(gc[s.o]);
            const float go = (p.extrapolate != 1 && !s.inb) ? 0.f * gv : gv;
            gg[v][0] = __builtin_fmaf(r[1], go, gg[v][0]);
            gg[v][1] = __builtin_fmaf(r[2], go, gg[v][1]);
            gg[v][2] = __builtin_fmaf(r[3], go, gg[v][2]);
            if (c == p.C - 1) {                        // the sample's sums are complete: stored here, like gather_tiled's outputs
                float *q = ggrid + (b * p.N + s.o) * D;
#pragma unroll
                for (int d = 0; d < D; ++d) q[d] = gg[v][(3 - D) + d];
            }
        }
        // slow list: one wave per sample, lanes = taps; accumulated straight into ggrid (the same lane of the same wave owns a slow
        // sample for every channel)
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a[3] = { 0.f, 0.f, 0.f };
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off; float gr[3];
                    tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, gr);
                    const float vv = (t0 + lane < NTAP) ? Cvt<float, T>::ld(vc[off]) : 0.f;
                    a[0] += gr[0] * vv; a[1] += gr[1] * vv; a[2] += gr[2] * vv;
                }
                const float go = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : Cvt<float, T>::ld(gc[o]);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float r = wave_sum(a[(3 - D) + d]);
                    if (lane == 0) {
                        float *q = ggrid + (b * p.N + o) * D + d;
                        *q = (c == 0 ? 0.f : *q) + r * go;
                    }
                }
            }
        }
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Fused backward of push / count (pushpull.py:262-299), one staged gather per channel:
//   gval[b,c,o]  = mask * pull(gvol_out)[c]                              (if gval)
//   ggrid[b,o,d] = mask * sum_c val[b,c,o] * d/dx_d pull(gvol_out)[c]    (if ggrid)
// val == nullptr: count backward (val = 1).
// ---------------------------------------------------------------------------
template <typename C>
__global__ __launch_bounds__(C::NT) void pushbwd_tiled(KParams p, const typename C::T *__restrict__ gvol_out,
                                                       const typename C::T *__restrict__ val, const float *__restrict__ grid,
                                                       typename C::T *__restrict__ gval, float *__restrict__ ggrid,
                                                       int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch, DeferArgs defer)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    constexpr int D = C::D;
    using T = typename C::T;
    const int tid = threadIdx.x;
    const Lattice L = make_lattice<C>(p, (int)sizeof(T));
    const WorkRange wr(ntiles * nbatch);
    for (int work = wr.first; work < wr.end; work += wr.step) {
    const int64_t b = work / ntiles;
    const TileGeom g = tile_geom<C>(work % ntiles, gx, gy, gz, nty, ntz);
    Box<C> box;
    const unsigned fastmask = box.build(p, L, grid, b, g, sm);
    const int nslow = sm.nslow;
    if (hand_back<C>(defer, work, b, g, nslow, p, grid, sm)) continue;
    float gg[C::VPT][3];
#pragma unroll
    for (int v = 0; v < C::VPT; ++v) { gg[v][0] = 0.f; gg[v][1] = 0.f; gg[v][2] = 0.f; }

    for (int c = 0; c < p.C; ++c) {
        const T *vc = gvol_out + b * p.vol_sb + c * p.vol_sc;
        const T *ic = val ? val + b * p.val_sb + c * p.val_sc : nullptr;
        T *oc = gval ? gval + b * p.val_sb + c * p.val_sc : nullptr;
        __syncthreads();
        stage_box<C>(vc, box.S, sm);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;
            const Sample<C> s = load_sample<C>(p, grid, b, g, tid, v);
            if (!s.valid) continue;
            float r[4] = { 0.f, 0.f, 0.f, 0.f };
            if (fast) gather_box<C, true>(sm, box, s, L, r);
            else {
                r[0] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], -1);
                if (D == 3) r[1] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 0);
                r[2] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 1);
                r[3] = gather_one_thread<T>(L, vc, s.i0[0], s.i0[1], s.i0[2], s.t[0], s.t[1], s.t[2], 2);
            }
            const float m = (p.extrapolate != 1 && !s.inb) ? 0.f : 1.f;
            if (oc) oc[s.o] = Cvt<float, T>::st(r[0] * m);
            const float sv = m * (ic ? Cvt<float, T>::ld(ic[s.o]) : 1.f);
            gg[v][0] = __builtin_fmaf(r[1], sv, gg[v][0]);
            gg[v][1] = __builtin_fmaf(r[2], sv, gg[v][1]);
            gg[v][2] = __builtin_fmaf(r[3], sv, gg[v][2]);
        }
        if (nslow > 0 && nslow <= SLOWCAP) {
            const int wave = tid >> 6, lane = tid & 63;
            const int NTAP = (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1);
            for (int sidx = wave; sidx < nslow; sidx += C::NT / 64) {
                float x[3];
                const int64_t o = slow_sample<C>(g, sm.slow[sidx], p, grid, b, x);
                float a[4] = { 0.f, 0.f, 0.f, 0.f };
                for (int t0 = 0; t0 < NTAP; t0 += 64) {
                    int off; float gr[3];
                    const float w = tap_weight<C>(L, x[0], x[1], x[2], t0 + lane, &off, gr);
                    const float vv = (t0 + lane < NTAP) ? Cvt<float, T>::ld(vc[off]) : 0.f;
                    a[0] += w * vv; a[1] += gr[0] * vv; a[2] += gr[1] * vv; a[3] += gr[2] * vv;
                }
                const float m = (p.extrapolate != 1 && !coords_inb<C>(p, x)) ? 0.f : 1.f;
                const float r0 = wave_sum(a[0]);
                if (lane == 0 && oc) oc[o] = Cvt<float, T>::st(r0 * m);
                const float sv = m * (ic ? Cvt<float, T>::ld(ic[o]) : 1.f);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float r = wave_sum(a[1 + (3 - D) + d]);
                    if (lane == 0 && ggrid) {
                        float *q = ggrid + (b * p.N + o) * D + d;
                        *q = (c == 0 ? 0.f : *q) + r * sv;
                    }
                }
            }
        }
    }
    if (ggrid) {
#pragma unroll
        for (int v = 0; v < C::VPT; ++v) {
            const bool fast = (fastmask >> v) & 1;
            if (!fast && nslow <= SLOWCAP) continue;   // slow-list samples were written above
            int ox, oy, oz;
            sample_pos<C>(g, tid, v, ox, oy, oz);
            if (!(ox < gx && oy < gy && oz < gz)) continue;
            const int64_t o = ((int64_t)ox * gy + oy) * gz + oz;
            float *q = ggrid + (b * p.N + o) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] = gg[v][(3 - D) + d];
        }
    }
    __syncthreads();                                   // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
template <typename C>
struct TileCount {
    int gx, gy, gz, ntx, nty, ntz;
    explicit TileCount(const interpol_problem *p)
    {
        gx = C::D == 3 ? (int)p->grid_shape[0] : 1;
        gy = (int)p->grid_shape[C::D == 3 ? 1 : 0];
        gz = (int)p->grid_shape[C::D == 3 ? 2 : 1];
        ntx = (gx + C::TX - 1) / C::TX; nty = (gy + C::TY - 1) / C::TY; ntz = (gz + C::TZ - 1) / C::TZ;
    }
    int ntiles() const { return ntx * nty * ntz; }
    // persistent launch: as many blocks as the chip holds (LDS-limited blocks per CU x CUs)
    dim3 grid(int B) const
    {
        static int cus = 0;
        if (!cus) {
            int dev = 0; hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            if (cus <= 0) cus = 256;
        }
        const long long total = (long long)ntiles() * B;
        const long long per_cu = (long long)(160 * 1024) / (long long)smem_bytes<C>();
        const long long want = (long long)cus * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
        return dim3((unsigned)(total < want ? total : want), 1u);
    }
};

// Kernels that need more than 64 KiB of dynamic LDS must opt in, once per KERNEL and device.  (The
// cache is keyed on the kernel's address: the function-pointer TYPE is shared by every kernel of a
// family, a static per template instantiation would opt in only the first one launched.)
template <typename C, typename F>
static int big_lds(F kernel)
{
    struct Seen { const void *fn; int dev; };
    static Seen seen[256];
    static int nseen = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const void *fn = (const void *)kernel;
    for (int i = 0; i < nseen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<C>());
    if (e != hipSuccess) return (int)e;
    if (nseen < 256) { seen[nseen].fn = fn; seen[nseen].dev = dev; ++nseen; }     // (a lost race only repeats the call)
    return 0;
}

#define IP_CHECK_LAUNCH() do { const hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 1 : (int)e_; } while (0)
// ... then the generic kernel of the operator on the tiles handed back (defer.hip)
#define IP_CHECK_LAUNCH_THEN(deferred) do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; \
                                            const int rc_ = (deferred); return rc_ ? rc_ : 1; } while (0)
#define IP_DEFER(df, t, CC) const Defer df(k, st, (t).ntiles(), p->batch, (t).ntx, (t).nty, (t).ntz, CC::TX, CC::TY, CC::TZ)

// Orders 4 - 5: the pull kernels take tiles of 16 x 8 x 16 samples (two per thread) instead of the 8 x 8 x 16 of the
// other kernels of these orders: less halo staged per sample (config 3 pull 3.55 -> 2.91 ms); the gradient and scatter
// kernels keep one sample per thread (registers).
template <typename C> struct WideTile { using type = C; };
template <typename T, bool ISO, int GM> struct WideTile<Cfg<T, 4, ISO, 3, 8, 8, 16, 1024, 32, GM>> { using type = Cfg<T, 4, ISO, 3, 16, 16, 16, 1024, 32, GM>; };
template <typename T, bool ISO, int GM> struct WideTile<Cfg<T, 5, ISO, 3, 8, 8, 16, 1024, 32, GM>> { using type = Cfg<T, 5, ISO, 3, 16, 16, 16, 1024, 32, GM>; };
template <typename T, bool ISO, int GM> struct WideTile<Cfg<T, 6, ISO, 3, 8, 8, 8, 512, 32, GM>> { using type = Cfg<T, 6, ISO, 3, 8, 16, 16, 512, 32, GM>; };
template <typename T, bool ISO, int GM> struct WideTile<Cfg<T, 7, ISO, 3, 8, 8, 8, 512, 32, GM>> { using type = Cfg<T, 7, ISO, 3, 8, 16, 16, 512, 32, GM>; };

template <typename C0>
static int launch_pull2_impl(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using C = typename WideTile<C0>::type;
    using T = typename C::T;
    if constexpr (C::D == 3) {
        const int attr = big_lds<C>(pull2_tiled<C>);
        if (attr) return attr;
        const TileCount<C> t(p);
        IP_DEFER(df, t, C);
        hipLaunchKernelGGL((pull2_tiled<C>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                           k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
        IP_CHECK_LAUNCH_THEN(df.desc ? DeferOps<T>::pull(k, vol, grid, val, df.tl, st) : 0);
    } else {
        return 0;
    }
}

// single channel at `vol` / `val` (already offset to the channel), shifted-pair kernel
template <typename C0>
static int launch_pull1s_impl(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using C = typename WideTile<C0>::type;
    using T = typename C::T;
    if constexpr (C::D == 3 && C::ISO) {
        const int attr = big_lds<C>(pull1s_tiled<C>);
        if (attr) return attr;
        const TileCount<C> t(p);
        IP_DEFER(df, t, C);
        hipLaunchKernelGGL((pull1s_tiled<C>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                           k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
        IP_CHECK_LAUNCH_THEN(df.desc ? DeferOps<T>::pull(k, vol, grid, val, df.tl, st) : 0);
    } else {
        return 0;
    }
}

template <typename C0>
static int launch_grad1s_impl(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using C = typename WideTile<C0>::type;
    using T = typename C::T;
    if constexpr (C::D == 3 && C::ISO) {
        const int attr = big_lds<C>(grad1s_tiled<C>);
        if (attr) return attr;
        const TileCount<C> t(p);
        IP_DEFER(df, t, C);
        hipLaunchKernelGGL((grad1s_tiled<C>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                           k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
        IP_CHECK_LAUNCH_THEN(df.desc ? DeferOps<T>::grad(k, vol, grid, val, df.tl, st) : 0);
    } else {
        return 0;
    }
}

template <typename C, bool GRAD>
static int launch_gather_impl(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using T = typename C::T;
    // pull with an even channel count: two channels per LDS slot (unless disabled for A/B tests)
    if (!GRAD && C::D == 3 && p->channels % 2 == 0 && !(k.dbg & 4)) return launch_pull2_impl<C>(p, k, vol, grid, val, st);
    if constexpr (!GRAD && C::D == 3 && C::ISO)
        if (p->channels == 1 && !(k.dbg & (4 | 16))) return launch_pull1s_impl<C>(p, k, vol, grid, val, st);
    if constexpr (GRAD && C::D == 3 && C::ISO && C::VPT == 1)
        if (!(k.dbg & (4 | 16))) return launch_grad1s_impl<C>(p, k, vol, grid, val, st);
    if (!GRAD && C::D == 3 && p->channels >= 3 && !(k.dbg & 4)) {
        // odd channel count: the pair kernel on the first C - 1 channels, this kernel on the last one
        KParams kp = k;
        kp.C = (int)p->channels - 1;
        const int rc = launch_pull2_impl<C>(p, kp, vol, grid, val, st);
        if (rc != 1) return rc;
        KParams k1 = k;
        k1.C = 1;
        if constexpr (C::ISO && !GRAD)
            if (!(k.dbg & 16))
                return launch_pull1s_impl<C>(p, k1, (const T *)vol + (p->channels - 1) * k.vol_sc, grid, (T *)val + (p->channels - 1) * k.val_sc, st);
        const int attr1 = big_lds<C>(gather_tiled<C, GRAD>);
        if (attr1) return attr1;
        const TileCount<C> t1(p);
        IP_DEFER(df1, t1, C);
        hipLaunchKernelGGL((gather_tiled<C, GRAD>), t1.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st, k1,
                           (const T *)vol + (p->channels - 1) * k.vol_sc, (const float *)grid, (T *)val + (p->channels - 1) * k.val_sc,
                           t1.gx, t1.gy, t1.gz, t1.nty, t1.ntz, t1.ntiles(), (int)p->batch, df1.args);
        IP_CHECK_LAUNCH_THEN(df1.desc ? DeferOps<T>::pull(k1, (const T *)vol + (p->channels - 1) * k.vol_sc, grid, (T *)val + (p->channels - 1) * k.val_sc, df1.tl, st) : 0);
    }
    const int attr = big_lds<C>(gather_tiled<C, GRAD>);
    if (attr) return attr;
    const TileCount<C> t(p);
    IP_DEFER(df, t, C);
    hipLaunchKernelGGL((gather_tiled<C, GRAD>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                       k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    IP_CHECK_LAUNCH_THEN(!df.desc ? 0 : (GRAD ? DeferOps<T>::grad(k, vol, grid, val, df.tl, st) : DeferOps<T>::pull(k, vol, grid, val, df.tl, st)));
}

template <typename C0>
static int launch_push_impl(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    using C = typename WideTile<C0>::type;
    using T = typename C::T;
    if (k.cc) {
        // values + count in one pass: own instantiations for fp32 storage and isotropic orders 1-3
        // (else declined: the caller runs push and count one after the other)
        if constexpr (std::is_same<T, float>::value && C::ISO && C::K <= 3) {
            if (!val) return 0;
            const int attr = big_lds<C>(push_tiled<C, false, true>);
            if (attr) return attr;
            const TileCount<C> t(p);
            IP_DEFER(df, t, C);
            hipLaunchKernelGGL((push_tiled<C, false, true>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                               k, (const T *)val, (const float *)grid, (float *)vol, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
            IP_CHECK_LAUNCH_THEN(df.template push<T>(k, val, grid, vol, st));
        } else {
            return 0;
        }
    }
    const int attr = val ? big_lds<C>(push_tiled<C, false>) : big_lds<C>(push_tiled<C, true>);
    if (attr) return attr;
    const TileCount<C> t(p);
    IP_DEFER(df, t, C);
    if (val)
        hipLaunchKernelGGL((push_tiled<C, false>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                           k, (const T *)val, (const float *)grid, (float *)vol, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    else
        hipLaunchKernelGGL((push_tiled<C, true>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                           k, (const T *)nullptr, (const float *)grid, (float *)vol, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    IP_CHECK_LAUNCH_THEN(df.template push<T>(k, val, grid, vol, st));
}

// Coordinate mode dispatch (Cfg::GM): the dense-grid instantiation, or the general one for
// separable lattices / displacement fields (forward kernels only; the fused backward kernels
// leave those modes to the generic kernels -- the backward of pull w.r.t. the image alone is a push).
template <typename C, bool GRAD>
static int launch_gather(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    if (k.sep == 3) return 0;                              // affine lattices: class-sorted tiles or generic kernels
    if (k.sep) {
        // own instantiations for fp32 storage, 3-D, isotropic orders 1-3 (linear ... cubic resize
        // and displacement fields); everything else in these modes runs the generic kernels --
        // keeps the build short
        if constexpr (std::is_same<typename C::T, float>::value && C::D == 3 && C::ISO && C::K <= 3) {
            if (k.sep == 1) return launch_gather_impl<typename C::template Mode<1>, GRAD>(p, k, vol, grid, val, st);
            return launch_gather_impl<typename C::template Mode<2>, GRAD>(p, k, vol, grid, val, st);
        } else {
            return 0;
        }
    }
    return launch_gather_impl<C, GRAD>(p, k, vol, grid, val, st);
}
template <typename C>
static int launch_push(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    if (k.sep == 3) return 0;
    if (k.sep) {
        if constexpr (std::is_same<typename C::T, float>::value && C::D == 3 && C::ISO && C::K <= 3) {
            if (k.sep == 1) return launch_push_impl<typename C::template Mode<1>>(p, k, val, grid, vol, st);
            return launch_push_impl<typename C::template Mode<2>>(p, k, val, grid, vol, st);
        } else {
            return 0;
        }
    }
    return launch_push_impl<C>(p, k, val, grid, vol, st);
}

template <typename C>
static int launch_pullbwd(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid,
                          void *gvol, void *ggrid, int64_t gsb, int64_t gsc, hipStream_t st)
{
    using T = typename C::T;
    if (k.sep || gvol || !ggrid) return 0;                 // declined: generic kernels (the tiles hold the grid gradient only, see pullbwd_tiled)
    if constexpr (C::D == 3 && C::ISO && C::VPT == 1) {
        if (!(k.dbg & 16)) {
            // grid gradient alone, high orders: the shifted-pair gather
            using CW = typename WideTile<C>::type;
            const int attr1 = big_lds<CW>(gradc1s_tiled<CW>);
            if (attr1) return attr1;
            const TileCount<CW> t1(p);
            IP_DEFER(df1, t1, CW);
            hipLaunchKernelGGL((gradc1s_tiled<CW>), t1.grid((int)p->batch), dim3(CW::NT), smem_bytes<CW>(), st,
                               k, (const T *)gout, (const T *)vol, (const float *)grid, (float *)ggrid,
                               t1.gx, t1.gy, t1.gz, t1.nty, t1.ntz, t1.ntiles(), (int)p->batch, df1.args);
            IP_CHECK_LAUNCH_THEN(df1.template gradc<T>(k, gout, vol, grid, ggrid, st));
        }
    }
    const int attr = big_lds<C>(pullbwd_tiled<C>);
    if (attr) return attr;
    const TileCount<C> t(p);
    IP_DEFER(df, t, C);
    hipLaunchKernelGGL((pullbwd_tiled<C>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                       k, (const T *)gout, (const T *)vol, (const float *)grid, (float *)ggrid,
                       t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    IP_CHECK_LAUNCH_THEN(df.desc ? DeferOps<T>::pullbwd(k, gout, vol, grid, nullptr, ggrid, 0, 0, df.tl, st) : 0);
}

template <typename C>
static int launch_pushbwd(const interpol_problem *p, const KParams &k, const void *gvol_out, const void *val, const void *grid,
                          void *gval, void *ggrid, hipStream_t st)
{
    using T = typename C::T;
    if (k.sep) return 0;
    const int attr = big_lds<C>(pushbwd_tiled<C>);
    if (attr) return attr;
    const TileCount<C> t(p);
    IP_DEFER(df, t, C);
    hipLaunchKernelGGL((pushbwd_tiled<C>), t.grid((int)p->batch), dim3(C::NT), smem_bytes<C>(), st,
                       k, (const T *)gvol_out, (const T *)val, (const float *)grid, (T *)gval, (float *)ggrid,
                       t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    IP_CHECK_LAUNCH_THEN(df.desc ? DeferOps<T>::pushbwd(k, gvol_out, val, grid, gval, ggrid, df.tl, st) : 0);
}

// Tile shapes: the box must hold tile + K + 2 * halo lattice points per dim.
//   3-D, K <= 3 : 16x16x16 samples, 1024 threads (4 samples each), box <= 33x33x32
//   3-D, K = 4,5: 8x8x16,           1024 threads (1 sample each)
//   3-D, K = 6,7: 8x8x8,             512 threads
//   2-D         : 1x16x32,           512 threads, LDS pitch 64 (box <= 72x64, 36 KiB)
// Mixed per-dim orders use the K = 3 (max order <= 3) or K = 7 tile with ISO = false.
template <typename T, int K, bool ISO> struct Tile3 { using type = Cfg<T, K, ISO, 3, 16, 16, 16, 1024, 32>; };
template <typename T, bool ISO> struct Tile3<T, 4, ISO> { using type = Cfg<T, 4, ISO, 3, 8, 8, 16, 1024, 32>; };
template <typename T, bool ISO> struct Tile3<T, 5, ISO> { using type = Cfg<T, 5, ISO, 3, 8, 8, 16, 1024, 32>; };
template <typename T, bool ISO> struct Tile3<T, 6, ISO> { using type = Cfg<T, 6, ISO, 3, 8, 8, 8, 512, 32>; };
template <typename T, bool ISO> struct Tile3<T, 7, ISO> { using type = Cfg<T, 7, ISO, 3, 8, 8, 8, 512, 32>; };
template <typename T, int K, bool ISO> struct Tile2 { using type = Cfg<T, K, ISO, 2, 1, 32, 32, 1024, 64>; };

} // namespace tiled

// Which tile serves the problem: { order of the tile, iso }, or order < 0 = not eligible.
// Covered: 2-D / 3-D, storage IP_TT, per-dim orders 0..7 (not all 0), any bound / extrapolate.
struct TiledPick { int K; bool iso; };
static TiledPick tiled_pick(const interpol_problem *p, const KParams &k)
{
    const TiledPick no = { -1, false };
    if (p->dim != 3 && p->dim != 2) return no;
    if (p->batch > 65535) return no;
    int64_t n = 1, nt = p->batch;
    for (int d = 0; d < p->dim; ++d) {
        if (p->grid_shape[d] > 0x7fffffff / 4) return no;
        n *= p->grid_shape[d];
        nt *= (p->grid_shape[d] + 7) / 8;
    }
    if (n < 4096 || nt > 0x7fffffff) return no;     // tiny problems: the generic kernel has less fixed cost
    if ((uint64_t)n * (uint64_t)p->dim * 4ull > 0xffffffffull) return no;   // 32-bit byte offsets into one item's grid
    bool same = true; int mx = 0;
    for (int d = 0; d < p->dim; ++d) { same = same && k.order[d] == k.order[0]; mx = k.order[d] > mx ? k.order[d] : mx; }
    if (mx < 1 || mx > 7) return no;
    if (same) return TiledPick{ mx, true };
    return TiledPick{ mx <= 3 ? 3 : 7, false };
}

#define IP_T3(K, ISO) typename tiled::Tile3<IP_TT, K, ISO>::type
#define IP_T2(K, ISO) typename tiled::Tile2<IP_TT, K, ISO>::type
#define IP_BY_ORDER(FN, ...)                                                             \
    const TiledPick pick = tiled_pick(p, k);                                             \
    if (pick.K < 0) return 0;                                                            \
    if (!pick.iso) {                                                                     \
        if (p->dim == 3) { if (pick.K == 3) return FN<IP_T3(3, false) __VA_ARGS__; return FN<IP_T3(7, false) __VA_ARGS__; } \
        if (pick.K == 3) return FN<IP_T2(3, false) __VA_ARGS__; return FN<IP_T2(7, false) __VA_ARGS__; \
    }                                                                                    \
    if (p->dim == 3) switch (pick.K) {                                                   \
        case 1: return FN<IP_T3(1, true) __VA_ARGS__;                                    \
        case 2: return FN<IP_T3(2, true) __VA_ARGS__;                                    \
        case 3: return FN<IP_T3(3, true) __VA_ARGS__;                                    \
        case 4: return FN<IP_T3(4, true) __VA_ARGS__;                                    \
        case 5: return FN<IP_T3(5, true) __VA_ARGS__;                                    \
        case 6: return FN<IP_T3(6, true) __VA_ARGS__;                                    \
        default: return FN<IP_T3(7, true) __VA_ARGS__; }                                 \
    switch (pick.K) {                                                                    \
        case 1: return FN<IP_T2(1, true) __VA_ARGS__;                                    \
        case 2: return FN<IP_T2(2, true) __VA_ARGS__;                                    \
        case 3: return FN<IP_T2(3, true) __VA_ARGS__;                                    \
        case 4: return FN<IP_T2(4, true) __VA_ARGS__;                                    \
        case 5: return FN<IP_T2(5, true) __VA_ARGS__;                                    \
        case 6: return FN<IP_T2(6, true) __VA_ARGS__;                                    \
        default: return FN<IP_T2(7, true) __VA_ARGS__; }

// Trilinear gathers stay on the generic kernel: 8 taps per sample do not pay for staging a tile.
// Measured (2x2x160^3): smooth field 0.08 (generic) vs 0.14 ms (tiled), i.i.d. sigma = 2 noise 0.26 vs
// 0.20 ms -- the generic kernel runs at the HBM roofline on the fields the operator is used with.
static bool linear_only(const interpol_problem *p, const KParams &k)
{
    if (p->flags & INTERPOL_FLAG_FORCE_TILED) return false;
    for (int d = 0; d < p->dim; ++d) if (k.order[d] > 1) return false;
    return true;
}

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

// One translation unit per storage type (IP_TT / IP_TSFX given on the command line) and kernel
// family (IP_TPART: 1 gathers, 2 scatters, 3 fused backward kernels; all when undefined), so
// that the build spreads over the cores;
// each returns 1 when it took the problem, 0 to decline (the generic kernels run).
// 2-D gathers stay on the generic kernel: with (K+1)^2 taps per sample the direct gather is
// already cheaper than staging a tile (measured at config 5: 1.36 ms generic vs 2.19 ms tiled).
// class-sorted tiles (ops_sorted.hip): 3-D, one order 2..3; declines everything else
int IP_SYM(try_sorted_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st);
int IP_SYM(try_sorted_gradc_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st);
int IP_SYM(try_fast_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st);
// lean 2-D tiles (ops_tiled2d.hip): per-dim orders 1..3
int IP_SYM(try_tiled2d_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st);
int IP_SYM(try_tiled2d_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st);
int IP_SYM(try_tiled2d_gradc_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st);

#if !defined(IP_TPART) || IP_TPART == 1
int IP_SYM(try_fast_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    if (p->dim == 2) {
        const int rc = IP_SYM(try_tiled2d_pull_, IP_TSFX)(p, k, vol, grid, val, st);
        if (rc != 0) return rc;
    }
    if (p->dim != 3 && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    if (linear_only(p, k)) return 0;
    {
        const int rc = IP_SYM(try_sorted_pull_, IP_TSFX)(p, k, vol, grid, val, st);
        if (rc != 0) return rc;
    }
    IP_BY_ORDER(tiled::launch_gather, , false>(p, k, vol, grid, val, st))
}

int IP_SYM(try_fast_grad_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    if (p->dim != 3 && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    if (linear_only(p, k)) return 0;
    // (a class-sorted grad kernel -- three packed derivative sums per sample -- was tried: 3.3 vs 3.7 ms at config 2's shape
    //  for cubic, sigma = 2, but 2.9 vs 2.5 ms at the identity and 3.7 vs 2.3 ms for quadratic: 48 accumulator registers
    //  spill.  The grid-gradient kernel, which contracts the channels first, is the one that pays: ops_sorted.hip)
    IP_BY_ORDER(tiled::launch_gather, , true>(p, k, vol, grid, val, st))
}

#endif

int IP_SYM(try_sorted_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st);
// 1-D scatter tiles (push1d.hip)
int IP_SYM(try_push1d_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st);

#if !defined(IP_TPART) || IP_TPART == 2
// `vol` is the (already zero-filled or accumulating) FLOAT target; `val` == NULL means count.
int IP_SYM(try_fast_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    {
        const int rc = p->dim == 2 ? IP_SYM(try_tiled2d_push_, IP_TSFX)(p, k, val, grid, vol, st)
                     : (p->dim == 1 ? IP_SYM(try_push1d_, IP_TSFX)(p, k, val, grid, vol, st) : IP_SYM(try_sorted_push_, IP_TSFX)(p, k, val, grid, vol, st));
        if (rc != 0) return rc;
    }
    IP_BY_ORDER(tiled::launch_push, >(p, k, val, grid, vol, st))
}

#endif

#if !defined(IP_TPART) || IP_TPART == 3
int IP_SYM(try_fast_pullbwd_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid,
                                       void *gvol, void *ggrid, int64_t gsb, int64_t gsc, hipStream_t st)
{
    // the LDS tiles hold the GRID gradient only (round 6: the scatter half of pullbwd_tiled is gone, see the kernel): abi.hip splits every
    // backward -- the image gradient is a push of grad_out; where that push declined, the generic fused kernel takes what is left
    if (gvol || !ggrid) return 0;
    if (p->dim == 2 && !gvol && ggrid) {
        // 2-D, grid gradient only, orders 1..3: the lean tile with the channels contracted per tap (ops_tiled2d.hip)
        const int rc = IP_SYM(try_tiled2d_gradc_, IP_TSFX)(p, k, gout, vol, grid, ggrid, st);
        if (rc != 0) return rc;
    }
    // 2-D, grid gradient only, other orders: a gather like 2-D pull / grad -- the generic kernel is faster than the
    // round-1 tiles there (config 5 shape: 1.7 vs 2.9 ms)
    if (p->dim != 3 && !gvol && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    // 3-D trilinear, grid gradient only: eight taps per sample leave nothing for an LDS box to amortise -- the generic fused
    // kernel is twice as fast on smooth fields (4 x 2 x 256^3: 1.3 vs 2.6 ms at the identity, 1.6 vs 3.1 on a smooth field;
    // the tiles only win under i.i.d. noise of sigma >= 2 voxels with several channels, 3.7 vs 4.3 ms)
    if (p->dim == 3 && !gvol && k.mode == MODE_ISO1 && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    if (p->dim == 3 && !gvol && ggrid) {
        // grid gradient alone, 3-D quadratic / cubic: the class-sorted gather (ops_sorted.hip)
        const int rc = IP_SYM(try_sorted_gradc_, IP_TSFX)(p, k, gout, vol, grid, ggrid, st);
        if (rc != 0) return rc;
    }
    IP_BY_ORDER(tiled::launch_pullbwd, >(p, k, gout, vol, grid, gvol, ggrid, gsb, gsc, st))
}

int IP_SYM(try_fast_pushbwd_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gvol_out, const void *val,
                                       const void *grid, void *gval, void *ggrid, hipStream_t st)
{
    if (p->dim == 2 && ggrid && (val || !gval)) {
        // 2-D, orders 1..3: the same split as in 3-D below, with the lean tiles
        const int rc = IP_SYM(try_tiled2d_gradc_, IP_TSFX)(p, k, val, gvol_out, grid, ggrid, st);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1) {
            if (!gval) return 1;
            const int rc2 = IP_SYM(try_fast_pull_, IP_TSFX)(p, k, gvol_out, grid, gval, st);
            if (rc2 != 0) return rc2;
            ggrid = nullptr;
        }
    }
    if (p->dim != 3 && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    // 3-D trilinear: the generic fused kernel (see try_fast_pullbwd_; 4 x 2 x 256^3: 0.85 vs 2.5 ms at the identity)
    if (p->dim == 3 && k.mode == MODE_ISO1 && !(p->flags & INTERPOL_FLAG_FORCE_TILED)) return 0;
    if (p->dim == 3 && ggrid && (val || !gval)) {                  // (val == NULL: the backward of count, grad_out of ones)
        // 3-D quadratic / cubic: the grid gradient of push IS the grid gradient of pull with the roles of the two
        // images swapped (pushpull.py:278-281 vs 256-257) -- the class-sorted gather -- and the value gradient a pull
        // of grad_vol_out (4x2x256^3 cubic: 1.4 + 2.5 ms instead of 4.5 fused)
        const int rc = IP_SYM(try_sorted_gradc_, IP_TSFX)(p, k, /* grad_out := */ val, /* vol := */ gvol_out, grid, ggrid, st);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1) {
            if (!gval) return 1;
            const int rc2 = IP_SYM(try_fast_pull_, IP_TSFX)(p, k, gvol_out, grid, gval, st);
            if (rc2 != 0) return rc2;
            ggrid = nullptr;                                         // (declined: the fused kernel below does the values only)
        }
    }
    IP_BY_ORDER(tiled::launch_pushbwd, >(p, k, gvol_out, val, grid, gval, ggrid, st))
}
#endif

} // namespace ip

#ifdef IP_PROF
// one counter block per translation unit: interpol_debug_prof_<dtype>_<part>
#ifndef IP_TPART
#define IP_TPART 0
#endif
#define IP_PROF_NAME3(s, q) interpol_debug_prof_##s##_##q
#define IP_PROF_NAME2(s, q) IP_PROF_NAME3(s, q)
extern "C" __attribute__((visibility("default"))) int IP_PROF_NAME2(IP_TSFX, IP_TPART)(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::tiled::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::tiled::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
