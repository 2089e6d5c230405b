// ===========================================================================
// ops_sorted.hip -- LDS tiles with CLASS-SORTED lanes: 3-D, one spline order 2..3 for all
// dims (the pull: 1..3 -- round 5, trilinear under rough fields, behind the router of abi.hip: routed_pull / push_owner.hip: lin_probe),
// f32 / bf16 / f16 storage (fp32 math), any boundary / extrapolation mode.
//   pull  (gather)   : reference interpol/nd.py:80-143
//   push, count      : reference interpol/nd.py:146-213, pushpull.py:106-142   (see push section)
//
// Why another tile family.  The tap loop of a tiled gather reads (K+1)^3 LDS slots per sample at
// lane-uniform offsets from a per-lane base slot.  With an arbitrary deformation the bases of the
// 64 lanes of a wave are unrelated and the reads collide on the LDS banks: measured on gfx950
// (tools/microbench/lds_gather.hip) 7.6 clk per ds_read_b64 wave instruction with random bases,
// 25 clk per ds_read2_b64 -- against 2.7 clk when, inside each 32-lane bank group of the
// instruction, the lanes hold 32 DISTINCT values of (base slot mod 32).  All taps of a sample add
// the same offset in every lane, so a lane assignment that is conflict-free for the base is
// conflict-free for the whole stencil.  Hence:
//
//   1. a 512-thread workgroup owns a tile of 16^3 samples (8 per thread), two workgroups per CU
//      (76 KiB of LDS each): while one stages or sorts, the other runs its tap loop;
//   2. the samples of the tile are COUNTING-SORTED by class q = (base slot) mod 32 and dealt to
//      the lanes so that lane l of every half wave holds class l: rank r of class q goes to half-
//      wave slot r, lane q.  Classes hold 128 samples on average; the surplus of an over-full
//      class fills the holes left by the under-full ones (those half-wave slots pay a two-way
//      conflict), so every thread still processes 8 samples;
//   3. the box of lattice points the tile's stencils touch (<= 32 x 32 x 36, tile + K + halo) is
//      staged through LDS in FOUR PASSES over the residues of the box plane x mod 4: a cubic
//      stencil has exactly one x-tap in every pass, the partial sums stay in registers.  A pass
//      holds 8 planes of 32 rows of 36 slots of 8 bytes (two channels per slot, one ds_read_b64
//      feeds both): 73 728 B.  The plane pitch (32 * 36 slots) is a multiple of 32 slots, so the
//      class of a sample, (4 y0 + z0) mod 32, is the same in every pass; the row pitch 36 makes y0
//      count, which keeps the classes evenly filled for smooth deformations (where z0 alone takes
//      16 values) as well as for rough ones;
//   4. the boundary condition is applied while staging (wrapped offset and sign per box row /
//      column / slice, bounds.py:30-89): the tap loop is 16 ds_read_b64 at immediate offsets + 21
//      packed FMAs per sample and pass;
//   5. results return to the natural order through LDS and are stored coalesced;
//   6. samples whose support leaves the (clamped) box are handled tap-parallel by whole waves from
//      global memory (tile_common.hpp), pathological tiles per thread -- always correct.
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace sorted {

constexpr int NT = 512;                         // threads per workgroup
constexpr int NS = TS * TS * TS;                // samples per tile
constexpr int VPT = NS / NT;                    // samples per thread
constexpr int CAPX = 32, CAPY = 32, CAPZ = 36;  // box capacity (lattice points)
constexpr int PZ = 36;                          // row pitch (8-byte slots)
constexpr int PLANE = CAPY * PZ;                // plane pitch: 1152 = 36 * 32 slots
constexpr int NPL = CAPX / 4;                   // planes resident per pass
constexpr int BOXSLOTS = NPL * PLANE;           // 9216 slots = 73728 B
constexpr int NCLS = 32;                        // classes = 8-byte bank pairs
constexpr int NSLOT = NS / NCLS;                // half-wave slots per class
constexpr int SLOWCAP = 512;
constexpr int ROUGH = 512;                      // out-of-box samples beyond which a tile is left to the bricks of interpol_pull_ws: the slow list (a wave per sample,
                                                // ~0.035 us each) is full and the rest would be gathered one THREAD per sample
constexpr int HANDBACK = NS / 8;                // out-of-box samples beyond which the generic kernel takes a (smooth) tile, defer.hip;
                                                // measured: tools/handback_sweep.py, profiles/r02_handback.txt
constexpr int TABCAP = (BOXSLOTS * 8 - NS * 16) / 2;   // surplus samples the hole table can place (2048)
static_assert(PLANE % NCLS == 0, "the plane pitch must keep the class pass-independent");
static_assert(NS * 16 <= BOXSLOTS * 8, "sample records alias the box");

struct Smem {
    int   taboff[3][40];       // wrapped lattice offset (elements) of box plane / row / slice
    float tabsgn[3][40];       // boundary sign of the same
    int   lo[3], hi[3];        // block reductions of the first-tap indices
    int   nslow, pad[1];
    int   oobc[VPT][NT / 64];  // out-of-box samples per (sample slot, wave): count, then exclusive prefix -- their rank in the slow list
    int   cmax[8];             // scatter kernels: float bits of max |source| of the tile, per channel of the pair
    int   cnt[NCLS + 4];       // samples per class; [NCLS] collects the samples outside the box
    int   ooff[NCLS];          // surplus samples of the classes before this one
    unsigned short slow[SLOWCAP];
    int2  rowtab[NPL * CAPY];  // scatter kernels, per resident row of the pass: { taboff_x + taboff_y, bits of sign_x * sign_y }
    float2 box[BOXSLOTS];      // aliased: float4 rec[NS] + unsigned short holes[TABCAP]; float2 out[NS]
};


// The 16 LDS reads of one x-plane of a stencil (4 rows of 4 slots, row pitch PZ), both channels per
// read, as ONE block of ds_read_b64 at immediate offsets.  Written in assembly because the compiler
// (a) merges neighbouring reads into ds_read2_b64, which costs 3x more per byte on gfx950, and
// (b) interleaves the unrolled samples until their read results spill.  The block waits for its own
// reads (results handed to the compiler must be complete: it may move or spill them); the latency is
// covered by the other waves of the SIMD.
static_assert(PZ == 36, "the immediate offsets below are (row * PZ + k) * 8");
#define IP_RD(o, off) "ds_read_b64 %" #o ", %16 offset:" #off "\n\t"
__device__ __forceinline__ void stencil_reads(unsigned addr, f2 (&v)[16])
{
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16) IP_RD(3, 24)
                 IP_RD(4, 288) IP_RD(5, 296) IP_RD(6, 304) IP_RD(7, 312)
                 IP_RD(8, 576) IP_RD(9, 584) IP_RD(10, 592) IP_RD(11, 600)
                 IP_RD(12, 864) IP_RD(13, 872) IP_RD(14, 880) IP_RD(15, 888)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : "v"(addr) : "memory");
}
// K == 1: the 2 x 2 slots of one x-plane of a trilinear stencil
__device__ __forceinline__ void stencil_reads_k1(unsigned addr, f2 (&v)[4])
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:288\n\tds_read_b64 %3, %4 offset:296\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
}
#undef IP_RD

// ---------------------------------------------------------------------------
// Tile set-up shared by the gather and scatter kernels: coordinates (natural order), bounding
// box, boundary tables, classification, counting sort.  Leaves in registers the thread's 8
// SORTED samples (coordinates + natural id) and, for the natural order, which of the thread's
// own samples are fast / must be handled by the thread itself.
// ---------------------------------------------------------------------------
// MIX (K == 3 only): per-dim runtime orders 1..3 of KParams inside the cubic's four-tap stencil -- first tap floor(x - (k_d-1)/2),
// taps beyond a dim's order carry weight 0 and their slots are never used (cleared after the reads: a non-finite lattice point
// outside the true stencil must not reach the sums); box, classes, sort and passes are the cubic's.
template <int K, int GM, bool MIX = false>
struct Tile {
    int lo[3], S[3];
    // sorted samples: stencil coordinates t = x - i0 per dim and the packed key
    //   bits 0-4 x0 (first box plane), 5-15 y0 * PZ + z0 (slot inside a plane), 16-27 natural id,
    //   28 extrapolation mask (nd.py:10-27), 29 slot holds a sample
    float tx[VPT]; f2 tyz[VPT];
    int   key[VPT];
    unsigned fastmask, selfmask;       // natural order: sample v is in the sorted set / is left to this thread

    // coordinates of the thread's 8 samples (natural order)
    __device__ __forceinline__ static void load(const KParams &p, const float *__restrict__ grid, int64_t b, const TileGeom &g,
                                                const int tid, float (&c)[VPT][3])
    {
        const bool full = g.ox0 + TS <= g.gx && g.oy0 + TS <= g.gy && g.oz0 + TS <= g.gz;     // block-uniform
        if (GM == 0 && full) {
            // one address per thread; its samples lie two x-planes apart
            const float *gp = grid + b * p.grid_sb + (((int64_t)(g.ox0 + (tid >> 8)) * g.gy + (g.oy0 + ((tid >> 4) & 15))) * g.gz + (g.oz0 + (tid & 15))) * 3;
            const int64_t step = (int64_t)g.gy * g.gz * 6;
#pragma unroll
            for (int v = 0; v < VPT; ++v) { c[v][0] = gp[v * step]; c[v][1] = gp[v * step + 1]; c[v][2] = gp[v * step + 2]; }
        } else {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int ox, oy, oz;
                sample_pos(g, tid + NT * v, ox, oy, oz);
                // unconditional loads from a clamped position: the compiler batches them (one exposed round trip)
                ox = ox < g.gx ? ox : g.gx - 1; oy = oy < g.gy ? oy : g.gy - 1; oz = oz < g.gz ? oz : g.gz - 1;
                load_xyz<GM>(p, grid, b, g, ox, oy, oz, c[v]);
            }
        }
    }

    __device__ __forceinline__ void build(const KParams &p, const Lattice &L, const TileGeom &g, Smem &sm, const int tid, float (&c)[VPT][3])
    {
        if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        if (tid <= NCLS) sm.cnt[tid] = 0;
        // (Everything here runs once per sample and is counted in VALU instructions: the kernels are
        // bound by their issue rate, 4 cycles per wave instruction.)
        unsigned validmask = (1u << VPT) - 1u, inbmask = (1u << VPT) - 1u;
        const bool full = g.ox0 + TS <= g.gx && g.oy0 + TS <= g.gy && g.oz0 + TS <= g.gz;     // block-uniform
        if (!full) {
            validmask = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int ox, oy, oz;
                sample_pos(g, tid + NT * v, ox, oy, oz);
                if (ox < g.gx && oy < g.gy && oz < g.gz) validmask |= 1u << v;
            }
        }
        __syncthreads();
        prof_mark(4);
        if (p.extrapolate != 1) {                                    // nd.py:10-27
            inbmask = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (c[v][0] > p.mask_lo_f && c[v][0] < p.mask_hi_f[0] && c[v][1] > p.mask_lo_f && c[v][1] < p.mask_hi_f[1]
                    && c[v][2] > p.mask_lo_f && c[v][2] < p.mask_hi_f[2])
                    inbmask |= 1u << v;
        }
        // ---- first-tap index (kept as a float: exact, and it saturates nowhere) and stencil coordinate
        // i0 = floor(x - (K-1)/2), t = x - i0  (nd.py:45-46); block min / max of i0
        float fl[VPT][3];
        float fmn[3] = { 3e38f, 3e38f, 3e38f }, fmx[3] = { -3e38f, -3e38f, -3e38f };
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fl[v][d] = floorf(c[v][d] - (MIX ? 0.5f * (float)(p.order[d] - 1) : 0.5f * (float)(K - 1)));
                c[v][d] -= fl[v][d];                                 // c becomes t
                const float fv = ((validmask >> v) & 1) ? fl[v][d] : fmn[d];     // (folds away for full tiles)
                fmn[d] = __builtin_fminf(fmn[d], fv);
                fmx[d] = __builtin_fmaxf(fmx[d], ((validmask >> v) & 1) ? fl[v][d] : fmx[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float lim = 1073741824.f;
            const int a = wave_min(__float2int_rz(__builtin_fmaxf(__builtin_fminf(fmn[d], lim), -lim)));
            const int e = wave_max(__float2int_rz(__builtin_fmaxf(__builtin_fminf(fmx[d], lim), -lim)));
            if ((tid & 63) == 0) { atomicMin(&sm.lo[d], a); atomicMax(&sm.hi[d], e); }
        }
        __syncthreads();
        prof_mark(5);
        const int cap[3] = { CAPX, CAPY, CAPZ };
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int l = sm.lo[d], h = sm.hi[d] + K;           // supports span [l, h]
            if (h < l) { l = 0; h = 0; }                   // tile without valid samples
            int sz_ = h - l + 1;
            if (sz_ > cap[d]) { l += (sz_ - cap[d]) / 2; sz_ = cap[d]; }   // keep the centre; the rest goes to the slow list
            lo[d] = l; S[d] = sz_;
        }
        // boundary tables: box slot -> wrapped lattice offset and sign (bounds.py:30-89), one wave per dim
        {
            const int d = tid >> 6, slot = tid & 63;
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
        // ---- classification + histogram.  In the box <=> lo <= i0 <= lo + S - K - 1 in every dim.  Every
        // sample does ONE returning LDS add, unconditionally (the ranks of the 8 samples come back
        // together): on its class counter, or on the spare counter cnt[NCLS] when it is not in the box.
        const float flo[3] = { (float)lo[0], (float)lo[1], (float)lo[2] };
        const float fhi[3] = { (float)(lo[0] + S[0] - K - 1), (float)(lo[1] + S[1] - K - 1), (float)(lo[2] + S[2] - K - 1) };
        fastmask = 0; selfmask = 0;
        int kq[VPT], rk[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const bool in = (fl[v][0] >= flo[0]) & (fl[v][0] <= fhi[0]) & (fl[v][1] >= flo[1]) & (fl[v][1] <= fhi[1])
                          & (fl[v][2] >= flo[2]) & (fl[v][2] <= fhi[2]) & (bool)((validmask >> v) & 1);
            const int x0 = __float2int_rz(fl[v][0]) - lo[0], y0 = __float2int_rz(fl[v][1]) - lo[1], z0 = __float2int_rz(fl[v][2]) - lo[2];
            const int yz = y0 * PZ + z0;                             // slot inside a plane; (y0 * PZ + z0) mod 32 = class
            kq[v] = (x0 & 31) | ((yz & 2047) << 5) | ((tid + NT * v) << 16) | (int)(((inbmask >> v) & 1) << 28) | (1 << 29);
            if (in) fastmask |= 1u << v;
            rk[v] = atomicAdd(&sm.cnt[in ? (yz & (NCLS - 1)) : NCLS], 1);
        }
        // (rare) out-of-box samples: the first SLOWCAP of them -- in the order (sample slot, wave, lane), NOT in the order of
        // arrival: the slow list (a wave per sample) and the per-thread path sum in different orders, and which sample takes
        // which must not depend on the timing of atomics -- go to the slow list, the rest is left to its thread
        const unsigned oob = validmask & ~fastmask;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const unsigned long long bal = __ballot((oob >> v) & 1);
            if ((tid & 63) == 0) sm.oobc[v][tid >> 6] = __popcll(bal);
        }
        __syncthreads();
        if (tid < 64) {
            const int c = sm.oobc[tid >> 3][tid & 7];                // (VPT x NT / 64 = 64 counters, slot-major)
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
            sm.oobc[tid >> 3][tid & 7] = incl - c;
            if (tid == 63) sm.nslow = incl;
        }
        prof_mark(6);
        // holes and surplus: class q offers max(0, NSLOT - cnt) holes and has max(0, cnt - NSLOT) surplus samples
        const int q_l = tid & 31;
        const int cq = sm.cnt[q_l];
        const int holes = cq < NSLOT ? NSLOT - cq : 0, surplus = cq > NSLOT ? cq - NSLOT : 0;
        int nholes, nsurplus;
        const int hoff = half_excl_scan(holes, nholes);
        const int soff = half_excl_scan(surplus, nsurplus);
        if (tid < NCLS) sm.ooff[tid] = soff;
        unsigned short *holetab = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(sm.box) + NS * 16);
        {
            // hole m of class q_l: half-wave slot cq + m, lane q_l -- listed while surplus samples remain
            const int lim = nsurplus < TABCAP ? nsurplus : TABCAP;
            for (int m = tid >> 5; m < holes && hoff + m < lim; m += NT / 32)
                holetab[hoff + m] = (unsigned short)((cq + m) * 32 + q_l);
        }
        const int placed = nsurplus < TABCAP ? nsurplus : TABCAP;
        const int filled = placed - hoff < 0 ? 0 : (placed - hoff > holes ? holes : placed - hoff);
        const int cnteff = (cq < NSLOT ? cq : NSLOT) + filled;       // occupied half-wave slots of lane q_l
        __syncthreads();
        prof_mark(7);
        // slow list: rank = prefix of the (slot, wave) counters + position among the wave's lanes
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const unsigned long long bal = __ballot((oob >> v) & 1);
            if ((oob >> v) & 1) {
                const int rank = sm.oobc[v][tid >> 6] + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                if (rank < SLOWCAP) sm.slow[rank] = (unsigned short)(tid + NT * v);
                else selfmask |= 1u << v;
            }
        }
        float4 *rec = reinterpret_cast<float4 *>(sm.box);
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!((fastmask >> v) & 1)) continue;
            const int q = (kq[v] >> 5) & 31, r = rk[v];
            int pos = r * 32 + q;
            if (r >= NSLOT) {                                        // surplus sample: into a hole
                const int o = sm.ooff[q] + r - NSLOT;
                pos = o < TABCAP ? (int)holetab[o] : -1;
                if (pos < 0) { fastmask &= ~(1u << v); selfmask |= 1u << v; continue; }   // pathological tile
            }
            rec[pos] = make_float4(c[v][0], c[v][1], c[v][2], __int_as_float(kq[v]));
        }
        __syncthreads();
        prof_mark(8);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const float4 r = rec[tid + NT * j];                      // half-wave slot (tid >> 5) + 16 j, lane tid & 31
            const bool on = (tid >> 5) + (NT / 32) * j < cnteff;
            // an empty slot reads the box corner (a harmless address) and is dropped at the end
            tx[j] = on ? r.x : 0.5f * (float)(K - 1) + 0.25f;
            tyz[j] = on ? f2{ r.y, r.z } : f2{ 0.5f * (float)(K - 1) + 0.25f, 0.5f * (float)(K - 1) + 0.25f };
            key[j] = on ? __float_as_int(r.w) : 0;
        }
        __syncthreads();                                             // the records make way for the box
    }
};

// ---------------------------------------------------------------------------
// pull: val[b,c,o] = mask * sum_taps w vol        (nd.py:80-143)
// ---------------------------------------------------------------------------
template <typename T, int K, int GM, bool MIX = false>
__global__ __launch_bounds__(NT, 4) void pull_sorted(KParams p, const T *__restrict__ vol, const float *__restrict__ grid,
                                                     T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch,
                                                     DeferArgs defer)
{
    static_assert(!MIX || K == 3, "mixed orders live in the cubic's stencil");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = MIX ? p.order[d] : K; }
    L.lin = 0;
#ifndef IP_NOVERDICT
    // the probe of the call gave every tile to the bricks (abi.hip: routed_pull); gate_n == -3 (trilinear): the tiles run on verdict 1 alone
    if (p.verdict && (p.gate_n == -3 ? *p.verdict != 1 : *p.verdict == 1)) return;
#endif
    if (p.gate && p.gate_n > 0) {
        // interpol_pull_ws: the header, brick counters and brick list of the bricks' workspace lie in front of the tile flags; the
        // sample tiles clear them on their way (own_bin, launched behind this kernel, counts in them): no launch of its own
        int *z = const_cast<int *>(p.gate) - p.gate_n;
        for (int i = (int)blockIdx.x * NT + (int)threadIdx.x; i < p.gate_n; i += (int)gridDim.x * NT) z[i] = 0;
    }
    const WorkRange wr(ntiles * nbatch);
    if (p.gate && p.gate_n >= 0) {
        // (the workspace arrives dirty: the workgroup clears the flags of ITS tiles before it starts -- one round trip, under the
        //  first coordinate loads; a store per tile inside the loop is waited for at the tile's next barrier: 4 % of the kernel)
        for (int w_ = wr.first + (int)threadIdx.x * wr.step; w_ < wr.end; w_ += NT * wr.step) const_cast<int *>(p.gate)[w_] = 0;
    }
    // gate_n < 0: the small-box tiles of pull_direct.hip ran in front of this kernel and wrote every flag -- 0: served there
    const bool after_direct = p.gate && p.gate_n < 0;
    for (int work = wr.first; work < wr.end; work += wr.step) {
        // the thread index is made opaque per tile: everything derived from it would otherwise be
        // hoisted out of the persistent loop and held (spilled) across all phases
        const int tid = opaque((int)threadIdx.x);
        if (after_direct && p.gate[work] == 0) continue;              // (block-uniform)
        const int64_t b = work / ntiles;
        int tile = work % ntiles;
        const TileGeom g = tile_geom(tile, gx, gy, gz, nty, ntz);
        prof_mark(-1);
        Tile<K, GM, MIX> tl;
        float cnext[VPT][3];
        Tile<K, GM, MIX>::load(p, grid, b, g, tid, cnext);
        tl.build(p, L, g, sm, tid, cnext);
        const int nslow = sm.nslow < SLOWCAP ? sm.nslow : SLOWCAP;
        // interpol_pull_ws: a tile with many samples outside the box -- each costs a wave -- is left to the bricks of the image
        // (push_owner.hip: own_gather), whose cost does not depend on the deformation: flagged, skipped (block-uniform).  Rough
        // AND smooth-but-stretched tiles (folding fields, zooms: 4.7 -> 2.3 ms where the generic kernel used to take them)
        if (p.gate && sm.nslow > ROUGH) {
            if (tid == 0) const_cast<int *>(p.gate)[work] = 1;
            __syncthreads();
            continue;
        }
        // (after pull_direct: the flag said "left to pull_sorted"; served here -- every thread read it before the barriers of build)
        if (after_direct && tid == 0) const_cast<int *>(p.gate)[work] = 0;
        if (defer.flag) {                                                 // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
            bool hand_back = sm.nslow > (HANDBACK << ((p.dbg >> 9) & 7));
            if (hand_back) hand_back = tiled::tile_smooth(p, grid, b, 3, g.ox0, g.oy0, g.oz0, TS, TS, TS, g.gx, g.gy, g.gz, sm.hi);
            if (hand_back && tid == 0) defer_mark(defer, work, tile_desc(b, g.ox0 / TS, g.oy0 / TS, g.oz0 / TS));
            if (hand_back && defer.desc) { __syncthreads(); continue; }
        }
        // rows of the box are contiguous runs of the lattice's unit-stride dim, sign +1 throughout
        // (dst1 has sign 0 at index 0 -- quirk B-3 -- so its run must start at 1)
        const bool zlin = L.ss[2] == 1 && tl.lo[2] >= (L.bound[2] == B_DST1 ? 1 : 0) && tl.lo[2] + tl.S[2] <= L.n[2];
        // the boundary conditions of x and y never change the sign (replicate, dct1, dct2, dft: bounds.py:30-89)
        const bool plus = L.bound[0] != B_ZERO && L.bound[0] != B_DST1 && L.bound[0] != B_DST2
                       && L.bound[1] != B_ZERO && L.bound[1] != B_DST1 && L.bound[1] != B_DST2;
        prof_mark(0);

        for (int c = 0; ; c += 2) {
            const bool two = c + 1 < p.C;
            const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
            const T *vc1 = two ? vc0 + p.vol_sc : vc0;
            T *oc0 = val + b * p.val_sb + c * p.val_sc;
            T *oc1 = oc0 + p.val_sc;
            f2 acc[VPT];
#pragma unroll
            for (int j = 0; j < VPT; ++j) acc[j] = f2{ 0.f, 0.f };

            for (int ps = 0; ps < 4; ++ps) {
                const int tid = opaque((int)threadIdx.x);
                const int npl = (tl.S[0] - ps + 3) >> 2;            // box planes x = ps, ps + 4, ...
                // stage: slot (xq * 32 + y) * PZ + z = (sign * c0, sign * c1) of the wrapped lattice point
                const int nrow = npl * CAPY;
                const int omask = (p.dbg & 64) ? 0x3fff : -1;       // ablation: loads from a 64 KiB window
                __syncthreads();                                     // the previous pass's readers are done
                if (p.dbg & 1) {
                } else if (zlin) {
                    // the box's z-range lies inside the lattice and z is the unit-stride dim: rows are
                    // contiguous runs.  A thread moves QUADS of 4 slots: two 16-byte loads (one per
                    // channel), two 16-byte LDS stores.  (Narrow loads are what the vector L1 is slow at:
                    // it looks up one tag per cycle whatever the width -- 4-byte loads of the row ends
                    // alone cost as much as all the quads.)  ALL loads of the pass are issued before the
                    // first store: one exposed round trip.
                    const int nq = (tl.S[2] + 3) >> 2;               // quads per row; the last one is shifted to END at S_z
                    // thread = (row of the sweep, quad): 56 rows of 9 quads per sweep, 5 sweeps over the 256 resident rows -- the
                    // quad, its z offset and the row's slot are fixed per thread and pass (no division per quad)
                    constexpr int QPR = PZ / 4, RPS = NT / QPR, NU = (NPL * CAPY + RPS - 1) / RPS;
                    const int r0 = tid / QPR, qd = tid - r0 * QPR;
                    const bool qon = qd < nq && r0 < RPS;
                    const int zs = 4 * qd + 4 <= tl.S[2] ? 4 * qd : tl.S[2] - 4;
                    const int zoff = tl.lo[2] + zs;
                    float4 a0[NU], a1[NU]; float sg[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int r = r0 + RPS * u;
                        const bool on = qon && r < nrow && (r & 31) < tl.S[1];
                        const int xr = on ? 4 * (r >> 5) + ps : 0, yr = on ? r & 31 : 0;
                        const int off = (on ? sm.taboff[0][xr] + sm.taboff[1][yr] + zoff : 0) & omask;
                        sg[u] = plus ? 1.f : sm.tabsgn[0][xr] * sm.tabsgn[1][yr];
                        a0[u] = ld4<T>(vc0 + off);
                        a1[u] = ld4<T>(vc1 + off);
                    }
                    float2 *dst0 = sm.box + r0 * PZ + zs;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int r = r0 + RPS * u;
                        if (qon && r < nrow && (r & 31) < tl.S[1]) {
                            float2 *dst = dst0 + u * (RPS * PZ);
                            if (!plus) {
                                a0[u].x *= sg[u]; a0[u].y *= sg[u]; a0[u].z *= sg[u]; a0[u].w *= sg[u];
                                a1[u].x *= sg[u]; a1[u].y *= sg[u]; a1[u].z *= sg[u]; a1[u].w *= sg[u];
                            }
                            if (!(zs & 1)) {
                                reinterpret_cast<float4 *>(dst)[0] = make_float4(a0[u].x, a1[u].x, a0[u].y, a1[u].y);
                                reinterpret_cast<float4 *>(dst)[1] = make_float4(a0[u].z, a1[u].z, a0[u].w, a1[u].w);
                            } else {                                 // shifted last quad of an odd extent: 8-byte stores
                                dst[0] = make_float2(a0[u].x, a1[u].x); dst[1] = make_float2(a0[u].y, a1[u].y);
                                dst[2] = make_float2(a0[u].z, a1[u].z); dst[3] = make_float2(a0[u].w, a1[u].w);
                            }
                        }
                    }
                } else {
                    // general case (the box wraps in z, or z is strided): slot by slot through the z table
                    constexpr int U = 8;
                    const int z = tid & 31;
                    const bool zin = z < tl.S[2];
                    const int oz = zin ? sm.taboff[2][z] : 0;
                    const float sgz = zin ? sm.tabsgn[2][z] : 0.f;
                    for (int r0 = tid >> 5; r0 < nrow; r0 += (NT / 32) * U) {
                        float v0[U], v1[U], sg[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int r = r0 + u * (NT / 32);
                            const bool on = zin && r < nrow && (r & 31) < tl.S[1];
                            const int xr = on ? 4 * (r >> 5) + ps : 0, yr = on ? r & 31 : 0;
                            const int off = on ? sm.taboff[0][xr] + sm.taboff[1][yr] + oz : 0;
                            sg[u] = on ? sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sgz : 0.f;
                            v0[u] = Cvt<float, T>::ld(vc0[off]);
                            v1[u] = Cvt<float, T>::ld(vc1[off]);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int r = r0 + u * (NT / 32);
                            if (zin && r < nrow && (r & 31) < tl.S[1]) sm.box[r * PZ + z] = make_float2(v0[u] * sg[u], v1[u] * sg[u]);
                        }
                    }
                    for (int e = tid; tl.S[2] > 32 && e < nrow * 4; e += NT) {      // slices 32 ... 35
                        const int r = e >> 2, z2 = 32 + (e & 3);
                        if ((r & 31) < tl.S[1] && z2 < tl.S[2]) {
                            const int xr = 4 * (r >> 5) + ps, yr = r & 31;
                            const int off = sm.taboff[0][xr] + sm.taboff[1][yr] + sm.taboff[2][z2];
                            const float sgn = sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sm.tabsgn[2][z2];
                            sm.box[r * PZ + z2] = make_float2(Cvt<float, T>::ld(vc0[off]) * sgn, Cvt<float, T>::ld(vc1[off]) * sgn);
                        }
                    }
                }
                __syncthreads();
                prof_mark(1);
                if (p.dbg & 2) continue;
                const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
                // (mixed orders: the loop below once per combination of per-dim orders, chosen by a uniform branch -- the orders are
                //  compile-time constants inside, like K in the isotropic kernels)
#ifdef IP_SORTED_MIX_TU
                auto taps = [&](auto kxc, auto kyc, auto kzc) {
                [[maybe_unused]] constexpr int KX = decltype(kxc)::value, KY = decltype(kyc)::value, KZ = decltype(kzc)::value;
#else
                {                                                    // (the isotropic module: no lambda -- its code is kept as it was, to the byte)
                [[maybe_unused]] constexpr int KX = K, KY = K, KZ = K;
#endif
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    // (opaque per pass: the weights are recomputed -- hoisted out of the pass loop they
                    // would be 9 more live values per sample, i.e. spilled)
                    float tx = tl.tx[j]; f2 tyz = tl.tyz[j];
                    asm volatile("" : "+v"(tx), "+v"(tyz));
                    const int key = tl.key[j];
                    const int x0 = key & 31;
                    const int i = (ps - x0) & 3;                     // the x-tap of this pass
                    const int xq = (x0 + (K == 3 || i <= K ? i : 0)) >> 2;
                    if (K == 1) {
                        // trilinear: two of the four passes hold an x-tap of the sample, four slots each
                        if (i <= 1) {
                            f2 t4[4];
                            stencil_reads_k1(boxaddr + 8u * (unsigned)(xq * PLANE + ((key >> 5) & 2047)), t4);
                            const float wxi = i == 0 ? 1.f - tx : tx;
                            const f2 w0 = 1.f - tyz, w1 = tyz;
                            const f2 q0 = f2{ w0.y, w0.y } * t4[0] + f2{ w1.y, w1.y } * t4[1], q1 = f2{ w0.y, w0.y } * t4[2] + f2{ w1.y, w1.y } * t4[3];
                            acc[j] = f2{ wxi, wxi } * (f2{ w0.x, w0.x } * q0 + f2{ w1.x, w1.x } * q1) + acc[j];
                        }
                        asm volatile("" : "+v"(acc[j]));
                        continue;
                    }
                    f2 t2[16];
                    stencil_reads(boxaddr + 8u * (unsigned)(xq * PLANE + ((key >> 5) & 2047)), t2);
                    const float wxi = mixed_or_weight_x<K, MIX>(KX, tx, i);
                    f2 w[4];
                    if constexpr (MIX) mixed_weights_yz(KY, KZ, tyz, w);
                    else weights_yz<K>(tyz, w);
                    // (mixed orders: KY, KZ <= K = 3 bound the loops -- the slots beyond a dim's order are read but never used:
                    //  a non-finite lattice point outside the true stencil stays out of the sums)
                    f2 pp = { 0.f, 0.f };
#pragma unroll
                    for (int jy = 0; jy <= KY; ++jy) {
                        f2 q = { 0.f, 0.f };
#pragma unroll
                        for (int k = 0; k <= KZ; ++k) q = f2{ w[k].y, w[k].y } * t2[4 * jy + k] + q;
                        pp = f2{ w[jy].x, w[jy].x } * q + pp;
                    }
                    if constexpr (MIX) { if (i > KX) pp = f2{ 0.f, 0.f }; }     // (a plane beyond the x-stencil may hold anything)
                    acc[j] = f2{ wxi, wxi } * pp + acc[j];
                    // the sums are pinned here: otherwise the FMAs sink past the pass loop's back edge and
                    // the read results of several samples stay live across the barriers (spilled)
                    asm volatile("" : "+v"(acc[j]));
                }
#ifdef IP_SORTED_MIX_TU
                };
                if constexpr (MIX) mix_dispatch(L.k[0], L.k[1], L.k[2], taps);
                else taps(std::integral_constant<int, K>{}, std::integral_constant<int, K>{}, std::integral_constant<int, K>{});
#else
                }
#endif
                prof_mark(2);
            }
            {
            // back to the natural order through LDS
            __syncthreads();
            const int tid = opaque((int)threadIdx.x);
            float2 *outb = sm.box;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int key = opaque(tl.key[j]);                   // (else the address and the mask are computed -- and spilled -- before the passes)
                if (!((key >> 29) & 1)) continue;
                const float m = (float)((key >> 28) & 1);            // nd.py:139-140
                outb[(key >> 16) & (NS - 1)] = make_float2(acc[j].x * m, acc[j].y * m);
            }
            // out-of-box samples: one wave per sample, lanes = taps, straight from global memory
            if (nslow > 0) {
                const int wave = tid >> 6, lane = tid & 63;
                for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                    float a0, a1, m;
                    slow_taps<T, K, GM, MIX>(p, L, grid, b, g, sm.slow[sidx], lane, vc0, vc1, a0, a1, m);
                    a0 = wave_sum(a0); a1 = wave_sum(a1);
                    if (lane == 0) outb[sm.slow[sidx]] = make_float2(a0 * m, a1 * m);
                }
            }
            // pathological tiles (slow list or hole table overflowed): the thread gathers its sample itself
            if (tl.selfmask) {
                for (int v = 0; v < VPT; ++v) {
                    if (!((tl.selfmask >> v) & 1)) continue;
                    int ox, oy, oz; float x[3];
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    int ii[3]; float tt[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) split(MIX ? L.k[d] : K, x[d], ii[d], tt[d]);
                    const float m = inb_mask(p, x);
                    outb[tid + NT * v] = make_float2(m * tiled::gather_one_thread<T>(L, vc0, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1),
                                                     m * tiled::gather_one_thread<T>(L, vc1, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1));
                }
            }
            __syncthreads();
            if (g.ox0 + TS <= g.gx && g.oy0 + TS <= g.gy && g.oz0 + TS <= g.gz) {
                // whole tile: 16-byte stores of four z-neighbours (narrow stores are issue-bound)
#pragma unroll
                for (int u = 0; u < NS / 4 / NT; ++u) {
                    const int qi = tid + NT * u;                     // quad: x = qi >> 6, y = (qi >> 2) & 15, z = 4 (qi & 3)
                    const float4 *src = reinterpret_cast<const float4 *>(outb + 4 * qi);
                    const float4 lo_ = src[0], hi_ = src[1];
                    const int64_t o = ((int64_t)(g.ox0 + (qi >> 6)) * g.gy + (g.oy0 + ((qi >> 2) & 15))) * g.gz + (g.oz0 + 4 * (qi & 3));
                    st4<T>(oc0 + o, make_float4(lo_.x, lo_.z, hi_.x, hi_.z));
                    if (two) st4<T>(oc1 + o, make_float4(lo_.y, lo_.w, hi_.y, hi_.w));
                }
            } else {
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    int ox, oy, oz;
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    if (!(ox < g.gx && oy < g.gy && oz < g.gz)) continue;
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    const float2 r = outb[tid + NT * v];
                    oc0[o] = Cvt<float, T>::st(r.x);
                    if (two) oc1[o] = Cvt<float, T>::st(r.y);
                }
            }
            }
            prof_mark(3);
            if (c + 2 >= p.C) break;
            prof_mark(3);
        }
        __syncthreads();                                             // the next tile reuses the LDS tables / lists
    }
}


// ===========================================================================
// Grid gradient of pull (backward of grid_pull with respect to the grid, pushpull.py:256-257):
//   ggrid[b,o,d] = mask * sum_c gout[b,c,o] * d/dx_d pull(vol[b,c])(x_o)
// The class-sorted gather of pull_sorted with the channels contracted per tap and three derivative sums.
// ===========================================================================
template <typename T, int K, int GM, bool MIX = false>
__global__ __launch_bounds__(NT, 4) void gradc_sorted(KParams p, const T *__restrict__ vol, const T *__restrict__ gout, const float *__restrict__ grid,
                                                      float *__restrict__ ggrid, int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch,
                                                      DeferArgs defer)
{
    static_assert(!MIX || K == 3, "mixed orders live in the cubic's stencil");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = MIX ? p.order[d] : K; }
    L.lin = K == 1;                                                  // (all-linear: the reference's iso1 gradients -1, +1 in the out-of-box paths as well)
    // interpol_pull_backward with a bricks workspace: the probe's verdict lies -gate_n ints in front of the tile flags; 1 = every
    // tile goes to the bricks of the image (push_owner.hip: own_probe, nch < 0)
    if (p.gate && p.gate_n < 0 && p.gate[p.gate_n] == 1) return;
    if (!p.gate && p.verdict && p.gate_n == -3 && *p.verdict != 1) return;      // trilinear (abi.hip: routed_gradc): the tiles run on verdict 1 alone
    const WorkRange wr(ntiles * nbatch);
    if (p.gate)
        for (int w_ = wr.first + (int)threadIdx.x * wr.step; w_ < wr.end; w_ += NT * wr.step) const_cast<int *>(p.gate)[w_] = 0;
    for (int work = wr.first; work < wr.end; work += wr.step) {
        // the thread index is made opaque per tile: everything derived from it would otherwise be
        // hoisted out of the persistent loop and held (spilled) across all phases
        const int tid = opaque((int)threadIdx.x);
        const int64_t b = work / ntiles;
        int tile = work % ntiles;
        const TileGeom g = tile_geom(tile, gx, gy, gz, nty, ntz);
        prof_mark(-1);
        Tile<K, GM, MIX> tl;
        float cnext[VPT][3];
        Tile<K, GM, MIX>::load(p, grid, b, g, tid, cnext);
        tl.build(p, L, g, sm, tid, cnext);
        const int nslow = sm.nslow < SLOWCAP ? sm.nslow : SLOWCAP;
        // interpol_pull_backward with a bricks workspace: the tile is left to the bricks of the image (as pull_sorted)
        if (p.gate && sm.nslow > ROUGH) {
            if (tid == 0) const_cast<int *>(p.gate)[work] = 1;
            __syncthreads();
            continue;
        }
        if (defer.flag) {                                                 // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
            bool hand_back = sm.nslow > (HANDBACK << ((p.dbg >> 9) & 7));
            if (hand_back) hand_back = tiled::tile_smooth(p, grid, b, 3, g.ox0, g.oy0, g.oz0, TS, TS, TS, g.gx, g.gy, g.gz, sm.hi);
            if (hand_back && tid == 0) defer_mark(defer, work, tile_desc(b, g.ox0 / TS, g.oy0 / TS, g.oz0 / TS));
            if (hand_back && defer.desc) { __syncthreads(); continue; }
        }
        // rows of the box are contiguous runs of the lattice's unit-stride dim, sign +1 throughout
        // (dst1 has sign 0 at index 0 -- quirk B-3 -- so its run must start at 1)
        const bool zlin = L.ss[2] == 1 && tl.lo[2] >= (L.bound[2] == B_DST1 ? 1 : 0) && tl.lo[2] + tl.S[2] <= L.n[2];
        prof_mark(0);
        float ag[VPT][3];                                            // grid gradient of the thread's sorted samples, all channels
#pragma unroll
        for (int j = 0; j < VPT; ++j) { ag[j][0] = 0.f; ag[j][1] = 0.f; ag[j][2] = 0.f; }

        for (int c = 0; ; c += 2) {
            const bool two = c + 1 < p.C;
            const T *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
            const T *vc1 = two ? vc0 + p.vol_sc : vc0;
            // grad_out of the pair in the natural order (coalesced), handed to the sorted lanes through the box
            f2 go[VPT];
            {
                const T *gc0 = gout + b * p.val_sb + c * p.val_sc;
                const T *gc1 = two ? gc0 + p.val_sc : gc0;
                float2 *srcb = sm.box;
                float g0[VPT], g1[VPT];
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    int ox, oy, oz;
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    const bool ok = ox < g.gx && oy < g.gy && oz < g.gz;
                    const int64_t o = ok ? ((int64_t)ox * g.gy + oy) * g.gz + oz : 0;
                    g0[v] = gout ? Cvt<float, T>::ld(gc0[o]) : 1.f;             // (no grad_out: ones -- the backward of count)
                    g1[v] = two ? (gout ? Cvt<float, T>::ld(gc1[o]) : 1.f) : 0.f;
                }
                __syncthreads();                                     // the box is free (previous pair's readers are done)
#pragma unroll
                for (int v = 0; v < VPT; ++v) srcb[tid + NT * v] = make_float2(g0[v], g1[v]);
                __syncthreads();
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const float2 sv = srcb[(tl.key[j] >> 16) & (NS - 1)];
                    const bool on = (tl.key[j] >> 29) & 1;
                    go[j] = f2{ on ? sv.x : 0.f, on ? sv.y : 0.f };
                }
            }

            for (int ps = 0; ps < 4; ++ps) {
                const int tid = opaque((int)threadIdx.x);
                const int npl = (tl.S[0] - ps + 3) >> 2;            // box planes x = ps, ps + 4, ...
                // stage: slot (xq * 32 + y) * PZ + z = (sign * c0, sign * c1) of the wrapped lattice point
                const int nrow = npl * CAPY;
                const int omask = (p.dbg & 64) ? 0x3fff : -1;       // ablation: loads from a 64 KiB window
                __syncthreads();                                     // the previous pass's readers are done
                if (p.dbg & 1) {
                } else if (zlin) {
                    // the box's z-range lies inside the lattice and z is the unit-stride dim: rows are
                    // contiguous runs.  A thread moves QUADS of 4 slots: two 16-byte loads (one per
                    // channel), two 16-byte LDS stores.  (Narrow loads are what the vector L1 is slow at:
                    // it looks up one tag per cycle whatever the width -- 4-byte loads of the row ends
                    // alone cost as much as all the quads.)  ALL loads of the pass are issued before the
                    // first store: one exposed round trip.
                    const int nq = (tl.S[2] + 3) >> 2;               // quads per row; the last one is shifted to END at S_z
                    constexpr int QPR = PZ / 4, NU = (NPL * CAPY * QPR + NT - 1) / NT;
                    float4 a0[NU], a1[NU]; float sg[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int e = tid + NT * u, r = e / QPR, qd = e - r * QPR;
                        const bool on = qd < nq && r < nrow && (r & 31) < tl.S[1];
                        const int xr = on ? 4 * (r >> 5) + ps : 0, yr = on ? r & 31 : 0;
                        const int zs = 4 * qd + 4 <= tl.S[2] ? 4 * qd : tl.S[2] - 4;
                        const int off = (on ? sm.taboff[0][xr] + sm.taboff[1][yr] + tl.lo[2] + zs : 0) & omask;
                        sg[u] = sm.tabsgn[0][xr] * sm.tabsgn[1][yr];
                        a0[u] = ld4<T>(vc0 + off);
                        a1[u] = ld4<T>(vc1 + off);
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int e = tid + NT * u, r = e / QPR, qd = e - r * QPR;
                        if (qd < nq && r < nrow && (r & 31) < tl.S[1]) {
                            const int zs = 4 * qd + 4 <= tl.S[2] ? 4 * qd : tl.S[2] - 4;
                            float2 *dst = sm.box + r * PZ + zs;
                            if (!(zs & 1)) {
                                reinterpret_cast<float4 *>(dst)[0] = make_float4(a0[u].x * sg[u], a1[u].x * sg[u], a0[u].y * sg[u], a1[u].y * sg[u]);
                                reinterpret_cast<float4 *>(dst)[1] = make_float4(a0[u].z * sg[u], a1[u].z * sg[u], a0[u].w * sg[u], a1[u].w * sg[u]);
                            } else {                                 // shifted last quad of an odd extent: 8-byte stores
                                dst[0] = make_float2(a0[u].x * sg[u], a1[u].x * sg[u]); dst[1] = make_float2(a0[u].y * sg[u], a1[u].y * sg[u]);
                                dst[2] = make_float2(a0[u].z * sg[u], a1[u].z * sg[u]); dst[3] = make_float2(a0[u].w * sg[u], a1[u].w * sg[u]);
                            }
                        }
                    }
                } else {
                    // general case (the box wraps in z, or z is strided): slot by slot through the z table
                    constexpr int U = 8;
                    const int z = tid & 31;
                    const bool zin = z < tl.S[2];
                    const int oz = zin ? sm.taboff[2][z] : 0;
                    const float sgz = zin ? sm.tabsgn[2][z] : 0.f;
                    for (int r0 = tid >> 5; r0 < nrow; r0 += (NT / 32) * U) {
                        float v0[U], v1[U], sg[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int r = r0 + u * (NT / 32);
                            const bool on = zin && r < nrow && (r & 31) < tl.S[1];
                            const int xr = on ? 4 * (r >> 5) + ps : 0, yr = on ? r & 31 : 0;
                            const int off = on ? sm.taboff[0][xr] + sm.taboff[1][yr] + oz : 0;
                            sg[u] = on ? sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sgz : 0.f;
                            v0[u] = Cvt<float, T>::ld(vc0[off]);
                            v1[u] = Cvt<float, T>::ld(vc1[off]);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int r = r0 + u * (NT / 32);
                            if (zin && r < nrow && (r & 31) < tl.S[1]) sm.box[r * PZ + z] = make_float2(v0[u] * sg[u], v1[u] * sg[u]);
                        }
                    }
                    for (int e = tid; tl.S[2] > 32 && e < nrow * 4; e += NT) {      // slices 32 ... 35
                        const int r = e >> 2, z2 = 32 + (e & 3);
                        if ((r & 31) < tl.S[1] && z2 < tl.S[2]) {
                            const int xr = 4 * (r >> 5) + ps, yr = r & 31;
                            const int off = sm.taboff[0][xr] + sm.taboff[1][yr] + sm.taboff[2][z2];
                            const float sgn = sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sm.tabsgn[2][z2];
                            sm.box[r * PZ + z2] = make_float2(Cvt<float, T>::ld(vc0[off]) * sgn, Cvt<float, T>::ld(vc1[off]) * sgn);
                        }
                    }
                }
                __syncthreads();
                prof_mark(1);
                if (p.dbg & 2) continue;
                const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
                // (mixed orders: the loop below once per combination of per-dim orders, chosen by a uniform branch -- the orders are
                //  compile-time constants inside, like K in the isotropic kernels)
#ifdef IP_SORTED_MIX_TU
                auto taps = [&](auto kxc, auto kyc, auto kzc) {
                [[maybe_unused]] constexpr int KX = decltype(kxc)::value, KY = decltype(kyc)::value, KZ = decltype(kzc)::value;
#else
                {                                                    // (the isotropic module: no lambda -- its code is kept as it was, to the byte)
                [[maybe_unused]] constexpr int KX = K, KY = K, KZ = K;
#endif
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    // (opaque per pass: the weights are recomputed -- hoisted out of the pass loop they
                    // would be 9 more live values per sample, i.e. spilled)
                    float tx = tl.tx[j]; f2 tyz = tl.tyz[j];
                    asm volatile("" : "+v"(tx), "+v"(tyz));
                    const int key = tl.key[j];
                    const int x0 = key & 31;
                    const int i = (ps - x0) & 3;                     // the x-tap of this pass
                    const int xq = (x0 + (K == 3 || i <= K ? i : 0)) >> 2;
                    if (K == 1) {
                        // trilinear: two of the four passes hold an x-tap of the sample, four slots each
                        if (i <= 1) {
                            f2 t4[4];
                            stencil_reads_k1(boxaddr + 8u * (unsigned)(xq * PLANE + ((key >> 5) & 2047)), t4);
                            float sg4[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) sg4[k] = __builtin_fmaf(go[j].y, t4[k].y, go[j].x * t4[k].x);
                            const float wxi = i == 0 ? 1.f - tx : tx, gxi = i == 0 ? -1.f : 1.f;
                            const f2 w0 = 1.f - tyz, w1 = tyz;       // (.x: y, .y: z)
                            const float q0 = __builtin_fmaf(w1.y, sg4[1], w0.y * sg4[0]), q1 = __builtin_fmaf(w1.y, sg4[3], w0.y * sg4[2]);
                            const float z0 = sg4[1] - sg4[0], z1 = sg4[3] - sg4[2];
                            ag[j][0] = __builtin_fmaf(gxi, __builtin_fmaf(w1.x, q1, w0.x * q0), ag[j][0]);
                            ag[j][1] = __builtin_fmaf(wxi, q1 - q0, ag[j][1]);
                            ag[j][2] = __builtin_fmaf(wxi, __builtin_fmaf(w1.x, z1, w0.x * z0), ag[j][2]);
                        }
                        asm volatile("" : "+v"(ag[j][0]), "+v"(ag[j][1]), "+v"(ag[j][2]));
                        continue;
                    }
                    f2 t2[16];
                    stencil_reads(boxaddr + 8u * (unsigned)(xq * PLANE + ((key >> 5) & 2047)), t2);
                    float wxi, gxi;
                    f2 w[4], dq[4];
                    if constexpr (MIX) {
                        mixed_weights_yz(KY, KZ, tyz, w);
                        mixed_wgrads_yz(KY, KZ, tyz, dq);
                        wxi = mixed_weight_x(KX, tx, i);
                        gxi = mixed_wgrad_x(KX, tx, i);
                    } else {
                        wxi = weight_x<K>(tx, i); gxi = wgrad_x<K>(tx, i);
                        weights_yz<K>(tyz, w);
                        wgrads_yz<K>(tyz, dq);
                    }
                    // channels contracted with grad_out FIRST (s = g0 v0 + g1 v1 per tap), then the three derivative sums
                    // of a single image (pushpull.py:256-257)
                    float pp = 0.f, ppy = 0.f, ppz = 0.f;
#pragma unroll
                    for (int jy = 0; jy <= KY; ++jy) {                // (KY, KZ: = K, or the dims' own orders in the mixed kernels)
                        float q = 0.f, qz = 0.f;
#pragma unroll
                        for (int k = 0; k <= KZ; ++k) {
                            const float sgl = __builtin_fmaf(go[j].y, t2[4 * jy + k].y, go[j].x * t2[4 * jy + k].x);
                            q = __builtin_fmaf(w[k].y, sgl, q);
                            qz = __builtin_fmaf(dq[k].y, sgl, qz);
                        }
                        pp = __builtin_fmaf(w[jy].x, q, pp);
                        ppy = __builtin_fmaf(dq[jy].x, q, ppy);
                        ppz = __builtin_fmaf(w[jy].x, qz, ppz);
                    }
                    if constexpr (MIX) { if (i > KX) { pp = 0.f; ppy = 0.f; ppz = 0.f; } }     // (a plane beyond the x-stencil may hold anything)
                    ag[j][0] = __builtin_fmaf(gxi, pp, ag[j][0]);
                    ag[j][1] = __builtin_fmaf(wxi, ppy, ag[j][1]);
                    ag[j][2] = __builtin_fmaf(wxi, ppz, ag[j][2]);
                    asm volatile("" : "+v"(ag[j][0]), "+v"(ag[j][1]), "+v"(ag[j][2]));
                }
#ifdef IP_SORTED_MIX_TU
                };
                if constexpr (MIX) mix_dispatch(L.k[0], L.k[1], L.k[2], taps);
                else taps(std::integral_constant<int, K>{}, std::integral_constant<int, K>{}, std::integral_constant<int, K>{});
#else
                }
#endif
                prof_mark(2);
            }
            prof_mark(3);
            if (c + 2 >= p.C) break;
        }
        {
            // back to the natural order through LDS: (gx, gy, gz, -) per sample
            __syncthreads();
            const int tid = opaque((int)threadIdx.x);
            float4 *outb = reinterpret_cast<float4 *>(sm.box);
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int key = opaque(tl.key[j]);
                if (!((key >> 29) & 1)) continue;
                const float m = (float)((key >> 28) & 1);            // pushpull.py:256-257 with the mask of nd.py:139-140
                outb[(key >> 16) & (NS - 1)] = make_float4(ag[j][0] * m, ag[j][1] * m, ag[j][2] * m, 0.f);
            }
            // out-of-box samples: one wave per sample, lanes = taps, all channels, straight from global memory
            if (nslow > 0) {
                const int wave = tid >> 6, lane = tid & 63;
                for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                    int ox, oy, oz; float x[3];
                    sample_pos(g, sm.slow[sidx], ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    int off; float gr[3];
                    tiled::tap_weight_t<MIX ? -1 : K, MIX ? -1 : K>(L, x[0], x[1], x[2], lane, &off, gr);
                    float sgl = 0.f;
                    if (lane < (MIX ? (L.k[0] + 1) * (L.k[1] + 1) * (L.k[2] + 1) : (K + 1) * (K + 1) * (K + 1)))
                        for (int cc = 0; cc < p.C; ++cc)
                            sgl = __builtin_fmaf(gout ? Cvt<float, T>::ld(gout[b * p.val_sb + cc * p.val_sc + o]) : 1.f, Cvt<float, T>::ld(vol[b * p.vol_sb + cc * p.vol_sc + off]), sgl);
                    const float m = inb_mask(p, x);
                    const float a0 = wave_sum(gr[0] * sgl), a1 = wave_sum(gr[1] * sgl), a2 = wave_sum(gr[2] * sgl);
                    if (lane == 0) outb[sm.slow[sidx]] = make_float4(a0 * m, a1 * m, a2 * m, 0.f);
                }
            }
            // pathological tiles (slow list or hole table overflowed): the thread gathers its sample itself
            if (tl.selfmask) {
                for (int v = 0; v < VPT; ++v) {
                    if (!((tl.selfmask >> v) & 1)) continue;
                    int ox, oy, oz; float x[3];
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    int ii[3]; float tt[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) split(MIX ? L.k[d] : K, x[d], ii[d], tt[d]);
                    const float m = inb_mask(p, x);
                    float a[3] = { 0.f, 0.f, 0.f };
                    for (int cc = 0; cc < p.C; ++cc) {
                        const float gv = gout ? Cvt<float, T>::ld(gout[b * p.val_sb + cc * p.val_sc + o]) : 1.f;
                        for (int d = 0; d < 3; ++d)
                            a[d] = __builtin_fmaf(gv, tiled::gather_one_thread<T>(L, vol + b * p.vol_sb + cc * p.vol_sc, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], d), a[d]);
                    }
                    outb[tid + NT * v] = make_float4(a[0] * m, a[1] * m, a[2] * m, 0.f);
                }
            }
            __syncthreads();
            float *gb = ggrid + b * p.N * 3;                         // dense (B, *out, 3), whatever the batch stride of the grid (0: broadcast)
            if (g.ox0 + TS <= g.gx && g.oy0 + TS <= g.gy && g.oz0 + TS <= g.gz) {
                // whole tile: four z-neighbours = 12 contiguous floats, three 16-byte stores
#pragma unroll
                for (int u = 0; u < NS / 4 / NT; ++u) {
                    const int qi = tid + NT * u;                     // quad: x = qi >> 6, y = (qi >> 2) & 15, z = 4 (qi & 3)
                    const float4 r0 = outb[4 * qi], r1 = outb[4 * qi + 1], r2 = outb[4 * qi + 2], r3 = outb[4 * qi + 3];
                    const int64_t o = ((int64_t)(g.ox0 + (qi >> 6)) * g.gy + (g.oy0 + ((qi >> 2) & 15))) * g.gz + (g.oz0 + 4 * (qi & 3));
                    st4<float>(gb + 3 * o, make_float4(r0.x, r0.y, r0.z, r1.x));
                    st4<float>(gb + 3 * o + 4, make_float4(r1.y, r1.z, r2.x, r2.y));
                    st4<float>(gb + 3 * o + 8, make_float4(r2.z, r3.x, r3.y, r3.z));
                }
            } else {
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    int ox, oy, oz;
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    if (!(ox < g.gx && oy < g.gy && oz < g.gz)) continue;
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    const float4 r = outb[tid + NT * v];
                    gb[3 * o] = r.x; gb[3 * o + 1] = r.y; gb[3 * o + 2] = r.z;
                }
            }
        }
        __syncthreads();                                             // the next tile reuses the LDS tables / lists
    }
}


// ===========================================================================
// push / count : vol[b,c,tap] += w * mask * val[b,c,o]      (nd.py:146-213, pushpull.py:106-142)
//
// Same tiles, same sort.  The contributions of a tile are accumulated in the LDS box in FIXED
// POINT (LDS float atomics retire 0.3 lanes/clk on gfx950, integer ones 20x that -- and with
// class-sorted lanes a ds_add_u64 costs 6.3 clk instead of 12.6: tools/microbench/lds_gather.hip),
// two channels per 64-bit slot:  W += (q1 << 32) + sext(q0),  q = floor(src * w * 2^(29-ex-hb) + 1/2),
// where 2^ex bounds the tile's |src| (per channel) and hb is the headroom that the measured sample
// density demands (tiled::headroom32): neither 32-bit field can overflow, so they separate exactly
// at the flush, lo = (int32) W, hi = (W - lo) >> 32.  After every pass (box planes x = ps mod 4)
// the touched slots are added to the target with coalesced global atomics (slot sign applied,
// several slots may alias one lattice point under the boundary condition) and re-zeroed.
// Tiles the 32-bit fields cannot serve (strongly contracting deformations: density too high for the
// precision rule; non-finite sources) scatter tap-parallel straight to global memory, like the
// samples that fall outside the box.
//   MODE 0: values, 1: count (sources are ones; the target has one channel), 2: values + count in
//   one pass (INTERPOL_FLAG_WITH_COUNT: the target has C + 1 channels, the last receives the count)
// ===========================================================================
// the 4 LDS adds of one row of one x-plane of a stencil, at immediate offsets
static_assert(PZ == 36, "the immediate offsets below are (row * PZ + k) * 8");
#define IP_ADDROW(R) \
template <> __device__ __forceinline__ void row_adds<R>(unsigned addr, unsigned long long v0, unsigned long long v1, unsigned long long v2, unsigned long long v3) \
{ \
    asm volatile("ds_add_u64 %0, %1 offset:%5\n\tds_add_u64 %0, %2 offset:%6\n\tds_add_u64 %0, %3 offset:%7\n\tds_add_u64 %0, %4 offset:%8" \
                 :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(R * PZ * 8), "n"(R * PZ * 8 + 8), "n"(R * PZ * 8 + 16), "n"(R * PZ * 8 + 24) : "memory"); \
}
template <int R> __device__ __forceinline__ void row_adds(unsigned addr, unsigned long long v0, unsigned long long v1, unsigned long long v2, unsigned long long v3);
IP_ADDROW(0) IP_ADDROW(1) IP_ADDROW(2) IP_ADDROW(3)
#undef IP_ADDROW

template <typename T, int K, int GM, int MODE>
__global__ __launch_bounds__(NT, 4) void push_sorted(KParams p, const T *__restrict__ val, const float *__restrict__ grid,
                                                     float *__restrict__ vol, int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch,
                                                     DeferArgs defer)
{
    if (p.gate && *p.gate) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / 4; L.k[d] = K; }   // the target is float
    L.lin = 0;
    const int nch = MODE == 1 ? 1 : p.C + (MODE == 2 ? 1 : 0);       // target channels
    const WorkRange wr(ntiles * nbatch, false);
    float cnext[VPT][3];
    for (int work = wr.first; work < wr.end; work += wr.step) {
        const int tid = opaque((int)threadIdx.x);
        const int64_t b = work / ntiles;
        const TileGeom g = tile_geom(work % ntiles, gx, gy, gz, nty, ntz);
        prof_mark(-1);
        Tile<K, GM> tl;
        Tile<K, GM>::load(p, grid, b, g, tid, cnext);
        tl.build(p, L, g, sm, tid, cnext);
        const int nslow = sm.nslow < SLOWCAP ? sm.nslow : SLOWCAP;
        if (defer.flag) {                                                 // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
            bool hand_back = sm.nslow > ((3 * HANDBACK / 2) << ((p.dbg >> 9) & 7));   // scatter: 3/16 of the samples, see tiled::hand_back
            if (hand_back) hand_back = tiled::tile_smooth(p, grid, b, 3, g.ox0, g.oy0, g.oz0, TS, TS, TS, g.gx, g.gy, g.gz, sm.hi);
            if (hand_back && tid == 0) defer_mark(defer, work, tile_desc(b, g.ox0 / TS, g.oy0 / TS, g.oz0 / TS));
            if (hand_back && defer.desc) { __syncthreads(); continue; }
        }
        // ---- sample density: the largest number of sorted samples that share a first-tap cell bounds
        // what any lattice point can receive.  Counted in the (free) box: 16-bit counters, two per word.
        unsigned *cnt32 = reinterpret_cast<unsigned *>(sm.box);
        {
            float4 *z4 = reinterpret_cast<float4 *>(sm.box);
            for (int e = tid; e < BOXSLOTS / 2; e += NT) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid == 0) { sm.hi[0] = 0; }
            if (tid < 8) sm.cmax[tid] = 0;
        }
        __syncthreads();
        int cell[VPT];
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            cell[j] = (tl.key[j] & 31) * PLANE + ((tl.key[j] >> 5) & 2047);
            if ((tl.key[j] >> 29) & 1) atomicAdd(&cnt32[cell[j] >> 1], 1u << (16 * (cell[j] & 1)));
        }
        __syncthreads();
        {
            int m = 0;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                const int cv = (int)((cnt32[cell[j] >> 1] >> (16 * (cell[j] & 1))) & 0xffffu);
                m = ((tl.key[j] >> 29) & 1) && cv > m ? cv : m;
            }
            m = wave_max(m);
            if ((tid & 63) == 0 && m > 0) atomicMax(&sm.hi[0], m);
        }
        __syncthreads();
        const int hb = tiled::headroom32(L, sm.hi[0]);                // < 0: the 32-bit fields are not precise enough here
        {
            float4 *z4 = reinterpret_cast<float4 *>(sm.box);
            for (int e = tid; e < BOXSLOTS / 2; e += NT) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        prof_mark(0);

        for (int c = 0; c < nch; c += 2) {
            const int tid = opaque((int)threadIdx.x);
            const bool two = c + 1 < nch;
            // channel c (and c + 1) of this batch item; ones: the count image
            const bool ones0 = MODE == 1 || (MODE == 2 && c >= p.C), ones1 = MODE == 1 || (MODE == 2 && c + 1 >= p.C);
            const T *ic0 = ones0 ? nullptr : val + b * p.val_sb + c * p.val_sc;
            const T *ic1 = (ones1 || !two) ? nullptr : val + b * p.val_sb + (c + 1) * p.val_sc;
            float *vc0 = vol + b * p.vol_sb + c * p.vol_sc;
            float *vc1 = two ? vc0 + p.vol_sc : vc0;
            // sources of the tile in the natural order -- coalesced; the sorted order would be 16 scattered 4-byte loads
            // per thread (measured: 37 % of the kernel) -- handed to the sorted lanes through the (zero, free) box
            f2 src[VPT];
            float am0 = 0.f, am1 = 0.f;
            {
                float2 *srcb = reinterpret_cast<float2 *>(sm.box);
                float s0[VPT], s1[VPT];
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    int ox, oy, oz;
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    const bool ok = ox < g.gx && oy < g.gy && oz < g.gz;
                    const int64_t o = ok ? ((int64_t)ox * g.gy + oy) * g.gz + oz : 0;
                    s0[v] = ones0 ? 1.f : Cvt<float, T>::ld(ic0[o]);
                    s1[v] = !two ? 0.f : (ones1 ? 1.f : Cvt<float, T>::ld(ic1[o]));
                }
#pragma unroll
                for (int v = 0; v < VPT; ++v) srcb[tid + NT * v] = make_float2(s0[v], s1[v]);
                __syncthreads();
#pragma unroll
                for (int j = 0; j < VPT; ++j) {
                    const bool on = (tl.key[j] >> 29) & 1;
                    const float m = (float)((tl.key[j] >> 28) & 1);                  // masked: nd.py:201-203
                    const float2 sv = srcb[(tl.key[j] >> 16) & (NS - 1)];
                    const float a = on ? sv.x * m : 0.f, c_ = on ? sv.y * m : 0.f;
                    src[j] = f2{ a, c_ };
                    const float a0 = __builtin_fabsf(a), a1 = __builtin_fabsf(c_);
                    am0 = (a0 > am0 || a0 != a0) ? a0 : am0;             // NaN sticks
                    am1 = (a1 > am1 || a1 != a1) ? a1 : am1;
                }
                __syncthreads();
                float4 *z4 = reinterpret_cast<float4 *>(sm.box);
#pragma unroll
                for (int v = 0; v < NS / 2 / NT; ++v) z4[tid + NT * v] = make_float4(0.f, 0.f, 0.f, 0.f);   // the box is zero again
            }
            {
                const int b0 = wave_max(__float_as_int(am0)), b1 = wave_max(__float_as_int(am1));   // non-negative floats (and NaN) order like ints
                if ((tid & 63) == 0) { if (b0) atomicMax(&sm.cmax[0], b0); if (b1) atomicMax(&sm.cmax[1], b1); }
            }
            __syncthreads();                                         // maxima complete; the box is zero
            const int mb0 = sm.cmax[0], mb1 = sm.cmax[1];
            const bool fin = (mb0 & 0x7f800000) != 0x7f800000 && (mb1 & 0x7f800000) != 0x7f800000;
            const bool fixedpt = hb >= 0 && fin && !(p.dbg & 8);
            int ex0 = ((mb0 >> 23) & 0xff) - 127, ex1 = ((mb1 >> 23) & 0xff) - 127;
            ex0 = ex0 < -90 ? -90 : ex0; ex1 = ex1 < -90 ? -90 : ex1;
            const int hbc = hb < 0 ? 0 : hb;
            const f2 scale = { mb0 ? __int_as_float((127 + 29 - ex0 - hbc) << 23) : 0.f, mb1 ? __int_as_float((127 + 29 - ex1 - hbc) << 23) : 0.f };
            const float inv0 = __int_as_float((127 - 29 + ex0 + hbc) << 23), inv1 = __int_as_float((127 - 29 + ex1 + hbc) << 23);
            prof_mark(4);
            if (fixedpt) {
                const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
                for (int ps = 0; ps < 4; ++ps) {
                    const int tid = opaque((int)threadIdx.x);
                    if (tid < NPL * CAPY) {                          // row table of this pass (read by the flush)
                        const int xr = 4 * (tid >> 5) + ps, yr = tid & 31;
                        if (xr < tl.S[0] && yr < tl.S[1])
                            sm.rowtab[tid] = make_int2(sm.taboff[0][xr] + sm.taboff[1][yr], __float_as_int(sm.tabsgn[0][xr] * sm.tabsgn[1][yr]));
                    }
                    if (!(p.dbg & 2)) {
#pragma unroll
                    for (int j = 0; j < VPT; ++j) {
                        float tx = tl.tx[j]; f2 tyz = tl.tyz[j];
                        asm volatile("" : "+v"(tx), "+v"(tyz));
                        const int key = tl.key[j];
                        const int x0 = key & 31;
                        const int i = (ps - x0) & 3;                 // the x-tap of this pass
                        const int xq = (x0 + (K == 3 || i <= K ? i : 0)) >> 2;
                        const unsigned addr = boxaddr + 8u * (unsigned)(xq * PLANE + ((key >> 5) & 2047));
                        const float wxi = weight_x<K>(tx, i);
                        f2 w[4];
                        weights_yz<K>(tyz, w);
                        const f2 sx = src[j] * scale * f2{ wxi, wxi };
#pragma unroll
                        for (int jy = 0; jy <= K; ++jy) {
                            const f2 sy = sx * f2{ w[jy].x, w[jy].x };
                            unsigned long long v[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const f2 pr = sy * f2{ w[k].y, w[k].y };
                                const int q0 = tiled::cvt_rpi(pr.x), q1 = tiled::cvt_rpi(pr.y);
                                // (q1 << 32) + sext(q0): low word q0, high word q1 + (q0 < 0 ? -1 : 0)
                                v[k] = ((unsigned long long)(unsigned)(q1 + (q0 >> 31)) << 32) | (unsigned)q0;
                            }
                            if (jy == 0) row_adds<0>(addr, v[0], v[1], v[2], v[3]);
                            else if (jy == 1) row_adds<1>(addr, v[0], v[1], v[2], v[3]);
                            else if (jy == 2) row_adds<2>(addr, v[0], v[1], v[2], v[3]);
                            else row_adds<3>(addr, v[0], v[1], v[2], v[3]);
                        }
                    }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                    prof_mark(1);
                    if (!(p.dbg & 1)) {
                        // flush the planes of this pass: fixed point -> float, slot sign, one global atomic per touched
                        // slot and channel; touched slots are re-zeroed on the way.  Lanes run along z (a wave
                        // instruction covers whole rows: the L2 performs atomics line by line), 14 rows at a time.
                        const int npl = (tl.S[0] - ps + 3) >> 2;
                        const int nrow = npl * CAPY;
                        // Slices z < 32 go 32 lanes per row, two rows per instruction, four rows read before the first is used; the
                        // (up to four) slices z >= 32 go four lanes per row, sixteen rows per instruction -- not one nearly empty
                        // instruction per row pair.  (Reading ALL the thread's slots in one batch pushed the kernel into scratch:
                        // 0.93 -> 1.48 ms for everything else.)
                        // Measured (tools/ablate_sorted.py pushs, profiles/r02_push_ablation.txt; config 2, sigma = 2): set-up +
                        // sources 0.94 ms, tap loop 0.80, LDS side of the flush 0.17, its global atomics 1.55 -- additive: the L2
                        // retires ~0.33 G float lane-atomics per ms chip-wide (one per clock and channel), however they are
                        // coalesced, ordered over the tiles, or overlapped (fetching the next tile's coordinates before the last
                        // flush, or starting half the workgroups half a tile late, changed nothing).
                        auto flush_slot = [&](int r, int z, int offz, float f0, float f1, long long a) {
                            *reinterpret_cast<long long *>(sm.box + r * PZ + z) = 0ll;
                            if (p.dbg & 4) return;                     // (ablation: everything but the global atomics)
                            const int2 rt = sm.rowtab[r];
                            const int lo_ = (int)(a & 0xffffffffll);
                            const int hi_ = (int)((a - (long long)lo_) >> 32);
                            const int off = rt.x + offz;
                            const float sg = __int_as_float(rt.y);
                            if (lo_ != 0) __hip_atomic_fetch_add(vc0 + off, (float)lo_ * (f0 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (hi_ != 0) __hip_atomic_fetch_add(vc1 + off, (float)hi_ * (f1 * sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        };
                        {
                            const int z = tid & 31, rr = tid >> 5;
                            if (z < tl.S[2]) {
                                const int offz = sm.taboff[2][z];
                                const float f0 = inv0 * sm.tabsgn[2][z], f1 = inv1 * sm.tabsgn[2][z];
                                constexpr int RS = NT / 32, UF = 4;
                                for (int r0 = rr; r0 < nrow; r0 += UF * RS) {
                                    long long a[UF];
#pragma unroll
                                    for (int u = 0; u < UF; ++u) {
                                        const int r = r0 + u * RS;
                                        a[u] = (r < nrow && (r & 31) < tl.S[1]) ? *reinterpret_cast<long long *>(sm.box + r * PZ + z) : 0ll;
                                    }
#pragma unroll
                                    for (int u = 0; u < UF; ++u) {
                                        if (a[u] != 0) flush_slot(r0 + u * RS, z, offz, f0, f1, a[u]);
                                        else if ((p.dbg & 16) && r0 + u * RS < nrow && ((r0 + u * RS) & 31) < tl.S[1])     // (ablation: an atomic for every slot of the box, touched or not)
                                            flush_slot(r0 + u * RS, z, offz, f0, f1, 0x100000001ll);
                                    }
                                }
                            }
                        }
                        if (tl.S[2] > 32) {
                            const int z = 32 + (tid & 3), rr = tid >> 2;
                            if (z < tl.S[2]) {
                                const int offz = sm.taboff[2][z];
                                const float f0 = inv0 * sm.tabsgn[2][z], f1 = inv1 * sm.tabsgn[2][z];
                                for (int r = rr; r < nrow; r += NT / 4) {
                                    if ((r & 31) >= tl.S[1]) continue;
                                    const long long a = *reinterpret_cast<long long *>(sm.box + r * PZ + z);
                                    if (a != 0) flush_slot(r, z, offz, f0, f1, a);
                                }
                            }
                        }
                    }
                    __syncthreads();
                    prof_mark(2);
                }
            } else {
                // no fixed point for this tile: every sorted sample tap-parallel, one wave per sample (lanes =
                // taps), float atomics straight to global memory
                const int lane = tid & 63;
                for (int j = 0; j < VPT; ++j) {
                    for (int l = 0; l < 64; ++l) {
                        const int key = __shfl(tl.key[j], l);
                        if (!((key >> 29) & 1)) continue;            // wave-uniform
                        const float tx = __shfl(tl.tx[j], l), ty = __shfl(tl.tyz[j].x, l), tz = __shfl(tl.tyz[j].y, l);
                        const float s0 = __shfl(src[j].x, l), s1 = __shfl(src[j].y, l);
                        // coordinates back from (first tap, t): x = i0 + t
                        const int yz = (key >> 5) & 2047;
                        const float x = (float)(tl.lo[0] + (key & 31)) + tx, y = (float)(tl.lo[1] + yz / PZ) + ty, z = (float)(tl.lo[2] + yz % PZ) + tz;
                        int off;
                        const float wt = tiled::tap_weight_t<K, K>(L, x, y, z, lane, &off, nullptr);
                        if (lane < (K + 1) * (K + 1) * (K + 1)) {
                            __hip_atomic_fetch_add(vc0 + off, wt * s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (two) __hip_atomic_fetch_add(vc1 + off, wt * s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            // out-of-box samples: one wave per sample, lanes = taps
            if (nslow > 0 || tl.selfmask) {
                const int wave = tid >> 6, lane = tid & 63;
                for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                    int ox, oy, oz; float x[3];
                    sample_pos(g, sm.slow[sidx], ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    const float m = inb_mask(p, x);
                    const float s0 = (ones0 ? 1.f : Cvt<float, T>::ld(ic0[o])) * m;
                    const float s1 = !two ? 0.f : (ones1 ? 1.f : Cvt<float, T>::ld(ic1[o])) * m;
                    int off;
                    const float wt = tiled::tap_weight_t<K, K>(L, x[0], x[1], x[2], lane, &off, nullptr);
                    if (lane < (K + 1) * (K + 1) * (K + 1)) {
                        __hip_atomic_fetch_add(vc0 + off, wt * s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (two) __hip_atomic_fetch_add(vc1 + off, wt * s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // pathological tiles (slow list or hole table overflowed): the thread scatters its sample itself
                for (int v = 0; v < VPT; ++v) {
                    if (!((tl.selfmask >> v) & 1)) continue;
                    int ox, oy, oz; float x[3];
                    sample_pos(g, tid + NT * v, ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    const int64_t o = ((int64_t)ox * g.gy + oy) * g.gz + oz;
                    int ii[3]; float tt[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) split(K, x[d], ii[d], tt[d]);
                    const float m = inb_mask(p, x);
                    tiled::scatter_one_thread(L, vc0, (ones0 ? 1.f : Cvt<float, T>::ld(ic0[o])) * m, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2]);
                    if (two) tiled::scatter_one_thread(L, vc1, (ones1 ? 1.f : Cvt<float, T>::ld(ic1[o])) * m, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2]);
                }
            }
            __syncthreads();
            if (tid < 8) sm.cmax[tid] = 0;                           // for the next channel pair
            prof_mark(3);
        }
        __syncthreads();                                             // the next tile reuses the LDS tables / lists
    }
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
struct TileCount {
    int gx, gy, gz, ntx, nty, ntz;
    explicit TileCount(const interpol_problem *p)
    {
        gx = (int)p->grid_shape[0]; gy = (int)p->grid_shape[1]; gz = (int)p->grid_shape[2];
        ntx = (gx + TS - 1) / TS; nty = (gy + TS - 1) / TS; ntz = (gz + TS - 1) / TS;
    }
    int ntiles() const { return ntx * nty * ntz; }
    dim3 grid(int B) const
    {
        const long long total = (long long)ntiles() * B, want = 2ll * cu_count();
        return dim3((unsigned)(total < want ? total : want), 1u);
    }
    // one workgroup per tile: 8 * ceil(total / 8) workgroups, of which WorkRange (tile_common.hpp) gives each exactly one tile
    // of its XCD's eighth -- the dispatcher, not a stride, balances tiles of unequal cost; `mult` tiles per workgroup otherwise
    dim3 grid_each(int B, int mult) const
    {
        const long long total = (long long)ntiles() * B;
        if (total < 8) return dim3((unsigned)total, 1u);
        const long long per = (total + 7) >> 3, wg = (per + mult - 1) / mult;
        return dim3((unsigned)(8 * wg), 1u);
    }
};

template <typename T, int K, int GM, bool MIX = false>
static int launch_pull(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    const int attr = big_lds<pull_sorted<T, K, GM, MIX>>(sizeof(Smem));
    if (attr) return attr;
    const TileCount t(p);
    const Defer df(k, st, t.ntiles(), p->batch, t.ntx, t.nty, t.ntz, TS, TS, TS);
    // two tiles per workgroup, dealt by the dispatcher as workgroups retire: 1.36 -> 1.28 ms at config 2 against two persistent
    // workgroups per CU walking equal shares (tiles differ in cost: slow lists, box sizes); debug bits 13-15: 7 = persistent, 1-4 = tiles
    const int mopt = (k.dbg >> 13) & 7, mult = mopt == 0 ? 2 : mopt;
    hipLaunchKernelGGL((pull_sorted<T, K, GM, MIX>), mopt == 7 ? t.grid((int)p->batch) : t.grid_each((int)p->batch, mult), dim3(NT), sizeof(Smem), st,
                       k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = df.desc ? DeferOps<T>::pull(k, vol, grid, val, df.tl, st) : 0;
    return rc ? rc : 1;
}

template <typename T, int K, int GM, bool MIX = false>
static int launch_gradc(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st)
{
    const int attr = big_lds<gradc_sorted<T, K, GM, MIX>>(sizeof(Smem));
    if (attr) return attr;
    const TileCount t(p);
    const Defer df(k, st, t.ntiles(), p->batch, t.ntx, t.nty, t.ntz, TS, TS, TS);
    const int mopt = (k.dbg >> 13) & 7, mult = mopt == 0 ? 2 : mopt;   // (as launch_pull)
    // (behind the probe of interpol_pull_backward's router -- gate_n < 0 -- the kernel usually returns at once: two workgroups per CU
    //  instead of 8192 that are dispatched for nothing, 50 us)
    hipLaunchKernelGGL((gradc_sorted<T, K, GM, MIX>), (mopt == 7 || (k.gate && k.gate_n < 0)) ? t.grid((int)p->batch) : t.grid_each((int)p->batch, mult), dim3(NT), sizeof(Smem), st,
                       k, (const T *)vol, (const T *)gout, (const float *)grid, (float *)ggrid, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = df.template gradc<T>(k, gout, vol, grid, ggrid, st);
    return rc ? rc : 1;
}

// `vol` is the zero-filled (or accumulating) FLOAT target; val == NULL: count
template <typename T, int K, int GM>
static int launch_push(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    const TileCount t(p);
    const Defer df(k, st, t.ntiles(), p->batch, t.ntx, t.nty, t.ntz, TS, TS, TS);
#define IP_LAUNCH_PUSH(MODE)                                                                                          \
    {                                                                                                                 \
        const int attr = big_lds<push_sorted<T, K, GM, MODE>>(sizeof(Smem));                                          \
        if (attr) return attr;                                                                                        \
        hipLaunchKernelGGL((push_sorted<T, K, GM, MODE>), t.grid((int)p->batch), dim3(NT), sizeof(Smem), st,          \
                           k, (const T *)val, (const float *)grid, (float *)vol, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args); \
    }
    if (!val) IP_LAUNCH_PUSH(1)
    else if (k.cc) IP_LAUNCH_PUSH(2)
    else IP_LAUNCH_PUSH(0)
#undef IP_LAUNCH_PUSH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = df.template push<T>(k, val, grid, vol, st);
    return rc ? rc : 1;
}

} // namespace sorted

// Eligibility: 3-D, one order 2..3 for all dims, enough samples, 32-bit offsets into one item's grid.
constexpr int MIXED = 13;                                          // sorted_order: orders 1..3, not all equal
static int sorted_order(const interpol_problem *p, const KParams &k, bool linear = false)
{
    if (p->dim != 3 || p->batch > 65535) return -1;
    if (k.dbg & 32) return -1;                                     // A/B switch: the natural-order tiles of ops_tiled.hip
    int64_t n = 1, nt = p->batch;
    for (int d = 0; d < 3; ++d) {
        if (p->grid_shape[d] > 0x7fffffff / 4) return -1;
        n *= p->grid_shape[d];
        nt *= (p->grid_shape[d] + 15) / 16;
    }
    if (n < 4096 || nt > 0x7fffffff) return -1;
    if ((uint64_t)n * 12ull > 0xffffffffull) return -1;
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2]) {
        // mixed orders 1..3 (round 6): the cubic tiles with per-dim runtime weights -- gathers on dense grids only
        for (int d = 0; d < 3; ++d) if (k.order[d] < 1 || k.order[d] > 3) return -1;
        return linear && k.sep == 0 ? MIXED : -1;
    }
    if (k.order[0] < (linear ? 1 : 2) || k.order[0] > 3) return -1;
    return k.order[0];
}

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

// The mixed-order instantiations live in a translation unit of their own (the same file compiled with -DIP_SORTED_MIX_TU): with them in
// one module the register allocation of the ISOTROPIC kernels changes (pull_sorted<float, 3, 0>: 96 -> 128 B of scratch, +3.5 % at config 2).
int IP_SYM(sorted_pull_mixed_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st);
int IP_SYM(sorted_gradc_mixed_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st);
#ifdef IP_SORTED_MIX_TU
int IP_SYM(sorted_pull_mixed_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{ return sorted::launch_pull<IP_TT, 3, 0, true>(p, k, vol, grid, val, st); }
int IP_SYM(sorted_gradc_mixed_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st)
{ return sorted::launch_gradc<IP_TT, 3, 0, true>(p, k, gout, vol, grid, ggrid, st); }
#else

#ifdef IP_EXPERIMENTS
// the windowed gather (experiments/pull_window.hip, `make experiments`): every sample visited once -- measured slower, not in the product
int IP_SYM(try_window_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, int K, const void *vol, const void *grid, void *val, hipStream_t st);
#endif

// returns 1 when it took the problem, 0 to decline, anything else: error
int IP_SYM(try_sorted_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    const int K = sorted_order(p, k, true);
    if (K < 0) return 0;
    if (K == MIXED) return (k.dbg & 16) ? 0 : IP_SYM(sorted_pull_mixed_, IP_TSFX)(p, k, vol, grid, val, st);
    if (K == 1) {                                                    // trilinear (round 5): dense grids and displacement fields
        using T1 = IP_TT;
        if (k.sep == 0) return sorted::launch_pull<T1, 1, 0>(p, k, vol, grid, val, st);
        if constexpr (std::is_same<T1, float>::value) {
            return k.sep == 1 ? sorted::launch_pull<T1, 1, 1>(p, k, vol, grid, val, st)
                 : (k.sep == 2 ? sorted::launch_pull<T1, 1, 2>(p, k, vol, grid, val, st) : sorted::launch_pull<T1, 1, 3>(p, k, vol, grid, val, st));
        }
        return 0;
    }
#ifdef IP_EXPERIMENTS
    if (k.dbg & 4096) {                                            // opt-in: the windowed gather (experimental: 1.75 ms against 1.35 ms at config 2)
        const int rc = IP_SYM(try_window_pull_, IP_TSFX)(p, k, K, vol, grid, val, st);
        if (rc != 0) return rc;
    }
#endif
    using T = IP_TT;
    if (k.sep) {
        if constexpr (std::is_same<T, float>::value) {
            if (K == 3) return k.sep == 1 ? sorted::launch_pull<T, 3, 1>(p, k, vol, grid, val, st)
                             : (k.sep == 2 ? sorted::launch_pull<T, 3, 2>(p, k, vol, grid, val, st) : sorted::launch_pull<T, 3, 3>(p, k, vol, grid, val, st));
            return k.sep == 1 ? sorted::launch_pull<T, 2, 1>(p, k, vol, grid, val, st)
                 : (k.sep == 2 ? sorted::launch_pull<T, 2, 2>(p, k, vol, grid, val, st) : sorted::launch_pull<T, 2, 3>(p, k, vol, grid, val, st));
        } else {
            return 0;
        }
    }
    if (K == 3) return sorted::launch_pull<T, 3, 0>(p, k, vol, grid, val, st);
    return sorted::launch_pull<T, 2, 0>(p, k, vol, grid, val, st);
}

// grid gradient of pull alone (the image gradient of the same backward is a push of grad_out): dense grids and
// displacement fields; 1 when it took the problem, 0 to decline
int IP_SYM(try_sorted_gradc_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st)
{
    const int K = sorted_order(p, k, true);
    if (K < 0 || (k.dbg & 16)) return 0;
    using T = IP_TT;
    if (K == MIXED) return IP_SYM(sorted_gradc_mixed_, IP_TSFX)(p, k, gout, vol, grid, ggrid, st);
    if (K == 1) {                                                    // trilinear (round 5; mode iso1: every dim linear)
        if (k.mode != MODE_ISO1) return 0;
        if (k.sep == 0) return sorted::launch_gradc<T, 1, 0>(p, k, gout, vol, grid, ggrid, st);
        if constexpr (std::is_same<T, float>::value) { if (k.sep == 2) return sorted::launch_gradc<T, 1, 2>(p, k, gout, vol, grid, ggrid, st); }
        return 0;
    }
    if (k.sep == 2) {
        if constexpr (std::is_same<T, float>::value) {
            if (K == 3) return sorted::launch_gradc<T, 3, 2>(p, k, gout, vol, grid, ggrid, st);
            return sorted::launch_gradc<T, 2, 2>(p, k, gout, vol, grid, ggrid, st);
        } else {
            return 0;
        }
    }
    if (k.sep) return 0;
    if (K == 3) return sorted::launch_gradc<T, 3, 0>(p, k, gout, vol, grid, ggrid, st);
    return sorted::launch_gradc<T, 2, 0>(p, k, gout, vol, grid, ggrid, st);
}

// Not the default yet: at BASELINE config 2 the class-sorted scatter ties with the natural-order
// tiles of ops_tiled.hip (3.6 ms both: the tap loop drops from 1.8 to 0.7 ms, but the flush -- 1.2 G
// global atomics for the tile halos, the same in both -- is then exposed).  dbg bit 128 selects it.
int IP_SYM(try_sorted_push_, IP_TSFX)(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{
    if (!(k.dbg & 128)) return 0;
    const int K = sorted_order(p, k);
    if (K < 0) return 0;
    using T = IP_TT;
    if (k.sep == 3) return 0;                                        // affine lattices: the generic scatter
    if (k.sep) {
        if constexpr (std::is_same<T, float>::value) {
            if (K == 3) return k.sep == 1 ? sorted::launch_push<T, 3, 1>(p, k, val, grid, vol, st) : sorted::launch_push<T, 3, 2>(p, k, val, grid, vol, st);
            return k.sep == 1 ? sorted::launch_push<T, 2, 1>(p, k, val, grid, vol, st) : sorted::launch_push<T, 2, 2>(p, k, val, grid, vol, st);
        } else {
            return 0;
        }
    }
    if (K == 3) return sorted::launch_push<T, 3, 0>(p, k, val, grid, vol, st);
    return sorted::launch_push<T, 2, 0>(p, k, val, grid, vol, st);
}

#endif // IP_SORTED_MIX_TU

} // namespace ip

#if defined(IP_PROF) && !defined(IP_SORTED_MIX_TU)
#define IP_PROF_NAME3(s) interpol_debug_prof_sorted_##s
#define IP_PROF_NAME2(s) IP_PROF_NAME3(s)
extern "C" __attribute__((visibility("default"))) int IP_PROF_NAME2(IP_TSFX)(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::sorted::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::sorted::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
