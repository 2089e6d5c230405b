// ===========================================================================
// pull_window.hip -- grid_pull on class-sorted LDS tiles with a SLIDING WINDOW of box planes:
// 3-D, one spline order 2..3 for all dims, f32 / bf16 / f16 storage (fp32 math), any boundary /
// extrapolation mode, all four coordinate sources.
//   pull (gather): reference interpol/nd.py:80-143 (weights splines.py:30-80, indices bounds.py:30-89)
//
// What it replaces and why.  pull_sorted (ops_sorted.hip) keeps a quarter of the tile's box in LDS
// at a time (planes x = pass mod 4) and visits every sample in each of its four passes: the weights,
// the slot address and the bookkeeping of a sample are evaluated four times, and the kernel is bound
// by the VALU instructions it issues (478 per sample, 84 of them the packed FMAs of the stencil).
// Here every sample is visited ONCE:
//
//   1. a 512-thread workgroup owns a tile of 16^3 samples; two workgroups per CU (79 KiB of LDS each);
//   2. LDS holds a RING of 8 planes of the box (plane x in ring slot x mod 8; 32 rows of 36 slots of
//      8 bytes, two channels per slot).  The first-tap planes x0 of the samples are cut into WINDOWS of
//      RING - K consecutive planes (5 for cubic): all K + 1 x-planes of a sample of window w are among
//      the 8 planes [w * SPAN, w * SPAN + 8), which the ring holds while the window is processed.  Going
//      from one window to the next stages SPAN new planes over the slots of the SPAN oldest: every plane
//      of the box is staged once per tile, as before;
//   3. the samples are counting-sorted by (window, bank class of the first slot): ONE returning LDS add
//      per sample on a (first plane, class) histogram, from which a small planner derives, per window,
//      the number of half-wave rows, the rank of every sample inside its (window, class) and the holes
//      that the surplus of over-full classes fills (sorted_util / ops_sorted.hip explain the classes:
//      lane l of a half wave holds a sample whose first slot is in class l, so that all 32 lanes of a
//      ds_read_b64 group hit different bank pairs).  A window's rows are dealt to the 8 waves cyclically;
//      a thread keeps the records of its <= 9 rows in registers, each tagged with its window;
//   4. per window: barrier, stage, barrier, and every wave runs the rows it holds of that window: per
//      sample 12 weights, 4 plane addresses, 4 blocks of 16 ds_read_b64 at immediate offsets and 84
//      packed FMAs -- once;
//   5. results return to the natural order through LDS, stores are 16 bytes wide; samples whose support
//      leaves the (clamped) box are gathered tap-parallel by whole waves from global memory, tiles that
//      a smooth but stretched lattice spreads too far are handed back to the generic kernel (defer.hip).
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace window {

using namespace sorted;

constexpr int NT = 512;                         // threads per workgroup
constexpr int NS = TS * TS * TS;                // samples per tile
constexpr int VPT = NS / NT;                    // samples per thread (natural order)
constexpr int CAPX = 32, CAPY = 32, CAPZ = 36;  // box capacity (lattice points)
constexpr int PZ = 36;                          // row pitch (8-byte slots)
constexpr int PLANE = CAPY * PZ;                // plane pitch: 1152 slots = 36 * 32
constexpr int RING = 8;                         // planes resident
constexpr int BOXSLOTS = RING * PLANE;          // 9216 slots = 73728 B
constexpr int NCLS = 32;                        // classes = 8-byte bank pairs
constexpr int NXB = CAPX - 2;                   // first-tap planes a box can have (K = 2: 30, K = 3: 29)
constexpr int NWMAX = 6;                        // windows per tile: ceil(29 / 5), ceil(30 / 6)
constexpr int GCAP = NS / 64 + NWMAX;           // wave rows (64 sorted samples) of a tile: sum_w ceil(n_w / 64) <= 70
constexpr int NR = (GCAP + 7) / 8;              // record slots per thread
constexpr int HTCAP = (BOXSLOTS * 8 - GCAP * 64 * 16) / 2;   // entries of the hole table behind the records (1024)
constexpr int SLOWCAP = 512;
constexpr int HANDBACK = NS / 8;                // out-of-box samples beyond which the generic kernel takes a (smooth) tile, defer.hip
static_assert(PLANE % NCLS == 0, "the plane pitch must keep the class plane-independent");
static_assert(GCAP * 64 * 16 + HTCAP * 2 <= BOXSLOTS * 8, "records + hole table alias the ring");
static_assert(NR * 8 >= GCAP, "record slots");
// register arrays indexed by the (wave-uniform) slot number: vectors, so that a dynamic index becomes v_movrel instead of scratch
typedef float vNRf __attribute__((ext_vector_type(NR)));
typedef int vNRi __attribute__((ext_vector_type(NR)));

template <int K> struct Win {
    static constexpr int SPAN = RING - K;       // first-tap planes per window
    static_assert((CAPX - K + SPAN - 1) / SPAN <= NWMAX, "windows per tile");
};
// window of first-tap plane x0 (x0 / SPAN without a division; exact for x0 < 40)
template <int K> __device__ __forceinline__ int window_of(int x0) { return K == 3 ? (x0 * 13) >> 6 : (x0 * 11) >> 6; }

struct Smem {
    int   taboff[3][40];       // wrapped lattice offset (elements) of box plane / row / slice
    float tabsgn[3][40];       // boundary sign of the same
    int   lo[3], hi[3];        // block reductions of the first-tap indices
    int   nslow, pad[1];
    int   oobc[VPT][NT / 64];  // out-of-box samples per (sample slot, wave): count, then exclusive prefix -- their rank in the slow list
    int   hist[NXB][NCLS];     // samples per (first-tap plane, class); then: samples of the same window and class on earlier planes
    int   oob;                 // spare counter of the samples outside the box
    int   wrows[NWMAX + 1];    // half-wave rows of window w (even)
    int   wsur[NWMAX + 1];     // surplus samples of window w (they fill holes)
    int   wbase[NWMAX + 1];    // first half-wave row of window w
    unsigned char  cnteff[NWMAX][NCLS];   // occupied rows of lane q in window w
    unsigned short soffg[NWMAX][NCLS];    // hole-table index of the first surplus sample of (w, q)
    unsigned short slow[SLOWCAP];
    float2 box[BOXSLOTS];      // the ring; aliased: float4 rec[GCAP * 64] + unsigned short holes[HTCAP]; float2 out[NS]
};
static_assert(sizeof(Smem) <= 81920, "two workgroups per CU");

// The LDS reads of one x-plane of a stencil, both channels per read, as ONE block of ds_read_b64 at
// immediate offsets (the compiler would merge neighbours into ds_read2_b64 -- 3x the cost per byte on
// gfx950 -- and interleave samples until the results spill).  The block waits for its own reads.
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
#undef IP_RD
// two rows of a cubic stencil (the x-plane in two blocks: half the registers of the 16-read block; the quads of the next
// window are in flight under the taps and need theirs)
#define IP_RD(o, off) "ds_read_b64 %" #o ", %8 offset:" #off "\n\t"
__device__ __forceinline__ void stencil_reads(unsigned addr, f2 (&v)[8])
{
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16) IP_RD(3, 24)
                 IP_RD(4, 288) IP_RD(5, 296) IP_RD(6, 304) IP_RD(7, 312)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(addr) : "memory");
}
#undef IP_RD
#define IP_RD(o, off) "ds_read_b64 %" #o ", %9 offset:" #off "\n\t"
__device__ __forceinline__ void stencil_reads(unsigned addr, f2 (&v)[9])
{
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16)
                 IP_RD(3, 288) IP_RD(4, 296) IP_RD(5, 304)
                 IP_RD(6, 576) IP_RD(7, 584) IP_RD(8, 592)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
                 : "v"(addr) : "memory");
}
#undef IP_RD

// the K + 1 weights of the x-stencil (the scalar form of weights_yz, sorted_util.hpp; splines.py:30-80)
template <int K>
__device__ __forceinline__ void weights_x(float t, float *w)
{
    if (K == 3) {
        const float u = t - 1.f, v = 2.f - t;
        const float u2 = u * u, v2 = v * v;
        w[0] = (v2 * v) * (1.f / 6.f);
        w[3] = (u2 * u) * (1.f / 6.f);
        w[1] = u2 * (u * 0.5f - 1.f) + 2.f / 3.f;
        w[2] = v2 * (v * 0.5f - 1.f) + 2.f / 3.f;
    } else {
        const float a = 1.5f - t, c = t - 0.5f, m = t - 1.f;
        w[0] = (a * a) * 0.5f;
        w[1] = 0.75f - m * m;
        w[2] = (c * c) * 0.5f;
        w[3] = 0.f;
    }
}

// All taps of one sorted sample from the ring: returns the two channels' sums.
template <int K>
__device__ __forceinline__ f2 gather_sample(unsigned boxaddr, float tx, f2 tyz, int key)
{
    const int x0 = key & 31;
    const unsigned base = boxaddr + (((unsigned)key >> 5) & 2047u) * 8u;
    float wx[4];
    weights_x<K>(tx, wx);
    f2 w[4];
    weights_yz<K>(tyz, w);
    f2 a = { 0.f, 0.f };
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        const unsigned addr = base + (unsigned)((x0 + i) & (RING - 1)) * (unsigned)(PLANE * 8);
        f2 pp = { 0.f, 0.f };
        if (K == 3) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f2 t2[8];
                stencil_reads(addr + (unsigned)(h * 2 * PZ * 8), t2);
#pragma unroll
                for (int jy = 0; jy < 2; ++jy) {
                    f2 q = { 0.f, 0.f };
#pragma unroll
                    for (int k = 0; k <= K; ++k) q = f2{ w[k].y, w[k].y } * t2[4 * jy + k] + q;
                    pp = f2{ w[2 * h + jy].x, w[2 * h + jy].x } * q + pp;
                }
            }
        } else {
            f2 t2[(K + 1) * (K + 1)];
            stencil_reads(addr, t2);
#pragma unroll
            for (int jy = 0; jy <= K; ++jy) {
                f2 q = { 0.f, 0.f };
#pragma unroll
                for (int k = 0; k <= K; ++k) q = f2{ w[k].y, w[k].y } * t2[(K + 1) * jy + k] + q;
                pp = f2{ w[jy].x, w[jy].x } * q + pp;
            }
        }
        a = f2{ wx[i], wx[i] } * pp + a;
    }
    return a;
}

// ---------------------------------------------------------------------------
// Staging of box planes [p0, p1) into their ring slots: slot ((x & 7) * 32 + y) * PZ + z = (sign * c0, sign * c1)
// of the wrapped lattice point (bounds.py:30-89 through the tables).
// ---------------------------------------------------------------------------
// Rows of the box that are contiguous runs of the lattice's unit-stride dim with sign +1 move as QUADS of 4 slots (two
// 16-byte loads, one per channel; two 16-byte LDS stores).  The move is split: `quads_issue` computes the addresses and
// issues the loads -- before the tap loop of the window that is being processed --, `quads_commit` writes the ring once
// that window's readers are done: the round trip to memory runs under the taps.
template <int NU> struct Quads { float4 a0[NU], a1[NU]; };

// Thread -> (row of the sweep, quad), fixed for the tile: 56 rows of 9 quads per sweep of the workgroup; rows are (plane, y)
// pairs, 32 to the plane.  No division per quad, and the loads take a uniform base + a 32-bit byte offset.
constexpr int QPR = PZ / 4, RPS = NT / QPR;
struct StageMap { int r0, zs, boff; bool qon; };
__device__ __forceinline__ StageMap stage_map(int tid, int S2, int loz, int esize)
{
    StageMap m;
    m.r0 = tid / QPR;
    const int qd = tid - m.r0 * QPR, nq = (S2 + 3) >> 2;
    m.qon = qd < nq && m.r0 < RPS;
    m.zs = 4 * qd + 4 <= S2 ? 4 * qd : S2 - 4;                       // the last quad is shifted to END at S_z
    if (!m.qon) m.zs = 0;
    m.boff = (loz + m.zs) * esize;
    return m;
}

template <typename T>
__device__ __forceinline__ float4 ld4_at(const T *base, unsigned byte_off)
{
    return ld4<T>(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off));
}

// sweeps u0 .. u0 + NU - 1 over the rows of planes [p0, p0 + nrow / 32): loads issued, results left in Q
template <typename T, int NU>
__device__ __forceinline__ void quads_issue(Smem &sm, const T *__restrict__ vc0, const T *__restrict__ vc1, const StageMap &m, int p0, int u0,
                                            int S0, int S1, Quads<NU> &Q)
{
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int r = m.r0 + RPS * (u0 + u);
        // (rows that do not exist read a row that does: no select per load; the commit drops them)
        int x = p0 + (r >> 5), y = r & 31;
        x = x < S0 - 1 ? x : S0 - 1; y = y < S1 - 1 ? y : S1 - 1;
        const unsigned off = (unsigned)((sm.taboff[0][x] + sm.taboff[1][y]) * (int)sizeof(T) + m.boff);
        Q.a0[u] = ld4_at<T>(vc0, off);
        Q.a1[u] = ld4_at<T>(vc1, off);
    }
}

template <int NU>
__device__ __forceinline__ void quads_commit(Smem &sm, const StageMap &m, int p0, int u0, int nrow, int S1, bool plus, Quads<NU> &Q)
{
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int r = m.r0 + RPS * (u0 + u);
        if (!(m.qon && r < nrow && (r & 31) < S1)) continue;
        float2 *dst = sm.box + ((r + CAPY * p0) & (RING * CAPY - 1)) * PZ + m.zs;       // ring row ((x & 7) * 32 + y)
        float4 a0 = Q.a0[u], a1 = Q.a1[u];
        if (!plus) {
            const float sg = sm.tabsgn[0][p0 + (r >> 5)] * sm.tabsgn[1][r & 31];
            a0.x *= sg; a0.y *= sg; a0.z *= sg; a0.w *= sg;
            a1.x *= sg; a1.y *= sg; a1.z *= sg; a1.w *= sg;
        }
        if (!(m.zs & 1)) {
            reinterpret_cast<float4 *>(dst)[0] = make_float4(a0.x, a1.x, a0.y, a1.y);
            reinterpret_cast<float4 *>(dst)[1] = make_float4(a0.z, a1.z, a0.w, a1.w);
        } else {                                                     // shifted last quad of an odd extent: 8-byte stores
            dst[0] = make_float2(a0.x, a1.x); dst[1] = make_float2(a0.y, a1.y);
            dst[2] = make_float2(a0.z, a1.z); dst[3] = make_float2(a0.w, a1.w);
        }
    }
}

template <typename T>
__device__ __forceinline__ void stage_slots(Smem &sm, const T *__restrict__ vc0, const T *__restrict__ vc1, int tid, int p0, int p1, int S1, int S2)
{
    // general case (the box wraps in z, or z is strided): slot by slot through the z table
    constexpr int U = 8;
    const int nrow = (p1 - p0) * CAPY;
    const int z = tid & 31;
    const bool zin = z < S2;
    const int oz = zin ? sm.taboff[2][z] : 0;
    const float sgz = zin ? sm.tabsgn[2][z] : 0.f;
    for (int r0 = tid >> 5; r0 < nrow; r0 += (NT / 32) * U) {
        float v0[U], v1[U], sg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * (NT / 32);
            const bool on = zin && r < nrow && (r & 31) < S1;
            const int xr = on ? p0 + (r >> 5) : 0, yr = on ? r & 31 : 0;
            const int off = on ? sm.taboff[0][xr] + sm.taboff[1][yr] + oz : 0;
            sg[u] = on ? sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sgz : 0.f;
            v0[u] = Cvt<float, T>::ld(vc0[off]);
            v1[u] = Cvt<float, T>::ld(vc1[off]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * (NT / 32);
            if (zin && r < nrow && (r & 31) < S1)
                sm.box[(((p0 + (r >> 5)) & (RING - 1)) * CAPY + (r & 31)) * PZ + z] = make_float2(v0[u] * sg[u], v1[u] * sg[u]);
        }
    }
    for (int e = tid; S2 > 32 && e < nrow * 4; e += NT) {            // slices 32 ... 35
        const int r = e >> 2, z2 = 32 + (e & 3);
        if ((r & 31) < S1 && z2 < S2) {
            const int xr = p0 + (r >> 5), yr = r & 31;
            const int off = sm.taboff[0][xr] + sm.taboff[1][yr] + sm.taboff[2][z2];
            const float sgn = sm.tabsgn[0][xr] * sm.tabsgn[1][yr] * sm.tabsgn[2][z2];
            sm.box[((xr & (RING - 1)) * CAPY + yr) * PZ + z2] = make_float2(Cvt<float, T>::ld(vc0[off]) * sgn, Cvt<float, T>::ld(vc1[off]) * sgn);
        }
    }
}

// ---------------------------------------------------------------------------
// COMPACT tiles.  Under a smooth deformation the box of a tile is small (the identity: 19^3 lattice points): laid out with
// a row pitch of 24 slots it fits the 72 KiB of LDS WHOLE, and the natural order of the samples -- lanes along z, the
// two rows of a half wave two rows of the tile apart: 24 slots = 48 banks, twice that = 32 banks mod 64 -- is already
// free of bank conflicts.  Such a tile needs no sort, no record exchange, no passes: one staging round, then every thread
// gathers its own 8 samples (weights once, 64 reads at immediate offsets, 84 packed FMAs each).  Chosen per tile.
// ---------------------------------------------------------------------------
constexpr int CZ = 24;                          // row pitch of the compact layout (8-byte slots)
constexpr int CQ = CZ / 4, CRPS = NT / CQ;      // quads per row, rows per staging sweep of the workgroup
#define IP_RD(o, off) "ds_read_b64 %" #o ", %16 offset:" #off "\n\t"
__device__ __forceinline__ void compact_reads(unsigned addr, f2 (&v)[16])
{
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16) IP_RD(3, 24)
                 IP_RD(4, 192) IP_RD(5, 200) IP_RD(6, 208) IP_RD(7, 216)
                 IP_RD(8, 384) IP_RD(9, 392) IP_RD(10, 400) IP_RD(11, 408)
                 IP_RD(12, 576) IP_RD(13, 584) IP_RD(14, 592) IP_RD(15, 600)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
                 : "v"(addr) : "memory");
}
#undef IP_RD
#define IP_RD(o, off) "ds_read_b64 %" #o ", %9 offset:" #off "\n\t"
__device__ __forceinline__ void compact_reads(unsigned addr, f2 (&v)[9])
{
    asm volatile(IP_RD(0, 0) IP_RD(1, 8) IP_RD(2, 16)
                 IP_RD(3, 192) IP_RD(4, 200) IP_RD(5, 208)
                 IP_RD(6, 384) IP_RD(7, 392) IP_RD(8, 400)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
                 : "v"(addr) : "memory");
}
#undef IP_RD
static_assert(CZ * 8 == 192, "the immediate offsets above are (row * CZ + k) * 8");

// all taps of one sample from the compact box; pp8: plane pitch in bytes
template <int K>
__device__ __forceinline__ f2 compact_sample(unsigned addr, unsigned pp8, float tx, f2 tyz)
{
    float wx[4];
    weights_x<K>(tx, wx);
    f2 w[4];
    weights_yz<K>(tyz, w);
    f2 a = { 0.f, 0.f };
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        f2 t2[(K + 1) * (K + 1)];
        compact_reads(addr + (unsigned)i * pp8, t2);
        f2 pp = { 0.f, 0.f };
#pragma unroll
        for (int jy = 0; jy <= K; ++jy) {
            f2 q = { 0.f, 0.f };
#pragma unroll
            for (int k = 0; k <= K; ++k) q = f2{ w[k].y, w[k].y } * t2[(K + 1) * jy + k] + q;
            pp = f2{ w[jy].x, w[jy].x } * q + pp;
        }
        a = f2{ wx[i], wx[i] } * pp + a;
    }
    return a;
}

// Staging of the whole box in the compact layout: slot (x * S1 + y) * CZ + z = (sign * c0, sign * c1) of the wrapped
// lattice point (bounds.py:30-89 through the tables); the rows are contiguous runs of the lattice's unit-stride dim (zlin).
// Thread = (row of the sweep, quad): 85 rows of 6 quads per sweep, three sweeps per round trip.
template <typename T>
__device__ __forceinline__ void compact_stage(Smem &sm, const T *__restrict__ vc0, const T *__restrict__ vc1, int tid, int S0, int S1, int S2, int loz, bool plus)
{
    const int r0 = tid / CQ, qd = tid - r0 * CQ;
    const int nq = (S2 + 3) >> 2, nrow = S0 * S1;
    const bool qon = qd < nq && r0 < CRPS;
    const int zs = qon ? (4 * qd + 4 <= S2 ? 4 * qd : S2 - 4) : 0;   // the last quad is shifted to END at S_z
    const unsigned boff = (unsigned)((loz + zs) * (int)sizeof(T));
    const float rS1 = 1.f / (float)S1;
    for (int u0 = 0; u0 * CRPS < nrow; u0 += 3) {
        float4 a0[3], a1[3]; float sg[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            int r = r0 + CRPS * (u0 + u);
            r = r < nrow - 1 ? r : nrow - 1;                         // (rows that do not exist read the last one; dropped below)
            const int x = (int)(((float)r + 0.5f) * rS1), y = r - x * S1;
            const unsigned off = (unsigned)((sm.taboff[0][x] + sm.taboff[1][y]) * (int)sizeof(T)) + boff;
            sg[u] = plus ? 1.f : sm.tabsgn[0][x] * sm.tabsgn[1][y];
            a0[u] = ld4<T>(reinterpret_cast<const T *>(reinterpret_cast<const char *>(vc0) + off));
            a1[u] = ld4<T>(reinterpret_cast<const T *>(reinterpret_cast<const char *>(vc1) + off));
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = r0 + CRPS * (u0 + u);
            if (!(qon && r < nrow)) continue;
            float2 *dst = sm.box + r * CZ + zs;
            if (!plus) {
                a0[u].x *= sg[u]; a0[u].y *= sg[u]; a0[u].z *= sg[u]; a0[u].w *= sg[u];
                a1[u].x *= sg[u]; a1[u].y *= sg[u]; a1[u].z *= sg[u]; a1[u].w *= sg[u];
            }
            if (!(zs & 1)) {
                reinterpret_cast<float4 *>(dst)[0] = make_float4(a0[u].x, a1[u].x, a0[u].y, a1[u].y);
                reinterpret_cast<float4 *>(dst)[1] = make_float4(a0[u].z, a1[u].z, a0[u].w, a1[u].w);
            } else {                                                 // shifted last quad of an odd extent: 8-byte stores
                dst[0] = make_float2(a0[u].x, a1[u].x); dst[1] = make_float2(a0[u].y, a1[u].y);
                dst[2] = make_float2(a0[u].z, a1[u].z); dst[3] = make_float2(a0[u].w, a1[u].w);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// pull: val[b,c,o] = mask * sum_taps w vol        (nd.py:80-143)
// ---------------------------------------------------------------------------
// LEAN: every tile is whole and no sample is masked (extrapolate = 1) -- the launcher knows --: the validity and mask
// bookkeeping folds away, and a tile whose box needed no clamping classifies its samples without the in-box tests.
template <typename T, int K, int GM, bool LEAN>
__global__ __launch_bounds__(NT, 4) void pull_window(KParams p, const T *__restrict__ vol, const float *__restrict__ grid,
                                                     T *__restrict__ val, int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch,
                                                     DeferArgs defer)
{
    constexpr int SPAN = Win<K>::SPAN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    Lattice L;
#pragma unroll
    for (int d = 0; d < 3; ++d) { L.bound[d] = p.bound[d]; L.n[d] = p.vol_n[d]; L.ss[d] = p.vol_ss[d] / (int)sizeof(T); L.k[d] = K; }
    L.lin = 0;
    // One workgroup per tile (no persistent loop: nothing for the optimiser to hoist across tiles and hold -- spilled --
    // through all phases; the dispatcher balances tiles of unequal cost).  Workgroups are dealt round-robin to the 8 XCDs,
    // each with its own L2: XCD i takes the i-th contiguous eighth of the tile sequence (z-fastest order), so that the
    // tiles in flight on one XCD are neighbours and share their halos in that XCD's L2 (tile_common.hpp: WorkRange).
    {
        int tid = (int)threadIdx.x;
        int work;
        {
            const int total = ntiles * nbatch, bid = (int)blockIdx.x;
            if (total >= 8) {
                const int per = (total + 7) >> 3, xcd = bid & 7;
                work = xcd * per + (bid >> 3);
                const int end = (xcd + 1) * per < total ? (xcd + 1) * per : total;
                if (work >= end) return;
            } else {
                work = bid;
                if (work >= total) return;
            }
        }
        // (the divisions run on the VALU: bring the results back to scalar registers, or everything derived from the tile's
        //  position lives in vector registers)
        const int64_t b = __builtin_amdgcn_readfirstlane(work / ntiles);
        const int tile = work % ntiles;
        TileGeom g = tile_geom(tile, gx, gy, gz, nty, ntz);
        g.ox0 = __builtin_amdgcn_readfirstlane(g.ox0); g.oy0 = __builtin_amdgcn_readfirstlane(g.oy0); g.oz0 = __builtin_amdgcn_readfirstlane(g.oz0);
        prof_mark(-1);
        // the thread's samples: natural ids nid + 512 v, nid = the thread index with bits 4 and 5 swapped -- the two rows of a half
        // wave lie two rows of the tile apart (what the compact tiles want; the sorted ones do not care)
        const int nid = (tid & ~0x30) | ((tid & 0x10) << 1) | ((tid & 0x20) >> 1);
        const bool full = LEAN || (g.ox0 + TS <= g.gx && g.oy0 + TS <= g.gy && g.oz0 + TS <= g.gz);     // block-uniform

        // ---- coordinates of the thread's 8 samples (natural order: sample tid + 512 v) ----------------------
        float c[VPT][3];
        if (GM == 0 && full) {
            // one address per thread; its samples lie two x-planes apart
            const float *gp = grid + b * p.grid_sb + (((int64_t)(g.ox0 + (nid >> 8)) * g.gy + (g.oy0 + ((nid >> 4) & 15))) * g.gz + (g.oz0 + (nid & 15))) * 3;
            const int64_t step = (int64_t)g.gy * g.gz * 6;
#pragma unroll
            for (int v = 0; v < VPT; ++v) { c[v][0] = gp[v * step]; c[v][1] = gp[v * step + 1]; c[v][2] = gp[v * step + 2]; }
        } else {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int ox, oy, oz;
                sample_pos(g, nid + NT * v, ox, oy, oz);
                // unconditional loads from a clamped position: the compiler batches them (one exposed round trip)
                ox = ox < g.gx ? ox : g.gx - 1; oy = oy < g.gy ? oy : g.gy - 1; oz = oz < g.gz ? oz : g.gz - 1;
                load_xyz<GM>(p, grid, b, g, ox, oy, oz, c[v]);
            }
        }
        if (tid < 3) { sm.lo[tid] = 0x7fffffff; sm.hi[tid] = -0x7fffffff; }
        for (int e = tid; e < NXB * NCLS + 1; e += NT) (&sm.hist[0][0])[e] = 0;                 // (+ the spare counter behind it)
        unsigned validmask = (1u << VPT) - 1u, inbmask = (1u << VPT) - 1u;
        if (!full) {
            validmask = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                int ox, oy, oz;
                sample_pos(g, nid + NT * v, ox, oy, oz);
                if (ox < g.gx && oy < g.gy && oz < g.gz) validmask |= 1u << v;
            }
        }
        __syncthreads();
        if (!LEAN && p.extrapolate != 1) {                           // nd.py:10-27
            inbmask = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (c[v][0] > (float)p.mask_lo && c[v][0] < (float)p.mask_hi[0] && c[v][1] > (float)p.mask_lo && c[v][1] < (float)p.mask_hi[1]
                    && c[v][2] > (float)p.mask_lo && c[v][2] < (float)p.mask_hi[2])
                    inbmask |= 1u << v;
        }
        // ---- first-tap index (kept as a float: exact, saturates nowhere) and stencil coordinate:
        // i0 = floor(x - (K-1)/2), t = x - i0  (nd.py:45-46); block min / max of i0
        float fl[VPT][3];
        {
            float fmn[3] = { 3e38f, 3e38f, 3e38f }, fmx[3] = { -3e38f, -3e38f, -3e38f };
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    fl[v][d] = floorf(c[v][d] - 0.5f * (float)(K - 1));
                    c[v][d] -= fl[v][d];                             // c becomes t
                    const bool ok = (validmask >> v) & 1;            // (folds away for full tiles)
                    fmn[d] = __builtin_fminf(fmn[d], ok ? fl[v][d] : fmn[d]);
                    fmx[d] = __builtin_fmaxf(fmx[d], ok ? fl[v][d] : fmx[d]);
                }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float lim = 1073741824.f;
                const int a = wave_min(__float2int_rz(__builtin_fmaxf(__builtin_fminf(fmn[d], lim), -lim)));
                const int e = wave_max(__float2int_rz(__builtin_fmaxf(__builtin_fminf(fmx[d], lim), -lim)));
                if ((tid & 63) == 0) { atomicMin(&sm.lo[d], a); atomicMax(&sm.hi[d], e); }
            }
        }
        __syncthreads();
        prof_mark(4);
        int lo[3], S[3];
        bool whole = LEAN;                                           // the box holds the stencils of all the tile's (finite) samples
        bool unclamped = true;
        {
            const int cap[3] = { CAPX, CAPY, CAPZ };
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int l = sm.lo[d], h = sm.hi[d] + K;           // supports span [l, h]
                if (h < l) { l = 0; h = 0; }                   // tile without valid samples
                int sz_ = h - l + 1;
                if (sz_ > cap[d]) { l += (sz_ - cap[d]) / 2; sz_ = cap[d]; whole = false; unclamped = false; }   // keep the centre; the rest goes to the slow list
                lo[d] = l; S[d] = sz_;
            }
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
        // ---- COMPACT tile (block-uniform): whole tile, every stencil in the unclamped box, the box fits LDS with a row pitch of CZ
        // slots, contiguous rows: no sort, no windows -- staged once, every thread gathers its own samples
        {
            const bool zl = L.ss[2] == 1 && S[2] >= 4 && lo[2] >= (L.bound[2] == B_DST1 ? 1 : 0) && lo[2] + S[2] <= L.n[2];
            const bool compact = unclamped && validmask == (1u << VPT) - 1u && zl && S[2] <= CZ && S[0] * S[1] * CZ <= BOXSLOTS && (p.dbg & 8192);
            if (compact) {
                const bool pl = L.bound[0] != B_ZERO && L.bound[0] != B_DST1 && L.bound[0] != B_DST2
                             && L.bound[1] != B_ZERO && L.bound[1] != B_DST1 && L.bound[1] != B_DST2;
                unsigned cadr[VPT];
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    int x0 = __float2int_rz(fl[v][0]) - lo[0], y0 = __float2int_rz(fl[v][1]) - lo[1], z0 = __float2int_rz(fl[v][2]) - lo[2];
                    x0 = max(0, min(x0, S[0] - K - 1)); y0 = max(0, min(y0, S[1] - K - 1)); z0 = max(0, min(z0, S[2] - K - 1));
                    cadr[v] = (unsigned)(((x0 * S[1] + y0) * CZ + z0) * 8);
                }
                const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
                const unsigned pp8 = (unsigned)(S[1] * CZ * 8);
                for (int cch = 0; cch < p.C; cch += 2) {
                    const bool two = cch + 1 < p.C;
                    const T *vc0 = vol + b * p.vol_sb + cch * p.vol_sc;
                    const T *vc1 = two ? vc0 + p.vol_sc : vc0;
                    T *oc0 = val + b * p.val_sb + cch * p.val_sc;
                    T *oc1 = oc0 + p.val_sc;
                    __syncthreads();                                 // tables written / the previous pair's stores have read the box
                    compact_stage<T>(sm, vc0, vc1, tid, S[0], S[1], S[2], lo[2], pl);
                    __syncthreads();
                    f2 res[VPT];
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        float tx = c[v][0]; f2 tyz = f2{ c[v][1], c[v][2] };
                        asm volatile("" : "+v"(tx), "+v"(tyz));
                        res[v] = compact_sample<K>(boxaddr + opaque((int)cadr[v]), pp8, tx, tyz);
                        asm volatile("" : "+v"(res[v]));
                    }
                    __syncthreads();
                    float2 *outb = sm.box;
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        const float m = (float)((inbmask >> v) & 1);
                        outb[nid + NT * v] = make_float2(res[v].x * m, res[v].y * m);
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < NS / 4 / NT; ++u) {
                        const int qi = tid + NT * u;                 // quad: x = qi >> 6, y = (qi >> 2) & 15, z = 4 (qi & 3)
                        const float4 *src = reinterpret_cast<const float4 *>(outb + 4 * qi);
                        const float4 lo_ = src[0], hi_ = src[1];
                        const int64_t o = ((int64_t)(g.ox0 + (qi >> 6)) * g.gy + (g.oy0 + ((qi >> 2) & 15))) * g.gz + (g.oz0 + 4 * (qi & 3));
                        st4<T>(oc0 + o, make_float4(lo_.x, lo_.z, hi_.x, hi_.z));
                        if (two) st4<T>(oc1 + o, make_float4(lo_.y, lo_.w, hi_.y, hi_.w));
                    }
                }
                return;
            }
        }
        // ---- classification + histogram.  In the box <=> lo <= i0 <= lo + S - K - 1 in every dim.  Every sample does
        // ONE returning LDS add, unconditionally (the ranks of the 8 samples come back together): on the counter of its
        // (first-tap plane, class), or on the spare counter when it is not in the box.
        unsigned fastmask = 0, selfmask = 0;
        int kq[VPT], rk[VPT];
        if (whole) {
            // every sample is in the box: no tests.  (A coordinate that is not finite yields any counter -- clamped -- and a
            // NaN result.)
            fastmask = (1u << VPT) - 1u;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int x0 = __float2int_rz(fl[v][0]) - lo[0], y0 = __float2int_rz(fl[v][1]) - lo[1], z0 = __float2int_rz(fl[v][2]) - lo[2];
                const int yz = y0 * PZ + z0;                         // slot inside a plane; yz mod 32 = class
                unsigned bin = (unsigned)((x0 << 5) | (yz & (NCLS - 1)));
                bin = bin < (unsigned)(NXB * NCLS - 1) ? bin : (unsigned)(NXB * NCLS - 1);
                kq[v] = (int)(bin >> 5) | ((yz & 2047) << 5) | ((nid + NT * v) << 16) | (3 << 28);
                rk[v] = atomicAdd(&(&sm.hist[0][0])[bin], 1);
            }
            if (tid == 0) sm.nslow = 0;
        } else {
            const float flo[3] = { (float)lo[0], (float)lo[1], (float)lo[2] };
            const float fhi[3] = { (float)(lo[0] + S[0] - K - 1), (float)(lo[1] + S[1] - K - 1), (float)(lo[2] + S[2] - K - 1) };
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const bool in = (fl[v][0] >= flo[0]) & (fl[v][0] <= fhi[0]) & (fl[v][1] >= flo[1]) & (fl[v][1] <= fhi[1])
                              & (fl[v][2] >= flo[2]) & (fl[v][2] <= fhi[2]) & (bool)((validmask >> v) & 1);
                const int x0 = __float2int_rz(fl[v][0]) - lo[0], y0 = __float2int_rz(fl[v][1]) - lo[1], z0 = __float2int_rz(fl[v][2]) - lo[2];
                const int yz = y0 * PZ + z0;                         // slot inside a plane; yz mod 32 = class
                kq[v] = (x0 & 31) | ((yz & 2047) << 5) | ((nid + NT * v) << 16) | (int)(((inbmask >> v) & 1) << 28) | (1 << 29);
                if (in) fastmask |= 1u << v;
                rk[v] = atomicAdd(in ? &sm.hist[x0][yz & (NCLS - 1)] : &sm.oob, 1);
            }
        }
        // (rare) out-of-box samples: the first SLOWCAP of them -- in the order (sample slot, wave, lane), NOT in the order of
        // arrival -- go to the slow list, the rest is left to its thread
        const unsigned oob = validmask & ~fastmask;
        if (!whole) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const unsigned long long bal = __ballot((oob >> v) & 1);
                if ((tid & 63) == 0) sm.oobc[v][tid >> 6] = __popcll(bal);
            }
        }
        __syncthreads();
        prof_mark(5);
        // ---- the planner, step A: half wave w owns window w, lane q class q.  The histogram column of (w, q) becomes the
        // exclusive prefix over the window's planes; the window's rows, holes and surplus follow inside the half wave.
        const int nx = S[0] - K;                                     // first-tap planes of the box
        const int nw = (nx + SPAN - 1) / SPAN;                       // windows (<= NWMAX)
        int pc = 0, prows = 0, pholes = 0, phoff = 0, psoff = 0, pnsur = 0;      // planner registers of thread (w, q)
        {
            const int w = tid >> 5, q = tid & 31;
            if (w < nw) {
                int run = 0;
#pragma unroll
                for (int i = 0; i < SPAN; ++i) {
                    const int x = w * SPAN + i;
                    if (x < nx) { const int h = sm.hist[x][q]; sm.hist[x][q] = run; run += h; }
                }
                pc = run;
            }
            int n = pc;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) n += __shfl_xor(n, o, 32);
            prows = 2 * ((n + 63) >> 6);
            pholes = pc < prows ? prows - pc : 0;
            const int surplus = pc > prows ? pc - prows : 0;
            int nholes;
            phoff = half_excl_scan(pholes, nholes);
            psoff = half_excl_scan(surplus, pnsur);
            if (w <= NWMAX && q == 0) { sm.wrows[w] = prows; sm.wsur[w] = pnsur; }           // (zero beyond the last window)
        }
        if (!whole && tid >= NT - 64) {
            const int l = tid - (NT - 64);
            const int cc = sm.oobc[l >> 3][l & 7];                   // (VPT x NT / 64 = 64 counters, slot-major)
            int incl = cc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (l >= o) incl += t; }
            sm.oobc[l >> 3][l & 7] = incl - cc;
            if (l == 63) sm.nslow = incl;
        }
        __syncthreads();
        // ---- step B: first row and first hole-table entry of the window; occupied rows per lane; the hole table
        unsigned short *holetab = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(sm.box) + GCAP * 64 * 16);
        {
            const int w = tid >> 5, q = tid & 31;
            if (w < nw) {
                int rbase = 0, hbase = 0;
#pragma unroll
                for (int u = 0; u < NWMAX - 1; ++u) { rbase += u < w ? sm.wrows[u] : 0; hbase += u < w ? sm.wsur[u] : 0; }
                const int room = HTCAP - hbase < 0 ? 0 : HTCAP - hbase;
                const int placed = pnsur < room ? pnsur : room;      // surplus samples the table can place
                const int filled = placed - phoff < 0 ? 0 : (placed - phoff > pholes ? pholes : placed - phoff);
                sm.cnteff[w][q] = (unsigned char)((pc < prows ? pc : prows) + filled);
                sm.soffg[w][q] = (unsigned short)(hbase + psoff);
                if (q == 0) { sm.wbase[w] = rbase; if (w == nw - 1) sm.wbase[NWMAX] = rbase + prows; }       // ([NWMAX]: rows of the tile)
                // hole m of class q: row pc + m of the window, lane q -- listed while surplus samples remain
                for (int m = 0; m < filled; ++m) holetab[hbase + phoff + m] = (unsigned short)((rbase + pc + m) * 32 + q);
            } else if (w < NWMAX && q == 0) {
                sm.wbase[w] = 0x7fffffff;                            // no such window: it starts behind every row
            }
        }
        // slow list: rank = prefix of the (slot, wave) counters + position among the wave's lanes
#pragma unroll
        for (int v = 0; v < VPT && !whole; ++v) {
            const unsigned long long bal = __ballot((oob >> v) & 1);
            if ((oob >> v) & 1) {
                const int rank = sm.oobc[v][tid >> 6] + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                if (rank < SLOWCAP) sm.slow[rank] = (unsigned short)(nid + NT * v);
                else selfmask |= 1u << v;
            }
        }
        __syncthreads();
        prof_mark(6);
        // ---- records to their sorted places: row (wbase + r) of lane q, r = rank inside (window, class)
        float4 *rec = reinterpret_cast<float4 *>(sm.box);
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!((fastmask >> v) & 1)) continue;
            const int key = opaque(kq[v]);                           // (opaque: pieces of the classification's addresses would be kept -- spilled -- across the planner)
            const int x0 = key & 31, q = (key >> 5) & 31;
            const int w = window_of<K>(x0);
            const int r = sm.hist[x0][q] + rk[v];
            const int rows = sm.wrows[w];
            int pos = (sm.wbase[w] + r) * 32 + q;
            if (r >= rows) {                                         // surplus sample: into a hole
                const int o = (int)sm.soffg[w][q] + r - rows;
                pos = o < HTCAP ? (int)holetab[o] : -1;
                if (pos < 0) { fastmask &= ~(1u << v); selfmask |= 1u << v; continue; }   // pathological tile
            }
            rec[pos] = make_float4(c[v][0], c[v][1], c[v][2], __int_as_float(key));
        }
        __syncthreads();
        prof_mark(7);
        // ---- the thread's records: slot j = wave row j * 8 + wave, rows 2 g and 2 g + 1 (one per half wave)
        vNRf rtx, rty, rtz; vNRi rkey;
        vNRi swin;                                                   // window of the slot (wave-uniform), -1: none
        {
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            int wb[NWMAX + 1];
#pragma unroll
            for (int u = 0; u < NWMAX; ++u) wb[u] = __builtin_amdgcn_readfirstlane(sm.wbase[u]);
            const int total = __builtin_amdgcn_readfirstlane(sm.wbase[NWMAX]);
            int rb_[NR]; unsigned char ce[NR]; float4 rr[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int grow = 2 * (j * 8 + wave);                 // first of the wave's two rows
                int w = -1, rb = 0;                                  // the last window that starts at or before the row (an empty window shares its start with the next one)
#pragma unroll
                for (int u = 0; u < NWMAX; ++u) if (grow >= wb[u]) { w = u; rb = wb[u]; }
                if (grow >= total) w = -1;
                swin[j] = w;
                rb_[j] = rb;
                // (unconditional loads, all in flight together; a slot without a window reads window 0's row count and drops it)
                ce[j] = sm.cnteff[w < 0 ? 0 : w][tid & 31];
                rr[j] = rec[(j * 8 + wave) * 64 + (tid & 63)];
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int row = 2 * (j * 8 + wave) + ((tid >> 5) & 1) - rb_[j];
                const bool on = swin[j] >= 0 && row < (int)ce[j];
                // an empty place reads the box corner (a harmless address) and is dropped at the end
                const float mid = 0.5f * (float)(K - 1) + 0.25f;
                rtx[j] = on ? rr[j].x : mid; rty[j] = on ? rr[j].y : mid; rtz[j] = on ? rr[j].z : mid; rkey[j] = on ? __float_as_int(rr[j].w) : 0;
            }
        }
        const int nslow = sm.nslow < SLOWCAP ? sm.nslow : SLOWCAP;
        bool skip = false;
        if (defer.flag) {                                                 // (block-uniform) too rough for the box: the generic kernel takes the tile, defer.hip
            bool hand_back = sm.nslow > (HANDBACK << ((p.dbg >> 9) & 7));
            if (hand_back) hand_back = tiled::tile_smooth(p, grid, b, 3, g.ox0, g.oy0, g.oz0, TS, TS, TS, g.gx, g.gy, g.gz, sm.hi);
            if (hand_back && tid == 0) defer_mark(defer, work, tile_desc(b, g.ox0 / TS, g.oy0 / TS, g.oz0 / TS));
            skip = hand_back && defer.desc;
        }
        if (skip) return;
        // rows of the box are contiguous runs of the lattice's unit-stride dim, sign +1 throughout
        // (dst1 has sign 0 at index 0 -- quirk B-3 -- so its run must start at 1)
        const bool zlin = L.ss[2] == 1 && S[2] >= 4 && lo[2] >= (L.bound[2] == B_DST1 ? 1 : 0) && lo[2] + S[2] <= L.n[2];
        // the boundary conditions of x and y never change the sign (replicate, dct1, dct2, dft: bounds.py:30-89)
        const bool plus = L.bound[0] != B_ZERO && L.bound[0] != B_DST1 && L.bound[0] != B_DST2
                       && L.bound[1] != B_ZERO && L.bound[1] != B_DST1 && L.bound[1] != B_DST2;
        prof_mark(0);

        // One channel pair, or several: with at most two channels the records die as their samples are gathered and the
        // results take their registers (two code paths: the general one holds both across the window loop)
        auto pairs = [&](auto single_tag) {
        constexpr bool SINGLE = decltype(single_tag)::value;
        for (int cch = 0; ; cch += 2) {
            const bool two = cch + 1 < p.C;
            const T *vc0 = vol + b * p.vol_sb + cch * p.vol_sc;
            const T *vc1 = two ? vc0 + p.vol_sc : vc0;
            T *oc0 = val + b * p.val_sb + cch * p.val_sc;
            T *oc1 = oc0 + p.val_sc;
            vNRf acc0 = 0.f, acc1 = 0.f;
            // The windows in turn.  `Q` holds the quads of the NEXT window's new planes, loaded while this window's taps run.
            const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
            const bool pre = zlin && !(p.dbg & (1 | 4));             // (debug bit 4: loads and stores together, nothing in flight under the taps)
            int staged = 0;                                          // planes [.., staged) have been staged
            int w = 0;
            while (w < nw && __builtin_amdgcn_readfirstlane(sm.wrows[w]) == 0) ++w;      // first window that holds samples (block-uniform)
            Quads<3> Q;
            int qp0 = 0, qrows = 0;                                  // the quads in flight: planes [qp0, qp0 + qrows / 32)
            bool inflight = false;                                   // Q holds the quads of this window's new planes
            while (w < nw) {
                tid = opaque((int)threadIdx.x);
                const StageMap sm_ = stage_map(tid, S[2], lo[2], (int)sizeof(T));
                const int pa = w * SPAN, pb = pa + RING < S[0] ? pa + RING : S[0];
                const int p0 = pa > staged ? pa : staged;
                staged = pb;
                __syncthreads();                                     // the previous window's readers are done (first window: the records are in registers)
                if (p.dbg & 1) {
                } else if (inflight) {
                    quads_commit<3>(sm, sm_, qp0, 0, qrows, S[1], plus, Q);
                } else if (zlin) {                                   // (the first window stages all its planes, 8 x 32 rows: two round trips)
                    const int nrow = (pb - p0) * CAPY;
                    for (int u0 = 0; u0 * RPS < nrow; u0 += 3) {
                        quads_issue<T, 3>(sm, vc0, vc1, sm_, p0, u0, S[0], S[1], Q);
                        quads_commit<3>(sm, sm_, p0, u0, nrow, S[1], plus, Q);
                    }
                } else {
                    stage_slots<T>(sm, vc0, vc1, tid, p0, pb, S[1], S[2]);
                }
                inflight = false;
                __syncthreads();
                prof_mark(1);
                int wn = w + 1;
                while (wn < nw && __builtin_amdgcn_readfirstlane(sm.wrows[wn]) == 0) ++wn;
                if (wn < nw && pre) {                                // the next window's new planes: SPAN x 32 rows = 3 sweeps of 56
                    const int na = wn * SPAN, nb = na + RING < S[0] ? na + RING : S[0];
                    qp0 = na > staged ? na : staged;
                    qrows = (nb - qp0) * CAPY;
                    inflight = qrows <= 3 * RPS;                     // (more after a run of empty windows, or with SPAN = 6: those are staged in place)
                    if (inflight) quads_issue<T, 3>(sm, vc0, vc1, sm_, qp0, 0, S[0], S[1], Q);
                }
                if (!(p.dbg & 2)) {
                    // (a ROLLED loop over the thread's slots: the records and the results are register arrays indexed by the
                    //  wave-uniform slot number -- v_movrel --; unrolled, the nine copies of the body defeat the register allocator)
#pragma clang loop unroll(disable)
                    for (int j = 0; j < NR; ++j) {
                        if (__builtin_amdgcn_readfirstlane(swin[j]) != w) continue;      // (wave-uniform)
                        float tx = rtx[j]; f2 tyz = f2{ rty[j], rtz[j] };
                        asm volatile("" : "+v"(tx), "+v"(tyz));
                        const f2 r = gather_sample<K>(boxaddr, tx, tyz, opaque(rkey[j]));
                        if (SINGLE) { rtx[j] = r.x; rty[j] = r.y; }   // (the record is spent: the result takes its registers)
                        else { acc0[j] = r.x; acc1[j] = r.y; }
                    }
                }
                w = wn;
                prof_mark(2);
            }
            {
            // back to the natural order through LDS
            __syncthreads();
            tid = opaque((int)threadIdx.x);
            float2 *outb = sm.box;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int key = opaque(rkey[j]);
                if (!((key >> 29) & 1)) continue;
                const float m = (float)((key >> 28) & 1);            // nd.py:139-140
                outb[(key >> 16) & (NS - 1)] = SINGLE ? make_float2(rtx[j] * m, rty[j] * m) : make_float2(acc0[j] * m, acc1[j] * m);
            }
            // out-of-box samples: one wave per sample, lanes = taps, straight from global memory
            if (nslow > 0) {
                const int wave = tid >> 6, lane = tid & 63;
                for (int sidx = wave; sidx < nslow; sidx += NT / 64) {
                    float a0, a1, m;
                    slow_taps<T, K, GM>(p, L, grid, b, g, sm.slow[sidx], lane, vc0, vc1, a0, a1, m);
                    a0 = wave_sum(a0); a1 = wave_sum(a1);
                    if (lane == 0) outb[sm.slow[sidx]] = make_float2(a0 * m, a1 * m);
                }
            }
            // pathological tiles (slow list or hole table overflowed): the thread gathers its sample itself
            if (selfmask) {
                for (int v = 0; v < VPT; ++v) {
                    if (!((selfmask >> v) & 1)) continue;
                    int ox, oy, oz; float x[3];
                    sample_pos(g, nid + NT * v, ox, oy, oz);
                    load_xyz<GM>(p, grid, b, g, ox, oy, oz, x);
                    int ii[3]; float tt[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) split(K, x[d], ii[d], tt[d]);
                    const float m = inb_mask(p, x);
                    outb[nid + NT * v] = make_float2(m * tiled::gather_one_thread<T>(L, vc0, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1),
                                                     m * tiled::gather_one_thread<T>(L, vc1, ii[0], ii[1], ii[2], tt[0], tt[1], tt[2], -1));
                }
            }
            __syncthreads();
            if (full) {
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
            if (SINGLE || cch + 2 >= p.C) break;
        }
        };
        if (p.C <= 2) pairs(std::true_type{}); else pairs(std::false_type{});
    }
}

// ---------------------------------------------------------------------------
// Launcher
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
        const long long total = (long long)ntiles() * B;
        return dim3((unsigned)(total >= 8 ? 8 * ((total + 7) >> 3) : total), 1u);
    }
};

template <typename T, int K, int GM, bool LEAN>
static int launch_pull_(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    const int attr = big_lds<pull_window<T, K, GM, LEAN>>(sizeof(Smem));
    if (attr) return attr;
    const TileCount t(p);
    const Defer df(k, st, t.ntiles(), p->batch, t.ntx, t.nty, t.ntz, TS, TS, TS);
    hipLaunchKernelGGL((pull_window<T, K, GM, LEAN>), t.grid((int)p->batch), dim3(NT), sizeof(Smem), st,
                       k, (const T *)vol, (const float *)grid, (T *)val, t.gx, t.gy, t.gz, t.nty, t.ntz, t.ntiles(), (int)p->batch, df.args);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const int rc = df.desc ? DeferOps<T>::pull(k, vol, grid, val, df.tl, st) : 0;
    return rc ? rc : 1;
}

template <typename T, int K, int GM>
static int launch_pull(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{
    // dense grids whose tiles are all whole, nothing masked: the lean instantiation
    if (GM == 0 && k.extrapolate == 1 && p->grid_shape[0] % TS == 0 && p->grid_shape[1] % TS == 0 && p->grid_shape[2] % TS == 0 && !(k.dbg & 8))
        return launch_pull_<T, K, GM, GM == 0>(p, k, vol, grid, val, st);
    return launch_pull_<T, K, GM, false>(p, k, vol, grid, val, st);
}

} // namespace window

#define IP_SYM2(a, b) a##b
#define IP_SYM(a, b) IP_SYM2(a, b)

// Called by try_sorted_pull_* (ops_sorted.hip) once the problem is known to be eligible (3-D, one order K = 2..3);
// returns 1 when it took the problem, 0 to decline, anything else: error.
int IP_SYM(try_window_pull_, IP_TSFX)(const interpol_problem *p, const KParams &k, int K, const void *vol, const void *grid, void *val, hipStream_t st)
{
    using T = IP_TT;
    if (k.sep) {
        if constexpr (std::is_same<T, float>::value) {
            if (K == 3) return k.sep == 1 ? window::launch_pull<T, 3, 1>(p, k, vol, grid, val, st)
                             : (k.sep == 2 ? window::launch_pull<T, 3, 2>(p, k, vol, grid, val, st) : window::launch_pull<T, 3, 3>(p, k, vol, grid, val, st));
            return k.sep == 1 ? window::launch_pull<T, 2, 1>(p, k, vol, grid, val, st)
                 : (k.sep == 2 ? window::launch_pull<T, 2, 2>(p, k, vol, grid, val, st) : window::launch_pull<T, 2, 3>(p, k, vol, grid, val, st));
        } else {
            return 0;
        }
    }
    if (K == 3) return window::launch_pull<T, 3, 0>(p, k, vol, grid, val, st);
    return window::launch_pull<T, 2, 0>(p, k, vol, grid, val, st);
}

} // namespace ip

#ifdef IP_PROF
#define IP_PROF_NAME3(s) interpol_debug_prof_window_##s
#define IP_PROF_NAME2(s) IP_PROF_NAME3(s)
extern "C" __attribute__((visibility("default"))) int IP_PROF_NAME2(IP_TSFX)(unsigned long long *out, int reset)
{
    unsigned long long z[16] = { 0 };
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ip::sorted::g_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ip::sorted::g_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
