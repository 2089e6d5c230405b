// ===========================================================================
// pull_direct.hip -- grid_pull (reference interpol/nd.py:80-143) for SMOOTH deformations, 3-D quadratic / cubic, float32:
// the single-pass small-box tiles.
//
// The class-sorted tiles of ops_sorted.hip pay for rough deformations everywhere: a counting sort of the tile's samples by LDS
// bank class, four passes over a 32 x 32 x 36 box with the weights evaluated again in every pass (437 VALU instructions per
// sample, of which 84 are taps).  Where the deformation is smooth -- registration fields, affine maps, the identity -- none of
// that is needed: the stencils of HALF a tile (8 x 16 x 16 samples) then fit a box of 16 x 24 x 24 lattice points (72 KiB: two
// workgroups per CU), which is staged ONCE per channel pair; the samples stay in their natural order, a wave = four rows of 16
// z-neighbours, paired (y, y + 2 | y + 1, y + 3) so that on the identity the 32 lanes of a half wave read 32 different bank
// pairs (row pitch 24 slots: two rows apart = 48 = 16 mod 32); weights once, 64 reads at immediate offsets, 84 packed FMAs,
// coalesced stores.
//
// A workgroup walks a contiguous run of (batch item, tile) work items and writes one flag per item: 0 = served here, 2 = left to
// pull_sorted (launched behind this kernel with KParams::gate_n < 0: it serves the flagged items only and overwrites the flag
// with 0, or 1 = left to the bricks of the image, push_owner.hip).  A tile is left when either half does not fit the box (or
// holds non-finite coordinates); after two such tiles in a row the workgroup leaves the rest of its run without loading it --
// under i.i.d. noise the kernel costs two coordinate loads per workgroup (3 % of the grid), no probe launch is needed.
// MEASURED (tools/ab_direct.py, 4 x 2 x 256^3 cubic): identity 1.16 ms against 1.05 for the class-sorted tiles alone, smooth
// field 1.18 / 1.10, sigma = 0.5 1.41 / 1.12 (tiles that load and then leave), sigma = 2 1.36 / 1.30.  The phases of a half tile
// are latency-bound with two workgroups per CU: no staging and no taps still costs 0.51 ms, staging adds 0.45, taps 0.34.
// NOT the default: opt-in with INTERPOL_FLAG_SMALL_TILES (kept as a measured negative result, parity-tested).
// Needs the workspace of interpol_pull_ws (the flags); every boundary condition (a box slot is a lattice point through the
// tables), the three extrapolation modes, all four coordinate sources.
// ===========================================================================
#include "sorted_util.hpp"

namespace ip {
namespace direct {

using namespace sorted;

constexpr int NT = 512;                         // threads: two workgroups per CU
constexpr int HX = 8;                           // samples of a half tile along x (y, z: TS = 16)
constexpr int VPT = HX * TS * TS / NT;          // 4 samples per thread
constexpr int CX = 16, CY = 24, CZ = 24;        // box capacity, lattice points per dim
constexpr int PZ = CZ, PLANE = CY * PZ;         // pitches (8-byte slots)
constexpr int PAD = 128;                        // (quadratic stencils read a fourth row / plane they do not use)
constexpr int GIVEUP = 2;                       // tiles left in a row after which the workgroup leaves the rest of its run

struct Smem {
    int   taboff[2][3][CY];                     // per half
    float tabsgn[2][3][CY];
    int   wlo[NT / 64][2][4], whi[NT / 64][2][4];   // per wave and half: bounds of the first taps (x, y, z), [3]: non-finite coordinates
    float2 box[CX * PLANE + PAD];
};
static_assert(sizeof(Smem) <= 80 * 1024, "two workgroups per CU");

#define IP_RD(o, off) "ds_read_b64 %" #o ", %16 offset:" #off "\n\t"
// the 16 taps of one x-plane of a stencil: rows 192 bytes apart
__device__ __forceinline__ void plane_reads(unsigned addr, f2 (&v)[16])
{
    static_assert(PZ * 8 == 192, "the immediate offsets are (row * PZ + k) * 8");
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

template <int K, int GM>
__global__ __launch_bounds__(NT, 4) void pull_direct(KParams p, const float *__restrict__ vol, const float *__restrict__ grid, float *__restrict__ val,
                                                     int gx, int gy, int gz, int nty, int ntz, int ntiles, int nbatch,
                                                     int *__restrict__ flags, int nzero)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw);
    // the header, brick counters and brick list of the bricks' workspace lie in front of the flags: cleared on the way (own_bin,
    // launched behind pull_sorted, counts in them)
    for (int i = (int)blockIdx.x * NT + (int)threadIdx.x; i < nzero; i += (int)gridDim.x * NT) (flags - nzero)[i] = 0;
    const int total = ntiles * nbatch;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per, end = first + per < total ? first + per : total;
    int given_up = 0;
    for (int work = first; work < end; ++work) {
        const int tid = opaque((int)threadIdx.x);
        if (given_up >= GIVEUP) {                                     // (block-uniform)
            for (int w_ = work + tid; w_ < end; w_ += NT) flags[w_] = 2;
            break;
        }
        const int64_t b = work / ntiles;
        const TileGeom g = tile_geom(work % ntiles, gx, gy, gz, nty, ntz);
        // thread -> sample of the half tile: z = lane & 15, rows (y, y + 2 | y + 1, y + 3) of the wave's block of four, x = wave / 4 + 2 v
        const int wave = tid >> 6, lane = tid & 63;
        const int rr = lane >> 4;
        const int sy = (wave & 3) * 4 + (((rr & 1) << 1) | (rr >> 1)), sz = lane & 15;
        const int oy = g.oy0 + sy, oz = g.oz0 + sz;
        // ---- the coordinates of BOTH halves at once (one exposed round trip per tile), first taps i0 = floor(x - (K-1)/2) (nd.py:45)
        // and their bounding boxes per half: wave reductions, one LDS slot per wave, ONE barrier
        float c[2][VPT][3];
        unsigned valid = 0;
        const int xw = g.ox0 + (wave >> 2);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int ox = xw + h * HX + 2 * v;
                if (ox < gx && oy < gy && oz < gz) valid |= 1u << (h * VPT + v);
                // unconditional loads from a clamped position: the compiler batches them
                load_xyz<GM>(p, grid, b, g, ox < gx ? ox : gx - 1, oy < gy ? oy : gy - 1, oz < gz ? oz : gz - 1, c[h][v]);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int mn[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff }, mx[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff };
            int bad = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!((valid >> (h * VPT + v)) & 1)) continue;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float fl = floorf(c[h][v][d] - 0.5f * (float)(K - 1));
                    const bool ok = fl >= -1073741824.f && fl <= 1073741824.f;        // (NaN, infinities: pull_sorted's business)
                    bad |= ok ? 0 : 1;
                    const int i = ok ? __float2int_rz(fl) : 0;
                    mn[d] = i < mn[d] ? i : mn[d]; mx[d] = i > mx[d] ? i : mx[d];
                }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int a = wave_min(mn[d]), e = wave_max(mx[d]);
                if (lane == 0) { sm.wlo[wave][h][d] = a; sm.whi[wave][h][d] = e; }
            }
            const int anybad = wave_max(bad);
            if (lane == 0) sm.wlo[wave][h][3] = anybad;
        }
        __syncthreads();                                             // (also: the previous tile's readers of the box and tables are done)
        int lo2[2][3], S2[2][3];
        bool fits = true, any[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int hi[3] = { -0x7fffffff, -0x7fffffff, -0x7fffffff }, bad = 0;
            lo2[h][0] = lo2[h][1] = lo2[h][2] = 0x7fffffff;
#pragma unroll
            for (int w_ = 0; w_ < NT / 64; ++w_) {
#pragma unroll
                for (int d = 0; d < 3; ++d) { const int a = sm.wlo[w_][h][d], e = sm.whi[w_][h][d]; lo2[h][d] = a < lo2[h][d] ? a : lo2[h][d]; hi[d] = e > hi[d] ? e : hi[d]; }
                bad |= sm.wlo[w_][h][3];
            }
            any[h] = hi[0] >= lo2[h][0];                             // a sample of the grid in this half (block-uniform)
#pragma unroll
            for (int d = 0; d < 3; ++d) S2[h][d] = hi[d] - lo2[h][d] + K + 1;
            if (any[h] && (bad || S2[h][0] > CX || S2[h][1] > CY || S2[h][2] > CZ)) fits = false;
            if (!any[h]) { lo2[h][0] = lo2[h][1] = lo2[h][2] = 0; S2[h][0] = S2[h][1] = S2[h][2] = 0; }
        }
        const bool served = fits;                                    // (block-uniform)
        if (served) {
            // ---- box slot -> wrapped lattice offset and sign (bounds.py:30-89), both halves
            if (tid < 2 * 3 * 64) {
                const int h = tid / 192, d = (tid >> 6) % 3, slot = tid & 63;
                if (slot < CY) {
                    const int l = h == 0 ? (d == 0 ? lo2[0][0] : d == 1 ? lo2[0][1] : lo2[0][2]) : (d == 0 ? lo2[1][0] : d == 1 ? lo2[1][1] : lo2[1][2]);
                    const long long pk = wrap_outofline(p.bound[d], l + slot, p.vol_n[d]);
                    sm.taboff[h][d][slot] = (int)(pk & 0xffffffffll) * (p.vol_ss[d] / 4);
                    sm.tabsgn[h][d][slot] = (float)(int)(pk >> 32);
                }
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (!served || !any[half]) continue;                     // (block-uniform)
            const int lo[3] = { lo2[half][0], lo2[half][1], lo2[half][2] };
            const int S[3] = { S2[half][0], S2[half][1], S2[half][2] };
            const int xb = xw + half * HX;
            // rows of the box that are contiguous runs of the image's unit-stride dim with sign +1 move as quads (16-byte loads)
            const bool zlin = p.vol_ss[2] == 4 && S[2] >= 4 && lo[2] >= (p.bound[2] == B_DST1 ? 1 : 0) && lo[2] + S[2] <= p.vol_n[2];
            for (int ch = 0; ch < p.C; ch += 2) {
                const bool two = ch + 1 < p.C;
                const float *vc0 = vol + b * p.vol_sb + (int64_t)ch * p.vol_sc;
                const float *vc1 = two ? vc0 + p.vol_sc : vc0;
                float *oc0 = val + b * p.val_sb + (int64_t)ch * p.val_sc;
                __syncthreads();                                     // tables written / the previous pair's readers are done
                if (p.dbg & 1) {                                     // (ablation: no staging)
                } else if (zlin) {
                    constexpr int QPR = CZ / 4, NU = (CX * CY * QPR + NT - 1) / NT;      // 6 quads per row, 5 per thread
                    const int nq = (S[2] + 3) >> 2;                  // the last one is shifted to END at S_z
                    float4 a0[NU], a1[NU]; float sg[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int e = tid + NT * u, row = e / QPR, q = e - row * QPR;
                        const int x = row / CY, y = row - x * CY;
                        const bool on = q < nq && x < S[0] && y < S[1];
                        const int zs = 4 * q + 4 <= S[2] ? 4 * q : S[2] - 4;
                        const int off = on ? sm.taboff[half][0][x] + sm.taboff[half][1][y] + lo[2] + zs : 0;
                        sg[u] = on ? sm.tabsgn[half][0][x] * sm.tabsgn[half][1][y] : 0.f;
                        a0[u] = ld4<float>(vc0 + off);
                        a1[u] = ld4<float>(vc1 + off);
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int e = tid + NT * u, row = e / QPR, q = e - row * QPR;
                        const int x = row / CY, y = row - x * CY;
                        if (q < nq && x < S[0] && y < S[1]) {
                            const int zs = 4 * q + 4 <= S[2] ? 4 * q : S[2] - 4;
                            float2 *dst = sm.box + row * PZ + zs;
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
                    for (int e = tid; e < CX * CY * CZ; e += NT) {
                        const int x = e / (CY * CZ), y = (e / CZ) % CY, z = e % CZ;
                        if (x < S[0] && y < S[1] && z < S[2]) {
                            const int off = sm.taboff[half][0][x] + sm.taboff[half][1][y] + sm.taboff[half][2][z];
                            const float sgn = sm.tabsgn[half][0][x] * sm.tabsgn[half][1][y] * sm.tabsgn[half][2][z];
                            sm.box[(x * CY + y) * PZ + z] = make_float2(vc0[off] * sgn, vc1[off] * sgn);
                        }
                    }
                }
                __syncthreads();
                const unsigned boxaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(sm.box);
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    float x0 = c[half][v][0], x1 = c[half][v][1], x2 = c[half][v][2];
                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));  // (one sample at a time: nothing of the next one is hoisted above this one's reads)
                    const float fx = floorf(x0 - 0.5f * (float)(K - 1)), fy = floorf(x1 - 0.5f * (float)(K - 1)), fz = floorf(x2 - 0.5f * (float)(K - 1));
                    const float tx = x0 - fx; const f2 tyz = f2{ x1 - fy, x2 - fz };
                    int cx = __float2int_rz(fx) - lo[0], cy = __float2int_rz(fy) - lo[1], cz = __float2int_rz(fz) - lo[2];
                    // (inside the box by construction; clamped, should a sample outside the grid have been loaded from elsewhere)
                    cx = max(0, min(cx, CX - K - 1)); cy = max(0, min(cy, CY - K - 1)); cz = max(0, min(cz, CZ - K - 1));
                    const unsigned addr = boxaddr + (unsigned)((cx * CY + cy) * PZ + cz) * 8u;
                    float wx[4];
                    {   // the K + 1 weights of the x-stencil (scalar form of weights_yz; splines.py:30-80)
                        if (K == 3) { const float u = tx - 1.f, w_ = 2.f - tx, u2 = u * u, w2 = w_ * w_;
                                      wx[0] = (w2 * w_) * (1.f / 6.f); wx[3] = (u2 * u) * (1.f / 6.f); wx[1] = u2 * (u * 0.5f - 1.f) + 2.f / 3.f; wx[2] = w2 * (w_ * 0.5f - 1.f) + 2.f / 3.f; }
                        else { const float a = 1.5f - tx, cc = tx - 0.5f, m = tx - 1.f; wx[0] = (a * a) * 0.5f; wx[1] = 0.75f - m * m; wx[2] = (cc * cc) * 0.5f; wx[3] = 0.f; }
                    }
                    f2 w[4];
                    weights_yz<K>(tyz, w);
                    f2 a = { 0.f, 0.f };
#pragma unroll
                    for (int ii = 0; ii <= K; ++ii) {
                        if (p.dbg & 2) { a = f2{ w[0].x + wx[ii], w[1].y }; break; }      // (ablation: no taps)
                        f2 t2[16];
                        plane_reads(addr + (unsigned)(ii * PLANE * 8), t2);
                        f2 pp = { 0.f, 0.f };
#pragma unroll
                        for (int jy = 0; jy <= K; ++jy) {
                            f2 q = { 0.f, 0.f };
#pragma unroll
                            for (int k = 0; k <= K; ++k) q = f2{ w[k].y, w[k].y } * t2[4 * jy + k] + q;
                            pp = f2{ w[jy].x, w[jy].x } * q + pp;
                        }
                        a = f2{ wx[ii], wx[ii] } * pp + a;
                        asm volatile("" : "+v"(a));                  // (one x-plane at a time: the 16 reads of several planes in flight spill)
                    }
                    const float xyz[3] = { x0, x1, x2 };
                    const float m = inb_mask(p, xyz);                // nd.py:139-140
                    if ((valid >> (half * VPT + v)) & 1) {
                        const int64_t o = ((int64_t)(xb + 2 * v) * gy + oy) * gz + oz;
                        oc0[o] = a.x * m;
                        if (two) oc0[p.val_sc + o] = a.y * m;
                    }
                }
            }
        }
        if (tid == 0) flags[work] = served ? 0 : 2;
        given_up = served ? 0 : given_up + 1;
        if (!served) __syncthreads();                                // (the waves' bounds are read; a served tile passed a barrier since)
    }
}

} // namespace direct

// Launch the small-box tiles in front of pull_sorted (interpol_pull_ws, abi.hip).  `flags`: one int per (batch item, tile) in
// pull_sorted's order, `nzero` ints in front of them are cleared.  Returns 1 when launched, 0 to decline, else an error.
int try_pull_direct(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, int *flags, int nzero, hipStream_t st)
{
    using namespace direct;
    if (p->dim != 3 || p->dtype != INTERPOL_F32 || p->grid_dtype != INTERPOL_F32) return 0;
    if (k.order[0] != k.order[1] || k.order[0] != k.order[2] || k.order[0] < 2 || k.order[0] > 3) return 0;
    const int gx = (int)p->grid_shape[0], gy = (int)p->grid_shape[1], gz = (int)p->grid_shape[2];
    const int ntx = (gx + TS - 1) / TS, nty = (gy + TS - 1) / TS, ntz = (gz + TS - 1) / TS;
    const long long total = (long long)ntx * nty * ntz * p->batch;
    if (total <= 0 || total > 0x7fffffffll) return 0;
    // runs of about 32 work items per workgroup (the give-up costs two of them), at least two workgroups per CU
    long long nwg = (total + 31) / 32;
    const long long want = 2ll * cu_count();
    if (nwg < want) nwg = total < want ? total : want;
#define IP_PD(KK, GM)                                                                                                   \
    {                                                                                                                   \
        const int attr = big_lds<direct::pull_direct<KK, GM>>(sizeof(Smem));                                            \
        if (attr) return attr;                                                                                          \
        hipLaunchKernelGGL((direct::pull_direct<KK, GM>), dim3((unsigned)nwg), dim3(NT), sizeof(Smem), st, k, (const float *)vol, \
                           (const float *)grid, (float *)val, gx, gy, gz, nty, ntz, ntx * nty * ntz, (int)p->batch, flags, nzero); \
    }
#define IP_PD_GM(KK) { if (k.sep == 0) IP_PD(KK, 0) else if (k.sep == 1) IP_PD(KK, 1) else if (k.sep == 2) IP_PD(KK, 2) else IP_PD(KK, 3) }
    if (k.order[0] == 3) IP_PD_GM(3) else IP_PD_GM(2)
#undef IP_PD_GM
#undef IP_PD
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : (int)e;
}

} // namespace ip
