// Label-map sampling: arg-max over labels of the interpolated indicator images, in ONE pass.
//
// The reference (interpol/api.py:194-205) loops over `input.unique()`: for every label it builds
// the indicator image, pulls it (GridPull) and keeps `out[soft > pmax] = label` -- L full
// passes plus three elementwise passes each.  For a given output voxel only the labels under its
// (K+1)^D stencil can have a non-zero indicator value, so
//     soft_l = mask * sum_{taps t with label_t == l} w_t sign_t
// is evaluated for the labels of the stencil only, and the winner is the reference's: the largest
// soft value if it is > 0 (pmax starts at 0), the SMALLEST such label on ties (unique() is
// sorted ascending and the update is a strict `>`), else 0.  Covered: isotropic orders 0..3 in
// 1 / 2 / 3 dims (up to the 64 taps of the 3-D cubic: pull_labels_hash_kernel, float32 coordinates; pull_labels_wide_kernel, float64); the host keeps the
// reference's loop for anything else and for prefilter = True.
// No FMA contraction in this translation unit: the winner of an arg-max can hinge on the last
// bit of a weight (coordinates exactly half-way between voxels give exact ties in the reference),
// so the weights are evaluated with the reference's roundings (separate multiply and add).
#pragma clang fp contract(off)
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include "launch.hpp"
#include <hip/hip_runtime.h>

namespace ip {
namespace {

template <typename G, typename R, int D, int K>
__global__ __launch_bounds__(256) void pull_labels_kernel(KParams p, const int *__restrict__ vol, const G *__restrict__ grid,
                                                          int *__restrict__ val, int B)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.N) return;
    constexpr int T0 = K + 1, T1 = D > 1 ? K + 1 : 1, T2 = D > 2 ? K + 1 : 1, NT = T0 * T1 * T2;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, K, true, NEED_W> s;
        s.setup(p, x);
        for (int c = 0; c < p.C; ++c) {
            const char *v0 = reinterpret_cast<const char *>(vol + b * p.vol_sb + c * p.vol_sc);
            int lab[NT];
            R w[NT];
            // Cubic weights with the reference's own operations (splines.py:42-44 divides by 6;
            // the shared bspline_w multiplies by 1/6: one ulp apart, enough to flip an exact tie)
            R wd[3][K + 1];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int j = 0; j <= K; ++j) wd[d][j] = s.w[d][j];
            if constexpr (K == 3) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const R fl = sizeof(R) == 4 ? (R)floorf((float)(x[d] - R(1))) : (R)floor((double)(x[d] - R(1)));   // nd.py:45
                    const R t = x[d] - fl;                                                                    // nd.py:46
#pragma unroll
                    for (int j = 0; j <= K; ++j) {
                        R a = t - R(j); a = a < R(0) ? -a : a;
                        const R u = R(2) - a;
                        const R we = a < R(1) ? (a * a * (a - R(2)) * R(3) + R(4)) / R(6) : (u * u * u) / R(6);
                        wd[d][j] = s.w[d][j] == R(0) ? R(0) : (s.w[d][j] < R(0) ? -we : we);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < T0; ++i)
#pragma unroll
                for (int j = 0; j < T1; ++j)
#pragma unroll
                    for (int k = 0; k < T2; ++k) {
                        const int t = (i * T1 + j) * T2 + k;
                        lab[t] = *reinterpret_cast<const int *>(v0 + (s.off[0][i] + s.off[1][j] + s.off[2][k]));
                        w[t] = (wd[0][i] * wd[1][j]) * wd[2][k];             // node-major products, sign folded in
                    }
            int best_l = 0x7fffffff;
            R best_w = R(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int l = lab[t];
                R sum = R(0);
#pragma unroll
                for (int u = 0; u < NT; ++u) sum += (lab[u] == l) ? w[u] : R(0);   // tap order, as the per-label pull sums
                sum *= s.mask;
                if (sum > best_w || (sum == best_w && l < best_l)) { best_w = sum; best_l = l; }
            }
            val[b * p.val_sb + c * p.val_sc + o] = best_w > R(0) ? best_l : 0;
        }
    }
}

// The same arg-max for stencils of up to 64 taps (3-D cubic).  Comparing every tap's label with every
// other's is 4096 compare-adds per voxel there; instead the DISTINCT labels under the stencil are
// visited one by one (usually one to four of them): take the first tap not yet accounted for, sum the
// weights of all taps that carry its label -- in tap order, adding exact zeros for the others, i.e. the
// very sum the per-label pull forms -- and strike those taps off.  Weights are re-formed on the fly as
// (wx wy) wz, the node-major product of the reference.
constexpr int WIDE_NT = 128;                 // 32 KiB of staged labels per workgroup: five workgroups per CU
template <typename G, typename R, int D, int K>
__global__ __launch_bounds__(WIDE_NT, 3) void pull_labels_wide_kernel(KParams p, const int *__restrict__ vol, const G *__restrict__ grid,
                                                               int *__restrict__ val, int B)
{
    // the labels under a thread's stencil, staged once: column threadIdx.x of labs[tap][thread] (conflict-free: a wave reads a row)
    __shared__ int labs[64][WIDE_NT];
    const int64_t o = (int64_t)blockIdx.x * WIDE_NT + threadIdx.x;
    if (o >= p.N) return;
    constexpr int T0 = K + 1, T1 = D > 1 ? K + 1 : 1, T2 = D > 2 ? K + 1 : 1, NT = T0 * T1 * T2;
    static_assert(NT <= 64, "one bit per tap");
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, K, true, NEED_W> s;
        s.setup(p, x);
        R wd[3][K + 1];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j <= K; ++j) wd[d][j] = s.w[d][j];
        if constexpr (K == 3) {                                      // the reference's own operations (see above)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const R fl = sizeof(R) == 4 ? (R)floorf((float)(x[d] - R(1))) : (R)floor((double)(x[d] - R(1)));
                const R t = x[d] - fl;
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    R a = t - R(j); a = a < R(0) ? -a : a;
                    const R u = R(2) - a;
                    const R we = a < R(1) ? (a * a * (a - R(2)) * R(3) + R(4)) / R(6) : (u * u * u) / R(6);
                    wd[d][j] = s.w[d][j] == R(0) ? R(0) : (s.w[d][j] < R(0) ? -we : we);
                }
            }
        }
        R wxy[T0 * T1];
#pragma unroll
        for (int i = 0; i < T0; ++i)
#pragma unroll
            for (int j = 0; j < T1; ++j) wxy[i * T1 + j] = wd[0][i] * wd[1][j];
        for (int c = 0; c < p.C; ++c) {
            const char *v0 = reinterpret_cast<const char *>(vol + b * p.vol_sb + c * p.vol_sc);
            // The NT labels under the stencil are gathered ONCE, into LDS (held in registers they took the kernel to 256 VGPRs: one
            // wave per SIMD and nothing to hide the gathers behind, 6.4 ms for 192^3 samples); then one pass over them per DISTINCT
            // label -- one or two for most stencils of a label map, up to NT for noise.
#pragma unroll
            for (int i = 0; i < T0; ++i) {
                // one x-plane (T1 * T2 gathers) in flight at a time; its addresses are formed here, not kept across planes
                unsigned oi = s.off[0][i];
                asm volatile("" : "+v"(oi));
                int tmp[T1 * T2];
#pragma unroll
                for (int j = 0; j < T1; ++j)
#pragma unroll
                    for (int k = 0; k < T2; ++k) tmp[j * T2 + k] = *reinterpret_cast<const int *>(v0 + (oi + s.off[1][j] + s.off[2][k]));
#pragma unroll
                for (int jk = 0; jk < T1 * T2; ++jk) labs[i * T1 * T2 + jk][threadIdx.x] = tmp[jk];
                asm volatile("" ::: "memory");
            }
            unsigned long long todo = NT == 64 ? ~0ull : ((1ull << (NT & 63)) - 1ull);
            int best_l = 0x7fffffff;
            R best_w = R(0);
            // (volatile: the reads are loop-invariant to the compiler, which would hoist all NT of them into registers again)
            const volatile int *col = &labs[0][threadIdx.x];
            while (todo) {
                const int t = __ffsll((long long)todo) - 1;          // first tap not accounted for: its label
                const int l = col[t * WIDE_NT];
                R sum = R(0);
                unsigned lo_hit = 0u, hi_hit = 0u;
                R wz[T2];                                            // (laundered per pass: the NT products wxy * wz are invariant of this loop
#pragma unroll                                                       //  and would be hoisted into NT more registers)
                for (int k = 0; k < T2; ++k) { wz[k] = wd[2][k]; asm volatile("" : "+v"(wz[k])); }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const bool m = col[u * WIDE_NT] == l;
                    sum += m ? wxy[u / T2] * wz[u % T2] : R(0);      // tap order, as the per-label pull sums
                    if (u < 32) lo_hit |= m ? (1u << u) : 0u; else hi_hit |= m ? (1u << (u - 32)) : 0u;
                    if ((u & 7) == 7) asm volatile("" : "+v"(sum), "+v"(lo_hit), "+v"(hi_hit) :: "memory");   // (eight reads in flight, not NT)
                }
                todo &= ~(((unsigned long long)hi_hit << 32) | lo_hit);
                sum *= s.mask;
                if (sum > best_w || (sum == best_w && l < best_l)) { best_w = sum; best_l = l; }
            }
            val[b * p.val_sb + c * p.val_sc + o] = best_w > R(0) ? best_l : 0;
        }
    }
}

// The 3-D cubic arg-max for label maps with MANY labels under a stencil (i.i.d. labels: ~36 distinct ones among the 64 taps, where the
// pass-per-distinct-label walk above costs 36 x 64 compare-adds per voxel: 10 ms for 192^3).  One pass over the taps, in tap order, into
// a per-thread hash table of 64 (label, sum) slots in LDS -- slot s of thread t at [s][t]: a wave touches 64 different banks whatever the
// slots -- with linear probing and the occupancy in a 64-bit register (no sentinel: any int32 is a label).  A label's sum receives its
// taps' weights in tap order, starting from 0 + w: the very additions of the per-label pull (the zeros it adds for the other taps change
// nothing).  The label of the previous tap and its running sum stay in registers.  32 KiB per 64 threads: five waves per CU -- enough here, the
// gathers of a plane are in flight together.  192^3 voxels (tools/r5/labels_ab.py): i.i.d. labels out of 50 10.0 -> 3.0 ms, 8^3 blocks of labels
// 2.8 -> 2.2, i.i.d. labels out of 2 1.5 -> 2.1; 64 DISTINCT labels under every stencil (int32 noise) fill the table: 14 ms either way.
constexpr int HASH_NT = 64;
template <typename G, int D, int K>
__global__ __launch_bounds__(HASH_NT) void pull_labels_hash_kernel(KParams p, const int *__restrict__ vol, const G *__restrict__ grid,
                                                                      int *__restrict__ val, int B)
{
    typedef float R;
    __shared__ int keys[64][HASH_NT];
    __shared__ float sums[64][HASH_NT];
    const int64_t o = (int64_t)blockIdx.x * HASH_NT + threadIdx.x;
    if (o >= p.N) return;
    constexpr int T0 = K + 1, T1 = K + 1, T2 = K + 1, NT = T0 * T1 * T2;
    static_assert(D == 3 && K == 3 && NT == 64, "64 taps, 64 slots");
    const int tid = threadIdx.x;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        R x[D];
        load_coords<R, G, D>(p, grid, b, o, x);
        Stencil<R, D, K, true, NEED_W> s;
        s.setup(p, x);
        R wd[3][K + 1];
#pragma unroll
        for (int d = 0; d < D; ++d) {                                // the reference's own operations (see pull_labels_kernel)
            const R fl = (R)floorf((float)(x[d] - R(1)));
            const R t = x[d] - fl;
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                R a = t - R(j); a = a < R(0) ? -a : a;
                const R u = R(2) - a;
                const R we = a < R(1) ? (a * a * (a - R(2)) * R(3) + R(4)) / R(6) : (u * u * u) / R(6);
                wd[d][j] = s.w[d][j] == R(0) ? R(0) : (s.w[d][j] < R(0) ? -we : we);
            }
        }
        for (int c = 0; c < p.C; ++c) {
            const char *v0 = reinterpret_cast<const char *>(vol + b * p.vol_sb + c * p.vol_sc);
            unsigned long long occ = 0ull;
            // the label of the previous tap and its running sum stay in registers: a smooth map (one or two labels under most stencils)
            // touches the table a few times per voxel, not 64
            int cur_l = 0, cur_h = -1;
            R cur_s = R(0);
#pragma unroll 1
            for (int i = 0; i < T0; ++i) {
                // one x-plane (16 gathers) in flight at a time
                const unsigned oi = s.off[0][i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : 3];
                const R wx = i == 0 ? wd[0][0] : i == 1 ? wd[0][1] : i == 2 ? wd[0][2] : wd[0][3];
                int tmp[T1 * T2];
#pragma unroll
                for (int j = 0; j < T1; ++j)
#pragma unroll
                    for (int k = 0; k < T2; ++k) tmp[j * T2 + k] = *reinterpret_cast<const int *>(v0 + (oi + s.off[1][j] + s.off[2][k]));
#pragma unroll
                for (int jk = 0; jk < T1 * T2; ++jk) {
                    const int l = tmp[jk];
                    const R w = (wx * wd[1][jk / T2]) * wd[2][jk % T2];       // node-major product, sign folded in
                    if (cur_h >= 0 && l == cur_l) { cur_s += w; continue; }
                    if (cur_h >= 0) sums[cur_h][tid] = cur_s;
                    unsigned h = ((unsigned)l * 0x9E3779B1u) >> 26;
                    for (;;) {
                        if (!((occ >> h) & 1ull)) { keys[h][tid] = l; cur_s = R(0) + w; occ |= 1ull << h; break; }
                        if (keys[h][tid] == l) { cur_s = sums[h][tid] + w; break; }
                        h = (h + 1u) & 63u;
                    }
                    cur_l = l; cur_h = (int)h;
                }
            }
            if (cur_h >= 0) sums[cur_h][tid] = cur_s;
            int best_l = 0x7fffffff;
            R best_w = R(0);
            while (occ) {
                const int h = __ffsll((long long)occ) - 1;
                occ &= occ - 1ull;
                const int l = keys[h][tid];
                const R sum = sums[h][tid] * s.mask;
                if (sum > best_w || (sum == best_w && l < best_l)) { best_w = sum; best_l = l; }
            }
            val[b * p.val_sb + c * p.val_sc + o] = best_w > R(0) ? best_l : 0;
        }
    }
}

template <typename G, typename R>
int launch_g(const KParams &p, const void *vol, const void *grid, void *val, int B, hipStream_t st)
{
    const int K = p.order[0];
    for (int d = 1; d < p.dim; ++d) if (p.order[d] != K) return INTERPOL_E_ORDER;
#define IP_L(DD, KK) if (p.dim == DD && K == KK) { hipLaunchKernelGGL((pull_labels_kernel<G, R, DD, KK>), sample_grid(p, B), dim3(256), 0, st, \
                                                                        p, (const int *)vol, (const G *)grid, (int *)val, B); goto done; }
    IP_L(1, 0) IP_L(1, 1) IP_L(1, 2) IP_L(1, 3) IP_L(2, 0) IP_L(2, 1) IP_L(2, 2) IP_L(2, 3) IP_L(3, 0) IP_L(3, 1) IP_L(3, 2)
#undef IP_L
    if (p.dim == 3 && K == 3 && sizeof(R) == 4 && !(p.dbg & 2)) {     // (debug bit 2: the walk below, for A/B)
        const dim3 hgrid((unsigned)((p.N + HASH_NT - 1) / HASH_NT), (unsigned)(B < 65535 ? B : 65535), 1);
        hipLaunchKernelGGL((pull_labels_hash_kernel<G, 3, 3>), hgrid, dim3(HASH_NT), 0, st, p, (const int *)vol, (const G *)grid, (int *)val, B);
        goto done;
    }
    if (p.dim == 3 && K == 3) {
        const dim3 wgrid((unsigned)((p.N + WIDE_NT - 1) / WIDE_NT), (unsigned)(B < 65535 ? B : 65535), 1);
        hipLaunchKernelGGL((pull_labels_wide_kernel<G, R, 3, 3>), wgrid, dim3(WIDE_NT), 0, st, p, (const int *)vol, (const G *)grid, (int *)val, B);
        goto done;
    }
    return INTERPOL_E_ORDER;
done:
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace

int launch_pull_labels(const KParams &p, int grid_f64, const void *vol, const void *grid, void *val, int B, hipStream_t st)
{
    return grid_f64 ? launch_g<double, double>(p, vol, grid, val, B, st) : launch_g<float, float>(p, vol, grid, val, B, st);
}

} // namespace ip
