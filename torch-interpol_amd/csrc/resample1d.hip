// One-dimensional B-spline resampling along ONE dimension of a tensor viewed as
// (outer, n, inner): the building block of `resize` / `restrict` on tensor-product lattices
// (reference interpol/resize.py:96-117, restrict.py:88-118 build stack(meshgrid_ij(*lin), -1) and
// call the D-dimensional grid_pull / grid_push: (K+1)^D taps per voxel).  The stencil of a
// tensor-product lattice factorises, so D passes of K+1 taps give the same operator:
//     forward : dst[b, s, c]  = mask(x_s) * sum_j w_j(x_s) sign_j src[b, wrap(i0_s + j), c]
//     adjoint : dst[b, wrap(i0_s + j), c] += w_j(x_s) sign_j mask(x_s) src[b, s, c]
// with x_s = lin[s]; weights, indices, signs and mask are those of csrc/stencil.hpp
// (splines.py:30-80, bounds.py:30-89, nd.py:10-27).  One thread per (b, s, c): lanes run
// along c when inner > 1 (coalesced rows, wave-uniform stencil), along s for the last dim.
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace ip {
namespace {

template <typename T> __device__ __forceinline__ T ld_b(const T *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// grid: x = blocks over ns * inner, y = outer slices
template <typename T, typename G, typename R, int K, bool ADJ>
__global__ __launch_bounds__(256) void resample1d_kernel(KParams p, const T *__restrict__ src, const G *__restrict__ lin,
                                                         T *__restrict__ dst, unsigned ns, unsigned inner, int64_t nl, int64_t outer)
{
    const unsigned r = blockIdx.x * 256u + threadIdx.x;
    if (r >= ns * inner) return;
    const unsigned s = r / inner, c = r - s * inner;
    R x[1] = { (R)lin[s] };
    Stencil<R, 1, K, true, NEED_W> st;
    st.setup(p, x);
    for (int64_t b = blockIdx.y; b < outer; b += gridDim.y) {
        const T *sb = src + b * (ADJ ? (int64_t)ns : nl) * inner + c;
        T *db = dst + b * (ADJ ? nl : (int64_t)ns) * inner + c;
        if constexpr (!ADJ) {
            R acc = R(0);
#pragma unroll
            for (int j = 0; j <= K; ++j) acc = st.w[0][j] * Cvt<R, T>::ld(ld_b(sb, st.off[0][j])) + acc;
            db[(int64_t)s * inner] = Cvt<R, T>::st(acc * st.mask);
        } else {
            const R v = Cvt<R, T>::ld(sb[(int64_t)s * inner]) * st.mask;
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                T *q = reinterpret_cast<T *>(reinterpret_cast<char *>(db) + st.off[0][j]);
                __hip_atomic_fetch_add(q, (T)(st.w[0][j] * v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// fp32 forward pass, inner % 4 == 0: four consecutive c per thread, 16-byte loads and stores
template <int K>
__global__ __launch_bounds__(256) void resample1d_fwd_f32x4(KParams p, const float *__restrict__ src, const float *__restrict__ lin,
                                                            float *__restrict__ dst, unsigned ns, unsigned inner4, int64_t nl, int64_t outer)
{
    const unsigned r = blockIdx.x * 256u + threadIdx.x;
    if (r >= ns * inner4) return;
    const unsigned s = r / inner4, c = (r - s * inner4) * 4u;
    float x[1] = { lin[s] };
    Stencil<float, 1, K, true, NEED_W> st;
    st.setup(p, x);
    const int64_t inner = (int64_t)inner4 * 4;
    for (int64_t b = blockIdx.y; b < outer; b += gridDim.y) {
        const float *sb = src + b * nl * inner + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sb) + st.off[0][j]);
            const float w = st.w[0][j];
            acc.x = w * v.x + acc.x; acc.y = w * v.y + acc.y; acc.z = w * v.z + acc.z; acc.w = w * v.w + acc.w;
        }
        const float m = st.mask;
        *reinterpret_cast<float4 *>(dst + (b * ns + s) * inner + c) = make_float4(acc.x * m, acc.y * m, acc.z * m, acc.w * m);
    }
}

template <typename T, typename G, typename R, bool ADJ>
int launch_k(int order, const KParams &p, const void *src, const void *lin, void *dst, unsigned ns, unsigned inner, int64_t nl,
             int64_t outer, hipStream_t st)
{
    constexpr bool VEC_OK = !ADJ && std::is_same<T, float>::value && std::is_same<G, float>::value;
    const bool vec = VEC_OK && inner % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const unsigned cols = vec ? inner / 4 : inner;
    const unsigned blocks = (unsigned)(((uint64_t)ns * cols + 255) / 256);
    // The stencil of a thread depends on s only: every thread walks several outer slices with
    // the same weights / offsets (its set-up costs ~100 instructions, a tap 2).  Enough slices
    // in parallel to fill the chip (~16 K threads per CU), the rest in the loop.
    int64_t ny = (int64_t)(4u << 20) / ((int64_t)blocks * 256) + 1;
    ny = ny < outer ? ny : outer;
    ny = ny < 65535 ? ny : 65535;
    const dim3 grid(blocks, (unsigned)ny);
    if constexpr (VEC_OK) {
        if (vec) {
#define IP_R4(KK) case KK: hipLaunchKernelGGL((resample1d_fwd_f32x4<KK>), grid, dim3(256), 0, st, p, (const float *)src, \
                                              (const float *)lin, (float *)dst, ns, cols, nl, outer); break;
            switch (order) { IP_R4(0) IP_R4(1) IP_R4(2) IP_R4(3) IP_R4(4) IP_R4(5) IP_R4(6) IP_R4(7) default: return INTERPOL_E_ORDER; }
#undef IP_R4
            const hipError_t e4 = hipGetLastError();
            return e4 == hipSuccess ? 0 : (int)e4;
        }
    }
#define IP_R1(KK) case KK: hipLaunchKernelGGL((resample1d_kernel<T, G, R, KK, ADJ>), grid, dim3(256), 0, st, p, (const T *)src, \
                                              (const G *)lin, (T *)dst, ns, inner, nl, outer); break;
    switch (order) { IP_R1(0) IP_R1(1) IP_R1(2) IP_R1(3) IP_R1(4) IP_R1(5) IP_R1(6) IP_R1(7) default: return INTERPOL_E_ORDER; }
#undef IP_R1
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace

// Called by interpol_resample_1d (abi.hip).  16-bit storage: forward only.
int launch_resample1d(int dtype, int lin_f64, int order, const KParams &p, int adjoint, const void *src, const void *lin, void *dst,
                      unsigned ns, unsigned inner, int64_t nl, int64_t outer, hipStream_t st)
{
    if (dtype == INTERPOL_F64) {
        if (!lin_f64) return INTERPOL_E_DTYPE;
        return adjoint ? launch_k<double, double, double, true>(order, p, src, lin, dst, ns, inner, nl, outer, st)
                       : launch_k<double, double, double, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    }
    if (lin_f64) return INTERPOL_E_DTYPE;
    if (dtype == INTERPOL_F32)
        return adjoint ? launch_k<float, float, float, true>(order, p, src, lin, dst, ns, inner, nl, outer, st)
                       : launch_k<float, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    if (adjoint) return INTERPOL_E_DTYPE;
    if (dtype == INTERPOL_BF16) return launch_k<bf16_t, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    if (dtype == INTERPOL_F16) return launch_k<f16_t, float, float, false>(order, p, src, lin, dst, ns, inner, nl, outer, st);
    return INTERPOL_E_DTYPE;
}

} // namespace ip
