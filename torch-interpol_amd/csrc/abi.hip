// ===========================================================================
// abi.hip -- the C-ABI of libinterpol_hip.so (declared in include/interpol_hip.h).
//
// Validates the plain-C problem descriptor, converts it to the device-side
// KParams, picks the kernel family and enqueues it on the caller's stream.
// No allocation, no synchronisation, no global state.
//
// Reference interface replaced: the operator seam interpol/pushpull.py:35-325
// and interpol/coeff.py:288-313 of balbasty/torch-interpol @2024_10_08.
// ===========================================================================
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include "filter_params.hpp"
#include "defer.hpp"
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

namespace ip {

// typed launchers (ops_<dtype>.hip)
#define IP_DECL(sfx)                                                                                                   \
    int launch_pull_##sfx(const KParams &, const void *, const void *, void *, int, hipStream_t);                     \
    int launch_grad_##sfx(const KParams &, const void *, const void *, void *, int, hipStream_t);                     \
    int launch_push_##sfx(const KParams &, const void *, const void *, void *, int, hipStream_t);                     \
    int launch_pullbwd_##sfx(const KParams &, const void *, const void *, const void *, void *, void *, int, int64_t, int64_t, hipStream_t); \
    int launch_pushbwd_##sfx(const KParams &, const void *, const void *, const void *, void *, void *, int, hipStream_t); \
    int launch_narrow_##sfx(const void *, void *, int64_t, hipStream_t);
IP_DECL(f32) IP_DECL(f64) IP_DECL(bf16) IP_DECL(f16)
#undef IP_DECL
#define IP_DECL2(sfx)                                                                                                  \
    int launch_hess_##sfx(const KParams &, const void *, const void *, void *, int, hipStream_t);                     \
    int launch_pushgrad_##sfx(const KParams &, const void *, const void *, void *, int, hipStream_t);
IP_DECL2(f32) IP_DECL2(f64)
#undef IP_DECL2

int launch_filter(int dtype, const FilterParams &fp, const void *src, void *data, hipStream_t st);
bool resample1d_adjoint_gathers(int64_t n_samples, int64_t inner);
int64_t bricks_workspace_bytes(const KParams &p, int B, int shared);
int launch_push_bricks(const KParams &p, int B, int shared, const void *val, const void *grid, void *vol,
                       void *workspace, int64_t workspace_bytes, hipStream_t st);
int launch_pull_labels(const KParams &p, int grid_f64, const void *vol, const void *grid, void *val, int B, hipStream_t st);
int launch_resample1d(int dtype, int lin_f64, int order, const KParams &p, int adjoint, const void *src, const void *lin, void *dst,
                      unsigned ns, unsigned inner, int64_t nl, int64_t outer, hipStream_t st);

// fast paths (ops_tiled.hip, one translation unit per storage type); each returns 1 when it
// took the problem, 0 to decline, <0 / >0 on error
#define IP_DECL_TILED(sfx)                                                                                             \
    int try_fast_pull_##sfx(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t); \
    int try_fast_grad_##sfx(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t); \
    int try_fast_push_##sfx(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t); \
    int try_fast_pullbwd_##sfx(const interpol_problem *, const KParams &, const void *, const void *, const void *,     \
                               void *, void *, int64_t, int64_t, hipStream_t);                                         \
    int try_fast_pushbwd_##sfx(const interpol_problem *, const KParams &, const void *, const void *, const void *,     \
                               void *, void *, hipStream_t);
IP_DECL_TILED(f32) IP_DECL_TILED(bf16) IP_DECL_TILED(f16)
#undef IP_DECL_TILED

// owner-computes (target-stationary) scatter for same-resolution deformations (push_owner.hip)
int try_owner_push(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, hipStream_t, const int **);
int try_push_f64_tiles(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t);
int owner_pull_prepare(const interpol_problem *, const KParams &, void *, int64_t, hipStream_t, int **, int *);
int owner_pull_finish(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, bool, hipStream_t,
                      bool grad = false, const void *gout = nullptr, bool probed = false, bool spatial = false);
int owner_grad_probe(const interpol_problem *, const KParams &, const void *, void *, int64_t, hipStream_t, int mode = -1);
int linear_pull_probe(const interpol_problem *, const KParams &, const void *, void *, hipStream_t, const int **, int);
int64_t gather5_workspace_bytes(const interpol_problem *, const KParams &);
int try_gather5(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, int, const void *, hipStream_t, const int **);
int64_t scatter5_workspace_bytes(const interpol_problem *, const KParams &);
int try_scatter5(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, hipStream_t, const int **);
int try_backward5(const interpol_problem *, const KParams &, const KParams &, const void *, const void *, const void *, void *, void *, void *, int64_t,
                  hipStream_t, const int **);
int try_pushbwd5(const interpol_problem *, const KParams &, const void *, const void *, const void *, void *, void *, void *, int64_t, hipStream_t);
int64_t scatter2d_workspace_bytes(const interpol_problem *, const KParams &);
int64_t gather2d_workspace_bytes(const interpol_problem *, const KParams &);
int try_gather2d(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, int, const void *, hipStream_t, const int **);
int try_scatter2d(const interpol_problem *, const KParams &, const void *, const void *, void *, void *, int64_t, hipStream_t, const int **);
#ifdef IP_EXPERIMENTS
int try_pull_direct(const interpol_problem *, const KParams &, const void *, const void *, void *, int *, int, hipStream_t);
#endif
int try_sorted_pull_f32(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t);
int try_sorted_pull_bf16(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t);
int try_sorted_pull_f16(const interpol_problem *, const KParams &, const void *, const void *, void *, hipStream_t);
int try_sorted_gradc_f32(const interpol_problem *, const KParams &, const void *, const void *, const void *, void *, hipStream_t);
int64_t owner_pull_workspace_bytes(const interpol_problem *, const KParams &);
int64_t owner_workspace_bytes(const interpol_problem *, const KParams &, bool);

#define IP_TILED_BY_DTYPE(NAME, ...)                                                     \
    switch (p->dtype) {                                                                  \
    case INTERPOL_F32: return NAME##_f32(__VA_ARGS__);                                   \
    case INTERPOL_BF16: return NAME##_bf16(__VA_ARGS__);                                 \
    case INTERPOL_F16: return NAME##_f16(__VA_ARGS__);                                   \
    default: return 0; }
static int try_fast_pull(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{ IP_TILED_BY_DTYPE(try_fast_pull, p, k, vol, grid, val, st) }
static int try_fast_grad(const interpol_problem *p, const KParams &k, const void *vol, const void *grid, void *val, hipStream_t st)
{ IP_TILED_BY_DTYPE(try_fast_grad, p, k, vol, grid, val, st) }
static int try_fast_push(const interpol_problem *p, const KParams &k, const void *val, const void *grid, void *vol, hipStream_t st)
{ IP_TILED_BY_DTYPE(try_fast_push, p, k, val, grid, vol, st) }
static int try_fast_pullbwd(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid,
                            void *gvol, void *ggrid, int64_t gsb, int64_t gsc, hipStream_t st)
{ IP_TILED_BY_DTYPE(try_fast_pullbwd, p, k, gout, vol, grid, gvol, ggrid, gsb, gsc, st) }
static int try_fast_pushbwd(const interpol_problem *p, const KParams &k, const void *gvol_out, const void *val, const void *grid,
                            void *gval, void *ggrid, hipStream_t st)
{ IP_TILED_BY_DTYPE(try_fast_pushbwd, p, k, gvol_out, val, grid, gval, ggrid, st) }

static size_t esize(int dtype) { return dtype == INTERPOL_F64 ? 8 : (dtype == INTERPOL_F32 ? 4 : 2); }
static size_t acc_esize(int dtype) { return dtype == INTERPOL_F64 ? 8 : 4; }

enum Role { GATHER, SCATTER };

// Validate + convert.  `vol_elem_bytes`: element size of the indexed lattice as the
// kernel sees it (storage type for gathers, accumulation type for scatters).
static int make_params(const interpol_problem *p, Role role, int trailing, KParams *k, int *B, bool has_val = true)
{
    if (!p) return INTERPOL_E_NULL;
    if (p->abi_version != INTERPOL_ABI_VERSION) return INTERPOL_E_SHAPE;
    if (p->dim < 1 || p->dim > 3) return INTERPOL_E_DIM;
    if (p->extrapolate < 0 || p->extrapolate > 2) return INTERPOL_E_EXTRAP;
    if (p->dtype < 0 || p->dtype > 3) return INTERPOL_E_DTYPE;
    if (p->dtype == INTERPOL_F64 ? p->grid_dtype != INTERPOL_F64 : p->grid_dtype != INTERPOL_F32) return INTERPOL_E_DTYPE;
    if (p->batch < 1 || p->batch > 0x7fffffff || p->channels < 1 || p->channels > 0x7fffffff) return INTERPOL_E_SHAPE;
    memset(k, 0, sizeof(*k));
    k->dim = p->dim;
    k->extrapolate = p->extrapolate;
    bool all1 = true, all0 = true;
    int64_t N = 1;
    const size_t vb = role == GATHER ? esize(p->dtype) : acc_esize(p->dtype);
    uint64_t max_off = 0;
    for (int d = 0; d < 3; ++d) {
        if (d >= p->dim) { k->bound[d] = 1; k->order[d] = 0; k->vol_n[d] = 1; k->vol_ss[d] = 0; continue; }
        if (p->order[d] < 0 || p->order[d] > 7) return INTERPOL_E_ORDER;
        if (p->bound[d] < 0 || p->bound[d] > 6) return INTERPOL_E_BOUND;
        if (p->vol_shape[d] < 1 || p->vol_shape[d] > 0x3fffffff) return INTERPOL_E_SHAPE;
        if (p->grid_shape[d] < 1) return INTERPOL_E_SHAPE;
        if (p->vol_stride[2 + d] < 0) return INTERPOL_E_STRIDE;
        k->bound[d] = p->bound[d];
        k->order[d] = p->order[d];
        k->vol_n[d] = (int)p->vol_shape[d];
        const uint64_t sb = (uint64_t)p->vol_stride[2 + d] * vb;
        if (sb > 0x7fffffffull) return INTERPOL_E_STRIDE;
        k->vol_ss[d] = (int)sb;
        max_off += (uint64_t)(p->vol_shape[d] - 1) * sb;
        all1 = all1 && p->order[d] == 1;
        all0 = all0 && p->order[d] == 0;
        N *= p->grid_shape[d];
        k->mask_hi[d] = (double)(p->vol_shape[d] - 1) + (p->extrapolate == 2 ? 0.5 + 5e-2 : 5e-2);
    }
    if (max_off + vb > 0xffffffffull) return INTERPOL_E_SHAPE;      // one (b, c) image must fit 32-bit byte offsets
    k->mask_lo = -(p->extrapolate == 2 ? 0.5 + 5e-2 : 5e-2);
    k->mask_lo_f = (float)k->mask_lo;
    for (int d = 0; d < 3; ++d) k->mask_hi_f[d] = (float)k->mask_hi[d];
    // pushpull.py:48-66 : all orders 1 -> iso1 semantics, all 0 -> iso0, else nd
    k->mode = all1 ? MODE_ISO1 : (all0 ? MODE_ISO0 : MODE_ND);
    k->C = (int)p->channels;
    k->dbg = (p->flags >> 8) & 0xffff;
    k->gate_n = 0;
    k->N = N;
    k->vol_sb = p->vol_stride[0];
    k->vol_sc = p->vol_stride[1];
    // grid: spatial dims contiguous (row-major), component stride 1
    {
        const int modes = ((p->flags & INTERPOL_FLAG_SEPARABLE_GRID) ? 1 : 0) + ((p->flags & INTERPOL_FLAG_DISPLACEMENT) ? 1 : 0)
                        + ((p->flags & INTERPOL_FLAG_AFFINE_GRID) ? 1 : 0);
        if (modes > 1) return INTERPOL_E_STRIDE;                     // one coordinate source at a time
    }
    if (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_DISPLACEMENT | INTERPOL_FLAG_AFFINE_GRID)) {
        if (N > 0xffffffffll) return INTERPOL_E_SHAPE;               // the sample index is split in 32 bits
        k->sep = (p->flags & INTERPOL_FLAG_SEPARABLE_GRID) ? 1 : ((p->flags & INTERPOL_FLAG_DISPLACEMENT) ? 2 : 3);
    }
    for (int d = 0; d < 3; ++d) {
        if (d < p->dim && p->grid_shape[d] > 0x7fffffffll) return INTERPOL_E_SHAPE;
        k->gshape[d] = d < p->dim ? (int)p->grid_shape[d] : 1;
    }
    if (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_AFFINE_GRID)) {
        k->grid_sb = 0;
    } else {
        int64_t expect = p->dim;
        if (p->grid_stride[4] != 1 && p->dim > 1) return INTERPOL_E_STRIDE;
        for (int d = p->dim - 1; d >= 0; --d) {
            if (p->grid_shape[d] > 1 && p->grid_stride[1 + d] != expect) return INTERPOL_E_STRIDE;
            expect *= p->grid_shape[d];
        }
        k->grid_sb = p->grid_stride[0];
    }
    // val: spatial (+ trailing) dims contiguous
    if (has_val) {
        int64_t expect = trailing;
        for (int d = p->dim - 1; d >= 0; --d) {
            if (p->grid_shape[d] > 1 && p->val_stride[2 + d] != expect) return INTERPOL_E_STRIDE;
            expect *= p->grid_shape[d];
        }
        k->val_sb = p->val_stride[0];
        k->val_sc = p->val_stride[1];
    }
    *B = (int)p->batch;
    return 0;
}

static int64_t vol_numel(const interpol_problem *p)
{
    // batch stride 0 = ONE shared target that every batch item accumulates into
    int64_t n = (p->vol_stride[0] == 0 ? 1 : p->batch) * (p->channels + ((p->flags & INTERPOL_FLAG_WITH_COUNT) ? 1 : 0));
    for (int d = 0; d < p->dim; ++d) n *= p->vol_shape[d];
    return n;
}

// scatter targets must be dense (B, C, *shape) so they can be zero-filled / narrowed in one go
static bool vol_is_dense(const interpol_problem *p)
{
    int64_t expect = 1;
    for (int d = p->dim - 1; d >= 0; --d) {
        if (p->vol_shape[d] > 1 && p->vol_stride[2 + d] != expect) return false;
        expect *= p->vol_shape[d];
    }
    const int64_t nch = p->channels + ((p->flags & INTERPOL_FLAG_WITH_COUNT) ? 1 : 0);
    if (nch > 1 && p->vol_stride[1] != expect) return false;
    expect *= nch;
    if (p->batch > 1 && p->vol_stride[0] != expect && p->vol_stride[0] != 0) return false;
    return true;
}

template <typename F32, typename F64, typename BF, typename HF>
static int by_dtype(int dtype, F32 f32, F64 f64, BF bf, HF hf)
{
    switch (dtype) {
    case INTERPOL_F32: return f32();
    case INTERPOL_F64: return f64();
    case INTERPOL_BF16: return bf();
    case INTERPOL_F16: return hf();
    default: return INTERPOL_E_DTYPE;
    }
}

// Zero-fill of a target / accumulator on the stream.  A KERNEL, not hipMemsetAsync: inside a captured hipGraph the memset nodes of ROCm 7.2
// were observed to leave every fourth float of a 0.9 MB accumulator holding garbage from the second replay on, when other nodes of the
// graph had used the same addresses (tools/r5/graph2d_dbg.py: bf16 2-D push behind a bricks pull, one replay in two;
// push_owner.hip met the same with its header).  16 bytes per thread where the range allows, bytes at the edges.
__global__ __launch_bounds__(256) void zero_fill(unsigned char *__restrict__ ptr, size_t head, size_t n16, size_t tail)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    uint4 *q = reinterpret_cast<uint4 *>(ptr + head);
    for (size_t j = i; j < n16; j += step) q[j] = make_uint4(0u, 0u, 0u, 0u);
    if (i < head) ptr[i] = 0;
    if (i < tail) ptr[head + n16 * 16 + i] = 0;
}
static hipError_t zero_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    const size_t mis = (size_t)((uintptr_t)ptr & 15u);
    const size_t head = mis ? (16 - mis < bytes ? 16 - mis : bytes) : 0;
    const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
    size_t blocks = (n16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    hipLaunchKernelGGL(zero_fill, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)ptr, head, n16, tail);
    return hipGetLastError();
}

// common driver of the scatter-type operators
template <typename Launch>
static int scatter_driver(const interpol_problem *p, int trailing, bool need_val, const void *val, const void *grid,
                          void *vol, void *scratch, int64_t scratch_bytes, hipStream_t st, Launch launch)
{
    KParams k; int B;
    int rc = make_params(p, SCATTER, trailing, &k, &B, need_val);
    if (rc) return rc;
    if (!grid || !vol || (need_val && !val)) return INTERPOL_E_NULL;
    if (!vol_is_dense(p)) return INTERPOL_E_STRIDE;
    const int64_t numel = vol_numel(p);
    const bool lowp = (p->dtype == INTERPOL_BF16 || p->dtype == INTERPOL_F16);
    if (lowp && (p->flags & INTERPOL_FLAG_ACCUMULATE)) return INTERPOL_E_DTYPE;   // a 16-bit target is narrowed once, it cannot accumulate
    void *acc = vol;
    // scratch = [fp32 accumulator of a 16-bit target, 256-byte aligned size] [workspace of the binned scatter]
    void *ws = scratch;
    int64_t ws_bytes = scratch ? scratch_bytes : 0;
    if (lowp) {
        if (!scratch) return INTERPOL_E_NULL;
        if (scratch_bytes < numel * 4) return INTERPOL_E_SCRATCH;
        acc = scratch;
        const int64_t used = (numel * 4 + 255) & ~(int64_t)255;
        ws = (char *)scratch + used;
        ws_bytes = scratch_bytes - used;
    }
    if (ws_bytes <= 0) { ws = nullptr; ws_bytes = 0; }
    if (!(p->flags & INTERPOL_FLAG_ACCUMULATE) || lowp) {
        hipError_t e = zero_async(acc, (size_t)numel * acc_esize(p->dtype), st);
        if (e != hipSuccess) return (int)e;
    }
    rc = launch(k, B, acc, ws, ws_bytes);
    if (rc) return rc;
    if (lowp) {
        rc = p->dtype == INTERPOL_BF16 ? launch_narrow_bf16(acc, vol, numel, st) : launch_narrow_f16(acc, vol, numel, st);
    }
    return rc;
}

} // namespace ip

using namespace ip;

extern "C" {

int32_t interpol_abi_version(void) { return INTERPOL_ABI_VERSION; }

const char *interpol_error_string(int code)
{
    switch (code) {
    case INTERPOL_OK: return "ok";
    case INTERPOL_E_DIM: return "dim must be 1, 2 or 3";
    case INTERPOL_E_ORDER: return "spline order must be in 0..7";
    case INTERPOL_E_BOUND: return "unknown boundary code";
    case INTERPOL_E_DTYPE: return "unsupported dtype combination";
    case INTERPOL_E_SHAPE: return "bad or too large extent";
    case INTERPOL_E_NULL: return "required pointer is NULL";
    case INTERPOL_E_EXTRAP: return "extrapolate must be 0, 1 or 2";
    case INTERPOL_E_PREFILTER: return "prefilter not implemented for dst1/dst2";
    case INTERPOL_E_SCRATCH: return "scratch buffer too small";
    case INTERPOL_E_STRIDE: return "unsupported stride pattern";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int interpol_pull(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        rc = try_fast_pull(p, k, vol, grid, val, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    return by_dtype(p->dtype,
        [&] { return launch_pull_f32(k, vol, grid, val, B, st); },
        [&] { return launch_pull_f64(k, vol, grid, val, B, st); },
        [&] { return launch_pull_bf16(k, vol, grid, val, B, st); },
        [&] { return launch_pull_f16(k, vol, grid, val, B, st); });
}

/* grid_pull with a workspace: the deformation-independent organisation (bricks of the image, push_owner.hip: own_gather) for
 * fields too rough for the sample tiles, chosen by a probe of the call (INTERPOL_FLAG_AUTO_SCATTER) or always
 * (INTERPOL_FLAG_BINNED_SCATTER); without a (large enough, 256-byte aligned) workspace: interpol_pull. */
// the trilinear pull with a router (push_owner.hip: lin_probe): float32, dense grids and displacement fields, tiles' sizes
static bool linear_routed(const interpol_problem *p, const KParams &k)
{
    // (16-bit storage: dense grids, the pull alone -- what ops_sorted.hip instantiates)
    const bool lowp = p->dtype == INTERPOL_BF16 || p->dtype == INTERPOL_F16;
    if (p->dim != 3 || (p->dtype != INTERPOL_F32 && !lowp) || p->grid_dtype != INTERPOL_F32 || (k.sep != 0 && (k.sep != 2 || lowp))) return false;
    if (!(p->flags & (INTERPOL_FLAG_AUTO_SCATTER | INTERPOL_FLAG_BINNED_SCATTER)) || (k.dbg & 32)) return false;
    int64_t n = 1;
    for (int d = 0; d < 3; ++d) { if (k.order[d] != 1 || p->grid_shape[d] > 0x7fffffff / 4) return false; n *= p->grid_shape[d]; }
    return n >= 32768 && p->batch <= 65535 && (uint64_t)n * 12ull <= 0xffffffffull;
}

int64_t interpol_pull_workspace(const interpol_problem *p)
{
    KParams k; int B;
    if (make_params(p, GATHER, 1, &k, &B, false)) return 0;          // (the layout of `val` -- pull, grad, a gradient -- plays no part)
    if (p->flags & INTERPOL_FLAG_NO_FASTPATH) return 0;
    if (p->dim == 2) return gather2d_workspace_bytes(p, k);         // 2-D (scatter2d.hip: gather2d)
    if (linear_routed(p, k)) return 256;                             // trilinear: the verdict of the call's probe, nothing else
    const int64_t b5 = gather5_workspace_bytes(p, k);               // orders 4 and 5 (gather5.hip)
    return b5 ? b5 : owner_pull_workspace_bytes(p, k);
}

// The routed pull (HISTORY.md 4.2e).  1: done, 0: declined (nothing launched that the caller's fallback would not overwrite), else an error.
static int routed_pull(const interpol_problem *p, KParams k, int B, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes,
                       hipStream_t st)
{
    int *flags = nullptr;
    int nzero = 0;
    if (linear_routed(p, k) && workspace && workspace_bytes >= 256 && ((uintptr_t)workspace & 255u) == 0) {
        // trilinear (round 5): the class-sorted tiles for rough fields, the generic kernel for smooth ones, chosen by a probe of the call
        KParams kt = k;
        if (!(p->flags & INTERPOL_FLAG_BINNED_SCATTER)) {
            const int *gate = nullptr;
            int rl = linear_pull_probe(p, k, grid, workspace, st, &gate, 16);
            if (rl) return rl;
            kt.verdict = gate; kt.gate_n = -3;                       // the tiles run on verdict 1 ...
            k.gate = gate; k.gate_n = -1;                            // ... the generic kernel unless the verdict is 1
        }
        int rl = p->dtype == INTERPOL_F32 ? try_sorted_pull_f32(p, kt, vol, grid, val, st)
               : (p->dtype == INTERPOL_BF16 ? try_sorted_pull_bf16(p, kt, vol, grid, val, st) : try_sorted_pull_f16(p, kt, vol, grid, val, st));
        if (rl != 0 && rl != 1) return rl;
        if (rl == 1 && (p->flags & INTERPOL_FLAG_BINNED_SCATTER)) return 1;
        if (rl == 0) { k.gate = nullptr; k.gate_n = 0; }             // (the tiles declined: the generic kernel, unconditionally)
        rl = by_dtype(p->dtype,
            [&] { return launch_pull_f32(k, vol, grid, val, B, st); },
            [&] { return (int)INTERPOL_E_DTYPE; },
            [&] { return launch_pull_bf16(k, vol, grid, val, B, st); },
            [&] { return launch_pull_f16(k, vol, grid, val, B, st); });
        return rl ? rl : 1;
    }
    if (p->dim == 2) {
        // 2-D (scatter2d.hip: gather2d): the bricks always, or behind a probe of the call next to the lean tiles, which read the verdict
        const int *gate = nullptr;
        int r2 = try_gather2d(p, k, vol, grid, val, workspace, workspace_bytes, 0, nullptr, st, &gate);
        if (r2 != 2) return r2;
        k.gate = gate; k.gate_n = -1;
        r2 = try_fast_pull(p, k, vol, grid, val, st);
        if (r2 != 0 && r2 != 1) return r2;
        // the generic kernel: the third organisation (verdict 2: an expanding field); with the tiles declined, whatever the bricks left
        if (r2 == 1) k.gate_n = -2;
        r2 = by_dtype(p->dtype,
            [&] { return launch_pull_f32(k, vol, grid, val, B, st); },
            [&] { return (int)INTERPOL_E_DTYPE; },
            [&] { return launch_pull_bf16(k, vol, grid, val, B, st); },
            [&] { return launch_pull_f16(k, vol, grid, val, B, st); });
        return r2 ? r2 : 1;
    }
    if (k.order[0] >= 4) {
        // orders 4 and 5 (gather5.hip): the bricks always, or behind a probe of the call next to the tiles, which read the same verdict
        const int *gate = nullptr;
        int r5 = try_gather5(p, k, vol, grid, val, workspace, workspace_bytes, 0, nullptr, st, &gate);
        if (r5 != 2) return r5;
        k.gate = gate; k.gate_n = -1;
        r5 = try_fast_pull(p, k, vol, grid, val, st);
        if (r5 != 0) return r5;
        r5 = launch_pull_f32(k, vol, grid, val, B, st);                // (the tiles declined: the generic kernel, behind the same verdict)
        return r5 ? r5 : 1;
    }
    int rc = owner_pull_prepare(p, k, workspace, workspace_bytes, st, &flags, &nzero);
    if (rc != 1) return rc;
    if (p->flags & INTERPOL_FLAG_BINNED_SCATTER) { rc = owner_pull_finish(p, k, vol, grid, val, workspace, workspace_bytes, true, st); return rc ? rc : 1; }
    // A probe of the call first (round 5; stateless, no host synchronisation, hipGraph-safe -- the same kernel as the scatters' and
    // the grid gradient's): a rough dense sampling goes to the bricks altogether and pull_sorted returns at once.  Else the sample
    // tiles run and flag the tiles they leave to the bricks (too many samples outside their LDS box), as in round 4.
    const bool probe = p->dtype == INTERPOL_F32 && !(k.dbg & (32 | 16384));       // (debug bit 16384: no probe, the per-tile hand-over alone)
    if (probe) {
        rc = owner_grad_probe(p, k, grid, workspace, workspace_bytes, st, -2);     // (clears the header and the brick counters as well)
        if (rc) return rc;
    }
    k.gate = flags; k.gate_n = probe ? 0 : nzero;
    k.verdict = probe ? flags - nzero : nullptr;                     // ProbeHdr::gate, the first word of the header
#ifdef IP_EXPERIMENTS
    if (p->dtype == INTERPOL_F32 && !(k.dbg & (4096 | 32)) && (p->flags & INTERPOL_FLAG_SMALL_TILES)) {
        // (opt-in experiment) the single-pass small-box tiles first (pull_direct.hip: smooth deformations); they write every flag
        // -- 0: served, 2: left to pull_sorted, which then runs on the flagged tiles only (gate_n < 0).  Declined (0): as before.
        rc = try_pull_direct(p, k, vol, grid, val, flags, nzero, st);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1) {
            k.gate_n = -nzero;
            rc = try_sorted_pull_f32(p, k, vol, grid, val, st);
            if (rc != 0 && rc != 1) return rc;
            if (rc == 1) { rc = owner_pull_finish(p, k, vol, grid, val, workspace, workspace_bytes, false, st); return rc ? rc : 1; }
            k.gate = nullptr; k.gate_n = 0;                          // (pull_sorted declined: the generic kernel serves every sample again)
            rc = launch_pull_f32(k, vol, grid, val, B, st);
            return rc ? rc : 1;
        }
    }
#endif
    // (the class-sorted tiles themselves, not try_fast_pull: they alone clear the header and the brick counters on their way and
    //  write every tile's flag -- a kernel that ignored `gate_n` would leave own_bin / own_gather with a dirty workspace)
    if (p->dtype != INTERPOL_F32 || (k.dbg & 32)) return 0;
    rc = try_sorted_pull_f32(p, k, vol, grid, val, st);
    if (rc != 1) return rc;
    if (k.dbg & 32768) return 1;                                     // (ablation: the tiles with their flags, no brick kernels behind them)
    rc = owner_pull_finish(p, k, vol, grid, val, workspace, workspace_bytes, false, st, false, nullptr, probe);
    return rc ? rc : 1;
}

int interpol_pull_ws(const interpol_problem *p, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        rc = routed_pull(p, k, B, vol, grid, val, workspace, workspace_bytes, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
        rc = try_fast_pull(p, k, vol, grid, val, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    return by_dtype(p->dtype,
        [&] { return launch_pull_f32(k, vol, grid, val, B, st); },
        [&] { return launch_pull_f64(k, vol, grid, val, B, st); },
        [&] { return launch_pull_bf16(k, vol, grid, val, B, st); },
        [&] { return launch_pull_f16(k, vol, grid, val, B, st); });
}

int interpol_pull_labels(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream)
{
    if (p && (p->dtype != INTERPOL_F32 || p->grid_dtype != INTERPOL_F32)) return INTERPOL_E_DTYPE;   // int32 labels, float32 coordinates
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    return launch_pull_labels(k, 0, vol, grid, val, B, (hipStream_t)stream);
}

int interpol_grad(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, p ? p->dim : 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        rc = try_fast_grad(p, k, vol, grid, val, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    return by_dtype(p->dtype,
        [&] { return launch_grad_f32(k, vol, grid, val, B, st); },
        [&] { return launch_grad_f64(k, vol, grid, val, B, st); },
        [&] { return launch_grad_bf16(k, vol, grid, val, B, st); },
        [&] { return launch_grad_f16(k, vol, grid, val, B, st); });
}

// grid_grad with the bricks workspace of interpol_pull_workspace(p) (float32, 3-D quadratic / cubic; DESIGN.md 4.4, HISTORY.md 4.2e): the bricks of the
// image always (INTERPOL_FLAG_BINNED_SCATTER) or when the probe of the call finds a dense or rough sampling
// (INTERPOL_FLAG_AUTO_SCATTER: the tile / generic kernels are enqueued as well and return at once behind the probe's gate).
int interpol_grad_ws(const interpol_problem *p, const void *vol, const void *grid, void *val, void *workspace, int64_t workspace_bytes, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, p ? p->dim : 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    int *flags = nullptr;
    int nzero = 0;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH) && p->dtype == INTERPOL_F32 && linear_routed(p, k) && workspace && workspace_bytes >= 256
        && ((uintptr_t)workspace & 255u) == 0) {
        // trilinear (round 5): the generic kernel up to sigma ~ 1 (0.67 ms at the identity against 1.36 for the tiles, 4 x 2 x 256^3), the LDS
        // tiles beyond (sigma = 2 / 4: 1.8 / 3.6 against 2.7 / 4.7 ms) -- the probe of the trilinear pull, its threshold at 3.5 voxels of mean
        // |second difference| of the coordinates summed over the dims (sigma ~ 0.6: the gated generic launch, which strides over the batch,
        // loses to the tiles from there on -- sigma = 1: 2.1 against 1.5 ms); both kernels enqueued behind the verdict
        interpol_problem pt = *p;
        pt.flags |= INTERPOL_FLAG_FORCE_TILED;
        KParams kt = k, kg = k;
        if (!(p->flags & INTERPOL_FLAG_BINNED_SCATTER)) {
            const int *gate = nullptr;
            rc = linear_pull_probe(p, k, grid, workspace, st, &gate, 56);
            if (rc) return rc;
            kt.verdict = gate; kt.gate_n = -3;                       // the tiles run on verdict 1 ...
            kg.gate = gate; kg.gate_n = -1;                          // ... the generic kernel unless the verdict is 1
        }
        rc = try_fast_grad(&pt, kt, vol, grid, val, st);
        if (rc != 0 && rc != 1) return rc;
        if (rc == 1 && (p->flags & INTERPOL_FLAG_BINNED_SCATTER)) return 0;
        if (rc == 0) { kg.gate = nullptr; kg.gate_n = 0; }           // (the tiles declined: the generic kernel, unconditionally)
        return launch_grad_f32(kg, vol, grid, val, B, st);
    }
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH) && k.order[0] >= 4) {
        const int *gate = nullptr;
        rc = try_gather5(p, k, vol, grid, val, workspace, workspace_bytes, 2, nullptr, st, &gate);
        if (rc == 1) return 0;
        if (rc != 0 && rc != 2) return rc;
        if (rc == 2) {
            KParams kt = k;
            kt.gate = gate; kt.gate_n = -1;
            rc = try_fast_grad(p, kt, vol, grid, val, st);
            if (rc != 0 && rc != 1) return rc;
            return rc == 1 ? 0 : launch_grad_f32(kt, vol, grid, val, B, st);
        }
    }
    if ((p->flags & INTERPOL_FLAG_NO_FASTPATH) || p->dtype != INTERPOL_F32 || owner_pull_prepare(p, k, workspace, workspace_bytes, st, &flags, &nzero) != 1)
        return interpol_grad(p, vol, grid, val, stream);
    if (p->flags & INTERPOL_FLAG_BINNED_SCATTER) return owner_pull_finish(p, k, vol, grid, val, workspace, workspace_bytes, true, st, false, nullptr, false, true);
    rc = owner_grad_probe(p, k, grid, workspace, workspace_bytes, st);
    if (rc) return rc;
    KParams kt = k;
    kt.gate = flags - nzero; kt.gate_n = -1;                         // the probe's verdict (ProbeHdr::gate, the first word of the header): 1 = the bricks
    rc = try_fast_grad(p, kt, vol, grid, val, st);
    if (rc != 0 && rc != 1) return rc;
    if (rc == 0) {
        rc = launch_grad_f32(kt, vol, grid, val, B, st);
        if (rc) return rc;
    }
    return owner_pull_finish(p, k, vol, grid, val, workspace, workspace_bytes, true, st, false, nullptr, true, true);
}

int interpol_hess(const interpol_problem *p, const void *vol, const void *grid, void *val, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, p ? p->dim * p->dim : 1, &k, &B);
    if (rc) return rc;
    if (!vol || !grid || !val) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    switch (p->dtype) {
    case INTERPOL_F32: return launch_hess_f32(k, vol, grid, val, B, st);
    case INTERPOL_F64: return launch_hess_f64(k, vol, grid, val, B, st);
    default: return INTERPOL_E_DTYPE;          // second-order ops: f32 / f64 only (host upcasts)
    }
}

int interpol_push(const interpol_problem *p, const void *val, const void *grid, void *vol,
                  void *scratch, int64_t scratch_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const bool with_count = p && (p->flags & INTERPOL_FLAG_WITH_COUNT);
    return scatter_driver(p, 1, true, val, grid, vol, scratch, scratch_bytes, st, [&](const KParams &k0, int B, void *acc, void *ws, int64_t ws_bytes) {
        KParams k = k0;
        k.cc = with_count ? 1 : 0;
        bool routed2d = false;
        if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
            int rc = try_owner_push(p, k, val, grid, acc, ws, ws_bytes, st, &k.gate);     // needs its workspace: interpol_scatter_workspace
            if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;       // (2: launched behind the probe's gate; the kernels below read the same gate)
            if (rc == 0 && p->dtype == INTERPOL_F32) {
                rc = try_scatter5(p, k, val, grid, acc, ws, ws_bytes, st, &k.gate);      // orders 4 - 5 through bricks of the target (gather5.hip)
                if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;
            }
            if (rc == 0 && p->dim == 2) {
                rc = try_scatter2d(p, k, val, grid, acc, ws, ws_bytes, st, &k.gate);    // 2-D through bricks of the target (scatter2d.hip)
                if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;
                routed2d = rc == 2;
            }
            rc = try_fast_push(p, k, val, grid, acc, st);           // the tiled kernel splats values and count in one pass
            if (rc != 0 && !(rc == 1 && routed2d)) return rc == 1 ? 0 : rc;
            if (rc == 1) k.gate_n = -2;                              // (2-D router: the generic kernels below run on verdict 2 alone -- a target sampled sparsely)
            else if (routed2d) k.gate_n = -4;                        // (... and where the tiles declined they serve every verdict but 1, the bricks')
        }
        k.cc = 0;
        if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH) && p->dtype == INTERPOL_F64) {
            // float64 on LDS tiles (push_f64.hip); the count channel is a second launch into channel C of the accumulator
            int rc = try_push_f64_tiles(p, k, val, grid, acc, st);
            if (rc != 0 && rc != 1) return rc;
            if (rc == 1) {
                if (!with_count) return 0;
                KParams kc = k;
                kc.C = 1;
                char *accc = (char *)acc + (size_t)k.C * (size_t)k.vol_sc * acc_esize(p->dtype);
                rc = try_push_f64_tiles(p, kc, nullptr, grid, accc, st);
                if (rc == 1) return 0;
                return rc ? rc : launch_push_f64(kc, nullptr, grid, accc, B, st);
            }
        }
        int rc = by_dtype(p->dtype,
            [&] { return launch_push_f32(k, val, grid, acc, B, st); },
            [&] { return launch_push_f64(k, val, grid, acc, B, st); },
            [&] { return launch_push_bf16(k, val, grid, acc, B, st); },
            [&] { return launch_push_f16(k, val, grid, acc, B, st); });
        if (rc || !with_count) return rc;
        // generic kernels: the count is a second launch into channel C of the accumulator
        KParams kc = k;
        kc.C = 1;
        char *accc = (char *)acc + (size_t)k.C * (size_t)k.vol_sc * acc_esize(p->dtype);
        return by_dtype(p->dtype,
            [&] { return launch_push_f32(kc, nullptr, grid, accc, B, st); },
            [&] { return launch_push_f64(kc, nullptr, grid, accc, B, st); },
            [&] { return launch_push_bf16(kc, nullptr, grid, accc, B, st); },
            [&] { return launch_push_f16(kc, nullptr, grid, accc, B, st); });
    });
}

static int bricks_params(const interpol_problem *p, KParams *k, int *B)
{
    if (!p) return INTERPOL_E_NULL;
    if (p->dim != 3) return INTERPOL_E_DIM;
    if (p->dtype != INTERPOL_F32 || p->grid_dtype != INTERPOL_F32) return INTERPOL_E_DTYPE;
    int rc = make_params(p, SCATTER, 1, k, B, true);
    if (rc) return rc;
    if (!vol_is_dense(p)) return INTERPOL_E_STRIDE;
    k->cc = (p->flags & INTERPOL_FLAG_WITH_COUNT) ? 1 : 0;
    if (k->C + k->cc > 4 || (uint64_t)*B * (uint64_t)k->N > 0xffffffffull) return INTERPOL_E_SHAPE;
    return 0;
}

int64_t interpol_push_bricks_workspace(const interpol_problem *p)
{
    KParams k; int B;
    const int rc = bricks_params(p, &k, &B);
    if (rc) return rc;
    return bricks_workspace_bytes(k, B, p->vol_stride[0] == 0);
}

int interpol_push_bricks(const interpol_problem *p, const void *val, const void *grid, void *vol,
                         void *workspace, int64_t workspace_bytes, void *stream)
{
    KParams k; int B;
    int rc = bricks_params(p, &k, &B);
    if (rc) return rc;
    if (!val || !grid || !vol || !workspace) return INTERPOL_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_ACCUMULATE)) {
        const hipError_t e = zero_async(vol, (size_t)vol_numel(p) * 4, st);
        if (e != hipSuccess) return (int)e;
    }
    rc = launch_push_bricks(k, B, p->vol_stride[0] == 0, val, grid, vol, workspace, workspace_bytes, st);
    return rc == 1 ? 0 : (rc == 0 ? INTERPOL_E_SHAPE : rc);
}

int64_t interpol_scatter_workspace(const interpol_problem *p, int32_t count_only)
{
    KParams k; int B;
    if (!p || (p->flags & INTERPOL_FLAG_NO_FASTPATH)) return 0;
    if (make_params(p, SCATTER, 1, &k, &B, !count_only)) return 0;
    k.cc = (!count_only && (p->flags & INTERPOL_FLAG_WITH_COUNT)) ? 1 : 0;
    if (p->dtype != INTERPOL_F32 && p->dtype != INTERPOL_BF16 && p->dtype != INTERPOL_F16) return 0;
    int64_t ws = owner_workspace_bytes(p, k, count_only != 0);
    if (ws <= 0 && p->dtype == INTERPOL_F32 && (p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER)))
        ws = scatter5_workspace_bytes(p, k);                         // orders 4 - 5 (gather5.hip: scatter5)
    if (ws <= 0) ws = scatter2d_workspace_bytes(p, k);               // 2-D (scatter2d.hip)
    if (ws <= 0) return 0;
    const bool lowp = (p->dtype == INTERPOL_BF16 || p->dtype == INTERPOL_F16);
    return ws + (lowp ? ((vol_numel(p) * 4 + 255) & ~(int64_t)255) : 0);
}

int interpol_count(const interpol_problem *p, const void *grid, void *vol,
                   void *scratch, int64_t scratch_bytes, void *stream)
{
    if (p && p->channels != 1) return INTERPOL_E_SHAPE;
    if (p && (p->flags & INTERPOL_FLAG_WITH_COUNT)) return INTERPOL_E_STRIDE;      // interpol_push only
    hipStream_t st = (hipStream_t)stream;
    return scatter_driver(p, 1, false, nullptr, grid, vol, scratch, scratch_bytes, st, [&](const KParams &k0, int B, void *acc, void *ws, int64_t ws_bytes) {
        KParams k = k0;
        bool routed2d = false;
        if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
            int rc = try_owner_push(p, k, nullptr, grid, acc, ws, ws_bytes, st, &k.gate);
            if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;
            if (rc == 0 && p->dtype == INTERPOL_F32) {
                rc = try_scatter5(p, k, nullptr, grid, acc, ws, ws_bytes, st, &k.gate);
                if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;
            }
            if (rc == 0 && p->dim == 2) {
                rc = try_scatter2d(p, k, nullptr, grid, acc, ws, ws_bytes, st, &k.gate);
                if (rc != 0 && rc != 2) return rc == 1 ? 0 : rc;
                routed2d = rc == 2;
            }
            rc = try_fast_push(p, k, nullptr, grid, acc, st);
            if (rc != 0 && !(rc == 1 && routed2d)) return rc == 1 ? 0 : rc;
            if (rc == 1) k.gate_n = -2;
            else if (routed2d) k.gate_n = -4;                        // (the tiles declined: the generic kernel serves every verdict but the bricks')
        }
        if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH) && p->dtype == INTERPOL_F64) {
            const int rc = try_push_f64_tiles(p, k, nullptr, grid, acc, st);          // float64 on LDS tiles (push_f64.hip)
            if (rc != 0) return rc == 1 ? 0 : rc;
        }
        return by_dtype(p->dtype,
            [&] { return launch_push_f32(k, nullptr, grid, acc, B, st); },
            [&] { return launch_push_f64(k, nullptr, grid, acc, B, st); },
            [&] { return launch_push_bf16(k, nullptr, grid, acc, B, st); },
            [&] { return launch_push_f16(k, nullptr, grid, acc, B, st); });
    });
}

int interpol_pushgrad(const interpol_problem *p, const void *val, const void *grid, void *vol,
                      void *scratch, int64_t scratch_bytes, void *stream)
{
    if (p && p->dtype != INTERPOL_F32 && p->dtype != INTERPOL_F64) return INTERPOL_E_DTYPE;
    if (p && (p->flags & INTERPOL_FLAG_WITH_COUNT)) return INTERPOL_E_STRIDE;      // interpol_push only
    hipStream_t st = (hipStream_t)stream;
    return scatter_driver(p, p ? p->dim : 1, true, val, grid, vol, scratch, scratch_bytes, st, [&](const KParams &k, int B, void *acc, void *, int64_t) {
        return p->dtype == INTERPOL_F32 ? launch_pushgrad_f32(k, val, grid, acc, B, st)
                                        : launch_pushgrad_f64(k, val, grid, acc, B, st);
    });
}

// The grid gradient of a gather through the router (float32, 3-D quadratic / cubic, a bricks workspace of interpol_pull_workspace(p)
// bytes; DESIGN.md 4.4, HISTORY.md 4.2e): ggrid[b,o,:] = mask * sum_c gout[b,c,o] * grad pull(vol[b,c])(x_o) (pushpull.py:256-257; gout == NULL: ones).
// The sample tiles flag the tiles whose samples leave their LDS box; those samples go to the bricks of the image
// (own_gather<K, true>, push_owner.hip) -- 4 x 2 x 256^3 cubic, sigma = 6: 20 -> 2.9 ms.  1: done, 0: declined, else an error.
static int routed_gradc(const interpol_problem *p, const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid,
                        void *scratch, int64_t scratch_bytes, hipStream_t st)
{
    int *flags = nullptr;
    int nzero = 0;
    if (linear_routed(p, k) && p->dtype == INTERPOL_F32 && k.mode == MODE_ISO1 && scratch && scratch_bytes >= 256 && ((uintptr_t)scratch & 255u) == 0) {
        // trilinear (round 5): the class-sorted tiles for rough fields, the generic fused kernel for smooth ones (push_owner.hip: lin_probe)
        KParams kt = k, kg = k;
        if (!(p->flags & INTERPOL_FLAG_BINNED_SCATTER)) {
            const int *gate = nullptr;
            const int rl = linear_pull_probe(p, k, grid, scratch, st, &gate, 64);
            if (rl) return rl;
            kt.verdict = gate; kt.gate_n = -3;
            kg.gate = gate; kg.gate_n = -1;
        }
        int rl = try_sorted_gradc_f32(p, kt, gout, vol, grid, ggrid, st);
        if (rl != 0 && rl != 1) return rl;
        if (rl == 1 && (p->flags & INTERPOL_FLAG_BINNED_SCATTER)) return 1;
        if (rl == 0) { kg.gate = nullptr; kg.gate_n = 0; }
        const int B = (int)p->batch;
        rl = gout ? launch_pullbwd_f32(kg, gout, vol, grid, nullptr, ggrid, B, 0, 0, st) : launch_pushbwd_f32(kg, vol, nullptr, grid, nullptr, ggrid, B, st);
        return rl ? rl : 1;
    }
    if (p->dim == 2) {
        // 2-D (scatter2d.hip: gather2d, mode 1): the bricks always, or behind a probe of the call next to the lean tiles (gradc2d)
        const int *gate = nullptr;
        int r2 = try_gather2d(p, k, vol, grid, ggrid, scratch, scratch_bytes, 1, gout, st, &gate);
        if (r2 != 2) return r2;
        KParams kg = k;
        kg.gate = gate; kg.gate_n = -1;
        r2 = try_fast_pullbwd(p, kg, gout, vol, grid, nullptr, ggrid, 0, 0, st);
        if (r2 != 1) return r2;                                      // (0: the tiles declined -- the caller's kernels write the same numbers over the bricks')
        kg.gate_n = -2;                                              // the generic kernel: the organisation of expanding fields (verdict 2)
        const int B = (int)p->batch;
        r2 = gout ? by_dtype(p->dtype,
                [&] { return launch_pullbwd_f32(kg, gout, vol, grid, nullptr, ggrid, B, 0, 0, st); },
                [&] { return (int)INTERPOL_E_DTYPE; },
                [&] { return launch_pullbwd_bf16(kg, gout, vol, grid, nullptr, ggrid, B, 0, 0, st); },
                [&] { return launch_pullbwd_f16(kg, gout, vol, grid, nullptr, ggrid, B, 0, 0, st); })
                  : by_dtype(p->dtype,
                [&] { return launch_pushbwd_f32(kg, vol, nullptr, grid, nullptr, ggrid, B, st); },
                [&] { return (int)INTERPOL_E_DTYPE; },
                [&] { return launch_pushbwd_bf16(kg, vol, nullptr, grid, nullptr, ggrid, B, st); },
                [&] { return launch_pushbwd_f16(kg, vol, nullptr, grid, nullptr, ggrid, B, st); });
        return r2 ? r2 : 1;
    }
    if (k.order[0] >= 4) {                                           // orders 4 and 5: gather5.hip, the bricks always (1), or declined (0)
        const int r5 = try_gather5(p, k, vol, grid, ggrid, scratch, scratch_bytes, 1, gout, st, nullptr);
        return r5 == 2 ? INTERPOL_E_SHAPE : r5;
    }
    const int r = owner_pull_prepare(p, k, scratch, scratch_bytes, st, &flags, &nzero);
    if (r != 1) return r;
    int rc;
    if (p->flags & INTERPOL_FLAG_BINNED_SCATTER) {
        rc = owner_pull_finish(p, k, vol, grid, ggrid, scratch, scratch_bytes, true, st, true, gout);
        return rc ? rc : 1;
    }
    // a probe of the call first: dense samplings go to the bricks altogether (the tile kernel returns at once: gate_n < 0,
    // the probe's header lies -gate_n ints in front of the flags), else the tiles run and flag what they leave
    rc = owner_grad_probe(p, k, grid, scratch, scratch_bytes, st);
    if (rc) return rc;
    KParams kg = k;
    kg.gate = flags; kg.gate_n = -nzero;
    rc = try_sorted_gradc_f32(p, kg, gout, vol, grid, ggrid, st);
    if (rc != 1) return rc;
    if (k.dbg & 32768) return 1;                                     // (ablation: the tiles with their flags, no brick kernels behind them)
    rc = owner_pull_finish(p, k, vol, grid, ggrid, scratch, scratch_bytes, false, st, true, gout, true);
    return rc ? rc : 1;
}

int interpol_pull_backward(const interpol_problem *p, const void *grad_out, const void *vol, const void *grid,
                           void *grad_vol, void *grad_grid, void *scratch, int64_t scratch_bytes, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B);
    if (rc) return rc;
    if (!grad_out || !vol || !grid) return INTERPOL_E_NULL;
    if (!grad_vol && !grad_grid) return 0;
    if (grad_grid && (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_AFFINE_GRID))) return INTERPOL_E_STRIDE;   // no per-sample grid to differentiate
    hipStream_t st = (hipStream_t)stream;
    // grad_vol is a dense (B, C, *vol_shape) buffer; vol must be spatially contiguous so
    // that both share tap offsets (channel / batch strides are free)
    {
        int64_t expect = 1;
        for (int d = p->dim - 1; d >= 0; --d) {
            if (p->vol_shape[d] > 1 && p->vol_stride[2 + d] != expect) return INTERPOL_E_STRIDE;
            expect *= p->vol_shape[d];
        }
    }
    int64_t img = 1;
    for (int d = 0; d < p->dim; ++d) img *= p->vol_shape[d];
    const int64_t numel = img * p->channels * p->batch;
    const bool lowp = (p->dtype == INTERPOL_BF16 || p->dtype == INTERPOL_F16);
    void *acc = grad_vol;
    if (grad_vol) {
        if (lowp) {
            if (!scratch) return INTERPOL_E_NULL;
            if (scratch_bytes < numel * 4) return INTERPOL_E_SCRATCH;
            acc = scratch;
        }
        if (!(p->flags & INTERPOL_FLAG_ACCUMULATE) || lowp) {
            hipError_t e = zero_async(acc, (size_t)numel * acc_esize(p->dtype), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    const int64_t gsb = img * p->channels, gsc = img;
    rc = 0;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        bool vol_done = false, grid_done = false;
        // (one channel + grid gradient at orders <= 3: the fused kernel is faster; at orders >= 4 the
        //  grid part has its own shifted-pair kernel and the split wins again: config 3 10.9 -> 8.x ms)
        bool high = p->dim == 3 && p->order[0] >= 4;
        for (int d = 1; d < p->dim; ++d) high = high && p->order[d] == p->order[0];
        // (round 5: ALWAYS split -- with one channel and both gradients the fused LDS-tile kernel, pullbwd_tiled, was found to return wrong
        //  image gradients or to fault under rough fields once a workgroup serves several tiles (4 x 1 x 256^3 trilinear, sigma >= 4;
        //  96^3 at sigma = 4), and the split is as fast or faster now: trilinear 2.33 -> 1.94 ms at the identity, cubic 3.09 -> 3.08)
        if (grad_vol) {
            // gradient w.r.t. the image = push of grad_out (pushpull.py:252-253): the push kernels
            // (channel pairs per LDS atomic) beat the scatter half of the fused backward kernel
            // (4x2x256^3 cubic: 3.5 vs 5.3 ms; with the grid gradient 3.5 + 4.5 vs 8.6 ms fused)
            KParams kp = k;
            kp.vol_sb = gsb; kp.vol_sc = gsc;
            int64_t dense = 1;
            for (int d = p->dim - 1; d >= 0; --d) { kp.vol_ss[d] = (int)(dense * (int64_t)acc_esize(p->dtype)); dense *= p->vol_shape[d]; }
            rc = 0;
            if (high && grad_grid && scratch && p->dtype == INTERPOL_F32 && (p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER))) {
                // orders 4 - 7, both gradients (round 6): ONE binning of the samples for the image gradient (scatter5) and the grid gradient
                // (gather5<K, 1>) -- gather5.hip: try_backward5; 0: declined, the two halves go their own ways below
                rc = try_backward5(p, k, kp, grad_out, vol, grid, acc, grad_grid, scratch, scratch_bytes, st, &kp.gate);
                if (rc < 0 || rc > 2) return rc;
                grid_done = rc != 0;
                if (rc == 2) rc = 0;                                 // (the image gradient behind the probe: the tiles below read the same verdict)
            }
            if (rc == 0 && !grid_done && high && scratch && p->dtype == INTERPOL_F32 && (p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER))) {
                // orders 4 - 5 with the bricks' workspace: the image gradient through bricks of the target (gather5.hip: scatter5; the
                // grid gradient below reuses the workspace behind it on the stream)
                rc = try_scatter5(p, kp, grad_out, grid, acc, scratch, scratch_bytes, st, &kp.gate);
                if (rc < 0 || rc > 2) return rc;
                if (rc == 2) rc = 0;                                 // (behind the probe: the tiles below read the same verdict)
            }
            const bool behind_probe = kp.gate != nullptr;
            if (rc == 0) rc = try_fast_push(p, kp, grad_out, grid, acc, st);
            if (rc < 0 || rc > 1) return rc;
            if (rc == 0 && behind_probe) {
                // scatter5 is enqueued behind the probe and the tiles declined (a separable / affine lattice at these orders): the generic
                // push, which reads the same verdict, is the other organisation -- not the fused kernel below, which does not
                rc = launch_push_f32(kp, grad_out, grid, acc, B, st);
                if (rc) return rc;
                rc = 1;
            }
            vol_done = rc == 1;
            rc = (vol_done && (!grad_grid || grid_done)) ? 1 : 0;
        }
        if (rc == 0 && grad_grid && !grid_done && (vol_done || !grad_vol) && scratch && (p->dtype == INTERPOL_F32 || (p->dim == 2 && !grad_vol && p->dtype != INTERPOL_F64))) {
            // (a 16-bit image gradient uses `scratch` as its accumulator: the workspace only when there is none)
            rc = routed_gradc(p, k, grad_out, vol, grid, grad_grid, scratch, scratch_bytes, st);
            if (rc != 0 && rc != 1) return rc;
        }
        if (rc == 0) {
            rc = try_fast_pullbwd(p, k, grad_out, vol, grid, vol_done ? nullptr : acc, grad_grid, gsb, gsc, st);   // 1 = done, 0 = declined
            if (rc == 0 && vol_done) {
                // (the tiled grid-gradient kernel declined: the generic fused kernel below does the
                //  grid part only)
                rc = by_dtype(p->dtype,
                    [&] { return launch_pullbwd_f32(k, grad_out, vol, grid, nullptr, grad_grid, B, gsb, gsc, st); },
                    [&] { return launch_pullbwd_f64(k, grad_out, vol, grid, nullptr, grad_grid, B, gsb, gsc, st); },
                    [&] { return launch_pullbwd_bf16(k, grad_out, vol, grid, nullptr, grad_grid, B, gsb, gsc, st); },
                    [&] { return launch_pullbwd_f16(k, grad_out, vol, grid, nullptr, grad_grid, B, gsb, gsc, st); });
                rc = rc == 0 ? 1 : rc;
            }
        }
    }
    if (rc == 1) rc = 0;
    else if (rc == 0)
        rc = by_dtype(p->dtype,
            [&] { return launch_pullbwd_f32(k, grad_out, vol, grid, acc, grad_grid, B, gsb, gsc, st); },
            [&] { return launch_pullbwd_f64(k, grad_out, vol, grid, acc, grad_grid, B, gsb, gsc, st); },
            [&] { return launch_pullbwd_bf16(k, grad_out, vol, grid, acc, grad_grid, B, gsb, gsc, st); },
            [&] { return launch_pullbwd_f16(k, grad_out, vol, grid, acc, grad_grid, B, gsb, gsc, st); });
    if (rc) return rc;
    if (grad_vol && lowp)
        rc = p->dtype == INTERPOL_BF16 ? launch_narrow_bf16(acc, grad_vol, numel, st) : launch_narrow_f16(acc, grad_vol, numel, st);
    return rc;
}

int interpol_push_backward(const interpol_problem *p, const void *grad_vol_out, const void *val, const void *grid,
                           void *grad_val, void *grad_grid, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B);
    if (rc) return rc;
    if (!grad_vol_out || !val || !grid) return INTERPOL_E_NULL;
    if (!grad_val && !grad_grid) return 0;
    if (grad_grid && (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_AFFINE_GRID))) return INTERPOL_E_STRIDE;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        if (grad_val && !grad_grid) {
            // gradient w.r.t. the values alone = pull of grad_vol_out (pushpull.py:276-277)
            rc = try_fast_pull(p, k, grad_vol_out, grid, grad_val, st);
            if (rc != 0) return rc == 1 ? 0 : rc;
        }
        rc = try_fast_pushbwd(p, k, grad_vol_out, val, grid, grad_val, grad_grid, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    return by_dtype(p->dtype,
        [&] { return launch_pushbwd_f32(k, grad_vol_out, val, grid, grad_val, grad_grid, B, st); },
        [&] { return launch_pushbwd_f64(k, grad_vol_out, val, grid, grad_val, grad_grid, B, st); },
        [&] { return launch_pushbwd_bf16(k, grad_vol_out, val, grid, grad_val, grad_grid, B, st); },
        [&] { return launch_pushbwd_f16(k, grad_vol_out, val, grid, grad_val, grad_grid, B, st); });
}

// interpol_push_backward (val != NULL) / interpol_count_backward (val == NULL, grad_val == NULL) with the bricks workspace of the gathers
// they are made of (interpol_pull_workspace(p) bytes): the value gradient is the routed pull of grad_vol_out, the grid gradient
// the routed grid gradient with the two images' roles swapped (pushpull.py:276-281, 296-298).
int interpol_push_backward_ws(const interpol_problem *p, const void *grad_vol_out, const void *val, const void *grid,
                              void *grad_val, void *grad_grid, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!val && grad_val) return INTERPOL_E_NULL;
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B, val != nullptr);
    if (rc) return rc;
    if (!grad_vol_out || !grid) return INTERPOL_E_NULL;
    if (!grad_val && !grad_grid) return 0;
    if (grad_grid && (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_AFFINE_GRID))) return INTERPOL_E_STRIDE;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH) && (p->dtype == INTERPOL_F32 || (p->dim == 2 && p->dtype != INTERPOL_F64)) && workspace) {
        if (grad_grid && grad_val && val && p->dim == 3 && p->dtype == INTERPOL_F32 && k.order[0] >= 4
            && (p->flags & (INTERPOL_FLAG_BINNED_SCATTER | INTERPOL_FLAG_AUTO_SCATTER))) {
            // orders 4 - 7, both gradients (round 6): one binning of the samples for the pull of grad_vol_out and for its grid gradient (gather5.hip)
            rc = try_pushbwd5(p, k, grad_vol_out, val, grid, grad_val, grad_grid, workspace, workspace_bytes, st);
            if (rc != 0 && rc != 1) return rc;
            if (rc == 1) return 0;
        }
        if (grad_grid) {
            rc = routed_gradc(p, k, val, grad_vol_out, grid, grad_grid, workspace, workspace_bytes, st);
            if (rc != 0 && rc != 1) return rc;
            if (rc == 1) grad_grid = nullptr;
        }
        if (grad_val && !grad_grid) {                                // (the workspace is free again: the same stream)
            rc = routed_pull(p, k, B, grad_vol_out, grid, grad_val, workspace, workspace_bytes, st);
            if (rc != 0 && rc != 1) return rc;
            if (rc == 1) grad_val = nullptr;
        }
        if (!grad_val && !grad_grid) return 0;
    }
    return val ? interpol_push_backward(p, grad_vol_out, val, grid, grad_val, grad_grid, stream)
               : interpol_count_backward(p, grad_vol_out, grid, grad_grid, stream);
}

int interpol_count_backward(const interpol_problem *p, const void *grad_vol_out, const void *grid,
                            void *grad_grid, void *stream)
{
    KParams k; int B;
    int rc = make_params(p, GATHER, 1, &k, &B, false);
    if (rc) return rc;
    if (!grad_vol_out || !grid || !grad_grid) return INTERPOL_E_NULL;
    if (p->flags & (INTERPOL_FLAG_SEPARABLE_GRID | INTERPOL_FLAG_AFFINE_GRID)) return INTERPOL_E_STRIDE;
    hipStream_t st = (hipStream_t)stream;
    if (!(p->flags & INTERPOL_FLAG_NO_FASTPATH)) {
        rc = try_fast_pushbwd(p, k, grad_vol_out, nullptr, grid, nullptr, grad_grid, st);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    return by_dtype(p->dtype,
        [&] { return launch_pushbwd_f32(k, grad_vol_out, nullptr, grid, nullptr, grad_grid, B, st); },
        [&] { return launch_pushbwd_f64(k, grad_vol_out, nullptr, grid, nullptr, grad_grid, B, st); },
        [&] { return launch_pushbwd_bf16(k, grad_vol_out, nullptr, grid, nullptr, grad_grid, B, st); },
        [&] { return launch_pushbwd_f16(k, grad_vol_out, nullptr, grid, nullptr, grad_grid, B, st); });
}

int interpol_spline_filter(void *data, int32_t dtype, int64_t outer, int64_t n, int64_t inner,
                           int32_t bound, int32_t order, void *stream)
{
    return interpol_spline_filter_to(data, data, dtype, outer, n, inner, bound, order, stream);
}

int interpol_spline_filter_to(const void *src, void *data, int32_t dtype, int64_t outer, int64_t n, int64_t inner,
                              int32_t bound, int32_t order, void *stream)
{
    if (!data || !src) return INTERPOL_E_NULL;
    if (dtype < 0 || dtype > 3) return INTERPOL_E_DTYPE;
    if (order < 0 || order > 7) return INTERPOL_E_ORDER;
    if (bound < 0 || bound > 6) return INTERPOL_E_BOUND;
    if (outer < 0 || n < 0 || inner < 0) return INTERPOL_E_SHAPE;
    if (order >= 2 && (bound == 4 || bound == 5)) return INTERPOL_E_PREFILTER;   // coeff.py:243-244
    if (order < 2 || n <= 1 || outer == 0 || inner == 0) {    // coeff.py:306-307, 264-265: the identity
        if (src != data && outer * n * inner > 0) {
            const size_t esz = dtype == 0 ? 4 : (dtype == 1 ? 8 : 2);
            const hipError_t e = hipMemcpyAsync(data, src, esz * (size_t)(outer * n * inner), hipMemcpyDeviceToDevice, (hipStream_t)stream);
            return e == hipSuccess ? 0 : (int)e;
        }
        return 0;
    }
    FilterParams fp;
    fp.outer = outer; fp.n = n; fp.inner = inner;
    fp.bound = (bound == 0 || bound == 2) ? 0 : ((bound == 1 || bound == 3) ? 1 : 2);   // coeff.py:237-242
    // coeff.py:35-65 get_poles
    switch (order) {
    case 2: fp.npoles = 1; fp.pole[0] = sqrt(8.) - 3.; break;
    case 3: fp.npoles = 1; fp.pole[0] = sqrt(3.) - 2.; break;
    case 4: fp.npoles = 2;
        fp.pole[0] = sqrt(664. - sqrt(438976.)) + sqrt(304.) - 19.;
        fp.pole[1] = sqrt(664. + sqrt(438976.)) - sqrt(304.) - 19.; break;
    case 5: fp.npoles = 2;
        fp.pole[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
        fp.pole[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5; break;
    case 6: fp.npoles = 3;
        fp.pole[0] = -0.488294589303044755130118038883789062112279161239377608394;
        fp.pole[1] = -0.081679271076237512597937765737059080653379610398148178525368;
        fp.pole[2] = -0.00141415180832581775108724397655859252786416905534669851652709; break;
    default: fp.npoles = 3;
        fp.pole[0] = -0.5352804307964381655424037816816460718339231523426924148812;
        fp.pole[1] = -0.122554615192326690515272264359357343605486549427295558490763;
        fp.pole[2] = -0.0091486948096082769285930216516478534156925639545994482648003; break;
    }
    fp.gain = 1.;
    for (int i = 0; i < fp.npoles; ++i) fp.gain *= (1. - fp.pole[i]) * (1. - 1. / fp.pole[i]);   // coeff.py:69-73
    make_pole_pre(fp);
    return launch_filter(dtype, fp, src, data, (hipStream_t)stream);
}

int32_t interpol_resample_1d_gathers(int32_t dtype, int64_t n_samples, int64_t inner)
{
    return (dtype == INTERPOL_F32 || dtype == INTERPOL_F64) && resample1d_adjoint_gathers(n_samples, inner) ? 1 : 0;
}

int interpol_resample_1d(int32_t dtype, int32_t lin_dtype, int32_t order, int32_t bound, int32_t extrapolate, int32_t mode,
                         int32_t adjoint, int64_t outer, int64_t n_samples, int64_t n_lattice, int64_t inner,
                         const void *src, const void *lin, void *dst, void *stream)
{
    if (!src || !lin || !dst) return INTERPOL_E_NULL;
    if (dtype < 0 || dtype > 3 || (lin_dtype != INTERPOL_F32 && lin_dtype != INTERPOL_F64)) return INTERPOL_E_DTYPE;
    if (order < 0 || order > 7) return INTERPOL_E_ORDER;
    if (bound < 0 || bound > 6) return INTERPOL_E_BOUND;
    if (extrapolate < 0 || extrapolate > 2) return INTERPOL_E_EXTRAP;
    if (mode < 0 || mode > 2) return INTERPOL_E_SHAPE;
    if (outer < 0 || n_samples < 0 || n_lattice < 1 || inner < 0 || n_lattice > 0x3fffffff) return INTERPOL_E_SHAPE;
    const size_t es = esize(dtype);
    if ((uint64_t)n_lattice * (uint64_t)inner * es > 0xffffffffull) return INTERPOL_E_SHAPE;
    if ((uint64_t)n_samples * (uint64_t)inner > 0xffffffffull) return INTERPOL_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (adjoint && (!resample1d_adjoint_gathers(n_samples, inner) || outer == 0 || n_samples == 0 || inner == 0)) {
        const hipError_t e = zero_async(dst, (size_t)outer * (size_t)n_lattice * (size_t)inner * es, st);
        if (e != hipSuccess) return (int)e;
    }
    if (outer == 0 || n_samples == 0 || inner == 0) return 0;
    KParams k;
    memset(&k, 0, sizeof(k));
    k.dim = 1;
    k.extrapolate = extrapolate;
    k.mode = mode;
    k.bound[0] = bound; k.order[0] = order; k.vol_n[0] = (int)n_lattice; k.vol_ss[0] = (int)((uint64_t)inner * es);
    for (int d = 1; d < 3; ++d) { k.bound[d] = 1; k.vol_n[d] = 1; }
    k.C = 1;
    k.N = n_samples;
    k.mask_lo = -(extrapolate == 2 ? 0.5 + 5e-2 : 5e-2);
    k.mask_hi[0] = (double)(n_lattice - 1) + (extrapolate == 2 ? 0.5 + 5e-2 : 5e-2);
    k.mask_lo_f = (float)k.mask_lo; k.mask_hi_f[0] = (float)k.mask_hi[0];
    return launch_resample1d(dtype, lin_dtype == INTERPOL_F64, order, k, adjoint, src, lin, dst, (unsigned)n_samples, (unsigned)inner,
                             n_lattice, outer, st);
}

// ---- host-side scalar primitives (no GPU) ------------------------------------
int32_t interpol_host_bound_index(int32_t bound, int32_t i, int32_t n) { return wrap_index(bound, i, n); }
int32_t interpol_host_bound_sign(int32_t bound, int32_t i, int32_t n) { return wrap_sign_raw(bound, i, n); }
double interpol_host_weight(int32_t order, double x, int32_t which)
{
    return which == 0 ? bspline_w<double>(order, x) : (which == 1 ? bspline_g<double>(order, x) : bspline_h<double>(order, x));
}
float interpol_host_weight_f32(int32_t order, float x, int32_t which)
{
    return which == 0 ? bspline_w<float>(order, x) : (which == 1 ? bspline_g<float>(order, x) : bspline_h<float>(order, x));
}

int32_t interpol_set_handback(int32_t mode) { return defer_set_mode(mode); }
int32_t interpol_release_stream(void *stream) { return defer_release_stream((hipStream_t)stream); }
int32_t interpol_has_experiments(void)
{
#ifdef IP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

const char *interpol_kernel_name(const interpol_problem *p, const char *op)
{
    (void)op;
    if (!p) return "invalid";
    return "generic";
}

} // extern "C"
