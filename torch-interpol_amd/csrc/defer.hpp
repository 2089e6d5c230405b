// ===========================================================================
// defer.hpp -- host side of the tile hand-back (defer.hip): the descriptor list of a stream and the
// generic kernels restricted to the handed-back tiles (ops_instantiate.inc, IP_DEFER).
// ===========================================================================
#pragma once
#include "../../include/interpol_hip.h"
#include "stencil.hpp"
#include <hip/hip_runtime.h>

namespace ip {

// The descriptor list of this stream for a launch of `nwork` work items (args.desc == NULL: no hand-back), stamped with a
// fresh launch number, and the LEASE of the stream's slot (slot >= 0): nobody else launches through the slot until
// defer_release() -- call it once the deferred kernel is enqueued (struct Defer below does, in its destructor).
struct DeferLease { DeferArgs args; int dev, slot; };
DeferLease defer_acquire(hipStream_t st, int64_t nwork, int64_t batch, int ntx, int nty, int ntz);
void defer_release(const DeferLease &lease, hipStream_t st);
int defer_set_mode(int mode);                  // INTERPOL_HANDBACK_*; returns the previous mode
int defer_release_stream(hipStream_t st);      // 1: the stream held a slot and gave it back

#define IP_DEFER_DECL(sfx) \
int launch_pull_deferred_##sfx(const KParams &p, const void *vol, const void *grid, void *val, const TileList &tl, hipStream_t st); \
int launch_grad_deferred_##sfx(const KParams &p, const void *vol, const void *grid, void *val, const TileList &tl, hipStream_t st); \
int launch_push_deferred_##sfx(const KParams &p, const void *val, const void *grid, void *acc, const TileList &tl, hipStream_t st); \
int launch_pullbwd_deferred_##sfx(const KParams &p, const void *gout, const void *vol, const void *grid, void *gvol, void *ggrid, \
                                  int64_t gsb, int64_t gsc, const TileList &tl, hipStream_t st); \
int launch_pushbwd_deferred_##sfx(const KParams &p, const void *gvol_out, const void *val, const void *grid, void *gval, void *ggrid, \
                                  const TileList &tl, hipStream_t st);
IP_DEFER_DECL(f32) IP_DEFER_DECL(bf16) IP_DEFER_DECL(f16)
#undef IP_DEFER_DECL

// The hand-back of one launch: the descriptor list (NULL: none) and the generic kernels that finish the job.
template <typename T> struct DeferOps;
#define IP_DEFER_OPS(T_, sfx) \
template <> struct DeferOps<T_> { \
    static int pull(const KParams &p, const void *vol, const void *grid, void *val, const TileList &tl, hipStream_t st) { return launch_pull_deferred_##sfx(p, vol, grid, val, tl, st); } \
    static int grad(const KParams &p, const void *vol, const void *grid, void *val, const TileList &tl, hipStream_t st) { return launch_grad_deferred_##sfx(p, vol, grid, val, tl, st); } \
    static int push1(const KParams &p, const void *val, const void *grid, void *acc, const TileList &tl, hipStream_t st) { return launch_push_deferred_##sfx(p, val, grid, acc, tl, st); } \
    static int pullbwd(const KParams &p, const void *gout, const void *vol, const void *grid, void *gvol, void *ggrid, int64_t gsb, int64_t gsc, const TileList &tl, hipStream_t st) \
    { return launch_pullbwd_deferred_##sfx(p, gout, vol, grid, gvol, ggrid, gsb, gsc, tl, st); } \
    static int pushbwd(const KParams &p, const void *gvol_out, const void *val, const void *grid, void *gval, void *ggrid, const TileList &tl, hipStream_t st) \
    { return launch_pushbwd_deferred_##sfx(p, gvol_out, val, grid, gval, ggrid, tl, st); } \
};
IP_DEFER_OPS(float, f32) IP_DEFER_OPS(bf16_t, bf16) IP_DEFER_OPS(f16_t, f16)
#undef IP_DEFER_OPS

struct Defer {
    unsigned long long *desc;
    DeferArgs args;
    TileList tl;
    DeferLease lease;
    hipStream_t stream;
    // nwork = ntiles * batch work items of tiles ex x ey x ez (kernel dims x, y, z: the LAST dims of the problem)
    Defer(const KParams &k, hipStream_t st, int64_t ntiles, int64_t batch, int ntx, int nty, int ntz, int ex, int ey, int ez) : stream(st)
    {
        // No hand-back behind a device-side router (k.gate: the probe-routed scatters, interpol_pull_ws, interpol_grad_ws, the routed
        // backward passes): what the tiles cannot hold goes to the bricks, decided from the coordinates of the call alone -- the
        // operator is a function of its inputs whatever the stream did before (round 5).  dbg 256: A/B switch, no hand-back.
        lease = ((k.dbg & 256) || k.gate || k.verdict) ? DeferLease{ { nullptr, nullptr, nullptr, 0u }, -1, -1 } : defer_acquire(st, ntiles * batch, batch, ntx, nty, ntz);
        args = lease.args;
        desc = args.desc;
        tl.desc = desc; tl.gen = args.gen; tl.cur = args.cur; tl.nwork = (int)(ntiles * batch); tl.e[0] = ex; tl.e[1] = ey; tl.e[2] = ez;
    }
    ~Defer() { defer_release(lease, stream); }
    Defer(const Defer &) = delete;
    Defer &operator=(const Defer &) = delete;
    // push / count / push + count (k.cc) of the handed-back tiles into the float accumulator `acc`; val == NULL: count
    template <typename T> int push(const KParams &k, const void *val, const void *grid, void *acc, hipStream_t st) const
    {
        if (!desc) return 0;
        KParams kk = k;
        kk.cc = 0;
        int rc = DeferOps<T>::push1(kk, val, grid, acc, tl, st);
        if (rc || !k.cc || !val) return rc;
        kk.C = 1;                                       // the count image: channel C of the target
        return DeferOps<T>::push1(kk, nullptr, grid, (char *)acc + (size_t)k.C * (size_t)k.vol_sc * 4, tl, st);
    }
    // grid gradient alone: sum_c gout_c d/dx pull(vol_c); gout == NULL: ones (backward of count)
    template <typename T> int gradc(const KParams &k, const void *gout, const void *vol, const void *grid, void *ggrid, hipStream_t st) const
    {
        if (!desc) return 0;
        if (gout) return DeferOps<T>::pullbwd(k, gout, vol, grid, nullptr, ggrid, 0, 0, tl, st);
        return DeferOps<T>::pushbwd(k, vol, nullptr, grid, nullptr, ggrid, tl, st);
    }
};

} // namespace ip
