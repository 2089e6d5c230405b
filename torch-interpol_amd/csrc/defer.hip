// ===========================================================================
// defer.hip -- where the LDS-tiled kernels hand tiles back to the generic kernels.
//
// A tile kernel serves the samples of a tile whose stencils fall inside its LDS box; a few
// others per tile are gathered / scattered tap-parallel by a wave.  When more than ~3 % fall outside (the
// deformation stretches the tile beyond the box: a zoom factor >= 2, i.i.d. noise of sigma >~ 5
// voxels) the per-tile fallback is an order of magnitude slower than the plain generic kernel, so
// the tile kernel skips such a tile, writes its descriptor (stencil.hpp: TileList) here, and the
// generic kernel of the same operator -- launched right behind it on the same stream, persistent
// blocks over the descriptor list -- computes exactly those tiles.  Every work item writes its
// descriptor (0: served), so the list needs no reset.
//
// The descriptor lists live in a static device array, one slot per (device, stream) that ever used
// the library: calls on one stream are ordered, so a slot has one writer / reader pair at a time.
// The library still allocates nothing.  Streams beyond the slots, or problems beyond a slot's
// capacity, simply run without the hand-back (correct, slow on rough deformations).
// ===========================================================================
#include "stencil.hpp"
#include <hip/hip_runtime.h>
#include <mutex>

namespace ip {

constexpr int DEFER_SLOTS = 16;
constexpr int DEFER_DEVICES = 64;
constexpr int64_t DEFER_CAP = 1 << 18;            // work items (tiles x batch items) per launch; 2 MiB per slot

__device__ unsigned long long g_defer[DEFER_SLOTS][DEFER_CAP];

// The descriptor list of this stream for a launch of `nwork` work items, or NULL (no hand-back).
unsigned long long *defer_buffer(hipStream_t st, int64_t nwork, int64_t batch, int ntx, int nty, int ntz)
{
    if (nwork <= 0 || nwork > DEFER_CAP || batch > (1 << 20) || ntx > (1 << 14) || nty > (1 << 14) || ntz > (1 << 14)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DEFER_DEVICES) return nullptr;
    static std::mutex mu;
    static unsigned long long *base[DEFER_DEVICES];
    static hipStream_t owner[DEFER_DEVICES][DEFER_SLOTS];
    static int nused[DEFER_DEVICES];
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev]) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_defer)) != hipSuccess || !ptr) { (void)hipGetLastError(); return nullptr; }
        base[dev] = (unsigned long long *)ptr;
    }
    int slot = -1;
    for (int i = 0; i < nused[dev]; ++i) if (owner[dev][i] == st) { slot = i; break; }
    if (slot < 0) {
        if (nused[dev] >= DEFER_SLOTS) return nullptr;
        slot = nused[dev]++;
        owner[dev][slot] = st;
    }
    return base[dev] + (int64_t)slot * DEFER_CAP;
}

} // namespace ip
