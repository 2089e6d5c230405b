// ===========================================================================
// defer.hip -- where the LDS-tiled kernels hand tiles back to the generic kernels.
//
// A tile kernel serves the samples of a tile whose stencils fall inside its LDS box; a few
// others per tile are gathered / scattered tap-parallel by a wave.  When more than ~3 % fall outside (the
// deformation stretches the tile beyond the box: a zoom factor >= 2, i.i.d. noise of sigma >~ 5
// voxels) the per-tile fallback is an order of magnitude slower than the plain generic kernel, so
// the tile kernel skips such a tile, writes its descriptor (stencil.hpp: TileList) here, and the
// generic kernel of the same operator -- launched right behind it on the same stream, persistent
// blocks over the descriptor list -- computes exactly those tiles.  A descriptor carries the number of
// the launch that wrote it and counts only in that launch: the tile kernels write nothing for the
// tiles they serve (a store per tile, waited for at the next barrier, cost 4 % of the headline pull)
// and the list needs no reset.
//
// The descriptor lists live in a static device array, one slot per (device, stream) that ever used
// the library: calls on one stream are ordered, so a slot has one writer / reader pair at a time.
// The only allocation of the library is 1 KiB of pinned host memory per device (the mode flags, see
// defer_buffer).  Streams beyond the slots, or problems beyond a slot's capacity, simply run
// without the hand-back (correct, slow on stretched tiles).
// ===========================================================================
#include "stencil.hpp"
#include <hip/hip_runtime.h>
#include <mutex>

namespace ip {

constexpr int DEFER_SLOTS = 16;
constexpr int DEFER_DEVICES = 64;
constexpr int64_t DEFER_CAP = 1 << 18;            // work items (tiles x batch items) per launch; 2 MiB per slot

__device__ unsigned long long g_defer[DEFER_SLOTS][DEFER_CAP];
__device__ unsigned g_defer_gen[DEFER_SLOTS][DEFER_CAP];   // launch stamps, zero-initialised; launch numbers start at 1

// The hand-back of one launch on stream `st`.  `hand_back_now` = 0 asks only for the flag (see below).
//
// A second kernel behind every tile kernel costs 25 - 35 us of kernel boundary even when it finds nothing to do (7 % of
// the 2-D pull of config 5).  So a stream starts in the PLAIN mode: desc == NULL, the tile kernels serve everything
// themselves -- but a tile worth handing back stores the launch number into the stream's flag, a word of pinned host
// memory.  The host reads that word (no synchronisation: a value that is a launch or two stale only delays the switch)
// at the next launch; when it has changed the stream enters the HAND-BACK mode, and leaves it again after 8 launches in
// a row without such a tile.  Results are the same in either mode.
DeferArgs defer_buffer(hipStream_t st, int64_t nwork, int64_t batch, int ntx, int nty, int ntz)
{
    const DeferArgs none = { nullptr, nullptr, nullptr, 0u };
    if (nwork <= 0 || nwork > DEFER_CAP || batch > (1 << 20) || ntx > (1 << 14) || nty > (1 << 14) || ntz > (1 << 14)) return none;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DEFER_DEVICES) return none;
    // a launch that is being captured into a graph would replay with the SAME launch number: no hand-back there
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return none; }
    if (cap != hipStreamCaptureStatusNone) return none;
    struct Slot { hipStream_t owner; unsigned launches, seen; int mode, quiet; };
    struct Device { unsigned long long *desc; unsigned *gen; volatile unsigned *hflag; unsigned *dflag; bool failed; int nused; Slot slot[DEFER_SLOTS]; };
    static std::mutex mu;
    static Device devs[DEFER_DEVICES];
    std::lock_guard<std::mutex> lock(mu);
    Device &D = devs[dev];
    if (D.failed) return none;
    if (!D.desc) {
        void *ptr = nullptr, *gptr = nullptr, *h = nullptr, *d = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_defer)) != hipSuccess || !ptr ||
            hipGetSymbolAddress(&gptr, HIP_SYMBOL(g_defer_gen)) != hipSuccess || !gptr ||
            hipHostMalloc(&h, sizeof(unsigned) * 16 * DEFER_SLOTS, hipHostMallocMapped) != hipSuccess || !h) {      // the flags: one 64-byte line per slot
            (void)hipGetLastError(); D.failed = true; return none;
        }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); D.failed = true; return none; }
        for (int i = 0; i < 16 * DEFER_SLOTS; ++i) ((unsigned *)h)[i] = 0u;
        D.desc = (unsigned long long *)ptr; D.gen = (unsigned *)gptr; D.hflag = (volatile unsigned *)h; D.dflag = (unsigned *)d;
    }
    int slot = -1;
    for (int i = 0; i < D.nused; ++i) if (D.slot[i].owner == st) { slot = i; break; }
    if (slot < 0) {
        if (D.nused >= DEFER_SLOTS) return none;
        slot = D.nused++;
        D.slot[slot] = Slot{ st, 0u, 0u, 0, 0 };
    }
    Slot &S = D.slot[slot];
    unsigned cur = ++S.launches;
    if (cur == 0) {                                 // 2^32 launches on one stream: wipe the stamps, start over
        if (hipMemsetAsync(D.gen + (int64_t)slot * DEFER_CAP, 0, sizeof(unsigned) * DEFER_CAP, st) != hipSuccess) return none;
        cur = S.launches = 1;
    }
    const unsigned now = D.hflag[16 * slot];
    if (now != S.seen) { S.seen = now; S.mode = 1; S.quiet = 0; }
    else if (S.mode && ++S.quiet > 8) S.mode = 0;
    DeferArgs a = { nullptr, nullptr, D.dflag + 16 * slot, cur };
    if (S.mode) { a.desc = D.desc + (int64_t)slot * DEFER_CAP; a.gen = D.gen + (int64_t)slot * DEFER_CAP; }
    return a;
}

} // namespace ip
