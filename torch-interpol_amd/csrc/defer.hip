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
// STATE -- this file holds the only state of the library (include/interpol_hip.h, "State"):
//   * per device, on first use: 1 KiB of pinned host memory (the mode flags the tile kernels store into);
//   * per (device, stream) that entered the hand-back mode: one of 16 SLOTS = 3 MiB of device memory
//     (2^18 descriptors + stamps), allocated with hipMalloc when the slot first hands back, plus one event.
//     A 17th stream takes the least recently used slot (its first launch waits, on the device, for the
//     previous owner's last hand-back launch); interpol_release_stream() gives a slot back explicitly;
//     everything is freed at process exit (atexit).
//   * a slot is LEASED to one launch at a time: the lease (a mutex) is held from the moment the launch
//     number is drawn until the deferred kernel is enqueued, so two host threads that launch on the same
//     stream cannot interleave "tile kernel A, tile kernel B, deferred A, deferred B" (B's descriptors
//     would hide A's).
//   * the mode (interpol_set_handback / INTERPOL_HANDBACK): ADAPTIVE (default) decides per stream from the
//     flags its recent launches stored -- results then depend, in the last bits, on the history of the
//     stream (tiles and generic kernels sum in different orders); ALWAYS and NEVER do not: with either,
//     every operator is a deterministic function of its inputs.
// Launches that are being captured into a hipGraph never hand back (a replay would repeat the launch number).
// ===========================================================================
#include "defer.hpp"
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ip {

constexpr int DEFER_SLOTS = 16;
constexpr int DEFER_DEVICES = 64;
constexpr int64_t DEFER_CAP = 1 << 18;            // work items (tiles x batch items) per launch; 2 MiB + 1 MiB per slot

namespace {

struct Slot {
    hipStream_t owner = nullptr;
    bool used = false;
    unsigned launches = 0, seen = 0;
    int mode = 0, quiet = 0;
    unsigned long long tick = 0;                   // last use (LRU)
    unsigned long long *desc = nullptr;            // device: DEFER_CAP descriptors
    unsigned *gen = nullptr;                       // device: DEFER_CAP launch stamps (zero-initialised; launch numbers start at 1)
    hipEvent_t ev = nullptr;                       // recorded behind the last hand-back launch of the owner
    bool ev_valid = false;
    std::mutex lease;                              // held from defer_acquire to defer_release
};
struct Device {
    volatile unsigned *hflag = nullptr;            // pinned host memory, one 64-byte line per slot
    unsigned *dflag = nullptr;                     // the same, as the device sees it
    bool failed = false;
    unsigned long long clock = 0;
    Slot slot[DEFER_SLOTS];
};

std::mutex g_mu;                                   // guards the tables below (never held across a launch)
Device g_dev[DEFER_DEVICES];
std::atomic<int> g_mode{ -1 };                     // -1: not read from the environment yet

void defer_shutdown_fwd();

int current_mode()
{
    int m = g_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char *e = std::getenv("INTERPOL_HANDBACK");
    m = INTERPOL_HANDBACK_ADAPTIVE;
    if (e && (!std::strcmp(e, "always") || !std::strcmp(e, "1"))) m = INTERPOL_HANDBACK_ALWAYS;
    else if (e && (!std::strcmp(e, "never") || !std::strcmp(e, "0"))) m = INTERPOL_HANDBACK_NEVER;
    int expect = -1;
    g_mode.compare_exchange_strong(expect, m);
    return g_mode.load();
}

bool device_ready(Device &D)
{
    if (D.failed) return false;
    if (D.hflag) return true;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, sizeof(unsigned) * 16 * DEFER_SLOTS, hipHostMallocMapped) != hipSuccess || !h) { (void)hipGetLastError(); D.failed = true; return false; }
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); D.failed = true; return false; }
    for (int i = 0; i < 16 * DEFER_SLOTS; ++i) ((unsigned *)h)[i] = 0u;
    D.hflag = (volatile unsigned *)h; D.dflag = (unsigned *)d;
    static bool registered = false;
    if (!registered) { registered = true; std::atexit(defer_shutdown_fwd); }
    return true;
}

// descriptor list of a slot, allocated when the slot first hands back
bool slot_buffers(Slot &S, hipStream_t st)
{
    if (S.desc) return true;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, sizeof(unsigned long long) * DEFER_CAP) != hipSuccess || !a) { (void)hipGetLastError(); return false; }
    if (hipMalloc(&b, sizeof(unsigned) * DEFER_CAP) != hipSuccess || !b) { (void)hipGetLastError(); (void)hipFree(a); return false; }
    if (hipMemsetAsync(b, 0, sizeof(unsigned) * DEFER_CAP, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(a); (void)hipFree(b); return false; }
    S.desc = (unsigned long long *)a; S.gen = (unsigned *)b;
    return true;
}

} // namespace

// The hand-back of one launch on stream `st`; the returned lease must go to defer_release() once the deferred kernel
// (if any) is enqueued.
//
// A second kernel behind every tile kernel costs 25 - 35 us of kernel boundary even when it finds nothing to do (7 % of
// the 2-D pull of config 5).  So, in the ADAPTIVE mode, a stream starts PLAIN: desc == NULL, the tile kernels serve
// everything themselves -- but a tile worth handing back stores the launch number into the stream's flag, a word of pinned
// host memory.  The host reads that word (no synchronisation: a value that is a launch or two stale only delays the switch)
// at the next launch; when it has changed the stream hands back, and stops again after 8 launches in a row without such
// a tile.
DeferLease defer_acquire(hipStream_t st, int64_t nwork, int64_t batch, int ntx, int nty, int ntz)
{
    DeferLease none = { { nullptr, nullptr, nullptr, 0u }, -1, -1 };
    const int mode = current_mode();
    if (mode == INTERPOL_HANDBACK_NEVER) return none;
    if (nwork <= 0 || nwork > DEFER_CAP || batch > (1 << 20) || ntx > (1 << 14) || nty > (1 << 14) || ntz > (1 << 14)) return none;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DEFER_DEVICES) return none;
    // a launch that is being captured into a graph would replay with the SAME launch number: no hand-back there
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return none; }
    if (cap != hipStreamCaptureStatusNone) return none;
    Device &D = g_dev[dev];
    for (int attempt = 0; attempt < 64; ++attempt) {
        int slot = -1;
        bool fresh = false;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            if (!device_ready(D)) return none;
            for (int i = 0; i < DEFER_SLOTS; ++i) if (D.slot[i].used && D.slot[i].owner == st) { slot = i; break; }
            if (slot < 0) {
                // a free slot, else the least recently used one that no launch holds right now
                unsigned long long best = ~0ull;
                // (a free slot's lease can still be held by a launch that raced with interpol_release_stream: try_lock, never
                //  wait for a lease while holding g_mu -- the other paths take the lease first, g_mu second)
                for (int i = 0; i < DEFER_SLOTS; ++i) if (!D.slot[i].used && D.slot[i].lease.try_lock()) { slot = i; break; }
                if (slot < 0) {
                    for (int i = 0; i < DEFER_SLOTS; ++i) {
                        if (D.slot[i].tick < best && D.slot[i].lease.try_lock()) {
                            if (slot >= 0) D.slot[slot].lease.unlock();
                            slot = i; best = D.slot[i].tick;
                        }
                    }
                    if (slot < 0) return none;                    // every slot is in the middle of a launch
                }
                Slot &S = D.slot[slot];
                // the new owner's launches come behind the previous owner's last hand-back launch on the device
                if (S.ev_valid && hipStreamWaitEvent(st, S.ev, 0) != hipSuccess) { (void)hipGetLastError(); S.lease.unlock(); return none; }
                S.owner = st; S.used = true; S.mode = 0; S.quiet = 0; S.ev_valid = false;
                S.seen = D.hflag[16 * slot];
                fresh = true;
            }
        }
        Slot &S = D.slot[slot];
        if (!fresh) {
            S.lease.lock();                                       // another thread launching on this stream: wait for its launch pair
            std::lock_guard<std::mutex> lock(g_mu);
            if (!S.used || S.owner != st) { S.lease.unlock(); continue; }     // evicted meanwhile: start over
        }
        std::lock_guard<std::mutex> lock(g_mu);
        S.tick = ++D.clock;
        unsigned cur = ++S.launches;
        if (cur == 0) {                                           // 2^32 launches through one slot: wipe the stamps, start over
            if (S.gen && hipMemsetAsync(S.gen, 0, sizeof(unsigned) * DEFER_CAP, st) != hipSuccess) { (void)hipGetLastError(); S.lease.unlock(); return none; }
            cur = S.launches = 1;
        }
        if (mode == INTERPOL_HANDBACK_ALWAYS) {
            S.mode = 1;
        } else {
            const unsigned now = D.hflag[16 * slot];
            if (now != S.seen) { S.seen = now; S.mode = 1; S.quiet = 0; }
            else if (S.mode && ++S.quiet > 8) S.mode = 0;
        }
        DeferLease L = { { nullptr, nullptr, D.dflag + 16 * slot, cur }, dev, slot };
        if (S.mode) {
            if (slot_buffers(S, st)) { L.args.desc = S.desc; L.args.gen = S.gen; }
            else S.mode = 0;
        }
        return L;
    }
    return none;
}

void defer_release(const DeferLease &L, hipStream_t st)
{
    if (L.slot < 0) return;
    Slot &S = g_dev[L.dev].slot[L.slot];
    if (L.args.desc) {
        // behind the launch pair: what a later owner of the slot has to wait for
        if (!S.ev && hipEventCreateWithFlags(&S.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); S.ev = nullptr; }
        if (S.ev && hipEventRecord(S.ev, st) == hipSuccess) S.ev_valid = true; else (void)hipGetLastError();
    }
    S.lease.unlock();
}

int defer_set_mode(int mode)
{
    const int prev = current_mode();
    if (mode >= INTERPOL_HANDBACK_ADAPTIVE && mode <= INTERPOL_HANDBACK_NEVER) g_mode.store(mode);
    return prev;
}

// the slot of `st` on the current device goes back to the pool (its memory stays for the next owner)
int defer_release_stream(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DEFER_DEVICES) return 0;
    Device &D = g_dev[dev];
    for (;;) {
        Slot *S = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            for (int i = 0; i < DEFER_SLOTS; ++i) if (D.slot[i].used && D.slot[i].owner == st) { S = &D.slot[i]; break; }
            if (!S) return 0;
        }
        S->lease.lock();
        std::lock_guard<std::mutex> lock(g_mu);
        if (S->used && S->owner == st) { S->used = false; S->owner = nullptr; S->mode = 0; S->quiet = 0; S->lease.unlock(); return 1; }
        S->lease.unlock();
    }
}

// process exit: give everything back.  Registered with atexit() on first use -- i.e. AFTER the HIP runtime registered its own
// teardown, so this runs BEFORE it (a library destructor would run after: HIP calls from there crash).
static void defer_shutdown();
namespace { void defer_shutdown_fwd() { defer_shutdown(); } }
static void defer_shutdown()
{
    for (int d = 0; d < DEFER_DEVICES; ++d) {
        Device &D = g_dev[d];
        if (!D.hflag) continue;
        int cur = 0;
        const bool sw = hipGetDevice(&cur) == hipSuccess && hipSetDevice(d) == hipSuccess;
        for (int i = 0; i < DEFER_SLOTS; ++i) {
            Slot &S = D.slot[i];
            if (S.desc) (void)hipFree(S.desc);
            if (S.gen) (void)hipFree(S.gen);
            if (S.ev) (void)hipEventDestroy(S.ev);
            S.desc = nullptr; S.gen = nullptr; S.ev = nullptr;
        }
        (void)hipHostFree((void *)D.hflag);
        D.hflag = nullptr;
        if (sw) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    }
}

} // namespace ip
