// Microbenchmark: global (memory-side) atomic rates on gfx950, chip-wide lane-atomics per ms.
// Build: hipcc --offload-arch=gfx950 -O3 global_atomics.hip -o global_atomics.bin ; run on the GPU box.
//
// What the halo flush of the tiled scatters needs to know (VERDICT r2 #1a):
//   * lane-atomics per ms for global_atomic_add_f32 / _add_x2 (u64) / _add_f64 / _pk_add_bf16,
//   * coalesced (a half-wave = 32 contiguous floats, as the tile flush issues them) vs scattered lanes,
//   * with / without return, target footprint inside one L2 (2 MiB) vs the config-2 target (512 MiB),
//   * whether the "scope" of the atomic changes where it executes (the compiler emits the SAME instruction
//     for wavefront / workgroup / agent scope: only system scope adds sc1) and what the sc / nt bits do,
//   * the plain read-modify-write alternative (owner-computes flush: load + add + store, no atomic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int NT = 256;

enum Op { F32 = 0, U64, F64, PKBF16, F32_RET, F32_SC1, F32_NT, F32_SC0SC1, RMW_F32, RMW_F32X4, STORE_F32, U32, U64_SC1 };

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Each thread issues ITER operations.  PATTERN 0: "tile flush" -- a wave covers 64 contiguous elements of a row, rows of one
// workgroup are contiguous 256-element runs at pseudo-random row positions of the footprint (what a box flush looks like);
// PATTERN 1: every lane at its own pseudo-random element; PATTERN 2: streaming (block-linear, everything touched once).
template <int OP, int PATTERN>
__global__ __launch_bounds__(NT) void k(char *base, size_t nelem, int iter, unsigned seed, float *sink)
{
    const unsigned gid = blockIdx.x * NT + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iter; ++it) {
        size_t e;
        if (PATTERN == 0) {
            const size_t rows = nelem / NT;
            const size_t r = hash32(blockIdx.x * 9781u + it * 7919u + seed) % rows;
            e = r * NT + threadIdx.x;
        } else if (PATTERN == 1) {
            e = ((size_t)hash32(gid * 2654435761u + it * 40503u + seed) * 2654435761ull) % nelem;
        } else {
            e = ((size_t)it * gridDim.x * NT + gid) % nelem;
        }
        if (OP == F32) {
            float *p = (float *)base + e;
            asm volatile("global_atomic_add_f32 %0, %1, off" :: "v"(p), "v"(1.0f) : "memory");
        } else if (OP == F32_SC1) {
            float *p = (float *)base + e;
            asm volatile("global_atomic_add_f32 %0, %1, off sc1" :: "v"(p), "v"(1.0f) : "memory");
        } else if (OP == F32_NT) {
            float *p = (float *)base + e;
            asm volatile("global_atomic_add_f32 %0, %1, off nt" :: "v"(p), "v"(1.0f) : "memory");
        } else if (OP == F32_SC0SC1) {
            float *p = (float *)base + e; float r;
            asm volatile("global_atomic_add_f32 %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(1.0f) : "memory");
            acc += r;
        } else if (OP == F32_RET) {
            float *p = (float *)base + e; float r;
            asm volatile("global_atomic_add_f32 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(1.0f) : "memory");
            acc += r;
        } else if (OP == U32) {
            unsigned *p = (unsigned *)base + e;
            asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(1u) : "memory");
        } else if (OP == U64) {
            unsigned long long *p = (unsigned long long *)base + e;
            asm volatile("global_atomic_add_x2 %0, %1, off" :: "v"(p), "v"(0x100000001ull) : "memory");
        } else if (OP == U64_SC1) {
            unsigned long long *p = (unsigned long long *)base + e;
            asm volatile("global_atomic_add_x2 %0, %1, off sc1" :: "v"(p), "v"(0x100000001ull) : "memory");
        } else if (OP == F64) {
            double *p = (double *)base + e;
            asm volatile("global_atomic_add_f64 %0, %1, off" :: "v"(p), "v"(1.0) : "memory");
        } else if (OP == PKBF16) {
            unsigned *p = (unsigned *)base + e;
            asm volatile("global_atomic_pk_add_bf16 %0, %1, off" :: "v"(p), "v"(0x3f803f80u) : "memory");
        } else if (OP == RMW_F32) {
            float *p = (float *)base + e;
            *p = *p + 1.0f;
        } else if (OP == RMW_F32X4) {
            float4 *p = (float4 *)base + e;
            float4 v = *p; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *p = v;
        } else if (OP == STORE_F32) {
            float *p = (float *)base + e;
            *p = (float)it;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static const char *pat_name[] = { "rows256", "scattered", "stream" };

template <int OP, int PATTERN>
void run(const char *name, int esize, char *buf, size_t bytes, int blocks, int iter, float *sink)
{
    const size_t nelem = bytes / esize;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP, PATTERN><<<blocks, NT>>>(buf, nelem, 2, 99, sink);
    hipDeviceSynchronize();
    const int reps = 3;
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) k<OP, PATTERN><<<blocks, NT>>>(buf, nelem, iter, r, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double lanes = (double)blocks * NT * iter;
    printf("%-34s %-9s %4zu MiB  %8.3f ms  %7.3f Glane/ms  %7.1f GB/s payload\n", name, pat_name[PATTERN], bytes >> 20, ms,
           lanes / ms / 1e9, lanes * esize / ms / 1e6);
    fflush(stdout);
}

// correctness of mixed "scopes": every workgroup of the chip adds 1 to the same 4096 floats with the given instruction
template <int OP>
void check(const char *name, char *buf, float *sink)
{
    hipMemset(buf, 0, 4096 * 8);
    k<OP, 2><<<4096, NT>>>(buf, 4096, 16, 0, sink);
    hipDeviceSynchronize();
    float h[4]; hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost);
    printf("check %-28s expected %d per element, got %.0f %.0f %.0f\n", name, 4096 * NT * 16 / 4096, h[0], h[1], h[2]);
}

int main()
{
    const size_t big = 512ull << 20, small = 2ull << 20;
    char *buf; hipMalloc(&buf, big); hipMemset(buf, 0, big);
    float *sink; hipMalloc(&sink, 4);
    const int blocks = 256 * 16, iter = 64;          // 67 M lane operations per launch

    printf("# %d blocks x %d threads x %d ops\n", blocks, NT, iter);
    for (int pass = 0; pass < 2; ++pass) {
        const size_t bytes = pass ? small : big;
        run<F32, 0>("global_atomic_add_f32", 4, buf, bytes, blocks, iter, sink);
        run<F32, 1>("global_atomic_add_f32", 4, buf, bytes, blocks, iter, sink);
        run<F32, 2>("global_atomic_add_f32", 4, buf, bytes, blocks, iter, sink);
        run<U32, 0>("global_atomic_add (u32)", 4, buf, bytes, blocks, iter, sink);
        run<U64, 0>("global_atomic_add_x2 (u64)", 8, buf, bytes, blocks, iter, sink);
        run<U64, 1>("global_atomic_add_x2 (u64)", 8, buf, bytes, blocks, iter, sink);
        run<U64, 2>("global_atomic_add_x2 (u64)", 8, buf, bytes, blocks, iter, sink);
        run<F64, 0>("global_atomic_add_f64", 8, buf, bytes, blocks, iter, sink);
        run<F64, 1>("global_atomic_add_f64", 8, buf, bytes, blocks, iter, sink);
        run<PKBF16, 0>("global_atomic_pk_add_bf16", 4, buf, bytes, blocks, iter, sink);
        run<F32_SC1, 0>("global_atomic_add_f32 sc1", 4, buf, bytes, blocks, iter, sink);
        run<F32_NT, 0>("global_atomic_add_f32 nt", 4, buf, bytes, blocks, iter, sink);
        run<U64_SC1, 0>("global_atomic_add_x2 sc1", 8, buf, bytes, blocks, iter, sink);
        run<F32_RET, 0>("global_atomic_add_f32 sc0 (rtn)", 4, buf, bytes, blocks, 16, sink);
        run<F32_SC0SC1, 0>("global_atomic_add_f32 sc0 sc1 (rtn)", 4, buf, bytes, blocks, 16, sink);
        run<RMW_F32, 0>("load+add+store f32 (no atomic)", 4, buf, bytes, blocks, iter, sink);
        run<RMW_F32, 2>("load+add+store f32 (no atomic)", 4, buf, bytes, blocks, iter, sink);
        run<RMW_F32X4, 0>("load+add+store f32x4 (no atomic)", 16, buf, bytes, blocks, iter, sink);
        run<RMW_F32X4, 2>("load+add+store f32x4 (no atomic)", 16, buf, bytes, blocks, iter, sink);
        run<STORE_F32, 0>("store f32", 4, buf, bytes, blocks, iter, sink);
        run<STORE_F32, 2>("store f32", 4, buf, bytes, blocks, iter, sink);
    }
    check<F32>("add_f32", buf, sink);
    check<F32_SC1>("add_f32 sc1", buf, sink);
    check<F32_NT>("add_f32 nt", buf, sink);
    return 0;
}
