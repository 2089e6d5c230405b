// Microbenchmark: 8-byte LDS reads / atomics under three address patterns (gfx950):
//   random   : every lane an independent random 8-byte slot
//   residue  : random ROWS (row pitch 32 slots = 256 B) but, inside every aligned group of 16
//              lanes, 16 distinct values of (slot mod 16)  -> what sorting a tile's samples by
//              (z0 mod 16) would give the tap loops of ops_tiled.hip
//   linear   : consecutive slots
// Build: hipcc --offload-arch=gfx950 -O3 lds_banks.hip -o lds_banks ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NT = 1024, ITER = 256, SLOTS = 16384;     // 128 KiB of 8-byte slots

template <int MODE, int PAT>
__global__ __launch_bounds__(NT) void k(float *out, unsigned seed)
{
    __shared__ unsigned long long lds[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += NT) lds[i] = 0ull;
    __syncthreads();
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    float acc = 0.f;
    const unsigned lane = threadIdx.x & 15;
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned slot;
        if (PAT == 0) slot = (s >> 8) & (SLOTS - 4);
        else if (PAT == 1) { const unsigned row = (s >> 8) & 511, half = (s >> 20) & 1; slot = (row * 32 + ((lane + it) & 15) + 16 * half) & (SLOTS - 4); }
        else slot = (threadIdx.x + it * 64) & (SLOTS - 4);
        if (MODE == 0) { const float2 v = reinterpret_cast<const float2 *>(lds)[slot]; acc += v.x + v.y; }                       // ds_read_b64
        else if (MODE == 1) { const float2 v = reinterpret_cast<const float2 *>(lds)[slot], w = reinterpret_cast<const float2 *>(lds)[slot + 1];
                              acc += v.x + v.y + w.x + w.y; }                                                                   // ds_read2_b64
        else atomicAdd(&lds[slot], 0x100000001ull);                                                                            // ds_add_u64
    }
    __syncthreads();
    if (acc == 12345.f || lds[threadIdx.x] == 77ull) out[0] = acc;
}

template <int MODE, int PAT>
void run(const char *name, float *d)
{
    const char *pat[3] = { "random", "residue", "linear" };
    const int blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, PAT><<<blocks, NT>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<MODE, PAT><<<blocks, NT>>>(d, r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double winst = (double)blocks * NT * ITER / 64;
    printf("%-14s %-8s %8.3f ms  %6.2f clk per wave instruction per CU (2.4 GHz x 256 CUs)\n", name, pat[PAT], ms,
           (ms * 1e-3) * 2.4e9 * 256 / winst);
}

int main()
{
    float *d; hipMalloc(&d, 4);
    run<0, 0>("ds_read_b64", d);  run<0, 1>("ds_read_b64", d);  run<0, 2>("ds_read_b64", d);
    run<1, 0>("ds_read2_b64", d); run<1, 1>("ds_read2_b64", d); run<1, 2>("ds_read2_b64", d);
    run<2, 0>("ds_add_u64", d);   run<2, 1>("ds_add_u64", d);   run<2, 2>("ds_add_u64", d);
    return 0;
}
