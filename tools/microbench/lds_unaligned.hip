// Does gfx950 serve 4-byte-aligned ds_read_b128 / ds_read_b64 correctly, and how fast under random addresses?
// Build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NT = 1024, ITER = 256, WORDS = 32768;

template <int MODE>
__global__ __launch_bounds__(NT) void k(float *out, int *bad, unsigned seed)
{
    __shared__ float lds[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += NT) lds[i] = (float)i;
    __syncthreads();
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    float acc = 0.f; int nbad = 0;
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned idx = ((s >> 8) % (WORDS - 8)) | 1u;          // odd word index: never 8/16-byte aligned
        const unsigned addr = idx * 4u;
        float a, b, c, d;
        if (MODE == 0) {            // one 16-byte read at a 4-byte aligned address
            float4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            a = v.x; b = v.y; c = v.z; d = v.w;
        } else if (MODE == 1) {     // two ds_read2_b32
            a = lds[idx]; b = lds[idx + 1]; c = lds[idx + 2]; d = lds[idx + 3];
        } else {                    // two 8-byte reads at 4-byte aligned addresses
            float2 v, w;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(v), "=v"(w) : "v"(addr) : "memory");
            a = v.x; b = v.y; c = w.x; d = w.y;
        }
        nbad += (a != (float)idx) + (b != (float)(idx + 1)) + (c != (float)(idx + 2)) + (d != (float)(idx + 3));
        acc += a + b + c + d;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc == 12345.f) out[0] = acc;
}

template <int MODE> void run(const char *name, float *d, int *bad)
{
    const int blocks = 256 * 8;
    hipMemset(bad, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, NT>>>(d, bad, 1); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, NT>>>(d, bad, r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    int nb; hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-34s %8.3f ms  %6.2f clk per 16 bytes x 64 lanes per CU   wrong values: %d\n", name, ms,
           (ms * 1e-3) * 2.4e9 * 256 / ((double)blocks * NT * ITER / 64), nb);
}
int main()
{
    float *d; int *bad; hipMalloc(&d, 4); hipMalloc(&bad, 4);
    run<1>("2 x ds_read2_b32 (compiler)", d, bad);
    run<0>("ds_read_b128, 4-byte aligned", d, bad);
    run<2>("2 x ds_read_b64, 4-byte aligned", d, bad);
    return 0;
}
