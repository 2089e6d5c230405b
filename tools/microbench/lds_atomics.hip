// Microbenchmark: LDS atomic / read / write rates on gfx950 (lanes per CU-clock).
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NT = 1024, ITER = 256, WORDS = 32768;

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(NT) void k(float *out, unsigned seed)
{
    __shared__ float lds[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += NT) lds[i] = 0.f;
    __syncthreads();
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    float acc = 0.f;
    unsigned base = RANDOM ? 0 : threadIdx.x;
#pragma unroll 16
    for (int it = 0; it < ITER; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned idx = RANDOM ? ((s >> 8) & (WORDS - 2)) : ((base + it * 67) & (WORDS - 2));
        if (MODE == 0) atomicAdd(&lds[idx], 1.0f);                                   // ds_add_f32
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(lds) + idx, 1u);   // ds_add_u32
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(lds) + (idx >> 1), 1ull);  // ds_add_u64
        else if (MODE == 3) acc += lds[idx];                                          // ds_read_b32
        else if (MODE == 4) lds[idx] = (float)it;                                     // ds_write_b32
        else if (MODE == 5) { float v = lds[idx]; lds[idx] = v + 1.0f; }              // read-modify-write, non atomic
        else if (MODE == 6) acc += atomicAdd(&lds[idx], 1.0f);                        // ds_add_rtn_f32
        else if (MODE == 7) atomicMax(reinterpret_cast<int *>(lds) + idx, (int)it);  // ds_max_i32
    }
    __syncthreads();
    if (acc == 12345.f || lds[threadIdx.x] == -1.f) out[0] = acc;
}

template <int MODE, bool RANDOM>
void run(const char *name, float *d)
{
    const int blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, RANDOM><<<blocks, NT>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<MODE, RANDOM><<<blocks, NT>>>(d, r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double lanes = (double)blocks * NT * ITER;
    printf("%-28s %-7s %8.3f ms  %7.1f Glane/s  %6.2f lanes/clk/CU (2.4GHz x256)\n", name, RANDOM ? "random" : "linear",
           ms, lanes / ms / 1e6, lanes / (ms * 1e-3) / (256 * 2.4e9));
}

int main()
{
    float *d; hipMalloc(&d, 4);
    run<0, true>("ds_add_f32", d);  run<0, false>("ds_add_f32", d);
    run<1, true>("ds_add_u32", d);  run<1, false>("ds_add_u32", d);
    run<2, true>("ds_add_u64", d);  run<2, false>("ds_add_u64", d);
    run<3, true>("ds_read_b32", d); run<3, false>("ds_read_b32", d);
    run<4, true>("ds_write_b32", d); run<4, false>("ds_write_b32", d);
    run<5, true>("read+write (non atomic)", d); run<5, false>("read+write (non atomic)", d);
    run<6, true>("ds_add_rtn_f32", d); run<6, false>("ds_add_rtn_f32", d);
    run<7, true>("ds_max_i32", d); run<7, false>("ds_max_i32", d);
    return 0;
}
