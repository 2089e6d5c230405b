// Microbenchmark (gfx950), round 6: ds_add_u64 x 64 at immediate offsets from a per-lane base slot (the tap block of own_accumulate).
// What does a wave instruction cost when the lanes' base slots are
//   classed32 : 32 distinct (slot mod 32) in each 32-lane half (what own_accumulate's class queues arrange)
//   classed16 : 16 distinct (slot mod 16) in each group of 16 CONTIGUOUS lanes (enough if atomics bank like ds_write_b64: 4 x 16 lanes, 32 banks of 4 B)
//   foreign15 : classed32 with 15 % of the lanes holding a record of a random class (the surplus records that fill the holes of other classes)
//   idle20    : classed32 with 20 % of the lanes switched off (queues padded to the fullest class instead)
//   random    : independent random slots
// and the same with the packed FMAs of the real tap block in front of every add (fma variants).
// Build: hipcc --offload-arch=gfx950 -O3 lds_add_classes.hip -o lds_add_classes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NT = 512, SLOTS = 6859 + 64, NSAMP = 64;
constexpr int PY = 19, PX = 361;
constexpr int SPAN = 3 * PX + 3 * PY + 4;
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hash(unsigned s) { s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16; return s; }

template <int PAT, int FMA>
__global__ __launch_bounds__(NT, 4) void k(float *out, unsigned seed)
{
    extern __shared__ unsigned long long lds[];
    for (int i = threadIdx.x; i < SLOTS; i += NT) lds[i] = 0ull;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const float w = 1.0f + 1e-9f * seed;
    for (int s = 0; s < NSAMP; ++s) {
        const unsigned h = hash(seed * 77u + threadIdx.x * 2654435761u + blockIdx.x * 40503u + s * 977u);
        const unsigned rows = (SLOTS - SPAN) / 32;
        unsigned base;
        bool on = true;
        if (PAT == 0) base = h % (SLOTS - SPAN);
        else if (PAT == 1) base = (h % rows) * 32 + (lane & 31);
        else if (PAT == 2) base = (h % (2 * rows)) * 16 + (lane & 15);
        else if (PAT == 3) { base = (h % rows) * 32 + (lane & 31); if ((h >> 20) % 100 < 15) base = (h % rows) * 32 + ((h >> 8) & 31); }
        else { base = (h % rows) * 32 + (lane & 31); on = (h >> 20) % 100 >= 20; }
        if (!on) continue;
        unsigned long long *p = lds + base;
        f2 sv = { w * (float)(h & 255), w * (float)((h >> 8) & 255) };
        const float tx = 0.3f + 1e-3f * (float)(h & 7), ty = 0.6f, tz = 0.2f;
        f2 wy[4] = { f2{tx, ty}, f2{ty, tz}, f2{tz, tx}, f2{tx * ty, tz} };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 sx = FMA ? sv * f2{ wy[i].x, wy[i].x } : sv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 sy = FMA ? sx * f2{ wy[j].y, wy[j].y } : sx;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    unsigned long long v;
                    if (FMA) { const f2 pr = __builtin_elementwise_fma(sy, f2{ wy[kk].x, wy[kk].x }, f2{ 12582912.f, 12582912.f }); v = __builtin_bit_cast(unsigned long long, pr); }
                    else v = 0x100000001ull + h;
                    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(p + i * PX + j * PY + kk);
                    asm volatile("ds_add_u64 %0, %1" :: "v"(addr), "v"(v) : "memory");
                }
            }
        }
    }
    __syncthreads();
    if (lds[threadIdx.x] == 77ull) out[0] = 1.f;
}

template <int PAT, int FMA>
void run(const char *name, float *d)
{
    const int blocks = 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<PAT, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PAT, FMA><<<blocks, NT, 78 * 1024>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<PAT, FMA><<<blocks, NT, 78 * 1024>>>(d, r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double winst = (double)blocks * NT * NSAMP * 64 / 64;
    printf("%-12s %-4s %8.3f ms  %6.2f clk per wave instruction slot per CU (2 workgroups of 512 per CU, 78 KiB each)\n", name, FMA ? "fma" : "", ms, (ms * 1e-3) * 2.4e9 * 256 / winst);
}
int main()
{
    float *d; hipMalloc(&d, 4);
    run<0, 0>("random", d); run<1, 0>("classed32", d); run<2, 0>("classed16", d); run<3, 0>("foreign15", d); run<4, 0>("idle20", d);
    run<0, 1>("random", d); run<1, 1>("classed32", d); run<2, 1>("classed16", d); run<3, 1>("foreign15", d); run<4, 1>("idle20", d);
    return 0;
}
