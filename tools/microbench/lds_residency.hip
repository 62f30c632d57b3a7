// How many 256-thread workgroups with S bytes of dynamic LDS does a gfx950 CU keep resident?  Timing census: 2 x #CU
// workgroups that each spin for a fixed number of shader cycles - if two are resident per CU the launch takes one spin,
// otherwise two.  Also prints hipOccupancyMaxActiveBlocksPerMultiprocessor for the same kernel.
//   hipcc --offload-arch=gfx950 -O3 -o lds_residency lds_residency.hip && ./lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void spin(unsigned long long cycles, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    smem[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x + 1) % THREADS] == 123 && cycles == 7) sink[0] = 1;
}

template <int THREADS> static int probe(int lds, int ncu, int *sink)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&spin<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<THREADS>, THREADS, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms[3];
    for (int mult = 1; mult <= 3; ++mult) {
        spin<THREADS><<<ncu * mult, THREADS, lds>>>(200000ull, sink);  // warm
        CHECK(hipEventRecord(a));
        spin<THREADS><<<ncu * mult, THREADS, lds>>>(2000000ull, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms[mult - 1], a, b));
    }
    printf("threads %4d  LDS %6d B: occupancy API %d;  1x#CU %.2f ms, 2x#CU %.2f ms, 3x#CU %.2f ms  -> resident per CU ~ %s\n", THREADS, lds, occ,
           ms[0], ms[1], ms[2], ms[1] < 1.5f * ms[0] ? (ms[2] < 1.5f * ms[0] ? ">=3" : "2") : "1");
    return 0;
}

int main()
{
    int dev = 0, ncu = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    int *sink;
    CHECK(hipMalloc(&sink, 64));
    printf("CUs: %d\n", ncu);
    for (int lds : {16384, 32768, 49152, 65536, 66560, 73728, 76800, 79872, 81920, 98304, 129024, 163840})
        if (probe<256>(lds, ncu, sink)) return 1;
    for (int lds : {65536, 76800, 81920}) if (probe<512>(lds, ncu, sink)) return 1;
    return 0;
}
