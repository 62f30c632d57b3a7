// Does FILLING the matrix pipe pay on a power-capped board?  (round 5)
// The LDS-ring convolution kernels keep the matrix pipe 78 % busy at the 1.59 GHz the board gives them; every experiment that raised the
// occupancy without removing work returned nothing.  Two explanations predict different things for a bare MFMA loop whose duty cycle
// is varied: (a) energy per MFMA is constant in this range (the voltage is at its floor): FLOP/s at the power cap does not depend on the
// duty cycle until the clock saturates; (b) energy per MFMA falls with the clock (V^2): a fuller pipe at a lower clock delivers MORE.
// One wave per SIMD (256 workgroups of 4 waves), v_mfma_f32_32x32x16_bf16 on four accumulators, operands in registers; after every
// group of four MFMAs (128 pipe cycles) the wave sleeps SLEEP x 64 cycles: duty = 128 / (128 + 64 SLEEP + a few).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_duty_power mfma_duty_power.hip && ./mfma_duty_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <string>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SLEEP, int WAVES, int DATA> __global__ __launch_bounds__(WAVES * 64) void loop(float *sink, int iters, unsigned long long *ticks)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a[2], b[2];
    for (int k = 0; k < 2; ++k)
        for (int e = 0; e < 8; ++e) {
            if (DATA == 0) {   // small, slowly varying constants (few toggling bits)
                a[k][e] = (__bf16)(0.001f * (float)((lane + k + e) & 15)); b[k][e] = (__bf16)(0.002f * (float)((lane * 3 + k + e) & 15));
            } else {           // what a convolution layer feeds the pipe: weights ~ +-U(0.06), activations = ReLU of +-U(1) (half of them zero)
                unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ (unsigned)((k * 8 + e) * 40503u) ^ (blockIdx.x * 97u);
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
                const float u = (float)(h & 0xffff) / 32768.0f - 1.0f, v = (float)(h >> 16) / 32768.0f - 1.0f;
                a[k][e] = (__bf16)(0.06f * u); b[k][e] = (__bf16)(v > 0.0f ? v : 0.0f);
            }
        }
    f32x16 acc[2][2] = {};
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}


// How many independent accumulation chains does ONE wave need to keep the pipe full?  (the ring kernels afford two)
template <int CHAINS> __global__ __launch_bounds__(256) void chains(float *sink, int iters)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (float)((lane + e) & 15)); b[e] = (__bf16)(0.002f * (float)((lane * 3 + e) & 15)); }
    f32x16 acc[CHAINS] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
    if (s == 123.456f) sink[0] = s;
}
template <int CHAINS> static int run_chains(float *sink)
{
    const int wgs = 256, iters = 600000 / CHAINS;
    chains<CHAINS><<<wgs, 256>>>(sink, 100);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < 8; ++k) chains<CHAINS><<<wgs, 256>>>(sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 8.0 * wgs * 4 * (double)iters * CHAINS * 32768.0 / (ms * 1e-3) / 1e12;
    printf("one wave per SIMD, %d accumulation chain(s), constant data: %8.1f TFLOP/s = %5.1f %% of 2.5 PFLOP/s\n", CHAINS, tf, tf / 25.0);
    return 0;
}

static std::string smi()
{
    std::string out;
    FILE *f = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Socket Graphics Package Power|sclk clock' | sed 's/.*: //' | tr '\\n' ' '", "r");
    if (!f) return out;
    char buf[256];
    while (fgets(buf, sizeof buf, f)) out += buf;
    pclose(f);
    return out;
}

template <int SLEEP, int WAVES, int DATA> static int run(float *sink, unsigned long long *ticks)
{
    const int wgs = 256, iters = SLEEP >= 4 ? 100000 : 200000;
    const double flop_per_launch = (double)wgs * WAVES * iters * 4 * 32768.0;
    loop<SLEEP, WAVES, DATA><<<wgs, WAVES * 64>>>(sink, 100, ticks);
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    std::string mid;
    double kernel_s = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 8; ++k) loop<SLEEP, WAVES, DATA><<<wgs, WAVES * 64>>>(sink, iters, ticks);
        CHECK(hipEventRecord(e1));
        launches += 8;
        if (mid.empty() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.8) mid = smi();
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); kernel_s += ms * 1e-3;
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    unsigned long long tk = 0;
    CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
    const double per_launch = kernel_s / launches, ghz = (double)tk / per_launch / 1e9;
    const double tf = flop_per_launch * launches / kernel_s / 1e12;
    (void)ghz;
    printf("%s data, waves/SIMD %d, sleep %2d: %8.1f TFLOP/s = %5.1f %% of 2.5 PFLOP/s   mid-run [sclk, package W]: %s\n",
           DATA ? "conv-like" : "constant ", WAVES / 4, SLEEP, tf, tf / 25.0, mid.c_str());
    return 0;
}

int main()
{
    float *sink; unsigned long long *ticks;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&ticks, 64));
    if (run_chains<1>(sink) || run_chains<2>(sink) || run_chains<3>(sink) || run_chains<4>(sink)) return 1;
    if (run<0, 4, 0>(sink, ticks)) return 1;
    if (run<2, 4, 0>(sink, ticks)) return 1;
    if (run<4, 4, 0>(sink, ticks)) return 1;
    if (run<8, 4, 0>(sink, ticks)) return 1;
    if (run<0, 4, 1>(sink, ticks)) return 1;
    if (run<1, 4, 1>(sink, ticks)) return 1;
    if (run<2, 4, 1>(sink, ticks)) return 1;
    if (run<4, 4, 1>(sink, ticks)) return 1;
    if (run<8, 4, 1>(sink, ticks)) return 1;
    if (run<0, 8, 1>(sink, ticks)) return 1;
    if (run<4, 8, 1>(sink, ticks)) return 1;
    return 0;
}
