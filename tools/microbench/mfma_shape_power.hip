// The step is power-managed, so what matters for the convolution kernels is FLOP per joule, not FLOP
// per cycle.  MI355X_MICROARCH.md reports that bare and LDS-fed v_mfma_f32_16x16x32_bf16 loops deliver 1.12-1.15 x the FLOP/s of
// v_mfma_f32_32x32x16_bf16 loops on a power-limited board (half the fp32 accumulator traffic per MAC).  This probe measures that
// on this pool with the conv kernels' operand diet: every MFMA's A and B fragments re-read from LDS (ds_read_b128, 1 KiB each per
// wave, conflict-free), two waves per SIMD on every CU, ~2 s per shape, with package power and sclk sampled (rocm-smi) mid-run.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_power mfma_shape_power.hip && ./mfma_shape_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 32x32x16, 4 accumulators (2 A x 2 B fragments per step: 4 MFMAs per 4 KiB of LDS reads)
// SHAPE 1: 16x16x32, 16 accumulators (4 A x 4 B fragments per step: 16 MFMAs per 8 KiB) - the same LDS bytes per MAC
// READS = false: operands stay in registers (bare loop)
template <int SHAPE, bool READS>
__global__ __launch_bounds__(512) void loop(float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 4; i += 512) reinterpret_cast<float *>(smem)[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    const char *base = smem + wave * 8192 + lane * 16;   // 8 fragments of 1 KiB per wave, lane-linear = conflict free
    const unsigned la = (unsigned)(wave * 8192 + lane * 16);   // byte address inside the dynamic LDS (it starts at 0)
    bf16x8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = *reinterpret_cast<const bf16x8 *>(base + k * 1024); b[k] = *reinterpret_cast<const bf16x8 *>(base + (4 + k) * 1024); }
    if (SHAPE == 0) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
            if (READS) {   // real LDS reads inside the loop (inline asm: a volatile C++ load becomes a FLAT load), waited for before the MFMAs
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(a[k]) : "v"(la), "i"(k * 1024));
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(b[k]) : "v"(la), "i"((4 + k) * 1024));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        float s = 0;
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0];
        if (s == 123.456f) sink[0] = s;
    } else if (SHAPE == 2) {
        // 16x16x32 with the conv kernels' software pipeline: the next step's 8 fragments are read (ds_read_b128, inline asm so that
        // they are real LDS reads inside the loop) BEFORE this step's 16 MFMAs; counted lgkmcnt wait in front of the MFMAs
        f32x4 acc[4][4] = {};
        bf16x8 a2[2][4], b2[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a2[0][k] = a[k]; b2[0][k] = b[k]; }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(a2[ph ^ 1][k]) : "v"(la), "i"(k * 1024));
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(b2[ph ^ 1][k]) : "v"(la), "i"((4 + k) * 1024));
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ph][m], b2[ph][n], acc[m][n], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads had 16 MFMAs (256 cycles) to land
            }
        }
        float s = 0;
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0];
        if (s == 123.456f) sink[0] = s;
    } else {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
            if (READS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(a[k]) : "v"(la), "i"(k * 1024));
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(b[k]) : "v"(la), "i"((4 + k) * 1024));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        float s = 0;
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0];
        if (s == 123.456f) sink[0] = s;
    }
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

template <int SHAPE, bool READS> static int run(const char *name, float *sink)
{
    const int iters = 400000, wgs = 256;   // one 8-wave workgroup per CU = two waves per SIMD; ~50-100 ms per launch
    const double flop_per_launch = (double)wgs * 8 * iters * (SHAPE == 0 ? 4 * 32768.0 : 16 * 16384.0);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&loop<SHAPE, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    loop<SHAPE, READS><<<wgs, 512, 64 * 1024>>>(sink, 100);
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    std::string mid;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
        for (int k = 0; k < 8; ++k) loop<SHAPE, READS><<<wgs, 512, 64 * 1024>>>(sink, iters);
        launches += 8;
        if (mid.empty() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.8) mid = smi();   // the queue is several launches deep
        CHECK(hipDeviceSynchronize());
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%-44s %8.1f TFLOP/s   mid-run [sclk, package power W]: %s\n", name, flop_per_launch * launches / el / 1e12, mid.c_str());
    return 0;
}

int main()
{
    float *sink;
    CHECK(hipMalloc(&sink, 64));
    if (run<0, false>("32x32x16 bf16, operands in registers", sink)) return 1;
    if (run<1, false>("16x16x32 bf16, operands in registers", sink)) return 1;
    if (run<0, true>("32x32x16 bf16, LDS-fed (1 KiB / MFMA), reads waited for", sink)) return 1;
    if (run<1, true>("16x16x32 bf16, LDS-fed (0.5 KiB / MFMA), reads waited for", sink)) return 1;
    if (run<2, true>("16x16x32 bf16, LDS-fed, reads one step ahead", sink)) return 1;
    return 0;
}
