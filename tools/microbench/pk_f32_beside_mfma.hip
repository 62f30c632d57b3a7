// Found in round 2: the tiled warp kernel returned wrong pixels - always in lanes 48-63 of a wave,
// always the LOW half of a v_pk_mul_f32 result - but only while ANOTHER kernel with MFMAs (a rocBLAS GEMM was enough) shared
// the CUs from a second stream; the same source built with -fno-slp-vectorize (no packed f32 VALU ops) never failed.
// This probe takes the warp kernel out of the picture: are the packed single-precision VALU operations the compiler's SLP
// vectoriser emits (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, with and without op_sel) exact when waves of another
// kernel run MFMAs on the same SIMDs?
//
// victim<OP>: every lane computes one packed operation per iteration on lane- and iteration-dependent operands and checks
//             both halves, bit for bit, against the same arithmetic done with scalar v_mul_f32 / v_add_f32 / v_fma_f32;
//             mismatches are counted per (half, 16-lane group).
// aggressor:  one wave per SIMD on every CU spinning on v_mfma_f32_32x32x16_bf16 (or, as a control, on v_fma_f32) until the
//             host raises a flag.
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32_beside_mfma pk_f32_beside_mfma.hip && ./pk_f32_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: MFMA, 1: plain VALU fma (control), 2: packed f32 VALU
__global__ __launch_bounds__(256) void aggressor(float *sink, int rounds)
{
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    float v = (float)threadIdx.x;
    f32x2 pv = {v, v + 1.0f};
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(0.5f));
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv) : "v"(pv));
        }
    }
    if (acc[0] == 123.456f || v == 123.456f || pv[0] == 123.456f) sink[0] = acc[1] + v + pv[1];
}

__device__ __forceinline__ float mk(unsigned s) { return __uint_as_float(0x3f800000u | (s & 0x7fffffu)); }   // [1, 2)

// OP: 0 pk_mul op_sel:[1,0]   1 pk_mul   2 pk_add   3 pk_fma op_sel_hi:[0,1,1]   4 pk_fma
// PARTIAL: every iteration runs under a fresh wave-uniform EXEC mask = random 64 bits with a random subset of the four
// 16-lane groups switched off entirely (divergent code is where the warp kernel's packed operations sit).
template <int OP, bool PARTIAL>
__global__ __launch_bounds__(256) void victim(unsigned long long *bad, int iters)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned ws = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * 40503u + 977u);
    unsigned long long local[8] = {0};
    for (int it = 0; it < iters; ++it) {
        if (PARTIAL) {
            ws = ws * 1664525u + 1013904223u; const unsigned lo = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned hi = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned g = ws >> 28;
            unsigned long long m = ((unsigned long long)hi << 32) | lo;
            for (int q = 0; q < 4; ++q) if (!(g >> q & 1)) m &= ~(0xffffull << (16 * q));
            if (!(m >> lane & 1)) continue;
        }
        s = s * 1664525u + 1013904223u; const float x0 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float x1 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float y0 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float y1 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float z0 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float z1 = mk(s >> 3);
        f32x2 x = {x0, x1}, y = {y0, y1}, z = {z0, z1}, r;
        float e0, e1;
        if (OP == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x1), "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x1), "v"(y1));
        } else if (OP == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x0), "v"(y0));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x1), "v"(y1));
        } else if (OP == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(x0), "v"(y0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(x1), "v"(y1));
        } else if (OP == 3) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x0), "v"(y0), "v"(z0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x0), "v"(y1), "v"(z1));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x0), "v"(y0), "v"(z0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x1), "v"(y1), "v"(z1));
        }
        if (__float_as_uint(r[0]) != __float_as_uint(e0)) local[lane >> 4]++;
        if (__float_as_uint(r[1]) != __float_as_uint(e1)) local[4 + (lane >> 4)]++;
    }
    for (int i = 0; i < 8; ++i)
        if (local[i]) atomicAdd(&bad[i], local[i]);
}

template <int OP, bool PARTIAL> static int run_victim(const char *name, int aggr, hipStream_t sa, hipStream_t sv, unsigned long long *dbad, float *sink)
{
    CHECK(hipMemset(dbad, 0, 8 * 8));
    CHECK(hipDeviceSynchronize());
    const int rounds = 600000;   // long enough to outlast the victim launches (tens of milliseconds)
    if (aggr == 0) aggressor<0><<<256, 256, 0, sa>>>(sink, rounds);
    if (aggr == 1) aggressor<1><<<256, 256, 0, sa>>>(sink, rounds);
    if (aggr == 2) aggressor<2><<<256, 256, 0, sa>>>(sink, rounds / 2);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, sv));
    for (int l = 0; l < 20; ++l) victim<OP, PARTIAL><<<1024, 256, 0, sv>>>(dbad, 2000);
    CHECK(hipEventRecord(e1, sv));
    CHECK(hipStreamSynchronize(sv));
    const bool aggressor_still_running = aggr < 0 || hipStreamQuery(sa) == hipErrorNotReady;
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8];
    CHECK(hipMemcpy(h, dbad, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[i];
    static const char *an[] = {"alone", "beside MFMA waves", "beside v_fma_f32 waves", "beside v_pk_fma_f32 waves"};
    printf("%-40s %-26s %8.1f ms  wrong results %8llu of %.2e   low half by lane group [%llu %llu %llu %llu]  high half [%llu %llu %llu %llu]%s\n",
           name, an[aggr + 1], ms, tot, 20.0 * 1024 * 256 * 2000 * 2, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
           aggressor_still_running ? "" : "   (aggressor ended early)");
    return 0;
}

int main()
{
    unsigned long long *dbad; float *sink;
    CHECK(hipMalloc(&dbad, 64)); CHECK(hipMalloc(&sink, 64));
    hipStream_t sa, sv;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    for (int aggr = -1; aggr <= 2; ++aggr) {
        if (run_victim<0, false>("v_pk_mul_f32 op_sel:[1,0]", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<1, false>("v_pk_mul_f32", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<2, false>("v_pk_add_f32", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<3, false>("v_pk_fma_f32 op_sel_hi:[0,1,1]", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<4, false>("v_pk_fma_f32", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<0, true>("v_pk_mul_f32 op_sel:[1,0] partial EXEC", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<1, true>("v_pk_mul_f32 partial EXEC", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<2, true>("v_pk_add_f32 partial EXEC", aggr, sa, sv, dbad, sink)) return 1;
        if (run_victim<4, true>("v_pk_fma_f32 partial EXEC", aggr, sa, sv, dbad, sink)) return 1;
    }
    return 0;
}
