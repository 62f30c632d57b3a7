// Micro-benchmark: is a VALU write to an MFMA source operand, issued directly behind a burst of
// MFMAs, safe when two waves share a SIMD (matrix pipe contended)?  Each wave runs
//   repeat ITERS: { 8 x  acc += A * B  (same A, B);  B ^= TOGGLE  (VALU, no padding) }
// with small-integer bf16 operands, so the exact result is known: if any MFMA of a burst read the
// toggled B early, acc differs.  Reports mismatching lanes for 1 wave/SIMD (256 threads) and
// 2 waves/SIMD (512 threads).  Build: hipcc --offload-arch=gfx950 -O2 mfma_war.hip -o mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAD>
__global__ void k(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    // A[row r][k] = 1 for every k; B[k][col] = 1.0 (0x3F80) or, toggled, 2.0 (0x4000): xor 0x7F80 per half
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)1.0f; }
    u32x4 bb = __builtin_bit_cast(u32x4, b);
    f32x16 acc = {0};
    for (int it = 0; it < iters; ++it) {
        if (PAD == 0)
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t" "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t" "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t" "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t" "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_xor_b32 %3, 0x7f807f80, %3\n\t" "v_xor_b32 %4, 0x7f807f80, %4\n\t"
                "v_xor_b32 %5, 0x7f807f80, %5\n\t" "v_xor_b32 %6, 0x7f807f80, %6\n\t"
                : "+v"(acc) : "v"(a), "v"(bb), "v"(bb.x), "v"(bb.y), "v"(bb.z), "v"(bb.w));
        // the asm above cannot alias %2 with %3..%6 reliably; use the explicit form below instead
    }
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + i] = acc[i];
}

// explicit-register version: B lives in v[20:23], A in v[16:19], acc in a VGPR block chosen by the compiler
__global__ void k2(float *out, int iters, int nburst)
{
    f32x16 acc = {0};
    unsigned b0 = 0x3f803f80u, b1 = b0, b2 = b0, b3 = b0;       // 1.0 x8
    const unsigned a0 = 0x3f803f80u;                            // 1.0 x8
    for (int it = 0; it < iters; ++it) {
        asm volatile(
            "v_mov_b32 v16, %5\n\t v_mov_b32 v17, %5\n\t v_mov_b32 v18, %5\n\t v_mov_b32 v19, %5\n\t"
            "v_mov_b32 v20, %1\n\t v_mov_b32 v21, %2\n\t v_mov_b32 v22, %3\n\t v_mov_b32 v23, %4\n\t"
            "s_nop 4\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, v[16:19], v[20:23], %0\n\t"
            // WAR: overwrite the B operand registers immediately behind the burst (no wait states)
            "v_mov_b32 v20, 0\n\t v_mov_b32 v21, 0\n\t v_mov_b32 v22, 0\n\t v_mov_b32 v23, 0\n\t"
            "s_nop 7\n\t s_nop 7\n\t"
            : "+v"(acc) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(a0)
            : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
        b0 ^= 0x7f807f80u; b1 ^= 0x7f807f80u; b2 ^= 0x7f807f80u; b3 ^= 0x7f807f80u;   // 1.0 <-> 2.0
    }
    for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16 + i] = acc[i];
}

int main()
{
    const int iters = 2000;
    // each MFMA adds sum_k A*B = 16 * b; bursts alternate b = 1, 2 -> per pair of iterations 8*16*(1+2)
    const float expect = 8.0f * 16.0f * (1.0f + 2.0f) * (iters / 2);
    for (int threads : {256, 512, 1024}) {
        const int blocks = 256 * 4;
        float *d;
        hipMalloc(&d, (size_t)blocks * threads * 16 * sizeof(float));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k2, dim3(blocks), dim3(threads), 0, 0, d, iters, 8);
            hipDeviceSynchronize();
            std::vector<float> h((size_t)blocks * threads * 16);
            hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
            size_t bad = 0;
            float worst = expect;
            for (float v : h) if (v != expect) { ++bad; worst = v; }
            printf("threads/WG %4d (%d waves/SIMD when 1 WG/CU...): %zu of %zu accumulator values differ from %.0f%s\n", threads,
                   threads / 256, bad, h.size(), expect, bad ? " (e.g. " : "");
            if (bad) printf("     example value %.0f)\n", worst);
        }
        hipFree(d);
    }
    return 0;
}
