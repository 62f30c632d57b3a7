// Root cause of round 2's two-stream corruption of the warp kernel, isolated:
//
//   a packed single-precision VALU operation whose operand select swaps the halves of src1
//       v_pk_add_f32 / v_pk_mul_f32  vD, vA, vB  op_sel:[0,1] op_sel_hi:[1,0]        (lo = A.lo op B.HI, hi = A.hi op B.LO)
//   reads B.HI as ZERO for the low-half result in lanes 48-63 - sometimes, and only while waves of a kernel on ANOTHER STREAM
//   execute MFMAs on the same SIMD (the packed f32 operations share the matrix pipe).  The high half is right; alone, beside
//   plain or packed VALU waves, or beside MFMA waves of the SAME launch (own workgroup / other workgroups of the grid) it never
//   happens; the mirrored select (op_sel:[1,0] op_sel_hi:[0,1], src0 swapped) never fails.
//
// How it was found: the failing kernel's device assembly was patched instruction by instruction and re-linked into the
// library (all 136 packed ops -> scalar: 0 of 100 forwards wrong; only ONE op left packed at a time: wrong only for the
// src1-swapped v_pk_add_f32; that op alone -> scalar, or halves pre-swapped by v_mov + plain v_pk_add_f32: 0 of 100; 16 wait
// states in front of it, a fresh destination, every s_waitcnt vmcnt -> 0: still 100 of 100).  hipcc's SLP vectoriser emits
// this form for   c0 = a*x0 + b*x1;  c1 = a*y0 + b*y1   style code.
//
// victim<OP, EXECMODE>: one packed operation per lane and iteration on fresh operands in [1, 2), both halves checked bit for bit
//             against v_add_f32 / v_mul_f32 / v_fma_f32; mismatches counted per (half, 16-lane group); the first few are
//             dumped with their operands.
// aggressor<KIND>: waves of another stream spinning on MFMAs (several shapes), on v_fma_f32 or on v_pk_fma_f32 (controls).
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32_beside_mfma pk_f32_beside_mfma.hip && ./pk_f32_beside_mfma [all]
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

enum { AG_NONE = -1, AG_MFMA32_CHAIN, AG_MFMA32_2ACC, AG_MFMA16_4ACC, AG_MFMA4, AG_VALU, AG_PKVALU, AG_COUNT };
static const char *kAggr[] = {"alone", "32x32x16 bf16, one chain", "32x32x16 bf16, 2 accumulators", "16x16x32 bf16, 4 accumulators", "4x4x4 f16, 4 accumulators",
                              "v_fma_f32 (control)", "v_pk_fma_f32 (control)"};

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float *sink, int rounds)
{
    f32x16 a32[2] = {{0}, {0}};
    f32x4 a16[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a, b;
    f16x4 ha, hb;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    for (int i = 0; i < 4; ++i) { ha[i] = (_Float16)(float)(threadIdx.x & 7); hb[i] = (_Float16)(float)(i + 1); }
    float v = (float)threadIdx.x;
    f32x2 pv = {v, v + 1.0f};
    for (int r = 0; r < rounds; ++r) {
        if (KIND == AG_MFMA32_CHAIN) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a32[0], 0, 0, 0);
        } else if (KIND == AG_MFMA32_2ACC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a32[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a32[k & 1], 0, 0, 0);
        } else if (KIND == AG_MFMA16_4ACC) {
#pragma unroll
            for (int k = 0; k < 32; ++k) a16[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a16[k & 3], 0, 0, 0);
        } else if (KIND == AG_MFMA4) {
#pragma unroll
            for (int k = 0; k < 64; ++k) a16[k & 3] = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, a16[k & 3], 0, 0, 0);
        } else if (KIND == AG_VALU) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(0.5f));
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv) : "v"(pv));
        }
    }
    if (a32[0][0] == 123.456f || a32[1][0] == 123.456f || a16[0][0] + a16[1][0] + a16[2][0] + a16[3][0] == 123.456f || v == 123.456f || pv[0] == 123.456f)
        sink[0] = a32[0][1] + v + pv[1];
}

__device__ __forceinline__ float mk(unsigned s) { return __uint_as_float(0x3f800000u | (s & 0x7fffffu)); }   // [1, 2)

enum { OP_ADD_SWAP1, OP_MUL_SWAP1, OP_FMA_SWAP1, OP_ADD_SWAP0, OP_ADD_LO_CROSS1, OP_ADD_HI_CROSS1, OP_MUL_LO_CROSS0, OP_MUL_HI_BCAST0, OP_FMA_BCAST0, OP_ADD, OP_MUL, OP_FMA,
       OP_FMA_F16_SWAP1, OP_MUL_F16_BCAST0LO, OP_FMA_F16_BCAST1HI, OP_FMA_F16_BCAST1LO, OP_COUNT };
static const char *kOp[] = {
    "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]          (src1 halves swapped)",
    "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]          (src1 halves swapped)",
    "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]      (src1 halves swapped)",
    "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]          (src0 halves swapped)",
    "v_pk_add_f32 op_sel:[0,1]                          (lo takes src1.hi)",
    "v_pk_add_f32 op_sel_hi:[1,0]                       (hi takes src1.lo)",
    "v_pk_mul_f32 op_sel:[1,0]                          (lo takes src0.hi)",
    "v_pk_mul_f32 op_sel_hi:[0,1]                       (hi takes src0.lo)",
    "v_pk_fma_f32 op_sel_hi:[0,1,1]                     (hi takes src0.lo)",
    "v_pk_add_f32",
    "v_pk_mul_f32",
    "v_pk_fma_f32",
    "v_pk_fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]      (src1 halves swapped, f16)",
    // the three forms deform_pack3.inl's blend emits (a corner weight broadcast to both channels of a dword)
    "v_pk_mul_f16 op_sel_hi:[0,1]                       (src0.lo to both halves, f16)",
    "v_pk_fma_f16 op_sel:[0,1,0]                        (src1.hi to both halves, f16)",
    "v_pk_fma_f16 op_sel_hi:[1,0,1]                     (src1.lo to both halves, f16)",
};

// EXECMODE: 0 all lanes, 1 a fresh random wave-uniform mask per iteration (random bits, a random subset of the four 16-lane
// groups switched off entirely), 2 lanes 48-63 only
template <int OP, int EXECMODE>
__global__ __launch_bounds__(256) void victim(unsigned long long *bad, int iters, float *samples, unsigned *nsamples)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned ws = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) * 40503u + 977u);
    unsigned long long local[8] = {0};
    for (int it = 0; it < iters; ++it) {
        if (EXECMODE == 2 && lane < 48) continue;
        if (EXECMODE == 1) {
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
#define PK2(INS, MODS) asm volatile(INS " %0, %1, %2 " MODS : "=v"(r) : "v"(x), "v"(y))
#define PK3(INS, MODS) asm volatile(INS " %0, %1, %2, %3 " MODS : "=v"(r) : "v"(x), "v"(y), "v"(z))
#define REF2(INS, E, A, B) asm volatile(INS " %0, %1, %2" : "=v"(E) : "v"(A), "v"(B))
#define REF3(E, A, B, C) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(E) : "v"(A), "v"(B), "v"(C))
        if (OP == OP_ADD_SWAP1) { PK2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]"); REF2("v_add_f32", e0, x0, y1); REF2("v_add_f32", e1, x1, y0); }
        else if (OP == OP_MUL_SWAP1) { PK2("v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]"); REF2("v_mul_f32", e0, x0, y1); REF2("v_mul_f32", e1, x1, y0); }
        else if (OP == OP_FMA_SWAP1) { PK3("v_pk_fma_f32", "op_sel:[0,1,0] op_sel_hi:[1,0,1]"); REF3(e0, x0, y1, z0); REF3(e1, x1, y0, z1); }
        else if (OP == OP_ADD_SWAP0) { PK2("v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[0,1]"); REF2("v_add_f32", e0, x1, y0); REF2("v_add_f32", e1, x0, y1); }
        else if (OP == OP_ADD_LO_CROSS1) { PK2("v_pk_add_f32", "op_sel:[0,1]"); REF2("v_add_f32", e0, x0, y1); REF2("v_add_f32", e1, x1, y1); }
        else if (OP == OP_ADD_HI_CROSS1) { PK2("v_pk_add_f32", "op_sel_hi:[1,0]"); REF2("v_add_f32", e0, x0, y0); REF2("v_add_f32", e1, x1, y0); }
        else if (OP == OP_MUL_LO_CROSS0) { PK2("v_pk_mul_f32", "op_sel:[1,0]"); REF2("v_mul_f32", e0, x1, y0); REF2("v_mul_f32", e1, x1, y1); }
        else if (OP == OP_MUL_HI_BCAST0) { PK2("v_pk_mul_f32", "op_sel_hi:[0,1]"); REF2("v_mul_f32", e0, x0, y0); REF2("v_mul_f32", e1, x0, y1); }
        else if (OP == OP_FMA_BCAST0) { PK3("v_pk_fma_f32", "op_sel_hi:[0,1,1]"); REF3(e0, x0, y0, z0); REF3(e1, x0, y1, z1); }
        else if (OP == OP_ADD) { PK2("v_pk_add_f32", ""); REF2("v_add_f32", e0, x0, y0); REF2("v_add_f32", e1, x1, y1); }
        else if (OP == OP_MUL) { PK2("v_pk_mul_f32", ""); REF2("v_mul_f32", e0, x0, y0); REF2("v_mul_f32", e1, x1, y1); }
        else if (OP == OP_FMA) { PK3("v_pk_fma_f32", ""); REF3(e0, x0, y0, z0); REF3(e1, x1, y1, z1); }
        else if (OP == OP_MUL_F16_BCAST0LO || OP == OP_FMA_F16_BCAST1HI || OP == OP_FMA_F16_BCAST1LO) {
            // f16 operands in [1, 2); expected values from the plain form on an explicitly splatted operand
            unsigned xi = ((__float_as_uint(x0) >> 13) & 0x03ff03ffu) | 0x3c003c00u, yi = ((__float_as_uint(y0) >> 13) & 0x03ff03ffu) | 0x3c003c00u,
                     zi = ((__float_as_uint(z0) >> 13) & 0x03ff03ffu) | 0x3c003c00u, ri, ei;
            if (OP == OP_MUL_F16_BCAST0LO) {
                const unsigned xs = (xi & 0xffffu) | (xi << 16);
                asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(ri) : "v"(xi), "v"(yi));
                asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(ei) : "v"(xs), "v"(yi));
            } else if (OP == OP_FMA_F16_BCAST1HI) {
                const unsigned ys = (yi >> 16) | (yi & 0xffff0000u);
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(ri) : "v"(xi), "v"(yi), "v"(zi));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(ei) : "v"(xi), "v"(ys), "v"(zi));
            } else {
                const unsigned ys = (yi & 0xffffu) | (yi << 16);
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(ri) : "v"(xi), "v"(yi), "v"(zi));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(ei) : "v"(xi), "v"(ys), "v"(zi));
            }
            r[0] = __uint_as_float(ri & 0xffffu); e0 = __uint_as_float(ei & 0xffffu);
            r[1] = __uint_as_float(ri >> 16); e1 = __uint_as_float(ei >> 16);
        }
        else {   // the 16-bit packed fma the pack kernel's blend uses (regular VALU pipe): one dword = two halves
            unsigned xi = ((__float_as_uint(x0) >> 13) & 0x03ff03ffu) | 0x3c003c00u, yi = ((__float_as_uint(y0) >> 13) & 0x03ff03ffu) | 0x3c003c00u,
                     zi = ((__float_as_uint(z0) >> 13) & 0x03ff03ffu) | 0x3c003c00u, ri, ys = (yi >> 16) | (yi << 16), ei;
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(ri) : "v"(xi), "v"(yi), "v"(zi));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(ei) : "v"(xi), "v"(ys), "v"(zi));
            r[0] = __uint_as_float(ri & 0xffffu); e0 = __uint_as_float(ei & 0xffffu);
            r[1] = __uint_as_float(ri >> 16); e1 = __uint_as_float(ei >> 16);
        }
        if (__float_as_uint(r[0]) != __float_as_uint(e0)) {
            local[lane >> 4]++;
            const unsigned k = atomicAdd(nsamples, 1u);
            if (k < 4) { float *o = samples + k * 8; o[0] = r[0]; o[1] = e0; o[2] = x0; o[3] = x1; o[4] = y0; o[5] = y1; o[6] = r[1]; o[7] = (float)lane; }
        }
        if (__float_as_uint(r[1]) != __float_as_uint(e1)) local[4 + (lane >> 4)]++;
    }
    for (int i = 0; i < 8; ++i)
        if (local[i]) atomicAdd(&bad[i], local[i]);
}

// SPLIT_BY_BLOCK = false: 512 threads = two waves per SIMD; waves 0-3 spin on MFMAs (16x16x32, 4 accumulators), waves 4-7 run
// the victim loop with the src1-swapped packed add: the MFMAs come from the victim's OWN workgroup.
// SPLIT_BY_BLOCK = true: 256-thread workgroups, even blocks spin on MFMAs, odd blocks run the victim loop - the MFMAs come
// from ANOTHER WORKGROUP OF THE SAME KERNEL LAUNCH (round 1's case: two workgroups of one kernel per CU).
template <bool SPLIT_BY_BLOCK>
__global__ __launch_bounds__(512) void same_launch(unsigned long long *bad, int iters, float *sink)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (SPLIT_BY_BLOCK ? (blockIdx.x & 1) == 0 : wave < 4) {
        f32x4 a16[4] = {{0}, {0}, {0}, {0}};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
        for (int r = 0; r < iters / 4; ++r)
#pragma unroll
            for (int k = 0; k < 32; ++k) a16[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a16[k & 3], 0, 0, 0);
        if (a16[0][0] + a16[1][0] + a16[2][0] + a16[3][0] == 123.456f) sink[0] = a16[0][1];
        return;
    }
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long local[8] = {0};
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u; const float x0 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float x1 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float y0 = mk(s >> 3);
        s = s * 1664525u + 1013904223u; const float y1 = mk(s >> 3);
        f32x2 x = {x0, x1}, y = {y0, y1}, r;
        float e0, e1;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(x0), "v"(y1));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(x1), "v"(y0));
        if (__float_as_uint(r[0]) != __float_as_uint(e0)) local[lane >> 4]++;
        if (__float_as_uint(r[1]) != __float_as_uint(e1)) local[4 + (lane >> 4)]++;
    }
    for (int i = 0; i < 8; ++i)
        if (local[i]) atomicAdd(&bad[i], local[i]);
}

struct Ctx { hipStream_t sa, sv; unsigned long long *dbad; float *sink, *samples; unsigned *nsamples; };

static int launch_aggressor(int kind, const Ctx &c)
{
    const int rounds = 600000;   // outlasts the victim launches (a flag in the output says if it did not)
    switch (kind) {
    case AG_MFMA32_CHAIN: aggressor<AG_MFMA32_CHAIN><<<256, 256, 0, c.sa>>>(c.sink, rounds); break;
    case AG_MFMA32_2ACC: aggressor<AG_MFMA32_2ACC><<<512, 256, 0, c.sa>>>(c.sink, rounds); break;      // two waves per SIMD
    case AG_MFMA16_4ACC: aggressor<AG_MFMA16_4ACC><<<512, 256, 0, c.sa>>>(c.sink, rounds); break;
    case AG_MFMA4: aggressor<AG_MFMA4><<<512, 256, 0, c.sa>>>(c.sink, rounds); break;
    case AG_VALU: aggressor<AG_VALU><<<256, 256, 0, c.sa>>>(c.sink, rounds); break;
    case AG_PKVALU: aggressor<AG_PKVALU><<<256, 256, 0, c.sa>>>(c.sink, rounds / 2); break;
    default: break;
    }
    return (int)hipGetLastError();
}

template <int OP, int EXECMODE> static int run_victim(int aggr, const Ctx &c)
{
    CHECK(hipMemset(c.dbad, 0, 8 * 8));
    CHECK(hipMemset(c.nsamples, 0, 4));
    CHECK(hipDeviceSynchronize());
    if (launch_aggressor(aggr, c)) return 1;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, c.sv));
    const int launches = 20, wgs = 1024, iters = 2000;
    for (int l = 0; l < launches; ++l) victim<OP, EXECMODE><<<wgs, 256, 0, c.sv>>>(c.dbad, iters, c.samples, c.nsamples);
    CHECK(hipEventRecord(e1, c.sv));
    CHECK(hipStreamSynchronize(c.sv));
    const bool still = aggr < 0 || hipStreamQuery(c.sa) == hipErrorNotReady;
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8];
    CHECK(hipMemcpy(h, c.dbad, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[i];
    static const char *em[] = {"all lanes", "random EXEC", "lanes 48-63"};
    printf("  %-84s %-11s %7.1f ms  wrong %8llu of %.1e   low half by lane group [%llu %llu %llu %llu]  high half [%llu %llu %llu %llu]%s\n", kOp[OP], em[EXECMODE], ms, tot,
           (double)launches * wgs * 256 * iters * 2 * (EXECMODE == 0 ? 1.0 : 0.25), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
           still ? "" : "   (aggressor ended early)");
    unsigned ns = 0; float hs[32];
    CHECK(hipMemcpy(&ns, c.nsamples, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hs, c.samples, sizeof hs, hipMemcpyDeviceToHost));
    for (unsigned k = 0; k < ns && k < 2; ++k)
        printf("        lane %2.0f: low half got %.9g, expected %.9g;  src0 = {%.9g, %.9g}  src1 = {%.9g, %.9g};  high half %.9g\n", hs[k * 8 + 7], hs[k * 8], hs[k * 8 + 1],
               hs[k * 8 + 2], hs[k * 8 + 3], hs[k * 8 + 4], hs[k * 8 + 5], hs[k * 8 + 6]);
    return 0;
}

int main(int argc, char **)
{
    Ctx c;
    CHECK(hipMalloc(&c.dbad, 64)); CHECK(hipMalloc(&c.sink, 64)); CHECK(hipMalloc(&c.samples, 32 * 4)); CHECK(hipMalloc(&c.nsamples, 4));
    CHECK(hipStreamCreateWithFlags(&c.sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&c.sv, hipStreamNonBlocking));
    for (int split = 0; split < 2; ++split) {   // one kernel launch, one stream
        CHECK(hipMemset(c.dbad, 0, 64));
        for (int l = 0; l < 10; ++l) {
            if (split) same_launch<true><<<2048, 256, 0, c.sv>>>(c.dbad, 4000, c.sink);
            else same_launch<false><<<512, 512, 0, c.sv>>>(c.dbad, 4000, c.sink);
        }
        CHECK(hipDeviceSynchronize());
        unsigned long long h[8];
        CHECK(hipMemcpy(h, c.dbad, sizeof h, hipMemcpyDeviceToHost));
        printf("%s, v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] beside a 16x16x32 MFMA spin, no second stream:\n"
               "  wrong %llu of %.1e   low half by lane group [%llu %llu %llu %llu]  high half [%llu %llu %llu %llu]\n",
               split ? "same launch, OTHER workgroups (even blocks MFMA, odd blocks victim)" : "same workgroup (waves 0-3 MFMA, waves 4-7 victim)",
               h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6] + h[7], split ? 10.0 * 1024 * 256 * 4000 * 2 : 10.0 * 512 * 256 * 4000 * 2, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    for (int aggr = AG_NONE; aggr < AG_COUNT; ++aggr) {
        printf("aggressor on the other stream: %s\n", kAggr[aggr + 1]);
        if (run_victim<OP_ADD_SWAP1, 0>(aggr, c) || run_victim<OP_ADD_SWAP1, 1>(aggr, c) || run_victim<OP_ADD_SWAP1, 2>(aggr, c)) return 1;
        if (run_victim<OP_MUL_SWAP1, 0>(aggr, c) || run_victim<OP_MUL_SWAP1, 2>(aggr, c)) return 1;
        if (run_victim<OP_FMA_SWAP1, 0>(aggr, c) || run_victim<OP_FMA_SWAP1, 2>(aggr, c)) return 1;
        if (run_victim<OP_ADD_SWAP0, 0>(aggr, c) || run_victim<OP_ADD_SWAP0, 2>(aggr, c)) return 1;
        if (run_victim<OP_ADD_LO_CROSS1, 0>(aggr, c) || run_victim<OP_ADD_LO_CROSS1, 2>(aggr, c)) return 1;
        if (run_victim<OP_ADD_HI_CROSS1, 0>(aggr, c) || run_victim<OP_ADD_HI_CROSS1, 2>(aggr, c)) return 1;
        if (run_victim<OP_MUL_LO_CROSS0, 0>(aggr, c) || run_victim<OP_MUL_HI_BCAST0, 0>(aggr, c) || run_victim<OP_FMA_BCAST0, 0>(aggr, c)) return 1;
        if (run_victim<OP_FMA_F16_SWAP1, 0>(aggr, c) || run_victim<OP_FMA_F16_SWAP1, 2>(aggr, c)) return 1;
        if (run_victim<OP_MUL_F16_BCAST0LO, 0>(aggr, c) || run_victim<OP_MUL_F16_BCAST0LO, 2>(aggr, c)) return 1;
        if (run_victim<OP_FMA_F16_BCAST1HI, 0>(aggr, c) || run_victim<OP_FMA_F16_BCAST1HI, 2>(aggr, c)) return 1;
        if (run_victim<OP_FMA_F16_BCAST1LO, 0>(aggr, c) || run_victim<OP_FMA_F16_BCAST1LO, 2>(aggr, c)) return 1;
        if (argc < 2) continue;
        if (run_victim<OP_ADD, 0>(aggr, c) || run_victim<OP_MUL, 0>(aggr, c) || run_victim<OP_FMA, 0>(aggr, c)) return 1;
        if (run_victim<OP_ADD, 1>(aggr, c) || run_victim<OP_MUL, 1>(aggr, c) || run_victim<OP_FMA, 1>(aggr, c)) return 1;
        if (run_victim<OP_MUL_LO_CROSS0, 1>(aggr, c) || run_victim<OP_MUL_HI_BCAST0, 1>(aggr, c) || run_victim<OP_FMA_BCAST0, 1>(aggr, c)) return 1;
    }
    return 0;
}
