// Issue-rate / latency probe for the vector instructions the deformable-conv blend can be built from (gfx950).
// For every pattern: 64 instructions per loop trip on 8 independent destination registers (throughput) or on ONE
// register (dependent chain = latency), 1 / 2 / 4 waves per SIMD, every CU busy.  Reports shader cycles (s_memtime)
// per wave-instruction as seen by ONE wave (its own issue interval) and per SIMD (interval / waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { P_FMA, P_PKFMA_F32, P_PKFMA_F16, P_DOT2_BF16, P_DOT2C_BF16, P_DOT2_F16, P_PERM, P_CVTPK, P_FMAMIX, P_AND, P_PKMUL_F32,
       P_MFMA32, P_MFMA16, P_MFMA32_DOT2, P_MFMA32_FMA, P_MFMA32_PERM, P_MFMA32_PKF16, P_MFMA32X8, P_DSREAD128, NPAT };
static const char *kNames[NPAT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f16", "v_dot2_f32_bf16 (vop3p)", "v_dot2c_f32_bf16 (vop2)",
                                   "v_dot2_f32_f16", "v_perm_b32", "v_cvt_pk_bf16_f32", "v_fma_mix_f32", "v_and_b32", "v_pk_mul_f32",
                                   "mfma_32x32x16_bf16 alone", "mfma_16x16x32_bf16 alone", "mfma32 + 8 dot2 per gap", "mfma32 + 8 fma per gap",
                                   "mfma32 + 8 perm per gap", "mfma32 + 8 pk_fma_f16 per gap", "mfma_32x32x8_bf16_1k alone", "ds_read_b128 (conflict-free)"};

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define REP8(S) S S S S S S S S
// 8 independent destinations d0..d7
#define IND8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int PAT, bool DEP>
__global__ __launch_bounds__(1024) void probe(unsigned long long *cyc, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    float d[8], a = threadIdx.x * 0.001f + 1.0f, b = 0.5f, c = 0.25f;
    unsigned ua = threadIdx.x * 2654435761u, ub = 0x3f803f80u;
    for (int i = 0; i < 8; ++i) d[i] = i;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    f32x16 acc = {0}, acc2 = {0};
    f32x4 acc4 = {0};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x + i); fb[i] = (__bf16)(float)(i + 1); }
    bf16x4 ga = {fa[0], fa[1], fa[2], fa[3]}, gb = {fb[0], fb[1], fb[2], fb[3]};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#define D(i) (DEP ? d[0] : d[i])
        if constexpr (PAT == P_FMA) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(DEP ? d[0] : d[i]) : "v"(a), "v"(b));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_PKFMA_F32) {
            // 2 f32 per instruction: register pairs
            double *dd = reinterpret_cast<double *>(d);
            double pa = __hiloint2double(__float_as_int(a), __float_as_int(b)), pb = pa;
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dd[DEP ? 0 : (i & 3)]) : "v"(pa), "v"(pb));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_PKFMA_F16) {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(ub));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_DOT2_BF16) {
#define OP(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(ub));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_DOT2C_BF16) {
#define OP(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(ub));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_DOT2_F16) {
#define OP(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(ub));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_PERM) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(ub));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_CVTPK) {
#define OP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(DEP ? d[0] : d[i]) : "v"(a));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_FMAMIX) {
#define OP(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(DEP ? d[0] : d[i]) : "v"(ua), "v"(b));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_AND) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(DEP ? d[0] : d[i]) : "v"(ua));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_PKMUL_F32) {
            double *dd = reinterpret_cast<double *>(d);
            double pa = __hiloint2double(__float_as_int(a), __float_as_int(b));
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(dd[DEP ? 0 : (i & 3)]) : "v"(pa));
            REP8(IND8(OP))
#undef OP
        } else if constexpr (PAT == P_MFMA32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // 8 MFMAs per trip (counted as 8 instructions)
                if (DEP || (k & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
            }
        } else if constexpr (PAT == P_MFMA16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4, 0, 0, 0);
        } else if constexpr (PAT == P_MFMA32X8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (DEP || (k & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, ga), __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, gb), acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, ga), __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, gb), acc2, 0, 0, 0);
            }
        } else if constexpr (PAT == P_MFMA32_DOT2 || PAT == P_MFMA32_FMA || PAT == P_MFMA32_PERM || PAT == P_MFMA32_PKF16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // 8 x (1 MFMA + 8 fillers)
                if ((k & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
                if constexpr (PAT == P_MFMA32_DOT2) {
#define OP(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(ua), "v"(ub));
                    IND8(OP)
#undef OP
                } else if constexpr (PAT == P_MFMA32_FMA) {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));
                    IND8(OP)
#undef OP
                } else if constexpr (PAT == P_MFMA32_PERM) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(ua), "v"(ub));
                    IND8(OP)
#undef OP
                } else {
#define OP(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(ua), "v"(ub));
                    IND8(OP)
#undef OP
                }
            }
        } else if constexpr (PAT == P_DSREAD128) {
            f32x4 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const f32x4 *>(&lds[((threadIdx.x & 63) * 4 + k * 256) & 4095]);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] += r[k][0];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += d[i];
    s += acc[0] + acc2[3] + acc4[1] + c;
    if (s == 12345.678f) sink[0] = s;  // keep everything live
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int PAT, bool DEP> static int run(int waves_per_simd, int iters, unsigned long long *dcyc, float *sink)
{
    const int threads = 256 * waves_per_simd, blocks = 256;
    probe<PAT, DEP><<<blocks, threads>>>(dcyc, sink, 10);
    probe<PAT, DEP><<<blocks, threads>>>(dcyc, sink, iters);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * threads / 64);
    CHECK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    int per_trip = 64;
    if (PAT == P_MFMA32 || PAT == P_MFMA16 || PAT == P_MFMA32X8 || PAT == P_DSREAD128) per_trip = 8;
    if (PAT >= P_MFMA32_DOT2 && PAT <= P_MFMA32_PKF16) per_trip = 8;  // per (MFMA + 8 fillers) group
    const double per_wave = med / ((double)iters * per_trip);
    printf("%-34s %-4s waves/SIMD %d: %7.2f cyc per instr (one wave), %7.2f per SIMD\n", kNames[PAT], DEP ? "dep" : "ind", waves_per_simd,
           per_wave, per_wave / waves_per_simd);
    return 0;
}

#define RUNALL(PAT)                                                     \
    for (int w : {1, 2, 4}) if (run<PAT, false>(w, iters, dcyc, sink)) return 1; \
    if (run<PAT, true>(1, iters, dcyc, sink)) return 1;

int main()
{
    unsigned long long *dcyc;
    float *sink;
    CHECK(hipMalloc(&dcyc, 256 * 16 * 8));
    CHECK(hipMalloc(&sink, 64));
    const int iters = 2000;
    RUNALL(P_FMA) RUNALL(P_PKFMA_F32) RUNALL(P_PKFMA_F16) RUNALL(P_DOT2_BF16) RUNALL(P_DOT2C_BF16) RUNALL(P_DOT2_F16)
    RUNALL(P_PERM) RUNALL(P_CVTPK) RUNALL(P_FMAMIX) RUNALL(P_AND) RUNALL(P_PKMUL_F32)
    RUNALL(P_MFMA32) RUNALL(P_MFMA16) RUNALL(P_MFMA32X8)
    for (int w : {1, 2}) {
        if (run<P_MFMA32_DOT2, false>(w, iters, dcyc, sink)) return 1;
        if (run<P_MFMA32_FMA, false>(w, iters, dcyc, sink)) return 1;
        if (run<P_MFMA32_PERM, false>(w, iters, dcyc, sink)) return 1;
        if (run<P_MFMA32_PKF16, false>(w, iters, dcyc, sink)) return 1;
    }
    for (int w : {1, 2, 4}) if (run<P_DSREAD128, false>(w, iters, dcyc, sink)) return 1;
    return 0;
}
