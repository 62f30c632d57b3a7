// Round 2, two-stream corruption of the tiled warp kernel.  What the failing launches showed:
//   * only while a kernel from ANOTHER stream shared the GPU (a rocBLAS bf16 GEMM was the strongest trigger),
//   * always lanes 48-63 of a wave, always the term fed by ONE particular global_load_dword of a 12-load gather burst,
//   * the wrong product equals "the load never happened" (the destination register's previous content, a small integer,
//     read as a denormal float), and the consumer is the first VALU instruction behind the counted s_waitcnt vmcnt(10)
//     that retires that load,
//   * the same source with one s_waitcnt vmcnt(0) behind the whole burst never failed.
// This probe reproduces the instruction pattern without the warp kernel: every lane issues 12 scattered
// global_load_dword into registers that hold a poison value, then retires them one by one with counted waits
// (vmcnt(10) for the first two, then 9, 8, ... 0) and copies each register to a "captured" register with the first VALU
// instruction behind its wait (NOPS wait states in between).  After a final vmcnt(0) the captured copy must equal the
// table entry; a captured poison value means the wait released before the data was in the register.
// Aggressors on a second stream: none / a spin of v_mfma_f32_32x32x16_bf16 / rocBLAS bf16 GEMMs (2048^3).
// A second victim (victim_mirror) replays the failing build's own instruction sequence with its register numbers.
//   hipcc --offload-arch=gfx950 -O3 -o vmcnt_gather_beside_gemm vmcnt_gather_beside_gemm.hip -lrocblas
//   ./vmcnt_gather_beside_gemm        (mirror only)      ./vmcnt_gather_beside_gemm all     (mirror + the gather / EXEC-mask matrix)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_spin(float *sink, int rounds)
{
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    for (int r = 0; r < rounds; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

__host__ __device__ inline unsigned table_value(unsigned i) { return i * 2654435761u + 0x9e3779b9u; }   // never a small integer twice in a row

__global__ void fill(unsigned *t, unsigned n)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) t[i] = table_value(i);
}

#define NL 12
// COUNTED = true: waits vmcnt(10), 9, 8, ... as hipcc schedules the warp's gather; false: one vmcnt(0) behind the burst.
// DMA = true: the workgroup first fills a 49 KiB LDS window by global->LDS DMA (49 one-KiB pieces, waves take them round
// robin), waits vmcnt(0) and passes a barrier - the warp kernel's prologue - before the gather loop.
// EXECMODE: 0 all lanes, 1 only lanes 48-63, 2 only lanes 0-15, 3 a fresh random wave-uniform mask per burst
// (random bits with a random subset of the four 16-lane groups switched off), 4 lanes 32-63.
template <bool COUNTED, int NOPS, bool DMA, int EXECMODE>
__global__ __launch_bounds__(256) void victim(const unsigned *__restrict__ table, unsigned mask, unsigned long long *bad, int iters, unsigned salt)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 256u + threadIdx.x + salt) * 2654435761u + 99991u;
    unsigned long long local[4 * NL] = {0};
    if (DMA) {
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        __shared__ __attribute__((aligned(16))) unsigned win[49 * 256];
        const unsigned wave = threadIdx.x >> 6;
        const unsigned base = (blockIdx.x * 7919u + salt * 104729u) & mask & ~0x3fffu;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int j = i * 4 + wave;
            if (j < 49) __builtin_amdgcn_global_load_lds((gptr_t *)(table + ((base + (j * 64 + lane) * 4) & mask)), (lptr_t *)(win + j * 256), 16, 0, 0);
        }
        __syncthreads();
        if (win[(threadIdx.x * 37) % (49 * 256)] == 0x12345u) bad[0] += 1;   // keep the window alive
    }
    unsigned ws = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6) + salt * 131u) * 40503u + 977u);
    for (int it = 0; it < iters; ++it) {
        if (EXECMODE == 1 && lane < 48) continue;
        if (EXECMODE == 2 && lane >= 16) continue;
        if (EXECMODE == 4 && lane < 32) continue;
        if (EXECMODE == 3) {
            ws = ws * 1664525u + 1013904223u; const unsigned lo = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned hi = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned g = ws >> 28;
            unsigned long long m = ((unsigned long long)hi << 32) | lo;
            for (int q = 0; q < 4; ++q) if (!(g >> q & 1)) m &= ~(0xffffull << (16 * q));
            if (!(m >> lane & 1)) continue;
        }
        unsigned idx[NL];
        const unsigned *a[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            s = s * 1664525u + 1013904223u;
            idx[j] = (s >> 4) & mask;
            a[j] = table + idx[j];
        }
        unsigned r[NL], c[NL];
        if (COUNTED) {
            asm volatile(
                "v_mov_b32 %0, 7\n\tv_mov_b32 %1, 7\n\tv_mov_b32 %2, 7\n\tv_mov_b32 %3, 7\n\tv_mov_b32 %4, 7\n\tv_mov_b32 %5, 7\n\t"
                "v_mov_b32 %6, 7\n\tv_mov_b32 %7, 7\n\tv_mov_b32 %8, 7\n\tv_mov_b32 %9, 7\n\tv_mov_b32 %10, 7\n\tv_mov_b32 %11, 7\n\t"
                "s_nop 4\n\t"
                "global_load_dword %0, %24, off\n\tglobal_load_dword %1, %25, off\n\tglobal_load_dword %2, %26, off\n\t"
                "global_load_dword %3, %27, off\n\tglobal_load_dword %4, %28, off\n\tglobal_load_dword %5, %29, off\n\t"
                "global_load_dword %6, %30, off\n\tglobal_load_dword %7, %31, off\n\tglobal_load_dword %8, %32, off\n\t"
                "global_load_dword %9, %33, off\n\tglobal_load_dword %10, %34, off\n\tglobal_load_dword %11, %35, off\n\t"
                "s_waitcnt vmcnt(10)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %13, %1\n\tv_mov_b32 %12, %0\n\t"
                "s_waitcnt vmcnt(9)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %14, %2\n\t"
                "s_waitcnt vmcnt(8)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %15, %3\n\t"
                "s_waitcnt vmcnt(7)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %16, %4\n\t"
                "s_waitcnt vmcnt(6)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %17, %5\n\t"
                "s_waitcnt vmcnt(5)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %18, %6\n\t"
                "s_waitcnt vmcnt(4)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %19, %7\n\t"
                "s_waitcnt vmcnt(3)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %20, %8\n\t"
                "s_waitcnt vmcnt(2)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %21, %9\n\t"
                "s_waitcnt vmcnt(1)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %22, %10\n\t"
                "s_waitcnt vmcnt(0)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %23, %11\n\t"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
                  "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]),
                  "=&v"(c[6]), "=&v"(c[7]), "=&v"(c[8]), "=&v"(c[9]), "=&v"(c[10]), "=&v"(c[11])
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]),
                  "v"(a[11]), "i"(NOPS)
                : "memory");
        } else {
            asm volatile(
                "v_mov_b32 %0, 7\n\tv_mov_b32 %1, 7\n\tv_mov_b32 %2, 7\n\tv_mov_b32 %3, 7\n\tv_mov_b32 %4, 7\n\tv_mov_b32 %5, 7\n\t"
                "v_mov_b32 %6, 7\n\tv_mov_b32 %7, 7\n\tv_mov_b32 %8, 7\n\tv_mov_b32 %9, 7\n\tv_mov_b32 %10, 7\n\tv_mov_b32 %11, 7\n\t"
                "s_nop 4\n\t"
                "global_load_dword %0, %24, off\n\tglobal_load_dword %1, %25, off\n\tglobal_load_dword %2, %26, off\n\t"
                "global_load_dword %3, %27, off\n\tglobal_load_dword %4, %28, off\n\tglobal_load_dword %5, %29, off\n\t"
                "global_load_dword %6, %30, off\n\tglobal_load_dword %7, %31, off\n\tglobal_load_dword %8, %32, off\n\t"
                "global_load_dword %9, %33, off\n\tglobal_load_dword %10, %34, off\n\tglobal_load_dword %11, %35, off\n\t"
                "s_waitcnt vmcnt(0)\n\t.rept %c36\n\ts_nop 0\n\t.endr\n\t"
                "v_mov_b32 %23, %11\n\tv_mov_b32 %22, %10\n\tv_mov_b32 %21, %9\n\tv_mov_b32 %20, %8\n\tv_mov_b32 %19, %7\n\tv_mov_b32 %18, %6\n\t"
                "v_mov_b32 %17, %5\n\tv_mov_b32 %16, %4\n\tv_mov_b32 %15, %3\n\tv_mov_b32 %14, %2\n\tv_mov_b32 %13, %1\n\tv_mov_b32 %12, %0\n\t"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
                  "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]),
                  "=&v"(c[6]), "=&v"(c[7]), "=&v"(c[8]), "=&v"(c[9]), "=&v"(c[10]), "=&v"(c[11])
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]),
                  "v"(a[11]), "i"(NOPS)
                : "memory");
        }
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (c[j] != table_value(idx[j])) local[j * 4 + (lane >> 4)]++;
    }
    for (int i = 0; i < 4 * NL; ++i)
        if (local[i]) atomicAdd(&bad[i], local[i]);
}


// The failing build's own sequence (warp_tiled_kernel<3, void>, global path of one pixel, hipcc's register choice):
// 12 scattered loads, then scalar and PACKED f32 multiplies placed directly behind counted waits; weights are exact powers
// of two (2, 0.5, 4, 0.25) and the table holds floats in [1, 2), so every product is exact and checked bit for bit.
//   products: 0 ne*c0_01 (v_mul behind vmcnt(10))   1,2 [nw*c0_00, sw*c0_10] (v_pk_mul behind vmcnt(9))   3 se*c0_11 (v_mul, vmcnt(8))
//             4,5 [nw*c2_00, ne*c1_01] (v_pk_mul behind vmcnt(3); the high half is the term that went missing; its destination
//                 v[20:21] is the ADDRESS pair of load #10, which vmcnt(3) leaves in flight)
//             6,7 [nw*c1_00, ne*c2_01] (v_pk_mul behind vmcnt(2), overwrites the weight pair in place)
//             8,9 [sw*c1_10, sw*c2_10] (v_pk_mul op_sel_hi:[0,1] behind vmcnt(1))   10,11 se*c1_11, se*c2_11 (v_mul behind vmcnt(0))
template <int EXECMODE>
__global__ __launch_bounds__(256) void victim_mirror(const float *__restrict__ table, unsigned mask, unsigned long long *bad, int iters, unsigned salt)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned s = (blockIdx.x * 256u + threadIdx.x + salt) * 2654435761u + 99991u;
    unsigned long long local[4 * NL] = {0};
    unsigned ws = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6) + salt * 131u) * 40503u + 977u);
    for (int it = 0; it < iters; ++it) {
        if (EXECMODE == 1 && lane < 48) continue;
        if (EXECMODE == 3) {
            ws = ws * 1664525u + 1013904223u; const unsigned lo = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned hi = ws;
            ws = ws * 1664525u + 1013904223u; const unsigned g = ws >> 28;
            unsigned long long m = ((unsigned long long)hi << 32) | lo;
            for (int q = 0; q < 4; ++q) if (!(g >> q & 1)) m &= ~(0xffffull << (16 * q));
            if (!(m >> lane & 1)) continue;
        }
        unsigned idx[NL];
        const float *a[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            s = s * 1664525u + 1013904223u;
            idx[j] = (s >> 4) & mask;
            a[j] = table + idx[j];
        }
        float o[NL];
        asm volatile(
            "v_mov_b32 v1, 7\n\tv_mov_b32 v3, 7\n\tv_mov_b32 v48, 7\n\tv_mov_b32 v49, 7\n\tv_mov_b32 v50, 7\n\tv_mov_b32 v51, 7\n\t"
            "v_mov_b32 v52, 7\n\tv_mov_b32 v53, 7\n\tv_mov_b32 v54, 7\n\tv_mov_b32 v55, 7\n\tv_mov_b32 v56, 7\n\tv_mov_b32 v57, 7\n\t"
            "v_mov_b32 v20, %24\n\tv_mov_b32 v21, %25\n\t"   // load #10's address lives in v[20:21], as in the failing build
            "v_mov_b32 v12, 2.0\n\tv_mov_b32 v13, 0.5\n\tv_mov_b32 v15, 4.0\n\tv_mov_b32 v2, 0x3e800000\n\tv_mov_b32 v7, 0\n\t"
            "s_nop 4\n\t"
            "global_load_dword v48, %12, off\n\tglobal_load_dword v1, %13, off\n\tglobal_load_dword v49, %14, off\n\tglobal_load_dword v3, %15, off\n\t"
            "global_load_dword v50, %16, off\n\tglobal_load_dword v53, %17, off\n\tglobal_load_dword v54, %18, off\n\tglobal_load_dword v56, %19, off\n\t"
            "global_load_dword v52, %20, off\n\tglobal_load_dword v51, v[20:21], off\n\tglobal_load_dword v55, %22, off\n\tglobal_load_dword v57, %23, off\n\t"
            "v_mov_b32 v14, v12\n\tv_mov_b32 v6, v15\n\t"
            "s_waitcnt vmcnt(10)\n\tv_mul_f32 v1, v13, v1\n\t"
            "s_waitcnt vmcnt(9)\n\tv_pk_mul_f32 v[14:15], v[14:15], v[48:49]\n\t"
            "s_waitcnt vmcnt(8)\n\tv_mul_f32 v17, v2, v3\n\t"
            "s_waitcnt vmcnt(3)\n\tv_pk_mul_f32 v[20:21], v[12:13], v[52:53]\n\t"
            "s_waitcnt vmcnt(2)\n\tv_pk_mul_f32 v[12:13], v[12:13], v[50:51]\n\t"
            "s_waitcnt vmcnt(1)\n\tv_pk_mul_f32 v[6:7], v[6:7], v[54:55] op_sel_hi:[0,1]\n\t"
            "s_waitcnt vmcnt(0)\n\tv_mul_f32 v56, v2, v56\n\tv_mul_f32 v57, v2, v57\n\t"
            "v_mov_b32 %0, v1\n\tv_mov_b32 %1, v14\n\tv_mov_b32 %2, v15\n\tv_mov_b32 %3, v17\n\tv_mov_b32 %4, v20\n\tv_mov_b32 %5, v21\n\t"
            "v_mov_b32 %6, v12\n\tv_mov_b32 %7, v13\n\tv_mov_b32 %8, v6\n\tv_mov_b32 %9, v7\n\tv_mov_b32 %10, v56\n\tv_mov_b32 %11, v57\n\t"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
              "=&v"(o[10]), "=&v"(o[11])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]),
              "v"((unsigned)(unsigned long long)a[9]), "v"((unsigned)((unsigned long long)a[9] >> 32))
            : "memory", "v1", "v2", "v3", "v6", "v7", "v12", "v13", "v14", "v15", "v17", "v20", "v21", "v48", "v49", "v50", "v51", "v52", "v53",
              "v54", "v55", "v56", "v57");
        // load j -> (weight, product slot): #1 c0_00 #2 c0_01 #3 c0_10 #4 c0_11 #5 c1_00 #6 c1_01 #7 c1_10 #8 c1_11 #9 c2_00 #10 c2_01 #11 c2_10 #12 c2_11
        const float w[NL] = {2.0f, 0.5f, 4.0f, 0.25f, 2.0f, 0.5f, 4.0f, 0.25f, 2.0f, 0.5f, 4.0f, 0.25f};
        const int slot[NL] = {1, 0, 2, 3, 6, 5, 8, 10, 4, 7, 9, 11};
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const float expect = w[j] * __uint_as_float(0x3f800000u | (table_value(idx[j]) & 0x7fffffu));
            if (__float_as_uint(o[slot[j]]) != __float_as_uint(expect)) local[slot[j] * 4 + (lane >> 4)]++;
        }
    }
    for (int i = 0; i < 4 * NL; ++i)
        if (local[i]) atomicAdd(&bad[i], local[i]);
}

__global__ void fill_float(float *t, unsigned n)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) t[i] = __uint_as_float(0x3f800000u | (table_value(i) & 0x7fffffu));
}

struct Gemm {
    rocblas_handle h = nullptr;
    void *A = nullptr, *C = nullptr;
    int n = 2048;
    int init(hipStream_t s)
    {
        if (rocblas_create_handle(&h) != rocblas_status_success) return 1;
        rocblas_set_stream(h, s);
        CHECK(hipMalloc(&A, (size_t)n * n * 2)); CHECK(hipMalloc(&C, (size_t)n * n * 2));
        CHECK(hipMemset(A, 0x3c, (size_t)n * n * 2));
        return 0;
    }
    int run(int count)
    {
        const float alpha = 1.0f, beta = 0.0f;
        for (int i = 0; i < count; ++i) {
            rocblas_status st = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_none, n, n, n, &alpha, A, rocblas_datatype_bf16_r, n, A,
                                                rocblas_datatype_bf16_r, n, &beta, C, rocblas_datatype_bf16_r, n, C, rocblas_datatype_bf16_r, n,
                                                rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
            if (st != rocblas_status_success) { printf("rocblas_gemm_ex: %d\n", (int)st); return 1; }
        }
        return 0;
    }
};

template <bool COUNTED, int NOPS, bool DMA, int EXECMODE>
static int run(const char *name, int aggr, Gemm &g, hipStream_t sa, hipStream_t sv, const unsigned *table, unsigned mask, unsigned long long *dbad, float *sink)
{
    CHECK(hipMemset(dbad, 0, 4 * NL * 8));
    CHECK(hipDeviceSynchronize());
    if (aggr == 1) mfma_spin<<<256, 256, 0, sa>>>(sink, 1500000);
    if (aggr == 2 && g.run(DMA ? 12000 : 4000)) return 1;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, sv));
    const int launches = DMA ? 4000 : 400, wgs = 120, iters = DMA ? 8 : 64;   // the warp launch that failed: 120 workgroups of 256 threads, 8 bursts per thread
    for (int l = 0; l < launches; ++l) victim<COUNTED, NOPS, DMA, EXECMODE><<<wgs, 256, 0, sv>>>(table, mask, dbad, iters, (unsigned)l);
    CHECK(hipEventRecord(e1, sv));
    CHECK(hipStreamSynchronize(sv));
    const bool still = aggr == 0 || hipStreamQuery(sa) == hipErrorNotReady;
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(4 * NL);
    CHECK(hipMemcpy(h.data(), dbad, 4 * NL * 8, hipMemcpyDeviceToHost));
    unsigned long long tot = 0, grp[4] = {0};
    char perload[256]; int off = 0;
    for (int j = 0; j < NL; ++j) {
        unsigned long long lj = 0;
        for (int q = 0; q < 4; ++q) { lj += h[j * 4 + q]; grp[q] += h[j * 4 + q]; }
        tot += lj;
        off += snprintf(perload + off, sizeof perload - off, "%llu ", lj);
    }
    static const char *an[] = {"alone", "beside an MFMA spin", "beside rocBLAS bf16 GEMMs"};
    printf("%-34s %-26s %7.1f ms  stale captures %7llu of %.2e loads   by lane group [%llu %llu %llu %llu]   by load [%s]%s\n", name, an[aggr], ms, tot,
           (double)launches * wgs * 256 * iters * NL, grp[0], grp[1], grp[2], grp[3], perload, still ? "" : "  (aggressor ended early)");
    return 0;
}

template <int EXECMODE>
static int run_mirror(const char *name, int aggr, Gemm &g, hipStream_t sa, hipStream_t sv, const float *ftable, unsigned mask, unsigned long long *dbad, float *sink)
{
    CHECK(hipMemset(dbad, 0, 4 * NL * 8));
    CHECK(hipDeviceSynchronize());
    if (aggr == 1) mfma_spin<<<256, 256, 0, sa>>>(sink, 1500000);
    if (aggr == 2 && g.run(8000)) return 1;
    const int launches = 4000, wgs = 120, iters = 8;
    for (int l = 0; l < launches; ++l) victim_mirror<EXECMODE><<<wgs, 256, 0, sv>>>(ftable, mask, dbad, iters, (unsigned)l);
    CHECK(hipStreamSynchronize(sv));
    const bool still = aggr == 0 || hipStreamQuery(sa) == hipErrorNotReady;
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(4 * NL);
    CHECK(hipMemcpy(h.data(), dbad, 4 * NL * 8, hipMemcpyDeviceToHost));
    unsigned long long tot = 0, grp[4] = {0};
    char per[256]; int off = 0;
    for (int j = 0; j < NL; ++j) {
        unsigned long long lj = 0;
        for (int q = 0; q < 4; ++q) { lj += h[j * 4 + q]; grp[q] += h[j * 4 + q]; }
        tot += lj;
        off += snprintf(per + off, sizeof per - off, "%llu ", lj);
    }
    static const char *an[] = {"alone", "beside an MFMA spin", "beside rocBLAS bf16 GEMMs"};
    printf("%-34s %-26s wrong products %7llu   by lane group [%llu %llu %llu %llu]   by product [%s]%s\n", name, an[aggr], tot, grp[0], grp[1], grp[2], grp[3], per,
           still ? "" : "  (aggressor ended early)");
    return 0;
}

int main(int argc, char **)
{
    hipStream_t sa, sv;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const unsigned n = 1u << 20;   // 4 MiB table: the gathers hit L2, mostly miss the 32 KiB vector L1
    unsigned *table; unsigned long long *dbad; float *sink;
    CHECK(hipMalloc(&table, n * 4)); CHECK(hipMalloc(&dbad, 4 * NL * 8)); CHECK(hipMalloc(&sink, 64));
    fill<<<1024, 256>>>(table, n);
    CHECK(hipDeviceSynchronize());
    float *ftable;
    CHECK(hipMalloc(&ftable, n * 4));
    fill_float<<<1024, 256>>>(ftable, n);
    CHECK(hipDeviceSynchronize());
    Gemm g;
    if (g.init(sa)) return 1;
    if (g.run(20)) return 1;
    CHECK(hipDeviceSynchronize());
    for (int aggr = 0; aggr <= 2; ++aggr) {
        if (run_mirror<0>("failing build's sequence, all lanes", aggr, g, sa, sv, ftable, n - 1, dbad, sink)) return 1;
        if (run_mirror<1>("failing build's sequence, 48-63", aggr, g, sa, sv, ftable, n - 1, dbad, sink)) return 1;
        if (run_mirror<3>("failing build's sequence, random", aggr, g, sa, sv, ftable, n - 1, dbad, sink)) return 1;
    }
    for (int aggr = 0; aggr <= 2 && argc > 1; ++aggr) {
        if (run<true, 0, false, 0>("counted waits, all lanes", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, false, 1>("counted waits, lanes 48-63 only", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, false, 4>("counted waits, lanes 32-63 only", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, false, 2>("counted waits, lanes 0-15 only", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, false, 3>("counted waits, random EXEC", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 8, false, 1>("counted + 8 ws, lanes 48-63 only", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<false, 0, false, 1>("one vmcnt(0), lanes 48-63 only", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, true, 1>("DMA prologue + counted, 48-63", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
        if (run<true, 0, true, 3>("DMA prologue + counted, random", aggr, g, sa, sv, table, n - 1, dbad, sink)) return 1;
    }
    return 0;
}
