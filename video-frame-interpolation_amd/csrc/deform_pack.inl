// Shared pieces of the one-launch ModulatedDeformConvPack kernel (deform_pack3.inl): LDS read helpers, the bf16 -> f16 window
// conversion, the packed-f16 corner blend, the in-kernel stamp macros of the diagnostic build, and the 16-bit launcher.
// (The round-2 kernel that lived here - deform_pack_kernel: five k-groups per tap, all three fragments from global weight
// fragments, per-tap geometry - was replaced by deform_pack3.inl in round 3 and removed; git history has it.)
#include "deform.inl"
#include <mutex>
#include <type_traits>

#ifndef EMAVFI_DEFORM_ABL_NO_FALLBACK
#define EMAVFI_DEFORM_ABL_NO_FALLBACK 0  // timing-only ablation (wrong results): samples leaving the window read clamped window pixels
#endif

// Diagnostic build only (-DEMAVFI_DEFORM_STAMPS=1; cdna_hip_programming.md section 7, in-kernel stamps): s_memtime at
// the seams of the kernel, per-wave segment sums written to DeformParams::stamps (a buffer the diagnostic build of
// emavfi_api.hip allocates), read back with emavfi_debug_deform_stamps().  Never part of the shipped library.
#ifndef EMAVFI_DEFORM_STAMPS
#define EMAVFI_DEFORM_STAMPS 0
#endif
#if EMAVFI_DEFORM_STAMPS
__device__ __forceinline__ unsigned long long deform_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DEFORM_STAMP(var) const unsigned long long var = deform_stamp()
#elif defined(EMAVFI_DEFORM_FENCES)
#define DEFORM_STAMP(var) __builtin_amdgcn_sched_barrier(0)   // experiment: the stamped build's scheduling fences without its stamps
#else
#define DEFORM_STAMP(var)
#endif

typedef __attribute__((address_space(3))) const char lds_cchar_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ u32x4_t lds_read16(lds_cchar_t *p) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(p); }
__device__ __forceinline__ u32x2_t lds_read8(lds_cchar_t *p) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x2_t *>(p); }

// bf16 pair -> f16 pair (exact in f16's normal range; round-toward-zero only matters below 2^-14, saturates above 65504)
__device__ __forceinline__ unsigned bf16x2_to_f16x2(unsigned d)
{
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)));
}
template <typename TS> __device__ __forceinline__ u32x4_t to_f16_piece(u32x4_t v)
{
    if constexpr (std::is_same<TS, bf16_t>::value)
        return u32x4_t{bf16x2_to_f16x2(v[0]), bf16x2_to_f16x2(v[1]), bf16x2_to_f16x2(v[2]), bf16x2_to_f16x2(v[3])};
    return v;
}

// (w_lo, w_hi) packed f16 pair; blend on v_pk_mul_f16 / v_pk_fma_f16 with one half of the weight register broadcast to both
// channels of a dword.  Written as a splat the compiler folds into the instruction's op_sel modifiers (checked in the
// ISA: no v_perm / v_pack) - NOT as inline asm: an asm result consumed by the next MFMA gets no VALU-write -> MFMA-read
// wait states from hipcc (cdna_hip_programming.md 5.7), and the first version of this kernel read stale operand
// registers in exactly that way (odd fragment rows, first channel fragment only).
template <int HALF> __device__ __forceinline__ f16x2_t bcast_half(unsigned w)
{
    const f16x2_t p = __builtin_bit_cast(f16x2_t, w);
    return __builtin_shufflevector(p, p, HALF, HALF);
}
// four corner pieces (f16, NQ dwords each) -> one MFMA operand fragment; corners in the order (y0,x0) (y0,x1) (y1,x0) (y1,x1),
// weights w01 = (w00, w01), w23 = (w10, w11).  Accumulation in f16: four terms, the size of the final rounding.
template <int NQ> __device__ __forceinline__ f16x8 blend_corners(const u32x4_t (&v)[4], unsigned w01, unsigned w23)
{
    const f16x2_t z = {(half_t)0.0f, (half_t)0.0f};
    f16x2_t a[4] = {z, z, z, z};
    // explicit dword arrays: subscripting v[c][q] with the loop variable made hipcc (ROCm 7.2) use element 0 for every q
    const unsigned d0[4] = {v[0][0], v[0][1], v[0][2], v[0][3]}, d1[4] = {v[1][0], v[1][1], v[1][2], v[1][3]};
    const unsigned d2[4] = {v[2][0], v[2][1], v[2][2], v[2][3]}, d3[4] = {v[3][0], v[3][1], v[3][2], v[3][3]};
    // corner-major: the NQ chains are independent, so consecutive instructions never depend on each other
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_bit_cast(f16x2_t, d0[q]) * bcast_half<0>(w01);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d1[q]), bcast_half<1>(w01), a[q]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d2[q]), bcast_half<0>(w23), a[q]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d3[q]), bcast_half<1>(w23), a[q]);
    return f16x8{a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]};
}

struct OmTap { float dy, dx, mk; };
__device__ __forceinline__ OmTap load_om(const float *__restrict__ om, int tap, bool in_image)
{
    OmTap t;
    t.dy = om[2 * tap]; t.dx = om[2 * tap + 1]; t.mk = in_image ? om[18 + tap] : 0.0f;
    return t;
}

template <typename TS, bool FUSE_OFF> static int launch_deform_pack3(const DeformParams &p, hipStream_t s);

// 16-bit dtypes: the reference width runs the one-launch LDS-window kernel (deform_pack3.inl; weights in its layout, host:
// deform_pack3_shape), every other width the global-gather kernel of deform.inl
template <typename TS> static int launch_deform16(const DeformParams &p, hipStream_t s)
{
    if (p.pack3) {
        if (!deform_pack3_shape(p.ck, p.nf, p.cin_real, p.cout_real)) return -2;
        return p.off_w ? launch_deform_pack3<TS, true>(p, s) : launch_deform_pack3<TS, false>(p, s);
    }
    if (p.off_w) return -1;  // the host only asks for fusion after deform16_can_fuse_offset_conv()
    return launch_deform_any<TS>(p, s);
}
