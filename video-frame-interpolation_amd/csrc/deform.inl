// Modulated deformable 3x3 convolution (DCNv2), one offset group, one weight group.
// Replaces torchvision.ops.DeformConv2d as built at /root/reference/src/models/ema_vfi.py:45-51
// and called at :60 (three times per forward, :136-138).
//
//   out[o, y, x] = bias[o] + sum_{k=3i+j} sum_c W[o,c,i,j] * mask_k(y,x) * bilin(in[c], py, px)
//   py = y - 1 + i + dy_k(y,x),  px = x - 1 + j + dx_k(y,x)
//   bilin = 0 when py <= -1 || py >= H || px <= -1 || px >= W; corners outside the image add 0.
//
// One 256-thread workgroup owns an 8x32 tile of output pixels; wave w owns rows 2w and 2w+1
// (two 32-pixel MFMA fragments).  The contraction is D[cout][pixel] += W_k[cout][c] * S_k[c][pixel]
// with S_k the bilinearly sampled, mask-scaled input of tap k.  The MFMA B operand of lane (r, h)
// for k-group kg is exactly "pixel r, channels kg*CHKG + h*EPV .. +EPV" - one contiguous 16-byte
// piece of a channels-last pixel - so every lane GATHERS AND BLENDS ITS OWN OPERAND FRAGMENT in
// registers: four 16-byte corner loads, an fp32 blend with the corner weights, a pack to T, and
// the result feeds the matrix core directly.  There is no LDS "deformed im2col" tile and no
// sampling-table round trip; LDS only holds the tap's packed weights (double buffered, one
// barrier per tap).  Sampling positions, floor and corner weights are fp32 in both dtypes and
// are computed without comparisons (clamps), so no lane masks are held across memory waits.
// (Round 1 retired the accumulator chains before each gather as an empirical guard against a lane 48-63 corruption; its cause
// turned out to be a packed-f32 operand-select erratum - DESIGN.md section 5 - which the build now avoids, so the guard is gone.
// Regression test: tests/test_gpu_parity.py::test_deform_bf16_is_deterministic_at_two_workgroups_per_cu.)
//
// The gather reads global memory through L1/L2 (each input pixel is re-read by ~36 corner
// fetches of neighbouring pixels / taps; HBM sees it about once).
#include "common.h"
#include <mutex>

template <typename T, int CK, int NF> struct DeformCfg {
    using D = DT<T>;
    static constexpr int KG = CK / D::CHKG;
    static constexpr int WTAP = KG * NF * 1024;
    static constexpr int WVEC = KG * NF * 64;
    static constexpr int WPT = (WVEC + 255) / 256;
    static constexpr int LDS_BYTES = 2 * WTAP;
    // prefetch the next tap's weights through registers only while that costs <= 16 VGPRs
    static constexpr bool PREFETCH = WPT <= 4;
    // k-groups gathered per batch: bounds the registers holding loads in flight (16 per k-group)
    static constexpr int KB = (KG % 5 == 0) ? 5 : ((KG % 4 == 0) ? 4 : ((KG % 3 == 0) ? 3 : ((KG % 2 == 0) ? 2 : 1)));
    static_assert(KG % KB == 0, "k-group batch must divide KG");
    static_assert(LDS_BYTES <= 160 * 1024, "weights do not fit the 160 KiB LDS");
};

struct SampleTap {
    unsigned o[4];  // byte offsets of the four (clamped) corner pixels inside this sample's plane
    float w[4];     // bilinear weight x mask, 0 for a corner outside the image
};

// Compare-free: positions are clamped to [-2, size+1] (NaN -> -2), so the int conversions cannot
// overflow, and corner validity is an integer clamp to {0,1}.  A position <= -1 or >= size makes
// both of its corners invalid or zero-weighted = the operator's "outside -> 0" rule.
__device__ __forceinline__ SampleTap sample_tap_vals(float dy, float dx, float mk, int tap, int y, int x, int H, int W,
                                                     unsigned ps_bytes, int *yc0 = nullptr, int *yc1 = nullptr,
                                                     int *xc0 = nullptr, int *xc1 = nullptr)
{
    SampleTap t;
    const int i = tap / 3, j = tap - 3 * i;
    const float py = fminf(fmaxf((float)(y - 1 + i) + dy, -2.0f), (float)(H + 1));
    const float px = fminf(fmaxf((float)(x - 1 + j) + dx, -2.0f), (float)(W + 1));
    const float fy = floorf(py), fx = floorf(px);
    const int hl = (int)fy, wl = (int)fx, hh = hl + 1, wh = wl + 1;
    const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
    const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
    const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
    if (yc0) { *yc0 = hlc; *yc1 = hhc; *xc0 = wlc; *xc1 = whc; }  // clamped corner coordinates
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): rows, W and pixel indices are < 2^24
    const unsigned r0 = __umul24((unsigned)hlc, (unsigned)W), r1 = __umul24((unsigned)hhc, (unsigned)W);
    t.o[0] = __umul24(r0 + wlc, ps_bytes);
    t.o[1] = __umul24(r0 + whc, ps_bytes);
    t.o[2] = __umul24(r1 + wlc, ps_bytes);
    t.o[3] = __umul24(r1 + whc, ps_bytes);
    const int vhl = min(max(hl + 1, 0), 1) * min(max(H - hl, 0), 1);  // 0 <= hl <= H-1
    const int vhh = min(max(hh + 1, 0), 1) * min(max(H - hh, 0), 1);
    const int vwl = min(max(wl + 1, 0), 1) * min(max(W - wl, 0), 1);
    const int vwh = min(max(wh + 1, 0), 1) * min(max(W - wh, 0), 1);
    t.w[0] = mk * (uh * uw) * (float)(vhl * vwl);
    t.w[1] = mk * (uh * lw) * (float)(vhl * vwh);
    t.w[2] = mk * (lh * uw) * (float)(vhh * vwl);
    t.w[3] = mk * (lh * lw) * (float)(vhh * vwh);
    return t;
}
__device__ __forceinline__ SampleTap sample_tap(const float *__restrict__ om, int tap, int y, int x, int H, int W,
                                                unsigned ps_bytes, bool in_image)
{
    return sample_tap_vals(om[2 * tap], om[2 * tap + 1], in_image ? om[18 + tap] : 0.0f, tap, y, x, H, W, ps_bytes);
}

// fp32 blend of four 16-byte corner pieces -> one MFMA operand fragment
__device__ __forceinline__ bf16x8 blend4(const uint4 (&v)[4], const float (&w)[4], bf16_t)
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned d[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // a dword holds channels 2q (low half) and 2q+1 (high half)
            a[2 * q] = fmaf(w[c], __uint_as_float(d[q] << 16), a[2 * q]);
            a[2 * q + 1] = fmaf(w[c], __uint_as_float(d[q] & 0xffff0000u), a[2 * q + 1]);
        }
    }
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)a[4], (bf16_t)a[5], (bf16_t)a[6], (bf16_t)a[7]};
}
// Same blend on the dot-product unit: v_dot2c_f32_bf16 takes the packed pair as it sits in the register
// (no unpack) - weights (w, 0) pick the low channel of a dword, (0, w) the high one.  fp32 accumulation,
// but the four corner weights are rounded to bf16 (8 significant bits, like the blended value itself).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
struct BlendW { unsigned lo[4], hi[4]; };  // bf16(w) in the low / high half, other half zero
__device__ __forceinline__ BlendW blend_weights_bf16(const float (&w)[4])
{
    BlendW r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned short b = __builtin_bit_cast(unsigned short, (bf16_t)w[c]);
        r.lo[c] = b;
        r.hi[c] = (unsigned)b << 16;
    }
    return r;
}
// NQ = dwords of the piece that are blended (channels 2*NQ.. of the result are 0)
template <int NQ = 4>
__device__ __forceinline__ bf16x8 blend4_dot2(const uint4 (&v)[4], const BlendW &w)
{
    float a[8];
#pragma unroll
    for (int j = 2 * NQ; j < 8; ++j) a[j] = 0.0f;
    {   // first corner: the three-source form with a literal 0 addend (no accumulator to clear first)
        const unsigned d[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(a[2 * q]) : "v"(d[q]), "v"(w.lo[0]));
            asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(a[2 * q + 1]) : "v"(d[q]), "v"(w.hi[0]));
        }
    }
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const unsigned d[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a[2 * q] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, d[q]), __builtin_bit_cast(bf16x2_t, w.lo[c]), a[2 * q], false);
            a[2 * q + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, d[q]), __builtin_bit_cast(bf16x2_t, w.hi[c]), a[2 * q + 1], false);
        }
    }
    return bf16x8{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)a[4], (bf16_t)a[5], (bf16_t)a[6], (bf16_t)a[7]};
}
// f16, packed: v_pk_mul_f16 / v_pk_fma_f16 on the channel pairs as they sit in the registers, corner weights as
// (w, w) f16 pairs - 16 instructions per 8-channel piece, the result is already the packed MFMA operand (no
// conversion).  Accumulation in f16: four terms of one sign pattern, error ~2^-10 relative, the size of the final
// rounding the fp32 blend needs anyway.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
struct BlendWh { f16x2_t pk[4]; };
__device__ __forceinline__ BlendWh blend_weights_f16(const float (&w)[4])
{
    BlendWh r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r.pk[c] = f16x2_t{(half_t)w[c], (half_t)w[c]};
    return r;
}
template <int NQ = 4>
__device__ __forceinline__ f16x8 blend4_pk(const uint4 (&v)[4], const BlendWh &w)
{
    f16x2_t a[4];
#pragma unroll
    for (int q = NQ; q < 4; ++q) a[q] = f16x2_t{(half_t)0.0f, (half_t)0.0f};
    {
        const unsigned d[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
        for (int q = 0; q < NQ; ++q) a[q] = __builtin_bit_cast(f16x2_t, d[q]) * w.pk[0];
    }
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const unsigned d[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[q]), w.pk[c], a[q]);
    }
    return f16x8{a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]};
}
// f16: fp32 blend with fp32 weights on v_fma_mix_f32, which reads an f16 half of a packed register directly
// (op_sel_hi marks source 0 as f16, op_sel picks its high half) - no unpack.  Written as asm because hipcc
// otherwise prefers 2 x v_cvt_f32_f16 + v_pk_fma_f32 per channel pair (3 instructions instead of 2).
__device__ __forceinline__ f16x8 blend4(const uint4 (&v)[4], const float (&w)[4], half_t)
{
    float a[8];
    {   // first corner: literal 0 addend (no accumulator to clear first)
        const unsigned d[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(a[2 * q]) : "v"(d[q]), "v"(w[0]));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a[2 * q + 1]) : "v"(d[q]), "v"(w[0]));
        }
    }
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const unsigned d[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[2 * q]) : "v"(d[q]), "v"(w[c]));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[2 * q + 1]) : "v"(d[q]), "v"(w[c]));
        }
    }
    return f16x8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)a[4], (half_t)a[5], (half_t)a[6], (half_t)a[7]};
}
__device__ __forceinline__ f32x4 blend4(const uint4 (&v)[4], const float (&w)[4], float)
{
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 x = __builtin_bit_cast(f32x4, v[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(w[c], x[j], a[j]);
    }
    return a;
}

template <typename T, int CK, int NF>
__global__ __launch_bounds__(256, 2) void deform_kernel(const DeformParams p)
{
    using C = DeformCfg<T, CK, NF>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_w = smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int H = p.H, W = p.W;
    const unsigned ps_bytes = (unsigned)p.x_ps * (unsigned)sizeof(T);  // one sample's plane is < 4 GiB (host check)

    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];

    // this lane's two pixels (fragment rows m = 0, 1), shared by its h = 0 / h = 1 partner lanes
    const int px_x = blockIdx.x * 32 + r;
    int py_y[2];
    bool in_img[2];
    const float *om[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        py_y[m] = blockIdx.y * 8 + wave * 2 + m;
        in_img[m] = py_y[m] < H && px_x < W;
        om[m] = p.om + (((size_t)b * H + (in_img[m] ? py_y[m] : 0)) * W + (in_img[m] ? px_x : 0)) * 32;
    }
    const char *gx = (const char *)p.x + (size_t)b * H * W * ps_bytes + h * 16;

    // tap 0 weights -> LDS buffer 0
    for (int idx = tid; idx < C::WVEC; idx += 256)
        *reinterpret_cast<uint4 *>(lds_w + idx * 16) = *reinterpret_cast<const uint4 *>((const char *)p.w + idx * 16);
    __syncthreads();

    int cur = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        // prefetch the next tap's packed weights into registers (written to LDS behind the MFMAs)
        uint4 wr[C::WPT];
        if (C::PREFETCH && tap < 8) {
#pragma unroll
            for (int i = 0; i < C::WPT; ++i) {
                const int idx = tid + i * 256;
                if (idx < C::WVEC) wr[i] = *reinterpret_cast<const uint4 *>((const char *)p.w + (size_t)(tap + 1) * C::WTAP + idx * 16);
            }
        }
        const char *wb = lds_w + cur * C::WTAP + lane * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const SampleTap st = sample_tap(om[m], tap, py_y[m], px_x, H, W, ps_bytes, in_img[m]);
#pragma unroll
            for (int k0 = 0; k0 < C::KG; k0 += C::KB) {
                uint4 v[C::KB][4];
#pragma unroll
                for (int kk = 0; kk < C::KB; ++kk)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        v[kk][c] = *reinterpret_cast<const uint4 *>(gx + st.o[c] + (unsigned)((k0 + kk) * 32));
                vec xf[C::KB];
#pragma unroll
                for (int kk = 0; kk < C::KB; ++kk) xf[kk] = blend4(v[kk], st.w, T{});
#pragma unroll
                for (int kk = 0; kk < C::KB; ++kk)
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        const vec wv = *reinterpret_cast<const vec *>(wb + ((k0 + kk) * NF + n) * 1024);
                        mma_kg(acc[m][n], wv, xf[kk]);
                    }
            }
        }
        if (tap < 8) {
#pragma unroll
            for (int i = 0; i < C::WPT; ++i) {
                const int idx = tid + i * 256;
                if (idx < C::WVEC)
                    *reinterpret_cast<uint4 *>(lds_w + (cur ^ 1) * C::WTAP + idx * 16) =
                        C::PREFETCH ? wr[i]
                                    : *reinterpret_cast<const uint4 *>((const char *)p.w + (size_t)(tap + 1) * C::WTAP + idx * 16);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue (no activation: ema_vfi.py:136-138 chains the blocks directly) ----
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!in_img[m]) continue;
        T *op = reinterpret_cast<T *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (p.cstore - n * 32 > 0) store_frag(op + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
    }
}

template <typename T, int CK, int NF> static int launch_deform_inst(const DeformParams &p, hipStream_t s)
{
    using C = DeformCfg<T, CK, NF>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&deform_kernel<T, CK, NF>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    dim3 grid((p.W + 31) / 32, (p.H + 7) / 8, p.B);
    deform_kernel<T, CK, NF><<<grid, 256, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}

// (CK, NF): fusion widths mid+3 for mid in {8, 16, 32, 64} -> padded 16, 32, 48, 80.
#define EMAVFI_DEFORM_INSTANCES(X) X(16, 1) X(32, 1) X(48, 2) X(80, 3)

template <typename T> static int launch_deform_any(const DeformParams &p, hipStream_t s)
{
#define X(CK_, NF_) \
    if (p.ck == CK_ && p.nf == NF_) return launch_deform_inst<T, CK_, NF_>(p, s);
    EMAVFI_DEFORM_INSTANCES(X)
#undef X
    return -2;
}
