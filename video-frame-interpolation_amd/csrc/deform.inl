// Modulated deformable 3x3 convolution (DCNv2), one offset group, one weight group.
// Replaces torchvision.ops.DeformConv2d as built at /root/reference/src/models/ema_vfi.py:45-51
// and called at :60 (three times per forward, :136-138).
//
//   out[o, y, x] = bias[o] + sum_{k=3i+j} sum_c W[o,c,i,j] * mask_k(y,x) * bilin(in[c], py, px)
//   py = y - 1 + i + dy_k(y,x),  px = x - 1 + j + dx_k(y,x)
//   bilin = 0 when py <= -1 || py >= H || px <= -1 || px >= W; corners outside the image add 0.
//
// One 256-thread workgroup owns an 8x32 tile of output pixels.  Per tap:
//   A  one thread per pixel turns (dy, dx, mask) into four clamped corner pixel indices and four
//      corner weights (bilinear weight x mask, 0 for an invalid corner) in an LDS table -
//      coordinates, floor and weights are fp32 in both dtypes.  This phase is written WITHOUT
//      comparisons: positions are clamped with v_max/v_min and corner validity is an integer
//      clamp to {0,1} (no lane masks live in SGPRs across the mask load's wait);
//   B  all threads gather: one item = (pixel, 16-byte channel piece); four 16-byte loads from
//      the channels-last input (all channels of a corner are contiguous), fp32 blend, and the
//      blended piece goes to the LDS "deformed im2col" tile [pixel][CK];
//   C  the tile is contracted with the tap's packed weights on the matrix cores
//      (D[cout][pixel], same fragment scheme as conv3x3.inl); the accumulator chains are
//      retired (mfma_retire, common.h) before the next tap's phase A so that code never runs in
//      the shadow of queued MFMAs - regression test:
//      tests/test_gpu_parity.py::test_deform_bf16_is_deterministic_at_two_workgroups_per_cu.
// The gather reads global memory through L1/L2 (each input pixel is re-read by ~36 corner
// fetches of neighbouring pixels/taps; HBM sees it about once).
#include "common.h"

template <typename T, int CK, int NF> struct DeformCfg {
    using D = DT<T>;
    static constexpr int NPIX = 256;
    static constexpr int PSTR = LdsPix<T, CK>::BYTES;
    static constexpr int PIECES = CK * (int)sizeof(T) / 16;
    static constexpr int KG = CK / D::CHKG;
    static constexpr int WTAP = KG * NF * 1024;
    static constexpr int WVEC = KG * NF * 64;
    static constexpr int LDS_S = NPIX * PSTR;
    static constexpr int LDS_TAB = NPIX * 32;  // 4 int + 4 float per pixel
    static constexpr int LDS_BYTES = LDS_S + LDS_TAB + WTAP;
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
};

__device__ __forceinline__ void blend_piece(f32x4 &lo, f32x4 &hi, const uint4 &raw, float w, bf16_t)
{
    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lo[j] = fmaf(w, (float)v[j], lo[j]);
        hi[j] = fmaf(w, (float)v[4 + j], hi[j]);
    }
}
__device__ __forceinline__ void blend_piece(f32x4 &lo, f32x4 &, const uint4 &raw, float w, float)
{
    const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = fmaf(w, v[j], lo[j]);
}
__device__ __forceinline__ uint4 pack_piece(const f32x4 &lo, const f32x4 &hi, bf16_t)
{
    const bf16x8 v = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                      (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
    return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ uint4 pack_piece(const f32x4 &lo, const f32x4 &, float)
{
    return __builtin_bit_cast(uint4, lo);
}

template <typename T, int CK, int NF>
__global__ __launch_bounds__(256) void deform_kernel(const DeformParams p)
{
    using C = DeformCfg<T, CK, NF>;
    using vec = typename DT<T>::vec;
    constexpr int PSTR = C::PSTR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_s = smem;
    int *tab_i = reinterpret_cast<int *>(smem + C::LDS_S);
    float *tab_w = reinterpret_cast<float *>(smem + C::LDS_S + C::NPIX * 16);
    char *lds_w = smem + C::LDS_S + C::LDS_TAB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int H = p.H, W = p.W;

    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];

    // this thread's pixel for phase A
    const int ay = blockIdx.y * 8 + (tid >> 5), ax = blockIdx.x * 32 + (tid & 31);
    const bool a_in = ay < H && ax < W;
    const float *om = p.om + (((size_t)b * H + (a_in ? ay : 0)) * W + (a_in ? ax : 0)) * 32;
    const char *gx = (const char *)p.x + (size_t)b * H * W * p.x_ps * sizeof(T);

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        // ---- A: sampling table ----
        {
            const int i = tap / 3, j = tap - 3 * i;
            int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
            if (a_in) {
                const float dy = om[2 * tap], dx = om[2 * tap + 1], mk = om[18 + tap];
                // Compare-free formulation: positions are clamped to [-2, size+1] (NaN -> -2), so the
                // int conversions cannot overflow, and corner validity is an integer clamp to {0,1}.
                // A position <= -1 or >= size makes both of its corners invalid or zero-weighted,
                // which is exactly the operator's "outside -> 0" rule.
                const float py = fminf(fmaxf((float)(ay - 1 + i) + dy, -2.0f), (float)(H + 1));
                const float px = fminf(fmaxf((float)(ax - 1 + j) + dx, -2.0f), (float)(W + 1));
                const float fy = floorf(py), fx = floorf(px);
                const int hl = (int)fy, wl = (int)fx, hh = hl + 1, wh = wl + 1;
                const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
                const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
                const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
                o1 = hlc * W + wlc; o2 = hlc * W + whc; o3 = hhc * W + wlc; o4 = hhc * W + whc;
                const int vhl = min(max(hl + 1, 0), 1) * min(max(H - hl, 0), 1);   // 0 <= hl <= H-1
                const int vhh = min(max(hh + 1, 0), 1) * min(max(H - hh, 0), 1);
                const int vwl = min(max(wl + 1, 0), 1) * min(max(W - wl, 0), 1);
                const int vwh = min(max(wh + 1, 0), 1) * min(max(W - wh, 0), 1);
                w1 = mk * (uh * uw) * (float)(vhl * vwl);
                w2 = mk * (uh * lw) * (float)(vhl * vwh);
                w3 = mk * (lh * uw) * (float)(vhh * vwl);
                w4 = mk * (lh * lw) * (float)(vhh * vwh);
            }
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 ti = {o1, o2, o3, o4};
            const f32x4 tw = {w1, w2, w3, w4};
            *reinterpret_cast<i32x4 *>(tab_i + tid * 4) = ti;
            *reinterpret_cast<f32x4 *>(tab_w + tid * 4) = tw;
        }
        // ---- this tap's packed weights ----
        for (int idx = tid; idx < C::WVEC; idx += 256)
            *reinterpret_cast<uint4 *>(lds_w + idx * 16) =
                *reinterpret_cast<const uint4 *>((const char *)p.w + (size_t)tap * C::WTAP + idx * 16);
        __syncthreads();

        // ---- B: gather + blend into the deformed-im2col tile ----
        for (int it = tid; it < C::NPIX * C::PIECES; it += 256) {
            const int pix = it / C::PIECES, pc = it - pix * C::PIECES;
            const int4 o = *reinterpret_cast<const int4 *>(tab_i + pix * 4);
            const float4 w = *reinterpret_cast<const float4 *>(tab_w + pix * 4);
            const unsigned ps = (unsigned)p.x_ps * (unsigned)sizeof(T);  // per-sample plane < 4 GiB (checked on host)
            const uint4 v1 = *reinterpret_cast<const uint4 *>(gx + ((unsigned)o.x * ps + (unsigned)pc * 16u));
            const uint4 v2 = *reinterpret_cast<const uint4 *>(gx + ((unsigned)o.y * ps + (unsigned)pc * 16u));
            const uint4 v3 = *reinterpret_cast<const uint4 *>(gx + ((unsigned)o.z * ps + (unsigned)pc * 16u));
            const uint4 v4 = *reinterpret_cast<const uint4 *>(gx + ((unsigned)o.w * ps + (unsigned)pc * 16u));
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            blend_piece(lo, hi, v1, w.x, T{});
            blend_piece(lo, hi, v2, w.y, T{});
            blend_piece(lo, hi, v3, w.z, T{});
            blend_piece(lo, hi, v4, w.w, T{});
            *reinterpret_cast<uint4 *>(lds_s + pix * PSTR + pc * 16) = pack_piece(lo, hi, T{});
        }
        __syncthreads();

        // ---- C: contraction ----
        const char *xb0 = lds_s + ((wave * 2 + 0) * 32 + r) * PSTR + h * 16;
        const char *xb1 = lds_s + ((wave * 2 + 1) * 32 + r) * PSTR + h * 16;
        const char *wb = lds_w + lane * 16;
#pragma unroll
        for (int kg = 0; kg < C::KG; ++kg) {
            const vec x0 = *reinterpret_cast<const vec *>(xb0 + kg * 32);
            const vec x1 = *reinterpret_cast<const vec *>(xb1 + kg * 32);
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const vec wv = *reinterpret_cast<const vec *>(wb + (kg * NF + n) * 1024);
                mma_kg(acc[0][n], wv, x0);
                mma_kg(acc[1][n], wv, x1);
            }
        }
        // the next tap's phase A/B is VALU- and predicate-heavy: do not run it in the MFMA shadow
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            mfma_retire(acc[0][n]);
            mfma_retire(acc[1][n]);
        }
        __syncthreads();
    }

    // ---- epilogue (no activation: ema_vfi.py:136-138 chains the blocks directly) ----
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int y = blockIdx.y * 8 + wave * 2 + m, x = blockIdx.x * 32 + r;
        if (y >= H || x >= W) continue;
        T *op = reinterpret_cast<T *>(p.out) + (((size_t)b * H + y) * W + x) * p.out_ps;
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = n * 32 + 8 * g + 4 * h;
                if (c0 >= p.cstore) continue;
                store4(op + c0, acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
            }
    }
}

template <typename T, int CK, int NF> static int launch_deform_inst(const DeformParams &p, hipStream_t s)
{
    using C = DeformCfg<T, CK, NF>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&deform_kernel<T, CK, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
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
