// bf16 modulated deformable conv with the input tile staged in LDS (the "tiled gather with LDS
// halo" of BASELINE.json).  Same operator and fragment scheme as deform.inl; what changes is
// where the bilinear taps are fetched from:
//
//   * a 512-thread workgroup (8 waves = 2 per SIMD, so one wave's gather/blend VALU overlaps its
//     partner's MFMAs) owns a 16x32 output tile; wave w owns rows 2w, 2w+1;
//   * the input tile plus a halo of 1 (3x3 taps) + R (offset reach) + 1 (bilinear) pixels is
//     loaded ONCE, coalesced, into LDS: (19+2R) x (35+2R) pixels x CK bf16 (R = 2: 23 x 39 x
//     160 B = 140 KiB) - HBM/L2 sees each input pixel 1.75x instead of ~36 scattered corner
//     fetches that thrash the 32 KiB L1 (deform.inl v1 measured 5.0 ms per B=8 720p launch);
//   * every lane gathers its own MFMA operand pieces with ds_read_b128; a tap whose four corners
//     do not all lie inside the staged window (|offset| > R) falls back to the global gather of
//     deform.inl for that lane - results are identical either way;
//   * the tap's packed weights sit in one 15 KiB LDS buffer, prefetched through registers
//     (two barriers per tap).
#include "deform.inl"

template <int CK, int NF, int R> struct DeformLdsCfg {
    static constexpr int TROWS = 16, TCOLS = 32;
    static constexpr int TR = TROWS + 3 + 2 * R, TC = TCOLS + 3 + 2 * R;
    static constexpr int PSB = CK * 2;        // bytes per staged pixel
    static constexpr int PIECES = PSB / 16;
    static constexpr int KG = CK / 16;
    static constexpr int WTAP = KG * NF * 1024;
    static constexpr int WVEC = KG * NF * 64;
    static constexpr int WPT = (WVEC + 511) / 512;
    static constexpr int LDS_TILE = TR * TC * PSB;
    static constexpr int LDS_BYTES = LDS_TILE + WTAP;
    static constexpr int KB = (KG % 5 == 0) ? 5 : ((KG % 3 == 0) ? 3 : ((KG % 2 == 0) ? 2 : 1));
    static_assert(LDS_BYTES <= 160 * 1024, "tile + weights do not fit the 160 KiB LDS");
};

template <int CK, int NF, int R>
__global__ __launch_bounds__(512) void deform_lds_kernel(const DeformParams p)
{
    using C = DeformLdsCfg<CK, NF, R>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_x = smem;
    char *lds_w = smem + C::LDS_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int H = p.H, W = p.W;
    const unsigned ps_bytes = (unsigned)p.x_ps * 2u;
    const int ty0 = blockIdx.y * C::TROWS - 1 - R, tx0 = blockIdx.x * C::TCOLS - 1 - R;
    const char *gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;

    // ---- stage the input window (zero outside the image) and tap 0's weights ----
    for (int it = tid; it < C::TR * C::TC * C::PIECES; it += 512) {
        const int pix = it / C::PIECES, pc = it - pix * C::PIECES;
        const int ly = pix / C::TC, lx = pix - ly * C::TC;
        const int gy = ty0 + ly, gxx = tx0 + lx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy >= 0 && gy < H && gxx >= 0 && gxx < W)
            v = *reinterpret_cast<const uint4 *>(gplane + (size_t)(gy * W + gxx) * ps_bytes + pc * 16);
        *reinterpret_cast<uint4 *>(lds_x + pix * C::PSB + pc * 16) = v;
    }
    for (int idx = tid; idx < C::WVEC; idx += 512)
        *reinterpret_cast<uint4 *>(lds_w + idx * 16) = *reinterpret_cast<const uint4 *>((const char *)p.w + idx * 16);

    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];

    const int px_x = blockIdx.x * C::TCOLS + r;
    int py_y[2];
    bool in_img[2];
    const float *om[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        py_y[m] = blockIdx.y * C::TROWS + wave * 2 + m;
        in_img[m] = py_y[m] < H && px_x < W;
        om[m] = p.om + (((size_t)b * H + (in_img[m] ? py_y[m] : 0)) * W + (in_img[m] ? px_x : 0)) * 32;
    }
    const char *gx = gplane + h * 16;
    const char *lx0 = lds_x + h * 16;

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        uint4 wr[C::WPT];
        if (tap < 8) {
#pragma unroll
            for (int i = 0; i < C::WPT; ++i) {
                const int idx = tid + i * 512;
                if (idx < C::WVEC) wr[i] = *reinterpret_cast<const uint4 *>((const char *)p.w + (size_t)(tap + 1) * C::WTAP + idx * 16);
            }
        }
        __syncthreads();  // this tap's weights (and, first time, the staged window) are visible
        const char *wb = lds_w + lane * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            int yc0, yc1, xc0, xc1;
            const SampleTap st = sample_tap(om[m], tap, py_y[m], px_x, H, W, ps_bytes, in_img[m], &yc0, &yc1, &xc0, &xc1);
            // window-local byte offsets of the four corners; valid only when `inside`
            const bool inside = yc0 >= ty0 && yc1 <= ty0 + C::TR - 1 && xc0 >= tx0 && xc1 <= tx0 + C::TC - 1;
            const unsigned l00 = (unsigned)((yc0 - ty0) * C::TC + (xc0 - tx0)) * C::PSB;
            const unsigned l01 = (unsigned)((yc0 - ty0) * C::TC + (xc1 - tx0)) * C::PSB;
            const unsigned l10 = (unsigned)((yc1 - ty0) * C::TC + (xc0 - tx0)) * C::PSB;
            const unsigned l11 = (unsigned)((yc1 - ty0) * C::TC + (xc1 - tx0)) * C::PSB;
            const unsigned lo[4] = {l00, l01, l10, l11};
            const bool all_inside = __all(inside);
#pragma unroll
            for (int k0 = 0; k0 < C::KG; k0 += C::KB) {
                uint4 v[C::KB][4];
                if (all_inside) {
#pragma unroll
                    for (int kk = 0; kk < C::KB; ++kk)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            v[kk][c] = *reinterpret_cast<const uint4 *>(lx0 + lo[c] + (unsigned)((k0 + kk) * 32));
                } else {
#pragma unroll
                    for (int kk = 0; kk < C::KB; ++kk)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (inside) v[kk][c] = *reinterpret_cast<const uint4 *>(lx0 + lo[c] + (unsigned)((k0 + kk) * 32));
                            else v[kk][c] = *reinterpret_cast<const uint4 *>(gx + st.o[c] + (unsigned)((k0 + kk) * 32));
                        }
                }
                bf16x8 xf[C::KB];
#pragma unroll
                for (int kk = 0; kk < C::KB; ++kk) xf[kk] = blend4(v[kk], st.w, bf16_t{});
#pragma unroll
                for (int kk = 0; kk < C::KB; ++kk)
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wb + ((k0 + kk) * NF + n) * 1024);
                        mma_kg(acc[m][n], wv, xf[kk]);
                    }
#pragma unroll
                for (int n = 0; n < NF; ++n) mfma_retire(acc[m][n]);
            }
        }
        if (tap < 8) {
            __syncthreads();  // every wave has finished reading this tap's weights
#pragma unroll
            for (int i = 0; i < C::WPT; ++i) {
                const int idx = tid + i * 512;
                if (idx < C::WVEC) *reinterpret_cast<uint4 *>(lds_w + idx * 16) = wr[i];
            }
        }
    }

#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!in_img[m]) continue;
        bf16_t *op = reinterpret_cast<bf16_t *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
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

template <int CK, int NF, int R> static int launch_deform_lds(const DeformParams &p, hipStream_t s)
{
    using C = DeformLdsCfg<CK, NF, R>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&deform_lds_kernel<CK, NF, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid((p.W + C::TCOLS - 1) / C::TCOLS, (p.H + C::TROWS - 1) / C::TROWS, p.B);
    deform_lds_kernel<CK, NF, R><<<grid, 512, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}
