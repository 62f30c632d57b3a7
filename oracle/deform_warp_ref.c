/*
 * ORACLE - test infrastructure only, never linked into the product library.
 *
 * Plain-C scalar restatement of the two data-dependent gathers on the
 * EMA-VFI hot path, written independently of oracle/emavfi_oracle.py's
 * vectorised torch version so the two can check each other:
 *
 *   oracle_deform_conv2d  - torchvision.ops.deform_conv2d (DCNv2) as configured
 *                           at /root/reference/src/models/ema_vfi.py:45-51 and
 *                           called at :60.  torchvision is NOT vendored in the
 *                           reference and NOT installed in this image
 *                           (requirements.txt:2, unpinned): PARITY UNPINNED for
 *                           this op; it follows the operator's published
 *                           definition (deformable im2col with the
 *                           "h <= -1 || h >= H -> 0" rule, then a dense
 *                           contraction) and is pinned by the known-answer
 *                           tests in tests/test_oracle_deform.py.
 *   oracle_warp           - EMA_VFI.warp (ema_vfi.py:149-171): grid build,
 *                           normalisation with a true fp32 division, then
 *                           ATen grid_sampler_2d (bilinear, zeros,
 *                           align_corners=True) restated op for op
 *                           (ATen/native/GridSampler.h:27-36 for the
 *                           un-normalisation; cpu/GridSamplerKernel.cpp for
 *                           the interpolation weights).  Pinned against
 *                           torch's own F.grid_sample in tests/test_oracle_warp.py.
 *
 * All tensors are dense NCHW fp32.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stddef.h>

static float dcn_bilinear(const float *p, int H, int W, float h, float w)
{
    if (h <= -1.0f || h >= (float)H || w <= -1.0f || w >= (float)W)
        return 0.0f;
    int hl = (int)floorf(h), wl = (int)floorf(w);
    int hh = hl + 1, wh = wl + 1;
    float lh = h - (float)hl, lw = w - (float)wl;
    float uh = 1.0f - lh, uw = 1.0f - lw;
    float v1 = (hl >= 0 && wl >= 0) ? p[(size_t)hl * W + wl] : 0.0f;
    float v2 = (hl >= 0 && wh <= W - 1) ? p[(size_t)hl * W + wh] : 0.0f;
    float v3 = (hh <= H - 1 && wl >= 0) ? p[(size_t)hh * W + wl] : 0.0f;
    float v4 = (hh <= H - 1 && wh <= W - 1) ? p[(size_t)hh * W + wh] : 0.0f;
    return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
}

/* x [B,C,H,W], offset [B,18,H,W] (2k = dy, 2k+1 = dx, k = 3i+j), mask [B,9,H,W],
 * weight [O,C,3,3], bias [O] or NULL, out [B,O,H,W]. */
void oracle_deform_conv2d(const float *x, const float *offset, const float *mask,
                          const float *weight, const float *bias, float *out,
                          int B, int C, int O, int H, int W)
{
    const size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx) {
                const size_t pix = (size_t)y * W + xx;
                for (int o = 0; o < O; ++o) {
                    double acc = bias ? (double)bias[o] : 0.0;
                    for (int k = 0; k < 9; ++k) {
                        const int i = k / 3, j = k % 3;
                        const float dy = offset[((size_t)b * 18 + 2 * k) * hw + pix];
                        const float dx = offset[((size_t)b * 18 + 2 * k + 1) * hw + pix];
                        const float m = mask[((size_t)b * 9 + k) * hw + pix];
                        const float py = (float)(y - 1 + i) + dy;
                        const float px = (float)(xx - 1 + j) + dx;
                        for (int c = 0; c < C; ++c) {
                            const float v = m * dcn_bilinear(x + ((size_t)b * C + c) * hw, H, W, py, px);
                            acc += (double)weight[(((size_t)o * C + c) * 3 + i) * 3 + j] * (double)v;
                        }
                    }
                    out[((size_t)b * O + o) * hw + pix] = (float)acc;
                }
            }
}

/* frame2 [B,C,H,W], flow [B,2,H,W] (channel 0 = dx, 1 = dy, in pixels), out [B,C,H,W]. */
void oracle_warp(const float *frame2, const float *flow, float *out, int B, int C, int H, int W)
{
    const size_t hw = (size_t)H * W;
    const float wden = (float)((W - 1) > 1 ? (W - 1) : 1);
    const float hden = (float)((H - 1) > 1 ? (H - 1) : 1);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = (size_t)y * W + x;
                /* ema_vfi.py:162-166 */
                const float vx = (float)x + flow[((size_t)b * 2 + 0) * hw + pix];
                const float vy = (float)y + flow[((size_t)b * 2 + 1) * hw + pix];
                const float gx = 2.0f * vx / wden - 1.0f;
                const float gy = 2.0f * vy / hden - 1.0f;
                /* grid_sampler_unnormalize, align_corners=True */
                const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
                const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
                const float xw = floorf(ix), yn = floorf(iy);
                const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
                const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
                /* NaN / inf coordinates fail every range test below -> 0 */
                const int okx0 = (xw >= 0.0f && xw <= (float)(W - 1));
                const int okx1 = (xw + 1.0f >= 0.0f && xw + 1.0f <= (float)(W - 1));
                const int oky0 = (yn >= 0.0f && yn <= (float)(H - 1));
                const int oky1 = (yn + 1.0f >= 0.0f && yn + 1.0f <= (float)(H - 1));
                for (int c = 0; c < C; ++c) {
                    const float *p = frame2 + ((size_t)b * C + c) * hw;
                    float acc = 0.0f;
                    if (oky0 && okx0) acc += p[(size_t)(int)yn * W + (int)xw] * nw;
                    if (oky0 && okx1) acc += p[(size_t)(int)yn * W + (int)xw + 1] * ne;
                    if (oky1 && okx0) acc += p[(size_t)((int)yn + 1) * W + (int)xw] * sw;
                    if (oky1 && okx1) acc += p[(size_t)((int)yn + 1) * W + (int)xw + 1] * se;
                    out[((size_t)b * C + c) * hw + pix] = acc;
                }
            }
}
