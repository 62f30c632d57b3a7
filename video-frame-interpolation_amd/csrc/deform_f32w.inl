// Modulated deformable 3x3 convolution (DCNv2) in exact fp32 at the reference width (67 -> 67 channels) on an LDS-staged
// window - the fp32 counterpart of deform_pack3.inl.  Replaces torchvision.ops.DeformConv2d (ema_vfi.py:45-51, called at :60)
// in the parity mode (EMAVFI_F32) and in the autocast-policy mode (EMAVFI_AMP16: torchvision's deform_conv2d runs in fp32 there).
//
// deform_kernel<float, 80, 3> (deform.inl) gathers every corner from global memory (L1/L2) and contracts on
// v_mfma_f32_32x32x2_f32 with N = 96 and K = 72 issued for 67 x 67 real: 12.8 ms per launch at B = 8 x 720p = 0.30 of the fp32
// matrix peak, 40 % of the parity mode's step and 75 % of the autocast-policy mode's.  Here:
//   * v_mfma_f32_16x16x4_f32: 5 output blocks of 16 (80 issued channels instead of 96), K in steps of 4: 36 + 32 = 68 issued for 67
//     real (round 6: the second half's leftover step - channels 68..71, zero weights - is no longer issued: 85 instead of 90 MFMAs per
//     (tap, block); the kernel is matrix-pipe-bound at the clock the board holds, section 3.3b of DESIGN.md);
//   * the input window (tile + halo of 1 tap + R offset reach + 1 bilinear, zero outside the image) is staged in LDS by
//     global->LDS DMA, one HALF of the channels at a time (36 fp32 channels = 144-byte pixels = 9 sixteen-byte slots, an odd
//     slot stride: 76 KiB, two workgroups per CU); the nine taps run once per half;
//   * lane (j, kb) of a 16-pixel block reads ONE 16-byte piece per corner (channels 16 G + 4 kb .. + 3 of group G) and feeds
//     four MFMA steps from it: step t of the group contracts the channel set {16 G + 4 kb' + t}, the packed weights carry
//     that permutation (misc_kernels.hip, pack_deform_f32w_kernel); the 4 leftover channels of a half are one step on 4-byte reads;
//   * a tap's four corners are one window address + three immediates, zero padding comes from the window;
//   * samples that leave the window (|offset| > R near the tile edge: rare) contribute nothing in the tap loops and are
//     added by a fix-up loop from global memory (clamped corners, validity-masked weights - what deform_kernel computes).
// Arithmetic: sampling positions, corner weights and the blend are the fp32 expressions of deform.inl (sample_tap_vals,
// blend4(float)); the contraction is an exact fp32 FMA chain per output (MFMA f32), in a different channel order.
#pragma once
#include "deform.inl"

struct F32W {
    static constexpr int R = 2, TROWS = 16, TCOLS = 16, WAVES = 4, THREADS = 256;
    static constexpr int TR = TROWS + 3 + 2 * R, TC = TCOLS + 3 + 2 * R;                  // 23 x 23 window pixels
    static constexpr int HC = 36, SP = HC * 4 / 16, PSB = SP * 16, ROWB = TC * PSB;       // 36 fp32 channels per half: 144 B pixels
    static constexpr int LDS_BYTES = TR * ROWB;                                           // 76 176 B: two workgroups per CU
    static constexpr int SEG_PX = 7, SEG_BYTES = SEG_PX * PSB, LAST_PX = TC - 3 * SEG_PX;
    // packed weights: [tap][half][cout block 5]{ group 0: lane x 4 floats | group 1: lane x 4 floats | leftover: lane x 1 float }
    static constexpr int CB_BYTES = 2 * 1024 + 256, HALF_BYTES = 5 * CB_BYTES, TAP_BYTES = 2 * HALF_BYTES, W_BYTES = 9 * TAP_BYTES;
    static_assert((SP & 1) == 1 && SEG_PX * SP <= 64 && LAST_PX > 0, "odd slot stride, row segments");
    static_assert(ROWB + PSB + 8 * 16 + 15 < 65536 && 2 * LDS_BYTES <= 160 * 1024, "immediates, two workgroups per CU");
};

__device__ __forceinline__ void mma_f32_k4(f32x4 &acc, float w, float x) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, acc, 0, 0, 0); }
// EMAVFI_F32X3 (round 6): the same contraction as a three-term f16 split on v_mfma_f32_16x16x16_f16.  Lane (j, kb) of the B operand holds K
// elements 4 kb .. 4 kb + 3 - exactly the four blended channels {16 G + 4 kb + t} this kernel's lanes already hold, and lane (i, kb) of the A
// operand the four weights the fp32 form feeds to its four MFMA steps: ONE 8-cycle MFMA per term instead of four 32-cycle ones.
// hi = f16(v) (round toward zero is as good as any: lo takes the rest), lo = f16(v - hi): 22 bits of each operand, exact products, fp32 sums.
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
struct HiLo { f16x4_t hi, lo; };
__device__ __forceinline__ HiLo split_f16x4(const f32x4 v)
{
    HiLo r;
    r.hi = f16x4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    r.lo = f16x4_t{(half_t)(v[0] - (float)r.hi[0]), (half_t)(v[1] - (float)r.hi[1]), (half_t)(v[2] - (float)r.hi[2]), (half_t)(v[3] - (float)r.hi[3])};
    return r;
}
__device__ __forceinline__ void mma_x3(f32x4 &acc, const HiLo &w, const HiLo &x)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi, x.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(w.lo, x.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi, x.lo, acc, 0, 0, 0);
}

template <bool X3>
__global__ __launch_bounds__(256, 2) void deform_f32w_kernel(const DeformParams p)
{
    using C = F32W;
    constexpr int R = C::R;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((address_space(3))) const char lds_cchar_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_cchar_t *lds_r = (lds_cchar_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kb = lane >> 4;
    const int H = p.H, W = p.W;
    const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;

    // ---- tile of this workgroup (XCD-aware order: deform_pack.inl)
    const int ntx = (W + C::TCOLS - 1) / C::TCOLS, nty = (H + C::TROWS - 1) / C::TROWS, nt = ntx * nty;
    int tile_x, tile_y, b;
    {
        constexpr int SROWS = 4;
        const int nwg = gridDim.x, grp = blockIdx.x & 7, kk = blockIdx.x >> 3, qq = nwg >> 3, rr = nwg & 7;
        const int wg = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
        b = wg / nt;
        const int t = wg - b * nt, strip = t / (SROWS * ntx), tt = t - strip * SROWS * ntx;
        const int rows = min(SROWS, nty - strip * SROWS);
        tile_x = tt / rows;
        tile_y = strip * SROWS + (tt - tile_x * rows);
    }
    const unsigned ps_bytes = (unsigned)p.x_ps * 4u;
    const int ty0 = tile_y * C::TROWS - 1 - R, tx0 = tile_x * C::TCOLS - 1 - R;
    const char *gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;
    const char *zeros = (const char *)p.zeros;
    const char *wbase_g = (const char *)p.w;

    // ---- this lane's pixel in each of its wave's four 16-pixel blocks (block = one tile row), and its offsets / masks
    const int px_x = tile_x * C::TCOLS + j;
    int py_y[4];
    bool in_img[4];
    const float *om[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        py_y[blk] = tile_y * C::TROWS + wave * 4 + blk;
        in_img[blk] = py_y[blk] < H && px_x < W;
        om[blk] = p.om + (((size_t)b * H + (in_img[blk] ? py_y[blk] : 0)) * W + (in_img[blk] ? px_x : 0)) * 32;
    }
    const float fx_base = (float)(px_x - 1), fy_max = (float)(H + 1), fx_max = (float)(W + 1);

    // accumulators: D[16 c + 4 kb + e][pixel j of block blk]
    f32x4 acc[5][4];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][blk][e] = p.bias[16 * c + 4 * kb + e];

    // sampling geometry of (tap, block): window byte offset of the top-left corner (clamped into the window), the four
    // corner weights (bilinear x mask), and whether the sample leaves the window.  fp32, compare-free clamps (deform.inl).
    struct Geo { unsigned base; float w[4]; bool out; float py, px, mk; };
    auto geometry = [&](int tap, int blk, float dy, float dx, float mkraw) {
        Geo g;
        const int ti = tap / 3, tj = tap - 3 * ti;
        g.mk = in_img[blk] ? mkraw : 0.0f;
        g.py = fminf(fmaxf((float)(py_y[blk] - 1 + ti) + dy, -2.0f), fy_max);
        g.px = fminf(fmaxf((fx_base + (float)tj) + dx, -2.0f), fx_max);
        const float fy = floorf(g.py), fx = floorf(g.px);
        const int hl = (int)fy, wl = (int)fx;
        const float lh = g.py - fy, lw = g.px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
        const int ly0 = hl - ty0, lx0 = wl - tx0;
        const bool inside = (unsigned)ly0 <= (unsigned)(C::TR - 2) && (unsigned)lx0 <= (unsigned)(C::TC - 2);
        g.out = !inside && in_img[blk];
        const float keep = g.out ? 0.0f : 1.0f;   // parked for the fix-up loop: contributes nothing here
        g.w[0] = g.mk * (uh * uw) * keep; g.w[1] = g.mk * (uh * lw) * keep; g.w[2] = g.mk * (lh * uw) * keep; g.w[3] = g.mk * (lh * lw) * keep;
        g.base = __umul24((unsigned)min(max(ly0, 0), C::TR - 2), (unsigned)C::ROWB) + __umul24((unsigned)min(max(lx0, 0), C::TC - 2), (unsigned)C::PSB);
        return g;
    };
    constexpr int OFF[4] = {0, C::PSB, C::ROWB, C::ROWB + C::PSB};
    unsigned long long fb_mask = 0;   // wave-uniform: bit tap * 4 + blk set when a lane's sample of that (tap, block) left the window

    const bool dact = wave < 3 ? lane < C::SEG_PX * C::SP : lane < C::LAST_PX * C::SP;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();   // every wave has finished reading the first half's window
        // ---- DMA this half's window, row by row: wave w fetches pixels [7w, 7w + 7) (wave 3: 2 pixels) x 9 pieces of every row
        {
            const int dp = lane / C::SP, dpc = lane - dp * C::SP;
            const int dgx = tx0 + wave * C::SEG_PX + dp;
            const bool dcol = (unsigned)dgx < (unsigned)W;
            const long long pix0 = (long long)ty0 * W + dgx;
            const char *src = gplane + pix0 * (long long)ps_bytes + half * C::PSB + dpc * 16;
            const unsigned inc = (unsigned)W * ps_bytes;
            if (dact) {
#pragma unroll
                for (int ly = 0; ly < C::TR; ++ly) {
                    const bool ok = dcol && (unsigned)(ty0 + ly) < (unsigned)H;
                    const char *s = ok ? src : zeros;
                    __builtin_amdgcn_global_load_lds((gptr_t *)s, (lptr_t *)(smem + ly * C::ROWB + wave * C::SEG_BYTES), 16, 0, 0);
                    src += inc;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window DMA has landed (hipcc emits this wait today; not relied upon - ADVICE r5)
        __syncthreads();

        // offsets / masks of a tap are fetched one tap ahead (an L2 round trip in front of every tap's geometry otherwise)
        float omv[4][3];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { omv[blk][0] = om[blk][0]; omv[blk][1] = om[blk][1]; omv[blk][2] = om[blk][18]; }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // this (tap, half)'s A operands: lane (i = lane & 15, kb) holds W[16 c + i][half * 36 + 16 G + 4 kb + t], t = 0..3
            const char *wt = wbase_g + (size_t)tap * C::TAP_BYTES + half * C::HALF_BYTES;
            f32x4 a16[5][2];
            float a4[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                a16[c][0] = *reinterpret_cast<const f32x4 *>(wt + c * C::CB_BYTES + lane16);
                a16[c][1] = *reinterpret_cast<const f32x4 *>(wt + c * C::CB_BYTES + 1024 + lane16);
                a4[c] = half ? 0.0f : *reinterpret_cast<const float *>(wt + c * C::CB_BYTES + 2048 + lane4);
            }
            // EMAVFI_F32X3: the tap's weights as f16 (hi, lo) pairs; the leftover channel of lane (i, kb) sits at K = 4 kb of a K = 16 step
            // (kept as one packed dword per output block: hi | lo << 16).  The x3 model's blob carries them already split
            // (pack_deform_f32w_kernel, PackDesc::x3): a lane's 16 bytes are {hi[4], lo[4]} f16 instead of four floats
            HiLo w16[X3 ? 5 : 1][X3 ? 2 : 1];
            unsigned w4p[X3 ? 5 : 1];
            if constexpr (X3) {
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    w16[c][0] = __builtin_bit_cast(HiLo, a16[c][0]);
                    w16[c][1] = __builtin_bit_cast(HiLo, a16[c][1]);
                    w4p[c] = __float_as_uint(a4[c]);
                }
            }
            Geo g[4];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                g[blk] = geometry(tap, blk, omv[blk][0], omv[blk][1], omv[blk][2]);
                if (half == 0 && __any(g[blk].out)) fb_mask |= 1ull << (tap * 4 + blk);
            }
            {
                const int tn = tap < 8 ? tap + 1 : 8;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) { omv[blk][0] = om[blk][2 * tn]; omv[blk][1] = om[blk][2 * tn + 1]; omv[blk][2] = om[blk][18 + tn]; }
            }
            // units per tap: (block, group 0 | group 1 | leftover) - the leftover (channels 32..35 of the half) exists in the first half only
            // (channels 68..71 are padding); corner reads one unit ahead of their blend + MFMAs
            f32x4 vq[2][4];
            auto issue = [&](int u, f32x4 (&v)[4]) {
                const int blk = u / 3, part = u - 3 * blk;
                if (part == 2 && half) return;
                const unsigned a = g[blk].base + (part < 2 ? (unsigned)(part * 64 + kb * 16) : (unsigned)(128 + kb * 4));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (part < 2) v[c] = *reinterpret_cast<__attribute__((address_space(3))) const f32x4 *>(lds_r + a + OFF[c]);
                    else v[c][0] = *reinterpret_cast<__attribute__((address_space(3))) const float *>(lds_r + a + OFF[c]);
                }
            };
            issue(0, vq[0]);
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int blk = u / 3, part = u - 3 * blk;
                if (u + 1 < 12) issue(u + 1, vq[(u + 1) & 1]);
                const f32x4 (&v)[4] = vq[u & 1];
                const float (&w)[4] = g[blk].w;
                if (part < 2) {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = fmaf(w[c], v[c][e], x[e]);
                    if constexpr (X3) {
                        const HiLo xs = split_f16x4(x);
#pragma unroll
                        for (int c = 0; c < 5; ++c) mma_x3(acc[c][blk], w16[c][part], xs);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int c = 0; c < 5; ++c) mma_f32_k4(acc[c][blk], a16[c][part][t], x[t]);
                    }
                } else if (!half) {   // (wave-uniform)
                    float x = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) x = fmaf(w[c], v[c][0], x);
                    if constexpr (X3) {
                        const HiLo xs = split_f16x4(f32x4{x, 0.0f, 0.0f, 0.0f});
#pragma unroll
                        for (int c = 0; c < 5; ++c) {
                            typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
                            HiLo ws;
                            ws.hi = __builtin_bit_cast(f16x4_t, u2_t{w4p[c] & 0xffffu, 0u});
                            ws.lo = __builtin_bit_cast(f16x4_t, u2_t{w4p[c] >> 16, 0u});
                            mma_x3(acc[c][blk], ws, xs);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 5; ++c) mma_f32_k4(acc[c][blk], a4[c], x);
                    }
                }
            }
        }
    }

    // ---- fix-up: the marked (tap, block) samples that left the window, gathered from global memory with clamped corners and
    // validity-masked weights; every other lane takes part with zero weights
    if (__builtin_expect(fb_mask != 0, 0)) {
#pragma unroll 1
        for (unsigned long long left = fb_mask; left != 0; left &= left - 1) {
            const int bit = __builtin_ctzll(left), tap = bit >> 2, blkr = bit & 3;
            // (registers cannot be indexed by a runtime value: select the block's quantities)
            int yb = py_y[0]; bool inb = in_img[0]; const float *omb = om[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (blkr == q) { yb = py_y[q]; inb = in_img[q]; omb = om[q]; }
            const int ti = tap / 3, tj = tap - 3 * ti;
            const float dy = omb[2 * tap], dx = omb[2 * tap + 1], mk = inb ? omb[18 + tap] : 0.0f;
            const float py = fminf(fmaxf((float)(yb - 1 + ti) + dy, -2.0f), fy_max);
            const float px = fminf(fmaxf((fx_base + (float)tj) + dx, -2.0f), fx_max);
            const float fy = floorf(py), fx = floorf(px);
            const int hl = (int)fy, wl = (int)fx, hh = hl + 1, wh = wl + 1;
            const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
            const int ly0 = hl - ty0, lx0 = wl - tx0;
            const bool inside = (unsigned)ly0 <= (unsigned)(C::TR - 2) && (unsigned)lx0 <= (unsigned)(C::TC - 2);
            const bool need = !inside && inb;
            const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
            const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
            const bool vhl = (unsigned)hl < (unsigned)H, vhh = (unsigned)hh < (unsigned)H;
            const bool vwl = (unsigned)wl < (unsigned)W, vwh = (unsigned)wh < (unsigned)W;
            const float w[4] = {need && vhl && vwl ? mk * (uh * uw) : 0.0f, need && vhl && vwh ? mk * (uh * lw) : 0.0f,
                                need && vhh && vwl ? mk * (lh * uw) : 0.0f, need && vhh && vwh ? mk * (lh * lw) : 0.0f};
            const unsigned r0 = __umul24((unsigned)hlc, (unsigned)W), r1 = __umul24((unsigned)hhc, (unsigned)W);
            const unsigned o[4] = {(unsigned)__umul24(r0 + wlc, ps_bytes), (unsigned)__umul24(r0 + whc, ps_bytes), (unsigned)__umul24(r1 + wlc, ps_bytes),
                                   (unsigned)__umul24(r1 + whc, ps_bytes)};
            f32x4 accb[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const char *wt = wbase_g + (size_t)tap * C::TAP_BYTES + half * C::HALF_BYTES;
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    if (part == 2 && half) continue;   // channels 68..71: padding
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (need) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const char *src = gplane + o[c] + half * C::PSB + (part < 2 ? part * 64 + kb * 16 : 128 + kb * 4);
                            if (part < 2) {
                                const f32x4 v = *reinterpret_cast<const f32x4 *>(src);
#pragma unroll
                                for (int e = 0; e < 4; ++e) x[e] = fmaf(w[c], v[e], x[e]);
                            } else {
                                x[0] = fmaf(w[c], *reinterpret_cast<const float *>(src), x[0]);
                            }
                        }
                    }
                    HiLo xs;
                    if constexpr (X3) xs = split_f16x4(x);
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        if (part < 2) {
                            const f32x4 a = *reinterpret_cast<const f32x4 *>(wt + c * C::CB_BYTES + part * 1024 + lane16);
                            if constexpr (X3) {
                                mma_x3(accb[c], __builtin_bit_cast(HiLo, a), xs);
                            } else {
#pragma unroll
                                for (int t = 0; t < 4; ++t) mma_f32_k4(accb[c], a[t], x[t]);
                            }
                        } else {
                            const float a1 = *reinterpret_cast<const float *>(wt + c * C::CB_BYTES + 2048 + lane4);
                            if constexpr (X3) {
                                typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
                                const unsigned pk = __float_as_uint(a1);
                                HiLo ws;
                                ws.hi = __builtin_bit_cast(f16x4_t, u2_t{pk & 0xffffu, 0u});
                                ws.lo = __builtin_bit_cast(f16x4_t, u2_t{pk >> 16, 0u});
                                mma_x3(accb[c], ws, xs);
                            } else {
                                mma_f32_k4(accb[c], a1, x[0]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (blkr == q) acc[c][q] += accb[c];
        }
    }

    // ---- epilogue (no activation: ema_vfi.py:136-138 chains the blocks directly): lane (j, kb) holds channels 16 c + 4 kb .. + 3
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        if (!in_img[blk]) continue;
        float *op = reinterpret_cast<float *>(p.out) + (((size_t)b * H + py_y[blk]) * W + px_x) * p.out_ps;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (16 * c + 4 * kb < p.cstore) *reinterpret_cast<f32x4 *>(op + 16 * c + 4 * kb) = acc[c][blk];
        if (p.out16) {   // EMAVFI_AMP16: the fp16 rounding of the same values (round-to-nearest-even, as Tensor.half() does), in the same launch
            half_t *oh = reinterpret_cast<half_t *>(p.out16) + (((size_t)b * H + py_y[blk]) * W + px_x) * p.out16_ps;
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (16 * c + 4 * kb < p.cstore) {
                    store4(oh + 16 * c + 4 * kb, acc[c][blk][0], acc[c][blk][1], acc[c][blk][2], acc[c][blk][3]);
                    if (p.out16_lo_off > 0) {   // EMAVFI_F32X3: the lo half, so that hi + lo carries 22 bits of the fp32 value
                        float l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) l[e] = acc[c][blk][e] - (float)(half_t)acc[c][blk][e];
                        store4(oh + p.out16_lo_off + 16 * c + 4 * kb, l[0], l[1], l[2], l[3]);
                    }
                }
        }
    }
}

static int launch_deform_f32w(const DeformParams &p, hipStream_t s)
{
    using C = F32W;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    static PerDeviceOnce once3;
    const bool x3 = p.x3 != 0;   // EMAVFI_F32X3: the contraction as a three-term f16 split; the blob carries the weights as f16 (hi, lo) pairs
    if (const hipError_t e_ = x3 ? set_lds_limit(once3, reinterpret_cast<const void *>(&deform_f32w_kernel<true>), C::LDS_BYTES)
                                 : set_lds_limit(once, reinterpret_cast<const void *>(&deform_f32w_kernel<false>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const long long nwg = (long long)((p.W + C::TCOLS - 1) / C::TCOLS) * ((p.H + C::TROWS - 1) / C::TROWS) * p.B;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    if (x3) deform_f32w_kernel<true><<<(unsigned)nwg, C::THREADS, C::LDS_BYTES, s>>>(p);
    else deform_f32w_kernel<false><<<(unsigned)nwg, C::THREADS, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}
