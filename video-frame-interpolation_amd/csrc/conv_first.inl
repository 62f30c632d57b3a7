// feat_ext_conv1 (ema_vfi.py:73, conv_block(2 * in_channels, mid_channels)) fused with the concatenation in front of it
// (torch.cat([frame1, frame2], dim=1), :112) for the 16-bit modes at mid_channels = 64: reads the two NCHW fp32 frames as they are
// handed to forward(), writes the 64-channel channels-last activation.  Replaces pack_input (NCHW -> 16-channel pixels: 71 us,
// 413 MB of traffic) + conv3x3<16, 2, 1> (383 us; 6 real of 16 input channels per tap) at B = 8 x 720p.
//
// The layer is a store: 12 B + 12 B read and 128 B written per pixel against 6.9 kFLOP.  K = 6 channels x 9 taps is laid out as
// ten "tap slots" of 8 (6 channels + 2 zeros; slot 9 is zero): K = 16 kg + 8 h + c <-> tap 2 kg + h, channel c - so the B
// operand of lane (r, h) for k-group kg is the 16-byte LDS pixel of tap 2 kg + h exactly as it lies (no shuffles), and a tile
// costs 5 k-groups x 2 fragments x 2 rows = 20 MFMAs per wave instead of 9 x 2 x 2 = 36.  fp32 accumulation groups two taps per
// MFMA instead of one: results agree with the unfused layer to fp32 rounding (<= 1 ulp of the storage type after rounding).
#pragma once
#include "common.h"
#include <mutex>

struct FirstParams {
    const float *f1, *f2;   // NCHW fp32, 3 channels each
    void *out;              // channels-last T, out_ps elements per pixel (64 written)
    const void *w;          // [kg 5][nf 2][lane][16 B]: lane (r, h) holds W[32 nf + r][tap 2 kg + h][channel 0..5], 2 zeros
    const float *bias;      // [64]
    int out_ps, H, W, B, relu;
};

template <typename T>
__global__ __launch_bounds__(256, 3) void conv_first_kernel(const FirstParams p)
{
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPX = IH * IW;
    using vec = typename DT<T>::vec;
    __shared__ __attribute__((aligned(16))) char tile[NPX * 16];
    __shared__ __attribute__((aligned(16))) char ostage[4 * 32 * 144];   // per wave: one output row of 32 pixels, 128 B + 16 B pad each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const size_t plane = (size_t)H * W;

    // weights: 10 fragments, resident in registers for every tile of this workgroup
    vec wf[5][2];
#pragma unroll
    for (int kg = 0; kg < 5; ++kg)
#pragma unroll
        for (int n = 0; n < 2; ++n) wf[kg][n] = *reinterpret_cast<const vec *>((const char *)p.w + (kg * 2 + n) * 1024 + lane * 16);
    const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH, ntiles = ntx * nty * p.B;
#pragma unroll 1
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (ntx * nty), trem = t - b * (ntx * nty);
        const int ty = trem / ntx, tx = trem - ty * ntx;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const float *g1 = p.f1 + (size_t)b * 3 * plane, *g2 = p.f2 + (size_t)b * 3 * plane;
        if (t != (int)blockIdx.x) __syncthreads();   // every wave has read the previous tile
        // ---- stage the tile + halo: one 16-byte pixel = (frame1 c0..c2, frame2 c0..c2, 0, 0) in T; zero outside the image
#pragma unroll
        for (int k = 0; k < (NPX + 255) / 256; ++k) {
            const int q = k * 256 + tid;
            if (q < NPX) {
                const int ly = q / IW, lx = q - ly * IW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                    const size_t o = (size_t)gy * W + gx;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { v[c] = g1[c * plane + o]; v[3 + c] = g2[c * plane + o]; }
                }
                const vec px = {(T)v[0], (T)v[1], (T)v[2], (T)v[3], (T)v[4], (T)v[5], (T)0.0f, (T)0.0f};
                *reinterpret_cast<vec *>(tile + q * 16) = px;
            }
        }
        __syncthreads();
        // one tile row (32 pixels x 64 channels) at a time: 32 accumulator registers, so that four of these workgroups share a
        // SIMD and one's loads / stores run beside another's MFMAs (the layer is a store: 128 B out per pixel)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = wave * 2 + m;
            vec x[5];
#pragma unroll
            for (int kg = 0; kg < 5; ++kg) {
                const int tap = 2 * kg + h < 9 ? 2 * kg + h : 8;       // slot 9: zero weights, any finite data
                const int dy = tap / 3, dx = tap - 3 * dy;
                x[kg] = *reinterpret_cast<const vec *>(tile + ((row + dy) * IW + r + dx) * 16);
            }
            f32x16 acc[2];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // acc_channel(i, h) = (i & 3) + 8 (i >> 2) + 4 h: four runs of four consecutive channels
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.bias + n * 32 + 8 * g + 4 * h);
                    acc[n][4 * g] = b4[0]; acc[n][4 * g + 1] = b4[1]; acc[n][4 * g + 2] = b4[2]; acc[n][4 * g + 3] = b4[3];
                }
#pragma unroll
            for (int kg = 0; kg < 5; ++kg)
#pragma unroll
                for (int n = 0; n < 2; ++n) mma_kg(acc[n], wf[kg][n], x[kg]);
            // ---- store through LDS: the accumulator layout gives every store instruction 32 pixels x 32 bytes (64 partial lines);
            // staged per wave (144-byte pixels: conflict-free) and read back as (8 pixels x 128 bytes) per instruction, a wave
            // writes 1 KiB of consecutive addresses = eight whole 128-byte lines
            {
                typedef __attribute__((ext_vector_type(2))) T pair_t;
                typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
                typedef __attribute__((address_space(3))) char lchar_t;
                lchar_t *stg = (lchar_t *)ostage + wave * (32 * 144);
                const bool relu = p.relu != 0;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        unsigned a[2], c[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float v0 = acc[n][4 * g + 2 * q], v1 = acc[n][4 * g + 2 * q + 1], u0 = acc[n][4 * (g + 1) + 2 * q], u1 = acc[n][4 * (g + 1) + 2 * q + 1];
                            if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                            const pair_t pa = {(T)v0, (T)v1}, pb = {(T)u0, (T)u1};
                            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
                            a[q] = sw[0]; c[q] = sw[1];
                        }
                        // h = 0: channels 8g..8g+7 of fragment n, h = 1: 8(g+1)..8(g+1)+7 (store_frag16, common.h)
                        *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + r * 144 + (n * 32 + 8 * (g + h)) * 2) = u4_t{a[0], a[1], c[0], c[1]};
                    }
                const int y = ty * TH + row;
                char *orow = reinterpret_cast<char *>(p.out) + (((size_t)b * H + y) * W + tx * TW) * p.out_ps * sizeof(T);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = i * 8 + (lane >> 3), ch = lane & 7;
                    const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * 144 + ch * 16);
                    if (y < H && tx * TW + px < W) *reinterpret_cast<u4_t *>(orow + (size_t)px * p.out_ps * sizeof(T) + ch * 16) = v;
                }
            }
        }
    }
}

template <typename T> static int launch_conv_first_t(const FirstParams &p, hipStream_t s)
{
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const long long ntiles = (long long)((p.W + 31) / 32) * ((p.H + 7) / 8) * p.B;
    const long long grid = ntiles < 16LL * ncu ? ntiles : 16LL * ncu;   // three resident per CU (registers), ~5 rounds each
    conv_first_kernel<T><<<(unsigned)grid, 256, 0, s>>>(p);
    return (int)hipGetLastError();
}
