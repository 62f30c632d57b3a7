// The two planar heads at full resolution - motion_estimation.2 (64 -> 2, the flow: ema_vfi.py:91-92) and reconstruction.2
// (32 -> 3 + tanh, (t + 1) / 2: :105-106, :146) - in the 16-bit modes.  Both read a whole activation (128 / 64 bytes per pixel)
// to write 8 / 12 bytes: they are READS, and ran at 2.3-2.9 TB/s on kernels built for MFMA-bound layers (one lock-step eight-wave
// workgroup per CU with the weights in LDS: the CU waits for its tile's DMA, contracts for a moment, waits again; or the
// tile-per-workgroup kernel with a weight ring and a barrier per tap).  Here: 8 x 32-pixel tiles, the nine taps' weights of ONE
// 16-channel output block resident in REGISTERS (18 or 9 fragments per wave: v_mfma_f32_16x16x32), nothing but the input tile in
// LDS (48 / 27 KiB), so three (five) independent workgroups share a CU and one's DMA runs under the others' MFMAs.
#pragma once
#include "common.h"
#include <mutex>

template <typename T, int CK>
__global__ __launch_bounds__(256, CK == 64 ? 2 : 3) void conv_light_kernel(const ConvParams p)
{
    constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPX = IH * IW;
    constexpr int PIECES = CK * 2 / 16, SP = PIECES + 1, PSTR = SP * 16, K32 = CK / 32;
    constexpr int NSLOT = NPX * SP, NINST = (NSLOT + 63) / 64;
    using vec = typename DT<T>::vec;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    __shared__ __attribute__((aligned(16))) char tile[NINST * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kb = lane >> 4;
    const char *zeros = (const char *)p.zeros;
    const unsigned pixbytes = (unsigned)p.in_ps * 2u;

    // weights of the first 16-channel block: [tap][k32][block (2 packed, the first used)][lane][16 B] (PackDesc::mfma16)
    vec wf[9][K32];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < K32; ++k) wf[t][k] = *reinterpret_cast<const vec *>((const char *)p.w + ((t * K32 + k) * 2) * 1024 + lane * 16);
    float bias4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias4[e] = p.bias[kb * 4 + e];

    const int ntx = (p.Wout + TW - 1) / TW, nty = (p.Hout + TH - 1) / TH, ntiles = ntx * nty * p.B;
    const size_t plane = (size_t)p.Hout * p.Wout;
#pragma unroll 1
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (ntx * nty), trem = t - b * (ntx * nty);
        const int ty = trem / ntx, tx = trem - ty * ntx;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * pixbytes;
        if (t != (int)blockIdx.x) __syncthreads();   // every wave has read the previous tile
        // (an opaque copy of the lane index per tile: left loop-invariant, hipcc hoists every DMA instruction's (pixel, piece)
        // arithmetic in front of the tile loop - ~60 registers held across the MFMA section for nothing)
        int dlane = lane;
        asm volatile("" : "+v"(dlane));
#pragma unroll
        for (int i = 0; i < (NINST + 3) / 4; ++i) {
            const int jn = i * 4 + wave;
            if (jn < NINST) {
                const int sl = jn * 64 + dlane;
                const int pix = sl / SP, pc = sl - pix * SP;
                const int ly = pix / IW, lx = pix - ly * IW;
                const char *src = conv_dma_src(gin, zeros, iy0 + ly, ix0 + lx, pc, p.Hin, p.Win, pixbytes, sl < NSLOT && pc < PIECES);
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(tile + jn * 1024), 16, 0, 0);
            }
        }
        __syncthreads();   // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
        // wave w: rows 2w, 2w + 1; four blocks of 16 pixels; lane (j, kb) holds pixel j, input channels 8 kb .. + 7 (+ 32 k32)
        f32x4 acc[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pb][e] = bias4[e];
        // 9 * K32 steps of 4 reads + 4 MFMAs, operands one step ahead (fenced: left alone hipcc hoists all 72 reads: 244 registers)
        constexpr int NSTEP = 9 * K32;
        vec xq[2][4];
        auto load_step = [&](int st, vec (&xd)[4]) {
            const int tap = st / K32, k = st - tap * K32, dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int q = (wave * 2 + (pb >> 1) + dy) * IW + (pb & 1) * 16 + j + dx;
                xd[pb] = *reinterpret_cast<const vec *>(tile + q * PSTR + (kb + 4 * k) * 16);
            }
        };
        load_step(0, xq[0]);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + 1 < NSTEP) load_step(st + 1, xq[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) mma_k32(acc[pb], wf[st / K32][st % K32], xq[st & 1][pb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // planar fp32 epilogue (conv_epilogue's): <= 4 real channels = the four accumulator registers of the lanes with kb == 0
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int y = ty * TH + wave * 2 + (pb >> 1), x = tx * TW + (pb & 1) * 16 + j;
            if (kb != 0 || y >= p.Hout || x >= p.Wout) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < p.nplanes) {
                    float v = acc[pb][c];
                    if (p.round16) {
                        v = (float)(half_t)v;
                        if (p.epi == EPI_PLANAR_TANH01) v = (float)(half_t)((float)(half_t)tanhf(v) + 1.0f) / 2.0f;
                    } else if (p.epi == EPI_PLANAR_TANH01) v = (tanhf(v) + 1.0f) / 2.0f;
                    p.out_planar[((size_t)b * p.nplanes + c) * plane + (size_t)y * p.Wout + x] = v;
                }
        }
    }
}

template <typename T, int CK> static int launch_conv_light(const ConvParams &p, hipStream_t s)
{
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const long long ntiles = (long long)((p.Wout + 31) / 32) * ((p.Hout + 7) / 8) * p.B;
    const long long grid = ntiles < 12LL * ncu ? ntiles : 12LL * ncu;
    conv_light_kernel<T, CK><<<(unsigned)grid, 256, 0, s>>>(p);
    return (int)hipGetLastError();
}
