// conv_ring_first.inl - cat(frame1, frame2) + feat_ext_conv1 + ReLU + feat_ext_blocks.conv_block_0 + ReLU (ema_vfi.py:112-116,
// 73-76) in ONE launch for the 16-bit modes at mid_channels = 64.  Included by conv3x3_bf16.hip / conv3x3_f16.hip.
//
// conv_first.inl is a store (1.48 GB of HBM traffic for 6.9 kFLOP per pixel) and the 64 -> 64 layer behind it reads that tensor
// straight back.  Here conv_ring.inl's walk down a strip has TWO stages, and the 64-channel tensor between them exists only as
// an LDS ring of four rows:
//   stage A (step t: row t of feat_ext_conv1): the strip's 64 columns x 64 channels from a ring of four frame rows - 16-byte
//     pixels (frame1 c0..c2, frame2 c0..c2, 0, 0) converted from the NCHW fp32 frames, global loads issued two steps before the
//     row is written to LDS - with conv_first.inl's K layout (ten tap slots of 8: 5 MFMAs per wave and row); ReLU, rounded
//     to T, ZERO outside the image (the next convolution's padding), into the row ring (144-byte pixels: conflict-free);
//   stage B (the same step: row t - 2 of conv_block_0): conv3x3_ring_kernel's main loop on that ring, weights of the wave's
//     fragment in 144 VGPRs, outputs through the double-buffered LDS row, whole lines per store.
// A strip yields 62 output columns for 64 computed ones and a segment computes two extra stage-A rows.  No LDS-DMA at all (the
// frames are 24 B per pixel): ordinary loads and stores, the compiler's own waits.  Stage A's arithmetic is conv_first_kernel's
// and stage B's is conv3x3_ring_kernel's, operation for operation: the result is bit-identical to the two launches.
#pragma once
#include "conv_first.inl"

template <typename T> struct RingFirstCfg {
    static constexpr int TW = 64, TWO = TW - 2, PSTR = 144, MID = TW * PSTR, NMID = 4, STG = TW * PSTR;
    static constexpr int FW = TW + 2, FROW = FW * 16, NFR = 4;
    static constexpr int MID_OFF = 0, STG_OFF = NMID * MID, FR_OFF = STG_OFF + 2 * STG, BIAS_OFF = FR_OFF + NFR * FROW, WA_OFF = BIAS_OFF + 512;
    static constexpr int LDS_BYTES = WA_OFF + 5 * 2 * 1024;   // stage A's ten weight fragments [kg][fragment][lane][8]
    static_assert(sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024, "16-bit types; two workgroups per CU");
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_ringfirst_kernel(const FirstParams fp, const ConvParams p, const int nseg, const int seg_rows)
{
    using C = RingFirstCfg<T>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    typedef __attribute__((ext_vector_type(2))) T pair_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int frag = wave & 1, cb = wave >> 1;
    const int H = p.Hout, W = p.Wout;
    const size_t plane = (size_t)H * W;
    const int ntx = (W + C::TWO - 1) / C::TWO, nstrip = ntx * p.B;
    const bool relu = p.epi == EPI_RELU;

    // ---- this wave's fragment of both layers' weights
    vec wf[9][4];
    {
        const char *wb = (const char *)p.w + frag * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wf[t][kg] = *reinterpret_cast<const vec *>(wb + (t * 4 + kg) * 2048);
    }
    // stage A's weights stay in LDS (five more reads per row; 20 registers for the second set of frame values in flight)
    for (int i = tid; i < 5 * 2 * 64; i += 256) reinterpret_cast<u4_t *>(smem + C::WA_OFF)[i] = reinterpret_cast<const u4_t *>(fp.w)[i];
    if (tid < 32) reinterpret_cast<f32x4 *>(smem + C::BIAS_OFF)[tid] = reinterpret_cast<const f32x4 *>(tid < 16 ? fp.bias : p.bias)[tid & 15];
    for (int i = tid; i < C::NFR * C::FROW / 16; i += 256) reinterpret_cast<u4_t *>(smem + C::FR_OFF)[i] = u4_t{0u, 0u, 0u, 0u};   // the frame pixels' pad channels
    // stage A's five operand reads: tap slot 2 kg + h of this lane half (slot 9: zero weights, any finite data)
    int ady[5], adx[5];
#pragma unroll
    for (int kg = 0; kg < 5; ++kg) {
        const int tap = 2 * kg + h < 9 ? 2 * kg + h : 8;
        ady[kg] = tap / 3;
        adx[kg] = tap - 3 * ady[kg];
    }

    RING_STAMP_DECL;
    RingWork work(nstrip, H, nseg, seg_rows);
    int strip, ys, ye;
#pragma unroll 1
    while (work.next(strip, ys, ye)) {
        const int b = strip / ntx, tx = strip - b * ntx;
        const int a0 = ys - 1;                    // first stage-A row; its ring slot is 0
        const int ox0 = tx * C::TWO - 1;          // image column of stage A's column 0; frame-ring column 0 is ox0 - 1
        const float *g1 = fp.f1 + (size_t)b * 3 * plane, *g2 = fp.f2 + (size_t)b * 3 * plane;
        // one frame row: thread t loads TWO values - channel pair (t % 3) of pixel min(t / 3, 65): (f1 c0, f1 c1) | (f1 c2, f2 c0) |
        // (f2 c1, f2 c2) - UNCONDITIONALLY, from coordinates clamped into the image (a branch around the loads makes hipcc's vmcnt
        // bookkeeping pessimistic: it then waits for them at the next register reuse instead of at frame_put) ...
        // (round 5, tools/ring_stamps.py: every thread used to load all six values of pixel min(t, 65) and threads < 66 wrote whole
        // pixels - waves 0 and 1 carried the conversion and the LDS writes, waves 2 and 3 loaded 6 values for nothing and then waited
        // ~180 cycles per step at the barrier.  Same values, same conversion: still bit-identical to the two launches.)
        const int fpix = min(tid / 3, C::FW - 1), fpart = tid - 3 * (tid / 3);
        const int fgx = ox0 - 1 + fpix;
        const int fgxc = min(max(fgx, 0), W - 1);
        const float *fpa = (fpart == 0 ? g1 : fpart == 1 ? g1 + 2 * plane : g2 + plane) + fgxc;
        const float *fpb = (fpart == 0 ? g1 + plane : fpart == 1 ? g2 : g2 + 2 * plane) + fgxc;
        auto frame_load = [&](int gy, float (&v)[2]) {
#if defined(EMAVFI_RF_ABL) && (EMAVFI_RF_ABL & 1)   // timing-only: no frame loads
            for (int c = 0; c < 2; ++c) v[c] = (float)gy * 0.001f + c;
            return;
#endif
            const size_t o = (size_t)min(max(gy, 0), H - 1) * W;
            v[0] = fpa[o]; v[1] = fpb[o];
        };
        // ... and threads < 3 x 66 write their pair (zero outside the image) as 4 bytes of the 16-byte pixel in ring slot (gy - a0 + 1) & 3
        // (the pixel's last 4 bytes - channels 6, 7 - are zeroed once per kernel, below)
        auto frame_put = [&](int gy, const float (&v)[2]) {
            const bool in = (unsigned)gy < (unsigned)H && (unsigned)fgx < (unsigned)W;
            if (tid < 3 * C::FW) {
                const pair_t px = {(T)(in ? v[0] : 0.0f), (T)(in ? v[1] : 0.0f)};
                *reinterpret_cast<unsigned *>(smem + C::FR_OFF + ((gy - a0 + 1) & 3) * C::FROW + fpix * 16 + fpart * 4) = __builtin_bit_cast(unsigned, px);
            }
        };
        __syncthreads();   // the previous item's last reads of the rings (and the bias / stage-A weight tables' writes)
        float fold[2];     // frame row t + 2, loaded during step t - 1, written at the end of step t: two steps for the round trip
        {
            float v[2];
#pragma unroll 1
            for (int k = -1; k <= 1; ++k) { frame_load(a0 + k, v); frame_put(a0 + k, v); }
            frame_load(a0 + 2, fold);
        }
        const int xg = ox0 + cb * 32 + r;                 // image column of this lane's stage-A pixel
        const int oc = cb * 32 + r, ox = tx * C::TWO + oc;   // this lane's stage-B column in the strip / in the image
        (void)ox;
        char *obase = reinterpret_cast<char *>(p.out) + (((size_t)b * H * W + (size_t)tx * C::TWO) * p.out_ps + p.out_coff) * sizeof(T);
        const int npx = min(C::TWO, W - tx * C::TWO);
        unsigned soff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
            soff[i] = (px < npx && ch * 8 < p.cstore) ? (unsigned)px * (unsigned)p.out_ps * (unsigned)sizeof(T) + ch * 16u : 0x80000000u;
        }
        // always two store instructions (buffer stores: lanes outside the image, or !real, are dropped by the range check - no branch)
        auto store_row = [&](int y, bool real) {
            lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG;
            char *orow = obase + (size_t)(real ? y : ys) * W * p.out_ps * sizeof(T);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
                const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * C::PSTR + ch * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, soff[i], 0, 0);
            }
        };
#pragma unroll 1
        for (int t = a0; t <= ye + 1; ++t) {
            RING_STAMP(ts0);
            __syncthreads();   // stage A's row t - 1, stage B's staged row t - 3 and frame row t + 1 are visible
            RING_STAMP(ts1);
            float fnew[2];
            frame_load(t + 3, fnew);
            store_row(t - 3, t - 3 >= ys);
            RING_STAMP(ts2);
            // Round 5 (the duty-cycle law of DESIGN.md section 4.2): the step used to run stage A (operand reads -> five dependent MFMAs ->
            // epilogue) and then stage B, each a serial chain of its own.  Now B contracts first, A's five MFMAs queue behind B's last ones
            // (its first operands are read under them), B's epilogue runs while A's chain is in the pipe, A's epilogue last.  Same
            // arithmetic in the same order per stage: still bit-identical to the two launches.
#if defined(EMAVFI_RF_ABL) && (EMAVFI_RF_ABL & 2)   // timing-only: no stage A
            const bool doA = t <= ye && t == -12345;
#else
            const bool doA = t <= ye;
#endif
            const int yb = t - 2;
            const bool doB = yb >= ys;
            f32x16 acc[2];
            if (doB) {
                // ---- stage B: row yb of conv_block_0 from stage-A rows yb - 1 .. yb + 1 (conv3x3_ring_kernel's main loop)
                {
                    const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIAS_OFF + 256 + (frag * 32 + 4 * h) * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = lb[2 * g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[0][4 * g + e] = v[e];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
                {
                    constexpr int AH = EMAVFI_RING_AHEAD;
                    const char *xb[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) xb[dy] = smem + C::MID_OFF + ((yb - 1 + dy - a0) & 3) * C::MID + (cb * 32 + r) * C::PSTR + h * 16;
                    vec xq[AH + 1];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < AH; ++s) xq[s] = *reinterpret_cast<const vec *>(xb[s / 12] + ((s / 4) % 3) * C::PSTR + (s & 3) * 32);
#pragma unroll
                    for (int s = 0; s < 36; ++s) {
                        if (s + AH < 36) {
                            const int n = s + AH;
                            xq[n % (AH + 1)] = *reinterpret_cast<const vec *>(xb[n / 12] + ((n / 4) % 3) * C::PSTR + (n & 3) * 32);
                        }
                        mma_kg(acc[s & 1], wf[s >> 2][s & 3], xq[s % (AH + 1)]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            RING_STAMP(ts3);
            f32x16 acca;
            if (doA) {
                // ---- stage A: row t of feat_ext_conv1 (conv_first_kernel's arithmetic), its contraction
                {
                    const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIAS_OFF + (frag * 32 + 4 * h) * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = lb[2 * g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acca[4 * g + e] = v[e];
                    }
                }
                vec x[5], wa[5];
#pragma unroll
                for (int kg = 0; kg < 5; ++kg) {
                    x[kg] = *reinterpret_cast<const vec *>(smem + C::FR_OFF + ((t - a0 + ady[kg]) & 3) * C::FROW + (cb * 32 + r + adx[kg]) * 16);
                    wa[kg] = *reinterpret_cast<const vec *>(smem + C::WA_OFF + (kg * 2 + frag) * 1024 + lane * 16);
                }
#pragma unroll
                for (int kg = 0; kg < 5; ++kg) mma_kg(acca, wa[kg], x[kg]);
            }
            if (doB) {   // stage B's epilogue (VALU + two LDS writes) while stage A's chain is in the matrix pipe
                lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (yb & 1) * C::STG + (cb * 32 + r) * C::PSTR + frag * 64;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    unsigned a[2], c[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v0 = acc[0][4 * g + 2 * q] + acc[1][4 * g + 2 * q], v1 = acc[0][4 * g + 2 * q + 1] + acc[1][4 * g + 2 * q + 1];
                        float u0 = acc[0][4 * (g + 1) + 2 * q] + acc[1][4 * (g + 1) + 2 * q], u1 = acc[0][4 * (g + 1) + 2 * q + 1] + acc[1][4 * (g + 1) + 2 * q + 1];
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        const pair_t pa = {(T)v0, (T)v1}, pb = {(T)u0, (T)u1};
                        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
                        a[q] = sw[0]; c[q] = sw[1];
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + 16 * (g + h)) = u4_t{a[0], a[1], c[0], c[1]};
                }
            }
            if (doA) {   // stage A's epilogue: ReLU, rounded to T, ZERO outside the image, into the row ring
                const bool inside = (unsigned)t < (unsigned)H && (unsigned)xg < (unsigned)W;
                const unsigned keep = inside ? ~0u : 0u;
                lchar_t *mid = (lchar_t *)smem + C::MID_OFF + ((t - a0) & 3) * C::MID + (cb * 32 + r) * C::PSTR + frag * 64;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    unsigned a[2], c[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float v0 = fmaxf(acca[4 * g + 2 * q], 0.0f), v1 = fmaxf(acca[4 * g + 2 * q + 1], 0.0f);
                        const float u0 = fmaxf(acca[4 * (g + 1) + 2 * q], 0.0f), u1 = fmaxf(acca[4 * (g + 1) + 2 * q + 1], 0.0f);
                        const pair_t pa = {(T)v0, (T)v1}, pb = {(T)u0, (T)u1};
                        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa) & keep, __builtin_bit_cast(unsigned, pb) & keep, false, false);
                        a[q] = sw[0]; c[q] = sw[1];
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(mid + 16 * (g + h)) = u4_t{a[0], a[1], c[0], c[1]};
                }
            }
            RING_STAMP(ts4);
            frame_put(t + 2, fold);
#pragma unroll
            for (int c = 0; c < 2; ++c) fold[c] = fnew[c];
            RING_STAMP(ts5);
            RING_STAMP_ADD(0, ts0, ts1); RING_STAMP_ADD(1, ts1, ts2); RING_STAMP_ADD(2, ts2, ts3); RING_STAMP_ADD(3, ts3, ts4); RING_STAMP_ADD(4, ts4, ts5);
            RING_STAMP_STEP();
        }
        __syncthreads();
        store_row(ye - 1, true);
    }
    RING_STAMP_WRITE(p, 15, 4);
}

template <typename T> static int launch_conv_ringfirst_t(const FirstParams &fp, const ConvParams &p, hipStream_t s)
{
    using C = RingFirstCfg<T>;
    if (p.ring != 2 || p.stride != 1 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU) || !fp.relu || fp.H != p.Hout || fp.W != p.Wout) return -2;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ringfirst_kernel<T>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TWO - 1) / C::TWO) * p.B, grid = ((emavfi_switches() & SW_RING_ONE_WG) ? 1 : 2) * ncu;   // (SW_RING_ONE_WG: measurement switch, common.h)
    int nseg, seg_rows;
    const int nwg = conv_ring_work(nstrip, p.Hout, grid, &nseg, &seg_rows);
    conv3x3_ringfirst_kernel<T><<<nwg, 256, C::LDS_BYTES, s>>>(fp, p, nseg, seg_rows);
    return (int)hipGetLastError();
}
