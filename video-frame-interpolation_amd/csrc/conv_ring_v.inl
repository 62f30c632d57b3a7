// conv_ring_v.inl - conv_ring.inl's 64 -> 64 (and 67 -> 64) layers with the VERTICAL taps on the accumulators instead of on the
// operand reads (round 5; the idea that rebuilt the two fused heads, applied to the main contraction).  Included by conv3x3.inl.
//
// conv_ring.inl computes output row y from the ring's rows y-1, y, y+1: 36 operand reads (1 KiB each) for 36 MFMAs, every input row
// read three times from LDS.  Here a step CONSUMES one input row rho: each of its 12 operands (dx, k-group) is read once and feeds
// three MFMAs - with the weights of dy = 2, 1, 0 - into three accumulators, the pending sums of output rows rho-1, rho, rho+1.  After the
// step the first is a finished row (epilogue -> staging -> stored one step later), and the roles rotate for free: the FIRST MFMA of
// a chain reads the accumulator of the chain one row younger as its C operand and writes its own registers (vdst != src2), the
// youngest chain starts from 0 (the bias moved into the epilogue).  Same MFMAs, a third of the LDS operand traffic, three
// independent chains instead of two, a ring of D rows instead of D + 2 (30 KiB instead of 50), 16 more accumulator registers.
// A piece of n rows takes n + 2 steps (its first and last two steps also feed rows the piece does not own: 72 MFMAs per wave and piece
// that a per-step selection of the chains would save - built first, and hipcc then kept five accumulator sets and spilled 43
// registers; with RingWork's contiguous pieces (~1.4 per workgroup and ~225 rows at 720p) the two steps are ~1 %).
//
// Everything else - strips, segments, the asm LDS-DMA with its counted wait, staging and buffer stores, the bias table, the TAIL's
// im2col k-group - is conv_ring.inl's and follows its rules (exactly NDMA + NSTORE vector-memory instructions per step and wave).
#pragma once

template <typename T> __device__ __forceinline__ f32x16 mma_kg3(const typename DT<T>::vec &w, const typename DT<T>::vec &x, const f32x16 &c)
{
    if constexpr (std::is_same<T, bf16_t>::value) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
}

template <typename T, bool TAIL> struct ConvRingVCfg {
    using B = ConvRingCfg<T, TAIL, false>;
    static constexpr int PSTR = B::PSTR, SP = B::SP, TW = B::TW, ROWSLOT = B::ROWSLOT, ROWINST = B::ROWINST, ROWB = B::ROWB;
    // D - 1 rows in flight behind the one being read (row rho + D - 1 is issued in step rho, into the slot row rho - 1 has just left)
    static constexpr int D = EMAVFI_RING_DEPTH, RING = D;
    static constexpr int STG_PX = B::STG_PX, STG = B::STG, NSTG = 2;
    static constexpr int STG_OFF = RING * ROWB, BIAS_OFF = STG_OFF + NSTG * STG, BIAS_BYTES = 16 * 64 * 4;
    static constexpr int SCRATCH_OFF = BIAS_OFF + BIAS_BYTES, LDS_BYTES = SCRATCH_OFF + 1024;
    static constexpr int WMAIN = B::WMAIN;
    static constexpr int NDMA = B::NDMA, NSTORE = 3, VMWAIT = NSTORE + (NDMA + NSTORE) * (D - 2);
    static_assert(D >= 2 && sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024, "16-bit types; two workgroups per CU");
};

template <typename T, bool TAIL, bool ALT>
__global__ __launch_bounds__(256, 2) void conv3x3_ringv_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = ConvRingVCfg<T, TAIL>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u2_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(size_t)(lchar_t *)smem;
    const int r = lane & 31, h = lane >> 5;
    const int frag = wave & 1, cb = wave >> 1;
    const int ntx = (p.Wout + C::TW - 1) / C::TW, nstrip = ntx * p.B;
    const char *zeros = (const char *)p.zeros;
    const int npieces = TAIL ? 9 : 8;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const unsigned rowbytes = (unsigned)p.Win * pixbytes;
    const bool relu = p.epi == EPI_RELU;

    // ---- this wave's fragment of the weights: wf[dy * 3 + dx][kg] (conv_ring.inl's packing)
    vec wf[9][4], wt[TAIL ? 3 : 1];
    {
        const char *wb = (const char *)p.w + frag * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wf[t][kg] = *reinterpret_cast<const vec *>(wb + (t * 4 + kg) * 2048);
        if constexpr (TAIL) {
            // the im2col k-group of vertical tap dy: K = 8 h + e <-> horizontal tap 2 h + (e >> 2), channel 64 + (e & 3).  The blob packs
            // the nine taps as K = 16 j + 8 hh + e <-> tap 4 j + 2 hh + (e >> 2) (conv_ring.inl): tap t's four channels of row i are the
            // 8 bytes at fragment t >> 2, lane ((t >> 1) & 1, i), byte 8 (t & 1)
            const char *tb = (const char *)p.w + C::WMAIN + frag * 1024;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                u2_t q[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int dx = 2 * h + k, t = dy * 3 + (dx < 3 ? dx : 2);
                    q[k] = *reinterpret_cast<const u2_t *>(tb + (t >> 2) * 2048 + ((((t >> 1) & 1) * 32 + r) * 16) + 8 * (t & 1));
                    if (dx >= 3) q[k] = u2_t{0u, 0u};
                }
                wt[dy] = __builtin_bit_cast(vec, u4_t{q[0][0], q[0][1], q[1][0], q[1][1]});
            }
        }
    }
    // ---- lane constants of the row DMA: instruction jn covers 16-byte slots [64 jn, 64 jn + 64) of a ring row
    unsigned xoff[C::NDMA];
    unsigned xcol[C::NDMA];
#pragma unroll
    for (int i = 0; i < C::NDMA; ++i) {
        const int q = (i * 4 + wave) * 64 + lane;
        const int px = q / C::SP, pc = q - px * C::SP;
        xoff[i] = (unsigned)px * pixbytes + (unsigned)pc * 16u;
        xcol[i] = (q < C::ROWSLOT && pc < npieces) ? (unsigned)px : 0x40000000u;   // far outside any image: the zero page
    }

    RING_STAMP_DECL;
    RingWork work(nstrip, p.Hout, nseg, seg_rows);
    int strip, ys, ye;
#pragma unroll 1
    while (work.next(strip, ys, ye)) {
        const int b = strip / ntx, tx = strip - b * ntx;
        const int a0 = ys, a1 = ye - 1;   // output rows of this piece: [a0, a1]
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        const int ox0 = tx * C::TW, ix0 = ox0 - 1;
        auto dma_row = [&](int gy, int slot, bool live) {
            const bool rowok = live && (unsigned)gy < (unsigned)p.Hin;   // wave-uniform
            const char *rowp = gin + (size_t)(rowok ? gy : 0) * rowbytes + (ptrdiff_t)ix0 * (ptrdiff_t)pixbytes;
#pragma unroll
            for (int i = 0; i < C::NDMA; ++i) {
                const int jn = i * 4 + wave_u;
                const bool ok = rowok && (unsigned)(ix0 + (int)xcol[i]) < (unsigned)p.Win;
                const char *src = ok ? rowp + xoff[i] : zeros;
                const unsigned dst = lds0 + (jn < C::ROWINST ? (unsigned)(slot * C::ROWB + jn * 1024) : (unsigned)C::SCRATCH_OFF);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
            }
        };
        // the bias table in LDS: [border class 0..15][64] floats (bias_mode 0: one class) - no global loads inside the row loop
        {
            const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(p.bias + (p.bias_mode == 1 ? (size_t)b * 16 * 64 : 0));
            if (tid < (p.bias_mode == 1 ? 256 : 16)) reinterpret_cast<f32x4 *>(smem + C::BIAS_OFF)[tid] = bsrc[tid];
        }
        const int xg = ox0 + cb * 32 + r;   // this lane's output column
        const int xm = (xg >= 1 ? 1 : 0) | (xg <= p.Wout - 2 ? 2 : 0);

        // ---- the step's NSTORE store instructions (buffer stores: lanes outside the image, or !real, are dropped by the range check)
        char *obase = reinterpret_cast<char *>(p.out) + (((size_t)b * p.Hout * p.Wout + (size_t)tx * C::TW) * p.out_ps + p.out_coff) * sizeof(T);
        unsigned soff[C::NSTORE];
        {
            const int npx = min(C::TW, p.Wout - tx * C::TW);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
                soff[i] = (px < npx && ch * 8 < p.cstore) ? (unsigned)px * (unsigned)p.out_ps * (unsigned)sizeof(T) + ch * 16u : 0x80000000u;
            }
            soff[2] = (p.out_fill && tid < npx) ? (unsigned)tid * (unsigned)p.out_ps * (unsigned)sizeof(T) + 128u : 0x80000000u;
        }
        auto store_row = [&](int y, bool real) {
            lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG;
            char *orow = obase + (size_t)(real ? y : a0) * p.Wout * p.out_ps * sizeof(T);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
                const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * C::STG_PX + ch * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, soff[i], 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u4_t{0u, 0u, 0u, 0u}, rs, soff[2], 0, 0);
        };
        // input rows a0 - 1 .. a0 + D - 3 -> slots 0 .. D - 2, each followed by NSTORE dropped stores: the steady state's pattern
#pragma unroll 1
        for (int k = 0; k <= C::D - 2; ++k) {
            dma_row(a0 - 1 + k, k, a0 - 1 + k <= a1 + 1);
            store_row(a0, false);
        }
        // the pending sums: acc[dy] = output row (row being consumed) + 1 - dy
        f32x16 acc[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[d][i] = 0.0f;
        const char *xl = ring + (cb * 32 + r) * C::PSTR + h * 16;
        // one input row into the three chains; the first MFMA of chain dy continues chain dy - 1's registers (the pieces' first and
        // last two steps feed rows outside [a0, a1] too: never stored)
        auto contract = [&](int s0) {
            constexpr int AH = TAIL ? EMAVFI_RING_AHEAD - 1 : EMAVFI_RING_AHEAD;
            const char *xb = xl + s0 * C::ROWB;
            vec xt;
            if constexpr (TAIL) {
                const char *tl = ring + s0 * C::ROWB + (cb * 32 + r + 2 * h) * C::PSTR + 128;
                u2_t lo = *reinterpret_cast<const u2_t *>(tl);
                u2_t hi = *reinterpret_cast<const u2_t *>(tl + C::PSTR);
                if (h) { hi[0] = 0u; hi[1] = 0u; }   // horizontal tap 3 does not exist (its weights are zero, but 0 x Inf is not)
                xt = __builtin_bit_cast(vec, u4_t{lo[0], lo[1], hi[0], hi[1]});
            }
            vec xq[AH + 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < AH; ++s) xq[s] = *reinterpret_cast<const vec *>(xb + (s >> 2) * C::PSTR + (s & 3) * 32);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                if (s + AH < 12) {
                    const int n = s + AH;
                    xq[n % (AH + 1)] = *reinterpret_cast<const vec *>(xb + (n >> 2) * C::PSTR + (n & 3) * 32);
                }
                const vec &x = xq[s % (AH + 1)];
                const int dx = s >> 2, kg = s & 3;
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
                    acc[2] = mma_kg3<T>(wf[6 + dx][kg], x, acc[1]);
                    acc[1] = mma_kg3<T>(wf[3 + dx][kg], x, acc[0]);
                    acc[0] = mma_kg3<T>(wf[dx][kg], x, z);
                } else {
                    acc[2] = mma_kg3<T>(wf[6 + dx][kg], x, acc[2]);
                    acc[1] = mma_kg3<T>(wf[3 + dx][kg], x, acc[1]);
                    acc[0] = mma_kg3<T>(wf[dx][kg], x, acc[0]);
                }
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read to just above its MFMA)
            }
            if constexpr (TAIL) {
                acc[2] = mma_kg3<T>(wt[2], xt, acc[2]);
                acc[1] = mma_kg3<T>(wt[1], xt, acc[1]);
                acc[0] = mma_kg3<T>(wt[0], xt, acc[0]);
            }
        };
        int s0 = 0;   // ring slot of input row rho
#pragma unroll 1
        for (int rho = a0 - 1; rho <= a1 + 1; ++rho) {
            RING_STAMP(ts0);
            // this wave's part of input row rho (all but the youngest VMWAIT instructions) and its staging writes of row rho - 2
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::VMWAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            RING_STAMP(ts1);
            {
                int sl = s0 + C::D - 1; sl = sl >= C::RING ? sl - C::RING : sl;
                dma_row(rho + C::D - 1, sl, rho + C::D - 1 <= a1 + 1);
            }
            store_row(rho - 2, rho - 2 >= a0);
            RING_STAMP(ts2);
            contract(s0);
            RING_STAMP(ts3);
            // ---- output row rho - 1 is finished: bias, optional ReLU; this wave's 32 channels of its 32 pixels into the row's staging buffer
            {
                const int y = rho - 1;
                // motion_estimation.0 (bias_mode 1): the folded context half depends on the pixel's border class
                const int ym = (y >= 1 ? 1 : 0) | (y <= p.Hout - 2 ? 2 : 0), cls = p.bias_mode == 1 ? ym * 4 + xm : 0;
                const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIAS_OFF + (cls * 64 + frag * 32 + 4 * h) * 4);
                lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG + (cb * 32 + r) * C::STG_PX;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const f32x4 bv = lb[2 * g], bu = lb[2 * g + 2];
                    unsigned a[2], c[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v0 = acc[2][4 * g + 2 * q] + bv[2 * q], v1 = acc[2][4 * g + 2 * q + 1] + bv[2 * q + 1];
                        float u0 = acc[2][4 * (g + 1) + 2 * q] + bu[2 * q], u1 = acc[2][4 * (g + 1) + 2 * q + 1] + bu[2 * q + 1];
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        unsigned ua = pack16x2<T>(v0, v1, ALT), ub = pack16x2<T>(u0, u1, ALT);
                        const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
                        a[q] = sw[0]; c[q] = sw[1];
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + 16 * (frag * 4 + g + h)) = u4_t{a[0], a[1], c[0], c[1]};
                }
            }
            s0 = s0 + 1 >= C::RING ? 0 : s0 + 1;
            RING_STAMP(ts4);
            RING_STAMP_ADD(0, ts0, ts1); RING_STAMP_ADD(1, ts1, ts2); RING_STAMP_ADD(2, ts2, ts3); RING_STAMP_ADD(3, ts3, ts4);
            RING_STAMP_STEP();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the next item primes the same slots)
        __syncthreads();
        store_row(a1, true);
    }
    RING_STAMP_WRITE(p, 15 + (TAIL ? 1 : 0), 4);
}

template <typename T, bool TAIL, bool ALT> static int launch_conv_ringv_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvRingVCfg<T, TAIL>;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ringv_kernel<T, TAIL, ALT>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TW - 1) / C::TW) * p.B, grid = ((emavfi_switches() & SW_RING_ONE_WG) ? 1 : 2) * ncu;
    int nseg, seg_rows;
    const int nwg = conv_ring_work(nstrip, p.Hout, grid, &nseg, &seg_rows);
    conv3x3_ringv_kernel<T, TAIL, ALT><<<nwg, 256, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}
