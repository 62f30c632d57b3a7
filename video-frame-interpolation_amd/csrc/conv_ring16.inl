// conv_ring16.inl - EXPERIMENT (round 5, DESIGN.md section 7 "what comes next" item 1): conv_ring.inl's 64 -> 64 kernel on
// v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16.  Same structure - a workgroup of four waves walks down a strip of 64 output columns,
// the wave's 32 output channels x 576 K of weights stationary in 144 VGPRs, input rows in an LDS ring filled by inline-assembly LDS-DMA
// behind a counted s_waitcnt, outputs through a double-buffered LDS row - and the same MACs, LDS operand bytes and HBM bytes; what
// changes is the MFMA shape: per row and wave 72 MFMAs of 16 cycles on 4 accumulator quads instead of 36 MFMAs of 32 cycles on two
// 16-register accumulators.  The shape probe of round 2 (profiles/r02_mfma_shape_power.txt) ran this shape at 7 % less package power at
// equal LDS-fed throughput; the ring kernels are clock-bound at the board's power cap (section 3.2e), so the question this kernel
// answers is what that is worth in a real kernel.  Reached ONLY through the stage entry emavfi_conv3x3 with EMAVFI_CONV_RING16=1
// (64 -> 64, stride 1, 16-bit types, bias + optional ReLU); the forward never launches it.
//   * ring rows: UNPADDED 128-byte pixels, 16-byte unit u of pixel c at u ^ swz16(c) (conv_ring_tail.inl's input ring: the 16x16x32 operand
//     pattern - lane (j, kb) reads unit 4 k32 + kb of pixel c0 + j - is conflict-free that way); the DMA's lanes fetch the permuted piece;
//   * weights: the 16x16x32 packing [tap][k32][cout16 block 0..3][lane (i, kb)][8] (PackDesc::mfma16); wave (frag, cb) keeps blocks
//     2 frag, 2 frag + 1 of all 18 (tap, k32) steps;
//   * accumulator D[16 couts][16 pixels]: lane (j, ib) holds couts 4 ib .. 4 ib + 3 of pixel j: four 8-byte writes into the staging row.
#pragma once

template <typename T> struct ConvRing16Cfg {
    static constexpr int TW = 64, IW = TW + 2, IN_PX = 128, ROWSLOT = IW * 8, ROWINST = (ROWSLOT + 63) / 64, ROWB = ROWINST * 1024;
    static constexpr int D = 3, RING = D + 2;
    static constexpr int STG_PX = 144, STG = TW * STG_PX, STG_OFF = RING * ROWB, SCRATCH_OFF = STG_OFF + 2 * STG, LDS_BYTES = SCRATCH_OFF + 1024;
    static constexpr int NDMA = (ROWINST + 3) / 4, NSTORE = 3, VMWAIT = NSTORE + (NDMA + NSTORE) * (D - 2);
    static_assert(sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024 && NDMA == 3, "16-bit types; two workgroups per CU");
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_ring16_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = ConvRing16Cfg<T>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u2_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(size_t)(lchar_t *)smem;
    const int j = lane & 15, kb = lane >> 4;
    const int frag = wave_u & 1, cb = wave_u >> 1;
    const int ntx = (p.Wout + C::TW - 1) / C::TW, nstrip = ntx * p.B, nitems = nstrip * nseg;
    const char *zeros = (const char *)p.zeros;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const unsigned rowbytes = (unsigned)p.Win * pixbytes;
    const bool relu = p.epi == EPI_RELU;

    // ---- this wave's weights: 18 (tap, k32) steps x its two 16-channel output blocks
    vec wr[18][2];
    int xo[18], dyS[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int tap = s >> 1, k32 = s & 1, dy = tap / 3, dx = tap - 3 * dy;
        const char *wb = (const char *)p.w + (s * 4 + frag * 2) * 1024 + lane * 16;
        wr[s][0] = *reinterpret_cast<const vec *>(wb);
        wr[s][1] = *reinterpret_cast<const vec *>(wb + 1024);
        dyS[s] = dy;
        // operand of pixel block 0: pixel c = 32 cb + j + dx, unit (4 k32 + kb) ^ swz16(c) (block 1: + 16 pixels = 2048 bytes, same permutation)
        xo[s] = (cb * 32 + j + dx) * C::IN_PX + (((k32 * 4 + kb) ^ swz16(j + dx)) << 4);
    }
    float bia[2][4];   // bias of output channels 32 frag + 16 blk + 4 kb .. + 3 (the accumulator rows of this lane)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) bia[blk][e] = p.bias[frag * 32 + blk * 16 + kb * 4 + e];
    // ---- lane constants of the row DMA: 16-byte slot q of a ring row holds piece (q & 7) ^ swz16(q >> 3) of pixel q >> 3
    unsigned xoff[C::NDMA], xcol[C::NDMA];
#pragma unroll
    for (int i = 0; i < C::NDMA; ++i) {
        const int q = (i * 4 + wave) * 64 + lane, px = q >> 3, pc = (q & 7) ^ swz16(px);
        xoff[i] = (unsigned)px * pixbytes + (unsigned)pc * 16u;
        xcol[i] = q < C::ROWSLOT ? (unsigned)px : 0x40000000u;
    }

#pragma unroll 1
    for (int item = (int)blockIdx.x; item < nitems; item += (int)gridDim.x) {
        const int strip = item % nstrip, seg = item / nstrip;
        const int b = strip / ntx, tx = strip - b * ntx;
        const int ys = seg * seg_rows, ye = min(ys + seg_rows, p.Hout);
        const int a0 = ys, a1 = ye - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        const int ix0 = tx * C::TW - 1;
        auto dma_row = [&](int gy, int slot, bool live) {   // exactly NDMA instructions per wave (conv_ring.inl)
            const bool rowok = live && (unsigned)gy < (unsigned)p.Hin;
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
        char *obase = reinterpret_cast<char *>(p.out) + (((size_t)b * p.Hout * p.Wout + (size_t)tx * C::TW) * p.out_ps + p.out_coff) * sizeof(T);
        const int npx = min(C::TW, p.Wout - tx * C::TW);
        unsigned soff[C::NSTORE];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
            soff[i] = (px < npx && ch * 8 < p.cstore) ? (unsigned)px * (unsigned)p.out_ps * (unsigned)sizeof(T) + ch * 16u : 0x80000000u;
        }
        soff[2] = 0x80000000u;   // (the third store of conv_ring.inl's pattern - out_fill - is always dropped here: the count must match)
        auto store_row = [&](int y, bool real) {
            lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG;
            char *orow = obase + (size_t)(real ? y : ys) * p.Wout * p.out_ps * sizeof(T);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
                const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * C::STG_PX + ch * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, soff[i], 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u4_t{0u, 0u, 0u, 0u}, rs, soff[2], 0, 0);
        };
        // input rows a0 - 1 .. a0 + D - 1 -> slots 0 .. D, each followed by NSTORE dropped stores: the steady state's instruction pattern
#pragma unroll 1
        for (int k = 0; k <= C::D; ++k) {
            dma_row(a0 - 1 + k, k, a0 - 1 + k <= a1 + 1);
            store_row(ys, false);
        }
        int s0 = 0;   // ring slot of input row y - 1
#pragma unroll 1
        for (int y = a0; y <= a1; ++y) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::VMWAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                int sl = s0 + C::D + 1; sl = sl >= C::RING ? sl - C::RING : sl;
                dma_row(y + C::D, sl, y + C::D <= a1 + 1);
            }
            store_row(y - 1, y > a0);
            f32x4 acc[2][2];   // [pixel block][cout block]
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) acc[pb][blk] = f32x4{bia[blk][0], bia[blk][1], bia[blk][2], bia[blk][3]};
            {
                // LDS byte offsets, not pointers (an array of row bases indexed by a run-time dy loses its address space: flat loads)
                int xs[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    int sl = s0 + dy; sl = sl >= C::RING ? sl - C::RING : sl;
                    xs[dy] = sl * C::ROWB;
                }
                int xrow[18];
#pragma unroll
                for (int s = 0; s < 18; ++s) xrow[s] = (dyS[s] == 0 ? xs[0] : dyS[s] == 1 ? xs[1] : xs[2]) + xo[s];
                constexpr int AH = 4;
                vec xq[AH + 1];
                auto xread = [&](int q) { return *reinterpret_cast<const __attribute__((address_space(3))) vec *>((lchar_t *)smem + xrow[q >> 1] + (q & 1) * 2048); };
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < AH; ++q) xq[q] = xread(q);
#pragma unroll
                for (int q = 0; q < 36; ++q) {
                    if (q + AH < 36) xq[(q + AH) % (AH + 1)] = xread(q + AH);
                    mma_k32(acc[q & 1][0], wr[q >> 1][0], xq[q % (AH + 1)]);
                    mma_k32(acc[q & 1][1], wr[q >> 1][1], xq[q % (AH + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- optional ReLU; lane (j, kb) holds channels 32 frag + 16 blk + 4 kb .. + 3 of pixel 32 cb + 16 pb + j: four 8-byte writes
            {
                lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG;
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu ? fmaxf(acc[pb][blk][e], 0.0f) : acc[pb][blk][e];
                        const u2_t w2 = {pack16x2<T>(v[0], v[1], false), pack16x2<T>(v[2], v[3], false)};
                        *reinterpret_cast<__attribute__((address_space(3))) u2_t *>(stg + (cb * 32 + pb * 16 + j) * C::STG_PX + (frag * 32 + blk * 16 + kb * 4) * 2) = w2;
                    }
            }
            s0 = s0 + 1 >= C::RING ? 0 : s0 + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the next item primes the same slots)
        __syncthreads();
        store_row(a1, true);
    }
}

template <typename T> static int launch_conv_ring16(const ConvParams &p, hipStream_t s)
{
    using C = ConvRing16Cfg<T>;
    if (p.stride != 1 || p.nchunk != 1 || p.npass != 1 || p.nf != 2 || p.ck != 64 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU) || p.out_alt ||
        p.head_w || p.w2)
        return -2;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ring16_kernel<T>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TW - 1) / C::TW) * p.B, grid = 2 * ncu;
    int nseg, seg_rows;
    conv_ring_segments(nstrip, p.Hout, grid, &nseg, &seg_rows);
    const int nitems = nstrip * nseg;
    conv3x3_ring16_kernel<T><<<nitems < grid ? nitems : grid, 256, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}
