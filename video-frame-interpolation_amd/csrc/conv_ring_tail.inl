// conv_ring_tail.inl - reconstruction.1 + .2 (64 -> 32 + ReLU, 32 -> 3 + tanh, (t + 1) / 2: ema_vfi.py:104-106, 146) in ONE launch for
// the 16-bit modes at mid_channels = 64.  Included by conv3x3.inl behind conv_ring.inl, whose strip walk, LDS-DMA ring and counted
// s_waitcnt it shares.
//
// The two layers ran as conv3x3_persist16_kernel (373 us) + conv_light_kernel (179 us) and moved 2.1 GB for 1.0 GB of input and
// output: the 32-channel tensor between them was written and read back.  Here it exists only as an LDS ring of four rows:
//   * stage A (64 -> 32) on v_mfma_f32_16x16x32: wave w computes ALL 32 output channels (two 16-channel blocks) of the strip's columns
//     [16 w, 16 w + 16) - 18 (tap, k32) steps, 144 weight registers, 18 operand reads and 36 MFMAs per row, an operand read feeding
//     two MFMAs - adds the bias, applies the ReLU, rounds to T and writes its 16 pixels x 32 channels into the row ring (zero outside
//     the image) in the SAME step.  (Rounds 3-4 split the 18 steps between the two waves of a 32-column block - 72 weight registers
//     per wave - and handed the partner's half of the sums over through LDS, finished one step later: 16 KiB of LDS, two more LDS
//     round trips per step and one more step of lag in a kernel whose step is a chain of LDS round trips.  Round 5, after the stamps
//     of profiles/r05_ring_stamps.txt - 40 % matrix-pipe occupancy at 1.98 GHz, i.e. NOT clock-bound like the rest of the family -
//     removed the hand-off: the registers were there.)
//   * stage B (32 -> 3, one row behind stage A): wave w owns the 16 columns [16 w, 16 w + 16) of the strip's 62.  The head's VERTICAL taps
//     sit on the rows of the MFMA's A operand (row 4 c + dy = W[c][dy][dx][:]), so THREE MFMAs - one per horizontal tap, 12 weight
//     registers - add the newest row of the ring to the three output rows it belongs to, and a rotating accumulator hands a finished
//     row to the epilogue every step (see the kernel); tanh, (t + 1) / 2, ONE planar fp32 store for the three planes;
//   * both LDS images are UNPADDED and XOR-swizzled for the 16x16x32 operand pattern: the input ring (128-byte pixels, unit u of
//     pixel c at u ^ swz16(c) - the DMA's lanes fetch the permuted piece) and the row ring (64-byte pixels, unit u at
//     u ^ (((c >> 2) & 1) << 1)); both conflict-free on paper (tools/lds_swizzle_search.py);
//   * per step and wave exactly 3 DMA + 1 store instruction (round 5: the head's three planes leave in ONE store): the counted wait is
//     vmcnt(1 + 4 (D - 2)) = 5 at D = 3.
#ifndef EMAVFI_RT_AHEAD
#define EMAVFI_RT_AHEAD 4   // stage A's operand reads run this many ahead of their MFMAs
#endif
#ifndef EMAVFI_RT_ABL
#define EMAVFI_RT_ABL 0   // timing-only ablations (diagnostic builds): 1 every DMA reads the zero page, 2 no head, 4 no stage-A MFMAs
#endif
template <typename T> struct RingTailCfg {
    static constexpr int TW = 64, TWO = TW - 2, IW = TW + 2, IN_PX = 128, ROWSLOT = IW * 8, ROWINST = (ROWSLOT + 63) / 64, ROWB = ROWINST * 1024;
    static constexpr int D = 3, RING = D + 2, MID_PX = 64, MID = TW * MID_PX, NMID = 4;
    static constexpr int MID_OFF = RING * ROWB, SCRATCH_OFF = MID_OFF + NMID * MID, LDS_BYTES = SCRATCH_OFF + 1024;
    static constexpr int NDMA = (ROWINST + 3) / 4, NSTORE = 1, VMWAIT = NSTORE + (NDMA + NSTORE) * (D - 2);
    static_assert(sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024 && NDMA == 3, "16-bit types; two workgroups per CU");
};

template <typename T, bool R16, bool TANH>   // ConvParams::round16 (EMAVFI_AMP16) and epi2 == EPI_PLANAR_TANH01 as template arguments: the run-time
                                             // tests put ocml's branchy tanhf into every instance's row loop and made hipcc duplicate a store
                                             // into both arms of a branch - the code-object tests count the loop's VMEM instructions
__global__ __launch_bounds__(256, 2) void conv3x3_ringtail_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = RingTailCfg<T>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u2_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(size_t)(lchar_t *)smem;
    const int j = lane & 15, kb = lane >> 4;
    const int ntx = (p.Wout + C::TWO - 1) / C::TWO, nstrip = ntx * p.B;
    const char *zeros = (const char *)p.zeros;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const unsigned rowbytes = (unsigned)p.Win * pixbytes;
    const auto s4 = [](int c) { return ((c >> 2) & 1) << 1; };   // the row ring's unit permutation

    // ---- stage A: this wave's 18 (tap, k32) steps for both 16-channel output blocks
    // (weights: the 16x16x32 packing [tap][k32][cout16 block 0..1][lane (i, kb)][8])
    vec wr[18][2];
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const char *wb = (const char *)p.w + (s * 2) * 1024 + lane * 16;
        wr[s][0] = *reinterpret_cast<const vec *>(wb);
        wr[s][1] = *reinterpret_cast<const vec *>(wb + 1024);
    }
    // operand of step (tap, k32): pixel c = 16 w + j + dx, unit (4 k32 + kb) ^ swz16(c); six distinct lane offsets (dx, k32)
    int xoA[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int dx = i >> 1, k32 = i & 1;
        xoA[i] = (wave * 16 + j + dx) * C::IN_PX + (((k32 * 4 + kb) ^ swz16(j + dx)) << 4);
    }
    float ba[2][4];   // bias of output channels 16 blk + 4 kb .. + 3 (the accumulator rows of this lane)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) ba[blk][e] = p.bias[blk * 16 + kb * 4 + e];
    // ---- stage B: the 32 -> 3 head with its VERTICAL taps on the matrix core's rows.  Row 4 c + dy of the 16-row A operand of the MFMA for
    // horizontal tap dx holds W[c][dy][dx][0..31] (c < 3, dy < 3; the other rows read a zero row of the blob), the B operand is the
    // row ring's newest row rho, 16 pixels shifted by dx: ONE MFMA per dx adds row rho's contribution to the three output rows it
    // belongs to - register dy of the accumulator of lane (j, kb = c) is output row rho + 1 - dy, channel c, pixel j.  After the three
    // MFMAs register 2 is a finished row; the accumulator then rotates (2 <- 1 <- 0 <- 0).  Three operand reads and three MFMAs per head
    // row instead of nine and nine, 12 weight registers instead of 36, and still one epilogue and ONE store for the three planes.
    vec hw[3];
    {
        const int i = lane & 15, c = i >> 2, dy = i & 3;
        const bool realrow = c < p.nplanes && dy < 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            hw[dx] = *reinterpret_cast<const vec *>((const char *)p.head_w + (realrow ? dy * 3 + dx : 0) * 2048 + (kb * 16 + (realrow ? c : 15)) * 16);
    }
    const float hb = kb < p.nplanes ? p.head_bias[kb < p.nplanes ? kb : 0] : 0.0f;
    int hxo[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) hxo[dx] = (wave * 16 + j + dx) * C::MID_PX + ((kb ^ s4(wave * 16 + j + dx)) << 4);
    // ---- lane constants of the row DMA: 16-byte slot q of a ring row holds piece (q & 7) ^ swz16(q >> 3) of pixel q >> 3
    unsigned xoff[C::NDMA], xcol[C::NDMA];
#pragma unroll
    for (int i = 0; i < C::NDMA; ++i) {
        const int q = (i * 4 + wave) * 64 + lane, px = q >> 3, pc = (q & 7) ^ swz16(px);
        xoff[i] = (unsigned)px * pixbytes + (unsigned)pc * 16u;
        xcol[i] = q < C::ROWSLOT ? (unsigned)px : 0x40000000u;
    }

    RING_STAMP_DECL;
    RingWork work(nstrip, p.Hout, nseg, seg_rows);
    int strip, ys, ye;
#pragma unroll 1
    while (work.next(strip, ys, ye)) {
        const int b = strip / ntx, tx = strip - b * ntx;
        const int a0 = ys - 1, a1 = ye;             // stage-A rows of this item
        const int ox0 = tx * C::TWO - 1, ix0 = ox0 - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        auto dma_row = [&](int gy, int slot, bool live) {   // exactly NDMA instructions per wave (conv_ring.inl)
            const bool rowok = live && (unsigned)gy < (unsigned)p.Hin && !(EMAVFI_RT_ABL & 1);
            const char *rowp = gin + (size_t)(rowok ? gy : 0) * rowbytes + (ptrdiff_t)ix0 * (ptrdiff_t)pixbytes;
#pragma unroll
            for (int i = 0; i < C::NDMA; ++i) {
                const int jn = i * 4 + wave_u;
                const bool ok = rowok && (unsigned)(ix0 + (int)xcol[i]) < (unsigned)p.Win;
                const char *src = ok ? rowp + xoff[i] : zeros;
                const unsigned dst = lds0 + (jn < C::ROWINST ? (unsigned)(slot * C::ROWB + jn * 1024) : (unsigned)C::SCRATCH_OFF);
                // (no "memory" clobber on purpose: LDS reads of this step may be scheduled around the DMA issue.  Safe because the
                // slot written - ring row (y + D) % RING - is read by no wave before the NEXT step's counted wait + s_barrier, which is
                // followed by a compiler-level memory fence; no LDS WRITE of the kernel targets the input ring at all.  m0 IS
                // clobbered: ADVICE r3)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "m0");
            }
        };
        // the head's store: wave w owns head columns [16 w, 16 w + 16); lane (j, kb) holds channel kb of column 16 w + j: plane kb
        const int hc = wave * 16 + j, hx = tx * C::TWO + hc;
        const unsigned soff = (hc < C::TWO && hx < p.Wout && kb < p.nplanes) ? (unsigned)hx * 4u + (unsigned)kb * (unsigned)(p.Hout * p.Wout) * 4u : 0x80000000u;
        const size_t plane = (size_t)p.Hout * p.Wout;
        // stage A: row y of the 64 -> 32 layer for this wave's 16 columns, all 32 channels: 18 operand reads, 36 MFMAs, then bias, ReLU,
        // rounding and the row-ring write (zero outside the image: the rows are the head convolution's padding)
        // (round 5, end: FOUR accumulation chains - even and odd steps of either channel block - instead of two: a lone wave's two chains of
        // 16-cycle MFMAs left the pipe idle more than half of the contraction; and the step's order is contraction A, the head's three
        // MFMAs queued behind it, A's epilogue, the head's epilogue)
        f32x4 acc[2][2];
        auto stage_a_mma = [&](int s0) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) { acc[blk][0] = f32x4{ba[blk][0], ba[blk][1], ba[blk][2], ba[blk][3]}; acc[blk][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
            // LDS byte offsets, not pointers: a row-base array indexed by a run-time dy loses its address space and the reads become
            // flat_load (vmcnt AND lgkmcnt: the counted wait would be wrong - tests/test_cabi_cpu.py checks the code object)
            int xs[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                int sl = s0 + dy; sl = sl >= C::RING ? sl - C::RING : sl;
                xs[dy] = sl * C::ROWB;
            }
            constexpr int AH = EMAVFI_RT_AHEAD;
            vec xq[AH + 1];
            auto xread = [&](int q) {   // step q: tap q >> 1 (dy = tap / 3, dx = tap % 3), k32 = q & 1
                const int tap = q >> 1, dy = tap / 3, dx = tap - 3 * dy;
                return *reinterpret_cast<const __attribute__((address_space(3))) vec *>((lchar_t *)smem + xs[dy] + xoA[dx * 2 + (q & 1)]);
            };
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < AH; ++q) xq[q] = xread(q);
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                if (q + AH < 18) xq[(q + AH) % (AH + 1)] = xread(q + AH);
                if (!(EMAVFI_RT_ABL & 4) || q == 0) {
                    mma_k32(acc[0][q & 1], wr[q][0], xq[q % (AH + 1)]);
                    mma_k32(acc[1][q & 1], wr[q][1], xq[q % (AH + 1)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto stage_a_out = [&](int y) {
            const int c = wave * 16 + j;
            const bool inside = (unsigned)y < (unsigned)p.Hout && (unsigned)(ox0 + c) < (unsigned)p.Wout;
            const unsigned keepm = inside ? ~0u : 0u;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float sum = acc[blk][0][e] + acc[blk][1][e]; v[e] = p.epi == EPI_RELU ? fmaxf(sum, 0.0f) : sum; }
                typedef __attribute__((ext_vector_type(2))) T pair_t;
                const pair_t lo = {(T)v[0], (T)v[1]}, hi = {(T)v[2], (T)v[3]};
                // channels 16 blk + 4 kb .. + 3 = bytes 32 blk + 8 kb of the 64-byte pixel: unit 2 blk + (kb >> 1), half kb & 1
                lchar_t *mp = (lchar_t *)smem + C::MID_OFF + ((y - a0) & 3) * C::MID + c * C::MID_PX + (((2 * blk + (kb >> 1)) ^ s4(c)) << 4) + (kb & 1) * 8;
                *reinterpret_cast<__attribute__((address_space(3))) u2_t *>(mp) = u2_t{__builtin_bit_cast(unsigned, lo) & keepm, __builtin_bit_cast(unsigned, hi) & keepm};
            }
        };
        // stage B: the row ring's row rho (written in the step before) enters the head; head row rho - 1 = yb is finished by it.
        // NSTORE = 1 store (lane (j, kb): plane kb)
        f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};   // per item: register dy = the pending sums of output row (newest row consumed) + 1 - dy
        auto head_mma = [&](int yb) {
            const int mr = C::MID_OFF + ((yb + 1 - a0) & 3) * C::MID;
            vec hxv[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) hxv[dx] = *reinterpret_cast<const __attribute__((address_space(3))) vec *>((lchar_t *)smem + mr + hxo[dx]);
            hacc = f32x4{0.0f, hacc[0], hacc[1], 0.0f};
#pragma unroll
            for (int dx = 0; dx < ((EMAVFI_RT_ABL & 2) ? 1 : 3); ++dx) mma_k32(hacc, hw[dx], hxv[dx]);
        };
        auto head_out = [&](int yb, bool real) {
            float o = hacc[2] + hb;
            float *orow = p.out_planar + (size_t)b * p.nplanes * plane + (size_t)(real ? yb : ys) * p.Wout;
            if constexpr (TANH) {   // ema_vfi.py:106,146 (round16: every op rounds as an fp16 tensor op does under autocast)
                if constexpr (R16) {
                    // tanh in fp32, then the fp16 roundings of the tensor ops (what a CUDA fp16 tanh does): 1 - 2 / (1 + exp(2x)) on v_exp_f32 +
                    // v_rcp_f32 (a few fp32 ulp: invisible behind the rounding to fp16 except on exact ties; the cancellation near 0
                    // disappears in the `+ 1`), branch-free - ocml's tanhf brought branches into the row loop that scattered its blocks
                    o = (float)(half_t)o;
                    const float th = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * o));
                    o = (float)(half_t)((float)(half_t)th + 1.0f) / 2.0f;
                }
                // (tanh(x) + 1) / 2 = 1 / (1 + exp(-2x)): v_exp_f32 + v_rcp_f32 (2 ulp) instead of ocml's branchy tanhf, which cost this
                // kernel as much as its MFMAs
                else o = __frcp_rn(1.0f + __expf(-2.0f * o));
            } else if constexpr (R16) o = (float)(half_t)o;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, soff, 0, 0);
        };
        // input rows a0 - 1 .. a0 + D - 1 -> slots 0 .. D, each followed by NSTORE dropped stores: the steady state's instruction pattern
#pragma unroll 1
        for (int k = 0; k <= C::D; ++k) {
            dma_row(a0 - 1 + k, k, a0 - 1 + k <= a1 + 1);
            // (the store of the steady-state pattern, dropped: a buffer of zero records.  NOT head_mma: its accumulator carries state)
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(p.out_planar, 0, 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(0u, rs0, soff, 0, 0);
        }
        int s0 = 0;   // ring slot of input row y - 1
#pragma unroll 1
        for (int y = a0; y <= a1; ++y) {
            RING_STAMP(ts0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::VMWAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            RING_STAMP(ts1);
            {
                int sl = s0 + C::D + 1; sl = sl >= C::RING ? sl - C::RING : sl;
                dma_row(y + C::D, sl, y + C::D <= a1 + 1);
            }
            RING_STAMP(ts2);
            // stage A's row y; behind its contraction the head consumes row-ring row y - 1 (written before this step's barrier) and finishes
            // head row y - 2
            stage_a_mma(s0);
            RING_STAMP(ts3);
            head_mma(y - 2);
            stage_a_out(y);
            RING_STAMP(ts4);
            head_out(y - 2, y - 2 >= ys);
            s0 = s0 + 1 >= C::RING ? 0 : s0 + 1;
            RING_STAMP(ts5);
            RING_STAMP_ADD(0, ts0, ts1); RING_STAMP_ADD(1, ts1, ts2); RING_STAMP_ADD(2, ts2, ts3); RING_STAMP_ADD(3, ts3, ts4); RING_STAMP_ADD(4, ts4, ts5);
            RING_STAMP_STEP();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        head_mma(a1 - 1);         // (a1 = ye: the segment's last head row, from the rows ye - 2 .. ye)
        head_out(a1 - 1, true);
        __syncthreads();          // the next item's first rows overwrite the rings
    }
    RING_STAMP_WRITE(p, 14, 4);
}

template <typename T, bool R16, bool TANH> static int launch_conv_ringtail_t(const ConvParams &p, hipStream_t s)
{
    using C = RingTailCfg<T>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ringtail_kernel<T, R16, TANH>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TWO - 1) / C::TWO) * p.B, grid = ((emavfi_switches() & SW_RING_ONE_WG) ? 1 : 2) * ncu;   // (SW_RING_ONE_WG: measurement switch, common.h)
    int nseg, seg_rows;
    const int nwg = conv_ring_work(nstrip, p.Hout, grid, &nseg, &seg_rows);
    conv3x3_ringtail_kernel<T, R16, TANH><<<nwg, 256, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}

template <typename T> static int launch_conv_ringtail(const ConvParams &p, hipStream_t s)
{
    if (!p.mfma16 || p.ck != 64 || p.nf != 1 || p.stride != 1 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU) || !p.head_w || !p.head_bias ||
        !p.out_planar || p.nplanes < 1 || p.nplanes > 3)
        return -2;
    if (p.epi2 != EPI_PLANAR && p.epi2 != EPI_PLANAR_TANH01) return -2;
    const bool tanh01 = p.epi2 == EPI_PLANAR_TANH01;
    if (p.round16) return tanh01 ? launch_conv_ringtail_t<T, true, true>(p, s) : launch_conv_ringtail_t<T, true, false>(p, s);
    return tanh01 ? launch_conv_ringtail_t<T, false, true>(p, s) : launch_conv_ringtail_t<T, false, false>(p, s);
}
