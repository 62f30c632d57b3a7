// conv_ring2.inl - TWO consecutive 64 -> 64 conv_blocks in ONE launch (round 4; included by conv3x3.inl):
//   feat_ext_blocks.conv_block_1 + conv_block_2 (ema_vfi.py:75-76): A = conv_block_1, B = conv_block_2 (writes `feat` into the
//   fusion tensor), both + ReLU.
// conv_ring.inl's structure (weights of a wave's output fragment stationary in 144 VGPRs, input rows through an LDS ring filled by
// LDS-DMA behind a counted s_waitcnt, one s_barrier per output row) with a workgroup of EIGHT waves, one per CU:
//   * waves 0-3 = layer A.  Exactly conv3x3_ring_kernel's main loop; their output row (rounded to the storage type and ZEROED
//     outside the image: it is layer B's zero padding) is not stored to HBM but written into a second LDS ring of four rows;
//   * waves 4-7 = layer B, two rows behind: the same main loop on the mid ring (144-byte pixels: the same conflict-free operand
//     pattern), epilogue to a double-buffered staging row, whole-line buffer stores one step later.
// Per step every wave executes one s_barrier; A-waves issue their 3 DMA instructions and wait "all but the youngest 3" (D = 3), the
// B-waves' stores are never waited for (their data leaves through registers).  A strip of 64 A-columns yields 62 B-columns (B needs
// A's columns -1 .. 62) and a segment computes two extra A-rows: 64 / 62 x (rows + 2) / rows of layer A's work - in exchange the
// 0.94 GB tensor between the two layers (B = 8 x 720p) is neither written nor read, and these layers were HBM-bound (52 % of the
// HBM roofline at 48 % of the MFMA peak: DESIGN.md section 3.2b).  Both stages repeat the unfused kernels' arithmetic operation
// for operation (same accumulation chains, same rounding of A's rows), so the result is BIT-IDENTICAL to the two launches
// (tests/test_gpu_parity.py::test_fused_block_pair_equals_the_two_launch_path; EMAVFI_CONV_RING2=0 runs the two launches).
template <typename T> struct ConvRing2Cfg {
    static constexpr int PSTR = 144, SP = 9, TW = 64, IW = TW + 2, ROWSLOT = IW * SP, ROWINST = (ROWSLOT + 63) / 64, ROWB = ROWINST * 1024;
    static constexpr int TWO = TW - 2;                       // output columns a strip contributes
    static constexpr int D = 3, RING = D + 2;                // layer A's input ring: rows y - 1 .. y + 1 in use, y + 2 .. y + D in flight
    static constexpr int NMID = 4, MIDROW = IW * PSTR;       // the ring of A's rows: 66 pixels (A's columns 0 .. 63 + two never written, read
                                                             // only by B's dropped columns 62, 63)
    static constexpr int MID_OFF = RING * ROWB, STG_PX = 144, STG = TW * STG_PX, STG_OFF = MID_OFF + NMID * MIDROW;
    static constexpr int BIASA_OFF = STG_OFF + 2 * STG, BIASA_BYTES = 16 * 64 * 4, BIASB_OFF = BIASA_OFF + BIASA_BYTES, BIASB_BYTES = 256;
    static constexpr int SCRATCH_OFF = BIASB_OFF + BIASB_BYTES, LDS_BYTES = SCRATCH_OFF + 1024;
    static constexpr int NDMA = (ROWINST + 3) / 4, VMWAIT_A = NDMA * (D - 2), NSTORE = 3;
    static_assert(sizeof(T) == 2 && LDS_BYTES <= 160 * 1024 && MIDROW % 16 == 0, "16-bit types; one workgroup per CU");
};

template <typename T, bool ALT>   // ALT: layer B stores the other 16-bit type (ConvParams::out_alt: `feat` as f16 in the bf16 model)
__global__ __launch_bounds__(512, 2) void conv3x3_ring2_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = ConvRing2Cfg<T>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const bool roleB = __builtin_amdgcn_readfirstlane(wave8) >= 4;
    const int wave = wave8 & 3, tid4 = tid & 255;            // indices inside the role's four waves
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(size_t)(lchar_t *)smem;
    const int r = lane & 31, h = lane >> 5;
    const int frag = wave & 1, cb = wave >> 1;
    const int ntx = (p.Wout + C::TWO - 1) / C::TWO, nstrip = ntx * p.B;
    const char *zeros = (const char *)p.zeros;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const unsigned rowbytes = (unsigned)p.Win * pixbytes;
    const bool relu = p.epi == EPI_RELU;

    // ---- this wave's fragment of ITS layer's weights
    vec wf[9][4];
    {
        const char *wb = (const char *)(roleB ? p.w2 : p.w) + frag * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wf[t][kg] = *reinterpret_cast<const vec *>(wb + (t * 4 + kg) * 2048);
    }
    // ---- lane constants of layer A's row DMA (conv_ring.inl): instruction jn covers 16-byte slots [64 jn, 64 jn + 64) of a ring row
    unsigned xoff[C::NDMA], xcol[C::NDMA];
#pragma unroll
    for (int i = 0; i < C::NDMA; ++i) {
        const int q = (i * 4 + wave) * 64 + lane;
        const int px = q / C::SP, pc = q - px * C::SP;
        xoff[i] = (unsigned)px * pixbytes + (unsigned)pc * 16u;
        xcol[i] = (q < C::ROWSLOT && pc < 8) ? (unsigned)px : 0x40000000u;   // the ninth piece / slots past the row: the zero page
    }

    RingWork work(nstrip, p.Hout, nseg, seg_rows);
    int strip, ys, ye;
#pragma unroll 1
    while (work.next(strip, ys, ye)) {
        const int b = strip / ntx, tx = strip - b * ntx;
        const int a0 = ys - 1, a1 = ye;                      // layer A's rows [a0, a1]: one more on either side for B's taps
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        const int ox0 = tx * C::TWO - 1;                     // image column of A's column 0
        const int ix0 = ox0 - 1;
        auto dma_row = [&](int gy, int slot, bool live) {    // exactly NDMA instructions per A-wave
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
        // bias tables in LDS (no global load inside the row loop): A [border class][64] (bias_mode 1) or [64]; B [64]
        if (!roleB) {
            const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(p.bias + (p.bias_mode == 1 ? (size_t)b * 16 * 64 : 0));
            if (tid4 < (p.bias_mode == 1 ? 256 : 16)) reinterpret_cast<f32x4 *>(smem + C::BIASA_OFF)[tid4] = bsrc[tid4];
        } else if (tid4 < 16) {
            reinterpret_cast<f32x4 *>(smem + C::BIASB_OFF)[tid4] = reinterpret_cast<const f32x4 *>(p.bias2)[tid4];
        }
        const int xga = ox0 + cb * 32 + r;                   // image column of this lane's A-column
        const int xm = (xga >= 1 ? 1 : 0) | (xga <= p.Wout - 2 ? 2 : 0);

        // ---- layer B's store instructions (buffer stores: lanes outside the strip's 62 columns / the image are dropped by the range check)
        char *obase = reinterpret_cast<char *>(p.out) + (((size_t)b * p.Hout * p.Wout + (size_t)tx * C::TWO) * p.out_ps + p.out_coff) * sizeof(T);
        unsigned soff[C::NSTORE];
        {
            const int npx = min(C::TWO, p.Wout - tx * C::TWO);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid4, px = q >> 3, ch = q & 7;
                soff[i] = (px < npx && ch * 8 < p.cstore) ? (unsigned)px * (unsigned)p.out_ps * (unsigned)sizeof(T) + ch * 16u : 0x80000000u;
            }
            soff[2] = (p.out_fill && tid4 < npx) ? (unsigned)tid4 * (unsigned)p.out_ps * (unsigned)sizeof(T) + 128u : 0x80000000u;
        }
        auto store_row = [&](int y, bool real) {
            lchar_t *stg = (lchar_t *)smem + C::STG_OFF + (y & 1) * C::STG;
            char *orow = obase + (size_t)(real ? y : ys) * p.Wout * p.out_ps * sizeof(T);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid4, px = q >> 3, ch = q & 7;
                const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * C::STG_PX + ch * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, soff[i], 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u4_t{0u, 0u, 0u, 0u}, rs, soff[2], 0, 0);
        };
        // the 36 (read, MFMA) pairs of one output row from three ring rows xb[dy] (conv_ring.inl's main loop: operands four k-groups ahead)
        auto contract = [&](const char *const (&xb)[3], f32x16 (&acc)[2]) {
            constexpr int AH = 4;
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
        };
        // ReLU + rounding + the 32x32 fragment's four 16-byte units of this lane's pixel into an LDS row with 144-byte pixels
        auto write_row = [&](lchar_t *dst_px, const f32x16 (&acc)[2], bool keep_it, bool alt) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                unsigned a[2], c[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v0 = acc[0][4 * g + 2 * q] + acc[1][4 * g + 2 * q], v1 = acc[0][4 * g + 2 * q + 1] + acc[1][4 * g + 2 * q + 1];
                    float u0 = acc[0][4 * (g + 1) + 2 * q] + acc[1][4 * (g + 1) + 2 * q], u1 = acc[0][4 * (g + 1) + 2 * q + 1] + acc[1][4 * (g + 1) + 2 * q + 1];
                    if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                    unsigned ua = pack16x2<T>(v0, v1, alt), ub = pack16x2<T>(u0, u1, alt);
                    const unsigned keep = keep_it ? ~0u : 0u;
                    ua &= keep; ub &= keep;
                    const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
                    a[q] = sw[0]; c[q] = sw[1];
                }
                *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(dst_px + 16 * (frag * 4 + g + h)) = u4_t{a[0], a[1], c[0], c[1]};
            }
        };

        // ---- prime layer A's input ring: rows a0 - 1 .. a0 + D - 1 -> slots 0 .. D
        if (!roleB) {
#pragma unroll 1
            for (int k = 0; k <= C::D; ++k) dma_row(a0 - 1 + k, k, a0 - 1 + k <= a1 + 1);
        }
        int s0 = 0;                                          // input-ring slot of A's input row ya - 1
        const int nsteps = a1 - a0 + 2;                      // A: rows a0 .. a1, then one idle step; B: rows ys .. ye - 1, two steps behind
        // The two roles run their step in OPPOSITE order, so that one half's matrix work lies beside the other half's LDS / memory work
        // (in lock step all eight waves contract together and then all write together: 1 050 us for the pair against 2 x 456 unfused):
        //   A:  contract row ya             | DMA of input row ya + D, epilogue of row ya -> mid ring
        //   B:  store row yb - 2, epilogue of row yb - 1 -> staging (its sums waited in registers across the barrier) | contract row yb
        f32x16 acc[2];                                       // A: inside a step; B: from the end of a step to the start of the next
        bool pend = false;                                   // B: acc holds row yb - 1's sums
#pragma unroll 1
        for (int t = 0; t < nsteps; ++t) {
            const int ya = a0 + t, yb = ya - 2;
            // A: its part of input row ya + 1 has landed (all but the youngest NDMA (D - 2) DMA instructions) and its LDS writes of the
            // previous row are done; B: its staging writes / mid-ring reads of the previous step are done
            if (!roleB) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::VMWAIT_A) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!roleB) {
                const bool live = ya <= a1;
                if (live) {
                    {
                        const int ym = (ya >= 1 ? 1 : 0) | (ya <= p.Hout - 2 ? 2 : 0), cls = p.bias_mode == 1 ? ym * 4 + xm : 0;
                        const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIASA_OFF + (cls * 64 + frag * 32 + 4 * h) * 4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = lb[2 * g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[0][4 * g + e] = v[e];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
                    const char *xl = smem + (cb * 32 + r) * C::PSTR + h * 16;
                    const char *xb[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        int sl = s0 + dy; sl = sl >= C::RING ? sl - C::RING : sl;
                        xb[dy] = xl + sl * C::ROWB;
                    }
                    contract(xb, acc);
                }
                {   // (behind the contraction: the slot written held input row ya - 2, which no step reads any more)
                    int sl = s0 + C::D + 1; sl = sl >= C::RING ? sl - C::RING : sl;
                    dma_row(ya + C::D, sl, ya + C::D <= a1 + 1);
                }
                if (live) {
                    // the row into the mid ring, zero outside the image (layer B's padding)
                    const bool inside = (unsigned)ya < (unsigned)p.Hout && (unsigned)xga < (unsigned)p.Wout;
                    write_row((lchar_t *)smem + C::MID_OFF + (t & 3) * C::MIDROW + (cb * 32 + r) * C::PSTR, acc, inside, false);
                }
                s0 = s0 + 1 >= C::RING ? 0 : s0 + 1;
            } else {
                store_row(yb - 2, yb - 2 >= ys);             // its staging row was written in the previous step (three stores, dropped when there is none)
                if (pend) write_row((lchar_t *)smem + C::STG_OFF + ((yb - 1) & 1) * C::STG + (cb * 32 + r) * C::STG_PX, acc, true, ALT);
                pend = yb >= ys && yb < ye;
                if (pend) {
                    {
                        const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIASB_OFF + (frag * 32 + 4 * h) * 4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = lb[2 * g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[0][4 * g + e] = v[e];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
                    // B's row yb reads A's rows yb - 1 .. yb + 1 = mid slots (t - 3 .. t - 1) & 3; B's column j reads A's columns j .. j + 2
                    const char *xl = smem + C::MID_OFF + (cb * 32 + r) * C::PSTR + h * 16;
                    const char *xb[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) xb[dy] = xl + ((t + 1 + dy) & 3) * C::MIDROW;
                    contract(xb, acc);
                }
            }
        }
        // ---- B's last two rows; then everything of this item is out of the rings before the next item primes them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (roleB) {
            store_row(ye - 2, ye - 2 >= ys);
            if (pend) write_row((lchar_t *)smem + C::STG_OFF + ((ye - 1) & 1) * C::STG + (cb * 32 + r) * C::STG_PX, acc, true, ALT);
        }
        __syncthreads();
        if (roleB) store_row(ye - 1, true);
        __syncthreads();
    }
}

template <typename T, bool ALT> static int launch_conv_ring2_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvRing2Cfg<T>;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ring2_kernel<T, ALT>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TWO - 1) / C::TWO) * p.B, grid = ncu;   // one 8-wave workgroup per CU
    int nseg, seg_rows;
    const int nwg = conv_ring_work(nstrip, p.Hout, grid, &nseg, &seg_rows);
    conv3x3_ring2_kernel<T, ALT><<<nwg, 512, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}

template <typename T> static int launch_conv_ring2(const ConvParams &p, hipStream_t s)
{
    if (p.stride != 1 || p.nchunk != 1 || p.npass != 1 || p.nf != 2 || p.ring != 2 || p.bias_mode > 1 || !p.w2 || !p.bias2 || p.head_w ||
        (p.epi != EPI_NONE && p.epi != EPI_RELU))
        return -2;
    return p.out_alt ? launch_conv_ring2_t<T, true>(p, s) : launch_conv_ring2_t<T, false>(p, s);
}
