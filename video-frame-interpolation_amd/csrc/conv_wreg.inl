// context_encoding.1 / .2 at the reference width (ema_vfi.py:81-82: 128 -> 256 at stride 2, 256 -> 256), 16-bit types:
// conv3x3_wreg_kernel - the packed WEIGHTS stream from L2 straight into registers, the input tile goes through the LDS.
//
// Why another kernel (round 4).  These two layers have 0.3 / 1.2 MB of weights - no register file or LDS holds
// them - and conv3x3_kernel streams them through the LDS by DMA, tap by tap, for every 256-pixel tile and output pass: 576 KiB of
// weights + 176 KiB of input per (tile, pass) of 256 -> 256 through an LDS-DMA path that ingests ~12.5 B/clk per CU
// (MI355X_MICROARCH.md, ldsdma-fill) = 62 k cycles for 37 k cycles of MFMA time - and two output passes read the input twice
// (PMC: 2.06x the algorithmic bytes; 128 -> 256: 1.76x).  Here:
//   * one 256-thread workgroup = 4 waves = the four 64-channel QUARTERS of all 256 output channels of a 4 x 32 pixel tile
//     (8 accumulator fragments = 128 registers per lane): the input is read ONCE per tile;
//   * per k-step (one tap, 16 input channels) a wave loads its two 1 KiB weight fragments with global_load_dwordx4 (lane-linear,
//     L2 hits after the first tile) P steps ahead into a register ring, reads 4 pixel fragments from the LDS and issues 8 MFMAs:
//     2 KiB of L2 -> register traffic and 4 KiB of LDS reads per 256 cycles of matrix pipe, nothing through the LDS-DMA but the
//     input tile (S = 1: 2.8 B/clk per CU);
//   * the input tile + halo of a 64-channel chunk (S = 2: 32 channels, columns split by parity and 64-byte pixels XOR-swizzled so that
//     the stride-2 operand reads are conflict-free) is staged by LDS-DMA into one of two buffers: chunk c + 1 lands under chunk c's MFMAs;
//   * every VMEM instruction of the main loop is inline assembly behind COUNTED waits (hipcc does not see the DMA and would wait for
//     the weights with counts that drain it): DMA and weight loads retire in issue order (MI355X_MICROARCH.md, s_waitcnt), so
//     "the two fragments of step s have landed" is vmcnt(2 P [+ NI while the next chunk's DMA is younger than them]).
// Two workgroups per CU (LDS 64 / 80 KiB, <= 256 registers): independent workgroups overlap each other's prologue / epilogue
// (the round's lesson: conv_ring2.inl, profiles/r04_pack_one_vs_two_workgroups.txt).
// Weights: the regular packing [chunk][tap][kg][nf = 8][lane][16 B] (pack_conv_kernel with ck = CK, nf = 8, one pass).
#ifndef EMAVFI_WREG_ABL
#define EMAVFI_WREG_ABL 0   // timing-only ablations (never in the product): 1 no input DMA, 2 no epilogue, 4 no weight loads, 8 no pool reduction, 16 direct stores
#endif

#ifndef EMAVFI_WREG_R1
#define EMAVFI_WREG_R1 4
#define EMAVFI_WREG_P1 3
#endif

template <typename T, int S> struct ConvWregCfg {
    static constexpr int CK = S == 1 ? 64 : 32;       // input channels per chunk
    static constexpr int KG = CK / 16;
    static constexpr int TH = 4, TW = 32;             // output tile
    static constexpr int IH = (TH - 1) * S + 3;       // input rows of a tile
    static constexpr int IWL = S == 1 ? TW + 2 : TW + 1;   // pixels of one LDS row segment (S = 2: of one column parity)
    static constexpr int NPIX = IH * S * IWL;
    // S = 1: 144-byte pixels (an odd number of 16-byte slots: conflict-free ds_read_b128).  S = 2: 64-byte pixels, piece c of pixel p
    // in slot c ^ ((p >> 2) & 3) - the padded form (80 bytes) does not leave room for a second buffer.  Conflict-free: a ds_read_b128 is
    // serviced in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32: the other piece); the four pixels of a group that
    // share p mod 4 (one 64-byte quarter of a 256-byte bank row) are a, a + 12, a + 20, a + 24 / a + 4, a + 8, a + 16, a + 28:
    // their (p >> 2) & 3 are four different values for every a, so the XOR sends them to four different quarters
    static constexpr bool SWZ = S == 2;
    static constexpr int PSTR = SWZ ? CK * 2 : CK * 2 + 16, SP = PSTR / 16;
    static constexpr int NI = (NPIX * SP + 255) / 256;         // DMA instructions per wave and chunk (every wave issues exactly NI)
    static constexpr int LDS_BUF = NI * 4096;
    static constexpr int NBUF = 2;
    static constexpr int LDS_BYTES = NBUF * LDS_BUF;
    static constexpr int SPC = 9 * KG;                // k-steps per chunk
    static constexpr int R = S == 1 ? EMAVFI_WREG_R1 : 6;          // weight register ring (slots of two fragments)
    static constexpr int P = S == 1 ? EMAVFI_WREG_P1 : 4;          // steps a weight load is ahead of its MFMAs
    static_assert(SPC % R == 0 && P < R, "the ring slot of a step must not depend on the chunk");
    static_assert(((SP & 1) == 1 || SWZ) && LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

typedef __attribute__((ext_vector_type(4))) unsigned wreg_u4;
// f(integral_constant<int, I>) for I = I0 .. N - 1: the step index has to be a constant expression (s_waitcnt immediates, register arrays)
template <int I, int N, typename F> __device__ __forceinline__ void wreg_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wreg_static_for<I + 1, N>(f);
    }
}

template <typename T, int S>
__global__ __launch_bounds__(256, 2) void conv3x3_wreg_kernel(const ConvParams p, const int ntx, const int nty)
{
    using C = ConvWregCfg<T, S>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order (as warp_tiled_kernel's): workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the XCD), each XCD
    // gets a contiguous run of tiles, so the tiles that share halo rows and columns meet in ONE L2 instead of fetching them from HBM
    // in eight (PMC: 256 -> 256 read 412 MB for a 236 MB input in plain order).  Bijective for any grid size; placement only affects speed.
    int tile;
    {
        const int nwg = (int)gridDim.x, grp = (int)blockIdx.x & 7, kk = (int)blockIdx.x >> 3, qq = nwg >> 3, rr = nwg & 7;
        tile = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
    }
    const int b = tile / (ntx * nty), trem = tile - b * (ntx * nty), ty = trem / ntx, tx = trem - ty * ntx;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    // diagnostic build only (tools/wreg_stamps.py): s_memtime at the seams, one row of 8 per wave, in the buffer run_conv put into out_planar
    unsigned long long st[4];
#define WREG_STAMP(k) st[k] = __builtin_amdgcn_s_memtime()
#else
#define WREG_STAMP(k)
#endif
    WREG_STAMP(0);
    // ---- accumulators start at the bias: wave w owns output channels [64 w, 64 w + 64)
    f32x16 acc[C::TH][2];
    {
        const float *bp = p.bias + wave * 64;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float bv = bp[n * 32 + acc_channel(i, h)];
#pragma unroll
                for (int m = 0; m < C::TH; ++m) acc[m][n][i] = bv;
            }
    }

    // ---- this lane's DMA sources (byte offsets inside the sample for chunk 0; ~0: the zero page)
    const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
    const char *zeros = (const char *)p.zeros;
    unsigned doff[C::NI];
    {
        const int iy0 = ty * C::TH * S - 1, ix0 = tx * C::TW * S - 1;
        const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int q = (i * 4 + wave) * 64 + lane;
            const int px = q / C::SP;
            int pc = q - px * C::SP;
            if constexpr (C::SWZ) pc ^= (px >> 2) & 3;
            int ly, lx;
            if constexpr (S == 1) { ly = px / C::IWL; lx = px - ly * C::IWL; }
            else {
                ly = px / (2 * C::IWL);
                const int rem = px - ly * 2 * C::IWL, par = rem >= C::IWL ? 1 : 0;
                lx = 2 * (rem - par * C::IWL) + par;
            }
            const int gy = iy0 + ly, gx = ix0 + lx;
            const bool ok = px < C::NPIX && (C::SWZ || pc < C::SP - 1) && lx <= (C::TW - 1) * S + 2 && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win &&
                            !(EMAVFI_WREG_ABL & 1);
            doff[i] = ok ? ((unsigned)gy * (unsigned)p.Win + (unsigned)gx) * pixbytes + (unsigned)pc * 16u : 0xffffffffu;
        }
    }
    auto dma_chunk = [&](int chunk, int buf) {
        const char *gc = gin + (size_t)chunk * C::CK * sizeof(T);
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const char *src = doff[i] != 0xffffffffu ? gc + doff[i] : zeros;
            const unsigned dst = lds0 + (unsigned)(buf * C::LDS_BUF + (i * 4 + wave) * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
        }
    };
    // ---- weights: step sg = (chunk * 9 + tap) * KG + kg is 8 KiB; this wave's two fragments are KiB 2 w, 2 w + 1 of it
    const char *wbase = (const char *)p.w + wave * 2048;
    const unsigned wlane = (unsigned)lane * 16u;
    wreg_u4 wq[C::R][2];
    auto wload = [&](int slot, const char *ws) {
        if (EMAVFI_WREG_ABL & 4) { wq[slot][0] = wreg_u4{0, 0, 0, 0}; wq[slot][1] = wreg_u4{0, 0, 0, 0}; return; }
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wq[slot][0]) : "v"(wlane), "s"(ws) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wq[slot][1]) : "v"(wlane), "s"(ws) : "memory");
    };

    dma_chunk(0, 0);
#pragma unroll
    for (int q = 0; q < C::P; ++q) wload(q, wbase + (size_t)q * 8192);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((EMAVFI_WREG_ABL & 4) ? 0 : 2 * C::P) : "memory");   // chunk 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    WREG_STAMP(1);
    const int nchunk = p.nchunk;
    auto chunk_body = [&](auto last_c, int chunk) {
        constexpr bool LAST = decltype(last_c)::value;
        const int buf = C::NBUF == 2 ? (chunk & 1) : 0;
        if constexpr (!LAST && C::NBUF == 2) dma_chunk(chunk + 1, buf ^ 1);   // its buffer was last read in chunk - 1: every wave is past that barrier
        const char *wc = wbase + (size_t)chunk * C::SPC * 8192;
        const char *xb = smem + buf * C::LDS_BUF + r * C::PSTR + h * 16;
        // (S = 2: an opaque copy of the lane's column per chunk, or hipcc hoists all 72 swizzled addresses out of the chunk loop and spills)
        unsigned r_op = (unsigned)r;
        asm volatile("" : "+v"(r_op));
        // operand address of step sc (tap, k-group), output row m: S = 1 a constant offset; S = 2 the parity-split, swizzled tile
        auto xptr = [&](int sc, int m) -> const char * {
            const int tap = sc / C::KG, kg = sc - tap * C::KG, dy = tap / 3, dx = tap - 3 * dy;
            if constexpr (!C::SWZ) return xb + ((m + dy) * C::IWL + dx) * C::PSTR + kg * 32;
            else {
                const unsigned pp = r_op + (unsigned)(((2 * m + dy) * 2 + (dx & 1)) * C::IWL + (dx >> 1));
                return smem + buf * C::LDS_BUF + pp * 64u + ((((pp >> 2) ^ (unsigned)h ^ (unsigned)(2 * kg)) & 3u) << 4);
            }
        };
        vec xk[2][C::TH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < C::TH; ++m) xk[0][m] = *reinterpret_cast<const vec *>(xptr(0, m));
        wreg_static_for<0, C::SPC>([&](auto sc_c) {
            constexpr int sc = decltype(sc_c)::value;
            if constexpr (!LAST || sc + C::P < C::SPC) wload((sc + C::P) % C::R, wc + (size_t)(sc + C::P) * 8192);
            if constexpr (sc + 1 < C::SPC) {
#pragma unroll
                for (int m = 0; m < C::TH; ++m) xk[(sc + 1) & 1][m] = *reinterpret_cast<const vec *>(xptr(sc + 1, m));
            }
            // the two fragments of step sc have landed: younger than them are the loads of the next P steps and, in the first P
            // steps of a chunk, the next chunk's DMA (issued after those steps' loads, which the previous chunk prefetched)
            constexpr int ahead = LAST ? (C::SPC - 1 - sc < C::P ? C::SPC - 1 - sc : C::P) : C::P;
            constexpr int n_young = (EMAVFI_WREG_ABL & 4) ? 63 : 2 * ahead + ((!LAST && C::NBUF == 2 && sc < C::P) ? C::NI : 0);
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wq[sc % C::R][0]), "+v"(wq[sc % C::R][1]) : "n"(n_young) : "memory");
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < C::TH; ++m) mma_kg(acc[m][n], __builtin_bit_cast(vec, wq[sc % C::R][n]), xk[sc & 1][m]);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!LAST) {
            if constexpr (C::NBUF == 2) {
                // chunk + 1 (this wave's part) landed long ago: it is older than the last steps' weight loads, which the waits above
                // retired; what is still in flight are the 2 P loads of the next chunk's first steps
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((EMAVFI_WREG_ABL & 4) ? 0 : 2 * C::P) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // every wave has read the chunk
                asm volatile("" ::: "memory");
                dma_chunk(chunk + 1, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    };
#pragma unroll 1
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) chunk_body(std::false_type{}, chunk);
    chunk_body(std::true_type{}, nchunk - 1);
    WREG_STAMP(2);
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    auto stamp_out = [&]() {
        const unsigned row = (unsigned)tile * 4 + wave;
        if (p.out_planar && row < DEFORM_STAMP_ROWS && lane == 0) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(p.out_planar) + (size_t)row * 8;
            d[0] = st[0]; d[1] = st[1] - st[0]; d[2] = st[2] - st[1]; d[3] = __builtin_amdgcn_s_memtime() - st[2]; d[4] = 1;
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            d[5] = hwid; d[6] = S;
        }
    };
#endif

    // ---- epilogue: (ReLU) -> channels-last, or (context_encoding.2 + AdaptiveAvgPool2d, ema_vfi.py:82-83) this tile's per-channel sums
    if ((EMAVFI_WREG_ABL & 2) && p.B > 0) return;   // (a condition hipcc cannot fold: the MFMAs stay)
    const int x = tx * C::TW + r;
    const bool relu = p.epi == EPI_RELU;
    if (p.pool_part) {
        // the layer's output is only ever averaged: sum the values the tensor would have held (rounded to T) over the tile's pixels inside
        // the image - rows in registers, the 32 columns by DPP steps - and write 256 floats
        // per tile; pool_partial_kernel<float> and ctx_finish_kernel add the tiles in a fixed order (deterministic, no atomics)
        float sum[2][16];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t = 0.0f;
#pragma unroll
                for (int m = 0; m < C::TH; ++m) {
                    float v = acc[m][n][i];
                    if (relu) v = fmaxf(v, 0.0f);
                    v = (float)(T)v;
                    t += (ty * C::TH + m < p.Hout && x < p.Wout) ? v : 0.0f;
                }
                // sum over the 32 lanes of this half in VALU data-parallel-primitive steps (no LDS round trips): quads, 8, 16 lanes by the
                // mirror permutations (every lane of a group already holds the group's sum), then row_bcast:15 adds a row's lane 15
                // into the next row: lanes 16-31 / 48-63 end with the half's total
                if (!(EMAVFI_WREG_ABL & 8)) {
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, false));   // row_half_mirror
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xF, 0xF, false));   // row_mirror
                    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1, 3
                }
                sum[n][i] = t;
            }
        if (r == 31) {
            float *dst = p.pool_part + (size_t)tile * 256 + wave * 64;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[n * 32 + acc_channel(i, h)] = sum[n][i];
        }
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
        stamp_out();
#endif
        return;
    }
    if (p.cstore % 8 == 0 && !(EMAVFI_WREG_ABL & 16)) {
        // through the LDS (the tile buffers are dead once every wave is past the last chunk): a wave's 64 channels of a pixel are one
        // 128-byte line, and a store instruction then writes 8 whole lines instead of 32 bytes of 32 different lines.  Staging:
        // [128 pixels][8 slots of 16 bytes], slot s of pixel q at s ^ ((q >> 1) & 7) - the 16 lanes of a ds_write_b128 service group
        // ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) then hit 16 different bank quads; the read-back is linear
        typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
        typedef __attribute__((address_space(3))) char lchar_t;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        lchar_t *stg = (lchar_t *)smem + wave * 16384;
#pragma unroll
        for (int m = 0; m < C::TH; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    unsigned a[2], c[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v0 = acc[m][n][4 * g + 2 * q], v1 = acc[m][n][4 * g + 2 * q + 1], u0 = acc[m][n][4 * (g + 1) + 2 * q], u1 = acc[m][n][4 * (g + 1) + 2 * q + 1];
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        const auto sw = __builtin_amdgcn_permlane32_swap(pack16x2<T>(v0, v1, false), pack16x2<T>(u0, u1, false), false, false);
                        a[q] = sw[0]; c[q] = sw[1];
                    }
                    const int slot = (n * 4 + g + h) ^ ((r >> 1) & 7);
                    *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + (m * 32 + r) * 128 + slot * 16) = u4_t{a[0], a[1], c[0], c[1]};
                }
        char *obase = reinterpret_cast<char *>(p.out) + ((size_t)b * p.Hout * p.Wout * p.out_ps + p.out_coff + wave * 64) * sizeof(T);
        const unsigned pixbytes_o = (unsigned)p.out_ps * (unsigned)sizeof(T);
#pragma unroll
        for (int it = 0; it < C::TH * 4; ++it) {
            const int idx = it * 64 + lane, q = idx >> 3, sl = (idx & 7) ^ ((q >> 1) & 7);
            const int y = ty * C::TH + (q >> 5), xo = tx * C::TW + (q & 31);
            const u4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + idx * 16);
            if (y < p.Hout && xo < p.Wout && wave * 64 + sl * 8 < p.cstore)
                *reinterpret_cast<u4_t *>(obase + ((size_t)y * p.Wout + xo) * pixbytes_o + sl * 16) = v;
        }
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
        stamp_out();
#endif
        return;
    }
#pragma unroll
    for (int m = 0; m < C::TH; ++m) {
        const int y = ty * C::TH + m;
        if (y >= p.Hout || x >= p.Wout) continue;
        const size_t pix = ((size_t)b * p.Hout + y) * p.Wout + x;
        T *ob = reinterpret_cast<T *>(p.out) + pix * p.out_ps + p.out_coff + wave * 64;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int limit = p.cstore - (wave * 2 + n) * 32;
            if (limit <= 0) continue;
            if (relu) store_frag(ob + n * 32, acc[m][n], h, limit, [](float v, int) { return fmaxf(v, 0.0f); });
            else store_frag(ob + n * 32, acc[m][n], h, limit, [](float v, int) { return v; });
        }
    }
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    stamp_out();
#endif
}

// tiles per image of conv3x3_wreg_kernel (= rows of ConvParams::pool_part per sample)
static inline int conv_wreg_tiles(int Hout, int Wout) { return ((Wout + 31) / 32) * ((Hout + 3) / 4); }

template <typename T, int S> static int launch_conv_wreg_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvWregCfg<T, S>;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_wreg_kernel<T, S>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ntx = (p.Wout + C::TW - 1) / C::TW, nty = (p.Hout + C::TH - 1) / C::TH;
    const long long ntiles = (long long)ntx * nty * p.B;
    if (ntiles > 0x7fffffffLL) return -2;
    conv3x3_wreg_kernel<T, S><<<(unsigned)ntiles, 256, C::LDS_BYTES, s>>>(p, ntx, nty);
    return (int)hipGetLastError();
}
template <typename T> static int launch_conv_wreg(const ConvParams &p, hipStream_t s)
{
    if (p.pool_part && p.cstore != 256) return -2;
    if (p.nf != 8 || p.npass != 1 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU) || p.out_alt || p.nchunk < 1) return -2;
    if (p.in_ps < p.ck * p.nchunk) return -2;
    if (p.stride == 1 && p.ck == 64) return launch_conv_wreg_t<T, 1>(p, s);
    if (p.stride == 2 && p.ck == 32) return launch_conv_wreg_t<T, 2>(p, s);
    return -2;
}
