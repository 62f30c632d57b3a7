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
//   * the input tile + halo of a 64-channel chunk (S = 2: 32 channels, columns split by parity so that the stride-2 operand reads are
//     conflict-free) is staged by LDS-DMA, S = 1 double-buffered: chunk c + 1 lands under chunk c's MFMAs;
//   * every VMEM instruction of the main loop is inline assembly behind COUNTED waits (hipcc does not see the DMA and would wait for
//     the weights with counts that drain it): DMA and weight loads retire in issue order (MI355X_MICROARCH.md, s_waitcnt), so
//     "the two fragments of step s have landed" is vmcnt(2 P [+ NI while the next chunk's DMA is younger than them]).
// Two workgroups per CU (LDS 64 / 48 KiB, <= 256 registers): independent workgroups overlap each other's prologue / epilogue
// (the round's lesson: conv_ring2.inl, profiles/r04_pack_one_vs_two_workgroups.txt).
// Weights: the regular packing [chunk][tap][kg][nf = 8][lane][16 B] (pack_conv_kernel with ck = CK, nf = 8, one pass).
#ifndef EMAVFI_WREG_ABL
#define EMAVFI_WREG_ABL 0   // timing-only ablations (never in the product): 1 no input DMA, 2 no stores, 4 no weight loads
#endif

template <typename T, int S> struct ConvWregCfg {
    static constexpr int CK = S == 1 ? 64 : 32;       // input channels per chunk
    static constexpr int KG = CK / 16;
    static constexpr int TH = 4, TW = 32;             // output tile
    static constexpr int IH = (TH - 1) * S + 3;       // input rows of a tile
    static constexpr int IWL = S == 1 ? TW + 2 : TW + 1;   // pixels of one LDS row segment (S = 2: of one column parity)
    static constexpr int NPIX = IH * S * IWL;
    static constexpr int PSTR = CK * 2 + 16, SP = PSTR / 16;   // odd number of 16-byte slots: conflict-free ds_read_b128
    static constexpr int NI = (NPIX * SP + 255) / 256;         // DMA instructions per wave and chunk (every wave issues exactly NI)
    static constexpr int LDS_BUF = NI * 4096;
    static constexpr int NBUF = S == 1 ? 2 : 1;
    static constexpr int LDS_BYTES = NBUF * LDS_BUF;
    static constexpr int SPC = 9 * KG;                // k-steps per chunk
    static constexpr int R = S == 1 ? 4 : 6;          // weight register ring (slots of two fragments)
    static constexpr int P = S == 1 ? 3 : 4;          // steps a weight load is ahead of its MFMAs
    static_assert(SPC % R == 0 && P < R, "the ring slot of a step must not depend on the chunk");
    static_assert(((SP & 1) == 1) && LDS_BYTES <= 80 * 1024, "two workgroups per CU");
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
__global__ __launch_bounds__(256, 2) void conv3x3_wreg_kernel(const ConvParams p)
{
    using C = ConvWregCfg<T, S>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

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
            const int px = q / C::SP, pc = q - px * C::SP;
            int ly, lx;
            if constexpr (S == 1) { ly = px / C::IWL; lx = px - ly * C::IWL; }
            else {
                ly = px / (2 * C::IWL);
                const int rem = px - ly * 2 * C::IWL, par = rem >= C::IWL ? 1 : 0;
                lx = 2 * (rem - par * C::IWL) + par;
            }
            const int gy = iy0 + ly, gx = ix0 + lx;
            const bool ok = px < C::NPIX && pc < C::SP - 1 && lx <= (C::TW - 1) * S + 2 && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win &&
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

    const int nchunk = p.nchunk;
    auto chunk_body = [&](auto last_c, int chunk) {
        constexpr bool LAST = decltype(last_c)::value;
        const int buf = C::NBUF == 2 ? (chunk & 1) : 0;
        if constexpr (!LAST && C::NBUF == 2) dma_chunk(chunk + 1, buf ^ 1);   // its buffer was last read in chunk - 1: every wave is past that barrier
        const char *wc = wbase + (size_t)chunk * C::SPC * 8192;
        const char *xb = smem + buf * C::LDS_BUF + r * C::PSTR + h * 16;
        auto xoff = [](int sc, int m) {
            const int tap = sc / C::KG, kg = sc - tap * C::KG, dy = tap / 3, dx = tap - 3 * dy;
            if (S == 1) return ((m + dy) * C::IWL + dx) * C::PSTR + kg * 32;
            return (((2 * m + dy) * 2 + (dx & 1)) * C::IWL + (dx >> 1)) * C::PSTR + kg * 32;
        };
        vec xk[2][C::TH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < C::TH; ++m) xk[0][m] = *reinterpret_cast<const vec *>(xb + xoff(0, m));
        wreg_static_for<0, C::SPC>([&](auto sc_c) {
            constexpr int sc = decltype(sc_c)::value;
            if constexpr (!LAST || sc + C::P < C::SPC) wload((sc + C::P) % C::R, wc + (size_t)(sc + C::P) * 8192);
            if constexpr (sc + 1 < C::SPC) {
#pragma unroll
                for (int m = 0; m < C::TH; ++m) xk[(sc + 1) & 1][m] = *reinterpret_cast<const vec *>(xb + xoff(sc + 1, m));
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

    // ---- epilogue: (ReLU) -> channels-last
    if (EMAVFI_WREG_ABL & 2) return;
    const int x = tx * C::TW + r;
    const bool relu = p.epi == EPI_RELU;
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
}

template <typename T, int S> static int launch_conv_wreg_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvWregCfg<T, S>;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_wreg_kernel<T, S>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const dim3 grid((p.Wout + C::TW - 1) / C::TW, (p.Hout + C::TH - 1) / C::TH, p.B);
    conv3x3_wreg_kernel<T, S><<<grid, 256, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}
template <typename T> static int launch_conv_wreg(const ConvParams &p, hipStream_t s)
{
    if (p.nf != 8 || p.npass != 1 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU) || p.out_alt || p.nchunk < 1) return -2;
    if (p.in_ps < p.ck * p.nchunk) return -2;
    if (p.stride == 1 && p.ck == 64) return launch_conv_wreg_t<T, 1>(p, s);
    if (p.stride == 2 && p.ck == 32) return launch_conv_wreg_t<T, 2>(p, s);
    return -2;
}
