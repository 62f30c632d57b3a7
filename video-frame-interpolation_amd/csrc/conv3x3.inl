// 3x3 convolution (pad 1, stride 1 or 2) as an implicit GEMM on the CDNA4 matrix cores.
// Replaces nn.Conv2d at /root/reference/src/models/ema_vfi.py:7-14 (every conv / conv_block on
// the forward path: :73-76, :80-82, :90-92, :35-41, :103-105).
//
// One 256-thread workgroup (4 waves, one per SIMD) produces a tile of 4*MF rows x 32 columns of
// output pixels for NF*32 output channels:
//   * the input tile + 1-pixel halo for CK input channels is staged ONCE in LDS and re-read by
//     all 9 taps, so HBM sees each input element about once ((TH+2)(TW+2)/(TH*TW) = 1.33x for
//     the 8x32 tile).  Staging is direct global->LDS DMA (global_load_lds_dwordx4, 16 B per
//     lane): the LDS image is slot-linear ([pixel][SP 16-byte slots], the last slot of a pixel
//     is padding) so a wave-instruction's 64 lanes write 1 KiB contiguously while each lane's
//     SOURCE address is computed per slot; lanes that map to padding or to pixels outside the
//     image (the conv's zero padding) read a 16-byte zero page.  All of a wave's DMA
//     instructions are in flight together (the first version staged through registers with
//     one load in flight per thread: 11 serialized L2/HBM round trips per tile);
//   * the packed weights of one tap (KG*NF KiB, already in MFMA fragment order = lane-linear,
//     exactly what the DMA writes) sit in a 3-slot LDS ring filled by DMA two taps ahead; the
//     per-tap barrier is a raw s_barrier behind a COUNTED s_waitcnt vmcnt(N) that leaves the newest
//     tap's DMA in flight (2 slots / one tap ahead where a third slot would cost occupancy);
//   * a wave owns MF rows of 32 pixels (MF pixel fragments) x NF channel fragments:
//     D[cout][pixel] accumulates in MF*NF 32x32 fp32 tiles.
// LDS pixel stride is an odd number of 16-byte slots (LdsPix), so the ds_read_b128 operand
// fetches of 32 consecutive pixels are bank-conflict free for stride 1.
//
// Kernels in this file (which layer runs where: launch_conv16 / launch_conv_mfma16 at the end):
//   conv3x3_kernel            the tile-per-workgroup kernel described above (every shape; the only one for fp32, chunked K)
//   conv3x3_s2ring_kernel     64 -> 128 at stride 2: weights in registers, input rows through an LDS ring (one strip per workgroup)
//   conv3x3_ring_kernel       (conv_ring.inl) that structure at stride 1: the 64 -> 64 layers and reconstruction.0 (67 -> 64) (product)
//   conv3x3_ringtail_kernel   (conv_ring_tail.inl) reconstruction.1 + .2 (64 -> 32 -> 3) through a second LDS ring (product)
//   conv3x3_persist_kernel    persistent, nine taps' weights resident, 16 x 32 tiles, 8 waves in lock step (32x32x16 MFMAs)
//   conv3x3_persist16_kernel  the same on v_mfma_f32_16x16x32 with an unpadded XOR-swizzled tile: 64 -> 32 and 64 -> planes when the fused
//                             ring kernels are switched off, 64 -> 64 with EMAVFI_CONV_RING=0 (round 2's ping-pong variant of it,
//                             conv3x3_pingpong16_kernel, was removed in round 4: the ring kernels replaced it in the product)
//   (round 2's conv3x3_pingpong_kernel - the schedule on 32x32x16 MFMAs - and conv3x3_tail_kernel - reconstruction.1 + .2 through
//    the LDS - were measurement-only experiments that lost and were removed in round 3)
#include "common.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

// Source address of one 16-byte DMA piece: piece pc of input pixel (gy, gx) of the sample starting at gin, or the zero page when
// the pixel lies outside the image (or `valid` is false).  32-bit offset arithmetic (a sample is < 4 GiB: checked at the API) and
// a select instead of a branch: the straightforward 64-bit form cost ~25 instructions, two of them 64-bit multiply-adds, and an
// EXEC-masked branch PER DMA instruction - 590 cycles each beside the other group's MFMAs (tools/conv_stamps.py).
__device__ __forceinline__ const char *conv_dma_src(const char *gin, const char *zeros, int gy, int gx, int pc, int Hin, int Win, unsigned pixbytes, bool valid)
{
    const bool ok = valid && (unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win;
    const unsigned pix = __umul24((unsigned)gy & 0xffffffu, (unsigned)Win) + (unsigned)gx;
    const unsigned off = pix * pixbytes + (unsigned)pc * 16u;
    return ok ? gin + off : zeros;
}

#ifndef EMAVFI_CONV_TILE_PIPE
#define EMAVFI_CONV_TILE_PIPE 2   // tile-per-workgroup kernel, 16-bit: weight fragments read this many steps ahead (0: the plain loop)
#endif
#ifndef EMAVFI_CONV_INTERLEAVE
#define EMAVFI_CONV_INTERLEAVE 1
#endif
#ifndef EMAVFI_CONV_STAGED_STORE
#define EMAVFI_CONV_STAGED_STORE 1   // tile-per-workgroup kernel: 16-bit channels-last outputs are transposed through LDS (0: direct 32-byte stores)
#endif
#ifndef EMAVFI_CONV_PIPELINE
#define EMAVFI_CONV_PIPELINE 1   // persistent kernel: operands one step ahead of their MFMAs (0 = the compiler's own order)
#endif

// Persistent kernel: touch the next tile's input lines (L2 prefetch) in front of the MFMA loop.  OFF: the convolution itself
// gains 0-3.5 %, but the board answers with a lower clock (2034 vs 2157 MHz under load, tools/power_trace.py) and the whole
// step loses 1.6-4 % - kept as a switch because it shows the step is power-managed, not latency-bound.
#ifndef EMAVFI_CONV_PREFETCH
#define EMAVFI_CONV_PREFETCH 0
#endif

template <typename T, int CK, int NF, int S> struct ConvCfg {
    using D = DT<T>;
    static constexpr int MF = (S == 1) ? 2 : 1;
    static constexpr int TH = 4 * MF, TW = 32;
    static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    static constexpr int PSTR = LdsPix<T, CK>::BYTES;
    static constexpr int PIECES = CK * (int)sizeof(T) / 16;
    static constexpr int KG = CK / D::CHKG;
    static constexpr int WTAP = KG * NF * 1024;  // bytes of packed weights per tap
    static constexpr int WINST = KG * NF;        // 1-KiB DMA wave-instructions per tap
    static constexpr int SP = PSTR / 16;         // 16-byte slots per staged pixel (last = padding)
    static constexpr int NSLOT = IH * IW * SP;
    static constexpr int NINST = (NSLOT + 63) / 64;  // DMA wave-instructions per input tile
    static constexpr int LDS_IN = NINST * 1024;
    // weight ring depth: 3 slots (DMA two taps ahead, counted vmcnt) when that neither overflows the
    // 160 KiB LDS nor lowers the number of workgroups per CU; otherwise 2 slots (one tap ahead)
    static constexpr int LDS2 = LDS_IN + 2 * WTAP, LDS3 = LDS_IN + 3 * WTAP;
    static constexpr int WBUF = (LDS3 <= 160 * 1024 && (160 * 1024 / LDS3) == (160 * 1024 / LDS2)) ? 3 : 2;
    static constexpr int AHEAD = WBUF - 1;
    static constexpr int LDS_BYTES = LDS_IN + WBUF * WTAP;
    static_assert(CK % D::CHKG == 0, "CK must be a whole number of k-groups");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
};

// ---- shared by both kernels: accumulator initialisation and epilogue for MF pixel rows ----
// `y0` = first output row of this wave, x = output column of this lane.
template <int MF, int NF>
__device__ __forceinline__ void conv_init_acc(f32x16 (&acc)[MF][NF], const ConvParams &p, int b, int pass, int y0, int x, int h)
{
    const int coutpad = p.npass * NF * 32;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const float *bp = p.bias + pass * NF * 32;
        if (p.bias_mode == 1) {
            // motion_estimation.0: the spatially constant context half of the concatenated input
            // (ema_vfi.py:124) is folded into a bias that depends only on which taps fall inside
            // the image; the table is indexed by that border class (see ctx_finish_kernel).
            const int y = y0 + m;
            const int ym = (y >= 1 ? 1 : 0) | (y <= p.Hout - 2 ? 2 : 0);
            const int xm = (x >= 1 ? 1 : 0) | (x <= p.Wout - 2 ? 2 : 0);
            bp += ((size_t)b * 16 + ym * 4 + xm) * coutpad;
        }
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = bp[n * 32 + acc_channel(i, h)];
    }
}

// lane (r, h) holds pixel r = column x, 4 consecutive channels per register quad
template <typename T, int MF, int NF>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[MF][NF], const ConvParams &p, int b, int pass, int y0, int x, int h)
{
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const int y = y0 + m;
        if (y >= p.Hout || x >= p.Wout) continue;
        const size_t pix = ((size_t)b * p.Hout + y) * p.Wout + x;
        if (p.epi == EPI_PLANAR || p.epi == EPI_PLANAR_TANH01) {
            // flow head / reconstruction tail: <= 4 real channels, written as NCHW fp32 planes
            if (h == 0 && pass == 0) {
                const size_t plane = (size_t)p.Hout * p.Wout;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < p.nplanes) {
                        float v = acc[m][0][c];
                        if (p.round16) {  // fp16 tensors under autocast: one rounding per op
                            v = (float)(half_t)v;
                            if (p.epi == EPI_PLANAR_TANH01) v = (float)(half_t)((float)(half_t)tanhf(v) + 1.0f) / 2.0f;
                        } else if (p.epi == EPI_PLANAR_TANH01) v = (tanhf(v) + 1.0f) / 2.0f;  // ema_vfi.py:106,146
                        p.out_planar[((size_t)b * p.nplanes + c) * plane + (size_t)y * p.Wout + x] = v;
                    }
            }
            continue;
        }
        if (p.epi == EPI_OM) {
            // mask = sigmoid(second chunk), ema_vfi.py:59; routed to channels 18..26 at pack time
            const auto om_act = [](float v, int c) { return (c >= 18 && c < 27) ? 1.0f / (1.0f + expf(-v)) : v; };
            const auto om_act16 = [](float v, int c) {  // offset_conv's output and sigmoid(mask) are fp16 tensors under autocast
                v = (float)(half_t)v;
                return (c >= 18 && c < 27) ? (float)(half_t)(1.0f / (1.0f + expf(-v))) : v;
            };
            if (p.round16) store_frag(reinterpret_cast<float *>(p.out) + pix * p.out_ps, acc[m][0], h, p.cstore, om_act16);
            else store_frag(reinterpret_cast<float *>(p.out) + pix * p.out_ps, acc[m][0], h, p.cstore, om_act);
            continue;
        }
        T *ob = reinterpret_cast<T *>(p.out) + pix * p.out_ps + p.out_coff + pass * NF * 32;
        if constexpr (sizeof(T) == 2) {
            if (p.x3) {   // EMAVFI_F32X3: the fp32 result as two f16 halves, hi = f16(v), lo = f16(v - hi)
                const bool relu = p.epi == EPI_RELU;
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const int limit = p.cstore - (pass * NF + n) * 32;
                    if (limit <= 0) continue;
                    store_frag(ob + n * 32, acc[m][n], h, limit, [relu](float v, int) { return relu ? fmaxf(v, 0.0f) : v; });
                    store_frag(ob + p.out_lo_off + n * 32, acc[m][n], h, limit, [relu](float v, int) {
                        const float r = relu ? fmaxf(v, 0.0f) : v;
                        return r - (float)(T)r;
                    });
                    if (p.out32)
                        store_frag(p.out32 + pix * p.out32_ps + p.out_coff + (pass * NF + n) * 32, acc[m][n], h, limit,
                                   [relu](float v, int) { return relu ? fmaxf(v, 0.0f) : v; });
                }
                continue;
            }
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int limit = p.cstore - (pass * NF + n) * 32;
            if (limit <= 0) continue;
            if (p.epi == EPI_RELU) store_frag(ob + n * 32, acc[m][n], h, limit, [](float v, int) { return fmaxf(v, 0.0f); });
            else store_frag(ob + n * 32, acc[m][n], h, limit, [](float v, int) { return v; });
        }
    }
}

// waves per SIMD the register allocation must leave room for = workgroups per CU the LDS footprint allows (each
// workgroup is 4 waves, one per SIMD), capped at 2: without it hipcc spends up to 512 registers on the 8-fragment
// instances and a second workgroup cannot become resident
template <typename T, int CK, int NF, int S> constexpr int conv_wg_per_cu()
{
    return (160 * 1024 / ConvCfg<T, CK, NF, S>::LDS_BYTES) >= 2 ? 2 : 1;
}

template <typename T, int CK, int NF, int S>
__global__ __launch_bounds__(256, (conv_wg_per_cu<T, CK, NF, S>())) void conv3x3_kernel(const ConvParams p)
{
    using C = ConvCfg<T, CK, NF, S>;
    using vec = typename DT<T>::vec;
    constexpr int MF = C::MF, IW = C::IW, PSTR = C::PSTR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_in = smem;
    char *lds_w = smem + C::LDS_IN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tx = blockIdx.x, ty = blockIdx.y;
    const int b = blockIdx.z / p.npass, pass = blockIdx.z - b * p.npass;

    // ---- accumulators start at the bias ----
    f32x16 acc[MF][NF];
    conv_init_acc<MF, NF>(acc, p, b, pass, ty * C::TH + wave * MF, tx * 32 + r, h);

    const int iy0 = ty * C::TH * S - 1, ix0 = tx * 32 * S - 1;
    const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
    const char *wpass = (const char *)p.w + (size_t)pass * p.nchunk * 9 * C::WTAP;

    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const char *zeros = (const char *)p.zeros;
    const int npieces = p.in_pieces > 0 ? p.in_pieces : C::PIECES;

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        if (chunk) {
            __syncthreads();
        }
        // ---- DMA the input tile (+halo) for this channel chunk, and tap 0's weights ----
        // EMAVFI_F32X3: virtual chunk 3 c + t = real chunk c, term t: (x_hi, w_hi), (x_hi, w_lo), (x_lo, w_hi); term 1 reuses term 0's tile
        const int rchunk = p.x3 ? chunk / 3 : chunk, term = p.x3 ? chunk - 3 * rchunk : 0;
        const char *gchunk = gin + ((size_t)rchunk * CK + (term == 2 ? (size_t)p.x3_lo_off : 0)) * sizeof(T);
#pragma unroll
        for (int i = 0; i < (C::NINST + 3) / 4; ++i) {
            const int j = i * 4 + wave;
            if (j < C::NINST && term != 1) {
                const int sl = j * 64 + lane;
                const int pix = sl / C::SP, pc = sl - pix * C::SP;
                const int ly = pix / IW, lx = pix - ly * IW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                const bool ok = sl < C::NSLOT && pc < npieces && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                const char *src = ok ? gchunk + ((size_t)gy * p.Win + gx) * p.in_ps * sizeof(T) + pc * 16 : zeros;
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(lds_in + j * 1024), 16, 0, 0);
            }
        }
        const char *wc = wpass + (size_t)chunk * 9 * C::WTAP;
        // weights: a 3-slot LDS ring filled by DMA two taps ahead.  `issue_w(t)` = this wave's share
        // of tap t's 1-KiB blocks; every wave issues the same number NW or NW-1 of them, so a
        // counted s_waitcnt leaves exactly the newest tap in flight across the barrier.
        auto issue_w = [&](int t) {
#pragma unroll
            for (int i = 0; i < (C::WINST + 3) / 4; ++i) {
                const int j = i * 4 + wave;
                if (j < C::WINST)
                    __builtin_amdgcn_global_load_lds((gptr_t *)(wc + (size_t)t * C::WTAP + j * 1024 + lane * 16),
                                                     (lptr_t *)(lds_w + (t % C::WBUF) * C::WTAP + j * 1024), 16, 0, 0);
            }
        };
        issue_w(0);
        if (C::AHEAD == 2) issue_w(1);
        __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier: tile + first taps landed

#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // the slot being refilled was last read one barrier ago (3-slot ring) / is the idle one (2-slot)
            if (tap + C::AHEAD < 9) issue_w(tap + C::AHEAD);
            const int dy = tap / 3, dx = tap - 3 * dy;
            const char *xb[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m)
                xb[m] = lds_in + (((wave * MF + m) * S + dy) * IW + r * S + dx) * PSTR + h * 16;
            const char *wb = lds_w + (tap % C::WBUF) * C::WTAP + lane * 16;
            if constexpr (sizeof(T) == 2 && EMAVFI_CONV_TILE_PIPE) {
                // 16-bit: operands EMAVFI_CONV_TILE_PIPE weight fragments (and one k-group of pixel pieces) ahead of their MFMAs,
                // every step fenced.  The plain loop below read a step's two operands right in front of its MFMAs: one LDS round
                // trip per step and wave, which is what bounded the streamed-weight layers (context_encoding.1 / .2: 29 % / 42 % of
                // the MFMA rate)
                constexpr int PF = EMAVFI_CONV_TILE_PIPE, NS = C::KG * NF;
                vec xk[2][MF], wq[PF + 1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MF; ++m) xk[0][m] = *reinterpret_cast<const vec *>(xb[m]);
#pragma unroll
                for (int q = 0; q < PF && q < NS; ++q) wq[q] = *reinterpret_cast<const vec *>(wb + q * 1024);
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const int kg = q / NF, n = q - kg * NF;
                    if (q + PF < NS) wq[(q + PF) % (PF + 1)] = *reinterpret_cast<const vec *>(wb + (q + PF) * 1024);
                    if (n == 0 && kg + 1 < C::KG) {
#pragma unroll
                        for (int m = 0; m < MF; ++m) xk[(kg + 1) & 1][m] = *reinterpret_cast<const vec *>(xb[m] + (kg + 1) * 32);
                    }
#pragma unroll
                    for (int m = 0; m < MF; ++m) mma_kg(acc[m][n], wq[q % (PF + 1)], xk[kg & 1][m]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int kg = 0; kg < C::KG; ++kg) {
                    vec xv[MF];
#pragma unroll
                    for (int m = 0; m < MF; ++m) xv[m] = *reinterpret_cast<const vec *>(xb[m] + kg * 32);
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        const vec wv = *reinterpret_cast<const vec *>(wb + (kg * NF + n) * 1024);
#pragma unroll
                        for (int m = 0; m < MF; ++m) mma_kg(acc[m][n], wv, xv[m]);
                    }
                }
            }
            if (tap < 8) {
                // tap+1's weights must have landed (issued a whole tap ago); tap+2's may stay in
                // flight.  This wave issued nw2 DMA instructions for tap+2 (0 when tap+2 > 8).
                const int nw2 = (C::AHEAD == 2 && tap + 2 < 9) ? (C::WINST - wave + 3) / 4 : 0;
                if (nw2 >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (nw2 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (nw2 == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    // 16-bit channels-last outputs: through LDS, so that a store instruction writes whole pixels of consecutive addresses instead
    // of 32 bytes of 64 different pixels (conv_first.inl measured 25 % on this store pattern).  The tile and the weight ring are dead.
    constexpr int STG_PX = NF * 64 + 16, STG_WAVE = MF * 32 * STG_PX;
    if constexpr (sizeof(T) == 2 && EMAVFI_CONV_STAGED_STORE && 4 * STG_WAVE <= C::LDS_BYTES) {
        if ((p.epi == EPI_NONE || p.epi == EPI_RELU) && !p.x3) {   // wave-uniform
            typedef __attribute__((ext_vector_type(2))) T pair_t;
            typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
            typedef __attribute__((address_space(3))) char lchar_t;
            __syncthreads();
            lchar_t *stg = (lchar_t *)smem + wave * STG_WAVE;
            const bool relu = p.epi == EPI_RELU;
            const int limit = p.cstore - pass * NF * 32;   // channels of this pass that exist in the output (multiple of 8)
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < NF; ++n)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        unsigned a[2], c[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float v0 = acc[m][n][4 * g + 2 * q], v1 = acc[m][n][4 * g + 2 * q + 1], u0 = acc[m][n][4 * (g + 1) + 2 * q], u1 = acc[m][n][4 * (g + 1) + 2 * q + 1];
                            if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                            const pair_t pa = {(T)v0, (T)v1}, pb = {(T)u0, (T)u1};
                            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
                            a[q] = sw[0]; c[q] = sw[1];
                        }
                        *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + (m * 32 + r) * STG_PX + (n * 32 + 8 * (g + h)) * 2) = u4_t{a[0], a[1], c[0], c[1]};
                    }
            constexpr int NCH = NF * 4;   // 16-byte chunks per staged pixel
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                const int y = ty * C::TH + wave * MF + m;
                if (y >= p.Hout) continue;
                char *orow = reinterpret_cast<char *>(p.out) + ((((size_t)b * p.Hout + y) * p.Wout + (size_t)tx * 32) * p.out_ps + p.out_coff + pass * NF * 32) * sizeof(T);
#pragma unroll
                for (int i = 0; i < NCH / 2; ++i) {
                    const int q = i * 64 + lane, px = q / NCH, ch = q - px * NCH;
                    if (tx * 32 + px < p.Wout && ch * 8 < limit)
                        *reinterpret_cast<u4_t *>(orow + (size_t)px * p.out_ps * sizeof(T) + ch * 16) =
                            *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + (m * 32 + px) * STG_PX + ch * 16);
                }
            }
            return;
        }
    }
    conv_epilogue<T, MF, NF>(acc, p, b, pass, ty * C::TH + wave * MF, tx * 32 + r, h);
}

template <typename T, int CK, int NF, int S> static int launch_conv_inst(const ConvParams &p, hipStream_t s)
{
    using C = ConvCfg<T, CK, NF, S>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_kernel<T, CK, NF, S>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    dim3 grid((p.Wout + 31) / 32, (p.Hout + C::TH - 1) / C::TH, p.B * p.npass);
    conv3x3_kernel<T, CK, NF, S><<<grid, 256, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Persistent, weights-resident variant (bf16, stride 1, all input channels in one chunk, one pass):
// the tile-per-workgroup kernel above DMA-ingests the input tile AND all nine taps' weights for every
// 256-pixel tile (64->64: 49 + 72 KiB); at B=8 x 720p that is 3.5 GB per launch = 5.3 TB/s of
// LDS-DMA fill, ~83 % of the chip-wide LDS-DMA ceiling (MI355X_MICROARCH.md: 6.4 TB/s) - the kernel
// was ingest-bound, not MFMA-bound.  Here a workgroup stays resident (grid = CUs x workgroups per CU),
// DMAs the nine taps' weights ONCE, and walks tiles of WAVES*2 rows x 32 columns: per tile only the
// input tile is ingested, there is no per-tap weight DMA and no per-tap barrier (one barrier after the
// tile's DMA, one before the next tile overwrites it), so a wave issues its 9*KG*2*NF MFMAs back to back.
// ------------------------------------------------------------------------------------------
template <typename T, int CK, int NF, int WAVES> struct ConvPersistCfg {
    using D = DT<T>;
    static constexpr int MF = 2, TH = WAVES * MF, TW = 32, IH = TH + 2, IW = TW + 2;
    static constexpr int PSTR = LdsPix<T, CK>::BYTES;
    static constexpr int PIECES = CK * (int)sizeof(T) / 16;
    static constexpr int KG = CK / D::CHKG;
    static constexpr int WTAP = KG * NF * 1024, WINST = 9 * KG * NF;
    static constexpr int SP = PSTR / 16, NSLOT = IH * IW * SP, NINST = (NSLOT + 63) / 64;
    static constexpr int LDS_W = 9 * WTAP, LDS_IN = NINST * 1024, LDS_BYTES = LDS_W + LDS_IN;
    static constexpr int WG_PER_CU = (160 * 1024 / LDS_BYTES) > 4 ? 4 : (160 * 1024 / LDS_BYTES);
    static_assert(LDS_BYTES <= 160 * 1024, "resident weights + tile do not fit the 160 KiB LDS");
};

template <typename T, int CK, int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_persist_kernel(const ConvParams p)
{
    using C = ConvPersistCfg<T, CK, NF, WAVES>;
    using vec = typename DT<T>::vec;
    constexpr int MF = C::MF, IW = C::IW, PSTR = C::PSTR;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_w = smem;
    char *lds_in = smem + C::LDS_W;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const char *zeros = (const char *)p.zeros;
    const int npieces = p.in_pieces > 0 ? p.in_pieces : C::PIECES;

    // all nine taps' packed weights, once per workgroup
#pragma unroll 1
    for (int j = wave; j < C::WINST; j += WAVES)
        __builtin_amdgcn_global_load_lds((gptr_t *)((const char *)p.w + j * 1024 + lane * 16), (lptr_t *)(lds_w + j * 1024), 16, 0, 0);

    const int ntx = (p.Wout + 31) / 32, nty = (p.Hout + C::TH - 1) / C::TH;
    const int ntiles = ntx * nty * p.B;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (ntx * nty), trem = tile - b * (ntx * nty);
        const int ty = trem / ntx, tx = trem - ty * ntx;
        const int iy0 = ty * C::TH - 1, ix0 = tx * 32 - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        if (tile != (int)blockIdx.x) __syncthreads();  // every wave has finished reading the previous tile
#pragma unroll
        for (int i = 0; i < (C::NINST + WAVES - 1) / WAVES; ++i) {
            const int j = i * WAVES + wave;
            if (j < C::NINST) {
                const int sl = j * 64 + lane;
                const int pix = sl / C::SP, pc = sl - pix * C::SP;
                const int ly = pix / IW, lx = pix - ly * IW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                const bool ok = sl < C::NSLOT && pc < npieces && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
                const char *src = ok ? gin + ((size_t)gy * p.Win + gx) * p.in_ps * sizeof(T) + pc * 16 : zeros;
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(lds_in + j * 1024), 16, 0, 0);
            }
        }
        f32x16 acc[MF][NF];
        conv_init_acc<MF, NF>(acc, p, b, 0, ty * C::TH + wave * MF, tx * 32 + r, h);
        __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier: tile (and, first time, weights) landed

#if EMAVFI_CONV_PREFETCH
        // Every CU walks its tile list in step with the others, so the tiles' input DMAs hit HBM as chip-wide bursts and are
        // waited for with the matrix pipe idle.  Touch the NEXT tile's input lines now (one byte per 128-byte line, result
        // unused): HBM delivers them into this XCD's L2 under the MFMAs below, and the next DMA is an L2 hit.
        constexpr int LPR = (IW * C::PIECES * 16 + 127) / 128 + 1;   // lines a tile row can touch
        constexpr int NPF = (C::IH * LPR + 64 * WAVES - 1) / (64 * WAVES);
        unsigned pfr[NPF] = {};
        {
            const int nt = tile + (int)gridDim.x;
            if (nt < ntiles) {
                const int nb = nt / (ntx * nty), nrem = nt - nb * (ntx * nty);
                const int nty_ = nrem / ntx, ntx_ = nrem - nty_ * ntx;
                const int ny0 = nty_ * C::TH - 1, nx0 = ntx_ * 32 - 1;
                const size_t pixb = (size_t)p.in_ps * sizeof(T);
                const char *gn = (const char *)p.in + (size_t)nb * p.Hin * p.Win * pixb;
                const int xa = nx0 < 0 ? 0 : nx0, xb = nx0 + IW > p.Win ? p.Win : nx0 + IW;
#pragma unroll
                for (int sl = 0; sl < NPF; ++sl) {
                    const int s_ = sl * 64 * WAVES + tid, row = s_ / LPR, l = s_ - row * LPR;
                    const int gy = ny0 + row;
                    const char *a0 = gn + ((size_t)gy * p.Win + xa) * pixb, *a1 = gn + ((size_t)gy * p.Win + xb) * pixb;
                    const char *ln = (const char *)((size_t)a0 & ~(size_t)127) + (size_t)l * 128;
                    if (row < C::IH && gy >= 0 && gy < p.Hin && ln < a1) {
                        const char *t_ = ln < a0 ? a0 : ln;
                        // the destination register stays reserved (it is an operand of the s_waitcnt statement behind the
                        // epilogue), so the in-flight load cannot land in a register the compiler has given to something else
                        asm volatile("global_load_ubyte %0, %1, off" : "=v"(pfr[sl]) : "v"(t_) : "memory");
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the touches in front of the MFMA loop
#endif

#if EMAVFI_CONV_PIPELINE
        // Software pipeline over (tap, k-group pair) steps: the operands of step s+1 (MF pixel pieces + NF weight fragments
        // per k-group) are read from LDS BEFORE the MFMAs of step s issue, and sched_barrier pins that order (left alone,
        // hipcc places each read group directly in front of its MFMAs, exposing an LDS round trip per group).
        {
            constexpr int KGS = (C::KG % 2 == 0) ? 2 : 1;         // k-groups per step
            constexpr int SPT = C::KG / KGS, NSTEP = 9 * SPT;     // steps per tap, steps per tile
            static_assert(C::KG % KGS == 0, "k-groups per step must divide KG");
            vec xq[2][MF][KGS], wq[2][KGS][NF];
            auto load_step = [&](int s, vec (&xd)[MF][KGS], vec (&wd)[KGS][NF]) {
                const int tap = s / SPT, kg0 = (s - tap * SPT) * KGS;
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    const char *xb = lds_in + (((wave * MF + m) + dy) * IW + r + dx) * PSTR + h * 16;
#pragma unroll
                    for (int k = 0; k < KGS; ++k) xd[m][k] = *reinterpret_cast<const vec *>(xb + (kg0 + k) * 32);
                }
                const char *wb = lds_w + tap * C::WTAP + lane * 16;
#pragma unroll
                for (int k = 0; k < KGS; ++k)
#pragma unroll
                    for (int n = 0; n < NF; ++n) wd[k][n] = *reinterpret_cast<const vec *>(wb + ((kg0 + k) * NF + n) * 1024);
            };
            load_step(0, xq[0], wq[0]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + 1 < NSTEP) load_step(s + 1, xq[(s + 1) & 1], wq[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < KGS; ++k)
#pragma unroll
                    for (int n = 0; n < NF; ++n)
#pragma unroll
                        for (int m = 0; m < MF; ++m) mma_kg(acc[m][n], wq[s & 1][k][n], xq[s & 1][m][k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const char *xb[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) xb[m] = lds_in + (((wave * MF + m) + dy) * IW + r + dx) * PSTR + h * 16;
            const char *wb = lds_w + tap * C::WTAP + lane * 16;
#pragma unroll
            for (int kg = 0; kg < C::KG; ++kg) {
                vec xv[MF];
#pragma unroll
                for (int m = 0; m < MF; ++m) xv[m] = *reinterpret_cast<const vec *>(xb[m] + kg * 32);
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const vec wv = *reinterpret_cast<const vec *>(wb + (kg * NF + n) * 1024);
#pragma unroll
                    for (int m = 0; m < MF; ++m) mma_kg(acc[m][n], wv, xv[m]);
                }
            }
        }
#endif
        conv_epilogue<T, MF, NF>(acc, p, b, 0, ty * C::TH + wave * MF, tx * 32 + r, h);
#if EMAVFI_CONV_PREFETCH
        if constexpr (NPF == 1) asm volatile("s_waitcnt vmcnt(0)" ::"v"(pfr[0]) : "memory");
        else if constexpr (NPF == 2) asm volatile("s_waitcnt vmcnt(0)" ::"v"(pfr[0]), "v"(pfr[1]) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::"v"(pfr[0]), "v"(pfr[1]), "v"(pfr[NPF - 1]) : "memory");
        static_assert(NPF <= 3, "prefetch touches per thread");
#endif
    }
}

// ------------------------------------------------------------------------------------------
// The persistent kernel on v_mfma_f32_16x16x32 (16-bit types, CK = 64, two 32-channel fragments = 64 -> 64: five launches and a
// quarter of the step).  The board is power-managed and this MFMA shape does the same MACs with half
// the fp32 accumulator traffic: tools/microbench/mfma_shape_power.hip measures 11-15 % more FLOP/s in a bare loop and the same
// LDS-fed rate at 7 % less package power.  Per wave and tile still 2 rows x 32 pixels x 64 channels: 4 pixel blocks x 4 channel blocks of 16 x 16,
// 16 MFMAs per 32-channel step against 4 weight + 4 pixel fragments (the same 1 KiB of LDS per 16 K MACs as before).
// The tile is stored unpadded with the XOR slot swizzle of the ping-pong kernel (a 16-lane ds_read_b128 group here mixes two
// adjacent pieces of 8 + 8 consecutive pixels, which no padded stride serves without conflicts; the swizzle does for the
// taps with dx = 0 and leaves a two-way conflict on a few lanes otherwise).  Weights come packed for this shape
// (PackDesc::mfma16).  fp32 accumulation order inside a tap differs from the 32x32x16 kernels (32 channels per MFMA instead
// of 16), so results agree with them to fp32 rounding, not bit for bit.
// ------------------------------------------------------------------------------------------
// Slot permutation of the unpadded 128-byte tile pixels of the two 16x16x32 kernels: the 16-byte piece c of tile pixel q is stored
// in slot c ^ swz16(q).  A ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), i.e. a group
// mixes EIGHT pixels that read piece kb with EIGHT that read piece kb ^ 1 (the B-operand layout puts k-block kb = lane >> 4 on
// 16 lanes).  Round 2's permutation c ^ ((q >> 1) & 7) touches bit 0 of c, so for half of the tile alignments (odd q >> 1 at the
// group's first pixel: every tap with dx = 1 and every odd tile row) two lane pairs of a group met in one bank quad: exactly the
// 25.0 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE the round-2 counters showed.  This one only permutes bits
// 1-2 of the piece index: bit 0 keeps the two halves of a group apart, and the eight pixels of a half have eight different
// (q & 1, (q >> 1) & 3) = (bank-row half, permutation) pairs for EVERY alignment.  +16 pixels keeps the permutation; the second
// 32-channel step still flips bit 2 of the piece = byte 64 of the address.
#ifndef EMAVFI_CONV_SWZ_NEW
#define EMAVFI_CONV_SWZ_NEW 1
#endif
#ifndef EMAVFI_CONV_FASTDMA
#define EMAVFI_CONV_FASTDMA 1
#endif
#ifndef EMAVFI_CONV_PKRELU
#define EMAVFI_CONV_PKRELU 0   // ReLU as v_pk_max_i16 on the rounded pairs: half the instructions, and measured SLOWER (64 -> 64: 570 vs 532 us,
#endif                         // same box): VOP3P instructions beside the other group's MFMAs cost more than two plain v_max_f32

__device__ __forceinline__ int swz16(int q) { return EMAVFI_CONV_SWZ_NEW ? ((q >> 1) & 3) << 1 : (q >> 1) & 7; }

// NF = 32-channel fragments the layer is packed for (2 NF blocks of 16 in the packed weights); NB = blocks of 16 output channels
// the kernel computes: 2 NF, or 1 for the planar heads (flow: 2 channels), which then do half the MFMAs of a 32-wide fragment.
template <typename T, int CK, int NF, int NB> struct ConvP16Cfg {
    static constexpr int WAVES = 8, MF = 2, TH = WAVES * MF, TW = 32, IH = TH + 2, IW = TW + 2;
    static constexpr int PIECES = CK * (int)sizeof(T) / 16, PSTR = PIECES * 16, SP = PIECES;
    static constexpr int K32 = CK / 32, NBP = NF * 2, PB = MF * 2;
    static constexpr int WTAP = K32 * NBP * 1024, WINST = 9 * K32 * NBP;
    static constexpr int NSLOT = IH * IW * SP, NINST = (NSLOT + 63) / 64;
    static constexpr int LDS_W = 9 * WTAP, LDS_IN = NINST * 1024, LDS_BYTES = LDS_W + LDS_IN;
    static_assert(sizeof(T) == 2 && PIECES == 8, "16-bit storage, 64 input channels (8 pieces: the XOR swizzle's domain)");
    static_assert(NB == NBP || NB == 1, "all packed blocks, or the first one only");
    static_assert(LDS_BYTES <= 160 * 1024, "resident weights + tile do not fit the 160 KiB LDS");
};

template <typename T, int CK, int NF, int NB>
__global__ __launch_bounds__(512) void conv3x3_persist16_kernel(const ConvParams p)
{
    using C = ConvP16Cfg<T, CK, NF, NB>;
    using vec = typename DT<T>::vec;
    constexpr int MF = C::MF, IW = C::IW, PSTR = C::PSTR, PB = C::PB, NBP = C::NBP;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_w = smem;
    char *lds_in = smem + C::LDS_W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kb = lane >> 4;
    const char *zeros = (const char *)p.zeros;
    const int npieces = p.in_pieces > 0 ? p.in_pieces : C::PIECES;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);

#pragma unroll 1
    for (int i = wave; i < C::WINST; i += C::WAVES)
        __builtin_amdgcn_global_load_lds((gptr_t *)((const char *)p.w + i * 1024 + lane * 16), (lptr_t *)(lds_w + i * 1024), 16, 0, 0);

    const int ntx = (p.Wout + 31) / 32, nty = (p.Hout + C::TH - 1) / C::TH;
    const int ntiles = ntx * nty * p.B;
    const int coutpad = NF * 32;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (ntx * nty), trem = tile - b * (ntx * nty);
        const int ty = trem / ntx, tx = trem - ty * ntx;
        const int iy0 = ty * C::TH - 1, ix0 = tx * 32 - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        if (tile != (int)blockIdx.x) __syncthreads();  // every wave has finished reading the previous tile
#pragma unroll
        for (int i = 0; i < (C::NINST + C::WAVES - 1) / C::WAVES; ++i) {
            const int jn = i * C::WAVES + wave;
            if (jn < C::NINST) {
                const int sl = jn * 64 + lane;
                const int pix = sl >> 3, pc = (sl & 7) ^ swz16(pix);   // slot -> the piece stored there
                const int ly = pix / IW, lx = pix - ly * IW;
                const int gy = iy0 + ly, gx = ix0 + lx;
                const char *src = conv_dma_src(gin, zeros, gy, gx, pc, p.Hin, p.Win, pixbytes, sl < C::NSLOT && pc < npieces);
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(lds_in + jn * 1024), 16, 0, 0);
            }
        }
        // accumulators: block pb = 2 * m + c is row m of the wave, columns 16 c .. 16 c + 15; the lane's pixel is column j of it
        f32x4 acc[PB][NB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const float *bp = p.bias;
            if (p.bias_mode == 1) {   // border-class bias table (motion_estimation.0, see conv_init_acc)
                const int y = ty * C::TH + wave * MF + (pb >> 1), x = tx * 32 + (pb & 1) * 16 + j;
                const int ym = (y >= 1 ? 1 : 0) | (y <= p.Hout - 2 ? 2 : 0);
                const int xm = (x >= 1 ? 1 : 0) | (x <= p.Wout - 2 ? 2 : 0);
                bp += ((size_t)b * 16 + ym * 4 + xm) * coutpad;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pb][nb][e] = bp[nb * 16 + kb * 4 + e];
        }
        __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier: tile (and, first time, weights) landed

        {
            constexpr int NSTEP = 9 * C::K32;
            vec xq[2][PB], wq[2][NB];
            auto load_step = [&](int s, vec (&xd)[PB], vec (&wd)[NB]) {
                const int tap = s / C::K32, k32 = s - tap * C::K32;
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int m = 0; m < MF; ++m) {   // +16 pixels keeps the slot permutation, k32 flips byte 64
                    const int q = ((wave * MF + m) + dy) * IW + j + dx;
                    const int off = (q * PSTR + ((kb ^ swz16(q)) << 4)) ^ (k32 * 64);
                    xd[2 * m] = *reinterpret_cast<const vec *>(lds_in + off);
                    xd[2 * m + 1] = *reinterpret_cast<const vec *>(lds_in + off + 2048);
                }
                const char *wb = lds_w + tap * C::WTAP + k32 * NBP * 1024 + lane * 16;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wd[nb] = *reinterpret_cast<const vec *>(wb + nb * 1024);
            };
            load_step(0, xq[0], wq[0]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + 1 < NSTEP) load_step(s + 1, xq[(s + 1) & 1], wq[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) mma_k32(acc[pb][nb], wq[s & 1][nb], xq[s & 1][pb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if constexpr (NB == 1) {
            // planar heads (flow / frame): <= 4 real channels = the four accumulator registers of the lanes with kb == 0, written
            // as NCHW fp32 planes exactly as conv_epilogue does
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int y = ty * C::TH + wave * MF + (pb >> 1), x = tx * 32 + (pb & 1) * 16 + j;
                if (kb != 0 || y >= p.Hout || x >= p.Wout) continue;
                const size_t plane = (size_t)p.Hout * p.Wout;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < p.nplanes) {
                        float v = acc[pb][0][c];
                        if (p.round16) {
                            v = (float)(half_t)v;
                            if (p.epi == EPI_PLANAR_TANH01) v = (float)(half_t)((float)(half_t)tanhf(v) + 1.0f) / 2.0f;
                        } else if (p.epi == EPI_PLANAR_TANH01) v = (tanhf(v) + 1.0f) / 2.0f;
                        p.out_planar[((size_t)b * p.nplanes + c) * plane + (size_t)y * p.Wout + x] = v;
                    }
            }
        } else {
            // channels-last T, bias already in, optional ReLU: channel blocks nb = 2 t and 2 t + 1 are exchanged between the lane
            // rows with v_permlane16_swap so that every lane holds 8 consecutive channels of its pixel = one 16-byte store:
            // row kb stores channels (2 t + (kb & 1)) * 16 + (kb >> 1) * 8 .. + 7
            typedef __attribute__((ext_vector_type(2))) T pair_t;
            const bool relu = p.epi == EPI_RELU;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int y = ty * C::TH + wave * MF + (pb >> 1), x = tx * 32 + (pb & 1) * 16 + j;
                const bool inside = y < p.Hout && x < p.Wout;
                T *ob = reinterpret_cast<T *>(p.out) + (((size_t)b * p.Hout + (inside ? y : 0)) * p.Wout + (inside ? x : 0)) * p.out_ps + p.out_coff;
#pragma unroll
                for (int t = 0; t < NB / 2; ++t) {
                    unsigned a[2], c2[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v0 = acc[pb][2 * t][2 * i], v1 = acc[pb][2 * t][2 * i + 1], u0 = acc[pb][2 * t + 1][2 * i], u1 = acc[pb][2 * t + 1][2 * i + 1];
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        const pair_t pa = {(T)v0, (T)v1}, pc2 = {(T)u0, (T)u1};
                        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pc2), false, false);
                        a[i] = sw[0];
                        c2[i] = sw[1];
                    }
                    const int c0 = (2 * t + (kb & 1)) * 16 + (kb >> 1) * 8;
                    if (inside && c0 < p.cstore) *reinterpret_cast<uint4 *>(ob + c0) = make_uint4(a[0], a[1], c2[0], c2[1]);
                }
            }
        }
    }
}

template <typename T, int CK, int NF, int NB> static int launch_conv_persist16(const ConvParams &p, hipStream_t s)
{
    using C = ConvP16Cfg<T, CK, NF, NB>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_persist16_kernel<T, CK, NF, NB>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int ntiles = ((p.Wout + 31) / 32) * ((p.Hout + C::TH - 1) / C::TH) * p.B;
    conv3x3_persist16_kernel<T, CK, NF, NB><<<ntiles < ncu ? ntiles : ncu, 512, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}

#include "conv_light.inl"
#include "conv_ring.inl"
#include "conv_ring2.inl"
#include "conv_ring_tail.inl"

// weights packed for the 16x16x32 shape (ConvParams::mfma16): 64 -> 64 (four blocks), 64 -> 32 (two), 64 / 32 -> planes (one)
template <typename T> static int launch_conv_mfma16(const ConvParams &p, hipStream_t s)
{
    if (p.head_w) return launch_conv_ringtail<T>(p, s);   // reconstruction.1 + .2 in one launch (conv_ring_tail.inl)
    const bool planar = p.epi == EPI_PLANAR || p.epi == EPI_PLANAR_TANH01;
    if (p.ck == 32 && p.stride == 1 && p.nchunk == 1 && p.npass == 1 && p.nf == 1 && planar && p.nplanes <= 4) return launch_conv_light<T, 32>(p, s);
    if (p.ck != 64 || p.stride != 1 || p.nchunk != 1 || p.npass != 1) return -2;
    if (!planar && p.epi != EPI_NONE && p.epi != EPI_RELU) return -2;
    if (p.nf == 2 && !planar) return launch_conv_persist16<T, 64, 2, 4>(p, s);
    if (p.nf == 1 && !planar) return launch_conv_persist16<T, 64, 1, 2>(p, s);
    if (p.nf == 1 && planar && p.nplanes <= 4) {
        if (!(emavfi_switches() & SW_NO_CONV_LIGHT)) return launch_conv_light<T, 64>(p, s);   // EMAVFI_CONV_LIGHT=0: the lock-step persistent kernel (A/B)
        return launch_conv_persist16<T, 64, 1, 1>(p, s);
    }
    return -2;
}

template <typename T, int CK, int NF, int WAVES> static int launch_conv_persist(const ConvParams &p, hipStream_t s)
{
    using C = ConvPersistCfg<T, CK, NF, WAVES>;
    static PerDeviceOnce lds_once;
    if (const hipError_t e_ = set_lds_limit(lds_once, reinterpret_cast<const void *>(&conv3x3_persist_kernel<T, CK, NF, WAVES>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    static std::once_flag once;   // (the occupancy of the first device is taken for all: one process drives one GPU model)
    static int wg_per_cu = 0;  // resident workgroups per CU: registers and LDS both limit it
    static hipError_t init_err = hipSuccess;
    std::call_once(once, [] {
        int n = 0;
        init_err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_persist_kernel<T, CK, NF, WAVES>, 64 * WAVES, C::LDS_BYTES);
        if (init_err != hipSuccess) return;
        // a persistent grid must not exceed what is resident: queued workgroups would only start
        // when others have finished their whole tile list
        wg_per_cu = n < 1 ? 1 : (n > C::WG_PER_CU ? C::WG_PER_CU : n);
    });
    if (init_err != hipSuccess) return (int)init_err;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int ntiles = ((p.Wout + 31) / 32) * ((p.Hout + C::TH - 1) / C::TH) * p.B;
    const int resident = ncu * wg_per_cu;
    conv3x3_persist_kernel<T, CK, NF, WAVES><<<ntiles < resident ? ntiles : resident, 64 * WAVES, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// context_encoding.0 (64 -> 128, stride 2: ema_vfi.py:80): weights stationary in registers, input rows through an LDS ring.
//
// The tile-per-workgroup kernel takes stride-2 layers in 32-channel chunks (a 64-channel 4 x 32 tile + weights would leave one
// workgroup per CU), i.e. it DMAs 64 bytes of every 144-byte fusion pixel per chunk - and each chunk fetches every 128-byte line
// (PMC: 2.75 GB for 1.42 GB algorithmic; an XCD's L2 does not hold the lines for the ~9 us between the two chunk passes), and it
// streams 144 KiB of weights through LDS per tile.  Here a workgroup walks DOWN a strip of 32 output columns:
//   * wave w owns output fragment w (channels 32 w .. 32 w + 31) and keeps that fragment's 64 x 9 x 32 weights in 144 VGPRs for
//     the whole walk: no weights in LDS, no per-tap barrier, one ds_read_b128 per MFMA;
//   * input rows live in a ring of five (output row y reads input rows 2y-1, 2y, 2y+1; rows 2y+2, 2y+3 are in flight for y+1):
//     every input row is DMA'd once per strip - no vertical halo at all, 65 / 64 horizontally;
//   * a row's even and odd pixels are stored apart (pixel p -> slot (p >> 1) + 33 (p & 1)): lane r reads pixel 2r + dx, i.e.
//     slot r + const at 144-byte pitch = 36 r dwords, distinct 4-bank groups for any 16 lanes a ds_read_b128 serves together
//     (the interleaved order reads at 72 r dwords: two-way conflicts);
//   * outputs go through a double-buffered LDS row (32 pixels x 256 bytes) so that a store instruction writes whole lines.
// One barrier per output row.  A strip is cut into `nseg` vertical segments so that ~2 workgroups per CU exist.
// ------------------------------------------------------------------------------------------
template <typename T> struct ConvS2RCfg {
    static constexpr int CK = 64, NFRAG = 4, PSTR = 144, IW = 65, SP = 9, ROWSLOT = IW * SP, ROWINST = (ROWSLOT + 63) / 64, ROWB = ROWINST * 1024;
    static constexpr int RING = 5, STG_PX = NFRAG * 64 + 16, STG = 32 * STG_PX, LDS_BYTES = RING * ROWB + 2 * STG;
    static_assert(sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024, "16-bit types; two workgroups per CU");
};

template <typename T, bool ALT = false>   // ALT: ConvParams::out_alt
__global__ __launch_bounds__(256, 2) void conv3x3_s2ring_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = ConvS2RCfg<T>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ntx = (p.Wout + 31) / 32, nstrip = ntx * p.B;
    const int strip = (int)blockIdx.x % nstrip, seg = (int)blockIdx.x / nstrip;
    const int b = strip / ntx, tx = strip - b * ntx;
    const int ys = seg * seg_rows, ye = min(ys + seg_rows, p.Hout);
    if (ys >= ye) return;   // workgroup-uniform
    const char *zeros = (const char *)p.zeros;
    const int npieces = p.in_pieces > 0 ? p.in_pieces : 8;
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
    const int ix0 = tx * 64 - 1;

    // ---- this wave's fragment of the weights: 9 taps x 4 k-groups x (32 channels x 16 k), packed [tap][kg][fragment][lane][8]
    vec wf[9][4];
    {
        const char *wb = (const char *)p.w + wave * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wf[t][kg] = *reinterpret_cast<const vec *>(wb + (t * 4 + kg) * (C::NFRAG * 1024));
    }
    // ---- DMA of one input row into one ring slot: ROWINST instructions, instruction jn covers 16-byte slots [64 jn, 64 jn + 64)
    auto dma = [&](int gy, int slot, int jn) {
        const int q = jn * 64 + lane;
        const int ps = q / C::SP, pc = q - ps * C::SP;
        const int px = ps < 33 ? 2 * ps : 2 * (ps - 33) + 1;   // even pixels first, then the odd ones
        const char *src = conv_dma_src(gin, zeros, gy, ix0 + px, pc, p.Hin, p.Win, pixbytes, q < C::ROWSLOT && pc < npieces && pc < 8);
        __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(ring + slot * C::ROWB + jn * 1024), 16, 0, 0);
    };
    // rows 2 ys - 1 .. 2 ys + 1 -> slots (row + 1) % 5
    int s0 = (2 * ys) % C::RING;   // slot of input row 2y - 1
    for (int u = wave; u < 3 * C::ROWINST; u += 4) {
        const int k = u / C::ROWINST, jn = u - k * C::ROWINST;
        int sl = s0 + k; sl = sl >= C::RING ? sl - C::RING : sl;
        dma(2 * ys - 1 + k, sl, jn);
    }
    float bias[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias[i] = p.bias[wave * 32 + acc_channel(i, h)];
    const bool relu = p.epi == EPI_RELU;
    char *obase = reinterpret_cast<char *>(p.out) + (((size_t)b * p.Hout * p.Wout + (size_t)tx * 32) * p.out_ps + p.out_coff) * sizeof(T);
    auto store_row = [&](int y) {
        lchar_t *stg = (lchar_t *)smem + C::RING * C::ROWB + (y & 1) * C::STG;
        char *orow = obase + (size_t)y * p.Wout * p.out_ps * sizeof(T);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 256 + tid, px = q >> 4, ch = q & 15;
            if (tx * 32 + px < p.Wout && ch * 8 < p.cstore)
                *reinterpret_cast<u4_t *>(orow + (size_t)px * p.out_ps * sizeof(T) + ch * 16) =
                    *reinterpret_cast<const __attribute__((address_space(3))) u4_t *>(stg + px * C::STG_PX + ch * 16);
        }
    };
#pragma unroll 1
    for (int y = ys; y < ye; ++y) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the rows for y (and its stores of row y - 2)
        __syncthreads();
        if (y > ys) store_row(y - 1);
        if (y + 1 < ye) {   // rows 2y + 2, 2y + 3 -> slots s0 + 3, s0 + 4
#pragma unroll
            for (int i = 0; i < (2 * C::ROWINST + 3) / 4; ++i) {
                const int u = i * 4 + wave;
                if (u < 2 * C::ROWINST) {
                    const int k = u / C::ROWINST, jn = u - k * C::ROWINST;
                    int sl = s0 + 3 + k; sl = sl >= C::RING ? sl - C::RING : sl;
                    dma(2 * y + 2 + k, sl, jn);
                }
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = bias[i]; acc[1][i] = 0.0f; }
        {
            // operands three k-groups ahead of their MFMAs (fenced: the scheduler otherwise sinks every read to just above its MFMA)
            constexpr int AH = 3;
            const char *xb[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                int sl = s0 + dy; sl = sl >= C::RING ? sl - C::RING : sl;
                xb[dy] = ring + sl * C::ROWB + r * C::PSTR + h * 16;
            }
            // pixel 2r + dx: even -> slot r + dx / 2, odd -> slot 33 + r
            auto xptr = [&](int n) { const int dx = (n / 4) % 3; return xb[n / 12] + (dx == 1 ? 33 : (dx >> 1)) * C::PSTR + (n & 3) * 32; };
            vec xq[AH + 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < AH; ++s) xq[s] = *reinterpret_cast<const vec *>(xptr(s));
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                if (s + AH < 36) xq[(s + AH) % (AH + 1)] = *reinterpret_cast<const vec *>(xptr(s + AH));
                mma_kg(acc[s & 1], wf[s >> 2][s & 3], xq[s % (AH + 1)]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- bias is in; optional ReLU; this wave's 32 channels of the row's 32 pixels into the row's staging buffer
        {
            typedef __attribute__((ext_vector_type(2))) T pair_t;
            lchar_t *stg = (lchar_t *)smem + C::RING * C::ROWB + (y & 1) * C::STG + r * C::STG_PX + wave * 64;
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                unsigned a[2], c[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v0 = acc[0][4 * g + 2 * q] + acc[1][4 * g + 2 * q], v1 = acc[0][4 * g + 2 * q + 1] + acc[1][4 * g + 2 * q + 1];
                    float u0 = acc[0][4 * (g + 1) + 2 * q] + acc[1][4 * (g + 1) + 2 * q], u1 = acc[0][4 * (g + 1) + 2 * q + 1] + acc[1][4 * (g + 1) + 2 * q + 1];
                    if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                    const auto sw = __builtin_amdgcn_permlane32_swap(pack16x2<T>(v0, v1, ALT), pack16x2<T>(u0, u1, ALT), false, false);
                    a[q] = sw[0]; c[q] = sw[1];
                }
                *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + 16 * (g + h)) = u4_t{a[0], a[1], c[0], c[1]};
            }
        }
        s0 += 2; s0 = s0 >= C::RING ? s0 - C::RING : s0;
    }
    __syncthreads();
    store_row(ye - 1);
}

template <typename T, bool ALT> static int launch_conv_s2ring_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvS2RCfg<T>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_s2ring_kernel<T, ALT>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + 31) / 32) * p.B;
    int nseg = (2 * ncu) / nstrip;   // two workgroups per CU in one round
    nseg = nseg < 1 ? 1 : (nseg > p.Hout ? p.Hout : nseg);
    const int seg_rows = (p.Hout + nseg - 1) / nseg;
    nseg = (p.Hout + seg_rows - 1) / seg_rows;
    conv3x3_s2ring_kernel<T, ALT><<<nstrip * nseg, 256, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}
template <typename T> static int launch_conv_s2ring(const ConvParams &p, hipStream_t s)
{
    if (p.ck != 64 || p.nf != 4 || p.stride != 2 || p.nchunk != 1 || p.npass != 1 || p.bias_mode != 0 || (p.epi != EPI_NONE && p.epi != EPI_RELU)) return -2;
    return p.out_alt ? launch_conv_s2ring_t<T, true>(p, s) : launch_conv_s2ring_t<T, false>(p, s);
}

// (CK, NF, stride) instantiations of the tile-per-workgroup kernel (keep in sync with kConvInst in emavfi_api.hip)
#define EMAVFI_CONV_INSTANCES(X) \
    X(16, 1, 1) X(16, 2, 1) X(32, 1, 1) X(48, 1, 1) X(64, 1, 1) X(64, 2, 1) X(64, 4, 1) X(80, 1, 1) X(80, 2, 1) \
    X(16, 1, 2) X(32, 2, 2) X(32, 4, 2)

template <typename T> static int launch_conv_any(const ConvParams &p, hipStream_t s)
{
#define X(CK_, NF_, ST_) \
    if (p.ck == CK_ && p.nf == NF_ && p.stride == ST_) return launch_conv_inst<T, CK_, NF_, ST_>(p, s);
    EMAVFI_CONV_INSTANCES(X)
    if constexpr (sizeof(T) == 2) {  // 16-bit types only: eight output fragments in one pass (EMAVFI_CONV_WREG=0: context_encoding.1)
        X(32, 8, 2)
    }
    if constexpr (sizeof(T) == 4) {  // fp32 only (k-groups of 8 channels): 65..72 input channels as 9 k-groups instead of 10 (round 6: the
        X(72, 1, 1) X(72, 2, 1)      // fp32 layers are matrix-pipe-bound at the clock the board holds; offset_conv 67 -> 27, reconstruction.0 67 -> 64)
    }
#undef X
    return -2;  // no instantiation
}

#include "conv_wreg.inl"

// 16-bit launcher (bf16 / f16): full-resolution single-chunk layers go to the persistent, weights-resident kernel
// (CK, NF, waves): 64->64, 64->32 / 64->2, 67->27 of the mid_channels = 64 model.
template <typename T> static int launch_conv16(const ConvParams &p, hipStream_t s, bool no_persistent)
{
    if (p.mfma16) return launch_conv_mfma16<T>(p, s);
    if (p.ring == 1) return launch_conv_s2ring<T>(p, s);
    if (p.ring == 4) return launch_conv_wreg<T>(p, s);
    if (p.ring >= 2) return p.w2 ? launch_conv_ring2<T>(p, s) : launch_conv_ring<T>(p, s);
    if (!no_persistent && p.stride == 1 && p.nchunk == 1 && p.npass == 1) {
        // measured at B=8 x 720p in bf16 (us per launch, tile-per-workgroup -> persistent): 64->64 670 -> 644,
        // 64->32 / 64->2 414 -> 370, 67->27 685 -> 557.  NOT used where it loses: 67->64 with 4 waves
        // (771 -> 915: one 4-wave workgroup per CU cannot overlap its own phases), 6->64, 32->3 (no gain).
        if (p.ck == 64 && p.nf == 2) return launch_conv_persist<T, 64, 2, 8>(p, s);
        if (p.ck == 64 && p.nf == 1) return launch_conv_persist<T, 64, 1, 8>(p, s);
        if (p.ck == 80 && p.nf == 1) return launch_conv_persist<T, 80, 1, 8>(p, s);
    }
    return launch_conv_any<T>(p, s);
}
