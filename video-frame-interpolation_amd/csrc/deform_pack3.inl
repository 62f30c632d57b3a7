// ModulatedDeformConvPack (ema_vfi.py:23-60) at the reference width (67 -> 27 offsets/masks, 67 -> 67) as ONE launch,
// 16-bit storage types - round 3 rebuild of deform_pack.inl around what its counters and ISA showed:
// the kernel is bound by instruction issue at two waves per SIMD, so this version removes issued work.
//
//   * K = 64 + 3.  The three warped-frame channels (64..66) used to cost a fifth k-group in every tap (3 real of 16
//     input channels: 6 of 30 MFMAs, 6 of 15 weight-fragment loads, 8 corner reads and 16 blend instructions per tap).
//     Now every tap contracts exactly the 64 feature channels (4 k-groups), and the three tail channels of ALL NINE taps
//     are one im2col step at the end: each half-lane blends the tail of ITS OWN pixel per tap (4 eight-byte corner
//     reads, 8 packed FMAs), keeps the 9 x 3 values in registers, and three k-groups (K index = tap slot * 4 + channel,
//     27 real of 48) contract them after the tap loop.  The offset_conv does the same with the undeformed window.
//     Issued MFMAs per wave: 90 + 270 -> 78 + 234; blend instructions per tap 144 -> 136; LDS reads 40 -> 36.
//   * The third output fragment (channels 64..66: 3 real rows of 32) takes its A operand from a 4.5 KiB table in LDS
//     ([tap][k-group][row 0..2 | zero row][half][16 B], every lane of a zero row reads the same 16 bytes: a broadcast)
//     instead of a 1 KiB global fragment per k-group: weight-fragment traffic L2/L1 -> registers 15 -> 8 KiB per wave and tap
//     (64 B/clk per CU was a co-limit: 8 waves x 15 KiB per tap).
//   * Window DMA by rows: wave w owns pixels [7w, 7w+7) of every window row (wave 3: the last two), so a lane's
//     (pixel, piece) and its source pointer / row increment are computed ONCE and a DMA instruction costs one 64-bit add
//     and one select (was ~25 address instructions per DMA instruction, 6 200 cycles of issue per tile).
//   * The blend is issued corner-major (four independent chains), which hipcc's scheduler had turned back into four dependent
//     chains with a wait state behind every v_pk_fma_f16 (63 s_nop per tap); the sample-outside-the-window fix-up is a pass of its
//     own behind the tap loop (round 6: an arena in the then dead window, the same tap body), so the common path carries none of
//     its address arithmetic or EXEC regions.
//   * bf16 storage may hand f16 bit patterns between consecutive packs (DeformParams::in_f16 / out_f16): the window is f16 on
//     chip anyway (deform_pack.inl), so pack i+1 skips the in-LDS conversion pass and the intermediate fusion tensor keeps 11
//     significant bits instead of 8.
//
// Geometry, lane <-> pixel assignment, the f16 on-chip arithmetic and the fallback semantics are those of deform_pack.inl.
#pragma once
#include "deform_pack.inl"

// timing-only ablations (wrong results; the tap loop's, bits 0-3, 6, 7, leave the offsets alone: every sample stays in the window):
// bit 0 no weight-fragment loads in the tap loop, 1 undeformed (conflict-free) gathers, 2 no blend beyond its first four multiplies,
// 3 no 32x32x16 MFMAs in the tap loop, 4 no window DMA, 5 no offset_conv MFMAs, 6 no third-fragment MFMAs, 7 no corner reads
#ifndef EMAVFI_P3_ABL
#define EMAVFI_P3_ABL 0
#endif
// measurement build: request 100 KiB of LDS per workgroup = ONE workgroup (one wave per SIMD) per CU - what a wave's phases cost
// without a partner on its SIMD
#ifndef EMAVFI_P3_ONE_WG
#define EMAVFI_P3_ONE_WG 0
#endif
// A/B builds (round 6): what the census and the fix-up hand-shake cost a tile that has no sample outside its window.  NO_HANDSHAKE is
// only legal together with -DEMAVFI_DEFORM_ABL_NO_FALLBACK=1 (no wave ever waits): timing only, wrong results beyond the window.
#ifndef EMAVFI_P3_NO_CENSUS
#define EMAVFI_P3_NO_CENSUS 0
#endif
#ifndef EMAVFI_P3_NO_HANDSHAKE
#define EMAVFI_P3_NO_HANDSHAKE 0
#endif
struct Pack3 {
    static constexpr int R = 2, TROWS = 16, TCOLS = 16, WAVES = 4, THREADS = 256;
    static constexpr int TR = TROWS + 3 + 2 * R, TC = TCOLS + 3 + 2 * R;                 // 23 x 23 window pixels
    static constexpr int SP = 9, PSB = SP * 16, ROWB = TC * PSB, WIN_BYTES = TR * ROWB;  // 144 B pixels, 3312 B rows, 76 176 B
    static constexpr int SEG_PX = 7, SEG_BYTES = SEG_PX * PSB;                            // one DMA instruction = 7 pixels x 9 pieces
    static constexpr int LAST_PX = TC - 3 * SEG_PX;                                       // wave 3: the last 2 pixels of a row
    static constexpr int W3_OFF = WIN_BYTES, W3_TAP = 4 * 4 * 2 * 16, W3_BYTES = 9 * W3_TAP;  // third-fragment A operands
    // fix-up arena (round 6): once all four waves have left the tap loop the window is dead, and every wave owns a quarter of it as an
    // arena of 32 entries = {4 corners x 9 pieces | one pad slot} (an odd number of 16-byte slots: conflict-free like the window's pixels);
    // entry 31 is all zeros (what lanes without a sample in the round read against zero weights)
    // Layout: piece-major - row (2 i + hb) = 32 slots x 16 B holds piece i % 9 of corner (2 hb + i / 9) of every slot, i = 0..17 - because an
    // LDS-DMA instruction writes lane-linear: instruction i fetches row 2 i from lanes 0..31 and row 2 i + 1 from lanes 32..63, each lane
    // ONE slot's corner pair (2 hb, 2 hb + 1) for the whole round: a DMA instruction costs one 64-bit add.  Slot 31 is all zeros.
    static constexpr int NENT = 31, SLOTS = 32, AROW = SLOTS * 16, ADMA = 2 * SP;         // 512 B rows, 18 DMA instructions per round
    static constexpr int ARENA_BYTES = (WIN_BYTES / 4) & ~15;                             // 19 040 B per wave
    static constexpr int A_KG = 2 * 2 * AROW, A_H = 2 * AROW, A_C1 = SP * 2 * AROW, A_C2 = AROW;   // k-group / piece / corner strides
    static constexpr int SYNC_OFF = W3_OFF + W3_BYTES;                                    // u32: waves that have left the tap loop
    static constexpr int TAB_OFF = SYNC_OFF + 16, TAB_BYTES = 4 * 32 * 4;                 // per wave 32 corner descriptors
    static constexpr int LDS_BYTES = TAB_OFF + TAB_BYTES;                                 // 81 312 B: two workgroups per CU
    static_assert(4 * SP * AROW <= ARENA_BYTES && NENT < SLOTS, "arena");
    // packed weights (bytes): DCN = [tap][kg 4][nf 2][lane][16] | W3 table | tail [j 3][nf 3][lane][16]
    static constexpr int DCN_TAP = 4 * 2 * 1024, DCN_W3 = 9 * DCN_TAP, DCN_TAIL = DCN_W3 + W3_BYTES, DCN_BYTES = DCN_TAIL + 9 * 1024;
    // offset_conv = [tap][kg 4][lane][16] | tail [j 3][lane][16]
    static constexpr int OFF_TAP = 4 * 1024, OFF_TAIL = 9 * OFF_TAP, OFF_BYTES = OFF_TAIL + 3 * 1024;
    static_assert(SEG_PX * SP <= 64 && LAST_PX > 0 && LAST_PX <= SEG_PX, "row segments");
    static_assert(WIN_BYTES % 16 == 0 && 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static_assert(ROWB + PSB + 8 * 16 + 15 < 65536, "corner offsets must fit the ds_read immediate");
};

// window byte offset of plain tap t relative to tap 0 (taps past 8 re-read tap 8: finite data against zero weights)
__host__ __device__ constexpr int pack3_tap_off(int t) { return t < 9 ? ((t / 3) * Pack3::TC + (t % 3)) * Pack3::PSB : (2 * Pack3::TC + 2) * Pack3::PSB; }

// Pins a point of the hand-made schedule: an empty volatile asm that "rewrites" the four partial sums (instruction selection
// otherwise places plain arithmetic anywhere between its operands and its users, on either side of a scheduling fence - half of
// the steps came out with their blend sunk behind their MFMAs), then the fence for the machine scheduler.  The asm emits no
// instruction, so the hazard recogniser still sees the real producer of every MFMA operand.
#define PACK3_PIN(a)                                                                  \
    do {                                                                              \
        asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]));    \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

template <typename TS, bool FUSE_OFF>
__global__ __launch_bounds__(256, 2) void deform_pack3_kernel(const DeformParams p)
{
    using C = Pack3;
    constexpr int R = C::R;
    static_assert(sizeof(TS) == 2, "16-bit storage types only");
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_cchar_t *lds_r = (lds_cchar_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const unsigned lane16 = (unsigned)lane * 16u;
    DEFORM_STAMP(ts_begin);
#if EMAVFI_DEFORM_STAMPS
    unsigned long long sum_steps = 0, cnt_out = 0;
#endif

    // ---- tile of this workgroup (XCD-aware order: deform_pack.inl)
    const int ntx = (W + C::TCOLS - 1) / C::TCOLS, nty = (H + C::TROWS - 1) / C::TROWS, nt = ntx * nty;
    int tile_x, tile_y, b;
    {
        constexpr int SROWS = 4;
        const int nwg = gridDim.x, grp = blockIdx.x & 7, kk = blockIdx.x >> 3, qq = nwg >> 3, rr = nwg & 7;
        const int wg = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
        b = wg / nt;
        const int t = wg - b * nt, strip = t / (SROWS * ntx), tt = t - strip * SROWS * ntx;
        const int rows = min(SROWS, nty - strip * SROWS);
        tile_x = tt / rows;
        tile_y = strip * SROWS + (tt - tile_x * rows);
    }
    const unsigned ps_bytes = (unsigned)p.x_ps * 2u, tail_bytes = (unsigned)p.tail_ps * 2u;
    const int ty0 = tile_y * C::TROWS - 1 - R, tx0 = tile_x * C::TCOLS - 1 - R;
    const char *gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;
    const char *tplane = p.x_tail ? (const char *)p.x_tail + (size_t)b * H * W * tail_bytes : nullptr;
    const char *zeros = (const char *)p.zeros;
    const char *wbase_g = (const char *)p.w;       // wave-uniform bases: fragment loads are base + lane16 + immediate
    const char *owbase_g = (const char *)p.off_w;

    // ---- small loads first (L2-resident): the first two taps' offset_conv fragments
    f16x8 ow[FUSE_OFF ? 3 : 1][4];
    if constexpr (FUSE_OFF) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) ow[t][kg] = *reinterpret_cast<const f16x8 *>(owbase_g + (t * 4 + kg) * 1024 + lane16);
    }

    // ---- DMA the window, row by row: wave w fetches pixels [7w, 7w + 7) (wave 3: 2 pixels) x 9 pieces of every row.
    // A lane's pixel column and piece never change; out-of-image pixels read the zero page.
    const bool dact = wave < 3 ? lane < C::SEG_PX * C::SP : lane < C::LAST_PX * C::SP;
    {
        const int dp = lane / C::SP, dpc = lane - dp * C::SP;
        const int dgx = tx0 + wave * C::SEG_PX + dp;
        const bool dcol = (unsigned)dgx < (unsigned)W;
        const long long pix0 = (long long)ty0 * W + dgx;
        const bool from_tail = tplane != nullptr && dpc == C::SP - 1;   // channels 64..71 from the compact tail buffer
        const char *src = from_tail ? tplane + pix0 * (long long)tail_bytes : gplane + pix0 * (long long)ps_bytes + dpc * 16;
        const unsigned inc = (unsigned)W * (from_tail ? tail_bytes : ps_bytes);
        if (dact && !(EMAVFI_P3_ABL & 16)) {
#pragma unroll
            for (int ly = 0; ly < C::TR; ++ly) {
                const bool ok = dcol && (unsigned)(ty0 + ly) < (unsigned)H;
                const char *s = ok ? src : zeros;
                __builtin_amdgcn_global_load_lds((gptr_t *)s, (lptr_t *)(smem + ly * C::ROWB + wave * C::SEG_BYTES), 16, 0, 0);
                src += inc;
            }
        }
        // the third fragment's A operands (4 608 B): one DMA instruction per wave + half an instruction
        const char *w3g = wbase_g + C::DCN_W3;
        __builtin_amdgcn_global_load_lds((gptr_t *)(w3g + wave * 1024 + lane16), (lptr_t *)(smem + C::W3_OFF + wave * 1024), 16, 0, 0);
        if (wave == 0 && lane < 32)
            __builtin_amdgcn_global_load_lds((gptr_t *)(w3g + 4096 + lane16), (lptr_t *)(smem + C::W3_OFF + 4096), 16, 0, 0);
    }
    DEFORM_STAMP(ts_issued);
#if EMAVFI_DEFORM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    DEFORM_STAMP(ts_landed);
    if constexpr (std::is_same<TS, bf16_t>::value) {
        // bf16 -> f16 in place: every wave converts exactly the pieces its own DMA instructions fetched, so its own
        // vmcnt(0) is the only wait needed before it reads them back.  Skipped when the producer already wrote f16.
        if (!p.in_f16) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (dact) {
#pragma unroll
                for (int ly = 0; ly < C::TR; ++ly) {
                    lds_char_t *q = (lds_char_t *)smem + ly * C::ROWB + wave * C::SEG_BYTES + lane16;
                    const u32x4_t v = to_f16_piece<TS>(lds_read16(q));
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(q) = v;
                }
            }
        }
    }

    // ---- this lane's pixel in each of its wave's two fragments (2 rows x 16 columns; hardware ds_read_b128 lane groups
    // get one row of 16 consecutive pixels each: deform_pack.inl)
    const bool g2 = (r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28;
    const int fr_row = g2 ? 1 : 0;
    const int fr_col = g2 ? (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16)) : (r < 4 ? r : (r < 16 ? r - 8 : r - 12));
    const int px_x = tile_x * C::TCOLS + fr_col;
    int py_y[2], wrow[2];
    bool in_img[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        wrow[m] = (wave * 2 + m) * 2 + fr_row;
        py_y[m] = tile_y * C::TROWS + wrow[m];
        in_img[m] = py_y[m] < H && px_x < W;
    }
    const int my_y = h ? py_y[1] : py_y[0];
    const bool my_in = h ? in_img[1] : in_img[0];
    const float *om_my = p.om + (((size_t)b * H + (my_in ? my_y : 0)) * W + (my_in ? px_x : 0)) * 32;
    const float fy_base = (float)(my_y - 1), fx_base = (float)(px_x - 1);
    const float fy_max = (float)(H + 1), fx_max = (float)(W + 1);
    // LDS byte offset of this lane's piece (h) of the plain tap-0 pixel of fragment row m
    unsigned xbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) xbase[m] = (unsigned)(((wrow[m] + R) * C::TC + fr_col + R) * C::PSB + h * 16);
    // The third output fragment (channels 64..66) runs on v_mfma_f32_16x16x32 (round 5: half the matrix-pipe cycles of the 32x32x16 it
    // replaces, 4 accumulator registers per row instead of 16) on the SAME B register: read as a 16x16x32 operand, lane
    // L = 32 h + r supplies column j = r & 15, K slice kb = 2 h + (r >> 4) - the two pixel halves of the fragment row sit in
    // different K slices.  The A operand separates them again: row 4 ph + c holds W[64 + c][slice h] in K slice 2 h + ph and zeros in
    // the other pixel half's slices, so D[4 ph + c][j] is channel 64 + c of pixel 16 ph + j - lane L < 32 ends with channels
    // 64..67 of ITS OWN pixel in its four registers.  Same LDS table (rows 0..2 | zero row, two halves), another lane mapping.
    // (Non-finite data: the other pixel's contribution is removed by ZERO weights, so an Inf / NaN blended value at pixel r +- 16 - reachable
    // only through an f16 overflow - makes channels 64..66 of pixel r NaN too, where the 32x32x16 form and the reference confine it to the
    // offending pixel.  Documented in include/emavfi.h; not masked: a frame with a non-finite activation is garbage either way - ADVICE r5.)
    const int a3i = lane & 15, a3kb = lane >> 4;
    const bool a3real = (a3i >> 2) < 2 && (a3i & 3) < 3 && (a3kb & 1) == (a3i >> 2);
    const int a3row = a3real ? (a3i & 3) : 3, a3half = a3kb >> 1;
    const unsigned w3lane = (unsigned)(C::W3_OFF + (a3row * 2 + a3half) * 16);
    const int t3lane16 = (a3half * 32 + a3row) * 16;   // the same operand out of a 32x32x16 fragment of the blob (its row 3 is a zero row)
    DEFORM_STAMP(ts_converted);
    if (tid == 0) *reinterpret_cast<__attribute__((address_space(3))) unsigned *>((lds_char_t *)smem + C::SYNC_OFF) = 0u;   // (fix-up hand-shake below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window DMA has landed (hipcc emits this wait today; not relied upon - ADVICE r5)
    __syncthreads();
    DEFORM_STAMP(ts_window);

    f32x16 omr[FUSE_OFF ? 2 : 1];
    if constexpr (FUSE_OFF) {
        // ---- the pack's offset_conv (ema_vfi.py:41,56: 3x3, pad 1, 67 -> 27) on the staged window: 4 k-groups per tap
        // on the window pieces as they lie, then the three tail channels of all nine taps as three im2col k-groups
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) omr[m][i] = p.off_bias[acc_channel(i, h)];
        u32x4_t xq[2][2][4];
        auto load_x = [&](auto tc, u32x4_t (&dst)[2][4]) {
            constexpr int toff = pack3_tap_off(decltype(tc)::value);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) dst[m][kg] = lds_read16(lds_r + xbase[m] + (unsigned)(toff + kg * 32));
        };
        load_x(std::integral_constant<int, 0>{}, xq[0]);
        auto off_tap = [&](auto tc) {
            constexpr int tap = decltype(tc)::value;
            if constexpr (tap < 7) {  // weight fragments two taps ahead
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) ow[(tap + 2) % 3][kg] = *reinterpret_cast<const f16x8 *>(owbase_g + ((tap + 2) * 4 + kg) * 1024 + lane16);
            }
            if constexpr (tap < 8) load_x(std::integral_constant<int, tap + 1>{}, xq[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg)
                    if (!(EMAVFI_P3_ABL & 32) || kg == 0) mma_kg(omr[m], ow[tap % 3][kg], __builtin_bit_cast(f16x8, xq[tap & 1][m][kg]));
            __builtin_amdgcn_sched_barrier(0);
        };
        off_tap(std::integral_constant<int, 0>{}); off_tap(std::integral_constant<int, 1>{}); off_tap(std::integral_constant<int, 2>{});
        off_tap(std::integral_constant<int, 3>{}); off_tap(std::integral_constant<int, 4>{}); off_tap(std::integral_constant<int, 5>{});
        off_tap(std::integral_constant<int, 6>{}); off_tap(std::integral_constant<int, 7>{}); off_tap(std::integral_constant<int, 8>{});
        // tail: k-group j, lane (r, h) holds K = 16j + 8h + e = tap slot 4j + 2h + (e >> 2), channel 64 + (e & 3)
        {
            f16x8 ot[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) ot[j] = *reinterpret_cast<const f16x8 *>(owbase_g + C::OFF_TAIL + j * 1024 + lane16);
            u32x2_t ta[2][3][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned tb = xbase[m] - (unsigned)(h * 16) + 128u;   // tail piece of the plain tap-0 pixel
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned o = h ? (unsigned)pack3_tap_off(4 * j + 2 + u) : (unsigned)pack3_tap_off(4 * j + u);
                        ta[m][j][u] = lds_read8(lds_r + tb + o);
                    }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x4_t bq = {ta[m][j][0][0], ta[m][j][0][1], ta[m][j][1][0], ta[m][j][1][1]};
                    mma_kg(omr[m], ot[j], __builtin_bit_cast(f16x8, bq));
                }
        }
        // mask = sigmoid(third chunk), ema_vfi.py:59 (channels 18..26 after the pack-time routing)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = acc_channel(i, h);
                const float v = omr[m][i];
                // v_exp_f32 + v_rcp_f32 (1 ulp each): the value becomes an f16 blend weight; the IEEE division and libm expf of
                // the stand-alone layer cost ~25 instructions per value, 18 values per lane
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
                omr[m][i] = (c >= 18 && c < 27) ? sg : v;
            }
    }

    // ---- DCN.  Accumulators; the first weight fragments.
    f16x8 wq[2][2];  // [kg & 1][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) wq[0][n] = *reinterpret_cast<const f16x8 *>(wbase_g + n * 1024 + lane16);
    f32x16 acc[2][2];
    f32x4 acc3[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc3[m][e] = lane < 32 ? p.bias[64 + e] : 0.0f;
    }
    DEFORM_STAMP(ts_offconv);

    // ---- sampling geometry of ALL NINE taps up front (fp32, compare-free clamps: NaN -> -2; positions <= -1 or >= size sample
    // zeros).  Half-lane h handles the pixel of fragment row h.  Nine independent chains in one block instead of one dependent
    // chain in front of every tap's steps; the offset_conv's accumulators die here.  Per tap a lane keeps three registers:
    //   sample inside the staged window:  window byte offset of its top-left corner | (w00, w01) | (w10, w11) as f16 pairs
    //   sample outside (|offset| > R near the tile edge: rare):  py | px | mask as fp32 bits, bit `tap` of lane_fb set.
    // Such a sample contributes NOTHING in the tap loop (zero weights, window offset 0); the fix-up loop behind it adds the
    // missing samples from global memory.  The common path carries no fallback code.
    constexpr int OFF[4] = {0, C::PSB, C::ROWB, C::ROWB + C::PSB};
    // (three 16-element register vectors: a wave-uniform runtime index into an ext_vector lowers to an indexed register move,
    // where a switch over nine scalars was turned into a scratch-memory table by hipcc)
    typedef unsigned u32x16_t __attribute__((ext_vector_type(9)));
    u32x16_t gm0 = {}, gm1 = {}, gm2 = {};
    unsigned lane_fb = 0, fb_taps = 0;   // per-lane / wave-uniform masks over taps
    float omax = 0.0f;                   // census (DeformParams::census): largest |offset| this lane computed
    unsigned tl[12][2];                  // blended tail (channels 64..66 of this half-lane's own pixel) per tap slot; 9..11 zero
#pragma unroll
    for (int t = 9; t < 12; ++t) tl[t][0] = tl[t][1] = 0u;
    auto geom_tap = [&](auto tc) {
        constexpr int tap = decltype(tc)::value, ti = tap / 3, tj = tap - 3 * ti;
        OmTap o;
        if constexpr (FUSE_OFF) {
            // channel c of (row 0 | row 1) of this lane's pixels, delivered to (half 0 | half 1): one swap.
            // swap(a, b) -> {(a.lo, b.lo), (a.hi, b.hi)}; the channel lives in half-lane (c >> 2) & 1, register (c & 3) + 4 * (c >> 3)
            auto pick = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int reg = (c & 3) + 4 * (c >> 3);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(omr[0][reg]), __float_as_uint(omr[1][reg]), false, false);
                return __uint_as_float(((c >> 2) & 1) ? sw[1] : sw[0]);
            };
            o.dy = pick(std::integral_constant<int, 2 * tap>{});
            o.dx = pick(std::integral_constant<int, 2 * tap + 1>{});
            o.mk = pick(std::integral_constant<int, 18 + tap>{});
            if (!my_in) o.mk = 0.0f;   // pixels of the tile overhang contribute nothing (and are never stored)
        } else {
            o = load_om(om_my, tap, my_in);
        }
        if (!EMAVFI_P3_NO_CENSUS) omax = fmaxf(omax, fmaxf(fabsf(o.dy), fabsf(o.dx)));   // (one v_max3_f32 with |.| modifiers; NaN offsets are ignored)
        const float py = fminf(fmaxf((fy_base + (float)ti) + o.dy, -2.0f), fy_max);
        const float px = fminf(fmaxf((fx_base + (float)tj) + o.dx, -2.0f), fx_max);
        const float fy = floorf(py), fx = floorf(px);
        const int hl = (int)fy, wl = (int)fx;
        const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
        const float w4[4] = {o.mk * (uh * uw), o.mk * (uh * lw), o.mk * (lh * uw), o.mk * (lh * lw)};
        // window-local top-left corner; all four corners inside the staged window <=> 0 <= ly0 <= TR-2 and 0 <= lx0 <= TC-2
        const int ly0 = hl - ty0, lx0 = wl - tx0;
        const bool inside = (unsigned)ly0 <= (unsigned)(C::TR - 2) && (unsigned)lx0 <= (unsigned)(C::TC - 2);
        // pixels of the tile overhang (mask forced to 0) read their clamped window position: 0 x finite data
        const bool need_fb = !EMAVFI_DEFORM_ABL_NO_FALLBACK && !inside && my_in;
        const unsigned mybase = __umul24((unsigned)min(max(ly0, 0), C::TR - 2), (unsigned)C::ROWB) +
                                __umul24((unsigned)min(max(lx0, 0), C::TC - 2), (unsigned)C::PSB);
        // branch-free on purpose (bit masks, not ?: - hipcc turns the selects into EXEC branches that cut the block in nine)
        const unsigned keep = need_fb ? 0u : 0xffffffffu;
        const unsigned w01h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[0], (half_t)w4[1]}) & keep;
        const unsigned w23h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[2], (half_t)w4[3]}) & keep;
        gm0[tap] = (mybase & keep) | (__float_as_uint(py) & ~keep);
        gm1[tap] = w01h | (__float_as_uint(px) & ~keep);
        gm2[tap] = w23h | (__float_as_uint(o.mk) & ~keep);
        lane_fb |= ~keep & (1u << tap);
        fb_taps |= __any(need_fb) ? 1u << tap : 0u;
        // tail of this half-lane's own pixel: four 8-byte corner reads (clamped position: always a valid window address)
        // (no scheduling fences in this blend: the nine taps' chains are meant to interleave)
        u32x4_t vt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x2_t t2 = lds_read8(lds_r + mybase + (unsigned)(128 + OFF[c]));
            vt[c] = u32x4_t{t2[0], t2[1], 0u, 0u};
        }
        const u32x4_t td = __builtin_bit_cast(u32x4_t, blend_corners<2>(vt, w01h, w23h));
        tl[tap][0] = td[0]; tl[tap][1] = td[1];
    };
    geom_tap(std::integral_constant<int, 0>{}); geom_tap(std::integral_constant<int, 1>{}); geom_tap(std::integral_constant<int, 2>{});
    geom_tap(std::integral_constant<int, 3>{}); geom_tap(std::integral_constant<int, 4>{}); geom_tap(std::integral_constant<int, 5>{});
    geom_tap(std::integral_constant<int, 6>{}); geom_tap(std::integral_constant<int, 7>{}); geom_tap(std::integral_constant<int, 8>{});

    // ---- the tail channels of all nine taps: three im2col k-groups, contracted first.  Half-lane h holds its OWN row's values;
    // one swap per dword hands tap slots (4j + 2h, 4j + 2h + 1) of row m to lane (r, h) of fragment m.
    const char *wtl = wbase_g + C::DCN_TAIL;
    auto tail_mma = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        f16x8 wt[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) wt[n] = *reinterpret_cast<const f16x8 *>(wtl + (j * 3 + n) * 1024 + (n < 2 ? lane16 : t3lane16));
        unsigned bm[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const auto sw = __builtin_amdgcn_permlane32_swap(tl[4 * j + u][d], tl[4 * j + 2 + u][d], false, false);
                bm[0][2 * u + d] = sw[0];
                bm[1][2 * u + d] = sw[1];
            }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f16x8 xf = __builtin_bit_cast(f16x8, u32x4_t{bm[m][0], bm[m][1], bm[m][2], bm[m][3]});
#pragma unroll
            for (int n = 0; n < 2; ++n) mma_kg(acc[m][n], wt[n], xf);
            mma_k32(acc3[m], wt[2], xf);
        }
    };
    tail_mma(std::integral_constant<int, 0>{}); tail_mma(std::integral_constant<int, 1>{}); tail_mma(std::integral_constant<int, 2>{});
    DEFORM_STAMP(ts_geom_all);

    // ---- 9 taps x 4 k-groups x 2 rows: two fragments from global weight fragments, the third from the LDS table.
    // The tap body is a generic lambda because it runs from TWO operand sources: the staged window (corners one pixel / one row apart)
    // and, for the samples that left the window, the fix-up arena (round 6: four consecutive 144-byte corner records per entry).
    f16x8 xf_prev = {}, w3_prev = {}, w3_cur = {};   // software pipeline: the MFMAs of a step run inside the NEXT step's blend
#pragma unroll
    for (int n = 0; n < 2; ++n) wq[1][n] = f16x8{};  // read (against the zero xf_prev) by the first tap's first step
    // one tap: operands (g0, g1, g2) = {LDS byte offset of the top-left corner | (w00, w01) | (w10, w11)} of this half-lane's OWN pixel;
    // wnext = the NEXT tap's first weight fragments (null: none).  The last step's MFMAs stay pending in (xf_prev, w3_prev, wq[1]).
    auto tap_body = [&](auto arena_tag, const int tap, const char *wnext, const unsigned g0, const unsigned g1, const unsigned g2) {
        constexpr bool ARENA = decltype(arena_tag)::value;
        constexpr int OFFC[4] = {0, ARENA ? C::A_C1 : C::PSB, ARENA ? C::A_C2 : C::ROWB, ARENA ? C::A_C1 + C::A_C2 : C::ROWB + C::PSB};
        constexpr int KGS = ARENA ? C::A_KG : 32, HS = ARENA ? C::A_H : 16;   // byte strides of a k-group (two pieces) / of this half-lane's piece
        const char *wtap = wbase_g + (size_t)tap * C::DCN_TAP;   // wave-uniform
        unsigned base[2], w01[2], w23[2];
        {
            auto both = [&](unsigned x, unsigned (&out)[2]) {
                const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                out[0] = sw[0]; out[1] = sw[1];
            };
            both(g1, w01); both(g2, w23);
            // swap(a, b) = {(a.lo, b.lo), (a.hi, b.hi)}: the h = 1 receivers take b = base + 16 (their piece of the pixel)
            const auto sw = __builtin_amdgcn_permlane32_swap(g0, g0 + (unsigned)HS, false, false);
            base[0] = sw[0]; base[1] = sw[1];
            if (!ARENA && (EMAVFI_P3_ABL & 2)) { base[0] = xbase[0] + (unsigned)(tap * 16); base[1] = xbase[1] + (unsigned)(tap * 16); }
        }
        const unsigned w3a = w3lane + (unsigned)(tap * C::W3_TAP);
        // ---- the eight (k-group, row) steps, scheduled by hand.  Left to hipcc the block came out as runs of 4-6 back-to-back
        // MFMAs (the wave parked behind the matrix pipe, 32 cycles each) between runs of 16-32 blend instructions (the pipe
        // idle) - and the SIMD's other wave runs the same program.  Here the three MFMAs of step s - 1 are issued between the
        // thirds of step s's blend (5 / 5 / 6 packed FMAs, corner-major: 8 + ~24 issue cycles per 32-cycle MFMA), the corner
        // reads of step s + 1 and the next k-group's weight fragments go out at the head of step s, and sched_barrier(0)
        // pins that order.  The last step's MFMAs are carried into the next tap's first blend (into the final three behind
        // the loop); the first tap's "previous" B operand is zero.
        auto gather = [&](int s, unsigned (&d)[4][4]) {
            const int kg = s >> 1, m = s & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4_t v = lds_read16(lds_r + base[m] + (unsigned)(kg * KGS + OFFC[c]));
                d[c][0] = v[0]; d[c][1] = v[1]; d[c][2] = v[2]; d[c][3] = v[3];
            }
        };
        unsigned vb[2][4][4];
        gather(0, vb[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kg = s >> 1, m = s & 1;
            const int pm = (s + 7) & 1;              // row of the previous step (step 7 of the previous tap for s = 0)
            const int pkg = ((s + 7) & 7) >> 1;      // its k-group: weights still in wq[pkg & 1]
            if (s + 1 < 8 && !(EMAVFI_P3_ABL & 128)) gather(s + 1, vb[(s + 1) & 1]);
            f16x8 w3n;
            if (m == 0) w3n = __builtin_bit_cast(f16x8, lds_read16(lds_r + w3a + (unsigned)(kg * 128)));
            __builtin_amdgcn_sched_barrier(0);
            f16x2_t a[4];
            const unsigned (&d)[4][4] = vb[s & 1];
            const unsigned wa = w01[m], wb = w23[m];
            // ---- MFMA 0 of the previous step | blend ops 0..4
            if (!(EMAVFI_P3_ABL & 8)) mma_kg(acc[pm][0], wq[pkg & 1][0], xf_prev);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = __builtin_bit_cast(f16x2_t, d[0][q]) * bcast_half<0>(wa);
            if (!(EMAVFI_P3_ABL & 4)) a[0] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[1][0]), bcast_half<1>(wa), a[0]);
            PACK3_PIN(a);
            // ---- MFMA 1 | blend ops 5..9
            if (!(EMAVFI_P3_ABL & 8)) mma_kg(acc[pm][1], wq[pkg & 1][1], xf_prev);
            if (!(EMAVFI_P3_ABL & 4)) {
#pragma unroll
                for (int q = 1; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[1][q]), bcast_half<1>(wa), a[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[2][q]), bcast_half<0>(wb), a[q]);
            }
            PACK3_PIN(a);
            // ---- MFMA 2 | blend ops 10..15, then the weight fragments of the next k-group (behind the MFMAs that read wq[(kg+1)&1])
            if (!(EMAVFI_P3_ABL & 64)) mma_k32(acc3[pm], w3_prev, xf_prev);   // (ablation bit 6: no third-fragment MFMA in the tap loop - results wrong in channels 64..66 only)
            if (!(EMAVFI_P3_ABL & 4)) {
#pragma unroll
                for (int q = 2; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[2][q]), bcast_half<0>(wb), a[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[3][q]), bcast_half<1>(wb), a[q]);
            }
            if (m == 0 && !(EMAVFI_P3_ABL & 1)) {
                if (kg + 1 < 4) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) wq[(kg + 1) & 1][n] = *reinterpret_cast<const f16x8 *>(wtap + ((kg + 1) * 2 + n) * 1024 + lane16);
                } else if (wnext) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) wq[0][n] = *reinterpret_cast<const f16x8 *>(wnext + n * 1024 + lane16);
                }
            }
            if (m == 0) w3_cur = w3n;
            PACK3_PIN(a);
            xf_prev = f16x8{a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]};
            w3_prev = w3_cur;
        }
    };
    // the pending last step (k-group 3, row 1) of a tap sequence
    auto flush_taps = [&]() {
        mma_kg(acc[1][0], wq[1][0], xf_prev);
        mma_kg(acc[1][1], wq[1][1], xf_prev);
        mma_k32(acc3[1], w3_prev, xf_prev);
    };
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        DEFORM_STAMP(ts_tap);
        unsigned g0 = gm0[tap], g1 = gm1[tap], g2 = gm2[tap];   // wave-uniform index: indexed register moves
        if (__builtin_expect((fb_taps >> tap) & 1u, 0)) {   // wave-uniform, rare: lanes parked for the fix-up hold (py, px, mask)
            if ((lane_fb >> tap) & 1u) { g0 = 0u; g1 = 0u; g2 = 0u; }
        }
        tap_body(std::false_type{}, tap, tap < 8 ? wbase_g + (size_t)(tap + 1) * C::DCN_TAP : nullptr, g0, g1, g2);
#if EMAVFI_DEFORM_STAMPS
        DEFORM_STAMP(ts_end);
        sum_steps += ts_end - ts_tap;
#endif
    }
    flush_taps();
    DEFORM_STAMP(ts_taps_done);

    // ---- fix-up (round 6): the samples that left the window.  Round 5 gathered them from global memory in a second copy of the tap
    // body - a full wave contraction per flagged tap with its corner loads in front of the blend, 2-4 tap times each.
    // Now: every wave signals that it has left the tap loop; a wave with parked samples waits until all four have (the window is dead
    // then), takes ITS quarter of the window as an arena, fetches the parked samples' four corner records (4 x 144 B) into it with
    // LDS-DMA - up to 31 samples of ANY of its taps per round, one exposed round trip per round instead of one per tap -, and runs the
    // very tap body of the main loop on the arena (zero weights and the zero entry for every lane without a sample in the round).
    // Corners are clamped into the image and the weights of out-of-image corners are zero: the value deform_kernel computes.
    typedef __attribute__((address_space(3))) unsigned lds_u32_t;
    lds_u32_t *sync_word = reinterpret_cast<lds_u32_t *>((lds_char_t *)smem + C::SYNC_OFF);
    if (!EMAVFI_P3_NO_HANDSHAKE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's last window reads have returned
        if (lane == 0) __hip_atomic_fetch_add(sync_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned n_parked = 0;
#if EMAVFI_DEFORM_STAMPS
    unsigned long long fx_wait = 0, fx_issue = 0, fx_land = 0, fx_taps = 0;
#endif
    if (__builtin_expect(fb_taps != 0, 0)) {
#if EMAVFI_DEFORM_STAMPS
        cnt_out += __popc(fb_taps);
#endif
        const unsigned lds0 = (unsigned)(size_t)(lds_char_t *)smem;
        const unsigned arena = (unsigned)__builtin_amdgcn_readfirstlane(wave * C::ARENA_BYTES);   // (an SGPR: the DMA's M0 operand)
        lds_u32_t *table = reinterpret_cast<lds_u32_t *>((lds_char_t *)smem + C::TAB_OFF + wave * 128);
        const unsigned long long below = (1ull << lane) - 1ull;
        // (py, px, mask) of a parked lane -> clamped corner descriptor (top-left pixel | x1 - x0 << 24 | y1 - y0 << 25) and the
        // validity-masked corner weights; lanes without `sel` get zero weights
        auto fix_geom = [&](int tap, bool sel, unsigned &desc, unsigned &w01h, unsigned &w23h) {
            const unsigned g0 = gm0[tap], g1 = gm1[tap], g2 = gm2[tap];
            const float py = sel ? __uint_as_float(g0) : 0.0f, px = sel ? __uint_as_float(g1) : 0.0f;
            const float mk = sel ? __uint_as_float(g2) : 0.0f;
            const float fy = floorf(py), fx = floorf(px);
            const int hl = (int)fy, wl = (int)fx, hh = hl + 1, wh = wl + 1;
            const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
            const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
            const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
            const bool vhl = (unsigned)hl < (unsigned)H, vhh = (unsigned)hh < (unsigned)H;
            const bool vwl = (unsigned)wl < (unsigned)W, vwh = (unsigned)wh < (unsigned)W;
            const float w4[4] = {vhl && vwl ? mk * (uh * uw) : 0.0f, vhl && vwh ? mk * (uh * lw) : 0.0f,
                                 vhh && vwl ? mk * (lh * uw) : 0.0f, vhh && vwh ? mk * (lh * lw) : 0.0f};
            desc = (__umul24((unsigned)hlc, (unsigned)W) + (unsigned)wlc) | ((unsigned)(whc - wlc) << 24) | ((unsigned)(hhc - hlc) << 25);
            w01h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[0], (half_t)w4[1]});
            w23h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[2], (half_t)w4[3]});
        };
        // one LDS-DMA instruction = 64 consecutive 16-byte slots = two rows of the arena (Pack3: piece-major layout)
        auto dma = [&](const char *src, unsigned dst) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
        };
#pragma unroll 1
        for (unsigned left = fb_taps; left != 0; left &= left - 1)
            n_parked += (unsigned)__popcll(__ballot(((lane_fb >> __builtin_ctz(left)) & 1u) != 0));
        const bool convert = std::is_same<TS, bf16_t>::value && !p.in_f16;
        const char *wtl_fx = wbase_g + C::DCN_TAIL;
        const unsigned my_slot = (unsigned)lane & 31u, hb = (unsigned)lane >> 5;
#pragma unroll 1
        for (unsigned rbase = 0; rbase < n_parked; rbase += (unsigned)C::NENT) {
            const unsigned n_ent = min((unsigned)C::NENT, n_parked - rbase);
            DEFORM_STAMP(tr0);
            // -- A: the round's corner descriptors, slot by slot
            unsigned round_taps = 0, prefix = 0;
#pragma unroll 1
            for (unsigned left = fb_taps; left != 0; left &= left - 1) {
                const int tap = __builtin_ctz(left);
                const bool parked = ((lane_fb >> tap) & 1u) != 0;
                const unsigned long long bal = __ballot(parked);
                const unsigned cnt = (unsigned)__popcll(bal);
                if (prefix < rbase + n_ent && prefix + cnt > rbase) {
                    round_taps |= 1u << tap;
                    const unsigned slot = prefix + (unsigned)__popcll(bal & below) - rbase;
                    unsigned desc, wa_, wb_;
                    fix_geom(tap, parked, desc, wa_, wb_);
                    if (parked && slot < n_ent) table[slot] = desc;
                }
                prefix += cnt;
            }
            // (the first round's descriptors were computed while the other waves finished their tap loops: the window is needed from here on)
            if (rbase == 0) {
#if EMAVFI_DEFORM_STAMPS
                DEFORM_STAMP(tw0);
#endif
                // every wave of the workgroup reaches the increment above unconditionally: the wait ends
                while (*reinterpret_cast<volatile lds_u32_t *>(sync_word) < (unsigned)C::WAVES) __builtin_amdgcn_s_sleep(1);
#if EMAVFI_DEFORM_STAMPS
                { DEFORM_STAMP(tw1); fx_wait = tw1 - tw0; }
#endif
            }
            // -- B: fetch.  Lane (slot, hb) owns corners 2 hb and 2 hb + 1 of its slot: instructions 0..8 their first, 9..17 their second,
            // piece by piece (piece 8 = the tail channels, from the compact tail buffer when the pack has one).  Slots past the round's
            // entries and slot 31 read the zero page.  The DMA is inline asm: hipcc places no waits for it, C below does.
            {
                const unsigned d = table[my_slot];
                const bool live = my_slot < n_ent;
                const unsigned cpa = (d & 0xffffffu) + (hb && ((d >> 25) & 1u) ? (unsigned)W : 0u), cpb = cpa + ((d >> 24) & 1u);
                const char *pa = live ? gplane + (size_t)__umul24(cpa, ps_bytes) : zeros, *pb = live ? gplane + (size_t)__umul24(cpb, ps_bytes) : zeros;
                const char *ta = !live ? zeros : (tplane ? tplane + (size_t)__umul24(cpa, tail_bytes) : pa + 128);
                const char *tb = !live ? zeros : (tplane ? tplane + (size_t)__umul24(cpb, tail_bytes) : pb + 128);
                const unsigned step = live ? 16u : 0u;
                const unsigned dst0 = lds0 + arena;
#pragma unroll
                for (int i = 0; i < C::SP - 1; ++i) { dma(pa, dst0 + (unsigned)(i * 1024)); pa += step; }
                dma(ta, dst0 + (unsigned)((C::SP - 1) * 1024));
#pragma unroll
                for (int i = 0; i < C::SP - 1; ++i) { dma(pb, dst0 + (unsigned)((C::SP + i) * 1024)); pb += step; }
                dma(tb, dst0 + (unsigned)((2 * C::SP - 1) * 1024));
            }
            // the round's first weight fragments travel under the DMA
            {
                const char *w0 = wbase_g + (size_t)__builtin_ctz(round_taps) * C::DCN_TAP;
#pragma unroll
                for (int n = 0; n < 2; ++n) { wq[0][n] = *reinterpret_cast<const f16x8 *>(w0 + n * 1024 + lane16); wq[1][n] = f16x8{}; }
                xf_prev = f16x8{}; w3_prev = f16x8{};
            }
            // -- C: landed (in-order return: the fragments above too); bf16 storage: every lane converts the slots it fetched
            DEFORM_STAMP(tr1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DEFORM_STAMP(tr2);
            if (convert && my_slot < n_ent) {   // (the other slots hold zeros)
#pragma unroll
                for (int i = 0; i < C::ADMA; ++i) {
                    lds_char_t *q = (lds_char_t *)smem + arena + (unsigned)(i * 1024) + lane16;
                    const u32x4_t v = to_f16_piece<TS>(lds_read16(q));
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(q) = v;
                }
            }
            // -- D: the round's taps on the arena
            DEFORM_STAMP(tr3);
            prefix = 0;
#pragma unroll 1
            for (unsigned left = fb_taps; left != 0; left &= left - 1) {
                const int tap = __builtin_ctz(left);
                const bool parked = ((lane_fb >> tap) & 1u) != 0;
                const unsigned long long bal = __ballot(parked);
                const unsigned cnt = (unsigned)__popcll(bal);
                const unsigned slot = prefix + (unsigned)__popcll(bal & below) - rbase;
                prefix += cnt;
                if (!((round_taps >> tap) & 1u)) continue;
                const bool mine = parked && slot < n_ent;
                unsigned desc, w01h, w23h;
                fix_geom(tap, mine, desc, w01h, w23h);
                const unsigned ent = arena + (mine ? slot : (unsigned)C::NENT) * 16u;
                // the tail's weight fragments (one im2col k-group with this tap's slot alone) travel under the tap's steps
                const int j = tap >> 2, hsel = (tap >> 1) & 1, u = tap & 1;
                f16x8 wt[3];
#pragma unroll
                for (int n = 0; n < 3; ++n) wt[n] = *reinterpret_cast<const f16x8 *>(wtl_fx + (j * 3 + n) * 1024 + (n < 2 ? lane16 : t3lane16));
                const unsigned later = round_taps & ~((2u << tap) - 1u);
                tap_body(std::true_type{}, tap, later ? wbase_g + (size_t)__builtin_ctz(later) * C::DCN_TAP : nullptr, ent, w01h, w23h);
                // the tail channels (64..66) of the same samples
                {
                    constexpr int OC[4] = {0, C::A_C1, C::A_C2, C::A_C1 + C::A_C2};
                    u32x4_t vt[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x2_t t2 = lds_read8(lds_r + ent + (unsigned)((C::SP - 1) * C::A_H + OC[c]));
                        vt[c] = u32x4_t{t2[0], t2[1], 0u, 0u};
                    }
                    const u32x4_t td = __builtin_bit_cast(u32x4_t, blend_corners<2>(vt, w01h, w23h));
                    // K = 16 j + 8 h' + 4 u + channel with tap = 4 j + 2 h' + u: only lanes of half h' carry it, in dwords (2u, 2u + 1)
                    unsigned tm[2][2];
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(td[dd], td[dd], false, false);   // {row 0's, row 1's} in both halves
                        tm[0][dd] = h == hsel ? sw[0] : 0u;
                        tm[1][dd] = h == hsel ? sw[1] : 0u;
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const u32x4_t bq = u ? u32x4_t{0u, 0u, tm[m][0], tm[m][1]} : u32x4_t{tm[m][0], tm[m][1], 0u, 0u};
                        const f16x8 xf = __builtin_bit_cast(f16x8, bq);
#pragma unroll
                        for (int n = 0; n < 2; ++n) mma_kg(acc[m][n], wt[n], xf);
                        mma_k32(acc3[m], wt[2], xf);
                    }
                }
            }
            flush_taps();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the round's arena reads have returned before the next round's DMA overwrites them
#if EMAVFI_DEFORM_STAMPS
            { DEFORM_STAMP(tr4); fx_issue += tr1 - tr0; fx_land += tr3 - tr1; fx_taps += tr4 - tr3; }
#endif
        }
    }
    DEFORM_STAMP(ts_loop);
    // ---- census of this launch (emavfi_forward_census / emavfi_mdcn_census): (wave, tap) groups that took the fix-up, samples outside the
    // window, and the largest |offset| of a wave that had one - ONLY such waves pay for it (a same-box A/B priced an unconditional
    // wave reduction + atomic at 40-55 us per launch, 3-4 %: profiles/r06_experiments_that_lost.txt).  64 slots of {u32 x 4} per launch,
    // no-return atomics.  A launch without a flagged wave reports max |offset| 0 = "every sample inside the +-2 px window".
    if (!EMAVFI_P3_NO_CENSUS && __builtin_expect(fb_taps != 0, 0) && p.census) {
        float om = my_in ? omax : 0.0f;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) om = fmaxf(om, __shfl_xor(om, sh));
        if (lane == 0) {
            unsigned *cs = p.census + (blockIdx.x & 63u) * 4u;
            (void)__hip_atomic_fetch_add(cs, (unsigned)__popc(fb_taps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(cs + 1, n_parked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_max(cs + 2, __float_as_uint(om), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue (no activation: ema_vfi.py:136-138 chains the blocks directly)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!in_img[m]) continue;
        TS *op = reinterpret_cast<TS *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
        // the third fragment: lanes < 32 hold channels 64..67 of their own pixel; 68..71 have zero weights (their bias alone)
        auto store_third = [&](auto *o16) {
            typedef typename std::remove_pointer<decltype(o16)>::type O;
            typedef __attribute__((ext_vector_type(2))) O pair_t;
            if (p.cstore > 64 && h == 0) {
                const pair_t q0 = {(O)acc3[m][0], (O)acc3[m][1]}, q1 = {(O)acc3[m][2], (O)acc3[m][3]};
                const pair_t q2 = {(O)p.bias[68], (O)p.bias[69]}, q3 = {(O)p.bias[70], (O)p.bias[71]};
                *reinterpret_cast<uint4 *>(o16 + 64) = make_uint4(__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1),
                                                                  __builtin_bit_cast(unsigned, q2), __builtin_bit_cast(unsigned, q3));
            }
        };
        if (std::is_same<TS, bf16_t>::value && p.out_f16) {
            half_t *oh = reinterpret_cast<half_t *>(op);
#pragma unroll
            for (int n = 0; n < 2; ++n) store_frag(oh + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
            store_third(oh);
        } else {
#pragma unroll
            for (int n = 0; n < 2; ++n) store_frag(op + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
            store_third(op);
        }
    }
#if EMAVFI_DEFORM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DEFORM_STAMP(ts_done);
    if (p.stamps && lane == 0 && blockIdx.x % DEFORM_STAMP_STRIDE == 0) {
        const unsigned row = (blockIdx.x / DEFORM_STAMP_STRIDE) * C::WAVES + wave;
        if (row < DEFORM_STAMP_ROWS) {
            unsigned long long *o = p.stamps + (size_t)row * 8;
            // (o[2] bits 32..63: the fix-up pass - hand-shake wait, arena rounds, its taps; o[4]: flagged taps << 32, parked samples << 40)
            // (bits 32..63 of o[0] / o[1] / o[3] / o[5]: the fix-up's hand-shake wait / descriptor + DMA issue / DMA landing + conversion / taps)
            o[0] = (ts_window - ts_begin) | (fx_wait << 32); o[1] = (ts_offconv - ts_window) | ((fx_issue - fx_wait) << 32);
            o[2] = (ts_geom_all - ts_offconv) | ((ts_loop - ts_taps_done) << 32); o[3] = sum_steps | (fx_land << 32);
            o[4] = (ts_done - ts_loop) | (cnt_out << 32) | ((unsigned long long)n_parked << 40); o[5] = (ts_done - ts_begin) | (fx_taps << 32); o[6] = 1;
            auto q16 = [](unsigned long long v) { v >>= 2; return v > 0xffffull ? 0xffffull : v; };
            o[7] = q16(ts_issued - ts_begin) | (q16(ts_landed - ts_issued) << 16) | (q16(ts_converted - ts_landed) << 32) | (q16(ts_window - ts_converted) << 48);
        }
    }
#endif
}

template <typename TS, bool FUSE_OFF> static int launch_deform_pack3(const DeformParams &p, hipStream_t s)
{
    using C = Pack3;
    constexpr int LDS_REQ = EMAVFI_P3_ONE_WG ? 100 * 1024 : C::LDS_BYTES;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&deform_pack3_kernel<TS, FUSE_OFF>), LDS_REQ); e_ != hipSuccess) return (int)e_;
    const long long nwg = (long long)((p.W + C::TCOLS - 1) / C::TCOLS) * ((p.H + C::TROWS - 1) / C::TROWS) * p.B;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    deform_pack3_kernel<TS, FUSE_OFF><<<(unsigned)nwg, C::THREADS, LDS_REQ, s>>>(p);
    return (int)hipGetLastError();
}
