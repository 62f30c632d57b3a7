// ModulatedDeformConvPack (ema_vfi.py:23-60) at the reference width as ONE launch, 16-bit storage types.
// Same operator and MFMA fragment scheme as deform.inl (D[cout][pixel] += W_k[cout][c] * S_k[c][pixel], every lane
// gathers and blends its own B-operand piece); what this kernel adds:
//
//   * the input window = tile + halo of 1 (3x3 taps) + R (offset reach) + 1 (bilinear) pixels, 72 channels, is staged in
//     LDS once by global->LDS DMA (every lane its own source address; out-of-image pixels read a zero page), 9 sixteen-
//     byte slots per pixel = an odd slot stride, so neighbouring-pixel ds_read_b128 gathers are bank-conflict free;
//   * ON CHIP THE WINDOW IS ALWAYS IEEE f16.  bf16 storage is converted in place right after the DMA (each wave converts
//     the pieces it fetched itself: exact for |v| in [6.1e-5, 65504], v_cvt_pkrtz saturates beyond, keeps sub-normals),
//     the packed weights of the two convolutions are the bf16-rounded values stored as f16 (exact), and the contraction
//     runs on v_mfma_f32_32x32x16_f16.  Why: gfx950 has no packed bf16 arithmetic; the bilinear blend on
//     v_dot2_f32_bf16 (one useful product per instruction) measured 6.3 cycles per instruction and does NOT overlap the
//     MFMAs (tools/microbench/valu_rates.hip: MFMA + 8 dot2 = 83 cycles, MFMA + 8 v_pk_fma_f16 = 45), whereas
//     v_pk_fma_f16 blends two channels per instruction beside the matrix pipe.  The products W*S are the same real
//     numbers as in bf16 (weights and data are bf16 values), the blended sample keeps 11 significant bits instead of 8;
//   * a tap's four corners are ONE window address + three immediates (the window is zero outside the image, so corner
//     validity needs no per-corner clamp or weight mask); lanes whose sample leaves the window (|offset| > R near the
//     tile edge) take a rare divergent path: validity-masked weights and a global gather, same value either way;
//   * the sampling geometry of a pixel is computed once (half-lane h does fragment row h) and handed to the other
//     half with v_permlane32_swap: 4 dwords per tap; the (dy, dx, mask) values of the fused offset_conv never leave
//     the accumulator registers and are picked with one swap each;
//   * packed weights never touch LDS: MFMA A fragments come from L2/L1 one k-group ahead, no per-tap barrier;
//   * the last k-group holds 3 real channels (64..66): only 2 of its 4 dwords are gathered and blended.
//
// Tile: 16 x TCOLS pixels, a wave owns two 32-pixel fragments.  TCOLS = 32: 512 threads, 126 KiB window, one workgroup
// per CU.  TCOLS = 16: 256 threads, 75 KiB window, TWO workgroups per CU whose prologues / epilogues overlap the other's
// main loop (fragment = 2 rows x 16 columns).
#include "deform.inl"
#include <mutex>
#include <type_traits>

#ifndef EMAVFI_DEFORM_XCD_ORDER
#define EMAVFI_DEFORM_XCD_ORDER 1   // 0: plain row-major tile order (A/B switch)
#endif
#ifndef EMAVFI_DEFORM_ABL_NO_WLOADS
#define EMAVFI_DEFORM_ABL_NO_WLOADS 0  // timing-only ablation (wrong results): no weight-fragment loads after the first
#endif
#ifndef EMAVFI_DEFORM_ABL_NO_FALLBACK
#define EMAVFI_DEFORM_ABL_NO_FALLBACK 0  // timing-only ablation (wrong results): samples leaving the window read clamped window pixels
#endif

// Diagnostic build only (-DEMAVFI_DEFORM_STAMPS=1; cdna_hip_programming.md section 7, in-kernel stamps): s_memtime at
// the seams of the kernel, per-wave segment sums written to DeformParams::stamps (a buffer the diagnostic build of
// emavfi_api.hip allocates), read back with emavfi_debug_deform_stamps().  Never part of the shipped library.
#ifndef EMAVFI_DEFORM_STAMPS
#define EMAVFI_DEFORM_STAMPS 0
#endif
#if EMAVFI_DEFORM_STAMPS
__device__ __forceinline__ unsigned long long deform_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DEFORM_STAMP(var) const unsigned long long var = deform_stamp()
#elif defined(EMAVFI_DEFORM_FENCES)
#define DEFORM_STAMP(var) __builtin_amdgcn_sched_barrier(0)   // experiment: the stamped build's scheduling fences without its stamps
#else
#define DEFORM_STAMP(var)
#endif

template <int TCOLS_, int R_> struct PackCfg {
    static constexpr int CK = 80, NF = 3, CS = 72, R = R_;
    static constexpr int TROWS = 16, TCOLS = TCOLS_;
    static constexpr int FC = TCOLS < 32 ? TCOLS : 32, FR = 32 / FC;      // a 32-pixel fragment = FR rows x FC columns
    static constexpr int NFRAG = TROWS * TCOLS / 32, WAVES = NFRAG / 2, THREADS = 64 * WAVES;
    static constexpr int TR = TROWS + 3 + 2 * R, TC = TCOLS + 3 + 2 * R;
    static constexpr int SP = CS * 2 / 16, PSB = SP * 16, ROWB = TC * PSB;  // slots / bytes per pixel, bytes per window row
    static constexpr int KG = CK / 16, WTAP = KG * NF * 1024;
    static constexpr int NSLOT = TR * TC * SP, NINST = (NSLOT + 63) / 64, PER_WAVE = (NINST + WAVES - 1) / WAVES;
    static constexpr int LDS_BYTES = NINST * 1024;
    static_assert(FR * FC == 32 && TROWS % (2 * FR) == 0, "fragment shape");
    static_assert((SP & 1) == 1, "odd slot stride keeps neighbouring-pixel gathers conflict free");
    static_assert(ROWB + PSB + (SP - 1) * 16 + 15 < 65536, "corner offsets must fit the ds_read immediate");
    static_assert(LDS_BYTES <= 160 * 1024, "window does not fit the 160 KiB LDS");
};

typedef __attribute__((address_space(3))) const char lds_cchar_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ u32x4_t lds_read16(lds_cchar_t *p) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(p); }
__device__ __forceinline__ u32x2_t lds_read8(lds_cchar_t *p) { return *reinterpret_cast<__attribute__((address_space(3))) const u32x2_t *>(p); }

// bf16 pair -> f16 pair (exact in f16's normal range; round-toward-zero only matters below 2^-14, saturates above 65504)
__device__ __forceinline__ unsigned bf16x2_to_f16x2(unsigned d)
{
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)));
}
template <typename TS> __device__ __forceinline__ u32x4_t to_f16_piece(u32x4_t v)
{
    if constexpr (std::is_same<TS, bf16_t>::value)
        return u32x4_t{bf16x2_to_f16x2(v[0]), bf16x2_to_f16x2(v[1]), bf16x2_to_f16x2(v[2]), bf16x2_to_f16x2(v[3])};
    return v;
}

// (w_lo, w_hi) packed f16 pair; blend on v_pk_mul_f16 / v_pk_fma_f16 with one half of the weight register broadcast to both
// channels of a dword.  Written as a splat the compiler folds into the instruction's op_sel modifiers (checked in the
// ISA: no v_perm / v_pack) - NOT as inline asm: an asm result consumed by the next MFMA gets no VALU-write -> MFMA-read
// wait states from hipcc (cdna_hip_programming.md 5.7), and the first version of this kernel read stale operand
// registers in exactly that way (odd fragment rows, first channel fragment only).
template <int HALF> __device__ __forceinline__ f16x2_t bcast_half(unsigned w)
{
    const f16x2_t p = __builtin_bit_cast(f16x2_t, w);
    return __builtin_shufflevector(p, p, HALF, HALF);
}
// four corner pieces (f16, NQ dwords each) -> one MFMA operand fragment; corners in the order (y0,x0) (y0,x1) (y1,x0) (y1,x1),
// weights w01 = (w00, w01), w23 = (w10, w11).  Accumulation in f16: four terms, the size of the final rounding.
template <int NQ> __device__ __forceinline__ f16x8 blend_corners(const u32x4_t (&v)[4], unsigned w01, unsigned w23)
{
    const f16x2_t z = {(half_t)0.0f, (half_t)0.0f};
    f16x2_t a[4] = {z, z, z, z};
    // explicit dword arrays: subscripting v[c][q] with the loop variable made hipcc (ROCm 7.2) use element 0 for every q
    const unsigned d0[4] = {v[0][0], v[0][1], v[0][2], v[0][3]}, d1[4] = {v[1][0], v[1][1], v[1][2], v[1][3]};
    const unsigned d2[4] = {v[2][0], v[2][1], v[2][2], v[2][3]}, d3[4] = {v[3][0], v[3][1], v[3][2], v[3][3]};
    // corner-major: the NQ chains are independent, so consecutive instructions never depend on each other
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_bit_cast(f16x2_t, d0[q]) * bcast_half<0>(w01);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d1[q]), bcast_half<1>(w01), a[q]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d2[q]), bcast_half<0>(w23), a[q]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d3[q]), bcast_half<1>(w23), a[q]);
    return f16x8{a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]};
}

struct OmTap { float dy, dx, mk; };
__device__ __forceinline__ OmTap load_om(const float *__restrict__ om, int tap, bool in_image)
{
    OmTap t;
    t.dy = om[2 * tap]; t.dx = om[2 * tap + 1]; t.mk = in_image ? om[18 + tap] : 0.0f;
    return t;
}

// TS = storage type of x / out (bf16_t or half_t).  TQ = dwords of the LAST k-group's piece that can hold real channels
// (cin_real <= 64 + 2 * TQ).  p.w / p.off_w hold f16 fragments (for TS = bf16_t: the bf16-rounded weights as f16).
template <typename TS, int TCOLS, int R, bool FUSE_OFF, int TQ>
__global__ __launch_bounds__(16 * TCOLS, 2) void deform_pack_kernel(const DeformParams p)
{
    using C = PackCfg<TCOLS, R>;
    constexpr int NF = C::NF, KG = C::KG;
    static_assert(sizeof(TS) == 2, "16-bit storage types only");
    static_assert(C::THREADS == 16 * TCOLS, "launch bounds");
    static_assert((KG & 1) == 1, "cross-tap weight prefetch slot assumes an odd k-group count");
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_cchar_t *lds_r = (lds_cchar_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    DEFORM_STAMP(ts_begin);
#if EMAVFI_DEFORM_STAMPS
    unsigned long long sum_geom = 0, sum_steps = 0, cnt_out = 0, cnt_lanes = 0;
#endif

    // ---- tile of this workgroup.  XCD-aware order (placement only affects speed): workgroups are dealt round-robin
    // over the 8 XCDs, so blockIdx % 8 labels an XCD group; each group gets a contiguous run of tiles, and inside an
    // image tiles are walked in strips of SROWS tile rows, column by column, so the tiles an XCD has in flight form a
    // compact 2-D block whose window halos are re-read from that XCD's L2 instead of from HBM.
    const int ntx = (W + C::TCOLS - 1) / C::TCOLS, nty = (H + C::TROWS - 1) / C::TROWS, nt = ntx * nty;
    int tile_x, tile_y, b;
    if (EMAVFI_DEFORM_XCD_ORDER) {
        constexpr int SROWS = 4;
        const int nwg = gridDim.x, grp = blockIdx.x & 7, kk = blockIdx.x >> 3, qq = nwg >> 3, rr = nwg & 7;
        const int wg = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
        b = wg / nt;
        const int t = wg - b * nt, strip = t / (SROWS * ntx), tt = t - strip * SROWS * ntx;
        const int rows = min(SROWS, nty - strip * SROWS);
        tile_x = tt / rows;
        tile_y = strip * SROWS + (tt - tile_x * rows);
    } else {
        b = blockIdx.x / nt;
        const int t = blockIdx.x - b * nt;
        tile_y = t / ntx;
        tile_x = t - tile_y * ntx;
    }
    const unsigned ps_bytes = (unsigned)p.x_ps * 2u, tail_bytes = (unsigned)p.tail_ps * 2u;
    const int ty0 = tile_y * C::TROWS - 1 - R, tx0 = tile_x * C::TCOLS - 1 - R;
    const char *gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;
    const char *tplane = p.x_tail ? (const char *)p.x_tail + (size_t)b * H * W * tail_bytes : nullptr;
    const char *zeros = (const char *)p.zeros;

    // ---- small loads first (L2-resident): the first two taps' offset_conv fragments.  The DCN accumulators are only
    // initialised after the offset_conv: with their 96 registers live hipcc sank the fragment prefetches next to their
    // MFMAs (load -> vmcnt(0) -> MFMA, one L2 round trip per MFMA: 143 cycles per MFMA measured with in-kernel stamps)
    const char *wlane = (const char *)p.w + lane * 16;
    const char *owl = (const char *)p.off_w + lane * 16;
    f16x8 ow[FUSE_OFF ? 3 : 1][KG];
    if constexpr (FUSE_OFF) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) ow[t][kg] = *reinterpret_cast<const f16x8 *>(owl + (t * KG + kg) * 1024);
    }

    // ---- DMA the input window (zero outside the image).  Slot sl of the LDS image = 16-byte piece pc of window pixel pix;
    // a wave's consecutive DMA instructions are 64 * WAVES slots apart: (pix, pc) advance incrementally, no divisions.
    {
        constexpr int STEP = 64 * C::WAVES, DPIX = STEP / C::SP, DPC = STEP % C::SP, DLY = DPIX / C::TC, DLX = DPIX % C::TC;
        const int sl0 = wave * 64 + lane;
        int pix = sl0 / C::SP, pc = sl0 - pix * C::SP;
        int ly = pix / C::TC, lx = pix - ly * C::TC;
#pragma unroll
        for (int i = 0; i < C::PER_WAVE; ++i) {
            const int j = i * C::WAVES + wave;
            if (j < C::NINST) {  // wave-uniform
                const int gy = ty0 + ly, gxx = tx0 + lx;
                const bool ok = (i * STEP + sl0 < C::NSLOT) && (unsigned)gy < (unsigned)H && (unsigned)gxx < (unsigned)W;
                const unsigned pixel = __umul24((unsigned)gy, (unsigned)W) + (unsigned)gxx;
                const char *src = gplane + (size_t)(__umul24(pixel, ps_bytes) + (unsigned)pc * 16u);
                if (tplane && pc == C::SP - 1) src = tplane + (size_t)__umul24(pixel, tail_bytes);  // channels 64..71
                if (!ok) src = zeros;
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(smem + j * 1024), 16, 0, 0);
            }
            pc += DPC; lx += DLX; ly += DLY;
            if (pc >= C::SP) { pc -= C::SP; lx += 1; }
            if (lx >= C::TC) { lx -= C::TC; ly += 1; }
            if (lx >= C::TC) { lx -= C::TC; ly += 1; }
        }
    }
    DEFORM_STAMP(ts_issued);
#if EMAVFI_DEFORM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    DEFORM_STAMP(ts_landed);
    if constexpr (std::is_same<TS, bf16_t>::value) {
        // bf16 -> f16 in place: every wave converts exactly the pieces its own DMA instructions fetched, so its own
        // vmcnt(0) is the only wait needed before it reads them back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < C::PER_WAVE; ++i) {
            const int j = i * C::WAVES + wave;
            if (j < C::NINST) {
                lds_char_t *q = (lds_char_t *)smem + j * 1024 + lane * 16;
                const u32x4_t v = to_f16_piece<TS>(lds_read16(q));
                *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(q) = v;
            }
        }
    }

    // ---- this lane's pixel in each of its wave's two fragments, shared by its h = 0 / h = 1 partner lanes
    // Pixel of lane r inside its 32-pixel fragment.  A fragment of 2 rows x 16 columns is NOT laid out lane-linearly: a
    // ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS
    // section), and with lanes 0-15 on row 0 and 16-31 on row 1 every group mixed eight pixels of each row - whose bank
    // quads collide pairwise unless the window row stride is a multiple of 256 bytes (SQ_LDS_BANK_CONFLICT: 50 % of the
    // LDS cycles).  Giving each hardware group one row of 16 consecutive pixels makes the undeformed reads conflict free.
    int fr_row, fr_col;
    if (C::FC == 16) {
        const bool g2 = (r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28;
        fr_row = g2 ? 1 : 0;
        fr_col = g2 ? (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16)) : (r < 4 ? r : (r < 16 ? r - 8 : r - 12));
    } else {
        fr_row = r / C::FC;
        fr_col = r - fr_row * C::FC;
    }
    const int px_x = tile_x * C::TCOLS + fr_col;
    int py_y[2], wrow[2];   // image row; window row of the plain (undeformed) tap centre minus (1 + R)
    bool in_img[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        wrow[m] = (wave * 2 + m) * C::FR + fr_row;
        py_y[m] = tile_y * C::TROWS + wrow[m];
        in_img[m] = py_y[m] < H && px_x < W;
    }
    // half-lane h computes the sampling geometry of fragment row h (both rows end up in both halves by one swap each)
    const int my_y = h ? py_y[1] : py_y[0];
    const bool my_in = h ? in_img[1] : in_img[0];
    const float *om_my = p.om + (((size_t)b * H + (my_in ? my_y : 0)) * W + (my_in ? px_x : 0)) * 32;
    OmTap nxt;
    if (!FUSE_OFF) nxt = load_om(om_my, 0, my_in);
    const float fy_base = (float)(my_y - 1), fx_base = (float)(px_x - 1);
    const float fy_max = (float)(H + 1), fx_max = (float)(W + 1);
    DEFORM_STAMP(ts_converted);
    __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
    DEFORM_STAMP(ts_window);

    // fused: this lane's half of its pixels' (dy, dx, mask) values stays in registers in accumulator layout:
    // channel c sits in half-lane (c >> 2) & 1, register (c & 3) + 4 * (c >> 3)
    f32x16 omr[FUSE_OFF ? 2 : 1];
    if constexpr (FUSE_OFF) {
        // ---- the pack's offset_conv (ema_vfi.py:41,56: 3x3, pad 1, cin -> 27) on the staged window ----
        // Plain taps: the B operand of lane (r, h) is a 16-byte piece of window pixel (row + i + R, col + j + R), read as it
        // lies; out-of-image pixels were zero-filled by the DMA = the conv's zero padding.  Weight fragments come from L2
        // two taps ahead.  Same tap / k-group order and epilogue as the stand-alone conv3x3 EPI_OM layer.
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) omr[m][i] = p.off_bias[acc_channel(i, h)];
        // B operands of a tap: 2 rows x KG pieces, read one tap AHEAD of their MFMAs (double buffered; the DCN
        // accumulators are not live yet, so the registers are free).  Unpipelined, every MFMA waited for its own
        // ds_read: 125 cycles per MFMA.
        u32x4_t xq[2][2][KG];
        auto load_x = [&](int tap, u32x4_t (&dst)[2][KG]) {
            const int i = tap / 3, j = tap - 3 * i;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                lds_cchar_t *xp = lds_r + ((wrow[m] + i + R) * C::TC + (fr_col + j + R)) * C::PSB;
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    const int slot = (2 * kg + h < C::SP) ? 2 * kg + h : C::SP - 1;
                    dst[m][kg] = lds_read16(xp + slot * 16);
                }
            }
        };
        load_x(0, xq[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 7 && !EMAVFI_DEFORM_ABL_NO_WLOADS) {  // weight fragments two taps ahead (L2 latency is ~3 taps of MFMA time for one wave)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) ow[(tap + 2) % 3][kg] = *reinterpret_cast<const f16x8 *>(owl + ((tap + 2) * KG + kg) * 1024);
            }
            if (tap < 8) load_x(tap + 1, xq[(tap + 1) & 1]);
            // keep the prefetches where they are written: left alone, hipcc sinks each load next to its MFMA to save
            // registers (load -> wait -> MFMA: one L2 / LDS round trip per MFMA)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) mma_kg(omr[m], ow[tap % 3][kg], __builtin_bit_cast(f16x8, xq[tap & 1][m][kg]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // mask = sigmoid(third chunk), ema_vfi.py:59 (channels 18..26 after the pack-time routing)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = acc_channel(i, h);
                const float v = omr[m][i];
                omr[m][i] = (c >= 18 && c < 27) ? 1.0f / (1.0f + expf(-v)) : v;
            }
    }
    f16x8 wq[2][NF];  // [kg & 1]; the next tap's first k-group arrives in wq[1] (KG is odd) and moves to wq[0]
#pragma unroll
    for (int n = 0; n < NF; ++n) wq[1][n] = *reinterpret_cast<const f16x8 *>(wlane + n * 1024);
    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];
    DEFORM_STAMP(ts_offconv);

    const char *gx = gplane + h * 16;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        DEFORM_STAMP(ts_tap);
#pragma unroll
        for (int n = 0; n < NF; ++n) wq[0][n] = wq[1][n];
        const char *wtap = wlane + (size_t)tap * C::WTAP;
        // ---- (dy, dx, mask) of this half-lane's row for this tap
        OmTap o;
        if constexpr (FUSE_OFF) {
            // channel c of (row 0 | row 1) of this lane's pixels, delivered to (half 0 | half 1): one swap.
            // swap(a, b) -> {(a.lo, b.lo), (a.hi, b.hi)}; the channel lives in half-lane (c >> 2) & 1
            auto pick = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int reg = (c & 3) + 4 * (c >> 3);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(omr[0][reg]), __float_as_uint(omr[1][reg]), false, false);
                return __uint_as_float(((c >> 2) & 1) ? sw[1] : sw[0]);
            };
            auto take = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                o.dy = pick(std::integral_constant<int, 2 * k>{});
                o.dx = pick(std::integral_constant<int, 2 * k + 1>{});
                o.mk = pick(std::integral_constant<int, 18 + k>{});
            };
            switch (tap) {  // wave-uniform; registers cannot be indexed by a loop variable
            case 0: take(std::integral_constant<int, 0>{}); break;
            case 1: take(std::integral_constant<int, 1>{}); break;
            case 2: take(std::integral_constant<int, 2>{}); break;
            case 3: take(std::integral_constant<int, 3>{}); break;
            case 4: take(std::integral_constant<int, 4>{}); break;
            case 5: take(std::integral_constant<int, 5>{}); break;
            case 6: take(std::integral_constant<int, 6>{}); break;
            case 7: take(std::integral_constant<int, 7>{}); break;
            default: take(std::integral_constant<int, 8>{}); break;
            }
            if (!my_in) o.mk = 0.0f;  // pixels of the tile overhang contribute nothing (and are never stored)
        } else {
            o = nxt;
            if (tap < 8) nxt = load_om(om_my, tap + 1, my_in);
        }
        // ---- sampling geometry (fp32, compare-free clamps: NaN -> -2; positions <= -1 or >= size sample zeros)
        const int ti = tap / 3, tj = tap - 3 * ti;
        const float py = fminf(fmaxf((fy_base + (float)ti) + o.dy, -2.0f), fy_max);
        const float px = fminf(fmaxf((fx_base + (float)tj) + o.dx, -2.0f), fx_max);
        const float fy = floorf(py), fx = floorf(px);
        const int hl = (int)fy, wl = (int)fx;
        const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
        float w4[4] = {o.mk * (uh * uw), o.mk * (uh * lw), o.mk * (lh * uw), o.mk * (lh * lw)};
        // window-local top-left corner; all four corners inside the staged window <=> 0 <= ly0 <= TR-2 and 0 <= lx0 <= TC-2
        const int ly0 = hl - ty0, lx0 = wl - tx0;
        const bool inside = (unsigned)ly0 <= (unsigned)(C::TR - 2) && (unsigned)lx0 <= (unsigned)(C::TC - 2);
        // clamped so that every lane's LDS reads stay in bounds; bit 0 flags a lane that must gather from global memory
        unsigned wbase = __umul24((unsigned)min(max(ly0, 0), C::TR - 2), (unsigned)C::ROWB) +
                         __umul24((unsigned)min(max(lx0, 0), C::TC - 2), (unsigned)C::PSB);
        unsigned gpk = 0;  // fallback lanes: top-left pixel index | (x1 - x0) << 24 | (y1 - y0) << 25 (clamped corners)
        if (!inside) {     // rare, divergent: the sample leaves the window -> clamped corners, validity-masked weights
            const int hh = hl + 1, wh = wl + 1;
            const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
            const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
            const bool vhl = (unsigned)hl < (unsigned)H, vhh = (unsigned)hh < (unsigned)H;
            const bool vwl = (unsigned)wl < (unsigned)W, vwh = (unsigned)wh < (unsigned)W;
            if (!(vhl && vwl)) w4[0] = 0.0f;
            if (!(vhl && vwh)) w4[1] = 0.0f;
            if (!(vhh && vwl)) w4[2] = 0.0f;
            if (!(vhh && vwh)) w4[3] = 0.0f;
            gpk = (__umul24((unsigned)hlc, (unsigned)W) + (unsigned)wlc) | ((unsigned)(whc - wlc) << 24) | ((unsigned)(hhc - hlc) << 25);
            wbase |= 1u;
        }
        const unsigned w01h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[0], (half_t)w4[1]});
        const unsigned w23h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[2], (half_t)w4[3]});
        // both halves get both rows: swap(x, x) = {row 0's, row 1's}
        unsigned base[2], w01[2], w23[2], gp[2];
        {
            auto both = [&](unsigned x, unsigned (&out)[2]) {
                const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                out[0] = sw[0]; out[1] = sw[1];
            };
            both(wbase, base); both(w01h, w01); both(w23h, w23); both(gpk, gp);
        }
        bool lane_out[2], any_out[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            lane_out[m] = !EMAVFI_DEFORM_ABL_NO_FALLBACK && (base[m] & 1u) != 0;
            any_out[m] = __any(lane_out[m]);
#if EMAVFI_DEFORM_STAMPS
            cnt_out += any_out[m] ? 1 : 0;
            cnt_lanes += __popcll(__ballot(lane_out[m]));
#endif
            base[m] &= ~1u;
        }
        DEFORM_STAMP(ts_geom);

        // ---- software pipeline over the 2*KG (k-group, row) steps: the four corner pieces of step s+1 are in flight
        // while step s is blended and contracted
        auto gather = [&](int sidx, u32x4_t (&v)[4]) {
            const int kg = sidx >> 1, m = sidx & 1;   // k-group outer: one weight fragment set serves both rows
            const bool tail = kg == KG - 1;           // compile-time after unrolling
            // slot of this lane's piece in the staged pixel; pieces past the staged channels (zero weights) re-read the last slot
            const int slot = (2 * kg + h < C::SP) ? 2 * kg + h : C::SP - 1;
            lds_cchar_t *q = lds_r + base[m] + (unsigned)(slot * 16);
            constexpr int OFF[4] = {0, C::PSB, C::ROWB, C::ROWB + C::PSB};
            if (tail && TQ == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32x2_t t2 = lds_read8(q + OFF[c]);
                    v[c] = u32x4_t{t2[0], t2[1], 0u, 0u};
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = lds_read16(q + OFF[c]);
            }
            if (any_out[m]) {      // wave-uniform: some lane reaches past the window
                if (lane_out[m]) { // one divergent region per step
                    const unsigned pix = gp[m] & 0xffffffu, ddx = (gp[m] >> 24) & 1u, ddy = (gp[m] >> 25) & 1u;
                    const unsigned pc[4] = {pix, pix + ddx, pix + (ddy ? (unsigned)W : 0u), pix + (ddy ? (unsigned)W : 0u) + ddx};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const char *src = gx + (size_t)__umul24(pc[c], ps_bytes) + (unsigned)(kg * 32);
                        // last k-group: channels 64..71 come from the compact tail when the input is split; the h = 1
                        // half (channels 72..79: zero weights) reads the zero page - those channels of x may be unwritten
                        if (tail) src = h ? zeros : (tplane ? tplane + (size_t)__umul24(pc[c], tail_bytes) : src);
                        v[c] = to_f16_piece<TS>(*reinterpret_cast<const u32x4_t *>(src));
                    }
                }
            }
        };
        u32x4_t vb[2][4];
        gather(0, vb[0]);
#pragma unroll
        for (int sidx = 0; sidx < 2 * KG; ++sidx) {
            if (sidx + 1 < 2 * KG) gather(sidx + 1, vb[(sidx + 1) & 1]);
            const int kg = sidx >> 1, m = sidx & 1;
            if (m == 0 && !EMAVFI_DEFORM_ABL_NO_WLOADS) {  // fetch the next k-group's fragments (or the next tap's first) while this one is used
                if (kg + 1 < KG) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) wq[(kg + 1) & 1][n] = *reinterpret_cast<const f16x8 *>(wtap + ((kg + 1) * NF + n) * 1024);
                } else if (tap < 8) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) wq[1][n] = *reinterpret_cast<const f16x8 *>(wtap + C::WTAP + n * 1024);
                }
            }
            f16x8 xf;
            if (kg == KG - 1 && TQ == 2) xf = blend_corners<2>(vb[sidx & 1], w01[m], w23[m]);  // folded at compile time
            else xf = blend_corners<4>(vb[sidx & 1], w01[m], w23[m]);
#pragma unroll
            for (int n = 0; n < NF; ++n) mma_kg(acc[m][n], wq[kg & 1][n], xf);
        }
#if EMAVFI_DEFORM_STAMPS
        DEFORM_STAMP(ts_end);
        sum_geom += ts_geom - ts_tap;
        sum_steps += ts_end - ts_geom;
#endif
    }
    DEFORM_STAMP(ts_loop);

    // ---- epilogue (no activation: ema_vfi.py:136-138 chains the blocks directly)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!in_img[m]) continue;
        TS *op = reinterpret_cast<TS *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (p.cstore - n * 32 > 0) store_frag(op + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
    }
#if EMAVFI_DEFORM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DEFORM_STAMP(ts_done);
    // every DEFORM_STAMP_STRIDE-th workgroup records; one private row of 8 values per wave (no atomics)
    if (p.stamps && lane == 0 && blockIdx.x % DEFORM_STAMP_STRIDE == 0) {
        const unsigned row = (blockIdx.x / DEFORM_STAMP_STRIDE) * C::WAVES + wave;
        if (row < DEFORM_STAMP_ROWS) {
            unsigned long long *o = p.stamps + (size_t)row * 8;
            o[0] = ts_window - ts_begin; o[1] = ts_offconv - ts_window; o[2] = sum_geom; o[3] = sum_steps;
            o[4] = (ts_done - ts_loop) | (cnt_out << 32) | (cnt_lanes << 40); o[5] = ts_done - ts_begin; o[6] = 1;
            // prologue detail packed into o[7]: 16 bits each (units of 4 cycles): issue, landed - issued, convert, barrier wait
            auto q16 = [](unsigned long long v) { v >>= 2; return v > 0xffffull ? 0xffffull : v; };
            o[7] = q16(ts_issued - ts_begin) | (q16(ts_landed - ts_issued) << 16) | (q16(ts_converted - ts_landed) << 32) | (q16(ts_window - ts_converted) << 48);
        }
    }
#endif
}

template <typename TS, int TCOLS, int R, bool FUSE_OFF, int TQ> static int launch_deform_pack(const DeformParams &p, hipStream_t s)
{
    using C = PackCfg<TCOLS, R>;
    static std::once_flag once;   // the library is re-entrant: launchers may be called from several threads
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&deform_pack_kernel<TS, TCOLS, R, FUSE_OFF, TQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const long long nwg = (long long)((p.W + C::TCOLS - 1) / C::TCOLS) * ((p.H + C::TROWS - 1) / C::TROWS) * p.B;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    deform_pack_kernel<TS, TCOLS, R, FUSE_OFF, TQ><<<(unsigned)nwg, C::THREADS, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}

// tile width: 16 = two 4-wave workgroups per CU (one's prologue / epilogue beside the other's main loop): measured 3 %
// faster than 32 (one 8-wave workgroup per CU) at B=8 x 720p in both 16-bit types, at equal HBM traffic
#ifndef EMAVFI_DEFORM_TCOLS
#define EMAVFI_DEFORM_TCOLS 16
#endif
// the reference width (mid_channels 64 -> 67 channels, k-groups to 80): LDS-staged window of 72 channels
static inline bool deform16_lds_shape(int ck, int nf, int cin_real) { return ck == 80 && nf == 3 && cin_real <= 72; }

template <typename TS, bool FUSE_OFF> static int launch_deform_pack3(const DeformParams &p, hipStream_t s);

template <typename TS> static int launch_deform16(const DeformParams &p, hipStream_t s)
{
    if (p.pack3) {  // weights in the deform_pack3.inl layout (host: deform_pack3_shape)
        if (!deform_pack3_shape(p.ck, p.nf, p.cin_real, p.cout_real)) return -2;
        return p.off_w ? launch_deform_pack3<TS, true>(p, s) : launch_deform_pack3<TS, false>(p, s);
    }
    if (deform16_lds_shape(p.ck, p.nf, p.cin_real)) {
        constexpr int TC = EMAVFI_DEFORM_TCOLS;
        if (p.cin_real <= 68) {  // the reference width: 3 real channels in the last k-group
            if (p.off_w) return launch_deform_pack<TS, TC, 2, true, 2>(p, s);
            return launch_deform_pack<TS, TC, 2, false, 2>(p, s);
        }
        if (p.off_w) return launch_deform_pack<TS, TC, 2, true, 4>(p, s);
        return launch_deform_pack<TS, TC, 2, false, 4>(p, s);
    }
    if (p.off_w) return -1;  // the host only asks for fusion after deform16_can_fuse_offset_conv()
    return launch_deform_any<TS>(p, s);
}
