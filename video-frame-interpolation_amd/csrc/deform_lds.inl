// bf16 / f16 modulated deformable conv with the input window staged in LDS (the "tiled gather with LDS
// halo" of BASELINE.json).  Same operator and fragment scheme as deform.inl; what changes is
// where the bilinear taps are fetched from:
//
//   * one workgroup owns a 16x32 output tile; a wave owns RPW rows (RPW = 1: 1024 threads, 16 waves =
//     4 per SIMD, <= 128 VGPRs; RPW = 2: 512 threads, 8 waves) - the waves sharing a SIMD overlap one
//     wave's gather/blend VALU and LDS latency with another's MFMAs;
//   * the input window = tile + halo of 1 (3x3 taps) + R (offset reach) + 1 (bilinear) pixels
//     is brought into LDS ONCE by global->LDS DMA (global_load_lds_dwordx4, every lane with its
//     own source address, out-of-image pixels read a zero page): (19+2R) x (35+2R) pixels x CS
//     staged channels.  Only the CS = 72 channels that can be non-zero are staged (67 real, the
//     MFMA k-groups run to 80): 9 sixteen-byte slots per pixel - an ODD slot stride, so the
//     ds_read_b128 gathers of neighbouring pixels are bank-conflict free - and lanes whose
//     k-group piece would be channels 72..79 re-read slot 8: those channels only ever meet zero
//     weights.  R = 2: 23 x 39 x 144 B = 126 KiB; HBM/L2 sees each input pixel 1.75x instead of
//     ~36 scattered corner fetches thrashing the 32 KiB L1 (deform.inl measured 5.0 ms per
//     B=8 720p launch with the global gather);
//   * every lane gathers its own MFMA operand pieces with ds_read_b128; a tap whose four corners
//     do not all lie inside the window (|offset| > R) falls back to the global gather of
//     deform.inl for that lane - results are identical either way;
//   * packed weights never touch LDS: every wave loads its MFMA A fragments straight from the blob (L2/L1
//     resident) one k-group ahead, so after the window barrier the waves run decoupled (no per-tap barrier);
//   * with DeformParams::off_w set the kernel is the whole ModulatedDeformConvPack: it first runs the pack's
//     offset_conv on the staged window and keeps (dy, dx, mask) in registers; otherwise it reads them from
//     p.om one tap ahead.
#include "deform.inl"
#include <mutex>
#include <type_traits>

#ifndef EMAVFI_DEFORM_PKF16
#define EMAVFI_DEFORM_PKF16 1  // f16: blend on v_pk_fma_f16 (0 = fp32 blend on v_fma_mix_f32)
#endif
#ifndef EMAVFI_DEFORM_DOT2
#define EMAVFI_DEFORM_DOT2 1
#endif
#ifndef EMAVFI_DEFORM_SHARE_GEOMETRY
#define EMAVFI_DEFORM_SHARE_GEOMETRY 1
#endif
// A/B switches (timing experiments; the defaults are the product)
#ifndef EMAVFI_DEFORM_XCD_ORDER
#define EMAVFI_DEFORM_XCD_ORDER 1   // 0: plain row-major tile order
#endif
#ifndef EMAVFI_DEFORM_TRIM_TAIL
#define EMAVFI_DEFORM_TRIM_TAIL 1   // 0: blend all four dwords of the last k-group
#endif

template <int CK, int NF, int CS, int R, int RPW> struct DeformLdsCfg {
    static constexpr int WAVES = 16 / RPW, THREADS = 64 * WAVES;
    static constexpr int TROWS = 16, TCOLS = 32;
    static constexpr int TR = TROWS + 3 + 2 * R, TC = TCOLS + 3 + 2 * R;
    static constexpr int SP = CS * 2 / 16;    // 16-byte slots per staged pixel
    static constexpr int PSB = SP * 16;       // bytes per staged pixel
    static constexpr int KG = CK / 16;
    static constexpr int WTAP = KG * NF * 1024;
    static constexpr int WINST = KG * NF;
    static constexpr int NSLOT = TR * TC * SP;
    static constexpr int NINST = (NSLOT + 63) / 64;
    static constexpr int LDS_TILE = NINST * 1024;
    static constexpr int LDS_BYTES = LDS_TILE;  // the window only: weights go L2 -> registers
    static constexpr int KB = (KG % 5 == 0) ? 5 : ((KG % 3 == 0) ? 3 : ((KG % 2 == 0) ? 2 : 1));
    static_assert(CS % 8 == 0 && CS <= CK, "staged channels: whole 16-byte slots, at most CK");
    static_assert((SP & 1) == 1, "odd slot stride keeps neighbouring-pixel gathers conflict free");
    static_assert(LDS_BYTES <= 160 * 1024, "window + weights do not fit the 160 KiB LDS");
};

struct OmTap { float dy, dx, mk; };

// Diagnostic build only (-DEMAVFI_DEFORM_STAMPS=1; cdna_hip_programming.md section 7, in-kernel stamps): s_memtime at the
// seams of the kernel, per-wave segment sums written to DeformParams::stamps (a buffer the diagnostic build of
// emavfi_api.hip allocates), read back with emavfi_debug_deform_stamps().  Never part of the shipped library: no stamp
// executes there and the symbol does not exist.
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
#else
#define DEFORM_STAMP(var)
#endif

// LDS reads through explicit address-space-3 pointers (ds_read_b128 / ds_read_b64, never flat_load)
typedef __attribute__((address_space(3))) const char lds_cchar_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ uint4 lds_read16(lds_cchar_t *p)
{
    const u32x4_t t = *reinterpret_cast<__attribute__((address_space(3))) const u32x4_t *>(p);
    return make_uint4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ uint2 lds_read8(lds_cchar_t *p)
{
    const u32x2_t t = *reinterpret_cast<__attribute__((address_space(3))) const u32x2_t *>(p);
    return make_uint2(t[0], t[1]);
}

__device__ __forceinline__ OmTap load_om(const float *__restrict__ om, int tap, bool in_image)
{
    OmTap t;
    t.dy = om[2 * tap]; t.dx = om[2 * tap + 1]; t.mk = in_image ? om[18 + tap] : 0.0f;
    return t;
}

// TQ = dwords of the LAST k-group's 16-byte piece that can hold real channels (cin_real <= 64 + 2 * TQ): the reference
// width (67) has 3 real channels there, so only 2 of the piece's 4 dwords are gathered and blended (8-byte LDS reads,
// half the blend instructions of that k-group); the other channels of the k-group only ever meet zero weights.
template <typename T, int CK, int NF, int CS, int R, int RPW, bool FUSE_OFF, int TQ>
__global__ __launch_bounds__(64 * (16 / RPW)) void deform_lds_kernel(const DeformParams p)
{
    using C = DeformLdsCfg<CK, NF, CS, R, RPW>;
    using vec = typename DT<T>::vec;  // vec or f16x8
    // bf16 blends on the dot-product unit (corner weights rounded to bf16); f16 blends packed (v_pk_fma_f16, corner
    // weights rounded to f16) - or, with EMAVFI_DEFORM_PKF16=0, in fp32 on v_fma_mix_f32 with fp32 weights
    constexpr bool DOT2 = EMAVFI_DEFORM_DOT2 && std::is_same<T, bf16_t>::value;
    constexpr bool PKH = EMAVFI_DEFORM_PKF16 && std::is_same<T, half_t>::value;
    static_assert(sizeof(T) == 2, "the LDS-window kernel is for the 16-bit dtypes");
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((address_space(3))) const char lchar_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_x = smem;
    // LDS reads go through an explicit address-space-3 pointer: with a generic pointer hipcc merged the window read and
    // the global fallback of a tap's first gather into flat_load (slower, and it counts on both vmcnt and lgkmcnt)
    lchar_t *lds_r = (lchar_t *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    // XCD-aware tile order (placement only affects speed).  Workgroups are dealt round-robin over the 8 XCDs, so
    // blockIdx % 8 labels an XCD group; each group gets a contiguous run of tiles, and inside an image tiles are walked
    // in strips of SROWS tile rows, column by column: the ~32 tiles an XCD has in flight form a compact 2-D block whose
    // window halos (1.75x the tile on its own) are re-read from that XCD's L2 instead of from HBM.
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
    const unsigned ps_bytes = (unsigned)p.x_ps * 2u;
    const int ty0 = tile_y * C::TROWS - 1 - R, tx0 = tile_x * C::TCOLS - 1 - R;
    const char *gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;
    const char *zeros = (const char *)p.zeros;
    const unsigned tail_bytes = (unsigned)p.tail_ps * 2u;
    const char *tplane = p.x_tail ? (const char *)p.x_tail + (size_t)b * H * W * tail_bytes : nullptr;
    static_assert(CS == 72, "the split-input path assumes the tail starts at channel 64 = the last staged slot");

    DEFORM_STAMP(ts_begin);
#if EMAVFI_DEFORM_STAMPS
    unsigned long long sum_geom = 0, sum_steps = 0;
#endif
    // ---- DMA the input window (zero outside the image) and tap 0's weights ----
#pragma unroll
    for (int i = 0; i < (C::NINST + C::WAVES - 1) / C::WAVES; ++i) {
        const int j = i * C::WAVES + wave;
        if (j < C::NINST) {
            const int sl = j * 64 + lane;
            const int pix = sl / C::SP, pc = sl - pix * C::SP;
            const int ly = pix / C::TC, lx = pix - ly * C::TC;
            const int gy = ty0 + ly, gxx = tx0 + lx;
            const bool ok = sl < C::NSLOT && gy >= 0 && gy < H && gxx >= 0 && gxx < W;
            const char *src = ok ? gplane + (size_t)(gy * W + gxx) * ps_bytes + pc * 16 : zeros;
            if (tplane && ok && pc == C::SP - 1) src = tplane + (size_t)(gy * W + gxx) * tail_bytes;  // channels 64..71
            __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(lds_x + j * 1024), 16, 0, 0);
        }
    }
    // weights never touch LDS: every wave loads its MFMA A fragments straight from the packed blob (135 KiB,
    // L2/L1 resident, one k-group ahead) - no weight ring, no per-tap barrier, the waves run decoupled
    const char *wlane = (const char *)p.w + lane * 16;
    vec wq[2][NF];  // [kg & 1]; the next tap's first k-group arrives in wq[1] (KG is odd) and moves to wq[0]
    static_assert((C::KG & 1) == 1, "cross-tap prefetch slot assumes an odd k-group count");
#pragma unroll
    for (int n = 0; n < NF; ++n) wq[1][n] = *reinterpret_cast<const vec *>(wlane + n * 1024);

    f32x16 acc[RPW][NF];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];

    const int px_x = tile_x * C::TCOLS + r;
    int py_y[RPW];
    bool in_img[RPW];
    const float *om[RPW];
    OmTap nxt[RPW];
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
        py_y[m] = tile_y * C::TROWS + wave * RPW + m;
        in_img[m] = py_y[m] < H && px_x < W;
        om[m] = p.om + (((size_t)b * H + (in_img[m] ? py_y[m] : 0)) * W + (in_img[m] ? px_x : 0)) * 32;
        if (!FUSE_OFF) nxt[m] = load_om(om[m], 0, in_img[m]);
    }
    const char *gx = gplane + h * 16;
    // offset_conv weight fragments of the first two taps ride along with the window DMA
    const char *owl = (const char *)p.off_w + lane * 16;
    vec ow[FUSE_OFF ? 3 : 1][C::KG];
    if constexpr (FUSE_OFF) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kg = 0; kg < C::KG; ++kg) ow[t][kg] = *reinterpret_cast<const vec *>(owl + (t * C::KG + kg) * 1024);
    }
    __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
    DEFORM_STAMP(ts_window);

    // fused: this lane's half of its pixels' (dy, dx, mask) values stays in registers in accumulator layout:
    // channel c sits in half-lane (c >> 2) & 1, register (c & 3) + 4 * (c >> 3)
    f32x16 omr[FUSE_OFF ? RPW : 1];
    if constexpr (FUSE_OFF) {
        // ---- the pack's offset_conv (ema_vfi.py:41,56: 3x3, pad 1, cin -> 27) on the staged window ----
        // Plain (undeformed) taps: the B operand of lane (r, h) is a 16-byte piece of window pixel
        // (row + i + R, r + j + R), read as it lies; out-of-image pixels were zero-filled by the DMA = the
        // conv's zero padding.  Weight fragments come from L2 two taps ahead.  Same tap/k-group order and the
        // same epilogue as the stand-alone conv3x3 EPI_OM layer -> bit-identical om.
        f32x16 (&oacc)[RPW] = omr;
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[m][i] = p.off_bias[acc_channel(i, h)];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 7) {  // two taps ahead (L2 latency is ~3 taps of MFMA time for one wave)
#pragma unroll
                for (int kg = 0; kg < C::KG; ++kg) ow[(tap + 2) % 3][kg] = *reinterpret_cast<const vec *>(owl + ((tap + 2) * C::KG + kg) * 1024);
            }
            const int i = tap / 3, j = tap - 3 * i;
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                lchar_t *xp = lds_r + (((wave * RPW + m) + i + R) * C::TC + (r + j + R)) * C::PSB;
#pragma unroll
                for (int kg = 0; kg < C::KG; ++kg) {
                    const int slot = (2 * kg + h < C::SP) ? 2 * kg + h : C::SP - 1;
                    const vec xv = *reinterpret_cast<__attribute__((address_space(3))) const vec *>(xp + slot * 16);
                    mma_kg(oacc[m], ow[tap % 3][kg], xv);
                }
            }
        }
        // mask = sigmoid(third chunk), ema_vfi.py:59 (channels 18..26 after the pack-time routing); pixels outside
        // the image get mask 0.  Nothing goes through memory: the main loop picks (dy, dx, mask) of a tap out of
        // these registers with one v_permlane32_swap per value.
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = acc_channel(i, h);
                const float v = oacc[m][i];
                oacc[m][i] = (c >= 18 && c < 27) ? (in_img[m] ? 1.0f / (1.0f + expf(-v)) : 0.0f) : v;
            }
    }

    DEFORM_STAMP(ts_offconv);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        DEFORM_STAMP(ts_tap);
#pragma unroll
        for (int n = 0; n < NF; ++n) wq[0][n] = wq[1][n];
        const char *wtap = wlane + (size_t)tap * C::WTAP;
        OmTap now[RPW];
        if constexpr (FUSE_OFF) {
            // value of channel c of this lane's pixel, wherever its half-lane keeps it: a = b = reg, swap a.hi <-> b.lo
            auto pick = [&](const f32x16 &v, auto cc) {
                constexpr int c = decltype(cc)::value;
                const unsigned u = __float_as_uint(v[(c & 3) + 4 * (c >> 3)]);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                return __uint_as_float(((c >> 2) & 1) ? sw[1] : sw[0]);
            };
            auto take = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
#pragma unroll
                for (int m = 0; m < RPW; ++m) {
                    now[m].dy = pick(omr[m], std::integral_constant<int, 2 * k>{});
                    now[m].dx = pick(omr[m], std::integral_constant<int, 2 * k + 1>{});
                    now[m].mk = pick(omr[m], std::integral_constant<int, 18 + k>{});
                }
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
        } else {
#pragma unroll
            for (int m = 0; m < RPW; ++m) now[m] = nxt[m];
            if (tap < 8) {
#pragma unroll
                for (int m = 0; m < RPW; ++m) nxt[m] = load_om(om[m], tap + 1, in_img[m]);
            }
        }
        // sampling geometry of this lane's RPW pixels for this tap
        SampleTap st[RPW];
        unsigned lo[RPW][4];
        bool inside[RPW], all_inside[RPW];
        BlendW bw[DOT2 ? RPW : 1];
        BlendWh bh[PKH ? RPW : 1];
        // one row's geometry: positions, corner weights, global offsets (st), window-local LDS offsets (lw),
        // whether all four corners lie inside the staged window
        auto geometry = [&](const OmTap &o, int y, SampleTap &t, unsigned (&lw)[4]) -> bool {
            int yc0, yc1, xc0, xc1;
            t = sample_tap_vals(o.dy, o.dx, o.mk, tap, y, px_x, H, W, ps_bytes, &yc0, &yc1, &xc0, &xc1);
            // window-local corner offsets, clamped into the window so every lane's LDS read is in
            // bounds; lanes that are not `inside` overwrite what they read with the global gather
            const int ly0 = min(max(yc0 - ty0, 0), C::TR - 1), ly1 = min(max(yc1 - ty0, 0), C::TR - 1);
            const int lx0 = min(max(xc0 - tx0, 0), C::TC - 1), lx1 = min(max(xc1 - tx0, 0), C::TC - 1);
            const unsigned q0 = __umul24((unsigned)ly0, (unsigned)C::TC), q1 = __umul24((unsigned)ly1, (unsigned)C::TC);
            lw[0] = __umul24(q0 + lx0, (unsigned)C::PSB); lw[1] = __umul24(q0 + lx1, (unsigned)C::PSB);
            lw[2] = __umul24(q1 + lx0, (unsigned)C::PSB); lw[3] = __umul24(q1 + lx1, (unsigned)C::PSB);
            return yc0 >= ty0 && yc1 <= ty0 + C::TR - 1 && xc0 >= tx0 && xc1 <= tx0 + C::TC - 1;
        };
        if constexpr (RPW == 2 && EMAVFI_DEFORM_SHARE_GEOMETRY) {
            // The two half-lanes of a pixel column need the same numbers for both rows: half h computes row h
            // only, and one v_permlane32_swap per dword hands both halves both rows (swap(x, x) = {row 0's, row 1's}).
            OmTap oh;
            oh.dy = h ? now[1].dy : now[0].dy; oh.dx = h ? now[1].dx : now[0].dx; oh.mk = h ? now[1].mk : now[0].mk;
            SampleTap th;
            unsigned lh[4];
            const bool ih = geometry(oh, py_y[0] + h, th, lh);
            auto both = [&](unsigned x, unsigned &r0, unsigned &r1) {
                const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                r0 = sw[0]; r1 = sw[1];
            };
            unsigned i0, i1;
            both(ih ? 1u : 0u, i0, i1);
            inside[0] = i0 != 0; inside[1] = i1 != 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                both(lh[c], lo[0][c], lo[1][c]);
                both(th.o[c], st[0].o[c], st[1].o[c]);
            }
            if constexpr (DOT2) {
                // corner weights travel as two packed bf16 pairs
                typedef __attribute__((ext_vector_type(2))) __bf16 pair_t;
                unsigned p01[2], p23[2];
                both(__builtin_bit_cast(unsigned, pair_t{(bf16_t)th.w[0], (bf16_t)th.w[1]}), p01[0], p01[1]);
                both(__builtin_bit_cast(unsigned, pair_t{(bf16_t)th.w[2], (bf16_t)th.w[3]}), p23[0], p23[1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bw[m].lo[0] = p01[m] & 0xffffu; bw[m].hi[0] = p01[m] << 16;
                    bw[m].lo[1] = p01[m] >> 16;     bw[m].hi[1] = p01[m] & 0xffff0000u;
                    bw[m].lo[2] = p23[m] & 0xffffu; bw[m].hi[2] = p23[m] << 16;
                    bw[m].lo[3] = p23[m] >> 16;     bw[m].hi[3] = p23[m] & 0xffff0000u;
                }
            } else if constexpr (PKH) {
                // corner weights travel as two packed f16 pairs, then each is doubled to (w, w)
                unsigned p01[2], p23[2];
                both(__builtin_bit_cast(unsigned, f16x2_t{(half_t)th.w[0], (half_t)th.w[1]}), p01[0], p01[1]);
                both(__builtin_bit_cast(unsigned, f16x2_t{(half_t)th.w[2], (half_t)th.w[3]}), p23[0], p23[1]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f16x2_t a = __builtin_bit_cast(f16x2_t, p01[m]), b2 = __builtin_bit_cast(f16x2_t, p23[m]);
                    bh[m].pk[0] = f16x2_t{a[0], a[0]}; bh[m].pk[1] = f16x2_t{a[1], a[1]};
                    bh[m].pk[2] = f16x2_t{b2[0], b2[0]}; bh[m].pk[3] = f16x2_t{b2[1], b2[1]};
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned w0, w1;
                    both(__float_as_uint(th.w[c]), w0, w1);
                    st[0].w[c] = __uint_as_float(w0); st[1].w[c] = __uint_as_float(w1);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) all_inside[m] = __all(inside[m]);
        } else {
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                inside[m] = geometry(now[m], py_y[m], st[m], lo[m]);
                all_inside[m] = __all(inside[m]);
                if constexpr (DOT2) bw[m] = blend_weights_bf16(st[m].w);
                if constexpr (PKH) bh[m] = blend_weights_f16(st[m].w);
            }
        }
        // software pipeline over the RPW*KG (row, k-group) steps: the four corner pieces of step s+1
        // are in flight while step s is blended and contracted
        auto gather = [&](int sidx, uint4 (&v)[4]) {
            const int kg = sidx / RPW, m = sidx - kg * RPW;   // k-group outer: one weight fragment set serves RPW rows
            // slot of this lane's piece in the staged pixel; pieces past the staged channels
            // (zero weights) re-read the last slot
            const int slot = (2 * kg + h < C::SP) ? 2 * kg + h : C::SP - 1;
            const bool tail = kg == C::KG - 1;  // compile-time after unrolling
            if (tail && TQ == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint2 t2 = lds_read8(lds_r + lo[m][c] + (unsigned)(slot * 16));
                    v[c] = make_uint4(t2.x, t2.y, 0u, 0u);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = lds_read16(lds_r + lo[m][c] + (unsigned)(slot * 16));
            }
            if (!all_inside[m]) {   // wave-uniform: some lane reaches past the window
                if (!inside[m]) {   // one divergent region per step
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const char *src = gx + st[m].o[c] + (unsigned)(kg * 32);
                        // last k-group: channels 64..71 come from the compact tail when the input is split; the h = 1
                        // half (channels 72..79: zero weights) reads the zero page - those channels of x may be unwritten
                        if (tail) src = h ? zeros : (tplane ? tplane + (st[m].o[c] / ps_bytes) * tail_bytes : src);
                        v[c] = *reinterpret_cast<const uint4 *>(src);
                    }
                }
            }
        };
        DEFORM_STAMP(ts_geom);
        uint4 vb[2][4];
        gather(0, vb[0]);
#pragma unroll
        for (int sidx = 0; sidx < RPW * C::KG; ++sidx) {
            if (sidx + 1 < RPW * C::KG) gather(sidx + 1, vb[(sidx + 1) & 1]);
            const int kg = sidx / RPW, m = sidx - kg * RPW;
            if (m == 0) {  // fetch the next k-group's fragments (or the next tap's first) while this one is used
                if (kg + 1 < C::KG) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) wq[(kg + 1) & 1][n] = *reinterpret_cast<const vec *>(wtap + ((kg + 1) * NF + n) * 1024);
                } else if (tap < 8) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) wq[1][n] = *reinterpret_cast<const vec *>(wtap + C::WTAP + n * 1024);
                }
            }
            vec xf;
            if (kg == C::KG - 1 && TQ == 2) {  // folded at compile time
                if constexpr (DOT2) xf = blend4_dot2<2>(vb[sidx & 1], bw[m]);
                else if constexpr (PKH) xf = blend4_pk<2>(vb[sidx & 1], bh[m]);
                else xf = blend4(vb[sidx & 1], st[m].w, T{});
            } else {
                if constexpr (DOT2) xf = blend4_dot2<4>(vb[sidx & 1], bw[m]);
                else if constexpr (PKH) xf = blend4_pk<4>(vb[sidx & 1], bh[m]);
                else xf = blend4(vb[sidx & 1], st[m].w, T{});
            }
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                mma_kg(acc[m][n], wq[kg & 1][n], xf);
            }
        }
        // retire the tap's accumulator chains before the next tap's geometry code (common.h)
#pragma unroll
        for (int m = 0; m < RPW; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) mfma_retire(acc[m][n]);
#if EMAVFI_DEFORM_STAMPS
        DEFORM_STAMP(ts_end);
        sum_geom += ts_geom - ts_tap;
        sum_steps += ts_end - ts_geom;
#endif
    }
    DEFORM_STAMP(ts_loop);

#pragma unroll
    for (int m = 0; m < RPW; ++m) {
        if (!in_img[m]) continue;
        T *op = reinterpret_cast<T *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
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
            o[4] = ts_done - ts_loop; o[5] = ts_done - ts_begin; o[6] = 1; o[7] = ts_begin;
        }
    }
#endif
}

template <typename T, int CK, int NF, int CS, int R, int RPW, bool FUSE_OFF, int TQ> static int launch_deform_lds(const DeformParams &p, hipStream_t s)
{
    using C = DeformLdsCfg<CK, NF, CS, R, RPW>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(&deform_lds_kernel<T, CK, NF, CS, R, RPW, FUSE_OFF, TQ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const long long nwg = (long long)((p.W + C::TCOLS - 1) / C::TCOLS) * ((p.H + C::TROWS - 1) / C::TROWS) * p.B;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    deform_lds_kernel<T, CK, NF, CS, R, RPW, FUSE_OFF, TQ><<<(unsigned)nwg, C::THREADS, C::LDS_BYTES, s>>>(p);
    return (int)hipGetLastError();
}

#ifndef EMAVFI_DEFORM_RPW
#define EMAVFI_DEFORM_RPW 2  // rows per wave; 1 (16 waves, <=128 VGPRs) spills and is 7x slower
#endif
// the reference width (mid_channels 64 -> 67 channels, k-groups to 80): LDS-staged window of 72 channels
static inline bool deform16_lds_shape(int ck, int nf, int cin_real) { return ck == 80 && nf == 3 && cin_real <= 72; }

template <typename T> static int launch_deform16(const DeformParams &p, hipStream_t s)
{
    if (deform16_lds_shape(p.ck, p.nf, p.cin_real)) {
        if (EMAVFI_DEFORM_TRIM_TAIL && p.cin_real <= 68) {  // the reference width: 3 real channels in the last k-group
            if (p.off_w) return launch_deform_lds<T, 80, 3, 72, 2, EMAVFI_DEFORM_RPW, true, 2>(p, s);
            return launch_deform_lds<T, 80, 3, 72, 2, EMAVFI_DEFORM_RPW, false, 2>(p, s);
        }
        if (p.off_w) return launch_deform_lds<T, 80, 3, 72, 2, EMAVFI_DEFORM_RPW, true, 4>(p, s);
        return launch_deform_lds<T, 80, 3, 72, 2, EMAVFI_DEFORM_RPW, false, 4>(p, s);
    }
    if (p.off_w) return -1;  // the host only asks for fusion after deform16_can_fuse_offset_conv()
    return launch_deform_any<T>(p, s);
}
