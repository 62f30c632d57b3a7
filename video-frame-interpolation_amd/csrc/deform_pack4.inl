// ModulatedDeformConvPack (ema_vfi.py:23-60) as ONE launch - round 4: deform_pack3.inl's arithmetic, operation for operation (the two
// are bit-identical: tests/test_gpu_parity.py::test_pack4_equals_pack3), in a launch structure that pairs the phases of the two waves
// of a SIMD BY CONSTRUCTION.
//
// What round 4 measured (profiles/r04_pack_one_vs_two_workgroups.txt).  A tile is five phases in a row, each
// loading ONE resource: P window DMA (the wave waits on the vector-memory queue), O offset_conv (matrix pipe + LDS), G geometry of
// nine taps (plain VALU), T nine taps of gather + blend + MFMA (packed-f16 VALU issue: v_pk_fma_f16 runs at 4.4 cycles per SIMD
// whatever the number of waves, and 1 152 of them per wave and tile), S stores.  ONE wave per SIMD (one workgroup per CU) needs
// 31 300 cycles per tile: P 6 100, O 4 500, G 3 600, T 13 800, S 2 200.  TWO waves per SIMD (two independent workgroups per CU, the
// round-3 product) need 43 600 EACH - 21 800 per tile, a factor 1.28 from the second wave, because the pairing is random: 45 % of a
// wave's T phase runs beside the other wave's T phase (both want the same VALU port and matrix pipe) while P beside P leaves the
// SIMD idle.  A persistent pair of identical workgroups falls into step (round 3: slower still).
//
// Here: ONE workgroup of eight waves per CU, persistent.  Waves 0-3 (half A) and 4-7 (half B) each own a window and walk their own
// tiles through the SAME loop
//        S P | O G | T1 | T2        (stores of the previous tile, window DMA of this one | offset_conv, geometry | taps 0-4 | taps 5-8 + fix-up)
// with a workgroup barrier behind every slot, and half B enters the loop TWO SLOTS LATE (two extra barriers in front; half A runs
// two behind its last tile).  So at any time one wave of a SIMD is in a T slot and its partner in S P or O G:
//        A:  SP  OG  T1  T2 | SP  OG  T1  T2 | ...
//        B:  -   -   SP  OG | T1  T2  SP  OG | ...
// The barriers are ordinary s_barrier: every wave executes the same number of them whatever its tiles (invalid tiles skip the work,
// never the barrier), so there is no spin-wait and nothing to hang on.  Alone-wave slot times: T1 = T2 = 6 900, S P = 8 300,
// O G = 8 100 - balanced to 16 %.  One code copy serves both halves (the loop body is straight-line, so register liveness stays what
// it is in deform_pack3: the accumulators die at S, the offset_conv's registers at G).
#pragma once
#include "deform_pack3.inl"

struct Pack4 {
    static constexpr int THREADS = 512, HALF_WAVES = 4;
    static constexpr int WIN_BYTES = Pack3::WIN_BYTES, W3_OFF = 2 * WIN_BYTES, LDS_BYTES = W3_OFF + Pack3::W3_BYTES;   // 156 960 B: one workgroup per CU
    static constexpr int T1_TAPS = 5;                                                                                   // taps [0, 5) | [5, 9)
    static_assert(LDS_BYTES <= 160 * 1024, "two windows + the third fragment's table in one CU's LDS");
};

template <typename TS>
__global__ __launch_bounds__(512, 2) void deform_pack4_kernel(const DeformParams p, const int n_iter)
{
    using C = Pack3;
    using D = Pack4;
    constexpr int R = C::R;
    static_assert(sizeof(TS) == 2, "16-bit storage types only");
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
    const int half = wave8 >> 2, wave = wave8 & 3;      // wave: index inside the half (deform_pack3's wave)
    const int r = lane & 31, h = lane >> 5;
    const int H = p.H, W = p.W;
    const unsigned lane16 = (unsigned)lane * 16u;
    char *win = smem + half * D::WIN_BYTES;             // this half's window
    lds_cchar_t *lds_r = (lds_cchar_t *)win;
    lds_cchar_t *lds_w3 = (lds_cchar_t *)smem;          // third-fragment table: addressed from the LDS base (w3lane carries D::W3_OFF)

    const int ntx = (W + C::TCOLS - 1) / C::TCOLS, nty = (H + C::TROWS - 1) / C::TROWS, nt = ntx * nty;
    const int ntiles = nt * p.B, npairs = (ntiles + 1) >> 1;
    const unsigned ps_bytes = (unsigned)p.x_ps * 2u, tail_bytes = (unsigned)p.tail_ps * 2u;
    const char *zeros = (const char *)p.zeros;
    const char *wbase_g = (const char *)p.w;
    const char *owbase_g = (const char *)p.off_w;
    const char *wtl = wbase_g + C::DCN_TAIL;

    // ---- tile-independent lane geometry (deform_pack3.inl)
    const bool dact = wave < 3 ? lane < C::SEG_PX * C::SP : lane < C::LAST_PX * C::SP;
    const int dp = lane / C::SP, dpc = lane - dp * C::SP;
    const bool g2 = (r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28;
    const int fr_row = g2 ? 1 : 0;
    const int fr_col = g2 ? (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16)) : (r < 4 ? r : (r < 16 ? r - 8 : r - 12));
    int wrow[2];
    unsigned xbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        wrow[m] = (wave * 2 + m) * 2 + fr_row;
        xbase[m] = (unsigned)(((wrow[m] + R) * C::TC + fr_col + R) * C::PSB + h * 16);
    }
    const unsigned w3lane = (unsigned)(D::W3_OFF + (min(r, 3) * 2 + h) * 16);
    constexpr int OFF[4] = {0, C::PSB, C::ROWB, C::ROWB + C::PSB};

    // ---- the third fragment's A operands (4 608 B), once per workgroup: half A's four waves, as deform_pack3's do per tile.
    // Half A's first window barrier (vmcnt(0) in front) publishes it; half B first reads it many barriers later.
    if (half == 0) {
        const char *w3g = wbase_g + C::DCN_W3;
        __builtin_amdgcn_global_load_lds((gptr_t *)(w3g + wave * 1024 + lane16), (lptr_t *)(smem + D::W3_OFF + wave * 1024), 16, 0, 0);
        if (wave == 0 && lane < 32)
            __builtin_amdgcn_global_load_lds((gptr_t *)(w3g + 4096 + lane16), (lptr_t *)(smem + D::W3_OFF + 4096), 16, 0, 0);
    }

    // ---- per-tile state (the tile whose phases this wave is running)
    bool valid = false;
    int tile_x = 0, tile_y = 0, b = 0, ty0 = 0, tx0 = 0, px_x = 0;
    int py_y[2] = {0, 0};
    bool in_img[2] = {false, false};
    bool my_in = false;
    float fy_base = 0.0f, fx_base = 0.0f;
    const char *gplane = nullptr, *tplane = nullptr;
    const float fy_max = (float)(H + 1), fx_max = (float)(W + 1);

    auto set_tile = [&](int it) {
        // pair q of this workgroup in iteration `it`, mapped XCD-aware like deform_pack3's workgroups (blocks g and g + 8 share an XCD:
        // each XCD walks its own contiguous run of pairs); the two tiles of a pair are neighbours in the 4-row strip order
        const int q = it * (int)gridDim.x + (int)blockIdx.x;
        valid = false;
        if (q < npairs) {
            constexpr int SROWS = 4;
            const int grp = q & 7, kk = q >> 3, qq = npairs >> 3, rr = npairs & 7;
            const int wgp = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
            const int wg = 2 * wgp + half;
            if (wg < ntiles) {
                valid = true;
                b = wg / nt;
                const int t = wg - b * nt, strip = t / (SROWS * ntx), tt = t - strip * SROWS * ntx;
                const int rows = min(SROWS, nty - strip * SROWS);
                tile_x = tt / rows;
                tile_y = strip * SROWS + (tt - tile_x * rows);
            }
        }
        ty0 = tile_y * C::TROWS - 1 - R; tx0 = tile_x * C::TCOLS - 1 - R;
        gplane = (const char *)p.x + (size_t)b * H * W * ps_bytes;
        tplane = p.x_tail ? (const char *)p.x_tail + (size_t)b * H * W * tail_bytes : nullptr;
        px_x = tile_x * C::TCOLS + fr_col;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            py_y[m] = tile_y * C::TROWS + wrow[m];
            in_img[m] = valid && py_y[m] < H && px_x < W;
        }
        const int my_y = h ? py_y[1] : py_y[0];
        my_in = h ? in_img[1] : in_img[0];
        fy_base = (float)(my_y - 1); fx_base = (float)(px_x - 1);
    };

    // ---- state that crosses slots
    f32x16 acc[2][3];                               // G -> T1 -> T2 -> S
    typedef unsigned u32x16_t __attribute__((ext_vector_type(9)));
    u32x16_t gm0 = {}, gm1 = {}, gm2 = {};          // G -> T2 (+ fix-up)
    unsigned lane_fb = 0, fb_taps = 0;
    f16x8 wq[2][2];                                 // the tap loop's software pipeline (crosses the T1 | T2 barrier)
    f16x8 xf_prev = {}, w3_prev = {}, w3_cur = {};
    // this lane's byte offset inside a 1-KiB weight fragment, made OPAQUE once per iteration: hipcc otherwise hoists the ~40 64-bit
    // fragment addresses (base + k KiB + lane16) in front of the tile loop and spills them (round 3 met the same with the lane index)
    unsigned l16 = lane16;

    // ================================================================ phase P: window DMA of the current tile
    auto phase_P = [&]() {
        if (!valid) return;
        const int dgx = tx0 + wave * C::SEG_PX + dp;
        const bool dcol = (unsigned)dgx < (unsigned)W;
        const long long pix0 = (long long)ty0 * W + dgx;
        const bool from_tail = tplane != nullptr && dpc == C::SP - 1;
        const char *src = from_tail ? tplane + pix0 * (long long)tail_bytes : gplane + pix0 * (long long)ps_bytes + dpc * 16;
        const unsigned inc = (unsigned)W * (from_tail ? tail_bytes : ps_bytes);
        if (dact) {
#pragma unroll
            for (int ly = 0; ly < C::TR; ++ly) {
                const bool ok = dcol && (unsigned)(ty0 + ly) < (unsigned)H;
                const char *s = ok ? src : zeros;
                __builtin_amdgcn_global_load_lds((gptr_t *)s, (lptr_t *)(win + ly * C::ROWB + wave * C::SEG_BYTES), 16, 0, 0);
                src += inc;
            }
        }
        if constexpr (std::is_same<TS, bf16_t>::value) {
            if (!p.in_f16) {   // bf16 -> f16 in place: every wave converts exactly the pieces its own DMA instructions fetched
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (dact) {
#pragma unroll
                    for (int ly = 0; ly < C::TR; ++ly) {
                        lds_char_t *q = (lds_char_t *)win + ly * C::ROWB + wave * C::SEG_BYTES + lane16;
                        const u32x4_t v = to_f16_piece<TS>(lds_read16(q));
                        *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(q) = v;
                    }
                }
            }
        }
    };

    // ================================================================ phase O G: offset_conv, geometry + tail of all nine taps
    auto phase_OG = [&]() {
        if (!valid) return;
        f16x8 ow[3][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) ow[t][kg] = *reinterpret_cast<const f16x8 *>(owbase_g + (t * 4 + kg) * 1024 + l16);
        f32x16 omr[2];
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
            if constexpr (tap < 7) {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) ow[(tap + 2) % 3][kg] = *reinterpret_cast<const f16x8 *>(owbase_g + ((tap + 2) * 4 + kg) * 1024 + l16);
            }
            if constexpr (tap < 8) load_x(std::integral_constant<int, tap + 1>{}, xq[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) mma_kg(omr[m], ow[tap % 3][kg], __builtin_bit_cast(f16x8, xq[tap & 1][m][kg]));
            __builtin_amdgcn_sched_barrier(0);
        };
        off_tap(std::integral_constant<int, 0>{}); off_tap(std::integral_constant<int, 1>{}); off_tap(std::integral_constant<int, 2>{});
        off_tap(std::integral_constant<int, 3>{}); off_tap(std::integral_constant<int, 4>{}); off_tap(std::integral_constant<int, 5>{});
        off_tap(std::integral_constant<int, 6>{}); off_tap(std::integral_constant<int, 7>{}); off_tap(std::integral_constant<int, 8>{});
        {
            f16x8 ot[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) ot[j] = *reinterpret_cast<const f16x8 *>(owbase_g + C::OFF_TAIL + j * 1024 + l16);
            u32x2_t ta[2][3][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned tb = xbase[m] - (unsigned)(h * 16) + 128u;
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
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = acc_channel(i, h);
                const float v = omr[m][i];
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
                omr[m][i] = (c >= 18 && c < 27) ? sg : v;
            }

        // ---- DCN accumulators; the first weight fragments of the tap loop
#pragma unroll
        for (int n = 0; n < 2; ++n) wq[0][n] = *reinterpret_cast<const f16x8 *>(wbase_g + n * 1024 + l16);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = p.bias[n * 32 + acc_channel(i, h)];

        // ---- geometry of all nine taps + the blended tail of this half-lane's own pixel (deform_pack3.inl, item 3)
        lane_fb = 0; fb_taps = 0;
        gm0 = u32x16_t{}; gm1 = u32x16_t{}; gm2 = u32x16_t{};   // (fresh vectors: the previous tile's are dead here - keeps them out of the offset_conv's live set)
        unsigned tl[12][2];
#pragma unroll
        for (int t = 9; t < 12; ++t) tl[t][0] = tl[t][1] = 0u;
        auto geom_tap = [&](auto tc) {
            constexpr int tap = decltype(tc)::value, ti = tap / 3, tj = tap - 3 * ti;
            OmTap o;
            auto pick = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int reg = (c & 3) + 4 * (c >> 3);
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(omr[0][reg]), __float_as_uint(omr[1][reg]), false, false);
                return __uint_as_float(((c >> 2) & 1) ? sw[1] : sw[0]);
            };
            o.dy = pick(std::integral_constant<int, 2 * tap>{});
            o.dx = pick(std::integral_constant<int, 2 * tap + 1>{});
            o.mk = pick(std::integral_constant<int, 18 + tap>{});
            if (!my_in) o.mk = 0.0f;
            const float py = fminf(fmaxf((fy_base + (float)ti) + o.dy, -2.0f), fy_max);
            const float px = fminf(fmaxf((fx_base + (float)tj) + o.dx, -2.0f), fx_max);
            const float fy = floorf(py), fx = floorf(px);
            const int hl = (int)fy, wl = (int)fx;
            const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
            const float w4[4] = {o.mk * (uh * uw), o.mk * (uh * lw), o.mk * (lh * uw), o.mk * (lh * lw)};
            const int ly0 = hl - ty0, lx0 = wl - tx0;
            const bool inside = (unsigned)ly0 <= (unsigned)(C::TR - 2) && (unsigned)lx0 <= (unsigned)(C::TC - 2);
            const bool need_fb = !inside && my_in;
            const unsigned mybase = __umul24((unsigned)min(max(ly0, 0), C::TR - 2), (unsigned)C::ROWB) +
                                    __umul24((unsigned)min(max(lx0, 0), C::TC - 2), (unsigned)C::PSB);
            const unsigned keep = need_fb ? 0u : 0xffffffffu;
            const unsigned w01h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[0], (half_t)w4[1]}) & keep;
            const unsigned w23h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[2], (half_t)w4[3]}) & keep;
            gm0[tap] = (mybase & keep) | (__float_as_uint(py) & ~keep);
            gm1[tap] = w01h | (__float_as_uint(px) & ~keep);
            gm2[tap] = w23h | (__float_as_uint(o.mk) & ~keep);
            lane_fb |= ~keep & (1u << tap);
            fb_taps |= __any(need_fb) ? 1u << tap : 0u;
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

        auto tail_mma = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            f16x8 wt[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) wt[n] = *reinterpret_cast<const f16x8 *>(wtl + (j * 3 + n) * 1024 + l16);
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
                for (int n = 0; n < 3; ++n) mma_kg(acc[m][n], wt[n], xf);
            }
        };
        tail_mma(std::integral_constant<int, 0>{}); tail_mma(std::integral_constant<int, 1>{}); tail_mma(std::integral_constant<int, 2>{});

        // the tap loop's pipeline starts empty
        xf_prev = f16x8{}; w3_prev = f16x8{}; w3_cur = f16x8{};
#pragma unroll
        for (int n = 0; n < 2; ++n) wq[1][n] = f16x8{};
    };

    // ================================================================ phase T: taps [t0, t1) of gather + blend + MFMA (deform_pack3.inl, item 4)
    auto phase_T = [&](const int t0, const int t1) {
        if (!valid) return;
#pragma unroll 1
        for (int tap = t0; tap < t1; ++tap) {
            const char *wtap = wbase_g + (size_t)tap * C::DCN_TAP;
            unsigned g0 = gm0[tap], g1 = gm1[tap], g2v = gm2[tap];
            if (__builtin_expect((fb_taps >> tap) & 1u, 0)) {
                if ((lane_fb >> tap) & 1u) { g0 = 0u; g1 = 0u; g2v = 0u; }
            }
            unsigned base[2], w01[2], w23[2];
            {
                auto both = [&](unsigned x, unsigned (&out)[2]) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                    out[0] = sw[0]; out[1] = sw[1];
                };
                both(g1, w01); both(g2v, w23);
                const auto sw = __builtin_amdgcn_permlane32_swap(g0, g0 + 16u, false, false);
                base[0] = sw[0]; base[1] = sw[1];
            }
            const unsigned w3a = w3lane + (unsigned)(tap * C::W3_TAP);
            auto gather = [&](int s, unsigned (&d)[4][4]) {
                const int kg = s >> 1, m = s & 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32x4_t v = lds_read16(lds_r + base[m] + (unsigned)(kg * 32 + OFF[c]));
                    d[c][0] = v[0]; d[c][1] = v[1]; d[c][2] = v[2]; d[c][3] = v[3];
                }
            };
            unsigned vb[2][4][4];
            gather(0, vb[0]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int kg = s >> 1, m = s & 1;
                const int pm = (s + 7) & 1;
                const int pkg = ((s + 7) & 7) >> 1;
                if (s + 1 < 8) gather(s + 1, vb[(s + 1) & 1]);
                f16x8 w3n;
                if (m == 0) w3n = __builtin_bit_cast(f16x8, lds_read16(lds_w3 + w3a + (unsigned)(kg * 128)));
                __builtin_amdgcn_sched_barrier(0);
                f16x2_t a[4];
                const unsigned (&d)[4][4] = vb[s & 1];
                const unsigned wa = w01[m], wb = w23[m];
                mma_kg(acc[pm][0], wq[pkg & 1][0], xf_prev);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = __builtin_bit_cast(f16x2_t, d[0][q]) * bcast_half<0>(wa);
                a[0] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[1][0]), bcast_half<1>(wa), a[0]);
                PACK3_PIN(a);
                mma_kg(acc[pm][1], wq[pkg & 1][1], xf_prev);
#pragma unroll
                for (int q = 1; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[1][q]), bcast_half<1>(wa), a[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[2][q]), bcast_half<0>(wb), a[q]);
                PACK3_PIN(a);
                mma_kg(acc[pm][2], w3_prev, xf_prev);
#pragma unroll
                for (int q = 2; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[2][q]), bcast_half<0>(wb), a[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = __builtin_elementwise_fma(__builtin_bit_cast(f16x2_t, d[3][q]), bcast_half<1>(wb), a[q]);
                if (m == 0) {
                    if (kg + 1 < 4) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) wq[(kg + 1) & 1][n] = *reinterpret_cast<const f16x8 *>(wtap + ((kg + 1) * 2 + n) * 1024 + l16);
                    } else if (tap < 8) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) wq[0][n] = *reinterpret_cast<const f16x8 *>(wtap + C::DCN_TAP + n * 1024 + l16);
                    }
                    w3_cur = w3n;
                }
                PACK3_PIN(a);
                xf_prev = f16x8{a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], a[3][0], a[3][1]};
                w3_prev = w3_cur;
            }
        }
    };

    // ================================================================ the end of T2: the carried step, then the fix-up loop (deform_pack3.inl, item 6)
    auto phase_T_finish = [&]() {
        if (!valid) return;
        mma_kg(acc[1][0], wq[1][0], xf_prev);
        mma_kg(acc[1][1], wq[1][1], xf_prev);
        mma_kg(acc[1][2], w3_prev, xf_prev);
        if (__builtin_expect(fb_taps != 0, 0)) {
            const char *gx = gplane + h * 16;
#pragma unroll 1
            for (unsigned left = fb_taps; left != 0; left &= left - 1) {
                const int tap = __builtin_ctz(left);
                const char *wtap = wbase_g + (size_t)tap * C::DCN_TAP;
                const unsigned g0 = gm0[tap], g1 = gm1[tap], g2v = gm2[tap];
                const bool need_fb = ((lane_fb >> tap) & 1u) != 0;
                const float py = need_fb ? __uint_as_float(g0) : 0.0f, px = need_fb ? __uint_as_float(g1) : 0.0f;
                const float mk = need_fb ? __uint_as_float(g2v) : 0.0f;
                const float fy = floorf(py), fx = floorf(px);
                const int hl = (int)fy, wl = (int)fx, hh = hl + 1, wh = wl + 1;
                const float lh = py - fy, lw = px - fx, uh = 1.0f - lh, uw = 1.0f - lw;
                const int hlc = min(max(hl, 0), H - 1), wlc = min(max(wl, 0), W - 1);
                const int hhc = min(max(hh, 0), H - 1), whc = min(max(wh, 0), W - 1);
                const bool vhl = (unsigned)hl < (unsigned)H, vhh = (unsigned)hh < (unsigned)H;
                const bool vwl = (unsigned)wl < (unsigned)W, vwh = (unsigned)wh < (unsigned)W;
                const float w4[4] = {vhl && vwl ? mk * (uh * uw) : 0.0f, vhl && vwh ? mk * (uh * lw) : 0.0f,
                                     vhh && vwl ? mk * (lh * uw) : 0.0f, vhh && vwh ? mk * (lh * lw) : 0.0f};
                const unsigned gpk = (__umul24((unsigned)hlc, (unsigned)W) + (unsigned)wlc) | ((unsigned)(whc - wlc) << 24) |
                                     ((unsigned)(hhc - hlc) << 25) | (need_fb ? 1u << 26 : 0u);
                const unsigned w01h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[0], (half_t)w4[1]});
                const unsigned w23h = __builtin_bit_cast(unsigned, f16x2_t{(half_t)w4[2], (half_t)w4[3]});
                unsigned w01[2], w23[2], gp[2];
                {
                    auto both = [&](unsigned x, unsigned (&out)[2]) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
                        out[0] = sw[0]; out[1] = sw[1];
                    };
                    both(w01h, w01); both(w23h, w23); both(gpk, gp);
                }
                auto corners_of = [&](unsigned g, unsigned (&pc)[4]) {
                    const unsigned pix = g & 0xffffffu, ddx = (g >> 24) & 1u, ddy = (g >> 25) & 1u;
                    pc[0] = pix; pc[1] = pix + ddx; pc[2] = pix + (ddy ? (unsigned)W : 0u); pc[3] = pc[2] + ddx;
                };
                const unsigned w3a = w3lane + (unsigned)(tap * C::W3_TAP);
#pragma unroll 1
                for (int kg = 0; kg < 4; ++kg) {
                    f16x8 wf[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) wf[n] = *reinterpret_cast<const f16x8 *>(wtap + (kg * 2 + n) * 1024 + l16);
                    const f16x8 w3f = __builtin_bit_cast(f16x8, lds_read16(lds_w3 + w3a + (unsigned)(kg * 128)));
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        unsigned d[4][4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) d[c][0] = d[c][1] = d[c][2] = d[c][3] = 0u;
                        if ((gp[m] >> 26) & 1u) {
                            unsigned pc[4];
                            corners_of(gp[m], pc);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const u32x4_t v = to_f16_piece_rt<TS>(*reinterpret_cast<const u32x4_t *>(gx + (size_t)__umul24(pc[c], ps_bytes) + (unsigned)(kg * 32)), p.in_f16);
                                d[c][0] = v[0]; d[c][1] = v[1]; d[c][2] = v[2]; d[c][3] = v[3];
                            }
                        }
                        unsigned xd[4];
                        blend_corners_cm<4>(d, w01[m], w23[m], xd);
                        const f16x8 xf = __builtin_bit_cast(f16x8, u32x4_t{xd[0], xd[1], xd[2], xd[3]});
                        mma_kg(acc[m][0], wf[0], xf);
                        mma_kg(acc[m][1], wf[1], xf);
                        mma_kg(acc[m][2], w3f, xf);
                    }
                }
                {
                    unsigned vt[4][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) vt[c][0] = vt[c][1] = vt[c][2] = vt[c][3] = 0u;
                    if (need_fb) {
                        unsigned pc[4];
                        corners_of(gpk, pc);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const char *src = tplane ? tplane + (size_t)__umul24(pc[c], tail_bytes) : gplane + (size_t)__umul24(pc[c], ps_bytes) + 128;
                            const u32x2_t raw = *reinterpret_cast<const u32x2_t *>(src);
                            const u32x4_t cv = to_f16_piece_rt<TS>(u32x4_t{raw[0], raw[1], 0u, 0u}, p.in_f16);
                            vt[c][0] = cv[0]; vt[c][1] = cv[1];
                        }
                    }
                    unsigned td[4];
                    blend_corners_cm<2>(vt, w01h, w23h, td);
                    const int j = tap >> 2, hsel = (tap >> 1) & 1, u = tap & 1;
                    unsigned tm[2][2];
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(td[d], td[d], false, false);
                        tm[0][d] = h == hsel ? sw[0] : 0u;
                        tm[1][d] = h == hsel ? sw[1] : 0u;
                    }
                    f16x8 wt[3];
#pragma unroll
                    for (int n = 0; n < 3; ++n) wt[n] = *reinterpret_cast<const f16x8 *>(wtl + (j * 3 + n) * 1024 + l16);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const u32x4_t bq = u ? u32x4_t{0u, 0u, tm[m][0], tm[m][1]} : u32x4_t{tm[m][0], tm[m][1], 0u, 0u};
                        const f16x8 xf = __builtin_bit_cast(f16x8, bq);
#pragma unroll
                        for (int n = 0; n < 3; ++n) mma_kg(acc[m][n], wt[n], xf);
                    }
                }
            }
        }
    };

    // ================================================================ phase S: stores (no activation: ema_vfi.py:136-138 chains the blocks directly)
    auto phase_S = [&]() {
        if (!valid) return;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (!in_img[m]) continue;
            TS *op = reinterpret_cast<TS *>(p.out) + (((size_t)b * H + py_y[m]) * W + px_x) * p.out_ps;
            if (std::is_same<TS, bf16_t>::value && p.out_f16) {
                half_t *oh = reinterpret_cast<half_t *>(op);
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    if (p.cstore - n * 32 > 0) store_frag(oh + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
            } else {
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    if (p.cstore - n * 32 > 0) store_frag(op + n * 32, acc[m][n], h, p.cstore - n * 32, [](float v, int) { return v; });
            }
        }
    };

    // rendezvous without memory semantics (the T1 | T2 seam and the O G seam: nothing is handed over there, the barrier only keeps the
    // two halves two slots apart) - a bare s_barrier: no vmcnt / lgkmcnt drain of the tap loop's prefetched weight fragments
    auto rendezvous = [&]() { asm volatile("s_barrier" ::: "memory"); };

    // ================================================================ the schedule
    // One copy of every phase; iteration `it` stores tile it - 1, then takes tile `it` through P | O G | T1 | T2 (iteration n_iter has no
    // tile left: it only stores the last one and keeps the barrier count).  `valid` still describes the previous tile when S runs.
    if (half == 1) { rendezvous(); rendezvous(); }   // half B runs two slots behind half A
#pragma unroll 1
    for (int it = 0; it <= n_iter; ++it) {
        asm volatile("" : "+v"(l16));
        phase_S();
        set_tile(it);
        phase_P();
        __syncthreads();                             // window landed (vmcnt(0) in front) and visible to the half's four waves
        phase_OG();
        rendezvous();
        phase_T(0, D::T1_TAPS);
        rendezvous();
        phase_T(D::T1_TAPS, 9);
        phase_T_finish();
        __syncthreads();                             // every wave of the half is done with its window before the next tile's DMA overwrites it
    }
    if (half == 0) { rendezvous(); rendezvous(); }
}

template <typename TS> static int launch_deform_pack4(const DeformParams &p, hipStream_t s)
{
    using C = Pack3;
    using D = Pack4;
    static PerDeviceOnce once;
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&deform_pack4_kernel<TS>), D::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const long long ntiles = (long long)((p.W + C::TCOLS - 1) / C::TCOLS) * ((p.H + C::TROWS - 1) / C::TROWS) * p.B;
    if (ntiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const long long npairs = (ntiles + 1) / 2;
    const int grid = (int)(npairs < ncu ? npairs : ncu);          // one persistent workgroup per CU
    const int n_iter = (int)((npairs + grid - 1) / grid);          // every workgroup runs the same number of iterations (and barriers)
    deform_pack4_kernel<TS><<<(unsigned)grid, D::THREADS, D::LDS_BYTES, s>>>(p, n_iter);
    return (int)hipGetLastError();
}
