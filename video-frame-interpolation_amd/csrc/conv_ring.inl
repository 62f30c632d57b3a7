// conv_ring.inl - the 64 -> 64 layers (feat_ext_blocks, motion_estimation.0 / .1: ema_vfi.py:74-76, 90-91) and reconstruction.0
// (67 -> 64: ema_vfi.py:103) with the weights stationary in registers and the input rows in an LDS ring.  Included by conv3x3.inl.
//
// The structure of conv3x3_s2ring_kernel at stride 1 (measured there: 543 -> 343 us against a tile kernel that streamed weights
// through LDS), for the layers conv3x3_pingpong16_kernel served at 533 us = 4.15 TB/s of real HBM traffic, 17 % of it vertical halo:
//   * a workgroup of four waves walks DOWN a strip of 64 output columns, one output row per step, one barrier per step;
//   * wave w owns output fragment (w & 1) (channels 32 (w & 1) .. + 31) of column block (w >> 1) (32 pixels) and keeps the
//     fragment's 64 x 9 x 32 weights in 144 VGPRs for the whole kernel: no weights in LDS, one ds_read_b128 per MFMA;
//   * input rows live in a ring of D + 2 (output row y reads rows y-1, y, y+1; rows y+2 .. y+D are in flight, D = 3: an HBM
//     round trip under load is several steps long): every input row is DMA'd once per strip segment - no vertical halo,
//     66 / 64 horizontally.  The DMA is issued from inline asm and drained by a COUNTED s_waitcnt: hipcc drains every
//     __builtin_amdgcn_global_load_lds it knows of (vmcnt(0)) in front of the next ds_read and of __syncthreads(), which made the
//     prefetch one step deep at best.  Every wave therefore issues exactly 3 DMA + 2 store instructions per step (dummy DMA
//     into a scratch KiB, stores as buffer stores whose out-of-range lanes are dropped instead of branched over), so that "all
//     but the youngest 2 + 5 (D - 2)" retires exactly the row the step needs;  Pixels keep the 144-byte pitch of the fusion tensors
//     (36 dwords: conflict-free for the 16 lanes a ds_read_b128 serves together); 128-byte inputs get a zero ninth piece;
//   * TAIL (reconstruction.0): channels 64..66 of the nine taps are three more k-groups, K = 16 j + 8 h + e <-> tap slot
//     4 j + 2 h + (e >> 2), channel 64 + (e & 3) (deform_pack3.inl's im2col tail): 39 MFMAs per row instead of the 45 of CK = 80,
//     operands straight from the ninth piece of the ring pixels (two ds_read_b64);
//   * outputs go through a double-buffered LDS row (64 pixels x 128 bytes) so that a store instruction writes whole lines;
//   * HEAD (motion_estimation.1 + .2, ema_vfi.py:90-92): the 64 -> 64 rows are NOT stored; they go (zeroed outside the image: the
//     next convolution's padding) into a second LDS ring of two rows, and one step behind them the same workgroup feeds each row
//     to the 64 -> <= 2 planar head (the flow) on v_mfma_f32_16x16x32 with the head's VERTICAL taps on the MFMA's rows (six MFMAs
//     per wave and row add the row's contribution to the three head rows it belongs to; see the kernel).  A strip then
//     yields 62 head columns for 64 computed ones and a segment computes two extra rows; in exchange one launch, 0.94 GB of
//     writes and 0.94 GB of reads per B = 8 x 720p disappear (DMA depth 2 instead of 3: the two rings fill the 80 KiB);
//   * work items = (strip, vertical segment), dealt round-robin to 2 workgroups per CU; the host picks the segment height so
//     that the item count fills whole rounds (launch_conv_ring).
#ifndef EMAVFI_RING_ABL
#define EMAVFI_RING_ABL 0   // timing-only ablations (diagnostic builds): 1 every DMA reads the zero page, 2 no row is stored, 4 no epilogue arithmetic
#endif
#ifndef EMAVFI_RING_DEPTH
#define EMAVFI_RING_DEPTH 3
#endif
#ifndef EMAVFI_RING_AHEAD
#define EMAVFI_RING_AHEAD 4
#endif
// ---- what a workgroup of a persistent ring kernel computes: pieces (strip, output rows [ys, ye)).
//   * chunked (nseg == 0; round 5, the default): the launch's nstrip * Hout output rows, strip after strip, are cut into gridDim.x equal
//     contiguous ranges - a workgroup walks down ONE range, i.e. one or two pieces.  Every piece primes its rings and reads (or, in the
//     fused kernels, computes) two halo rows once: 720p on 512 workgroups is 1.4 pieces of ~225 rows per workgroup instead of five
//     segments of 45 rows;
//   * segments (nseg > 0; EMAVFI_RING_CHUNK=0): every strip is cut into nseg segments of seg_rows rows, dealt round-robin
//     (conv_ring_segments picks nseg so that the items fill whole rounds).
struct RingWork {
    int left, strip0, y0;              // chunked: rows left in this workgroup's range, the next piece's strip and first row
    int cur, end, stride;              // segments: the next item, the item count, the grid size (0 = chunked)
    int hout, nstrip, seg_rows;
    __device__ RingWork(int nstrip_, int hout_, int nseg, int seg_rows_) : hout(hout_), nstrip(nstrip_), seg_rows(seg_rows_)
    {
        if (nseg == 0) {   // seg_rows = (nstrip * hout) / gridDim.x (the host's division); the first `rem` workgroups take one row more
            const int g = (int)blockIdx.x, q = seg_rows_, rem = nstrip * hout - q * (int)gridDim.x;
            const unsigned first = (unsigned)(g * q + min(g, rem));
            left = q + (g < rem ? 1 : 0);
            strip0 = (int)(first / (unsigned)hout);
            y0 = (int)first - strip0 * hout;
            cur = end = stride = 0;
        } else {
            cur = (int)blockIdx.x;
            end = nstrip * nseg;
            stride = (int)gridDim.x;
            left = strip0 = y0 = 0;
        }
    }
    __device__ bool next(int &strip, int &ys, int &ye)
    {
        if (stride == 0) {
            if (left <= 0) return false;
            strip = strip0;
            ys = y0;
            ye = min(hout, y0 + left);
            left -= ye - ys;
            ++strip0;
            y0 = 0;
            return true;
        }
        if (cur >= end) return false;
        strip = cur % nstrip;
        ys = (cur / nstrip) * seg_rows;
        ye = min(ys + seg_rows, hout);
        cur += stride;
        return true;
    }
};
static void conv_ring_segments(int nstrip, int Hout, int grid, int *nseg_out, int *seg_rows_out);
// grid size and (nseg, seg_rows) of a persistent ring launch
static int conv_ring_work(int nstrip, int Hout, int grid, int *nseg, int *seg_rows)
{
    if (!(emavfi_switches() & SW_NO_RING_CHUNK) && (long)nstrip * Hout < (1l << 30)) {
        const int total = nstrip * Hout, by8 = (total + 7) / 8;   // (tiny launches: at least eight rows per workgroup)
        const int nwg = by8 < grid ? by8 : grid;
        *nseg = 0;
        *seg_rows = total / nwg;
        return nwg;
    }
    conv_ring_segments(nstrip, Hout, grid, nseg, seg_rows);
    const int nitems = nstrip * *nseg;
    return nitems < grid ? nitems : grid;
}

#ifndef EMAVFI_HEAD_DEPTH
#define EMAVFI_HEAD_DEPTH 2
#endif
template <typename T, bool TAIL, bool HEAD> struct ConvRingCfg {
    static constexpr int PSTR = 144, SP = 9, TW = 64, IW = TW + 2, ROWSLOT = IW * SP, ROWINST = (ROWSLOT + 63) / 64, ROWB = ROWINST * 1024;
    static constexpr int TWO = HEAD ? TW - 2 : TW;   // columns a strip contributes to the launch's output
    static constexpr int D = HEAD ? EMAVFI_HEAD_DEPTH : EMAVFI_RING_DEPTH, RING = D + 2;
    // !HEAD: two output staging rows, 144-byte pixels (conflict-free for the 32x32 epilogue's writes and the 8-lanes-per-pixel store reads).
    // HEAD: the ring of two 64 -> 64 rows (one being written, one being read by the head), UNPADDED 128-byte pixels whose 16-byte unit u of pixel c lies at u ^ swz16(c): the head's
    // 16x16x32 operand reads (lane (j, kb): unit 4 k32 + kb of pixel c0 + j) are conflict-free that way; at 144 bytes every service
    // group of every read had a two-way conflict (20.3 % of the kernel's LDS cycles; tools/lds_swizzle_search.py)
    static constexpr int STG_PX = HEAD ? 128 : 128 + 16, STG = TW * STG_PX, NSTG = 2;
    static constexpr int STG_OFF = RING * ROWB, BIAS_OFF = STG_OFF + NSTG * STG, BIAS_BYTES = HEAD ? 256 : 16 * 64 * 4;
    static constexpr int HW_OFF = BIAS_OFF + BIAS_BYTES, HW_BYTES = HEAD ? 512 : 0;   // (HEAD: columns 62, 63 of the head read two pixels past a row)
    static constexpr int SCRATCH_OFF = HW_OFF + HW_BYTES, LDS_BYTES = SCRATCH_OFF + 1024;
    // stores per step and wave: HEAD one (lane (j, kb): plane kb of head pixel j); else two for the row's 512 16-byte units + one for the 64 pad units of
    // ConvParams::out_fill (all lanes out of range when it is off: the count must not depend on it)
    static constexpr int NDMA = (ROWINST + 3) / 4, NSTORE = HEAD ? 1 : 3, VMWAIT = NSTORE + (NDMA + NSTORE) * (D - 2);
    static constexpr int WMAIN = 9 * 4 * 2 * 1024;   // bytes of [tap][kg][fragment][lane][8]; the tail [j 3][fragment][lane][8] follows
    static_assert(sizeof(T) == 2 && 2 * LDS_BYTES <= 160 * 1024 && !(TAIL && HEAD), "16-bit types; two workgroups per CU");
};

template <typename T, bool TAIL, bool HEAD, bool ALT = false>   // ALT: ConvParams::out_alt (a template argument: a run-time select in
                                                                // the epilogue cost the 64 -> 64 layers 20 us each)
__global__ __launch_bounds__(256, 2) void conv3x3_ring_kernel(const ConvParams p, const int nseg, const int seg_rows)
{
    using C = ConvRingCfg<T, TAIL, HEAD>;
    using vec = typename DT<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;
    typedef __attribute__((address_space(3))) char lchar_t;
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u2_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = (unsigned)(size_t)(lchar_t *)smem;
    const int r = lane & 31, h = lane >> 5;
    const int frag = wave & 1, cb = wave >> 1;
    const int ntx = (p.Wout + C::TWO - 1) / C::TWO, nstrip = ntx * p.B;
    const char *zeros = (const char *)p.zeros;
    const int npieces = TAIL ? 9 : 8;   // pieces of an input pixel that are read (a 64-channel layer never reads the ninth)
    const unsigned pixbytes = (unsigned)p.in_ps * (unsigned)sizeof(T);
    const unsigned rowbytes = (unsigned)p.Win * pixbytes;
    const bool relu = p.epi == EPI_RELU;

    // ---- this wave's fragment of the weights
    vec wf[9][4], wt[TAIL ? 3 : 1];
    {
        const char *wb = (const char *)p.w + frag * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) wf[t][kg] = *reinterpret_cast<const vec *>(wb + (t * 4 + kg) * 2048);
        if (TAIL) {
#pragma unroll
            for (int j = 0; j < 3; ++j) wt[j] = *reinterpret_cast<const vec *>(wb + C::WMAIN + j * 2048);
        }
    }
    // ---- HEAD: the 64 -> 2 head with its VERTICAL taps on the matrix core's rows (as conv_ring_tail.inl's head).  Row 4 c + dy of the
    // 16-row A operand of the MFMA for (horizontal tap dx, k32) holds W[c][dy][dx][32 k32 ..] (c < 2, dy < 3; the other rows read a
    // zero row of the blob), the B operand is the newest 64 -> 64 row rho, 16 pixels shifted by dx: SIX MFMAs add row rho's
    // contribution to the three head rows it belongs to - register dy of lane (j, kb = c) is head row rho + 1 - dy, plane c, pixel j;
    // register 2 is then a finished row and the accumulator rotates.  Wave w owns head columns [16 w, 16 w + 16): 6 operand reads and
    // 6 MFMAs per wave and step (was 18 + 18 and a round trip of partial sums through LDS), 24 weight registers, ONE store.
    // Weights: block 0 of the 16x16x32 packing [tap][k32][cout16 block 0..1][lane (i, kb)][8] (rows >= 2 are zero in the blob).
    float hb = 0.0f;   // the head's bias (loaded once: a global load inside the row loop would make hipcc wait vmcnt(0), i.e. for the
                       // whole DMA ring)
    vec hwr[HEAD ? 6 : 1];
    int hxo[HEAD ? 6 : 1];
    if constexpr (HEAD) {
        const int j = lane & 15, kb = lane >> 4, c = j >> 2, dy = j & 3;
        const bool realrow = c < p.nplanes && dy < 3;
        hb = kb < p.nplanes ? p.head_bias[kb < p.nplanes ? kb : 0] : 0.0f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int dx = q >> 1, k32 = q & 1;
            hwr[q] = *reinterpret_cast<const vec *>((const char *)p.head_w + (realrow ? (dy * 3 + dx) * 2 + k32 : 0) * 2048 + (kb * 16 + (realrow ? c : 15)) * 16);
            // byte offset of this lane's operand inside a row: pixel c = 16 wave + j + dx, unit (4 k32 + kb) ^ swz16(c) (swz16 has
            // period 8 pixels)
            hxo[q] = wave * 2048 + (((j + dx) * C::STG_PX + ((kb ^ swz16(j + dx)) << 4)) ^ (k32 * 64));
        }
    }
    // ---- lane constants of the row DMA: instruction jn covers 16-byte slots [64 jn, 64 jn + 64) of a ring row
    unsigned xoff[C::NDMA];
    unsigned xcol[C::NDMA];
#pragma unroll
    for (int i = 0; i < C::NDMA; ++i) {
        const int q = (i * 4 + wave) * 64 + lane;
        const int px = q / C::SP, pc = q - px * C::SP;
        xoff[i] = (unsigned)px * pixbytes + (unsigned)pc * 16u;
        xcol[i] = (q < C::ROWSLOT && pc < npieces) ? (unsigned)px : 0x40000000u;   // far outside any image: the zero page
    }
    // the tail's two 8-byte reads per k-group: this lane half's tap slots 4 j + 2 h, 4 j + 2 h + 1
    int tdy[6], tdx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ts = 4 * (k >> 1) + 2 * h + (k & 1), tap = ts < 9 ? ts : 8;
        tdy[k] = tap / 3;
        tdx[k] = tap - 3 * tdy[k];
    }

    RING_STAMP_DECL;
    RingWork work(nstrip, p.Hout, nseg, seg_rows);
    int strip, ys, ye;
#pragma unroll 1
    while (work.next(strip, ys, ye)) {
        const int b = strip / ntx, tx = strip - b * ntx;
        // rows of the 64 -> 64 convolution this item computes: [a0, a1] (HEAD: one more on either side for the head's taps)
        const int a0 = HEAD ? ys - 1 : ys, a1 = HEAD ? ye : ye - 1;
        const char *gin = (const char *)p.in + (size_t)b * p.Hin * p.Win * p.in_ps * sizeof(T);
        const int ox0 = tx * C::TWO - (HEAD ? 1 : 0);   // image column of the 64 -> 64 convolution's column 0
        const int ix0 = ox0 - 1;
        // exactly NDMA instructions per wave (instruction slots past the row land in a scratch KiB; rows outside the image, or
        // not needed by this segment, read the zero page)
        auto dma_row = [&](int gy, int slot, bool live) {
            const bool rowok = live && (unsigned)gy < (unsigned)p.Hin && !(EMAVFI_RING_ABL & 1);   // wave-uniform
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
        // the bias table in LDS: [border class 0..15][64] floats (bias_mode 0: one class) - no global loads inside the row loop
        // (hipcc would wait for them with vmcnt(0), i.e. for the whole DMA ring)
        {
            const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(p.bias + (p.bias_mode == 1 ? (size_t)b * 16 * 64 : 0));
            if (tid < ((!HEAD && p.bias_mode == 1) ? 256 : 16)) reinterpret_cast<f32x4 *>(smem + C::BIAS_OFF)[tid] = bsrc[tid];
        }
        const int xg = ox0 + cb * 32 + r;   // this lane's column of the 64 -> 64 convolution
        const int xm = (xg >= 1 ? 1 : 0) | (xg <= p.Wout - 2 ? 2 : 0);

        // ---- the step's NSTORE store instructions (buffer stores: lanes outside the image, or !real, are dropped by the range check)
        char *obase = nullptr;
        unsigned soff[C::NSTORE];
        if constexpr (!HEAD) {
            obase = reinterpret_cast<char *>(p.out) + (((size_t)b * p.Hout * p.Wout + (size_t)tx * C::TW) * p.out_ps + p.out_coff) * sizeof(T);
            const int npx = min(C::TW, p.Wout - tx * C::TW);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, ch = q & 7;
                soff[i] = (px < npx && ch * 8 < p.cstore) ? (unsigned)px * (unsigned)p.out_ps * (unsigned)sizeof(T) + ch * 16u : 0x80000000u;
            }
            soff[2] = (p.out_fill && tid < npx) ? (unsigned)tid * (unsigned)p.out_ps * (unsigned)sizeof(T) + 128u : 0x80000000u;
        } else {
            // the head: wave w owns head columns [16 w, 16 w + 16) of the strip's 62; lane (j, kb) holds plane kb of pixel j
            const int hc = wave * 16 + (lane & 15), hx = tx * C::TWO + hc, kb = lane >> 4;
            const bool hok = hc < C::TWO && hx < p.Wout && kb < p.nplanes;
            soff[0] = hok ? ((unsigned)hx + (unsigned)kb * (unsigned)p.Hout * (unsigned)p.Wout) * 4u : 0x80000000u;
        }
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
            __builtin_amdgcn_raw_buffer_store_b128(u4_t{0u, 0u, 0u, 0u}, rs, soff[C::NSTORE - 1], 0, 0);
        };
        // HEAD: the 64 -> 64 row rho = yb + 1 (mid-ring slot (rho - a0) & 1, written in the step before) enters the head; head row yb is
        // finished by it.  hacc: register dy = the pending sums of head row rho + 1 - dy (rows below ys are never stored: whatever the
        // registers and the ring hold when an item starts only reaches those)
        f32x4 hacc = {0.0f, 0.0f, 0.0f, 0.0f};
        // (two halves: in the row loop the six MFMAs queue behind the main contraction's and the main epilogue runs in their shadow)
        auto head_mma = [&](int yb) {
            if constexpr (HEAD) {
                const lchar_t *mr = (const lchar_t *)smem + C::STG_OFF + ((yb + 1 - a0) & 1) * C::STG;
                vec hxv[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) hxv[q] = *reinterpret_cast<const __attribute__((address_space(3))) vec *>(mr + hxo[q]);
                hacc = f32x4{0.0f, hacc[0], hacc[1], 0.0f};
#pragma unroll
                for (int q = 0; q < 6; ++q) mma_k32(hacc, hwr[q], hxv[q]);
            }
        };
        auto head_out = [&](int yb, bool real) {
            if constexpr (HEAD) {
                float o = hacc[2] + hb;
                if (p.round16) o = (float)(half_t)o;
                float *orow = p.out_planar + (size_t)b * p.nplanes * p.Hout * p.Wout + (size_t)(real ? yb : ys) * p.Wout;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(orow, 0, real ? 0x7ffffff0 : 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, soff[0], 0, 0);
            }
        };
        // input rows a0 - 1 .. a0 + D - 1 -> slots 0 .. D, each followed by NSTORE dropped stores: the steady state's instruction pattern
#pragma unroll 1
        for (int k = 0; k <= C::D; ++k) {
            dma_row(a0 - 1 + k, k, a0 - 1 + k <= a1 + 1);
            if constexpr (HEAD) {   // (the store of the steady-state pattern, dropped.  NOT head_row: its accumulator carries state)
                const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(p.out_planar, 0, 0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(0u, rs0, soff[0], 0, 0);
            } else
                store_row(ys, false);
        }
        int s0 = 0;   // ring slot of input row y - 1
#pragma unroll 1
        for (int y = a0; y <= a1; ++y) {
            RING_STAMP(ts0);
            // this wave's part of input row y + 1 (all but the youngest VMWAIT instructions) and its LDS writes of row y - 1
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::VMWAIT) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            RING_STAMP(ts1);
            {
                int sl = s0 + C::D + 1; sl = sl >= C::RING ? sl - C::RING : sl;
                dma_row(y + C::D, sl, y + C::D <= a1 + 1);
            }
            if constexpr (!HEAD) store_row(y - 1, y > a0 && !(EMAVFI_RING_ABL & 2));   // (HEAD: the head stage follows the main contraction - head_mma / head_out below)
            RING_STAMP(ts2);
            f32x16 acc[2];
            {
                // motion_estimation.0 (bias_mode 1): the folded context half depends on the pixel's border class
                const int ym = (y >= 1 ? 1 : 0) | (y <= p.Hout - 2 ? 2 : 0), cls = (!HEAD && p.bias_mode == 1) ? ym * 4 + xm : 0;
                const f32x4 *lb = reinterpret_cast<const f32x4 *>(smem + C::BIAS_OFF + (cls * 64 + frag * 32 + 4 * h) * 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = lb[2 * g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][4 * g + e] = v[e];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
            const char *xl = ring + (cb * 32 + r) * C::PSTR + h * 16;
            // TAIL: the three im2col MFMAs (round 5, end) no longer open the step behind their own exposed reads - the reads go out inside
            // the main loop's last iterations, where the operand queue has free registers, and the MFMAs queue behind the 36
            const char *tl = ring + (cb * 32 + r) * C::PSTR + 128;
            int toff[TAIL ? 6 : 1];
            if constexpr (TAIL) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    int sl = s0 + tdy[k]; sl = sl >= C::RING ? sl - C::RING : sl;
                    toff[k] = sl * C::ROWB + tdx[k] * C::PSTR;
                }
            }
            vec tq[TAIL ? 3 : 1];
            {
                // operands EMAVFI_RING_AHEAD k-groups ahead of their MFMAs: one wave's read -> MFMA chain must not expose the LDS latency
                constexpr int AH = (TAIL || HEAD) ? EMAVFI_RING_AHEAD - 1 : EMAVFI_RING_AHEAD;   // (TAIL: 12 more weight registers; HEAD: 24 + the head stage behind the loop)
                const char *xb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    int sl = s0 + dy; sl = sl >= C::RING ? sl - C::RING : sl;
                    xb[dy] = xl + sl * C::ROWB;
                }
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
                    if constexpr (TAIL) {
                        if (s >= 33) {   // one im2col operand per iteration 33, 34, 35 (two 8-byte reads each)
                            const int j = s - 33;
                            u2_t lo = *reinterpret_cast<const u2_t *>(tl + toff[2 * j]);
                            u2_t hi = *reinterpret_cast<const u2_t *>(tl + toff[2 * j + 1]);
                            if (j == 2) {   // tap slots 9..11 do not exist: zero operand (their weights are zero, but 0 x Inf is not)
                                const unsigned keep0 = h ? 0u : ~0u;
                                lo[0] &= keep0; lo[1] &= keep0; hi[0] = 0u; hi[1] = 0u;
                            }
                            tq[j] = __builtin_bit_cast(vec, u4_t{lo[0], lo[1], hi[0], hi[1]});
                        }
                    }
                    mma_kg(acc[s & 1], wf[s >> 2][s & 3], xq[s % (AH + 1)]);
                    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read to just above its MFMA)
                }
                if constexpr (TAIL) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) mma_kg(acc[j & 1], wt[j], tq[j]);
                }
            }
            RING_STAMP(ts3);
            if constexpr (HEAD) head_mma(y - 2);   // the 64 -> 64 row of the step before enters the head: six MFMAs behind the 36
            // ---- optional ReLU; this wave's 32 channels of its 32 pixels into the row's staging buffer (HEAD: the ring of 64 -> 64
            // rows, zero outside the image: they are the head convolution's padding)
            {
                const int slot = HEAD ? ((y - a0) & 1) : (y & 1);
                const bool inside = !HEAD || ((unsigned)y < (unsigned)p.Hout && (unsigned)xg < (unsigned)p.Wout);
                lchar_t *stg = (lchar_t *)smem + C::STG_OFF + slot * C::STG + (cb * 32 + r) * C::STG_PX;
                const int usw = HEAD ? swz16(cb * 32 + r) : 0;
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    unsigned a[2], c[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float v0 = acc[0][4 * g + 2 * q] + acc[1][4 * g + 2 * q], v1 = acc[0][4 * g + 2 * q + 1] + acc[1][4 * g + 2 * q + 1];
                        float u0 = acc[0][4 * (g + 1) + 2 * q] + acc[1][4 * (g + 1) + 2 * q], u1 = acc[0][4 * (g + 1) + 2 * q + 1] + acc[1][4 * (g + 1) + 2 * q + 1];
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        if (EMAVFI_RING_ABL & 4) { v0 = acc[0][4 * g + 2 * q]; v1 = acc[1][4 * g + 2 * q]; u0 = v0; u1 = v1; }
                        unsigned ua = pack16x2<T>(v0, v1, ALT), ub = pack16x2<T>(u0, u1, ALT);
                        if (HEAD) { const unsigned keep = inside ? ~0u : 0u; ua &= keep; ub &= keep; }
                        const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
                        a[q] = sw[0]; c[q] = sw[1];
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) u4_t *>(stg + 16 * ((frag * 4 + g + h) ^ usw)) = u4_t{a[0], a[1], c[0], c[1]};
                }
            }
            if constexpr (HEAD) head_out(y - 2, y - 2 >= ys && !(EMAVFI_RING_ABL & 2));
            s0 = s0 + 1 >= C::RING ? 0 : s0 + 1;
            RING_STAMP(ts4);
            RING_STAMP_ADD(0, ts0, ts1); RING_STAMP_ADD(1, ts1, ts2); RING_STAMP_ADD(2, ts2, ts3); RING_STAMP_ADD(3, ts3, ts4);
            RING_STAMP_STEP();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the next item primes the same slots)
        __syncthreads();
        if constexpr (HEAD) {
            head_mma(a1 - 1);         // (a1 = ye: the segment's last head row, from the 64 -> 64 rows ye - 2 .. ye)
            head_out(a1 - 1, true);
            __syncthreads();          // the next item's first rows overwrite the ring of 64 -> 64 rows
        } else
            store_row(a1, true);
    }
    RING_STAMP_WRITE(p, 10 + (TAIL ? 1 : 0) + (HEAD ? 2 : 0), 4);
}

// Segment height: the item count should fill whole rounds of the 2-per-CU grid (a strip's segment re-reads two halo rows).
static void conv_ring_segments(int nstrip, int Hout, int grid, int *nseg_out, int *seg_rows_out)
{
    double best = -1.0;
    int bn = 1, br = Hout;
    for (int nseg = 1; nseg <= (Hout + 7) / 8; ++nseg) {
        const int rows = (Hout + nseg - 1) / nseg, ns = (Hout + rows - 1) / rows;
        const long items = (long)nstrip * ns, rounds = (items + grid - 1) / grid;
        const double eff = (double)items / (double)(rounds * grid) * (double)rows / (double)(rows + 2);
        if (eff > best + 1e-9) { best = eff; bn = ns; br = rows; }
    }
    *nseg_out = bn;
    *seg_rows_out = br;
}

template <typename T, bool TAIL, bool HEAD, bool ALT = false> static int launch_conv_ring_t(const ConvParams &p, hipStream_t s)
{
    using C = ConvRingCfg<T, TAIL, HEAD>;
    static PerDeviceOnce once;   // (the library is re-entrant and serves several devices per process)
    if (const hipError_t e_ = set_lds_limit(once, reinterpret_cast<const void *>(&conv3x3_ring_kernel<T, TAIL, HEAD, ALT>), C::LDS_BYTES); e_ != hipSuccess) return (int)e_;
    const int ncu = device_cu_count();
    if (ncu <= 0) return (int)hipErrorInvalidDevice;
    const int nstrip = ((p.Wout + C::TWO - 1) / C::TWO) * p.B, grid = ((emavfi_switches() & SW_RING_ONE_WG) ? 1 : 2) * ncu;   // (SW_RING_ONE_WG: measurement switch, common.h)
    int nseg, seg_rows;
    const int nwg = conv_ring_work(nstrip, p.Hout, grid, &nseg, &seg_rows);
    conv3x3_ring_kernel<T, TAIL, HEAD, ALT><<<nwg, 256, C::LDS_BYTES, s>>>(p, nseg, seg_rows);
    return (int)hipGetLastError();
}

template <typename T> static int launch_conv_ring(const ConvParams &p, hipStream_t s)
{
    if (p.stride != 1 || p.nchunk != 1 || p.npass != 1 || p.nf != 2 || p.bias_mode > 1 || (p.epi != EPI_NONE && p.epi != EPI_RELU)) return -2;
    if (p.head_w) {   // + a planar head of <= 2 channels computed from the rows in LDS (NSTORE stores per step = its planes)
        if (p.ring != 2 || p.bias_mode != 0 || !p.head_bias || !p.out_planar || p.nplanes < 1 || p.nplanes > 2) return -2;
        return launch_conv_ring_t<T, false, true>(p, s);
    }
    if (p.out_alt) return p.ring == 2 ? launch_conv_ring_t<T, false, false, true>(p, s) : -2;
    return p.ring == 3 ? launch_conv_ring_t<T, true, false>(p, s) : launch_conv_ring_t<T, false, false>(p, s);
}
