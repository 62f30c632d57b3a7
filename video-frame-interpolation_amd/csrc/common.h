// Shared device-side types for libemavfi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Activations live in HBM pixel-major / channel-minor ("channels-last"):
// element (b, y, x, c) at ((b*H + y)*W + x) * pstride + c, with the channel
// count padded to a multiple of 16 and the pad channels holding exact zeros.
// One pixel's channels are then one contiguous run, which is what both the
// MFMA operand fetch (8 bf16 / 4 fp32 consecutive channels per lane = 16 B) and
// the bilinear / deformable gathers (all channels of one tap) want.

// One "k-group" = what one 16-byte operand load per lane feeds to the matrix core:
//   bf16: 16 input channels -> one v_mfma_f32_32x32x16_bf16
//         (lane (r, h) holds channels 8h..8h+7 of row/col r)
//   f16:  the same shape on v_mfma_f32_32x32x16_f16 (EMAVFI_F16: what torch.cuda.amp.autocast() computes convs in)
//   fp32:  8 input channels -> four v_mfma_f32_32x32x2_f32
//         (lane (r, h) holds channels 4h..4h+3; MFMA j contracts channels {j, 4+j})
template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int CHKG = 8;  // channels per k-group
    static constexpr int EPV = 4;   // elements per 16-byte vector
    using vec = f32x4;
};
template <> struct DT<bf16_t> {
    static constexpr int CHKG = 16;
    static constexpr int EPV = 8;
    using vec = bf16x8;
};
template <> struct DT<half_t> {
    static constexpr int CHKG = 16;
    static constexpr int EPV = 8;
    using vec = f16x8;
};

// D[cout][pixel] += W[cout][k] * X[k][pixel] for one k-group.  Orientation: weights are the
// A operand (rows = output channels), pixels are the B operand (cols), so each lane ends up
// holding 4 consecutive output channels of ONE pixel per register quad - contiguous in the
// channels-last output.
__device__ __forceinline__ void mma_kg(f32x16 &acc, const bf16x8 &w, const bf16x8 &x)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_kg(f32x16 &acc, const f16x8 &w, const f16x8 &x)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc, 0, 0, 0);
}
// the 16x16x32 shape: D[16 cout][16 pixel] += W[16][32] * X[32][16].  Lane (i = lane & 15, kb = lane >> 4) holds row / column i,
// K elements kb*8 .. kb*8+7; the result lane (j = lane & 15, ib = lane >> 4) holds rows ib*4 .. ib*4+3 of column j.
// tools/microbench/mfma_shape_power.hip: on this power-managed board it delivers 11-15 % more FLOP/s than 32x32x16 in a bare loop and
// the same LDS-fed rate at 7 % less package power.
__device__ __forceinline__ void mma_k32(f32x4 &acc, const bf16x8 &w, const bf16x8 &x)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_k32(f32x4 &acc, const f16x8 &w, const f16x8 &x)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_kg(f32x16 &acc, const f32x4 &w, const f32x4 &x)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[3], acc, 0, 0, 0);
}

// two fp32 values rounded (to nearest even) to T, or to the other 16-bit type, as one 32-bit register (ConvParams::out_alt)
template <typename T> struct Other16;
template <> struct Other16<bf16_t> { using type = half_t; };
template <> struct Other16<half_t> { using type = bf16_t; };
template <> struct Other16<float> { using type = float; };
template <typename T> __device__ __forceinline__ unsigned pack16x2(float a, float b, bool alt)
{
    typedef __attribute__((ext_vector_type(2))) T pair_t;
    typedef typename Other16<T>::type O;
    typedef __attribute__((ext_vector_type(2))) O opair_t;
    if (alt) {
        if (sizeof(T) == 2 && !__is_same(T, half_t)) {   // bf16 kernel storing f16: SATURATE at the f16 range like the packs' own conversion
            a = a > 65504.0f ? 65504.0f : (a < -65504.0f ? -65504.0f : a);   // (include/emavfi.h; NaN stays NaN)
            b = b > 65504.0f ? 65504.0f : (b < -65504.0f ? -65504.0f : b);
        }
        const opair_t q = {(O)a, (O)b};
        return __builtin_bit_cast(unsigned, q);
    }
    const pair_t q = {(T)a, (T)b};
    return __builtin_bit_cast(unsigned, q);
}

// Accumulator register i of lane (r, h) is output channel (i&3) + 8*(i>>2) + 4*h of the
// 32-channel fragment, pixel r (C/D map of the 32x32 MFMA, cdna_hip_programming.md section 3).
__device__ __forceinline__ int acc_channel(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__device__ __forceinline__ void store4(float *p, float a, float b, float c, float d)
{
    *reinterpret_cast<f32x4 *>(p) = f32x4{a, b, c, d};
}
__device__ __forceinline__ void store4(bf16_t *p, float a, float b, float c, float d)
{
    *reinterpret_cast<bf16x4 *>(p) = bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
}
__device__ __forceinline__ void store4(half_t *p, float a, float b, float c, float d)
{
    *reinterpret_cast<f16x4 *>(p) = f16x4{(half_t)a, (half_t)b, (half_t)c, (half_t)d};
}

// Channels-last epilogue store of one 32-channel accumulator fragment of pixel r.
// Lane (r, h) holds channel groups {8g + 4h .. +3}, g = 0..3.  fp32: a group is already 16 bytes.
// bf16: a group is 8 bytes, and 8-byte-per-lane stores are issue-bound (cdna_hip_programming.md T21),
// so groups g and g+1 are exchanged between the h = 0 / h = 1 lanes of the pixel with
// v_permlane32_swap: lane h=0 ends with channels 8g..8g+7, lane h=1 with 8(g+1)..8(g+1)+7 - one
// 16-byte store each.  `limit` = channels of this fragment to store (multiple of 8).
template <typename F>
__device__ __forceinline__ void store_frag(float *base, const f32x16 &acc, int h, int limit, F act)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = 8 * g + 4 * h;
        if (c0 < limit) store4(base + c0, act(acc[4 * g], c0), act(acc[4 * g + 1], c0 + 1), act(acc[4 * g + 2], c0 + 2), act(acc[4 * g + 3], c0 + 3));
    }
}
template <typename T16, typename F>
__device__ __forceinline__ void store_frag16(T16 *base, const f32x16 &acc, int h, int limit, F act)
{
    typedef __attribute__((ext_vector_type(2))) T16 pair_t;
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        unsigned a[2], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ca = 8 * g + 4 * h + 2 * j, cb = 8 * (g + 1) + 4 * h + 2 * j;
            const pair_t pa = {(T16)act(acc[4 * g + 2 * j], ca), (T16)act(acc[4 * g + 2 * j + 1], ca + 1)};
            const pair_t pb = {(T16)act(acc[4 * (g + 1) + 2 * j], cb), (T16)act(acc[4 * (g + 1) + 2 * j + 1], cb + 1)};
            a[j] = __builtin_bit_cast(unsigned, pa);
            b[j] = __builtin_bit_cast(unsigned, pb);
            const auto sw = __builtin_amdgcn_permlane32_swap(a[j], b[j], false, false);
            a[j] = sw[0];
            b[j] = sw[1];
        }
        const int c0 = 8 * (g + h);  // h=0: channels 8g..8g+7, h=1: 8(g+1)..8(g+1)+7
        if (c0 < limit) *reinterpret_cast<uint4 *>(base + c0) = make_uint4(a[0], a[1], b[0], b[1]);
    }
}
template <typename F>
__device__ __forceinline__ void store_frag(bf16_t *base, const f32x16 &acc, int h, int limit, F act) { store_frag16(base, acc, h, limit, act); }
template <typename F>
__device__ __forceinline__ void store_frag(half_t *base, const f32x16 &acc, int h, int limit, F act) { store_frag16(base, acc, h, limit, act); }

// LDS pixel stride for CK channels of T: one 16-byte slot of padding makes the stride an odd
// number of slots, so the 16 lanes of a ds_read_b128 group (consecutive pixels, same channel
// offset) fall on 16 different slots of the 256-byte bank row: conflict-free.
template <typename T, int CK> struct LdsPix {
    static constexpr int BYTES = CK * (int)sizeof(T) + 16;
    static_assert(((BYTES / 16) & 1) == 1, "LDS pixel stride must be an odd number of 16-B slots");
};

// ---- kernel parameter blocks (host fills, passed by value) ----
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_OM = 2, EPI_PLANAR = 3, EPI_PLANAR_TANH01 = 4 };

struct ConvParams {
    const void *in;     // channels-last T
    void *out;          // channels-last T (EPI_NONE/RELU) or fp32 [px][32] (EPI_OM)
    float *out_planar;  // NCHW fp32 (EPI_PLANAR*)
    const void *w;      // packed [pass][chunk][tap][kg][nf][lane][16 B]
    const float *bias;  // [npass*NF*32] or, bias_mode 1, [B][16][npass*NF*32]
    const void *zeros;  // >= 16 bytes of zeros (source of padding / out-of-image DMA lanes)
    int in_ps, out_ps, out_coff;  // pixel strides / channel offset, in elements
    int Hin, Win, Hout, Wout, B;
    int nchunk, npass;
    int cstore;   // channels to store across all passes (multiple of 4)
    int epi, nplanes, bias_mode;
    int ck, nf, stride;  // host-side template selectors
    int round16;         // EMAVFI_AMP16: fp32-stored results (flow, offsets / masks, the frame) hold fp16-rounded values and
                         // sigmoid / tanh / (t+1)/2 round after every op, as fp16 tensors do under autocast
    int mfma16;          // weights packed for v_mfma_f32_16x16x32 (conv3x3_persist16_kernel): [tap][k32][cout16 block][lane][16 B]
    int ring;            // weights in registers, input rows through an LDS ring: 1 = conv3x3_s2ring_kernel (64 -> 128 at stride 2, context_encoding.0),
                         // 2 = conv3x3_ring_kernel (64 -> 64), 3 = the same with the im2col tail (65..67 -> 64: reconstruction.0);
                         // 4 = conv3x3_wreg_kernel (conv_wreg.inl: -> 256 channels, weights streamed into registers, input tile in LDS)
    const void *w2;          // conv_ring2.inl (ring == 2, both layers 64 -> 64): a SECOND conv_block behind this one in the same launch - its packed
    const float *bias2;      // weights (ring layout) and bias; `out*` / cstore / out_alt / out_fill then describe the second layer's output
    const void *head_w;      // conv_ring.inl, ring == 2 only: fuse a 64 -> nplanes (<= 2) planar head (its weights in the 16x16x32 packing,
    const float *head_bias;  // its bias) behind this layer: `out` is not written, `out_planar` gets the head (round16 applies)
    int out_alt;             // ring kernels (64 -> 64, 64 -> 128 stride 2), channels-last epilogue: store the OTHER 16-bit type (a bf16 kernel writes
                             // IEEE f16 bit patterns and vice versa).  bf16 model: `feat` lives as f16 (what the first pack wants on chip,
                             // no conversion pass), its other readers run the f16 kernels on bf16-rounded weights and hand bf16 on
    int out_fill;            // conv3x3_ring_kernel, 64 channels into 144-byte pixels (`feat` into the fusion tensor): also write the pixel's last
                             // 16 bytes (zeros).  128 of every 144 bytes leave a hole in every 128-byte line - partial-line writes that cost
                             // the layer ~240 us at B = 8 x 720p; the bytes belong to nobody yet (the warp writes them later, or never)
    int epi2;                // conv_ring_tail.inl (64 -> 32 -> nplanes <= 3, this layer packed mfma16 with nf == 1): the head's epilogue
                             // (EPI_PLANAR or EPI_PLANAR_TANH01)
    float *pool_part;    // conv_wreg.inl (ring == 4, 256 channels): do NOT store the layer's output, write the per-channel sums of every 4 x 32 pixel
                         // tile instead: [B][tiles][256] floats (context_encoding.2 feeds AdaptiveAvgPool2d and nothing else)
    int in_pieces;       // 16-byte pieces of an input pixel (single-chunk layers) that exist in memory; 0 = all CK of them.
                         // Pieces beyond read as zeros: the 72-channel fusion buffers feed CK = 80 layers this way.
    unsigned long long *stamps;  // diagnostic build (-DEMAVFI_DEFORM_STAMPS=1) only, else null: per-wave phase sums of the LDS-ring kernels
                                 // (tools/ring_stamps.py; EMAVFI_STAMP_RING selects the launch)
    // EMAVFI_F32X3 (round 6): fp32-accurate contraction on the f16 matrix pipe.  Activations are stored as TWO f16 halves per pixel,
    // [hi: Cpad channels | lo: Cpad channels] with v = hi + lo (hi = f16(v), lo = f16(v - hi): 22 significant bits), weights are packed as
    // three chunk copies per 64-channel chunk - (w_hi, w_lo, w_hi) - and the tile kernel runs three virtual chunks per real one:
    // x_hi w_hi, x_hi w_lo, x_lo w_hi (the products are exact in fp32; x_lo w_lo, 2^-22 relative, is dropped).  nchunk counts the VIRTUAL
    // chunks; x3_lo_off / out_lo_off = element offset of the lo half inside an input / output pixel; in_ps / out_ps are whole-pixel strides.
    int x3, x3_lo_off, out_lo_off;
    float *out32;   // x3 only: ALSO store the fp32 value, channels-last with pixel stride out32_ps (the last feature layer writes `feat` into the
    int out32_ps;   // fp32 fusion tensor the exact DCN reads: saves the widening pass' read of 1.9 GB at B = 8 x 720p)
};

// In-kernel stamps of the LDS-ring convolution kernels (diagnostic build only; cdna_hip_programming.md section 7): s_memtime at the
// seams of a row step, per-wave sums over all steps of all items, one row of eight u64 per (workgroup, wave) at the end of the
// kernel: {seg0..seg4, kernel total, kind, steps}.  The stamp waits lgkmcnt(0) (s_memtime is an SMEM instruction), i.e. it also drains
// the wave's outstanding LDS accesses at the seam - the stamped kernel runs a few per cent slower than the product; read SHARES.
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
__device__ __forceinline__ unsigned long long ring_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define RING_STAMP_DECL unsigned long long rsum_[5] = {0, 0, 0, 0, 0}, rsteps_ = 0; const unsigned long long rbegin_ = ring_stamp()
#define RING_STAMP(v) const unsigned long long v = ring_stamp()
#define RING_STAMP_ADD(i, a, b) rsum_[i] += (b) - (a)
#define RING_STAMP_STEP() ++rsteps_
#define RING_STAMP_WRITE(p, kind, nwaves)                                                                  \
    do {                                                                                                   \
        const unsigned long long rend_ = ring_stamp();                                                     \
        const unsigned row_ = blockIdx.x * (nwaves) + (threadIdx.x >> 6);                                  \
        if ((p).stamps && (threadIdx.x & 63) == 0 && row_ < DEFORM_STAMP_ROWS) {                           \
            unsigned long long *o_ = (p).stamps + (size_t)row_ * 8;                                        \
            o_[0] = rsum_[0]; o_[1] = rsum_[1]; o_[2] = rsum_[2]; o_[3] = rsum_[3]; o_[4] = rsum_[4];      \
            o_[5] = rend_ - rbegin_; o_[6] = (kind); o_[7] = rsteps_;                                      \
        }                                                                                                  \
    } while (0)
#else
#define RING_STAMP_DECL
#define RING_STAMP(v)
#define RING_STAMP_ADD(i, a, b)
#define RING_STAMP_STEP()
#define RING_STAMP_WRITE(p, kind, nwaves)
#endif

struct DeformParams {
    const void *x;     // channels-last T, CK channels used
    float *om;         // [px][32] fp32: 18 offsets (dy,dx per tap), 9 masks, 5 pad (read; written first when fused)
    void *out;         // channels-last T
    const void *w;     // packed [tap][kg][nf][lane][16 B]
    const float *bias; // [NF*32]
    const void *zeros; // >= 16 bytes of zeros (DMA source for out-of-image window pixels)
    // fused ModulatedDeformConvPack (bf16 LDS kernel only): when off_w is set the kernel first runs the pack's
    // offset_conv (3x3, cin -> 27, packed like a (CK, nf=1) conv layer) on the staged window and WRITES om
    const void *off_w;
    const float *off_bias;
    // split input (bf16/f16 LDS kernel only): channels [64, 72) of every pixel come from the compact channels-last
    // buffer x_tail[px][tail_ps = 4 since round 6: 8 bytes per pixel; 8 before] instead of x (the warp writes one contiguous record per pixel there instead of 6
    // useful bytes into every 160-byte fusion pixel); channels 72.. are zero
    const void *x_tail;
    int tail_ps;
    int x_ps, out_ps;
    int H, W, B;
    int cstore;
    int cin_real;  // real (unpadded) input channels
    int cout_real; // real output channels (0 = unknown: nf * 32)
    int ck, nf;    // host-side template selectors
    int pack3;     // weights are in the deform_pack3.inl layout (K = 64 per tap + im2col tail, third fragment as an LDS table);
                   // 3: fp32, the deform_f32w.inl layout
    int in_f16, out_f16;  // bf16 storage only: x (and x_tail) / out hold IEEE f16 bit patterns (tensors handed between consecutive packs)
    void *out16;          // fp32 LDS-window kernel (deform_f32w.inl) only, EMAVFI_AMP16: ALSO write the result's fp16 rounding, channels-last with
    int out16_ps;         // pixel stride out16_ps (elements) - what the fp16 offset_conv / reconstruction.0 read (was a separate conversion pass)
    int out16_lo_off;     // EMAVFI_F32X3: > 0 = also write the lo half f16(v - f16(v)) at this element offset of the out16 pixel
    int x3;               // EMAVFI_F32X3: deform_f32w.inl contracts with the three-term f16 split (16x16x16 f16 MFMAs) instead of fp32 MFMAs
    unsigned long long *stamps;  // diagnostic build (-DEMAVFI_DEFORM_STAMPS=1) only, else null
    unsigned *census;            // deform_pack3.inl: null, or 64 slots x 4 u32 (zeroed by the host before the launch) that receive
                                 // {(wave, tap) groups in the fix-up, samples outside the window, max |offset| as float bits, 0}
};
#define DEFORM_CENSUS_SLOTS 64
#define DEFORM_STAMP_STRIDE 14
#define DEFORM_STAMP_ROWS 16384

// A/B switches of the launch sequence and of kernel selection.  They are read from the environment ONCE per process (first use)
// into one atomic word - never per call: getenv racing a setenv from another thread is undefined behaviour, and a flip between
// graph capture and replay, or between two ranks, silently changed the launch sequence.  The parity tests flip bits
// in-process through emavfi_debug_switches() (include/emavfi.h) instead.  None of them changes the packed layout (those are
// LayoutEnv in emavfi_api.hip: fixed per process, exported as emavfi_layout_tag(), verified against the blob header).
enum {
    SW_NO_CONV_FIRST = 1,        // EMAVFI_CONV_FIRST=0: pack_input + conv3x3<16,2,1> instead of conv_first
    SW_NO_FIRSTRING = 2,         // EMAVFI_CONV_FIRSTRING=0: conv_first + ring kernel instead of conv_ring_first
    SW_NO_HEAD = 4,              // EMAVFI_CONV_HEAD=0: motion_estimation.1 and .2 as two launches
    SW_NO_TAILFUSE = 8,          // EMAVFI_CONV_TAILFUSE=0: reconstruction.1 and .2 as two launches
    SW_NO_CONV_LIGHT = 16,       // EMAVFI_CONV_LIGHT=0: planar heads on conv3x3_persist16_kernel
    SW_NO_PERSISTENT_CONV = 32,  // EMAVFI_NO_PERSISTENT_CONV: tile-per-workgroup kernel where the persistent one is the default
    SW_NO_RING2 = 64,            // EMAVFI_CONV_RING2=0: conv_block_1 + conv_block_2 / motion_estimation.0 + .1 + .2 as separate launches
    SW_NO_POOLFUSE = 128,        // EMAVFI_CONV_POOLFUSE=0: context_encoding.2 stores its output and avg_pool_partial reads it back
    SW_RING_ONE_WG = 256,        // EMAVFI_RING_ONE_WG=1 (measurement, round 5): the persistent LDS-ring kernels launch ONE workgroup per CU instead
                                 // of two, leaving half of every CU's LDS to a kernel of another stream (the pipelined forward's pack kernel)
    SW_NO_RING_CHUNK = 512,      // EMAVFI_RING_CHUNK=0: the persistent ring kernels walk (strip, 45-row segment) items dealt round-robin instead of
                                 // one contiguous range of rows per workgroup (conv_ring.inl, RingWork)
};
unsigned emavfi_switches();

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function handle of ONE device: apply it once per (kernel, device).
// (a per-process once-flag left every device but the first one of a multi-device process with the 64 KiB default.)
// Thread-safe; a racing first call on a device merely sets the attribute twice.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};   // bit d: applied on device d (device ids >= 64: applied at every launch)
};
inline hipError_t set_lds_limit(PerDeviceOnce &o, const void *fn, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (o.done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) o.done.fetch_or(bit, std::memory_order_release);
    return e;
}

// compute units of the CURRENT device (persistent kernels size their grid by it); cached per device, thread-safe
int device_cu_count();

// feat_ext_conv1 fused with cat(frame1, frame2) (conv_first.inl; 16-bit types, mid_channels 64)
struct FirstParams;
int launch_conv_first_bf16(const FirstParams &p, hipStream_t s);
int launch_conv_first_f16(const FirstParams &p, hipStream_t s);
// conv_ring_first.inl: feat_ext_conv1 + conv_block_0 in one launch (p: the 64 -> 64 layer, ring == 2; p.in is not read)
int launch_conv_ringfirst_bf16(const FirstParams &fp, const ConvParams &p, hipStream_t s);
int launch_conv_ringfirst_f16(const FirstParams &fp, const ConvParams &p, hipStream_t s);
int launch_conv3x3_f32(const ConvParams &p, hipStream_t s);
int launch_conv3x3_bf16(const ConvParams &p, hipStream_t s);
int launch_conv3x3_f16(const ConvParams &p, hipStream_t s);
int launch_deform_f32(const DeformParams &p, hipStream_t s);
int launch_deform_bf16(const DeformParams &p, hipStream_t s);
int launch_deform_f16(const DeformParams &p, hipStream_t s);
// 16-bit dtypes at the reference width: the whole ModulatedDeformConvPack is one launch (deform_pack3.inl / deform_pack.inl)
#ifndef EMAVFI_PACK3
#define EMAVFI_PACK3 1   // 0: the round-2 kernel (deform_pack.inl) and its weight layout (A/B builds)
#endif
// shape served by deform_pack3_kernel: 65..67 input channels (4 k-groups + a 3-channel tail), 65..67 output channels
static inline bool deform_pack3_shape(int ck, int nf, int cin_real, int cout_real)
{
    return EMAVFI_PACK3 && ck == 80 && nf == 3 && cin_real > 64 && cin_real <= 67 && cout_real > 64 && cout_real <= 67;
}
// shape served by deform_f32w_kernel (fp32, LDS window): the reference width, input channels in (64, 72], outputs <= 80
#ifndef EMAVFI_F32W
#define EMAVFI_F32W 1   // 0: deform_kernel<float, 80, 3> (global gathers, 32x32x2 MFMAs) and its weight layout (A/B builds)
#endif
static inline bool deform_f32w_shape(int ck, int nf, int cin_real, int cout_real)
{
    return EMAVFI_F32W && ck == 80 && nf == 3 && cin_real > 64 && cin_real <= 68 && cout_real > 64 && cout_real <= 80;   // (K = 36 + 32 issued)
}
bool deform16_can_fuse_offset_conv(int ck, int nf, int cin_real, int off_ck, int off_nf);
