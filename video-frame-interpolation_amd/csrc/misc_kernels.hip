// HBM-bound helper kernels of libemavfi: weight packing, layout conversion at the NCHW
// boundary, global-average-pool + context folding, and the backward bilinear warp.
#include "common.h"
#include "misc_kernels.h"
#include <atomic>
#include <type_traits>
#include <cstddef>
#include <cstdlib>

int device_cu_count()
{
    static std::atomic<int> cache[64];  // zero-initialised; a racing first call just queries twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -1;
    if (dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
    if (dev < 64) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// ------------------------------------------------------------------------------------------
// Weight packing: OIHW fp32 -> [pass][chunk][tap][kg][nf][lane][16 B] in MFMA operand order.
// Lane (r, h) of fragment nf holds output channel pass*NF*32 + nf*32 + r and input channels
// chunk*CK + kg*CHKG + h*EPV + e (e < EPV).  Everything outside the real tensor packs as 0,
// which is what keeps the pad channels of every activation exactly zero.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int route_cout(int pos, int perm)
{
    // perm 1 = ModulatedDeformConvPack routing (ema_vfi.py:57-59): the 27 raw channels are
    // chunked (static offsets | mask | dynamic offsets); packed channel order is
    // [18 offsets = cat(first, third)] [9 mask].
    if (perm == 0) return pos;
    if (pos < 9) return pos;
    if (pos < 18) return pos + 9;
    if (pos < 27) return pos - 9;
    return 1 << 30;
}

template <typename T>
__global__ void pack_conv_kernel(const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ wp,
                                 float *__restrict__ bp, PackDesc d)
{
    constexpr int CHKG = DT<T>::CHKG, EPV = DT<T>::EPV;
    if (d.ring == 3) { d.ck = 64; }   // channels 64.. go to the tail below
    const int KG = d.ck / CHKG;
    const size_t total = (size_t)d.npass * d.nchunk * 9 * KG * d.nf * 64 * EPV;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int e = t % EPV; t /= EPV;
        const int lane = t % 64; t /= 64;
        const int n = t % d.nf; t /= d.nf;
        const int kg = t % KG; t /= KG;
        const int tap = t % 9; t /= 9;
        const int chunk = t % d.nchunk; t /= d.nchunk;
        const int pass = (int)t;
        int co = route_cout(pass * d.nf * 32 + n * 32 + (lane & 31), d.perm);
        int ci = chunk * d.ck + kg * CHKG + (lane >> 5) * EPV + e;
        if (d.mfma16) {   // same number of elements, regrouped: (kg, n) enumerates (k32, cout16 block) = KG * nf = (KG / 2) * (2 nf)
            const int flat = kg * d.nf + n, nb16 = 2 * d.nf, k32 = flat / nb16, blk = flat - k32 * nb16;
            co = route_cout(pass * d.nf * 32 + blk * 16 + (lane & 15), d.perm);
            ci = chunk * d.ck + k32 * 32 + (lane >> 4) * 8 + e;
        }
        int term = 0;
        if (d.x3) {   // virtual chunk 3 c + t -> real chunk c
            const int rc = chunk / 3;
            term = chunk - 3 * rc;
            ci = rc * d.ck + kg * CHKG + (lane >> 5) * EPV + e;
        }
        float v = 0.0f;
        if (co < d.cout && ci < d.cin_take && ci < d.ck * (d.x3 ? d.nchunk / 3 : d.nchunk)) v = w[((size_t)co * d.cin_raw + d.cin_off + ci) * 9 + tap];
        if (d.via_bf16) v = (float)(bf16_t)v;
        if (term == 1) v = v - (float)(T)v;   // the lo part of the weight
        wp[idx] = (T)v;
    }
    if (d.ring == 3 && EPV == 8) {
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)3 * 2 * 64 * 8; idx += (size_t)gridDim.x * blockDim.x) {
            size_t t = idx;
            const int e = t % 8; t /= 8;
            const int lane = t % 64; t /= 64;
            const int n = t % 2; t /= 2;
            const int j = (int)t;
            const int ts = 4 * j + 2 * (lane >> 5) + (e >> 2), ci = 64 + (e & 3), co = n * 32 + (lane & 31);
            float v = 0.0f;
            if (ts < 9 && ci < d.cin_take && co < d.cout) v = w[((size_t)co * d.cin_raw + d.cin_off + ci) * 9 + ts];
            wp[total + idx] = (T)v;
        }
    }
    if (d.first6 && EPV == 8) {   // a second copy behind the regular fragments: conv_first.inl's ten tap slots of 8
        for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)5 * 2 * 64 * 8; idx += (size_t)gridDim.x * blockDim.x) {
            size_t t = idx;
            const int e = t % 8; t /= 8;
            const int lane = t % 64; t /= 64;
            const int n = t % 2; t /= 2;
            const int kg = (int)t;
            const int tap = 2 * kg + (lane >> 5), co = n * 32 + (lane & 31);
            float v = 0.0f;
            if (tap < 9 && e < d.cin_take && co < d.cout) v = w[((size_t)co * d.cin_raw + d.cin_off + e) * 9 + tap];
            wp[total + idx] = (T)v;
        }
    }
    const int coutpad = d.npass * d.nf * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < coutpad; i += gridDim.x * blockDim.x) {
        const int co = route_cout(i, d.perm);
        float bv = (co < d.cout && bias) ? bias[co] : 0.0f;
        if (d.bias_f16) bv = (float)(half_t)bv;
        bp[i] = bv;
    }
}

// deform_pack3.inl layouts (always f16 elements; via_bf16 rounds to bf16 first).  Input channels 0..63 are four k-groups per
// tap; channels 64..66 of all nine taps form three im2col k-groups: K = 16 j + 8 h + e <-> tap slot 4 j + 2 h + (e >> 2),
// channel 64 + (e & 3) (slots 9..11 and channel 67 are zero).
//   pack3 = 1 (DCN):          [tap][kg 4][nf 2][lane][8] | third fragment as a table [tap][kg 4][row 0..3][h][8] (row 3 = zeros)
//                             | tail [j 3][nf 3][lane][8]
//   pack3 = 2 (offset_conv):  [tap][kg 4][lane][8] | tail [j 3][lane][8]
__global__ void pack_deform3_kernel(const float *__restrict__ w, const float *__restrict__ bias, half_t *__restrict__ wp,
                                    float *__restrict__ bp, PackDesc d)
{
    const int nA = d.pack3 == 1 ? 9 * 4 * 2 * 64 * 8 : 9 * 4 * 64 * 8;
    const int nB = d.pack3 == 1 ? 9 * 4 * 4 * 2 * 8 : 0;
    const int nC = d.pack3 == 1 ? 3 * 3 * 64 * 8 : 3 * 64 * 8;
    const int nfm = d.pack3 == 1 ? 2 : 1, nft = d.pack3 == 1 ? 3 : 1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nA + nB + nC; idx += gridDim.x * blockDim.x) {
        int co, ci, tap;
        bool real = true;
        if (idx < nA) {
            int t = idx;
            const int e = t % 8; t /= 8;
            const int lane = t % 64; t /= 64;
            const int n = t % nfm; t /= nfm;
            const int kg = t % 4; t /= 4;
            tap = t;
            co = n * 32 + (lane & 31);
            ci = kg * 16 + (lane >> 5) * 8 + e;
        } else if (idx < nA + nB) {
            int t = idx - nA;
            const int e = t % 8; t /= 8;
            const int hh = t % 2; t /= 2;
            const int row = t % 4; t /= 4;
            const int kg = t % 4; t /= 4;
            tap = t;
            co = 64 + row;
            real = row < 3;
            ci = kg * 16 + hh * 8 + e;
        } else {
            int t = idx - nA - nB;
            const int e = t % 8; t /= 8;
            const int lane = t % 64; t /= 64;
            const int n = t % nft; t /= nft;
            const int j = t;
            tap = 4 * j + 2 * (lane >> 5) + (e >> 2);
            co = n * 32 + (lane & 31);
            ci = 64 + (e & 3);
            real = tap < 9 && (e & 3) < 3;
            if (!real) tap = 0;
        }
        co = route_cout(co, d.perm);
        float v = 0.0f;
        if (real && co < d.cout && ci < d.cin_take) v = w[((size_t)co * d.cin_raw + d.cin_off + ci) * 9 + tap];
        if (d.via_bf16) v = (float)(bf16_t)v;
        wp[idx] = (half_t)v;
    }
    const int coutpad = d.npass * d.nf * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < coutpad; i += gridDim.x * blockDim.x) {
        const int co = route_cout(i, d.perm);
        float bv = (co < d.cout && bias) ? bias[co] : 0.0f;
        if (d.bias_f16) bv = (float)(half_t)bv;
        bp[i] = bv;
    }
}

// deform_f32w.inl layout (fp32, 67 -> <= 80 channels): [tap][half 2][cout block 5]{ group 0: [lane][4] | group 1: [lane][4] |
// leftover: [lane][1] }; lane (i = lane & 15, kb = lane >> 4) of a group holds W[16 c + i][36 half + 16 G + 4 kb + t] for the
// group's four MFMA steps t = 0..3 (step t contracts the channel set {16 G + 4 kb' + t}: the order in which a lane's 16-byte
// gather piece feeds the steps), the leftover step W[16 c + i][36 half + 32 + kb].
__global__ void pack_deform_f32w_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ wp,
                                        float *__restrict__ bp, PackDesc d)
{
    constexpr int CB = 64 * 4 * 2 + 64, HALF = 5 * CB, TAP = 2 * HALF;   // floats
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 9 * TAP; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int tap = t / TAP; t -= tap * TAP;
        const int half = t / HALF; t -= half * HALF;
        const int c = t / CB; t -= c * CB;
        int lane, ci;
        if (t < 512) { const int G = t / 256, r = t - G * 256; lane = r >> 2; ci = 36 * half + 16 * G + 4 * (lane >> 4) + (r & 3); }
        else { lane = t - 512; ci = 36 * half + 32 + (lane >> 4); }
        const int co = 16 * c + (lane & 15);
        if (!d.x3) {
            wp[idx] = (co < d.cout && ci < d.cin_take) ? w[((size_t)co * d.cin_raw + d.cin_off + ci) * 9 + tap] : 0.0f;
            continue;
        }
        // EMAVFI_F32X3: the same bytes hold f16 (hi, lo) pairs - a lane's four floats become {hi[4], lo[4]}, the leftover float {hi, lo}:
        // float slot r of the lane's quad (r = 0..3) holds half elements (2 r, 2 r + 1) of that 8-half vector
        half_t *hp = reinterpret_cast<half_t *>(wp + idx);
        auto wval = [&](int cin) { return (co < d.cout && cin < d.cin_take) ? w[((size_t)co * d.cin_raw + d.cin_off + cin) * 9 + tap] : 0.0f; };
        if (t < 512) {
            const int r = t & 3, cbase = ci - r;   // the quad's first channel
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * r + q;           // element of {hi0..hi3, lo0..lo3}
                const float v = wval(cbase + (e & 3));
                const half_t hi = (half_t)v;
                hp[q] = e < 4 ? hi : (half_t)(v - (float)hi);
            }
        } else {
            const float v = wval(ci);
            const half_t hi = (half_t)v;
            hp[0] = hi;
            hp[1] = (half_t)(v - (float)hi);
        }
    }
    const int coutpad = d.npass * d.nf * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < coutpad; i += gridDim.x * blockDim.x) bp[i] = (i < d.cout && bias) ? bias[i] : 0.0f;
}

int launch_pack_conv(const float *w, const float *bias, void *wp, float *bp, const PackDesc &d, int dtype, hipStream_t s)
{
    if (d.pack3 == 3) {
        pack_deform_f32w_kernel<<<64, 256, 0, s>>>(w, bias, (float *)wp, bp, d);
        return (int)hipGetLastError();
    }
    if (d.pack3) {
        pack_deform3_kernel<<<64, 256, 0, s>>>(w, bias, (half_t *)wp, bp, d);
        return (int)hipGetLastError();
    }
    if (dtype == 0)
        pack_conv_kernel<float><<<256, 256, 0, s>>>(w, bias, (float *)wp, bp, d);
    else if (dtype == 2)
        pack_conv_kernel<half_t><<<256, 256, 0, s>>>(w, bias, (half_t *)wp, bp, d);
    else
        pack_conv_kernel<bf16_t><<<256, 256, 0, s>>>(w, bias, (bf16_t *)wp, bp, d);
    return (int)hipGetLastError();
}

// raw fp32 copies the context kernels need: Linear weight/bias, the context half of
// motion_estimation.0.0.weight as [o][c][tap], its bias.
__global__ void pack_ctx_kernel(const float *__restrict__ lin_w, const float *__restrict__ lin_b,
                                const float *__restrict__ w9, const float *__restrict__ b9, float *__restrict__ dst, int m, int round16)
{
    // round16: autocast casts Linear's and the convolution's weight and bias to fp16 (EMAVFI_AMP16)
    const auto rq = [round16](float v) { return round16 ? (float)(half_t)v : v; };
    float *o_lw = dst, *o_lb = o_lw + (size_t)m * 4 * m, *o_w9 = o_lb + m, *o_b9 = o_w9 + (size_t)m * m * 9;
    const int n_lw = m * 4 * m, n_w9 = m * m * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lw + n_w9 + 2 * m; i += gridDim.x * blockDim.x) {
        if (i < n_lw) {   // Linear weight transposed to [c][o]: the threads of ctx_finish_kernel (one per o) read consecutive addresses
            const int c = i / m, o = i - c * m;
            o_lw[i] = rq(lin_w[(size_t)o * 4 * m + c]);
        } else if (i < n_lw + m) o_lb[i - n_lw] = rq(lin_b[i - n_lw]);
        else if (i < n_lw + m + n_w9) {   // context half of motion_estimation.0's weight as [c][o * 9 + tap]
            const int k = i - n_lw - m, c = k / (m * 9), rem = k - c * m * 9, o = rem / 9, tap = rem - o * 9;
            o_w9[k] = rq(w9[((size_t)o * 2 * m + m + c) * 9 + tap]);
        } else o_b9[i - n_lw - m - n_w9] = rq(b9[i - n_lw - m - n_w9]);
    }
}
int launch_pack_ctx(const float *lin_w, const float *lin_b, const float *w9, const float *b9, float *dst, int m, int round16,
                    hipStream_t s)
{
    pack_ctx_kernel<<<64, 256, 0, s>>>(lin_w, lin_b, w9, b9, dst, m, round16);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Layout conversion at the boundary.
// ------------------------------------------------------------------------------------------
// torch.cat([frame1, frame2], dim=1) (ema_vfi.py:112) fused with the NCHW -> channels-last pack.
template <typename T>
__global__ void pack_input_kernel(const float *__restrict__ f1, const float *__restrict__ f2, T *__restrict__ dst,
                                  int B, int C, int H, int W, int cpad)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        T *o = dst + i * cpad;
        if (cpad == 16) {  // the model's case (2*in_channels <= 16): one pixel = 16 channels, 16-byte stores
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                v[c] = 0.0f;
                if (c < C) v[c] = f1[(b * C + c) * plane + pix];
                else if (c < 2 * C) v[c] = f2[(b * C + (c - C)) * plane + pix];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) store4(o + 4 * q, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            continue;
        }
        for (int c = 0; c < cpad; ++c) {
            float v = 0.0f;
            if (c < C) v = f1[(b * C + c) * plane + pix];
            else if (c < 2 * C) v = f2[(b * C + (c - C)) * plane + pix];
            o[c] = (T)v;
        }
    }
}
int launch_pack_input(const float *f1, const float *f2, void *dst, int B, int C, int H, int W, int cpad, int dtype, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    if (dtype == 0) pack_input_kernel<float><<<grid, 256, 0, s>>>(f1, f2, (float *)dst, B, C, H, W, cpad);
    else if (dtype == 2) pack_input_kernel<half_t><<<grid, 256, 0, s>>>(f1, f2, (half_t *)dst, B, C, H, W, cpad);
    else pack_input_kernel<bf16_t><<<grid, 256, 0, s>>>(f1, f2, (bf16_t *)dst, B, C, H, W, cpad);
    return (int)hipGetLastError();
}

template <typename T>
__global__ void nchw_to_cl_kernel(const float *__restrict__ src, T *__restrict__ dst, int B, int C, int c0, int ctake, int H, int W, int ps)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        T *o = dst + i * ps;
        for (int c = 0; c < ps; ++c) o[c] = (T)(c < ctake ? src[(b * C + c0 + c) * plane + pix] : 0.0f);
    }
}
template <typename T>
__global__ void cl_to_nchw_kernel(const T *__restrict__ src, float *__restrict__ dst, int B, int C, int H, int W, int ps, int coff)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        const T *o = src + i * ps + coff;
        for (int c = 0; c < C; ++c) dst[(b * C + c) * plane + pix] = (float)o[c];
    }
}
// EMAVFI_F32X3: fp32 <-> two f16 halves per pixel
__global__ void nchw_to_cl_x3_kernel(const float *__restrict__ src, half_t *__restrict__ dst, int B, int C, int H, int W, int psh)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        half_t *o = dst + i * 2 * psh;
        for (int c = 0; c < psh; ++c) {
            const float v = c < C ? src[(b * C + c) * plane + pix] : 0.0f;
            const half_t hi = (half_t)v;
            o[c] = hi;
            o[psh + c] = (half_t)(v - (float)hi);
        }
    }
}
__global__ void cl_to_nchw_x3_kernel(const half_t *__restrict__ src, float *__restrict__ dst, int B, int C, int H, int W, int psh, int coff)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        const half_t *o = src + i * 2 * psh + coff;
        for (int c = 0; c < C; ++c) dst[(b * C + c) * plane + pix] = (float)o[c] + (float)o[psh + c];
    }
}
int launch_nchw_to_cl_x3(const float *src, void *dst, int B, int C, int H, int W, int ps_half, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    nchw_to_cl_x3_kernel<<<grid, 256, 0, s>>>(src, (half_t *)dst, B, C, H, W, ps_half);
    return (int)hipGetLastError();
}
int launch_cl_to_nchw_x3(const void *src, float *dst, int B, int C, int H, int W, int ps_half, int coff, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    cl_to_nchw_x3_kernel<<<grid, 256, 0, s>>>((const half_t *)src, dst, B, C, H, W, ps_half, coff);
    return (int)hipGetLastError();
}
__global__ void pack_input_x3_kernel(const float *__restrict__ f1, const float *__restrict__ f2, half_t *__restrict__ dst, int B, int C, int H, int W, int cpad)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        half_t *o = dst + i * 2 * cpad;
        if (cpad == 16) {   // the model's case: 16-byte stores (the scalar loop below cost 1.1 ms at B = 8 x 720p)
            float v[16], l[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                v[c] = 0.0f;
                if (c < C) v[c] = f1[(b * C + c) * plane + pix];
                else if (c < 2 * C) v[c] = f2[(b * C + (c - C)) * plane + pix];
                l[c] = v[c] - (float)(half_t)v[c];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                store4(o + 4 * q, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                store4(o + 16 + 4 * q, l[4 * q], l[4 * q + 1], l[4 * q + 2], l[4 * q + 3]);
            }
            continue;
        }
        for (int c = 0; c < cpad; ++c) {
            float v = 0.0f;
            if (c < C) v = f1[(b * C + c) * plane + pix];
            else if (c < 2 * C) v = f2[(b * C + (c - C)) * plane + pix];
            const half_t hi = (half_t)v;
            o[c] = hi;
            o[cpad + c] = (half_t)(v - (float)hi);
        }
    }
}
int launch_pack_input_x3(const float *f1, const float *f2, void *dst, int B, int C, int H, int W, int cpad, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    pack_input_x3_kernel<<<grid, 256, 0, s>>>(f1, f2, (half_t *)dst, B, C, H, W, cpad);
    return (int)hipGetLastError();
}

int launch_nchw_to_cl_sub(const float *src, void *dst, int B, int C, int c0, int ctake, int H, int W, int ps, int dtype, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    if (dtype == 0) nchw_to_cl_kernel<float><<<grid, 256, 0, s>>>(src, (float *)dst, B, C, c0, ctake, H, W, ps);
    else if (dtype == 2) nchw_to_cl_kernel<half_t><<<grid, 256, 0, s>>>(src, (half_t *)dst, B, C, c0, ctake, H, W, ps);
    else nchw_to_cl_kernel<bf16_t><<<grid, 256, 0, s>>>(src, (bf16_t *)dst, B, C, c0, ctake, H, W, ps);
    return (int)hipGetLastError();
}
int launch_nchw_to_cl(const float *src, void *dst, int B, int C, int H, int W, int ps, int dtype, hipStream_t s)
{
    return launch_nchw_to_cl_sub(src, dst, B, C, 0, C, H, W, ps, dtype, s);
}
int launch_cl_to_nchw(const void *src, float *dst, int B, int C, int H, int W, int ps, int coff, int dtype, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    if (dtype == 0) cl_to_nchw_kernel<float><<<grid, 256, 0, s>>>((const float *)src, dst, B, C, H, W, ps, coff);
    else if (dtype == 2) cl_to_nchw_kernel<half_t><<<grid, 256, 0, s>>>((const half_t *)src, dst, B, C, H, W, ps, coff);
    else cl_to_nchw_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t *)src, dst, B, C, H, W, ps, coff);
    return (int)hipGetLastError();
}

// EMAVFI_AMP16 keeps the fusion tensor twice: fp32 (what the fp32 DCN reads and writes) and fp16 (what the fp16
// convolutions read).  One thread = 4 channels of one pixel.
template <bool WIDEN>
__global__ void convert_cl_kernel(const void *__restrict__ src, void *__restrict__ dst, size_t npx, int ps_src, int ps_dst, int c0, int nc, int lo_off)
{
    const int groups = nc / 4;
    const size_t total = npx * groups;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i / groups;
        const int c = c0 + 4 * (int)(i - px * groups);
        if (WIDEN) {
            const f16x4 v = *reinterpret_cast<const f16x4 *>((const half_t *)src + px * ps_src + c);
            f32x4 o = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            if (lo_off > 0) {
                const f16x4 l = *reinterpret_cast<const f16x4 *>((const half_t *)src + px * ps_src + lo_off + c);
                o = f32x4{o[0] + (float)l[0], o[1] + (float)l[1], o[2] + (float)l[2], o[3] + (float)l[3]};
            }
            *reinterpret_cast<f32x4 *>((float *)dst + px * ps_dst + c) = o;
        } else {
            const f32x4 v = *reinterpret_cast<const f32x4 *>((const float *)src + px * ps_src + c);
            const f16x4 hi = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<f16x4 *>((half_t *)dst + px * ps_dst + c) = hi;
            if (lo_off > 0)
                *reinterpret_cast<f16x4 *>((half_t *)dst + px * ps_dst + lo_off + c) =
                    f16x4{(half_t)(v[0] - (float)hi[0]), (half_t)(v[1] - (float)hi[1]), (half_t)(v[2] - (float)hi[2]), (half_t)(v[3] - (float)hi[3])};
        }
    }
}
// EMAVFI_F32X3, the model's case of the narrow conversion: 16 fp32 channels of every pixel -> their f16 (hi, lo) halves, one thread per
// pixel, whole 16-byte stores (the generic kernel above ran four threads per pixel on 8-byte stores: 0.8 ms at B = 8 x 720p for 16 channels)
__global__ void split16_cl_kernel(const float *__restrict__ src, half_t *__restrict__ dst, size_t npx, int ps_src, int ps_dst, int c0, int lo_off)
{
    for (size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (size_t)gridDim.x * blockDim.x) {
        const float *i = src + px * ps_src + c0;
        half_t *o = dst + px * ps_dst + c0;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(i + 4 * q);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
        unsigned hi[8], lo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const half_t h0 = (half_t)v[2 * q], h1 = (half_t)v[2 * q + 1];
            const half_t l0 = (half_t)(v[2 * q] - (float)h0), l1 = (half_t)(v[2 * q + 1] - (float)h1);
            hi[q] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            lo[q] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<uint4 *>(o + 8 * q) = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
            *reinterpret_cast<uint4 *>(o + lo_off + 8 * q) = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
        }
    }
}
int launch_convert_cl(const void *src, void *dst, size_t npx, int ps_src, int ps_dst, int c0, int nc, int widen, hipStream_t s, int lo_off)
{
    if (!widen && lo_off > 0 && nc == 16 && (c0 & 7) == 0 && (lo_off & 7) == 0 && (ps_dst & 7) == 0 && (ps_src & 3) == 0) {
        const int grid16 = (int)std::min<size_t>((npx + 255) / 256, 65535 * 4);
        split16_cl_kernel<<<grid16, 256, 0, s>>>((const float *)src, (half_t *)dst, npx, ps_src, ps_dst, c0, lo_off);
        return (int)hipGetLastError();
    }
    const int grid = (int)std::min<size_t>((npx * (nc / 4) + 255) / 256, 65535 * 4);
    if (widen) convert_cl_kernel<true><<<grid, 256, 0, s>>>(src, dst, npx, ps_src, ps_dst, c0, nc, lo_off);
    else convert_cl_kernel<false><<<grid, 256, 0, s>>>(src, dst, npx, ps_src, ps_dst, c0, nc, lo_off);
    return (int)hipGetLastError();
}

// offset [B,18,H,W] + mask [B,9,H,W] (torchvision argument layout) -> om [px][32]
__global__ void om_from_nchw_kernel(const float *__restrict__ off, const float *__restrict__ msk, float *__restrict__ om,
                                    int B, int H, int W)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        float *o = om + i * 32;
        for (int c = 0; c < 18; ++c) o[c] = off[(b * 18 + c) * plane + pix];
        for (int c = 0; c < 9; ++c) o[18 + c] = msk[(b * 9 + c) * plane + pix];
        for (int c = 27; c < 32; ++c) o[c] = 0.0f;
    }
}
int launch_om_from_nchw(const float *off, const float *msk, float *om, int B, int H, int W, hipStream_t s)
{
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    om_from_nchw_kernel<<<grid, 256, 0, s>>>(off, msk, om, B, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// AdaptiveAvgPool2d(1) -> Flatten -> Linear (ema_vfi.py:83-85) and the folding of the
// broadcast context into motion_estimation.0's bias (ema_vfi.py:124 concat eliminated).
// Deterministic: fixed partition, fixed summation order, no atomics - shards of a batch give
// bit-identical results on any GPU.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool_partial_kernel(const T *__restrict__ src, float *__restrict__ part, int npix,
                                                           int cp, int ps, int nparts)
{
    const int b = blockIdx.y, part_i = blockIdx.x;
    const int per = (npix + nparts - 1) / nparts;
    const int p0 = part_i * per, p1 = min(npix, p0 + per);
    // 16-byte path (the model's case: cp = 256 channels): a thread sums VEC consecutive channels, cp / VEC threads cover a pixel,
    // 256 * VEC / cp pixels are in flight per iteration (the scalar path below issued one 2-byte load per thread and iteration:
    // 112 us for 236 MB); the pixel lanes are then combined through the LDS in a fixed order
    constexpr int VEC = 16 / (int)sizeof(T);
    if (cp % VEC == 0 && ps % VEC == 0 && 256 % (cp / VEC) == 0 && cp / VEC <= 256) {
        __shared__ float redv[256 * VEC];
        typedef __attribute__((ext_vector_type(VEC))) T vecT;
        const int tpp = cp / VEC, nlv = 256 / tpp;
        const int cv = threadIdx.x % tpp, plv = threadIdx.x / tpp;
        float a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 0.0f;
        for (int px = p0 + plv; px < p1; px += nlv) {
            const vecT v = *reinterpret_cast<const vecT *>(src + ((size_t)b * npix + px) * ps + cv * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += (float)v[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) redv[(plv * tpp + cv) * VEC + e] = a[e];
        __syncthreads();
        for (int c = threadIdx.x; c < cp; c += 256) {
            float t = 0.0f;
            for (int l = 0; l < nlv; ++l) t += redv[l * cp + c];
            part[((size_t)b * nparts + part_i) * cp + c] = t;
        }
        return;
    }
    __shared__ float red[256];
    const int nl = 256 / cp;  // pixel lanes
    const int c = threadIdx.x % cp, pl = threadIdx.x / cp;
    float s = 0.0f;
    for (int px = p0 + pl; px < p1; px += nl) s += (float)src[((size_t)b * npix + px) * ps + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0) {
        for (int l = 1; l < nl; ++l) s += red[l * cp + c];
        part[((size_t)b * nparts + part_i) * cp + c] = s;
    }
}
int launch_pool_partial(const void *src, float *part, int B, int npix, int cp, int ps, int nparts, int dtype, hipStream_t s)
{
    dim3 grid(nparts, B);
    if (dtype == 0) pool_partial_kernel<float><<<grid, 256, 0, s>>>((const float *)src, part, npix, cp, ps, nparts);
    else if (dtype == 2) pool_partial_kernel<half_t><<<grid, 256, 0, s>>>((const half_t *)src, part, npix, cp, ps, nparts);
    else pool_partial_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t *)src, part, npix, cp, ps, nparts);
    return (int)hipGetLastError();
}

// One block of 1024 threads per sample (m <= 64: 4m divides 256).  ctxw = [lin_w TRANSPOSED 4m x m][lin_b m][w9c as [c][o * 9 + tap]][b9 m]
// (pack_ctx_kernel): every loop below reads consecutive addresses from consecutive threads.
// Writes ctx[b][m] and the border-class bias table R[b][16][coutpad]:
//   R[cls][o] = b9[o] + sum over taps (ky,kx) inside the image for that class of
//               sum_c W9[o, m + c, ky, kx] * ctx[c]
// cls = ym*4 + xm; bit0 of ym: row y-1 exists, bit1: row y+1 exists (same for xm / columns).
// The kernel is a chain of four dependent phases of a few hundred KB each - latency, not bandwidth: round 4 spreads every phase's
// sum over all 1024 threads (partial sums meet in LDS, added in a fixed order: deterministic), which cut the loads a thread issues
// one after another from 192 to 44 in the longest phase (37 -> ~15 us at B = 8).
constexpr int kCtxThreads = 1024;
__global__ __launch_bounds__(kCtxThreads) void ctx_finish_kernel(const float *__restrict__ part, const float *__restrict__ ctxw,
                                                                 float *__restrict__ ctx_out, float *__restrict__ table,
                                                                 int m, int cp, int nparts, int npix, int coutpad, int round16, int lo_off)
{
    // round16 (EMAVFI_AMP16): the pooled mean and the Linear output are fp16 tensors under autocast
    const auto rq = [round16](float v) { return round16 ? (float)(half_t)v : v; };
    constexpr int NT = kCtxThreads;
    extern __shared__ float sm[];
    float *mean = sm, *ctx = sm + 4 * m, *tsum = ctx + m, *red = tsum + 9 * m;  // tsum [m][9]; red: max(NT, 3 * 9 m) floats
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *lw = ctxw, *lb = lw + (size_t)m * 4 * m, *w9 = lb + m, *b9 = w9 + (size_t)m * m * 9;
    {   // ---- mean over the partial sums: thread (c, g) adds parts g, g + G, ...
        const int C4 = 4 * m, G = NT / C4, c = tid % C4, g = tid / C4;
        float a = 0.0f;
        if (g < G)
            for (int q = g; q < nparts; q += G) {
                a += part[((size_t)b * nparts + q) * cp + c];
                if (lo_off > 0) a += part[((size_t)b * nparts + q) * cp + lo_off + c];   // EMAVFI_F32X3: the lo halves' sums
            }
        if (g < G) red[g * C4 + c] = a;
        __syncthreads();
        if (tid < C4) {
            float s = 0.0f;
            for (int q = 0; q < G; ++q) s += red[q * C4 + tid];
            mean[tid] = rq(s / (float)npix);
        }
        __syncthreads();
    }
    {   // ---- Linear (ema_vfi.py:91, 121): thread (o, g) sums a slice of the 4m inputs from the [c][o] copy (coalesced), four chains
        const int G = NT / m, o = tid % m, g = tid / m, per = (4 * m + G - 1) / G;
        if (g < G) {
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int c1 = (g + 1) * per < 4 * m ? (g + 1) * per : 4 * m;
            int c = g * per;
            for (; c + 4 <= c1; c += 4)
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = fmaf(lw[(size_t)(c + u) * m + o], mean[c + u], a[u]);
            for (; c < c1; ++c) a[0] = fmaf(lw[(size_t)c * m + o], mean[c], a[0]);
            red[g * m + o] = (a[0] + a[1]) + (a[2] + a[3]);
        }
        __syncthreads();
        if (tid < m) {
            float t = lb[tid];
            for (int q = 0; q < G; ++q) t += red[q * m + tid];
            t = rq(t);
            ctx[tid] = t;
            ctx_out[(size_t)b * m + tid] = t;
        }
        __syncthreads();
    }
    {   // ---- tsum[k = o * 9 + tap] = sum_c w9[c][k] ctx[c]: blocks of KB outputs, thread (k, cg) sums a third (or less) of the channels
        const int K9 = 9 * m, KB = K9 < 288 ? K9 : 288;
        int CG = NT / KB; CG = CG > m ? m : CG; CG = CG > 3 ? 3 : CG;
        const int per = (m + CG - 1) / CG, kk = tid % KB, cg = tid / KB;
        for (int k0 = 0; k0 < K9; k0 += KB) {
            const int k = k0 + kk;
            if (cg < CG && k < K9) {
                float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const int c1 = (cg + 1) * per < m ? (cg + 1) * per : m;
                int c = cg * per;
                for (; c + 4 <= c1; c += 4)
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = fmaf(w9[(size_t)(c + u) * K9 + k], ctx[c + u], a[u]);
                for (; c < c1; ++c) a[0] = fmaf(w9[(size_t)c * K9 + k], ctx[c], a[0]);
                red[cg * K9 + k] = (a[0] + a[1]) + (a[2] + a[3]);
            }
        }
        __syncthreads();
        for (int k = tid; k < K9; k += NT) {
            float t = 0.0f;
            for (int q = 0; q < CG; ++q) t += red[q * K9 + k];
            tsum[k] = t;
        }
        __syncthreads();
    }
    for (int k = tid; k < 16 * coutpad; k += NT) {
        const int cls = k / coutpad, o = k - cls * coutpad;
        float s = 0.0f;
        if (o < m) {
            const int ym = cls >> 2, xm = cls & 3;
            s = b9[o];
            for (int ky = 0; ky < 3; ++ky) {
                if ((ky == 0 && !(ym & 1)) || (ky == 2 && !(ym & 2))) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    if ((kx == 0 && !(xm & 1)) || (kx == 2 && !(xm & 2))) continue;
                    s += tsum[o * 9 + ky * 3 + kx];
                }
            }
        }
        table[((size_t)b * 16 + cls) * coutpad + o] = s;
    }
}
int launch_ctx_finish(const float *part, const float *ctxw, float *ctx_out, float *table, int B, int m, int cp, int nparts,
                      int npix, int coutpad, int round16, hipStream_t s, int lo_off)
{
    const int nred = kCtxThreads > 27 * m ? kCtxThreads : 27 * m;
    const size_t sh = (size_t)(4 * m + m + 9 * m + nred) * sizeof(float);
    ctx_finish_kernel<<<B, kCtxThreads, sh, s>>>(part, ctxw, ctx_out, table, m, cp, nparts, npix, coutpad, round16, lo_off);
    return (int)hipGetLastError();
}

// The forward's LAST launch: the blob's header against what the call expects (misc_kernels.h, BlobGuard).  A match costs one
// 36-byte compare per thread of a 64-workgroup grid; a foreign blob (another library version / model / dtype / layout-switch
// setting, or no header at all) overwrites the frame with NaN - the forward cannot return a code for device-resident bytes
// without synchronising, but it must not hand back plausible garbage either.  (Not through the arithmetic: ReLU - v_max_f32 -
// turns a poisoned activation into 0.)
__global__ __launch_bounds__(256) void blob_guard_kernel(const BlobGuard guard, float *__restrict__ out, size_t n)
{
    const uint32_t *got = (const uint32_t *)guard.hdr, *want = (const uint32_t *)&guard.expect;
    bool foreign = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) foreign |= got[i] != want[i];   // magic, version, header_bytes, in, mid, nb, dtype, layout_tag
    if (!foreign) return;
    const float poison = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = poison;
}
int launch_blob_guard(const BlobGuard &guard, float *out, size_t n, hipStream_t s)
{
    blob_guard_kernel<<<64, 256, 0, s>>>(guard, out, n);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Blob header (misc_kernels.h, BlobHeader): written behind the pack kernels on the same stream, then the payload's checksum is
// added into it.  Integer addition commutes, so the atomics leave a deterministic value.
// ------------------------------------------------------------------------------------------
__global__ void blob_header_kernel(BlobHeader *dst, const BlobHeader h)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *dst = h;
    // (the rest of the 256 header bytes is zeroed by the host-side memset in front of this launch)
}
__global__ __launch_bounds__(256) void blob_checksum_kernel(const uint32_t *__restrict__ payload, size_t nwords, unsigned long long *sum)
{
    unsigned long long a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
        a += ((unsigned long long)payload[i] + 0x9E3779B9ull) * (2ull * i + 1ull);
    __shared__ unsigned long long part[256];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(sum, part[0]);
}
int launch_blob_seal(void *blob, const BlobHeader &h, hipStream_t s)
{
    if (hipMemsetAsync(blob, 0, kBlobHeaderBytes, s) != hipSuccess) return (int)hipGetLastError();
    BlobHeader w = h;
    w.checksum = 0;
    blob_header_kernel<<<1, 64, 0, s>>>((BlobHeader *)blob, w);
    const size_t nwords = (size_t)(h.total_bytes - kBlobHeaderBytes) / 4;
    blob_checksum_kernel<<<64, 256, 0, s>>>((const uint32_t *)((const char *)blob + kBlobHeaderBytes), nwords,
                                            (unsigned long long *)((char *)blob + offsetof(BlobHeader, checksum)));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Backward bilinear warp = EMA_VFI.warp (ema_vfi.py:149-171).  The grid the reference builds on
// the host (:153-160), the add (:162), the normalisation (:165-166, true fp32 division) and
// ATen's grid_sampler_2d (un-normalise, floor, four masked taps) are one kernel; the operation
// order of the coordinate math is kept exactly, because sampling directly at x+flow differs from
// the reference by up to 5e-4 at 720p.  Compiled without fast-math.
// ------------------------------------------------------------------------------------------
struct WarpTap { int o00, o01, o10, o11; float nw, ne, sw, se; int xa, xb, ya, yb; };

__device__ __forceinline__ WarpTap warp_tap(int x, int y, float fx, float fy, int H, int W, float wden, float hden)
{
    const float vx = (float)x + fx, vy = (float)y + fy;
    const float gx = 2.0f * vx / wden - 1.0f;
    const float gy = 2.0f * vy / hden - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = ix - xw, e = 1.0f - w, n = iy - yn, s = 1.0f - n;
    // range tests in float: a finite coordinate far outside fails all of them -> 0.  A NaN / infinite coordinate (NaN or
    // infinite flow, or a flow so large that 2 * v overflows: same operation order as the reference, so the same cases)
    // yields NaN in every channel, as ATen's CPU grid_sample does (its corner weights are NaN - NaN / inf - inf there and
    // 0 * NaN = NaN): ema_vfi.py:169 on the CPU forward, which is the oracle this path is held to.
    const bool bad = !(fabsf(ix) <= 3.402823466e38f) || !(fabsf(iy) <= 3.402823466e38f);
    const bool x0 = xw >= 0.0f && xw <= (float)(W - 1), x1 = xw + 1.0f >= 0.0f && xw + 1.0f <= (float)(W - 1);
    const bool y0 = yn >= 0.0f && yn <= (float)(H - 1), y1 = yn + 1.0f >= 0.0f && yn + 1.0f <= (float)(H - 1);
    const int xi = x0 ? (int)xw : (x1 ? (int)xw : 0), yi = y0 ? (int)yn : (y1 ? (int)yn : 0);
    const int xa = max(xi, 0), xb = min(xi + 1, W - 1), ya = max(yi, 0), yb = min(yi + 1, H - 1);
    WarpTap t;
    t.o00 = ya * W + xa; t.o01 = ya * W + xb; t.o10 = yb * W + xa; t.o11 = yb * W + xb;
    t.xa = xa; t.xb = xb; t.ya = ya; t.yb = yb;
    const float qnan = __builtin_nanf("");
    t.nw = bad ? qnan : ((y0 && x0) ? s * e : 0.0f);
    t.ne = bad ? qnan : ((y0 && x1) ? s * w : 0.0f);
    t.sw = bad ? qnan : ((y1 && x0) ? n * e : 0.0f);
    t.se = bad ? qnan : ((y1 && x1) ? n * w : 0.0f);
    return t;
}
__device__ __forceinline__ float warp_sample(const float *__restrict__ p, const WarpTap &t)
{
    // accumulation order nw, ne, sw, se as ATen's CPU kernel
    float a = p[t.o00] * t.nw;
    a += p[t.o01] * t.ne;
    a += p[t.o10] * t.sw;
    a += p[t.o11] * t.se;
    return a;
}

// NCHW -> NCHW (the C-ABI emavfi_warp; 32 algorithmic bytes per pixel at C = 3).
// One thread = 4 consecutive pixels of a row: 16-byte flow loads and output stores.
template <bool VEC4>
__global__ __launch_bounds__(256) void warp_nchw_kernel(const float *__restrict__ frame2, const float *__restrict__ flow,
                                                        float *__restrict__ out, int B, int C, int H, int W)
{
    const size_t plane = (size_t)H * W;
    const float wden = (float)max(W - 1, 1), hden = (float)max(H - 1, 1);
    constexpr int V = VEC4 ? 4 : 1;
    const size_t nitem = (size_t)B * plane / V;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < nitem; it += (size_t)gridDim.x * blockDim.x) {
        const size_t i = it * V;
        const size_t b = i / plane, pix = i - b * plane;
        const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
        float fx[V], fy[V];
        if (VEC4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(flow + (b * 2) * plane + pix);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(flow + (b * 2 + 1) * plane + pix);
#pragma unroll
            for (int k = 0; k < V; ++k) { fx[k] = a[k]; fy[k] = c[k]; }
        } else {
            fx[0] = flow[(b * 2) * plane + pix];
            fy[0] = flow[(b * 2 + 1) * plane + pix];
        }
        WarpTap t[V];
#pragma unroll
        for (int k = 0; k < V; ++k) t[k] = warp_tap(x + k, y, fx[k], fy[k], H, W, wden, hden);
        for (int c = 0; c < C; ++c) {
            const float *p = frame2 + (b * C + c) * plane;
            float v[V];
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = warp_sample(p, t[k]);
            float *o = out + (b * C + c) * plane + pix;
            if (VEC4) *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
            else o[0] = v[0];
        }
    }
}
// Tiled variant (the one emavfi_warp launches when W % 4 == 0 and C == 3): a 256-thread block
// owns a 32x64 pixel tile.  The frame2 window the tile can reach with |flow| <= R = 8 px
// (tile + R each side + 1 for the bilinear neighbour: 49 rows x 84 columns x 3 planes = 48 KiB)
// is brought into LDS by global->LDS DMA in 16-byte pieces (rows outside the image read a zero
// page), so HBM/L2 traffic is coalesced full lines instead of 12 scattered 4-byte gathers per
// pixel; the four taps are then read from LDS.  A pixel whose taps leave the window (|flow| > R)
// gathers from global memory as the simple kernel does - same arithmetic, same result.
__device__ const float g_zero_page[64] = {0};

// T = void: NCHW fp32 planes (emavfi_warp).  T = float / bf16_t: the forward's fused variant, which
// writes channels [coff, ps) of the channels-last fusion buffer (warped RGB, then zero padding).
// TH = tile rows (32: a 49-row window, 48 KiB, three workgroups per CU).  Round 6 measured TH = 16 (33 KiB, four per CU) for the forward's
// variant: 66.4 against 64.0 us - the kernel is bound by the LDS-DMA ingest of its window (46 KB per 2 048 pixels = 1.87 x the tile,
// 169 MB per launch through a path that takes ~ 6.4 TB/s chip-wide) and by ~ 100 issued instructions per pixel, not by occupancy;
// smaller tiles only raise the halo share (profiles/r06_experiments_that_lost.txt)
template <int C, typename T, int TH = 32>
__global__ __launch_bounds__(256) void warp_tiled_kernel(const float *__restrict__ frame2, const float *__restrict__ flow,
                                                         float *__restrict__ out, int B, int H, int W, void *cl_dst, int ps,
                                                         int coff, void *cl_dst16 = nullptr, int ps16 = 0)
{
    constexpr int TW = 64, R = 8, WR = TH + 2 * R + 1, WC = 84, PCS = WC / 4;
    constexpr int NPIECE = C * WR * PCS, NINST = (NPIECE + 63) / 64;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    __shared__ __attribute__((aligned(16))) float win[NINST * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the
    // XCD group); give each group a contiguous run of tiles so x-neighbours, which share halo
    // columns, hit the same XCD's L2.  Bijective for any grid size; placement only affects speed.
    const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH, nwg = gridDim.x;
    const int grp = blockIdx.x & 7, kk = blockIdx.x >> 3, qq = nwg >> 3, rr = nwg & 7;
    const int wg = (grp < rr ? grp * (qq + 1) : rr * (qq + 1) + (grp - rr) * qq) + kk;
    const int b = wg / (ntx * nty), trem = wg - b * (ntx * nty);
    const int ty = (trem / ntx) * TH, tx = (trem - (trem / ntx) * ntx) * TW;
    const int wy0 = ty - R, wx0 = tx - R;  // tx is a multiple of 64, so wx0 is 16-byte aligned when W % 4 == 0
    const size_t plane = (size_t)H * W;
    const float *src_b = frame2 + (size_t)b * C * plane;
#pragma unroll
    for (int i = 0; i < (NINST + 3) / 4; ++i) {
        const int j = i * 4 + wave;
        if (j < NINST) {
            const int idx = j * 64 + lane;
            const int ch = idx / (WR * PCS), rem = idx - ch * (WR * PCS);
            const int row = rem / PCS, pc = rem - row * PCS;
            const int gy = wy0 + row, gx = wx0 + 4 * pc;
            const bool ok = idx < NPIECE && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const float *src = ok ? src_b + (size_t)ch * plane + (size_t)gy * W + gx : g_zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(win + j * 256), 16, 0, 0);
        }
    }
    const float wden = (float)max(W - 1, 1), hden = (float)max(H - 1, 1);
    if constexpr (std::is_same<T, bf16_t>::value || std::is_same<T, half_t>::value) {
        if ((ps == 8 || ps == 4) && coff == 0) {
            // The forward's 16-bit variant (compact tail of the fusion input: 8 channels = 16 bytes per pixel, or - round 6 - 4 channels =
            // 8 bytes, which is what the forward passes: 22 % fewer bytes moved): a wave takes whole ROWS of the
            // tile, lane = column.  A store instruction then writes 1 KiB of consecutive addresses (the 4-pixels-per-thread mapping
            // below wrote 16 bytes of every 64: four partial passes over each line), the corner reads of a wave are consecutive dwords
            // of the window (conflict-free for a smooth flow; lanes four pixels apart were a four-way bank conflict) and the flow loads
            // are 256 contiguous bytes per instruction.  Same warp_tap / warp_sample arithmetic per pixel: identical results.
            constexpr int NR = TH / 4;
            float gx[NR], gy[NR];
            const int x = tx + lane;
#pragma unroll
            for (int k = 0; k < NR; ++k) {   // while the DMA is in flight
                const int y = ty + wave + 4 * k;
                const bool ok = y < H && x < W;
                const size_t pix = (size_t)(ok ? y : 0) * W + (ok ? x : 0);
                gx[k] = flow[((size_t)b * 2) * plane + pix];
                gy[k] = flow[((size_t)b * 2 + 1) * plane + pix];
            }
            // the window is filled by LDS-DMA, which hipcc drains (vmcnt(0)) ahead of the barrier today; pinned here so that a compiler
            // that moves its wait behind the barrier cannot let a wave read another wave's unlanded window.  Free: the
            // flow loads were issued after the DMA and return in order.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            typedef __attribute__((ext_vector_type(8))) T vec8;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int y = ty + wave + 4 * k;
                if (y >= H || x >= W) continue;
                const WarpTap t = warp_tap(x, y, gx[k], gy[k], H, W, wden, hden);
                const bool inside = t.ya >= wy0 && t.yb <= wy0 + WR - 1 && t.xa >= wx0 && t.xb <= wx0 + WC - 1;
                float v[C];
                if (inside) {
                    const int l00 = (t.ya - wy0) * WC + (t.xa - wx0), l01 = (t.ya - wy0) * WC + (t.xb - wx0);
                    const int l10 = (t.yb - wy0) * WC + (t.xa - wx0), l11 = (t.yb - wy0) * WC + (t.xb - wx0);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float *p = win + c * (WR * WC);
                        float a = p[l00] * t.nw;
                        a += p[l01] * t.ne;
                        a += p[l10] * t.sw;
                        a += p[l11] * t.se;
                        v[c] = a;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) v[c] = warp_sample(src_b + (size_t)c * plane, t);
                }
                const T z = (T)0.0f;
                T *o = reinterpret_cast<T *>(cl_dst) + ((size_t)b * plane + (size_t)y * W + x) * ps;
                if (ps == 4) {
                    typedef __attribute__((ext_vector_type(4))) T vec4;
                    *reinterpret_cast<vec4 *>(o) = vec4{(T)v[0], C > 1 ? (T)v[C > 1 ? 1 : 0] : z, C > 2 ? (T)v[C > 2 ? 2 : 0] : z, z};
                } else {
                    *reinterpret_cast<vec8 *>(o) = vec8{(T)v[0], C > 1 ? (T)v[C > 1 ? 1 : 0] : z, C > 2 ? (T)v[C > 2 ? 2 : 0] : z, z, z, z, z, z};
                }
            }
            return;
        }
    }
    // flow for this thread's items (4 pixels each; two per thread for the 32-row tile) while the DMA is in flight
    constexpr int NI = TH * TW / 4 / 256;
    f32x4 fx[NI], fy[NI];
    int iy[NI], ix[NI];
    bool live[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int item = tid + 256 * k;
        iy[k] = ty + item / (TW / 4);
        ix[k] = tx + 4 * (item % (TW / 4));
        live[k] = iy[k] < H && ix[k] < W;
        const size_t pix = (size_t)(live[k] ? iy[k] : 0) * W + (live[k] ? ix[k] : 0);
        fx[k] = *reinterpret_cast<const f32x4 *>(flow + ((size_t)b * 2) * plane + pix);
        fy[k] = *reinterpret_cast<const f32x4 *>(flow + ((size_t)b * 2 + 1) * plane + pix);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed before the barrier, whatever hipcc does with its own wait
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        if (!live[k]) continue;
        float v[C][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const WarpTap t = warp_tap(ix[k] + q, iy[k], fx[k][q], fy[k][q], H, W, wden, hden);
            const bool inside = t.ya >= wy0 && t.yb <= wy0 + WR - 1 && t.xa >= wx0 && t.xb <= wx0 + WC - 1;
            if (inside) {
                const int l00 = (t.ya - wy0) * WC + (t.xa - wx0), l01 = (t.ya - wy0) * WC + (t.xb - wx0);
                const int l10 = (t.yb - wy0) * WC + (t.xa - wx0), l11 = (t.yb - wy0) * WC + (t.xb - wx0);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float *p = win + c * (WR * WC);
                    float a = p[l00] * t.nw;
                    a += p[l01] * t.ne;
                    a += p[l10] * t.sw;
                    a += p[l11] * t.se;
                    v[c][q] = a;
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) v[c][q] = warp_sample(src_b + (size_t)c * plane, t);
            }
        }
        const size_t pix = (size_t)iy[k] * W + ix[k];
        if constexpr (std::is_void<T>::value) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                *reinterpret_cast<f32x4 *>(out + ((size_t)b * C + c) * plane + pix) = f32x4{v[c][0], v[c][1], v[c][2], v[c][3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T *o = reinterpret_cast<T *>(cl_dst) + ((size_t)b * plane + pix + q) * ps + coff;
                if constexpr (std::is_same<T, float>::value) {
                    // EMAVFI_AMP16: the fp16 rounding of the same values into channels [coff, ps16) of a second (fp16) fusion tensor, in the
                    // same launch (was a conversion pass of its own: fusion_round_warped)
                    if (cl_dst16) {
                        half_t *o16 = reinterpret_cast<half_t *>(cl_dst16) + ((size_t)b * plane + pix + q) * ps16 + coff;
                        if (ps16 - coff == 16 && (coff & 7) == 0 && (ps16 & 7) == 0) {   // mid_channels 64: channels 64..79 as two 16-byte stores
                            typedef __attribute__((ext_vector_type(8))) half_t h8;
                            const half_t z = (half_t)0.0f;
                            *reinterpret_cast<h8 *>(o16) = h8{(half_t)v[0][q], C > 1 ? (half_t)v[C > 1 ? 1 : 0][q] : z, C > 2 ? (half_t)v[C > 2 ? 2 : 0][q] : z, z, z, z, z, z};
                            *reinterpret_cast<h8 *>(o16 + 8) = h8{z, z, z, z, z, z, z, z};
                        } else {
#pragma unroll
                            for (int c = 0; c < C; ++c) o16[c] = (half_t)v[c][q];
                            for (int c = C; c < ps16 - coff; ++c) o16[c] = (half_t)0.0f;
                        }
                    }
                }
                if (ps - coff == 16 && (coff & 7) == 0) {  // mid_channels 64: channels 64..79, aligned
                    store4(o, v[0][q], C > 1 ? v[C > 1 ? 1 : 0][q] : 0.0f, C > 2 ? v[C > 2 ? 2 : 0][q] : 0.0f, 0.0f);
                    store4(o + 4, 0.0f, 0.0f, 0.0f, 0.0f);
                    store4(o + 8, 0.0f, 0.0f, 0.0f, 0.0f);
                    store4(o + 12, 0.0f, 0.0f, 0.0f, 0.0f);
                    continue;
                }
                if constexpr (sizeof(T) == 2) {
                    if (ps == 8 && coff == 0) {  // compact 8-channel tail of the fusion input: one 16-byte store per pixel
                        typedef __attribute__((ext_vector_type(8))) T vec8;
                        const T z = (T)0.0f;
                        *reinterpret_cast<vec8 *>(o) = vec8{(T)v[0][q], C > 1 ? (T)v[C > 1 ? 1 : 0][q] : z, C > 2 ? (T)v[C > 2 ? 2 : 0][q] : z, z, z, z, z, z};
                        continue;
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] = (T)v[c][q];
                for (int c = C; c < ps - coff; ++c) o[c] = (T)0.0f;
            }
        }
    }
}

int launch_warp_nchw(const float *frame2, const float *flow, float *out, int B, int C, int H, int W, hipStream_t s)
{
    const size_t n = (size_t)B * H * W;
    if ((W & 3) == 0 && C == 3) {
        const int nwg = ((W + 63) / 64) * ((H + 31) / 32) * B;
        warp_tiled_kernel<3, void><<<nwg, 256, 0, s>>>(frame2, flow, out, B, H, W, nullptr, 0, 0);
        return (int)hipGetLastError();
    }
    if ((W & 3) == 0) {
        const int grid = (int)std::min<size_t>((n / 4 + 255) / 256, 256 * 32);
        warp_nchw_kernel<true><<<grid, 256, 0, s>>>(frame2, flow, out, B, C, H, W);
    } else {
        const int grid = (int)std::min<size_t>((n + 255) / 256, 256 * 32);
        warp_nchw_kernel<false><<<grid, 256, 0, s>>>(frame2, flow, out, B, C, H, W);
    }
    return (int)hipGetLastError();
}

// Forward-path variant: writes the warped RGB straight into channels [coff, ps) of the
// channels-last fusion buffer (the torch.cat of ema_vfi.py:134 never materialises); channels
// past coff + C are the zero padding.
template <typename T>
__global__ __launch_bounds__(256) void warp_fused_kernel(const float *__restrict__ frame2, const float *__restrict__ flow,
                                                         T *__restrict__ dst, int B, int C, int H, int W, int ps, int coff,
                                                         half_t *__restrict__ dst16 = nullptr, int ps16 = 0)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    const float wden = (float)max(W - 1, 1), hden = (float)max(H - 1, 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
        const WarpTap t = warp_tap(x, y, flow[(b * 2) * plane + pix], flow[(b * 2 + 1) * plane + pix], H, W, wden, hden);
        T *o = dst + i * ps + coff;
        for (int c = 0; c < ps - coff; ++c) {
            const float v = c < C ? warp_sample(frame2 + (b * C + c) * plane, t) : 0.0f;
            o[c] = (T)v;
            if (dst16 && c < ps16 - coff) dst16[i * ps16 + coff + c] = (half_t)v;
        }
    }
}
int launch_warp_fused(const float *frame2, const float *flow, void *dst, int B, int C, int H, int W, int ps, int coff, int dtype,
                      hipStream_t s, void *dst16, int ps16)
{
    if (dst16 && (dtype != 0 || ps16 - coff > ps - coff)) return (int)hipErrorInvalidValue;   // the second (fp16) destination belongs to the fp32 variant
    if ((W & 3) == 0 && C == 3) {
        const int nwg = ((W + 63) / 64) * ((H + 31) / 32) * B;
        if (dtype == 0) warp_tiled_kernel<3, float><<<nwg, 256, 0, s>>>(frame2, flow, nullptr, B, H, W, dst, ps, coff, dst16, ps16);
        else if (dtype == 2) warp_tiled_kernel<3, half_t><<<nwg, 256, 0, s>>>(frame2, flow, nullptr, B, H, W, dst, ps, coff);
        else warp_tiled_kernel<3, bf16_t><<<nwg, 256, 0, s>>>(frame2, flow, nullptr, B, H, W, dst, ps, coff);
        return (int)hipGetLastError();
    }
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 256 * 64);
    if (dtype == 0) warp_fused_kernel<float><<<grid, 256, 0, s>>>(frame2, flow, (float *)dst, B, C, H, W, ps, coff, (half_t *)dst16, ps16);
    else if (dtype == 2) warp_fused_kernel<half_t><<<grid, 256, 0, s>>>(frame2, flow, (half_t *)dst, B, C, H, W, ps, coff);
    else warp_fused_kernel<bf16_t><<<grid, 256, 0, s>>>(frame2, flow, (bf16_t *)dst, B, C, H, W, ps, coff);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Frame pre/post-processing on the GPU: what the reference does on the
// host around every forward.
//   preprocess:  transforms.ToTensor() + Normalize(mean, std)  (inference.py:38-41, 44-48):
//                uint8 HWC -> float32 /255 -> (x - mean) / std, NCHW.  True fp32 divisions, same order.
//   postprocess: denormalize_frame (inference.py:51-58): NCHW fp32 -> HWC, x * std + mean (the
//                reference applies this to an output that is already in [0,1] - kept behind `denorm`),
//                clip to [0,1], * 255, astype(uint8) (truncation).  numpy promotes to float64 there
//                (np.array([...]) is float64), so the arithmetic here is double: bit-exact bytes.
// ------------------------------------------------------------------------------------------
struct Stats4 { float mean[4], stdv[4]; };
struct Stats4d { double mean[4], stdv[4]; };  // numpy's float64 constants in denormalize_frame

__global__ void preprocess_u8_kernel(const unsigned char *__restrict__ src, float *__restrict__ dst, int B, int H, int W, int C,
                                     Stats4 st)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        const unsigned char *p = src + i * C;
        for (int c = 0; c < C; ++c) {
            const float v = (float)p[c] / 255.0f;
            dst[(b * C + c) * plane + pix] = (v - st.mean[c]) / st.stdv[c];
        }
    }
}
__global__ void postprocess_u8_kernel(const float *__restrict__ src, unsigned char *__restrict__ dst, int B, int H, int W, int C,
                                      Stats4d st, int denorm)
{
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / plane, pix = i - b * plane;
        unsigned char *p = dst + i * C;
        for (int c = 0; c < C; ++c) {
            double v = (double)src[(b * C + c) * plane + pix];
            if (denorm) v = v * st.stdv[c] + st.mean[c];
            v = fmin(fmax(v, 0.0), 1.0) * 255.0;   // NaN -> 0 (fmax), as np.clip then astype would not define
            p[c] = (unsigned char)v;
        }
    }
}
// C = 3, H*W % 4 == 0: one thread = 4 consecutive pixels = 12 contiguous bytes (three dword accesses on the uint8 side,
// three 16-byte accesses on the fp32 planes).  Same per-element arithmetic as the scalar kernels; what changes is the
// access width - the uint8 side may be pinned host memory reached over PCIe, where single-byte accesses crawl.
__global__ void preprocess_u8x4_kernel(const unsigned *__restrict__ src, float *__restrict__ dst, int B, size_t plane, Stats4 st)
{
    const size_t groups = plane / 4, total = (size_t)B * groups;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t b = g / groups, pix = (g - b * groups) * 4;
        const unsigned w0 = src[g * 3], w1 = src[g * 3 + 1], w2 = src[g * 3 + 2];
        const unsigned char by[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                      (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                      (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = (float)by[q * 3 + c] / 255.0f;
                o[q] = (v - st.mean[c]) / st.stdv[c];
            }
            *reinterpret_cast<f32x4 *>(dst + (b * 3 + c) * plane + pix) = o;
        }
    }
}
__global__ void postprocess_u8x4_kernel(const float *__restrict__ src, unsigned *__restrict__ dst, int B, size_t plane, Stats4d st,
                                        int denorm)
{
    const size_t groups = plane / 4, total = (size_t)B * groups;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t b = g / groups, pix = (g - b * groups) * 4;
        unsigned char by[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(src + (b * 3 + c) * plane + pix);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double v = (double)x[q];
                if (denorm) v = v * st.stdv[c] + st.mean[c];
                v = fmin(fmax(v, 0.0), 1.0) * 255.0;
                by[q * 3 + c] = (unsigned char)v;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dst[g * 3 + k] = (unsigned)by[4 * k] | ((unsigned)by[4 * k + 1] << 8) | ((unsigned)by[4 * k + 2] << 16) | ((unsigned)by[4 * k + 3] << 24);
    }
}
// C = 3, the whole batch a multiple of 16 bytes on the uint8 side: one thread = ONE 16-byte quad of the interleaved byte stream
// (5 1/3 pixels), so that a wave's access to the uint8 side is 1 KiB of consecutive addresses per instruction.  That side is pinned
// host memory in the streaming harness (emavfi/stream.py): fine-grained, not cached by the GPU, every instruction its own PCIe
// transactions - the dword kernels above touch every 64-byte line three times (three 12-byte-strided instructions per thread).  The
// fp32 planes (HBM) are read / written as scalars at a lane stride of 5 1/3 floats: cached, cheap.  Byte g of the batch is channel
// g % 3 of global pixel g / 3; same per-element arithmetic as the scalar kernels (round 5).
__global__ __launch_bounds__(256) void preprocess_u8q_kernel(const uint4 *__restrict__ src, float *__restrict__ dst, size_t nquads, size_t plane,
                                                             Stats4 st)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[q];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        const size_t g0 = q * 16;
        size_t px = g0 / 3;
        int c = (int)(g0 - px * 3);
        size_t b = px / plane, pix = px - b * plane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float u = (float)((w[j >> 2] >> (8 * (j & 3))) & 0xffu) / 255.0f;
            dst[(b * 3 + c) * plane + pix] = (u - st.mean[c]) / st.stdv[c];
            if (++c == 3) { c = 0; if (++pix == plane) { pix = 0; ++b; } }
        }
    }
}
__global__ __launch_bounds__(256) void postprocess_u8q_kernel(const float *__restrict__ src, uint4 *__restrict__ dst, size_t nquads, size_t plane,
                                                              Stats4d st, int denorm)
{
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (size_t)gridDim.x * blockDim.x) {
        const size_t g0 = q * 16;
        size_t px = g0 / 3;
        int c = (int)(g0 - px * 3);
        size_t b = px / plane, pix = px - b * plane;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double v = (double)src[(b * 3 + c) * plane + pix];
            if (denorm) v = v * st.stdv[c] + st.mean[c];
            v = fmin(fmax(v, 0.0), 1.0) * 255.0;
            w[j >> 2] |= (unsigned)(unsigned char)v << (8 * (j & 3));
            if (++c == 3) { c = 0; if (++pix == plane) { pix = 0; ++b; } }
        }
        dst[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// measurement knob (round 5): cap the grid of the quad kernels (EMAVFI_PREPOST_WGS=n; 0 / unset: one thread per quad up to 262 140 blocks)
static int prepost_grid(size_t nq)
{
    static const int cap = [] { const char *e = getenv("EMAVFI_PREPOST_WGS"); return e ? atoi(e) : 0; }();
    const size_t want = (nq + 255) / 256;
    return (int)std::min<size_t>(want, cap > 0 ? (size_t)cap : (size_t)65535 * 4);
}
static bool u8q_ok(const void *u8, int B, int H, int W, int C)
{
    return C == 3 && ((size_t)B * H * W * 3) % 16 == 0 && ((uintptr_t)u8 & 15) == 0;
}
static bool u8x4_ok(const void *u8, const void *f32, int H, int W, int C)
{
    return C == 3 && ((size_t)H * W) % 4 == 0 && ((uintptr_t)u8 & 3) == 0 && ((uintptr_t)f32 & 15) == 0;
}
int launch_preprocess_u8(const unsigned char *src, float *dst, int B, int H, int W, int C, const float *mean, const float *stdv,
                         hipStream_t s)
{
    Stats4 st{};
    for (int c = 0; c < C && c < 4; ++c) { st.mean[c] = mean[c]; st.stdv[c] = stdv[c]; }
    if (u8q_ok(src, B, H, W, C)) {
        const size_t nq = (size_t)B * H * W * 3 / 16;
        const int grid = prepost_grid(nq);
        preprocess_u8q_kernel<<<grid, 256, 0, s>>>((const uint4 *)src, dst, nq, (size_t)H * W, st);
        return (int)hipGetLastError();
    }
    if (u8x4_ok(src, dst, H, W, C)) {
        const int grid = (int)std::min<size_t>(((size_t)B * H * W / 4 + 255) / 256, 65535 * 4);
        preprocess_u8x4_kernel<<<grid, 256, 0, s>>>((const unsigned *)src, dst, B, (size_t)H * W, st);
        return (int)hipGetLastError();
    }
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    preprocess_u8_kernel<<<grid, 256, 0, s>>>(src, dst, B, H, W, C, st);
    return (int)hipGetLastError();
}
int launch_postprocess_u8(const float *src, unsigned char *dst, int B, int H, int W, int C, const double *mean, const double *stdv,
                          int denorm, hipStream_t s)
{
    Stats4d st{};
    for (int c = 0; c < C && c < 4; ++c) { st.mean[c] = mean[c]; st.stdv[c] = stdv[c]; }
    if (u8q_ok(dst, B, H, W, C)) {
        const size_t nq = (size_t)B * H * W * 3 / 16;
        const int grid = prepost_grid(nq);
        postprocess_u8q_kernel<<<grid, 256, 0, s>>>(src, (uint4 *)dst, nq, (size_t)H * W, st, denorm);
        return (int)hipGetLastError();
    }
    if (u8x4_ok(dst, src, H, W, C)) {
        const int grid4 = (int)std::min<size_t>(((size_t)B * H * W / 4 + 255) / 256, 65535 * 4);
        postprocess_u8x4_kernel<<<grid4, 256, 0, s>>>(src, (unsigned *)dst, B, (size_t)H * W, st, denorm);
        return (int)hipGetLastError();
    }
    const int grid = (int)std::min<size_t>(((size_t)B * H * W + 255) / 256, 65535 * 4);
    postprocess_u8_kernel<<<grid, 256, 0, s>>>(src, dst, B, H, W, C, st, denorm);
    return (int)hipGetLastError();
}

// ---- census of the one-launch packs (deform_pack3.inl, DeformParams::census): one wave per block adds the 64 slots
struct CensusTotals { unsigned long long t[8]; };
__global__ void census_reduce_kernel(const unsigned *__restrict__ census, unsigned long long *__restrict__ out, CensusTotals totals)
{
    const int blk = blockIdx.x, lane = threadIdx.x;   // 64 threads
    const unsigned *c = census + ((size_t)blk * 64 + lane) * 4;
    unsigned long long fix = c[0], par = c[1];
    unsigned mx = c[2];
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        fix += __shfl_xor(fix, sh);
        par += __shfl_xor(par, sh);
        const unsigned o = __shfl_xor(mx, sh);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) {
        out[blk * 4 + 0] = fix; out[blk * 4 + 1] = totals.t[blk]; out[blk * 4 + 2] = par; out[blk * 4 + 3] = mx;
    }
}
int launch_census_reduce(const unsigned *census, unsigned long long *out, int nblocks, const unsigned long long *totals, hipStream_t s)
{
    if (nblocks < 1 || nblocks > 8) return (int)hipErrorInvalidValue;
    CensusTotals t{};
    for (int i = 0; i < nblocks; ++i) t.t[i] = totals[i];
    census_reduce_kernel<<<nblocks, 64, 0, s>>>(census, out, t);
    return (int)hipGetLastError();
}
