// Host-callable launchers of misc_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stddef.h>

struct PackDesc {
    int cout, cin_raw, cin_off, cin_take;  // raw OIHW tensor: take input channels [cin_off, cin_off + cin_take)
    int ck, nchunk, nf, npass;             // packed geometry
    int perm;                              // 0 = identity, 1 = offset_conv routing (offsets | mask)
    int via_bf16;                          // 1 = round every weight to bf16 first, then store it in the packed type
                                           // (f16 fragments holding the bf16 model's weights exactly: deform_pack.inl)
    int bias_f16;                          // 1 = the fp32 bias table holds fp16-rounded values (autocast casts the bias too)
    int mfma16;                            // 1 = fragments for v_mfma_f32_16x16x32: [tap][k32][cout16 block][lane (i, kb)][8 elements]
    int ring;                              // Layer::ring; 3: the main fragments take input channels 0..63 (ck = 64) and conv_ring.inl's tail
                                           // [j 3][nf 2][lane (r, h)][8] follows: W[32 nf + r][64 + (e & 3)][tap slot 4 j + 2 h + (e >> 2)]
    int first6;                            // conv_first.inl layout (6 -> 64, 16-bit): [kg 5][nf 2][lane (r, h)][8]: W[32 nf + r][channel e][tap 2 kg + h]
    int pack3;                             // deform_pack3.inl layouts (f16 elements, cin_take = 67): 1 = DCN 67 -> <= 67, 2 = offset_conv 67 -> 27;
                                           // 3 = deform_f32w.inl (fp32 elements, the fp32 DCN on an LDS window)
};

int launch_pack_conv(const float *w, const float *bias, void *wp, float *bp, const PackDesc &d, int dtype, hipStream_t s);
int launch_pack_ctx(const float *lin_w, const float *lin_b, const float *w9, const float *b9, float *dst, int m, int round16,
                    hipStream_t s);
// channels-last conversion of channels [c0, c0 + nc) of every pixel (nc % 4 == 0): fp16 -> fp32 (widen) or fp32 -> fp16
int launch_convert_cl(const void *src, void *dst, size_t npx, int ps_src, int ps_dst, int c0, int nc, int widen, hipStream_t s);
int launch_pack_input(const float *f1, const float *f2, void *dst, int B, int C, int H, int W, int cpad, int dtype, hipStream_t s);
int launch_nchw_to_cl(const float *src, void *dst, int B, int C, int H, int W, int ps, int dtype, hipStream_t s);
int launch_cl_to_nchw(const void *src, float *dst, int B, int C, int H, int W, int ps, int coff, int dtype, hipStream_t s);
int launch_om_from_nchw(const float *off, const float *msk, float *om, int B, int H, int W, hipStream_t s);
int launch_pool_partial(const void *src, float *part, int B, int npix, int cp, int ps, int nparts, int dtype, hipStream_t s);
int launch_ctx_finish(const float *part, const float *ctxw, float *ctx_out, float *table, int B, int m, int cp, int nparts,
                      int npix, int coutpad, int round16, hipStream_t s);
int launch_warp_nchw(const float *frame2, const float *flow, float *out, int B, int C, int H, int W, hipStream_t s);
int launch_warp_fused(const float *frame2, const float *flow, void *dst, int B, int C, int H, int W, int ps, int coff, int dtype,
                      hipStream_t s);
int launch_preprocess_u8(const unsigned char *src, float *dst, int B, int H, int W, int C, const float *mean, const float *stdv,
                         hipStream_t s);
int launch_postprocess_u8(const float *src, unsigned char *dst, int B, int H, int W, int C, const double *mean, const double *stdv,
                          int denorm, hipStream_t s);
