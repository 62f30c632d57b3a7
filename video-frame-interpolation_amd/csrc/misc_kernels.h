// Host-callable launchers of misc_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stddef.h>
#include <stdint.h>

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
    int x3;                                // EMAVFI_F32X3: nchunk counts VIRTUAL chunks 3 c + t; t = 0, 2 carry f16(w), t = 1 carries f16(w - f16(w))
    int pack3;                             // deform_pack3.inl layouts (f16 elements, cin_take = 67): 1 = DCN 67 -> <= 67, 2 = offset_conv 67 -> 27;
                                           // 3 = deform_f32w.inl (fp32 elements, the fp32 DCN on an LDS window)
};

// The packed weight blob is self-describing: its first 256 bytes hold this header, the payload
// (what build_plan lays out) follows.  include/emavfi.h documents the same layout for callers that move blobs around.
struct BlobHeader {
    char magic[8];                                                // "EMAVFIPK"
    uint32_t version, header_bytes;                               // EMAVFI_VERSION of the packing library; 256
    uint32_t in_ch, mid, nb, dtype;                               // the model and the REQUESTED dtype (EMAVFI_AMP16 is a layout of its own)
    uint32_t layout_tag, reserved;                                // emavfi_layout_tag() of the packing process
    uint64_t total_bytes;                                         // header + payload
    uint64_t checksum;                                            // blob_checksum() of the payload
    uint64_t reserved2;
};
static_assert(sizeof(BlobHeader) == 64, "blob header layout");
constexpr unsigned kBlobHeaderBytes = 256;
// order-independent position-weighted sum over the payload's 32-bit words: sum (w[i] + 0x9E3779B9) * (2 i + 1) mod 2^64
inline uint64_t blob_checksum_host(const void *payload, size_t bytes)
{
    const uint32_t *w = (const uint32_t *)payload;
    uint64_t s = 0;
    for (size_t i = 0; i < bytes / 4; ++i) s += ((uint64_t)w[i] + 0x9E3779B9ull) * (2 * (uint64_t)i + 1);
    return s;
}
// forward-time guard: the forward's last launch compares the blob's header with what the call expects and overwrites the frame
// with NaN on a mismatch (misc_kernels.hip, blob_guard_kernel)
struct BlobGuard { const BlobHeader *hdr; BlobHeader expect; };
int launch_blob_guard(const BlobGuard &guard, float *out, size_t n, hipStream_t s);
// writes the header (pack time) and adds the payload checksum into it
int launch_blob_seal(void *blob, const BlobHeader &h, hipStream_t s);

int launch_pack_conv(const float *w, const float *bias, void *wp, float *bp, const PackDesc &d, int dtype, hipStream_t s);
int launch_pack_ctx(const float *lin_w, const float *lin_b, const float *w9, const float *b9, float *dst, int m, int round16,
                    hipStream_t s);
// channels-last conversion of channels [c0, c0 + nc) of every pixel (nc % 4 == 0): fp16 -> fp32 (widen) or fp32 -> fp16
// lo_off > 0 (EMAVFI_F32X3): the 16-bit side holds [hi | lo] halves, lo at element offset lo_off: widen = hi + lo, narrow writes both
int launch_convert_cl(const void *src, void *dst, size_t npx, int ps_src, int ps_dst, int c0, int nc, int widen, hipStream_t s, int lo_off = 0);
int launch_pack_input(const float *f1, const float *f2, void *dst, int B, int C, int H, int W, int cpad, int dtype, hipStream_t s);
int launch_nchw_to_cl(const float *src, void *dst, int B, int C, int H, int W, int ps, int dtype, hipStream_t s);
// channels [c0, c0 + ctake) of an NCHW tensor with C channels -> channels 0.. of ps-channel pixels (the rest zero)
int launch_nchw_to_cl_sub(const float *src, void *dst, int B, int C, int c0, int ctake, int H, int W, int ps, int dtype, hipStream_t s);
int launch_cl_to_nchw(const void *src, float *dst, int B, int C, int H, int W, int ps, int coff, int dtype, hipStream_t s);
int launch_om_from_nchw(const float *off, const float *msk, float *om, int B, int H, int W, hipStream_t s);
int launch_pool_partial(const void *src, float *part, int B, int npix, int cp, int ps, int nparts, int dtype, hipStream_t s);
// lo_off > 0 (EMAVFI_F32X3): a row of `part` holds the sums of the hi halves and, lo_off floats further, of the lo halves: added
int launch_ctx_finish(const float *part, const float *ctxw, float *ctx_out, float *table, int B, int m, int cp, int nparts,
                      int npix, int coutpad, int round16, hipStream_t s, int lo_off = 0);
// EMAVFI_F32X3 layouts: NCHW fp32 <-> channels-last f16 halves [hi: ps_half | lo: ps_half]; pack_input the same for cat(frame1, frame2)
int launch_nchw_to_cl_x3(const float *src, void *dst, int B, int C, int H, int W, int ps_half, hipStream_t s);
int launch_cl_to_nchw_x3(const void *src, float *dst, int B, int C, int H, int W, int ps_half, int coff, hipStream_t s);
int launch_pack_input_x3(const float *f1, const float *f2, void *dst, int B, int C, int H, int W, int cpad, hipStream_t s);
int launch_warp_nchw(const float *frame2, const float *flow, float *out, int B, int C, int H, int W, hipStream_t s);
// dst16 / ps16 (fp32 variant only, EMAVFI_AMP16): also write the fp16 rounding of the same values into channels [coff, ps16) of a second
// channels-last tensor
int launch_warp_fused(const float *frame2, const float *flow, void *dst, int B, int C, int H, int W, int ps, int coff, int dtype,
                      hipStream_t s, void *dst16 = nullptr, int ps16 = 0);
int launch_preprocess_u8(const unsigned char *src, float *dst, int B, int H, int W, int C, const float *mean, const float *stdv,
                         hipStream_t s);
int launch_postprocess_u8(const float *src, unsigned char *dst, int B, int H, int W, int C, const double *mean, const double *stdv,
                          int denorm, hipStream_t s);
// deform_pack3.inl's census: sums the 64 atomic slots of each of `nblocks` launches ([block][64][4] u32) into out[block][4] u64 =
// {fix-up wave-taps, totals[block], samples outside the window, max |offset| as fp32 bits}
int launch_census_reduce(const unsigned *census, unsigned long long *out, int nblocks, const unsigned long long *totals, hipStream_t s);
