/*
 * emavfi.h - C-ABI of libemavfi.so, the MI355X-native (gfx950) EMA-VFI
 * inference path.
 *
 * The reference (424635328/video-frame-interpolation) has no FFI layer: its hot
 * path is the Python class EMA_VFI in src/models/ema_vfi.py, whose work is done
 * by torch / torchvision operators.  Each entry below replaces the operator
 * call sites cited next to it.  The Python mirror of the reference class
 * (video-frame-interpolation_amd/emavfi/model.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer on the current HIP device, 16-byte
 *    aligned, to a dense tensor.  Image tensors crossing this boundary are
 *    NCHW fp32, exactly what the reference's forward() receives and returns.
 *  - The library never allocates, frees or retains device memory and never
 *    synchronises the device: the caller owns inputs, outputs, the packed
 *    weight blob and the workspace (sized by the *_bytes queries) and all work
 *    is enqueued on the hipStream_t passed as `stream` (void* here so the
 *    header needs no HIP include; NULL = the default stream).
 *  - Every int-returning entry returns 0 on success and a negative EMAVFI_E_*
 *    code on failure; emavfi_last_error() then holds a thread-local message.
 *    Nothing aborts the process (the reference wraps its loop in try/except,
 *    inference.py:207-208).
 *  - `dtype` selects the arithmetic of the contractions: EMAVFI_F32 computes
 *    every convolution with fp32-input MFMA (exact fp32 FMA chains; this is the
 *    parity mode, <= 1e-3 max-abs vs the reference's CPU forward);
 *    EMAVFI_BF16 stores activations and weights in bf16 and accumulates in
 *    fp32 (BASELINE.json configs[2]: "bf16 convs + fp32 warp").  One stage
 *    leaves bf16: the one-launch ModulatedDeformConvPack kernel (mid_channels
 *    64) works on the IEEE f16 image of its input window - bf16 values convert
 *    exactly inside f16's normal range, keep 11 instead of 8 significant bits
 *    through the bilinear blend, lose trailing bits below 6.1e-5 and SATURATE
 *    at +-65504 (v_cvt_pkrtz; no infinities are produced) - contracts bf16-
 *    rounded weights stored as f16 (weights below 6.1e-5 in magnitude are f16
 *    subnormals there: absolute error <= 3e-8 each), and hands the tensor
 *    between two consecutive packs on as f16 bit patterns.  Since round 3 the feature map
 *    that enters the fusion stage (`feat`, and the warped frame beside it) is
 *    itself stored as f16 in this mode, saturating the same way, and its two
 *    other readers (context_encoding.0, motion_estimation.0) contract it with
 *    bf16-rounded weights stored as f16.  Activations beyond +-65504 in `feat`
 *    and in the fusion stage are therefore clamped there, not propagated
 *    (tests/test_gpu_parity.py::test_bf16_pack_saturates_at_the_f16_range).
 *    EMAVFI_F16 is the same data flow in IEEE half precision - the arithmetic
 *    torch.cuda.amp.autocast() gives the reference's convolutions on a GPU
 *    (inference.py:159); its fused deformable kernel blends the four corners
 *    in packed f16; values beyond +-65504 overflow to inf exactly as they
 *    would there.  NON-FINITE ACTIVATIONS inside the one-launch pack (reachable only through such an
 *    f16 overflow): the reference confines an Inf / NaN sample to the output pixels that sample it; this
 *    kernel may also turn channels 64..66 of the pixel 16 columns to the left / right in the same 2 x 16
 *    fragment row into NaN (its third output fragment contracts two pixels per MFMA column and separates
 *    them by zero weights: 0 x Inf), and a lane whose sample is parked for the fix-up pass reads window
 *    offset 0 / the arena's zero slot against zero weights.  Finite inputs are unaffected; a frame that
 *    contains any non-finite value is garbage in the reference as well.  A NaN / infinite flow (or one whose `2 * v` overflows)
 *    warps to NaN in every channel, a finite flow far outside to 0 - what
 *    the reference's CPU grid_sample returns (ema_vfi.py:169).  Flow, warp coordinates,
 *    deformable offsets / masks / sampling positions, the pooled context
 *    vector and the output are fp32 in every mode.
 *  - The model is identified by the reference constructor's three integers
 *    (ema_vfi.py:64): in_channels, mid_channels, num_blocks.
 */
#ifndef EMAVFI_H
#define EMAVFI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMAVFI_VERSION 403 /* 0.4.3: emavfi_forward_census, emavfi_mdcn_census (round 6; the workspace grows by 8 KiB, the packed layout is unchanged); 0.4.2: emavfi_forward_staged, emavfi_mdcn_profiled (round 5; the packed layout is 0.4.1's, but a blob says which library packed it: re-pack); 0.4.1: context_encoding.1 / .2 re-packed for conv_wreg.inl; 0.4.0: the packed blob starts with a 256-byte self-describing header, emavfi_forward takes packed_bytes (round 4): re-pack */

#define EMAVFI_F32 0
#define EMAVFI_BF16 1
#define EMAVFI_F16 2
/* The reference's forward under torch.cuda.amp.autocast() (inference.py:159), op policy restated: fp16 nn.Conv2d /
 * nn.Linear (input, weight AND bias cast to fp16, fp32 accumulation, fp16 result), fp16 tensors between those layers,
 * but grid_sample (ema_vfi.py:169; promotes to its widest argument, and frame2 and the grid are fp32) in fp32 on the
 * fp16-valued flow and torchvision's deform_conv2d (ema_vfi.py:60)
 * in fp32 on the UNROUNDED fp32 fusion tensor with the fp32 master weights (its Autocast kernel casts every argument
 * to float and the result back to the input's dtype, which is fp32 because cat(feat, warped) promotes), offsets /
 * sigmoid(mask) as fp16 values, tanh and (t+1)/2 in fp16.  The frame holds fp16-representable values.  Since 0.4.3 the fp32 deform_conv2d
 * of THIS mode contracts as a three-term f16 split (22 bits of each operand, exact products, fp32 accumulation: 3e-7 .. 1.4e-6 relative,
 * the spread between two fp32 summation orders - which is what separates any GPU run of the reference from its CPU run anyway); its
 * sampling positions, corner weights and blend are fp32 as before.  EMAVFI_F32 is the mode with exact fp32 FMA chains.
 * EMAVFI_F16 stays the FAST half-precision mode (DCN contraction and blend in fp16 too). */
#define EMAVFI_AMP16 3
/* fp32-ACCURATE contractions on the 16-bit matrix pipe (round 6; gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the f16 rate):
 * every nn.Conv2d / nn.Linear computes x_hi w_hi + x_hi w_lo + x_lo w_hi with x = x_hi + x_lo, w = w_hi + w_lo in IEEE f16 (22
 * significant bits of each operand; the products are exact in fp32, accumulation in fp32, biases and activations in fp32), activations
 * between the layers are stored as the two f16 halves of the fp32 value; flow, warp, sampling geometry, the pooled context and the three
 * deform_conv2d are the EXACT fp32 ones of EMAVFI_F32.  Passes the fp32 mode's parity gates (<= 1e-3 on the frame, <= 5e-4 relative on
 * every stage) on every reference-run fixture; a SEPARATELY NAMED mode: EMAVFI_F32 stays the exact-fp32 parity mode.  Needs |activation|
 * < 65504; the low half underflows below 6e-5 (an absolute 6e-8 per product).  Stage entries: emavfi_conv3x3 only. */
#define EMAVFI_F32X3 4

#define EMAVFI_OK 0
#define EMAVFI_E_ARG (-1)         /* bad shape / dtype / null pointer / misaligned pointer */
#define EMAVFI_E_UNSUPPORTED (-2) /* model widths this build has no kernel instantiation for */
#define EMAVFI_E_WORKSPACE (-3)   /* workspace or packed buffer too small */
#define EMAVFI_E_LAUNCH (-4)      /* hipGetLastError() after a launch */

/* Activation selector of emavfi_conv3x3 (conv vs conv_block, ema_vfi.py:7-14). */
#define EMAVFI_ACT_NONE 0
#define EMAVFI_ACT_RELU 1
#define EMAVFI_ACT_TANH01 2 /* tanh then (t+1)/2: reconstruction tail, ema_vfi.py:106,146 */

int emavfi_version(void);
const char *emavfi_last_error(void);

/* 0 if (in_channels, mid_channels, num_blocks, dtype) has kernels in this build. */
int emavfi_supported(int in_channels, int mid_channels, int num_blocks, int dtype);

/* Number of tensors in the reference state_dict (ema_vfi.py:63-107): 16 + 8*num_blocks... see
 * emavfi_param_count(); order = the reference's registration order:
 *   feat_ext_conv1.0.{weight,bias}; feat_ext_blocks.conv_block_i.0.{w,b} (i < num_blocks);
 *   context_encoding.{0.0,1.0,2.0}.{w,b}; context_encoding.5.{w,b};
 *   motion_estimation.{0.0,1.0,2}.{w,b};
 *   attention_blocks.i.offset_conv.{w,b}, attention_blocks.i.dcn_v2.{w,b} (i < num_blocks);
 *   reconstruction.{0.0,1.0,2}.{w,b}. */
int emavfi_param_count(int num_blocks);

/* Re-pack the fp32 OIHW parameters into the MFMA fragment order the kernels stream
 * (replaces nothing in the reference: torch keeps OIHW; this runs once after
 * load_state_dict, inference.py:69).  `params[i]` are device pointers in the order above. */
size_t emavfi_packed_bytes(int in_channels, int mid_channels, int num_blocks, int dtype);
int emavfi_pack_weights(int in_channels, int mid_channels, int num_blocks,
                        const void *const *params, int n_params,
                        void *packed, size_t packed_bytes, int dtype, void *stream);

/* The packed blob is self-describing.  Bytes [0, 256): header, little endian -
 *   char magic[8] = "EMAVFIPK"; u32 version (EMAVFI_VERSION); u32 header_bytes (256); u32 in_channels, mid_channels, num_blocks;
 *   u32 dtype (as requested: EMAVFI_AMP16 is a layout of its own); u32 layout_tag (emavfi_layout_tag() of the packing process);
 *   u32 reserved; u64 total_bytes (= emavfi_packed_bytes); u64 checksum; zeros up to byte 256;
 * bytes [256, total_bytes): the payload.  checksum = sum over the payload's 32-bit words w[i] of (w[i] + 0x9E3779B9) * (2 i + 1)
 * mod 2^64.  The reference has no counterpart (torch.load of a state_dict, inference.py:69); the blob is what this library
 * caches on disk and broadcasts between ranks, so it has to say what it is.
 *
 * emavfi_layout_tag: bit mask of the process-wide environment switches the packed LAYOUT depends on (read once per process):
 *   1 EMAVFI_CONV_MFMA16=0, 2 EMAVFI_CONV_RING=0, 4 EMAVFI_CONV_S2RING=0, (8: unused since 0.4.1), 16 EMAVFI_PACK_F16_CHAIN=0,
 *   32 EMAVFI_NO_FUSED_OFFSET, 64 EMAVFI_CONV_WREG=0.  Cache keys must use it (not the environment, which may have changed since the library latched it).
 * emavfi_packed_check: verifies header (magic, version, model, dtype, layout tag, size) and checksum of a blob in device OR host
 *   memory; EMAVFI_E_ARG with a message naming the mismatch.  THE ONE ENTRY THAT SYNCHRONISES: for a device blob it makes the
 *   blob's device current, waits for EVERY stream of that device (hipDeviceSynchronize - the producer may have been the pack
 *   kernels, an RCCL broadcast or a cache upload on any stream, also a non-blocking one), copies the blob to the host and restores
 *   the caller's device.  Host memory is read in place: the caller orders its own writes.  Call it when a blob arrives - from a
 *   file, another rank, another process - not per frame.
 * emavfi_forward itself (a) returns EMAVFI_E_ARG when packed_bytes is smaller than the model / dtype needs and (b) compares the
 *   header ON THE DEVICE with what the call expects, in its LAST launch (blob_guard, ~2 us): for a blob of another version /
 *   model / dtype / layout tag every kernel still runs on the foreign bytes, and that last launch then overwrites `out` with NaN -
 *   an all-NaN frame, never plausible garbage; it cannot return a code for device-resident bytes without synchronising.
 *   The `taps` (a test hook) are NOT guarded: they hold whatever the kernels computed from the foreign bytes. */
int emavfi_layout_tag(void);
int emavfi_packed_check(int in_channels, int mid_channels, int num_blocks, int dtype, const void *packed, size_t packed_bytes);

/* EMA_VFI.forward(frame1, frame2) -> out, ema_vfi.py:110-147.
 * frame1, frame2: [B, in_channels, H, W] fp32; out: [B, in_channels, H, W] fp32 in [0,1].
 * `taps` is NULL, or 5 + num_blocks device pointers (any may be NULL) that receive NCHW fp32
 * copies of the intermediates the golden vectors hold:
 *   [0] feat [B,mid,H,W]  [1] ctx [B,mid]  [2] flow [B,2,H,W]  [3] warped [B,in_channels,H,W]
 *   [4] reserved  [5+i] output of attention block i [B,mid+3,H,W]. */
size_t emavfi_workspace_bytes(int in_channels, int mid_channels, int num_blocks,
                              int B, int H, int W, int dtype);
int emavfi_forward(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes,
                   const float *frame1, const float *frame2, float *out,
                   void *workspace, size_t workspace_bytes,
                   int B, int H, int W, int dtype, float *const *taps, void *stream);

/* Measurement hooks (no reference counterpart: the reference has no profiling).
 * emavfi_forward_launches enumerates the kernel launches one forward enqueues, in order: returns
 * their number (also when capacity == 0), and for capacity >= that number fills `names`
 * (newline-separated, "kernel<instantiation> reference-layer"), and each launch's ALGORITHMIC
 * flops and bytes (real, unpadded channels; every tensor touched once).
 * emavfi_forward_profiled is emavfi_forward with launch i bracketed by hipEventRecord on
 * events[2i] / events[2i+1] (caller-created hipEvent_t, timing enabled) on `stream`. */
int emavfi_forward_launches(int in_channels, int mid_channels, int num_blocks, int B, int H, int W, int dtype,
                            char *names, size_t names_bytes, double *flops, double *bytes, int capacity);
/* emavfi_forward with STAGE events, for a caller that pipelines pieces of a batch over several streams (frame pairs are
 * independent, ema_vfi.py:110-147 has no cross-sample op; emavfi/model.py, EMAVFI_PIPELINE): `stage_events` is NULL or three
 * caller-created hipEvent_t (any may be NULL), recorded on `stream` behind
 *   [0] the front of the forward - feature extraction, context encoding, motion estimation, warp (ema_vfi.py:112-130),
 *   [1] the last attention block (:136-138),   [2] the reconstruction (:144-146; the forward's last launch).
 * `events` / `n_events` as in emavfi_forward_profiled, or NULL / 0.  No `taps`. */
int emavfi_forward_staged(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes,
                          const float *frame1, const float *frame2, float *out,
                          void *workspace, size_t workspace_bytes,
                          int B, int H, int W, int dtype, void *const *stage_events, void *const *events, int n_events, void *stream);
int emavfi_forward_profiled(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes,
                            const float *frame1, const float *frame2, float *out,
                            void *workspace, size_t workspace_bytes,
                            int B, int H, int W, int dtype, void *const *events, int n_events, void *stream);

/* EMA_VFI.warp(frame2, feature, flow), ema_vfi.py:149-171 (grid build + normalise +
 * F.grid_sample bilinear/zeros/align_corners=True), fused into one HBM-bound kernel.
 * frame2 [B,C,H,W], flow [B,2,H,W] (channel 0 = dx, 1 = dy, pixels), out [B,C,H,W]; fp32. */
int emavfi_warp(const float *frame2, const float *flow, float *out,
                int B, int C, int H, int W, void *stream);

/* Frame pre/post-processing around the forward (the reference does both on the host, per frame).
 * emavfi_preprocess_u8: transforms.ToTensor() + Normalize(mean, std), inference.py:38-41 / :44-48
 *   (cv2.resize excluded): frames_hwc uint8 [B,H,W,C] -> out_nchw fp32 [B,C,H,W] = ((u8/255) - mean[c]) / std[c].
 * emavfi_postprocess_u8: denormalize_frame, inference.py:51-58: frames_nchw fp32 [B,C,H,W] -> out_hwc uint8
 *   [B,H,W,C] = uint8(clip(x * std[c] + mean[c], 0, 1) * 255) (truncation; float64 arithmetic as numpy's
 *   promotion makes it there).  denormalize = 0 skips the x*std+mean step, which the reference applies to an
 *   output that is already in [0,1].
 * `mean` and `std` are HOST pointers to C values (C <= 4): fp32 for preprocess (torchvision builds fp32
 * tensors), float64 for postprocess (numpy's np.array([...]) constants); the frame pointers are device pointers -
 * the uint8 side may also be pinned (device-mapped) host memory, which the kernel then reads / writes over PCIe. */
int emavfi_preprocess_u8(const unsigned char *frames_hwc, float *out_nchw, int B, int H, int W, int C,
                         const float *mean, const float *std, void *stream);
int emavfi_postprocess_u8(const float *frames_nchw, unsigned char *out_hwc, int B, int H, int W, int C,
                          const double *mean, const double *std, int denormalize, void *stream);

/* One conv / conv_block (ema_vfi.py:7-14): Conv2d(k=3, p=1, stride 1 or 2) + activation.
 * x [B,Cin,H,W], weight [Cout,Cin,3,3], bias [Cout], y [B,Cout,ceil(H/stride),ceil(W/stride)]. */
size_t emavfi_conv3x3_workspace_bytes(int B, int Cin, int Cout, int H, int W, int stride, int dtype);
int emavfi_conv3x3(const float *x, const float *weight, const float *bias, float *y,
                   int B, int Cin, int Cout, int H, int W, int stride, int act, int dtype,
                   void *workspace, size_t workspace_bytes, void *stream);

/* torchvision.ops.deform_conv2d(x, offset, weight, bias, padding=1, mask=mask) as configured
 * at ema_vfi.py:45-51 / :60 (3x3, stride 1, pad 1, one offset group, one weight group).
 * x [B,C,H,W], offset [B,18,H,W] (2k = dy, 2k+1 = dx of tap k = 3i+j), mask [B,9,H,W],
 * weight [O,C,3,3], bias [O], y [B,O,H,W]. */
size_t emavfi_deform_conv2d_workspace_bytes(int B, int C, int O, int H, int W, int dtype);
int emavfi_deform_conv2d(const float *x, const float *offset, const float *mask,
                         const float *weight, const float *bias, float *y,
                         int B, int C, int O, int H, int W, int dtype,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ModulatedDeformConvPack.forward(x), ema_vfi.py:53-60, as one stage:
 *   raw = offset_conv(x) (3x3, pad 1, C -> 27, :35-43); offset = cat(raw[:, 0:9], raw[:, 18:27]); mask = sigmoid(raw[:, 9:18]) (:56-59);
 *   y = dcn_v2(x, offset, mask) (:60).
 * x, y [B,C,H,W] fp32 NCHW; offset_weight [27,C,3,3], offset_bias [27], dcn_weight [C,C,3,3], dcn_bias [C] (may be NULL).
 * C must be mid_channels + 3 of a supported model (the only width the reference builds the pack at, ema_vfi.py:97).
 * Routed exactly as one attention block of emavfi_forward: EMAVFI_BF16 / EMAVFI_F16 at C = 67 run the ONE-LAUNCH kernel
 * (offset_conv on the staged window, offsets / masks in registers, DCN, fix-up loop for samples that leave the window); EMAVFI_F32
 * runs conv3x3 + the fp32 LDS-window DCN; EMAVFI_AMP16 the fp16 offset_conv + fp32 DCN pair; other widths conv3x3 + the
 * global-gather DCN.  `flags` reproduce the forms the forward hands the tensor over in (one-launch kernel only):
 *   EMAVFI_MDCN_IN_F16 / _OUT_F16  bf16 model: x is stored / y is produced as IEEE f16 bit patterns (hand-off between packs, `feat`);
 *   EMAVFI_MDCN_SPLIT_TAIL         channels mid.. of x reach the kernel through the compact tail buffer - 4 channels per pixel - (the first pack's input). */
#define EMAVFI_MDCN_IN_F16 1
#define EMAVFI_MDCN_OUT_F16 2
#define EMAVFI_MDCN_SPLIT_TAIL 4
size_t emavfi_mdcn_workspace_bytes(int B, int C, int H, int W, int dtype, int flags);
int emavfi_mdcn(const float *x, const float *offset_weight, const float *offset_bias, const float *dcn_weight, const float *dcn_bias,
                float *y, int B, int C, int H, int W, int dtype, int flags, void *workspace, size_t workspace_bytes, void *stream);
/* Measurement hook: emavfi_mdcn with the stage's own launches (one in the 16-bit modes at C = 67, else two) bracketed by
 * hipEventRecord on events[2i] / events[2i+1]; the layout conversions around them are not bracketed (bench.py's
 * also_pack_vs_offset_spread: what the dominant kernel costs when the offsets leave its staged window). */
int emavfi_mdcn_profiled(const float *x, const float *offset_weight, const float *offset_bias, const float *dcn_weight, const float *dcn_bias,
                         float *y, int B, int C, int H, int W, int dtype, int flags, void *workspace, size_t workspace_bytes,
                         void *const *events, int n_events, void *stream);

/* Census of the one-launch ModulatedDeformConvPack kernel (measurement hook; the reference bounds its offsets nowhere, ema_vfi.py:55-60,
 * and the kernel stages a window that holds offsets up to +-2 px beyond the tap: samples that leave it take a fix-up pass).  The kernel
 * counts, while it runs: the (wave, tap) groups - 4 rows x 16 pixels x one tap - that took the fix-up, the samples outside the window and
 * the largest |offset| of the waves that had one (only those pay for the census: 0 = every sample was inside the window).  emavfi_forward_census reads the counters the LAST emavfi_forward* call on `workspace` left there
 * (same model, B, H, W, dtype; enqueue it on the same stream), emavfi_mdcn_census those of the last emavfi_mdcn* call:
 *   out[block][4] (unsigned 64-bit, DEVICE memory, num_blocks rows - one row for mdcn) =
 *     {fix-up wave-taps, all wave-taps (0: this block did not run the one-launch kernel, nothing was counted), samples outside the
 *      window (of B*H*W*9), max |offset| in px as fp32 bits}.
 * One tiny launch; nothing is synchronised. */
int emavfi_forward_census(int in_channels, int mid_channels, int num_blocks, int B, int H, int W, int dtype,
                          const void *workspace, size_t workspace_bytes, unsigned long long *out, void *stream);
int emavfi_mdcn_census(int B, int C, int H, int W, int dtype, int flags, const void *workspace, size_t workspace_bytes,
                       unsigned long long *out, void *stream);

/* context_encoding(feat) -> ctx, ema_vfi.py:79-86 (called at :120): conv stride 2 + ReLU, conv stride 2 + ReLU, conv + ReLU,
 * AdaptiveAvgPool2d(1), Flatten, Linear.  feat [B,mid,H,W]; params = 8 device pointers in the Sequential's registration order:
 * context_encoding.{0.0,1.0,2.0}.{weight,bias} ([2m,m,3,3] [2m] [4m,2m,3,3] [4m] [4m,4m,3,3] [4m]) and context_encoding.5.{weight [m,4m],
 * bias [m]}; ctx [B,mid] fp32.
 * reconstruction(fused) -> out, ema_vfi.py:102-107 (called at :144-146): conv + ReLU, conv + ReLU, conv, tanh, then (t + 1) / 2.
 * fused [B,mid+3,H,W]; params = 6 device pointers: reconstruction.{0.0,1.0,2}.{weight,bias}; out [B,3,H,W] in [0,1].
 * Both run the launches emavfi_forward runs for the stage (16-bit modes at mid_channels 64: the LDS-ring kernels, reconstruction.1 + .2
 * as ONE launch), on EMA_VFI(3, mid_channels, 3)'s plan; storage rounding of the inputs as the forward's tensors have it. */
size_t emavfi_context_workspace_bytes(int B, int mid_channels, int H, int W, int dtype);
int emavfi_context(const float *feat, const float *const *params, float *ctx, int B, int mid_channels, int H, int W, int dtype,
                   void *workspace, size_t workspace_bytes, void *stream);
size_t emavfi_reconstruct_workspace_bytes(int B, int mid_channels, int H, int W, int dtype);
int emavfi_reconstruct(const float *fused, const float *const *params, float *out, int B, int mid_channels, int H, int W, int dtype,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Test hook: the A/B switches of the launch sequence (EMAVFI_CONV_FIRST / _FIRSTRING / _HEAD / _TAILFUSE / _LIGHT / _RING2 /
 * _POOLFUSE = 0, EMAVFI_RING_CHUNK = 0, EMAVFI_NO_PERSISTENT_CONV) are read from the environment ONCE per process into one word; this replaces it by
 * (word & and_mask) | or_mask and returns the previous value (bits: 1 no conv_first, 2 no fused first two layers, 4 no fused flow
 * head, 8 no fused reconstruction tail, 16 no planar-head kernel, 32 no persistent conv, 64 no two-layer ring fusions, 128
 * context_encoding.2 stores its output instead of fusing the average pool, 256 (EMAVFI_RING_ONE_WG=1, a measurement switch) the
 * persistent LDS-ring kernels launch one workgroup per CU instead of two, 512 (EMAVFI_RING_CHUNK=0) they walk 45-row segments dealt
 * round-robin instead of one contiguous range of rows per workgroup).  None of them changes the packed layout.  Not for
 * production callers. */
int emavfi_debug_switches(int and_mask, int or_mask);

#ifdef __cplusplus
}
#endif
#endif /* EMAVFI_H */
