// libemavfi.so host side: the C-ABI of include/emavfi.h, the layer plan (which kernel
// instantiation serves which reference layer), weight packing and workspace carving.
// No device allocation, no synchronisation, no retained pointers.
#include "../../include/emavfi.h"
#include "common.h"
#include "misc_kernels.h"
#include "conv_first.inl"

#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// common.h, SW_*: the A/B switches of the launch sequence, latched from the environment at first use
std::atomic<unsigned> g_switches{0};
std::once_flag g_switches_once;
void init_switches()
{
    std::call_once(g_switches_once, [] {
        const auto off = [](const char *name) { const char *e = getenv(name); return e && e[0] == '0'; };
        unsigned v = 0;
        if (off("EMAVFI_CONV_FIRST")) v |= SW_NO_CONV_FIRST;
        if (off("EMAVFI_CONV_FIRSTRING")) v |= SW_NO_FIRSTRING;
        if (off("EMAVFI_CONV_HEAD")) v |= SW_NO_HEAD;
        if (off("EMAVFI_CONV_TAILFUSE")) v |= SW_NO_TAILFUSE;
        if (off("EMAVFI_CONV_LIGHT")) v |= SW_NO_CONV_LIGHT;
        if (off("EMAVFI_CONV_RING2")) v |= SW_NO_RING2;
        if (off("EMAVFI_CONV_POOLFUSE")) v |= SW_NO_POOLFUSE;
        if (off("EMAVFI_RING_CHUNK")) v |= SW_NO_RING_CHUNK;
        if (getenv("EMAVFI_NO_PERSISTENT_CONV") != nullptr) v |= SW_NO_PERSISTENT_CONV;
        if (const char *e = getenv("EMAVFI_RING_ONE_WG"); e && e[0] == '1') v |= SW_RING_ONE_WG;
        g_switches.store(v, std::memory_order_relaxed);
    });
}

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
inline size_t rup256(size_t v) { return (v + 255) & ~(size_t)255; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- which (CK, NF, stride) / (CK, NF) instantiations exist (keep in sync with the .inl lists) ----
const int kConvInst[][3] = {{16, 1, 1}, {16, 2, 1}, {32, 1, 1}, {48, 1, 1}, {64, 1, 1}, {64, 2, 1},
                            {64, 4, 1}, {80, 1, 1}, {80, 2, 1}, {16, 1, 2}, {32, 2, 2}, {32, 4, 2}};
// 16-bit types only: all eight output fragments of a stride-2 layer in one pass (128 accumulator registers)
const int kConvInst16[][3] = {{32, 8, 2}};
// fp32 only: 65..72 input channels as nine k-groups of 8
const int kConvInst32[][3] = {{72, 1, 1}, {72, 2, 1}};
const int kDeformInst[][2] = {{16, 1}, {32, 1}, {48, 2}, {80, 3}};

bool conv_inst_exists(int ck, int nf, int st, int esize)
{
    for (auto &i : kConvInst)
        if (i[0] == ck && i[1] == nf && i[2] == st) return true;
    if (esize == 2)
        for (auto &i : kConvInst16)
            if (i[0] == ck && i[1] == nf && i[2] == st) return true;
    if (esize == 4)
        for (auto &i : kConvInst32)
            if (i[0] == ck && i[1] == nf && i[2] == st) return true;
    return false;
}

struct Layer {
    // raw tensor
    int cout = 0, cin_raw = 0, cin_off = 0, cin_take = 0, stride = 1, perm = 0, param = 0;
    // packed geometry
    int cin_pad = 0, ck = 0, nchunk = 0, nf = 0, npass = 0, coutpad = 0;
    size_t w_off = 0, b_off = 0, w_bytes = 0;
    bool deform = false;
    bool f16_of_bf16 = false;  // bf16 model, layer consumed by deform_pack_kernel: bf16-rounded weights stored as f16
    bool mfma16 = false;       // packed for and run by conv3x3_persist16_kernel (v_mfma_f32_16x16x32): 16-bit full-resolution 64 -> (1..64) layers
    int pack3 = 0;             // deform_pack3.inl layouts: 1 = DCN, 2 = offset_conv (f16 elements)
    int ring = 0;              // ConvParams::ring: 1 = stride 2, 64 -> 128 (regular packing for ck 64, nf 4); 2 = 64 -> 64 (regular packing for
                               // ck 64, nf 2); 3 = 65..67 -> 64 (that + the im2col tail of channels 64..66, 6 KiB)
    bool first6 = false;       // feat_ext_conv1 at mid_channels 64, 16-bit: a second copy of the weights in conv_first.inl's layout (10 KiB)
    bool x3 = false;           // EMAVFI_F32X3: three virtual chunks per real one (w_hi, w_lo, w_hi), activations as [hi | lo] f16 halves
};

// Environment switches the PACKED LAYOUT depends on.  emavfi_pack_weights and emavfi_forward must agree on them, so the plan of
// a model reads them ONCE per process (process_layout_env: a switch flipped between packing and running cannot desynchronise
// the two); only the stage-level entry emavfi_conv3x3, which packs and runs inside one call, reads them per call (the
// parity tests compare kernels inside one process that way).  A blob packed by ANOTHER process under other switches is the
// caller's responsibility: the Python binding keys its caches by them (emavfi/lib.py, layout_switches).
struct LayoutEnv { bool m16_off, ring_off, s2ring_off, wreg_off; };
LayoutEnv read_layout_env()
{
    const auto off = [](const char *name) { const char *e = getenv(name); return e && e[0] == '0'; };
    return LayoutEnv{off("EMAVFI_CONV_MFMA16"), off("EMAVFI_CONV_RING"), off("EMAVFI_CONV_S2RING"), off("EMAVFI_CONV_WREG")};
}
const LayoutEnv &process_layout_env()
{
    static const LayoutEnv env = read_layout_env();
    return env;
}

// bit i of emavfi_layout_tag(): every process-wide switch a packed blob's LAYOUT depends on (the blob header carries the tag of the
// packing process; emavfi_packed_check and the forward's guard compare it with this process's)
bool pack_f16_chain();
unsigned layout_tag_of(const LayoutEnv &e)
{
    return (e.m16_off ? 1u : 0u) | (e.ring_off ? 2u : 0u) | (e.s2ring_off ? 4u : 0u) | (pack_f16_chain() ? 0u : 16u) |   // (bit 3 was round 2's EMAVFI_CONV_S2_CK64 experiment, removed in round 4)
          
           (deform16_can_fuse_offset_conv(80, 3, 67, 80, 1) ? 0u : 32u) | (e.wreg_off ? 64u : 0u);   // (EMAVFI_NO_FUSED_OFFSET, latched in deform_bf16.hip)
}

bool conv_geometry(Layer &L, int esize, const LayoutEnv &env_in, bool x3 = false)
{
    // EMAVFI_F32X3 runs every layer on the generic tile kernel (weights through LDS): the weights-in-registers kernels cannot hold three
    // weight sets; the virtual chunks are appended at the end of this function
    const LayoutEnv env = x3 ? LayoutEnv{true, true, true, true} : env_in;
    L.x3 = false;
    L.cin_pad = rup(L.cin_take, 16);
    // Stride-2 layers on the tile kernel: 32-channel chunks, two workgroups per CU (round 2 measured 64-channel chunks - every record read
    // once, but one workgroup per CU - SLOWER: 757 vs 513 us, 534 vs 513 us; that experiment and its instances were removed in round 4,
    // when conv_wreg.inl took the 256-channel layers)
    // 64 -> 128 at stride 2 (context_encoding.0): one 64-channel chunk, each wave keeps one output fragment's weights in registers
    // (conv3x3.inl, conv3x3_s2ring_kernel); EMAVFI_CONV_S2RING=0 keeps the 32-channel-chunk plan (changes the packing: set before packing)
    const bool s2r_off = env.s2ring_off;
    L.ring = (L.stride == 2 && esize == 2 && L.cin_pad == 64 && (L.cout + 31) / 32 == 4 && !s2r_off) ? 1 : 0;
    if (L.ring) {
        L.ck = 64; L.nchunk = 1; L.nf = 4; L.npass = 1; L.coutpad = 128;
        L.w_bytes = (size_t)9 * 4 * 4 * 1024;
        L.mfma16 = false;
        return true;
    }
    // 64 -> 64 and 65..67 -> 64 at stride 1 (feat_ext_blocks, motion_estimation.0 / .1, reconstruction.0): conv_ring.inl.
    // EMAVFI_CONV_RING=0 keeps round 2's plans (conv3x3_pingpong16_kernel / the CK = 80 tile kernel; changes the packing: set before packing)
    const bool ring_off = env.ring_off;
    if (L.stride == 1 && esize == 2 && L.cout > 32 && L.cout <= 64 && L.cin_take >= 64 && L.cin_take <= 67 && !ring_off) {
        L.ring = L.cin_take == 64 ? 2 : 3;
        L.ck = L.cin_pad; L.nchunk = 1; L.nf = 2; L.npass = 1; L.coutpad = 64;
        L.w_bytes = (size_t)9 * 4 * 2 * 1024 + (L.ring == 3 ? 3 * 2 * 1024 : 0);
        L.mfma16 = false;
        return true;
    }
    // 256 output channels from >= 128 inputs (context_encoding.1 / .2 at mid_channels 64): conv_wreg.inl - all eight output fragments in one
    // pass, weights streamed into registers.  EMAVFI_CONV_WREG=0 keeps round 3's plans (changes the packing: set before packing)
    if (esize == 2 && L.cout > 224 && L.cout <= 256 && L.cin_pad >= 128 && !env.wreg_off &&
        ((L.stride == 1 && L.cin_pad % 64 == 0) || (L.stride == 2 && L.cin_pad % 32 == 0))) {
        L.ring = 4;
        L.ck = L.stride == 1 ? 64 : 32; L.nchunk = L.cin_pad / L.ck; L.nf = 8; L.npass = 1; L.coutpad = 256;
        L.w_bytes = (size_t)L.nchunk * 9 * (L.ck / 16) * 8 * 1024;
        L.mfma16 = false;
        return true;
    }
    if (L.stride == 2) L.ck = (L.cin_pad % 32 == 0) ? 32 : 16;
    else if (esize == 4 && L.cin_take > 64 && L.cin_take <= 72) { L.ck = 72; L.cin_pad = 72; }   // fp32 k-groups are 8 channels: 9 instead of 10
    else if (L.cin_pad <= 80) L.ck = L.cin_pad;
    else if (L.cin_pad % 64 == 0) L.ck = 64;
    else return false;
    L.nchunk = L.cin_pad / L.ck;
    const int frags = (L.cout + 31) / 32;
    L.nf = (frags % 4 == 0) ? 4 : (frags % 2 == 0) ? 2 : 1;
    // 16-bit stride-2 layers with >= 256 output channels take all of them in ONE pass (8 fragments, 128 accumulator
    // registers per lane): every extra pass re-reads the whole input from HBM (FETCH_SIZE calibration,
    // tools/microbench/fetch_calib.hip: reading 64 bytes of every record fetches every 128-byte line)
    if (L.stride == 2 && esize == 2 && frags % 8 == 0 && L.ck == 32) L.nf = 8;
    if (L.stride == 2 && L.ck == 16) L.nf = 1;
    if (L.stride == 2 && L.ck == 32 && L.nf == 1) { L.ck = 16; L.nchunk = L.cin_pad / 16; }
    L.npass = frags / L.nf;
    L.coutpad = frags * 32;
    // fall back to narrower fragments if the preferred width has no instantiation
    while (!conv_inst_exists(L.ck, L.nf, L.stride, esize) && L.nf > 1) { L.nf /= 2; L.npass = frags / L.nf; }
    if (!conv_inst_exists(L.ck, L.nf, L.stride, esize)) return false;
    L.w_bytes = (size_t)L.npass * L.nchunk * 9 * (L.ck * esize / 32) * L.nf * 1024;
    // full-resolution 64-channel layers with two output fragments: the 16x16x32 MFMA shape (conv3x3.inl, conv3x3_persist16_kernel).
    // EMAVFI_CONV_MFMA16=0 keeps the 32x32x16 kernels.
    L.mfma16 = esize == 2 && L.stride == 1 && L.ck == 64 && (L.nf == 2 || L.nf == 1) && L.nchunk == 1 && L.npass == 1 && !env.m16_off;
    // 32 -> <= 4 channels (reconstruction.2): the planar-head kernel on 16x16x32 (conv_light.inl) reads the same regrouped packing
    if (esize == 2 && L.stride == 1 && L.ck == 32 && L.nf == 1 && L.nchunk == 1 && L.npass == 1 && L.cout <= 4 && !env.m16_off) L.mfma16 = true;
    if (x3) {
        if (esize != 2 || L.mfma16 || L.ring) return false;
        L.x3 = true;
        L.nchunk *= 3;      // virtual chunks 3 c + t
        L.w_bytes *= 3;
    }
    return true;
}

bool deform_geometry(Layer &L, int esize)
{
    const int cpad = rup(L.cin_take, 16), frags = (L.cout + 31) / 32;
    for (auto &i : kDeformInst)
        if (i[0] >= cpad && i[1] >= frags) {
            L.cin_pad = L.ck = i[0];
            L.nchunk = 1;
            L.nf = i[1];
            L.npass = 1;
            L.coutpad = L.nf * 32;
            L.w_bytes = (size_t)9 * (L.ck * esize / 32) * L.nf * 1024;
            L.deform = true;
            return true;
        }
    return false;
}

// bf16 model: consecutive one-launch packs hand each other f16 bit patterns (the window is f16 on chip anyway; the
// receiving pack skips its in-LDS conversion pass).  EMAVFI_PACK_F16_CHAIN=0 keeps bf16 between them (A/B switch).
bool pack_f16_chain()
{
    static const bool off = [] { const char *e = getenv("EMAVFI_PACK_F16_CHAIN"); return e && e[0] == '0'; }();
    return !off;
}

constexpr int kMaxBlocks = 8;

struct Plan {
    bool feat16 = false;       // bf16 model with the one-launch packs: `feat` (and the warped tail) is stored as IEEE f16 - what the first
                               // pack wants on chip (no conversion pass in it) -; its other readers (context_encoding.0,
                               // motion_estimation.0) run the f16 ring kernels on bf16-rounded weights and hand bf16 on
    int in_ch, mid, nb, dtype, esize;  // dtype = the KERNEL storage type (EMAVFI_AMP16 runs the f16 kernels)
    bool amp;                          // EMAVFI_AMP16: autocast op policy (fp32 DCN on an fp32 fusion tensor, fp16 roundings)
    bool x3;                           // EMAVFI_F32X3: fp32-accurate three-term f16 split in every convolution, exact fp32 DCN; the data flow of
                                       // the autocast mode (an fp32 fusion tensor beside the 16-bit one) without any of its roundings
    bool wide() const { return amp || x3; }   // the fusion tensor also exists in fp32 (what the fp32 DCN reads and writes)
    Layer dcn32[kMaxBlocks];           // amp: the deformable convolutions' fp32 master weights
    int fpad, p_mid;  // padded fusion / feature widths
    int fps;          // pixel stride (elements) of the fusion buffers: 72 when the 16-bit LDS-window pack serves them (the
                      // only channels any consumer reads; 144-byte pixels instead of 160: -10 % HBM traffic on every tensor
                      // the three packs and reconstruction.0 touch), else fpad
    Layer conv1, blk[kMaxBlocks], c0, c1, c2, m0, m1, m2, off[kMaxBlocks], dcn[kMaxBlocks], r0, r1, r2;
    Layer offh[kMaxBlocks];  // bf16 model at the reference width: second copy of offset_conv for the one-launch pack
    bool has_offh;
    int lin_param;
    size_t ctx_off, zero_off, total;
    const char *why;
};

Layer mk(int param, int cout, int cin_raw, int stride = 1, int cin_off = 0, int cin_take = -1, int perm = 0)
{
    Layer L;
    L.param = param; L.cout = cout; L.cin_raw = cin_raw; L.stride = stride;
    L.cin_off = cin_off; L.cin_take = cin_take < 0 ? cin_raw : cin_take; L.perm = perm;
    return L;
}

bool build_plan(Plan &P, int in_ch, int mid, int nb, int dtype)
{
    P.why = "";
    if (dtype != EMAVFI_F32 && dtype != EMAVFI_BF16 && dtype != EMAVFI_F16 && dtype != EMAVFI_AMP16 && dtype != EMAVFI_F32X3) {
        P.why = "dtype must be EMAVFI_F32, EMAVFI_BF16, EMAVFI_F16, EMAVFI_AMP16 or EMAVFI_F32X3";
        return false;
    }
    P.amp = dtype == EMAVFI_AMP16;
    P.x3 = dtype == EMAVFI_F32X3;
    if (P.wide()) dtype = EMAVFI_F16;  // every convolution runs the f16 kernels; what differs is in forward_impl / pack
    // the reference's fusion width is the literal mid_channels + 3 (ema_vfi.py:97): with any other in_channels its
    // forward raises a channel mismatch at the first attention block, so this build refuses instead of padding / dropping
    if (in_ch != 3) { P.why = "in_channels must be 3 (the reference's fusion width is mid_channels + 3, ema_vfi.py:97)"; return false; }
    if (nb < 1 || nb > kMaxBlocks) { P.why = "num_blocks must be 1..8"; return false; }
    if (mid < 8 || mid % 8 != 0) { P.why = "mid_channels must be a positive multiple of 8"; return false; }
    P.in_ch = in_ch; P.mid = mid; P.nb = nb; P.dtype = dtype; P.esize = dtype == EMAVFI_F32 ? 4 : 2;
    const int f = mid + 3;  // ema_vfi.py:97 (the +3 is literal in the reference)
    P.fpad = rup(f, 16); P.p_mid = rup(mid, 16);
    int q = 0;  // running index into the state_dict tensor list (weight, bias pairs)
    P.conv1 = mk(q, mid, 2 * in_ch); q += 2;
    for (int i = 0; i < nb; ++i) { P.blk[i] = mk(q, mid, mid); q += 2; }
    P.c0 = mk(q, 2 * mid, mid, 2); q += 2;
    P.c1 = mk(q, 4 * mid, 2 * mid, 2); q += 2;
    P.c2 = mk(q, 4 * mid, 4 * mid); q += 2;
    P.lin_param = q; q += 2;
    P.m0 = mk(q, mid, 2 * mid, 1, 0, mid); q += 2;  // feat half only; ctx half is folded into the bias table
    P.m1 = mk(q, mid, mid); q += 2;
    P.m2 = mk(q, 2, mid); q += 2;
    for (int i = 0; i < nb; ++i) {
        P.off[i] = mk(q, 27, f, 1, 0, -1, 1); q += 2;
        P.dcn[i] = mk(q, f, f); q += 2;
    }
    P.r0 = mk(q, mid, f); q += 2;
    P.r1 = mk(q, mid / 2, mid); q += 2;
    P.r2 = mk(q, in_ch, mid / 2); q += 2;

    size_t o = kBlobHeaderBytes;   // the blob's header (misc_kernels.h, BlobHeader) comes first
    auto place = [&](Layer &L, bool deform) {
        const bool ok = deform ? deform_geometry(L, P.esize) : conv_geometry(L, P.esize, process_layout_env(), P.x3);
        if (!ok) return false;
        L.w_off = o; o = rup256(o + L.w_bytes);
        L.b_off = o; o = rup256(o + (size_t)L.coutpad * sizeof(float));
        return true;
    };
    // 16-bit, 6 -> 64: the fused cat + conv kernel (conv_first.inl) reads its own 10 KiB fragment copy behind the regular one
    // (the blob always carries both layouts: EMAVFI_CONV_FIRST=0, read per forward, runs pack_input + conv3x3<16,2,1> - A/B, parity test)
    bool ok = conv_geometry(P.conv1, P.esize, process_layout_env(), P.x3);
    if (ok) {
        P.conv1.first6 = P.esize == 2 && P.conv1.cout == 64 && P.conv1.cin_take == 6 && !P.x3;
        P.conv1.w_off = o; o = rup256(o + P.conv1.w_bytes + (P.conv1.first6 ? 10240 : 0));
        P.conv1.b_off = o; o = rup256(o + (size_t)P.conv1.coutpad * sizeof(float));
    }
    for (int i = 0; i < nb && ok; ++i) ok = place(P.blk[i], false);
    ok = ok && place(P.c0, false) && place(P.c1, false) && place(P.c2, false);
    ok = ok && place(P.m0, false) && place(P.m1, false) && place(P.m2, false);
    for (int i = 0; i < nb && ok; ++i) ok = place(P.off[i], false) && place(P.dcn[i], true);
    ok = ok && place(P.r0, false) && place(P.r1, false) && place(P.r2, false);
    // The one-launch pack kernel (deform_pack.inl) keeps its window in f16 on chip whatever the storage type: in a bf16
    // model its two weight sets are the bf16-rounded values stored as f16.  The stand-alone offset_conv (conv3x3, used
    // when the pack is not fused) still reads the bf16 copy.
    P.fps = P.fpad;
    if (ok && !P.wide() && dtype != EMAVFI_F32 && nb > 0 && deform_pack3_shape(P.dcn[0].ck, P.dcn[0].nf, P.dcn[0].cin_take, P.dcn[0].cout)) P.fps = 72;
    if (ok && P.wide()) {
        for (int i = 0; i < nb && ok; ++i) {
            P.dcn32[i] = mk(P.dcn[i].param, f, f);
            ok = deform_geometry(P.dcn32[i], 4);

            P.dcn32[i].w_off = o; o = rup256(o + P.dcn32[i].w_bytes);
            P.dcn32[i].b_off = o; o = rup256(o + (size_t)P.dcn32[i].coutpad * sizeof(float));
        }
        if (!ok) { P.why = "no fp32 deformable-conv instantiation for these channel widths"; return false; }
    }
    if (ok && dtype == EMAVFI_F32)
        for (int i = 0; i < nb; ++i)
            if (deform_f32w_shape(P.dcn[i].ck, P.dcn[i].nf, P.dcn[i].cin_take, P.dcn[i].cout)) P.dcn[i].pack3 = 3;
    if (ok && P.wide())
        for (int i = 0; i < nb; ++i)
            if (deform_f32w_shape(P.dcn32[i].ck, P.dcn32[i].nf, P.dcn32[i].cin_take, P.dcn32[i].cout)) {
                P.dcn32[i].pack3 = 3;
                // deform_f32w.inl's x3 form (weights as f16 (hi, lo) pairs, three 16x16x16 f16 MFMAs per K = 16 step: 22 bits of each operand,
                // exact products, fp32 accumulation - the error class of an fp32 convolution's own summation order).  Also the autocast-policy
                // mode's DCN since round 6: torchvision's CUDA kernel is an fp32 computation in ANOTHER summation order, every result of this
                // mode is rounded to fp16 behind it, and the op is 73 % of that mode's step (6.9 -> 4.7 ms per launch).  EMAVFI_F32 stays exact.
                P.dcn32[i].x3 = P.x3 || P.amp;
            }
    P.has_offh = false;
    P.feat16 = false;
    const bool p3 = ok && !P.wide() && dtype != EMAVFI_F32 && nb > 0 && deform_pack3_shape(P.dcn[0].ck, P.dcn[0].nf, P.dcn[0].cin_take, P.dcn[0].cout) &&
                    P.off[0].nchunk == 1 && P.off[0].npass == 1 && P.off[0].ck == 80 && P.off[0].nf == 1;
    if (p3) {
        // a second copy of offset_conv for the one-launch pack: f16 fragments (bf16 model: the bf16-rounded values), and in
        // the deform_pack3.inl layout where that kernel serves the shape (both 16-bit models); off[i] keeps the conv3x3 layout
        P.has_offh = true;
        // (feat16: see Plan; needs the ring kernels on both sides of `feat` and at least one block in front of the last one)
        P.feat16 = dtype == EMAVFI_BF16 && pack_f16_chain() && nb >= 2 && P.blk[nb - 1].ring == 2 && P.c0.ring == 1 && P.m0.ring == 2 && P.fpad - mid == 16 &&
                   deform16_can_fuse_offset_conv(P.dcn[0].ck, P.dcn[0].nf, P.dcn[0].cin_take, P.off[0].ck, P.off[0].nf);   // (EMAVFI_NO_FUSED_OFFSET: once per process)
        if (P.feat16) P.c0.f16_of_bf16 = P.m0.f16_of_bf16 = true;
        for (int i = 0; i < nb && ok; ++i) {
            P.dcn[i].f16_of_bf16 = dtype == EMAVFI_BF16;
            P.dcn[i].pack3 = p3 ? 1 : 0;
            P.offh[i] = P.off[i];
            P.offh[i].f16_of_bf16 = dtype == EMAVFI_BF16;
            P.offh[i].pack3 = p3 ? 2 : 0;
            P.offh[i].w_off = o; o = rup256(o + P.offh[i].w_bytes);  // shares off[i]'s bias
        }
    }
    if (!ok) { P.why = "no kernel instantiation for these channel widths (built: mid_channels 8, 16, 32, 64)"; return false; }
    if (256 % rup(4 * mid, 16) != 0) { P.why = "4*mid_channels must divide 256 for the pooling kernel"; return false; }
    P.ctx_off = o;
    o = rup256(o + ((size_t)mid * 4 * mid + mid + (size_t)mid * mid * 9 + mid) * sizeof(float));
    P.zero_off = o;
    o += 256;
    P.total = o;
    return true;
}

#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
unsigned long long *debug_stamp_buffer();
#endif
int run_conv(const Plan &P, const Layer &L, const void *packed, const void *in, int in_ps, int Hin, int Win, void *out,
             int out_ps, int out_coff, int cstore, int epi, int B, hipStream_t s, const float *bias_table = nullptr,
             float *planar = nullptr, int nplanes = 0, const void *zeros = nullptr, const Layer *head = nullptr,
             const FirstParams *first = nullptr, int epi2 = 0, int out_alt = 0, const Layer *second = nullptr, float *pool_part = nullptr,
             float *out32 = nullptr, int out32_ps = 0)
{
    ConvParams c{};
    c.out32 = L.x3 ? out32 : nullptr; c.out32_ps = out32_ps;
    c.pool_part = pool_part;
    if (second) {   // conv_ring2.inl: `second` runs behind L in the same launch; out / out_ps / cstore / out_alt are ITS output's
        c.w2 = (const char *)packed + second->w_off;
        c.bias2 = (const float *)((const char *)packed + second->b_off);
    }
    c.epi2 = epi2;
    c.out_alt = out_alt;
    c.out_fill = (L.ring == 2 && !head && !first && out && out_coff == 0 && cstore == 64 && (size_t)out_ps * P.esize == 144) ? 1 : 0;   // (also the fused pair's second layer)
    if (head) {   // conv_ring.inl / conv_ring_tail.inl: L's rows stay in LDS, the planar head `head` is computed from them (planar / nplanes are the head's)
        c.head_w = (const char *)packed + head->w_off;
        c.head_bias = (const float *)((const char *)packed + head->b_off);
    }
    // a single-chunk layer wider than its input's pixel stride (CK = 80 fed from the 72-channel fusion buffers) reads
    // the missing pieces as zeros
    if (L.nchunk == 1 && in_ps < L.ck) c.in_pieces = in_ps * P.esize / 16;
    c.round16 = P.amp ? 1 : 0;
    if (L.x3) {   // EMAVFI_F32X3: pixels are [hi | lo] halves of the widths the caller names
        c.x3 = 1; c.x3_lo_off = in_ps; in_ps *= 2;
        if (epi == EPI_NONE || epi == EPI_RELU) { c.out_lo_off = out_ps; out_ps *= 2; }   // (EPI_OM / planar outputs are fp32 as in every mode)
    }
    c.zeros = zeros ? zeros : (const char *)packed + P.zero_off;
    c.in = in; c.out = out; c.out_planar = planar;
    c.w = (const char *)packed + L.w_off;
    c.bias = bias_table ? bias_table : (const float *)((const char *)packed + L.b_off);
    c.in_ps = in_ps; c.out_ps = out_ps; c.out_coff = out_coff;
    c.Hin = Hin; c.Win = Win;
    c.Hout = (Hin + L.stride - 1) / L.stride; c.Wout = (Win + L.stride - 1) / L.stride;
    c.B = B; c.nchunk = L.nchunk; c.npass = L.npass; c.cstore = cstore;
    c.epi = epi; c.nplanes = nplanes; c.bias_mode = bias_table ? 1 : 0;
    c.ck = L.ck; c.nf = L.nf; c.stride = L.stride; c.mfma16 = L.mfma16 ? 1 : 0; c.ring = L.ring;
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    if (L.ring == 4 && !planar && !getenv("EMAVFI_STAMP_RING")) c.out_planar = reinterpret_cast<float *>(debug_stamp_buffer());   // diagnostic build: conv_wreg.inl's stamps (tools/wreg_stamps.py)
    {   // diagnostic build: the LDS-ring kernels' stamps (tools/ring_stamps.py); EMAVFI_STAMP_RING = ring | tail | head | ringtail | ringfirst
        const char *sel = getenv("EMAVFI_STAMP_RING");
        const char *kind = first ? "ringfirst" : (head && L.ring == 2) ? "head" : (head && L.mfma16) ? "ringtail" : L.ring == 3 ? "tail" : (L.ring == 2 && !second) ? "ring" : "";
        if (sel && kind[0] && strcmp(sel, kind) == 0) c.stamps = debug_stamp_buffer();
    }
#endif
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    if (const char *rep = getenv("EMAVFI_DEBUG_REPEAT_CONV"))   // diagnostic build: as EMAVFI_DEBUG_REPEAT_PACK, for the plain layers
        if (!first && !(L.f16_of_bf16 && !L.deform))
            for (int i = 1; i < atoi(rep); ++i)
                if (const int rc = P.dtype == EMAVFI_F32 ? launch_conv3x3_f32(c, s) : P.dtype == EMAVFI_F16 ? launch_conv3x3_f16(c, s) : launch_conv3x3_bf16(c, s)) return rc;
#endif
    if (first) return P.dtype == EMAVFI_F16 ? launch_conv_ringfirst_f16(*first, c, s) : launch_conv_ringfirst_bf16(*first, c, s);
    if (L.f16_of_bf16 && !L.deform) return launch_conv3x3_f16(c, s);   // feat16: f16 activations in, bf16-rounded weights stored as f16
    return P.dtype == EMAVFI_F32 ? launch_conv3x3_f32(c, s) : P.dtype == EMAVFI_F16 ? launch_conv3x3_f16(c, s) : launch_conv3x3_bf16(c, s);
}

#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
// diagnostic build only (never the shipped library): a lazily allocated stamp buffer and its reader
unsigned long long *debug_stamp_buffer()
{
    static unsigned long long *buf = nullptr;
    if (!buf && hipMalloc(&buf, (size_t)DEFORM_STAMP_ROWS * 8 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(buf, 0, (size_t)DEFORM_STAMP_ROWS * 8 * sizeof(unsigned long long));
    return buf;
}
#endif

int run_deform(const Plan &P, const Layer &L, const void *packed, const void *x, int x_ps, float *om, void *out,
               int out_ps, int cstore, int B, int H, int W, hipStream_t s, const void *zeros = nullptr, const Layer *off = nullptr,
               const void *x_tail = nullptr, int tail_ps = 0, int force_dtype = -1, int in_f16 = 0, int out_f16 = 0, void *out16 = nullptr, int out16_ps = 0,
               unsigned *census = nullptr, int out16_lo_off = 0)
{
    const int kd = force_dtype >= 0 ? force_dtype : P.dtype;
    DeformParams d{};
    d.census = census;
    d.out16_lo_off = out16_lo_off;
    d.x3 = L.x3 && L.pack3 == 3 ? 1 : 0;
    d.x = x; d.om = om; d.out = out;
    d.x_tail = x_tail; d.tail_ps = tail_ps;
    if (off) {  // fused ModulatedDeformConvPack: the kernel computes om itself (off = the copy in the kernel's on-chip type)
        d.off_w = (const char *)packed + off->w_off;
        d.off_bias = (const float *)((const char *)packed + off->b_off);
    }
    d.w = (const char *)packed + L.w_off;
    d.bias = (const float *)((const char *)packed + L.b_off);
    d.zeros = zeros ? zeros : (const char *)packed + P.zero_off;
    d.x_ps = x_ps; d.out_ps = out_ps; d.H = H; d.W = W; d.B = B; d.cstore = cstore; d.cin_real = L.cin_take; d.ck = L.ck; d.nf = L.nf;
    d.cout_real = L.cout; d.pack3 = L.pack3; d.in_f16 = in_f16; d.out_f16 = out_f16;
    d.out16 = out16; d.out16_ps = out16_ps;
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    {   // diagnostic build: EMAVFI_STAMP_PACK=i records only the i-th deformable launch of every 3 (default: every launch,
        // i.e. what is read back is the last pack of a forward)
        static int calls = 0;
        const char *sel = getenv("EMAVFI_STAMP_PACK");
        if ((!sel || (calls % 3) == atoi(sel)) && !getenv("EMAVFI_STAMP_RING")) d.stamps = debug_stamp_buffer();   // (the ring kernels' stamps share the buffer)
        ++calls;
    }
#endif
#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
    // diagnostic build: EMAVFI_DEBUG_REPEAT_PACK=n launches the kernel n times back to back (same operands, same result: the input is
    // not the output) - what the board's power and clock are under THIS kernel alone (tools/power_per_kernel.py)
    if (const char *rep = getenv("EMAVFI_DEBUG_REPEAT_PACK"))
        for (int i = 1; i < atoi(rep); ++i)
            if (const int rc = kd == EMAVFI_F32 ? launch_deform_f32(d, s) : kd == EMAVFI_F16 ? launch_deform_f16(d, s) : launch_deform_bf16(d, s)) return rc;
#endif
    return kd == EMAVFI_F32 ? launch_deform_f32(d, s) : kd == EMAVFI_F16 ? launch_deform_f16(d, s) : launch_deform_bf16(d, s);
}

int pack_layer(const Layer &L, const void *const *params, void *packed, int dtype, hipStream_t s, bool bias_f16 = false)
{
    PackDesc d{L.cout, L.cin_raw, L.cin_off, L.cin_take, L.ck, L.nchunk, L.nf, L.npass, L.perm, L.f16_of_bf16 ? 1 : 0, bias_f16 ? 1 : 0};
    d.mfma16 = L.mfma16 ? 1 : 0;
    d.ring = L.ring;
    d.first6 = L.first6 ? 1 : 0;
    d.pack3 = L.pack3;
    d.x3 = L.x3 ? 1 : 0;
    return launch_pack_conv((const float *)params[L.param], (const float *)params[L.param + 1], (char *)packed + L.w_off,
                            (float *)((char *)packed + L.b_off), d, L.f16_of_bf16 ? (int)EMAVFI_F16 : dtype, s);
}

struct Workspace {
    char *base; size_t cap, used;
    void *take(size_t bytes) { void *p = base ? base + used : nullptr; used = rup256(used + bytes); return p; }
};

constexpr size_t kCensusBlock = (size_t)DEFORM_CENSUS_SLOTS * 4 * sizeof(unsigned);   // one launch's census
constexpr size_t kCensusBytes = (size_t)kMaxBlocks * kCensusBlock;

// The compact tail of the fusion input (channels mid .. mid + 2, the warped frame): FOUR 16-bit channels = 8 bytes per pixel (round 6; was 8
// channels = 16 bytes).  The pack kernel's window DMA fetches 16 bytes per pixel from it - the pixel's tail and its right neighbour's,
// which land in channels 68..71 of the window slot: finite values against zero weights, never read by the 8-byte tail reads - so the
// buffer keeps 8 bytes of slack behind its last pixel (it lives in the 16-channel packed-input buffer: plenty).  The warp then moves
// 8 + 12 + 8 = 28 bytes per pixel instead of 36.
constexpr int kTailPs = 4;

struct FwdBuffers {
    void *in16, *fA, *fB, *fu0, *fu1, *c1, *c2, *c3;
    float *fuF0, *fuF1;  // amp: fp32 copies of the fusion tensor (input / output of the fp32 DCN)
    float *part, *ctx, *table, *flow, *om;
    float *tpart;        // context_encoding.2 fused with the pool (conv_wreg.inl): per-tile channel sums [B][ntiles][p4]
    int ntiles, nparts2; // ... and the partial sums avg_pool_partial reduces them to
    unsigned *census;    // deform_pack3.inl's per-launch census: [kMaxBlocks][DEFORM_CENSUS_SLOTS][4] u32, zeroed at the head of every forward
    int nparts, H2, W2, H4, W4, p2, p4, p_half;
};

void carve_forward(const Plan &P, Workspace &ws, FwdBuffers &f, int B, int H, int W)
{
    const size_t px = (size_t)B * H * W, e = (size_t)P.esize * (P.x3 ? 2 : 1);   // (EMAVFI_F32X3: [hi | lo] halves per pixel)
    f.H2 = (H + 1) / 2; f.W2 = (W + 1) / 2; f.H4 = (f.H2 + 1) / 2; f.W4 = (f.W2 + 1) / 2;
    f.p2 = rup(2 * P.mid, 16); f.p4 = rup(4 * P.mid, 16); f.p_half = rup(P.mid / 2, 16);
    const int npix4 = f.H4 * f.W4;
    f.nparts = npix4 >= 256 * 64 ? 256 : (npix4 >= 64 ? npix4 / 64 : 1);  // >= 64 pixels per partial sum
    f.in16 = ws.take(px * 16 * e);
    f.fA = ws.take(px * P.p_mid * e);
    f.fB = ws.take(px * P.p_mid * e);
    f.fu0 = ws.take(px * P.fps * e);
    f.fu1 = ws.take(px * P.fps * e);
    f.c1 = ws.take((size_t)B * f.H2 * f.W2 * f.p2 * e);
    f.c2 = ws.take((size_t)B * npix4 * f.p4 * e);
    f.c3 = ws.take((size_t)B * npix4 * f.p4 * e);
    f.part = (float *)ws.take((size_t)B * f.nparts * f.p4 * sizeof(float) * (P.x3 ? 2 : 1));
    f.ntiles = ((f.W4 + 31) / 32) * ((f.H4 + 3) / 4);   // conv_wreg_tiles()
    f.nparts2 = f.ntiles >= 32 ? (f.ntiles / 16 < 32 ? f.ntiles / 16 : 32) : 1;
    if (f.nparts2 > f.nparts) f.nparts2 = f.nparts;
    f.tpart = P.c2.ring == 4 ? (float *)ws.take((size_t)B * f.ntiles * f.p4 * sizeof(float)) : nullptr;
    f.ctx = (float *)ws.take((size_t)B * P.mid * sizeof(float));
    f.table = (float *)ws.take((size_t)B * 16 * P.m0.coutpad * sizeof(float));
    f.flow = (float *)ws.take(px * 2 * sizeof(float));
    f.om = (float *)ws.take(px * 32 * sizeof(float));
    f.fuF0 = f.fuF1 = nullptr;
    if (P.wide()) {
        f.fuF0 = (float *)ws.take(px * P.fpad * sizeof(float));
        f.fuF1 = (float *)ws.take(px * P.fpad * sizeof(float));
    }
    f.census = (unsigned *)ws.take(kCensusBytes);   // (last: every other offset is what it was before round 6)
}

// ---- launch recorder: names every kernel launch of a forward, its algorithmic work, and can
// bracket each launch with caller-owned hipEvents on the launch stream (bench.py's roofline leg).
struct LaunchRec { std::string name; double flops, bytes; };
struct Recorder {
    std::vector<LaunchRec> *recs = nullptr;  // filled when enumerating
    void *const *events = nullptr;           // 2 per launch: start, stop
    int n_events = 0, idx = 0;
    bool dry = false;                        // enumerate only, launch nothing
    void *const *stage_events = nullptr;     // emavfi_forward_staged: up to 3 hipEvent_t recorded behind the front / the attention blocks / the reconstruction
};

// records stage event i of a staged forward on the launch stream (no-op for every other entry)
#define EMAVFI_STAGE_EVENT(rec, i)                                                                              \
    do {                                                                                                        \
        if (!(rec).dry && (rec).stage_events && (rec).stage_events[i] &&                                        \
            hipEventRecord((hipEvent_t)(rec).stage_events[i], s) != hipSuccess)                                 \
            return fail(EMAVFI_E_LAUNCH, "forward_staged: hipEventRecord of stage event %d failed", (int)(i));  \
    } while (0)

const char *dtype_name(int dtype) { return dtype == EMAVFI_F32 ? "f32" : dtype == EMAVFI_F16 ? "f16" : "bf16"; }

std::string conv_name(const Plan &P, const Layer &L)
{
    char b[96];
    snprintf(b, sizeof b, "conv3x3<%s,ck=%d,nf=%d,s=%d>", L.f16_of_bf16 ? "f16" : dtype_name(P.dtype), L.ck, L.nf, L.stride);   // (feat16: f16 kernels in a bf16 model)
    return b;
}
std::string deform_name(const Plan &P, const Layer &L)
{
    char b[96];
    snprintf(b, sizeof b, "deform<%s,ck=%d,nf=%d>", dtype_name(P.dtype), L.ck, L.nf);
    return b;
}
// algorithmic work of one conv launch: real (unpadded) channels, every tensor touched once
void conv_work(const Plan &P, const Layer &L, int B, int Hin, int Win, double out_esize, double &flops, double &bytes)
{
    const double Ho = (Hin + L.stride - 1) / L.stride, Wo = (Win + L.stride - 1) / L.stride;
    flops = 2.0 * 9.0 * L.cin_take * L.cout * Ho * Wo * B;
    bytes = (double)P.esize * L.cin_take * Hin * Win * B + out_esize * L.cout * Ho * Wo * B + 9.0 * L.cin_take * L.cout * P.esize;
}

#define EMAVFI_TRY(expr, what)                                                                       \
    do {                                                                                             \
        const int e_ = (expr);                                                                       \
        if (e_ == -2) return fail(EMAVFI_E_UNSUPPORTED, "%s: no kernel instantiation", what);        \
        if (e_ != 0) return fail(EMAVFI_E_LAUNCH, "%s: %s", what, hipGetErrorString((hipError_t)e_)); \
    } while (0)

#define EMAVFI_STEP(rec, label, fl, by, expr)                                                          \
    do {                                                                                               \
        if ((rec).recs) (rec).recs->push_back(LaunchRec{(label), (double)(fl), (double)(by)});         \
        if (!(rec).dry) {                                                                              \
            const bool ev_ = (rec).events && 2 * (rec).idx + 1 < (rec).n_events;                       \
            if (ev_ && hipEventRecord((hipEvent_t)(rec).events[2 * (rec).idx], s) != hipSuccess)       \
                return fail(EMAVFI_E_LAUNCH, "hipEventRecord failed");                                 \
            EMAVFI_TRY(expr, "forward");                                                               \
            if (ev_ && hipEventRecord((hipEvent_t)(rec).events[2 * (rec).idx + 1], s) != hipSuccess)   \
                return fail(EMAVFI_E_LAUNCH, "hipEventRecord failed");                                 \
        }                                                                                              \
        (rec).idx++;                                                                                   \
    } while (0)

// ---- one ModulatedDeformConvPack.forward (ema_vfi.py:53-60) as the forward runs it; shared by forward_impl and the stage-level entry
// emavfi_mdcn so that the stage test exercises exactly the product's routing, packing and flags.
// Whether block i runs as ONE launch (offset_conv computed on the staged window, offsets / masks never leave the registers)
bool pack_fuses(const Plan &P, int i)
{
    return P.dtype != EMAVFI_F32 && !P.amp && P.off[i].nchunk == 1 && P.off[i].npass == 1 && P.off[i].stride == 1 &&
           deform16_can_fuse_offset_conv(P.dcn[i].ck, P.dcn[i].nf, P.dcn[i].cin_take, P.off[i].ck, P.off[i].nf) &&
           (P.dcn[i].pack3 == 0 || P.has_offh);
}
// bf16 model, consecutive one-launch packs in the deform_pack3 layout: pack i writes f16 bit patterns for pack i + 1
bool pack_f16_link(const Plan &P, int i)
{
    return P.dtype == EMAVFI_BF16 && pack_f16_chain() && i >= 0 && i + 1 < P.nb && P.dcn[i].pack3 && P.dcn[i + 1].pack3 && pack_fuses(P, i) &&
           pack_fuses(P, i + 1);
}
// 16-bit / fp32 models: x -> y (channels-last, pixel stride P.fps).  x_tail: the compact tail buffer (kTailPs channels per pixel) the first pack takes channels
// 64.. from (or null); in_f16 / out_f16: the bf16 model's f16 hand-off (DeformParams)
int attention_block(const Plan &P, int i, const void *packed, const void *x, void *y, float *om, const void *x_tail, int in_f16, int out_f16,
                    int B, int H, int W, hipStream_t s, Recorder &rec, unsigned *census = nullptr)
{
    const double px = (double)B * H * W, e = P.esize, cf = P.mid + 3;
    double fl, by;
    conv_work(P, P.off[i], B, H, W, 4.0, fl, by);
    if (pack_fuses(P, i)) {
        // one launch for the whole pack (ema_vfi.py:54-60): offset_conv on the staged window, then the DCN;
        // the input is read once and the offsets / masks never leave the registers
        EMAVFI_STEP(rec, deform_name(P, P.dcn[i]) + " offset_conv+dcn_v2", fl + 2.0 * 9.0 * cf * cf * px, px * (2.0 * cf * e) + 9.0 * cf * (cf + 27.0) * e,
                    run_deform(P, P.dcn[i], packed, x, P.fps, om, y, P.fps, P.fps, B, H, W, s, nullptr, P.has_offh ? &P.offh[i] : &P.off[i], x_tail, kTailPs,
                               -1, in_f16, out_f16, nullptr, 0, census));
    } else {
        if (x_tail || in_f16 || out_f16) return fail(EMAVFI_E_UNSUPPORTED, "attention block: split tail / f16 hand-off need the one-launch pack kernel");
        EMAVFI_STEP(rec, conv_name(P, P.off[i]) + " offset_conv", fl, by, run_conv(P, P.off[i], packed, x, P.fps, H, W, om, 32, 0, 32, EPI_OM, B, s));
        EMAVFI_STEP(rec, deform_name(P, P.dcn[i]) + " dcn_v2", 2.0 * 9.0 * cf * cf * px, px * (2.0 * cf * e + 27.0 * 4.0) + 9.0 * cf * cf * e,
                    run_deform(P, P.dcn[i], packed, x, P.fps, om, y, P.fps, P.fps, B, H, W, s));
    }
    return EMAVFI_OK;
}
// EMAVFI_AMP16: fp16 offset_conv on the fp16 rounding x16 of the fusion tensor, fp32 deform_conv2d on the fp32 tensor xF -> yF; y16 receives
// the fp16 rounding of yF - from the DCN's own epilogue where the fp32 LDS-window kernel serves the shape (round 5: was a conversion
// pass per block), else from launch_convert_cl
bool amp_dcn_writes_fp16(const Plan &P, int i) { return P.dcn32[i].pack3 == 3; }
int attention_block_amp(const Plan &P, int i, const void *packed, const void *x16, const float *xF, float *yF, bool want16, void *y16, float *om, int B,
                        int H, int W, hipStream_t s, Recorder &rec)
{
    const double px = (double)B * H * W, cf = P.mid + 3;
    const size_t npx = (size_t)B * H * W;
    double fl, by;
    conv_work(P, P.off[i], B, H, W, 4.0, fl, by);
    EMAVFI_STEP(rec, conv_name(P, P.off[i]) + " offset_conv", fl, by, run_conv(P, P.off[i], packed, x16, P.fps, H, W, om, 32, 0, 32, EPI_OM, B, s));
    // (want16, not the pointer, decides: the enumeration-only pass has no buffers and must list the launches the real pass runs)
    const bool both = amp_dcn_writes_fp16(P, i) && want16;
    EMAVFI_STEP(rec, both ? (P.x3 ? "deform<f32,ck=80,nf=3> dcn_v2 (exact fp32; writes fp32 + its f16 hi / lo halves)" : "deform<f32,ck=80,nf=3> dcn_v2 (fp32 under autocast; writes fp32 + its fp16 rounding)")
                          : "deform<f32,ck=80,nf=3> dcn_v2 (fp32 under autocast)",
                2.0 * 9.0 * cf * cf * px, px * (2.0 * cf * 4.0 + 27.0 * 4.0 + (both ? cf * 2.0 * (P.x3 ? 2 : 1) : 0.0)) + 9.0 * cf * cf * 4.0,
                run_deform(P, P.dcn32[i], packed, xF, P.fpad, om, yF, P.fpad, P.fpad, B, H, W, s, nullptr, nullptr, nullptr, 0, EMAVFI_F32, 0, 0,
                           both ? y16 : nullptr, P.fps * (P.x3 ? 2 : 1), nullptr, both && P.x3 ? P.fps : 0));
    if (!both && want16)
        EMAVFI_STEP(rec, "fusion_round", 0, px * cf * 6.0, launch_convert_cl(yF, y16, npx, P.fpad, P.fps * (P.x3 ? 2 : 1), 0, P.fpad, 0, s, P.x3 ? P.fps : 0));
    return EMAVFI_OK;
}

// ---- context_encoding (ema_vfi.py:79-86, called at :120) and reconstruction (ema_vfi.py:102-107, called at :144-146) as the forward
// runs them; shared by forward_impl and the stage-level entries emavfi_context / emavfi_reconstruct
int context_stage(const Plan &P, const void *packed, const void *feat_cl, const FwdBuffers &f, int B, int H, int W, hipStream_t s, Recorder &rec)
{
    const double e = P.esize;
    const int mid = P.mid, dtype = P.dtype;
    double fl, by;
    conv_work(P, P.c0, B, H, W, e, fl, by);
    EMAVFI_STEP(rec, conv_name(P, P.c0) + " context_encoding.0", fl, by,
                run_conv(P, P.c0, packed, feat_cl, P.fps, H, W, f.c1, f.p2, 0, f.p2, EPI_RELU, B, s, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
                         P.feat16 ? 1 : 0));
    conv_work(P, P.c1, B, f.H2, f.W2, e, fl, by);
    EMAVFI_STEP(rec, conv_name(P, P.c1) + " context_encoding.1", fl, by,
                run_conv(P, P.c1, packed, f.c1, f.p2, f.H2, f.W2, f.c2, f.p4, 0, f.p4, EPI_RELU, B, s));
    conv_work(P, P.c2, B, f.H4, f.W4, e, fl, by);
    // conv_wreg.inl at the reference width: the layer's only reader is the pool, so the kernel writes per-tile channel sums instead of the
    // tensor (no 236 MB store + read at B = 8 x 720p); avg_pool_partial then adds tiles instead of pixels.  EMAVFI_CONV_POOLFUSE=0: A/B
    const bool poolfuse = P.c2.ring == 4 && f.p4 == 256 && !(emavfi_switches() & SW_NO_POOLFUSE);
    if (poolfuse) {
        EMAVFI_STEP(rec, conv_name(P, P.c2) + " context_encoding.2 + pool (tile sums)", fl, by - (double)B * f.H4 * f.W4 * 4 * mid * e,
                    run_conv(P, P.c2, packed, f.c2, f.p4, f.H4, f.W4, nullptr, f.p4, 0, f.p4, EPI_RELU, B, s, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr,
                             f.tpart));
        EMAVFI_STEP(rec, "avg_pool_partial", 0, (double)B * f.ntiles * 4 * mid * 4,
                    launch_pool_partial(f.tpart, f.part, B, f.ntiles, f.p4, f.p4, f.nparts2, EMAVFI_F32, s));
    } else {
        EMAVFI_STEP(rec, conv_name(P, P.c2) + " context_encoding.2", fl, by,
                    run_conv(P, P.c2, packed, f.c2, f.p4, f.H4, f.W4, f.c3, f.p4, 0, f.p4, EPI_RELU, B, s));
        const int hw = P.x3 ? 2 : 1;   // EMAVFI_F32X3: the hi and the lo halves are pooled as 2 x p4 channels and added in the fold
        EMAVFI_STEP(rec, "avg_pool_partial", 0, (double)B * f.H4 * f.W4 * 4 * mid * e * hw,
                    launch_pool_partial(f.c3, f.part, B, f.H4 * f.W4, hw * f.p4, hw * f.p4, f.nparts, dtype, s));
    }
    const int np = poolfuse ? f.nparts2 : f.nparts;
    EMAVFI_STEP(rec, "context_linear_fold", 0, (double)B * np * 4 * mid * 4,
                launch_ctx_finish(f.part, packed ? (const float *)((const char *)packed + P.ctx_off) : nullptr, f.ctx, f.table, B, mid,
                                  P.x3 ? 2 * f.p4 : f.p4, np, f.H4 * f.W4, P.m0.coutpad, P.amp ? 1 : 0, s, P.x3 ? f.p4 : 0));
    return EMAVFI_OK;
}

int reconstruction_stage(const Plan &P, const void *packed, const void *x, const FwdBuffers &f, float *out, int B, int H, int W, hipStream_t s, Recorder &rec)
{
    const double e = P.esize;
    const int C = P.in_ch;
    const unsigned sw = emavfi_switches();
    double fl, by;
    conv_work(P, P.r0, B, H, W, e, fl, by);
    EMAVFI_STEP(rec, conv_name(P, P.r0) + " reconstruction.0", fl, by,
                run_conv(P, P.r0, packed, x, P.fps, H, W, f.fA, P.p_mid, 0, P.p_mid, EPI_RELU, B, s));
    // reconstruction.1 + .2 in one launch (conv_ring_tail.inl): .1's rows never leave the LDS.  EMAVFI_CONV_TAILFUSE=0: two
    if (P.r1.mfma16 && P.r1.ck == 64 && P.r1.nf == 1 && P.r1.cout == 32 && P.r1.ring == 0 && P.r2.mfma16 && P.r2.ck == 32 && P.r2.nf == 1 && P.r2.cout <= 3 &&
        !(sw & SW_NO_TAILFUSE)) {
        double fl2, by2;
        conv_work(P, P.r1, B, H, W, e, fl, by);
        conv_work(P, P.r2, B, H, W, 4.0, fl2, by2);
        const double mid_bytes = (double)B * H * W * P.r1.cout * e;
        EMAVFI_STEP(rec, "conv3x3+tail<" + std::string(dtype_name(P.dtype)) + ",64->32->" + std::to_string(P.r2.cout) + "> reconstruction.1+.2(tanh)",
                    fl + fl2, by + by2 - 2 * mid_bytes,
                    run_conv(P, P.r1, packed, f.fA, P.p_mid, H, W, nullptr, 0, 0, 0, EPI_RELU, B, s, nullptr, out, C, nullptr, &P.r2, nullptr, EPI_PLANAR_TANH01));
    } else {
        conv_work(P, P.r1, B, H, W, e, fl, by);
        EMAVFI_STEP(rec, conv_name(P, P.r1) + " reconstruction.1", fl, by,
                    run_conv(P, P.r1, packed, f.fA, P.p_mid, H, W, f.fB, f.p_half, 0, f.p_half, EPI_RELU, B, s));
        conv_work(P, P.r2, B, H, W, 4.0, fl, by);
        EMAVFI_STEP(rec, conv_name(P, P.r2) + " reconstruction.2(tanh)", fl, by,
                    run_conv(P, P.r2, packed, f.fB, f.p_half, H, W, nullptr, 0, 0, 0, EPI_PLANAR_TANH01, B, s, nullptr, out, C));
    }
    return EMAVFI_OK;
}

BlobHeader expected_header(const Plan &P, int requested_dtype)
{
    BlobHeader h{};
    memcpy(h.magic, "EMAVFIPK", 8);
    h.version = EMAVFI_VERSION; h.header_bytes = kBlobHeaderBytes;
    h.in_ch = (uint32_t)P.in_ch; h.mid = (uint32_t)P.mid; h.nb = (uint32_t)P.nb; h.dtype = (uint32_t)requested_dtype;
    h.layout_tag = layout_tag_of(process_layout_env());
    h.total_bytes = P.total;
    return h;
}

int forward_impl(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes, const float *frame1,
                 const float *frame2, float *out, void *workspace, size_t workspace_bytes, int B, int H, int W, int dtype,
                 float *const *taps, void *stream, Recorder &rec)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s", P.why);
    const int requested_dtype = dtype;
    dtype = P.dtype;  // the kernel storage type from here on (EMAVFI_AMP16 -> EMAVFI_F16, with P.amp set)
    if (B < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "forward: B, H, W must be >= 1 (got %d, %d, %d)", B, H, W);
    if ((size_t)B * H * W >= ((size_t)1 << 31)) return fail(EMAVFI_E_ARG, "forward: B*H*W must be < 2^31");
    // the kernels' byte offsets inside one sample are 32-bit: guard with the WIDEST element any kernel of the plan reads
    // (EMAVFI_AMP16 runs the fp32 deformable kernel on an fp32 fusion tensor: 4-byte elements beside P.esize = 2)
    const size_t widest = P.wide() ? sizeof(float) : (size_t)P.esize;
    if ((size_t)H * W * P.fpad * widest >= ((size_t)1 << 32)) return fail(EMAVFI_E_ARG, "forward: one sample's activation plane must be < 4 GiB");
    if ((size_t)H * W >= ((size_t)1 << 24)) return fail(EMAVFI_E_ARG, "forward: H*W must be < 2^24 (24-bit pixel index arithmetic in the gather kernels)");
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    FwdBuffers f;
    carve_forward(P, ws, f, B, H, W);
    if (!rec.dry) {
        if (!packed || !frame1 || !frame2 || !out || !workspace) return fail(EMAVFI_E_ARG, "forward: null pointer");
        if (!aligned16(packed) || !aligned16(workspace) || !aligned16(frame1) || !aligned16(frame2) || !aligned16(out))
            return fail(EMAVFI_E_ARG, "forward: pointers must be 16-byte aligned");
        if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "forward: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
        // every kernel reads the blob at offsets of THIS plan: a shorter buffer is another model's, another dtype's or a truncated one
        if (packed_bytes < P.total)
            return fail(EMAVFI_E_ARG, "forward: packed blob has %zu bytes, EMA_VFI(%d, %d, %d) in this dtype needs %zu (wrong model / dtype / version?)",
                        packed_bytes, in_channels, mid_channels, num_blocks, P.total);
    }
    hipStream_t s = (hipStream_t)stream;
    const unsigned sw = emavfi_switches();
    // the one-launch packs count what left their window (emavfi_forward_census reads it back): 8 KiB, zeroed per forward
    if (!rec.dry && hipMemsetAsync(f.census, 0, kCensusBytes, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "forward: census memset failed");
    // the header of the blob is compared ON THE DEVICE with what this call expects (the forward's last launch, blob_guard_kernel): a
    // blob of another version / model / dtype / layout-switch setting yields an all-NaN frame instead of plausible garbage.
    // emavfi_packed_check() is the (synchronising) entry that returns a code for it.
    BlobGuard guard{rec.dry ? nullptr : (const BlobHeader *)packed, expected_header(P, requested_dtype)};
    const int mid = P.mid, C = P.in_ch;
    const double px = (double)B * H * W, e = P.esize;
    const size_t npx = (size_t)B * H * W;
    double fl, by;

    // bf16 model: `feat` and the warped tail as f16 (Plan::feat16) when the first pack is the one-launch kernel that wants them so
    const bool feat16 = P.feat16;
    const int feat_dtype = feat16 ? (int)EMAVFI_F16 : dtype;
    // --- feature extraction: cat + conv + ReLU, then num_blocks x (conv + ReLU)  (ema_vfi.py:112-116)
    conv_work(P, P.conv1, B, H, W, e, fl, by);
    int first_blk = 0;
    void *cur = f.fA, *nxt = f.fB;
    {
        // EMAVFI_CONV_FIRST=0 / EMAVFI_CONV_FIRSTRING=0 (the blob carries both layouts; the switch word is latched once per process and
        // flipped by the parity tests through emavfi_debug_switches)
        if (P.conv1.first6 && !(sw & SW_NO_CONV_FIRST)) {
            // cat(frame1, frame2) + conv + ReLU in one launch, straight from the NCHW fp32 frames (ema_vfi.py:112-113)
            // (the enumeration-only pass has no blob: no arithmetic on its null pointer - found by the UBSan build, tests/host/host_check.cpp)
            const char *blob = (const char *)packed;
            FirstParams fp{frame1, frame2, f.fA, blob ? blob + P.conv1.w_off + P.conv1.w_bytes : nullptr,
                           blob ? (const float *)(blob + P.conv1.b_off) : nullptr, P.p_mid, H, W, B, 1};
            if (P.nb >= 1 && P.blk[0].ring == 2 && !(sw & SW_NO_FIRSTRING)) {
                // ... and feat_ext_blocks.conv_block_0 + ReLU behind it in the SAME launch (conv_ring_first.inl): feat_ext_conv1's
                // tensor exists only as an LDS ring.  EMAVFI_CONV_FIRSTRING=0 (read per call): two launches
                double fl2, by2;
                const bool last = P.nb == 1;
                void *dst = last ? f.fu0 : f.fB;
                conv_work(P, P.blk[0], B, H, W, e, fl2, by2);
                EMAVFI_STEP(rec, std::string("conv_first+conv3x3<") + dtype_name(dtype) + ",6->64->64> cat+feat_ext_conv1+conv_block_0", fl + fl2,
                            px * (8.0 * C + mid * e) + 9.0 * 2 * C * mid * e + 9.0 * mid * mid * e,
                            run_conv(P, P.blk[0], packed, nullptr, P.p_mid, H, W, dst, last ? P.fps : P.p_mid, 0, P.p_mid, EPI_RELU, B, s, nullptr, nullptr, 0, nullptr, nullptr, &fp));
                first_blk = 1;
                cur = f.fB; nxt = f.fA;
            } else {
                EMAVFI_STEP(rec, std::string("conv_first<") + dtype_name(dtype) + ",6->64> cat+feat_ext_conv1", fl, px * (8.0 * C + mid * e) + 9.0 * 2 * C * mid * e,
                            dtype == EMAVFI_F16 ? launch_conv_first_f16(fp, s) : launch_conv_first_bf16(fp, s));
            }
        } else {
            EMAVFI_STEP(rec, "pack_input", 0, px * (8.0 * C + 2.0 * C * e * (P.x3 ? 2 : 1)),
                        P.x3 ? launch_pack_input_x3(frame1, frame2, f.in16, B, C, H, W, 16, s) : launch_pack_input(frame1, frame2, f.in16, B, C, H, W, 16, dtype, s));
            EMAVFI_STEP(rec, conv_name(P, P.conv1) + " feat_ext_conv1", fl, by,
                        run_conv(P, P.conv1, packed, f.in16, 16, H, W, f.fA, P.p_mid, 0, P.p_mid, EPI_RELU, B, s));
        }
    }
    for (int i = first_blk; i < P.nb; ++i) {
        // two consecutive conv_blocks as ONE launch (conv_ring2.inl): the tensor between them exists only as an LDS ring.
        // EMAVFI_CONV_RING2=0: one launch each
        const bool pair = i + 1 < P.nb && P.blk[i].ring == 2 && P.blk[i + 1].ring == 2 && !P.blk[i].f16_of_bf16 && !P.blk[i + 1].f16_of_bf16 && !(sw & SW_NO_RING2);
        const int j = pair ? i + 1 : i;   // the layer whose output leaves the launch
        const bool last = j == P.nb - 1;  // the last block writes feat straight into the fusion buffer
        void *dst = last ? f.fu0 : nxt;
        conv_work(P, P.blk[i], B, H, W, e, fl, by);
        if (pair) {
            double fl2, by2;
            conv_work(P, P.blk[j], B, H, W, e, fl2, by2);
            const double mid_bytes = (double)B * H * W * mid * e;   // the intermediate tensor is neither written nor read
            EMAVFI_STEP(rec, "conv3x3+conv3x3<" + std::string(dtype_name(dtype)) + ",64->64->64> feat_ext_blocks." + std::to_string(i) + "+" + std::to_string(j),
                        fl + fl2, by + by2 - 2 * mid_bytes,
                        run_conv(P, P.blk[i], packed, cur, P.p_mid, H, W, dst, last ? P.fps : P.p_mid, 0, P.p_mid, EPI_RELU, B, s, nullptr, nullptr, 0, nullptr,
                                 nullptr, nullptr, 0, last && feat16 ? 1 : 0, &P.blk[j]));
            i = j;
        } else {
            // (EMAVFI_F32X3: the last block also writes `feat` in fp32 into the fusion tensor the exact DCN reads)
            EMAVFI_STEP(rec, conv_name(P, P.blk[i]) + " feat_ext_blocks", fl, by + (last && P.x3 ? px * mid * 4.0 : 0.0),
                        run_conv(P, P.blk[i], packed, cur, P.p_mid, H, W, dst, last ? P.fps : P.p_mid, 0, P.p_mid, EPI_RELU, B, s, nullptr, nullptr, 0, nullptr,
                                 nullptr, nullptr, 0, last && feat16 ? 1 : 0, nullptr, nullptr, last && P.x3 ? f.fuF0 : nullptr, P.fpad));
        }
        if (!last) { void *t = cur; cur = nxt; nxt = t; }
    }
    if (!rec.dry && taps && taps[0])
        EMAVFI_TRY(P.x3 ? launch_cl_to_nchw_x3(f.fu0, taps[0], B, mid, H, W, P.fps, 0, s) : launch_cl_to_nchw(f.fu0, taps[0], B, mid, H, W, P.fps, 0, feat_dtype, s), "tap feat");

    // --- context encoding (ema_vfi.py:120): two stride-2 convs, one conv, global mean, linear
    if (const int rc = context_stage(P, packed, f.fu0, f, B, H, W, s, rec); rc != EMAVFI_OK) return rc;
    if (!rec.dry && taps && taps[1])
        if (hipMemcpyAsync(taps[1], f.ctx, (size_t)B * mid * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(EMAVFI_E_LAUNCH, "tap ctx copy failed");

    // --- motion estimation (ema_vfi.py:124-126); the broadcast-context concat is a per-border-class bias
    conv_work(P, P.m0, B, H, W, e, fl, by);
    EMAVFI_STEP(rec, conv_name(P, P.m0) + " motion_estimation.0(ctx folded)", fl, by,
                run_conv(P, P.m0, packed, f.fu0, P.fps, H, W, f.fA, P.p_mid, 0, P.p_mid, EPI_RELU, B, s, f.table, nullptr, 0, nullptr, nullptr, nullptr, 0,
                         feat16 ? 1 : 0));
    // motion_estimation.1 + .2 in one launch (conv_ring.inl, HEAD): .1's rows never leave the LDS.  EMAVFI_CONV_HEAD=0: two launches
    if (P.m1.ring == 2 && P.m2.mfma16 && P.m2.ck == 64 && P.m2.nf == 1 && P.m2.cout <= 2 && !(sw & SW_NO_HEAD)) {
        double fl2, by2;
        conv_work(P, P.m1, B, H, W, e, fl, by);
        conv_work(P, P.m2, B, H, W, 4.0, fl2, by2);
        // bytes: .1's input and weights, .2's weights and output (the intermediate tensor is neither written nor read)
        const double mid_bytes = (double)B * H * W * mid * e;
        EMAVFI_STEP(rec, "conv3x3+head<" + std::string(dtype_name(P.dtype)) + ",64->64->" + std::to_string(P.m2.cout) + "> motion_estimation.1+.2(flow)",
                    fl + fl2, by + by2 - 2 * mid_bytes,
                    run_conv(P, P.m1, packed, f.fA, P.p_mid, H, W, nullptr, 0, 0, 0, EPI_RELU, B, s, nullptr, f.flow, 2, nullptr, &P.m2));
    } else {
        conv_work(P, P.m1, B, H, W, e, fl, by);
        EMAVFI_STEP(rec, conv_name(P, P.m1) + " motion_estimation.1", fl, by,
                    run_conv(P, P.m1, packed, f.fA, P.p_mid, H, W, f.fB, P.p_mid, 0, P.p_mid, EPI_RELU, B, s));
        conv_work(P, P.m2, B, H, W, 4.0, fl, by);
        EMAVFI_STEP(rec, conv_name(P, P.m2) + " motion_estimation.2(flow)", fl, by,
                    run_conv(P, P.m2, packed, f.fB, P.p_mid, H, W, nullptr, 0, 0, 0, EPI_PLANAR, B, s, nullptr, f.flow, 2));
    }
    if (!rec.dry && taps && taps[2])
        if (hipMemcpyAsync(taps[2], f.flow, npx * 2 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(EMAVFI_E_LAUNCH, "tap flow copy failed");

    void *x = f.fu0, *y = f.fu1;
    if (P.wide()) {
        // --- EMAVFI_F32X3 takes the same data flow without any rounding: the fp32 fusion tensor is what the exact fp32 DCN reads and
        // writes, its [hi | lo] f16 halves (22 bits) are what the split convolutions read.
        // --- EMAVFI_AMP16: the autocast op policy for ema_vfi.py:130-138.  grid_sample runs in fp32 on the fp16-valued
        // flow and cat(feat, warped) promotes to fp32, so the fusion tensor exists twice: fp32 (fuF: what the fp32
        // deform_conv2d reads and writes, never rounded between blocks) and its fp16 rounding (fu: what the fp16
        // offset_conv / reconstruction.0 read).
        float *xF = f.fuF0, *yF = f.fuF1;
        // (round 5: the warp writes the fp32 values AND their fp16 rounding - was the conversion pass fusion_round_warped)
        if (P.x3) {
            EMAVFI_STEP(rec, "warp_fused<f32>", 24.0 * px, px * (8.0 + 8.0 * C), launch_warp_fused(frame2, f.flow, xF, B, C, H, W, P.fpad, mid, EMAVFI_F32, s));
            EMAVFI_STEP(rec, "fusion_split_warped", 0, px * (P.fpad - mid) * 8.0, launch_convert_cl(xF, f.fu0, npx, P.fpad, 2 * P.fps, mid, P.fpad - mid, 0, s, P.fps));
            // (`feat`'s fp32 copy is already in xF: the last feature layer wrote it beside its f16 halves; nb == 0 has no such layer)
            if (P.nb == 0) EMAVFI_STEP(rec, "fusion_widen_feat", 0, px * mid * 8.0, launch_convert_cl(f.fu0, xF, npx, 2 * P.fps, P.fpad, 0, mid, 1, s, P.fps));
        } else {
            EMAVFI_STEP(rec, "warp_fused<f32>", 24.0 * px, px * (8.0 + 8.0 * C + 2.0 * C),
                        launch_warp_fused(frame2, f.flow, xF, B, C, H, W, P.fpad, mid, EMAVFI_F32, s, f.fu0, P.fps));
            EMAVFI_STEP(rec, "fusion_widen_feat", 0, px * mid * 6.0, launch_convert_cl(f.fu0, xF, npx, P.fps, P.fpad, 0, mid, 1, s));
        }
        if (!rec.dry && taps && taps[3])
            EMAVFI_TRY(launch_cl_to_nchw(xF, taps[3], B, C, H, W, P.fpad, mid, EMAVFI_F32, s), "tap warped");
        EMAVFI_STAGE_EVENT(rec, 0);
        for (int i = 0; i < P.nb; ++i) {
            if (const int rc = attention_block_amp(P, i, packed, x, xF, yF, true, y, f.om, B, H, W, s, rec); rc != EMAVFI_OK) return rc;
            if (!rec.dry && taps && taps[5 + i])
                EMAVFI_TRY(launch_cl_to_nchw(yF, taps[5 + i], B, mid + 3, H, W, P.fpad, 0, EMAVFI_F32, s), "tap fused");
            void *t = x; x = y; y = t;
            float *tf = xF; xF = yF; yF = tf;
        }
    } else {
        // --- warp frame2 by the flow into channels [mid, fpad) of the fusion buffer (ema_vfi.py:130,134).
        // When the first pack runs as the one-launch LDS kernel, those 16 channels go to a compact buffer of their own
        // (kTailPs = 4 channels = 8 bytes per pixel, in the packed-input buffer, free since conv1) and the kernel's window DMA picks
        // them up from there: contiguous 16-byte pixels instead of 6 useful bytes scattered into every 160-byte fusion pixel.
        const bool split_tail = P.nb > 0 && pack_fuses(P, 0) && P.fpad - mid == 16;
        EMAVFI_STEP(rec, std::string("warp_fused<") + dtype_name(dtype) + ">", 24.0 * px, px * (8.0 + 4.0 * C + C * e),
                    split_tail ? launch_warp_fused(frame2, f.flow, f.in16, B, C, H, W, kTailPs, 0, feat_dtype, s)
                               : launch_warp_fused(frame2, f.flow, f.fu0, B, C, H, W, P.fps, mid, feat_dtype, s));
        if (!rec.dry && taps && taps[3])
            EMAVFI_TRY(split_tail ? launch_cl_to_nchw(f.in16, taps[3], B, C, H, W, kTailPs, 0, feat_dtype, s)
                                  : launch_cl_to_nchw(f.fu0, taps[3], B, C, H, W, P.fps, mid, feat_dtype, s), "tap warped");
        EMAVFI_STAGE_EVENT(rec, 0);

        // --- multi-attention fusion: num_blocks x ModulatedDeformConvPack, no activation (ema_vfi.py:136-138)
        for (int i = 0; i < P.nb; ++i) {
            if (const int rc = attention_block(P, i, packed, x, y, f.om, i == 0 && split_tail ? f.in16 : nullptr,
                                               (i == 0 ? feat16 : pack_f16_link(P, i - 1)) ? 1 : 0, pack_f16_link(P, i) ? 1 : 0, B, H, W, s, rec,
                                               rec.dry ? nullptr : f.census + (size_t)i * DEFORM_CENSUS_SLOTS * 4);
                rc != EMAVFI_OK)
                return rc;
            if (!rec.dry && taps && taps[5 + i])
                EMAVFI_TRY(launch_cl_to_nchw(y, taps[5 + i], B, mid + 3, H, W, P.fps, 0, pack_f16_link(P, i) ? (int)EMAVFI_F16 : dtype, s), "tap fused");
            void *t = x; x = y; y = t;
        }
    }

    EMAVFI_STAGE_EVENT(rec, 1);
    // --- reconstruction (ema_vfi.py:144-146)
    if (const int rc = reconstruction_stage(P, packed, x, f, out, B, H, W, s, rec); rc != EMAVFI_OK) return rc;
    EMAVFI_STEP(rec, "blob_guard", 0, 64.0, launch_blob_guard(guard, out, npx * (size_t)C, s));
    EMAVFI_STAGE_EVENT(rec, 2);
    return EMAVFI_OK;
}

}  // namespace

unsigned emavfi_switches()
{
    init_switches();
    return g_switches.load(std::memory_order_relaxed);
}

extern "C" {

#if defined(EMAVFI_DEFORM_STAMPS) && EMAVFI_DEFORM_STAMPS
// rows of {prologue, offset_conv, geometry, steps, epilogue, total, valid, start time} per sampled wave
int emavfi_debug_deform_stamps(unsigned long long *out, int rows, int reset)
{
    unsigned long long *buf = debug_stamp_buffer();
    if (!buf || rows > DEFORM_STAMP_ROWS) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (out && hipMemcpy(out, buf, (size_t)rows * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    if (reset && hipMemset(buf, 0, (size_t)DEFORM_STAMP_ROWS * 8 * sizeof(unsigned long long)) != hipSuccess) return -4;
    return 0;
}
#endif

int emavfi_version(void) { return EMAVFI_VERSION; }
const char *emavfi_last_error(void) { return g_err; }
int emavfi_param_count(int num_blocks) { return 2 * (11 + 3 * num_blocks); }

int emavfi_supported(int in_channels, int mid_channels, int num_blocks, int dtype)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s", P.why);
    return EMAVFI_OK;
}

size_t emavfi_packed_bytes(int in_channels, int mid_channels, int num_blocks, int dtype)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) { fail(EMAVFI_E_UNSUPPORTED, "%s", P.why); return 0; }
    return P.total;
}

int emavfi_pack_weights(int in_channels, int mid_channels, int num_blocks, const void *const *params, int n_params,
                        void *packed, size_t packed_bytes, int dtype, void *stream)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s", P.why);
    if (!params || !packed) return fail(EMAVFI_E_ARG, "pack_weights: null pointer");
    if (n_params != emavfi_param_count(num_blocks))
        return fail(EMAVFI_E_ARG, "pack_weights: expected %d tensors, got %d", emavfi_param_count(num_blocks), n_params);
    for (int i = 0; i < n_params; ++i)
        if (!params[i]) return fail(EMAVFI_E_ARG, "pack_weights: params[%d] is null", i);
    if (packed_bytes < P.total) return fail(EMAVFI_E_WORKSPACE, "pack_weights: need %zu bytes, got %zu", P.total, packed_bytes);
    if (!aligned16(packed)) return fail(EMAVFI_E_ARG, "pack_weights: packed buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const BlobHeader hdr = expected_header(P, dtype);
    dtype = P.dtype;            // kernel storage type (EMAVFI_AMP16 -> EMAVFI_F16)
    const bool b16 = P.amp;     // autocast casts a convolution's bias to fp16 as well (EMAVFI_F32X3 keeps every bias in fp32)
    // (the gaps between the 256-byte aligned layers are part of the checksummed payload: zero, not whatever the buffer held)
    if (hipMemsetAsync(packed, 0, P.total, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "pack: blob memset failed");
    EMAVFI_TRY(pack_layer(P.conv1, params, packed, dtype, s, b16), "pack conv1");
    for (int i = 0; i < P.nb; ++i) EMAVFI_TRY(pack_layer(P.blk[i], params, packed, dtype, s, b16), "pack feat block");
    EMAVFI_TRY(pack_layer(P.c0, params, packed, dtype, s, b16), "pack ctx0");
    EMAVFI_TRY(pack_layer(P.c1, params, packed, dtype, s, b16), "pack ctx1");
    EMAVFI_TRY(pack_layer(P.c2, params, packed, dtype, s, b16), "pack ctx2");
    EMAVFI_TRY(pack_layer(P.m0, params, packed, dtype, s, b16), "pack motion0");
    EMAVFI_TRY(pack_layer(P.m1, params, packed, dtype, s, b16), "pack motion1");
    EMAVFI_TRY(pack_layer(P.m2, params, packed, dtype, s, b16), "pack motion2");
    for (int i = 0; i < P.nb; ++i) {
        EMAVFI_TRY(pack_layer(P.off[i], params, packed, dtype, s, b16), "pack offset_conv");
        if (P.has_offh) EMAVFI_TRY(pack_layer(P.offh[i], params, packed, dtype, s), "pack offset_conv (f16 fragments)");
        EMAVFI_TRY(pack_layer(P.dcn[i], params, packed, dtype, s), "pack dcn_v2");
        if (P.wide()) EMAVFI_TRY(pack_layer(P.dcn32[i], params, packed, EMAVFI_F32, s), "pack dcn_v2 (fp32 master weights)");
    }
    EMAVFI_TRY(pack_layer(P.r0, params, packed, dtype, s, b16), "pack recon0");
    EMAVFI_TRY(pack_layer(P.r1, params, packed, dtype, s, b16), "pack recon1");
    EMAVFI_TRY(pack_layer(P.r2, params, packed, dtype, s, b16), "pack recon2");
    EMAVFI_TRY(launch_pack_ctx((const float *)params[P.lin_param], (const float *)params[P.lin_param + 1],
                               (const float *)params[P.m0.param], (const float *)params[P.m0.param + 1],
                               (float *)((char *)packed + P.ctx_off), P.mid, P.amp ? 1 : 0, s),
               "pack context");
    EMAVFI_TRY(launch_blob_seal(packed, hdr, s), "pack: blob header");
    return EMAVFI_OK;
}

int emavfi_layout_tag(void) { return (int)layout_tag_of(process_layout_env()); }

int emavfi_debug_switches(int and_mask, int or_mask)
{
    init_switches();
    unsigned old = g_switches.load(std::memory_order_relaxed), want;
    do want = (old & (unsigned)and_mask) | (unsigned)or_mask;
    while (!g_switches.compare_exchange_weak(old, want, std::memory_order_relaxed));
    return (int)old;
}

int emavfi_packed_check(int in_channels, int mid_channels, int num_blocks, int dtype, const void *packed, size_t packed_bytes)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s", P.why);
    if (!packed) return fail(EMAVFI_E_ARG, "packed_check: null pointer");
    if (packed_bytes < P.total) return fail(EMAVFI_E_ARG, "packed_check: blob has %zu bytes, this model / dtype needs %zu", packed_bytes, P.total);
    // device memory is copied to the host (this entry synchronises: it is for the moment a blob arrives - from a file, another rank,
    // another process -, not for the per-frame path); anything HIP does not know as device memory is read in place
    std::vector<unsigned char> host;
    const unsigned char *bytes = (const unsigned char *)packed;
    hipPointerAttribute_t attr{};
    const bool on_device = hipPointerGetAttributes(&attr, packed) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    (void)hipGetLastError();   // (a plain host pointer makes hipPointerGetAttributes fail: not an error of this call)
    if (on_device) {
        // The blob's producer (the pack kernels on the caller's stream, an RCCL broadcast, a cache upload) may have run on ANY stream of
        // the blob's device - also a non-blocking one, which the legacy null stream of a plain hipMemcpy does not wait for - and that
        // device need not be the current one.  So: make the blob's device current, wait for ALL of its streams, then copy.
        int cur_dev = 0;
        if (hipGetDevice(&cur_dev) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "packed_check: hipGetDevice failed");
        const bool other = attr.device != cur_dev;
        if (other && hipSetDevice(attr.device) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "packed_check: cannot select the blob's device %d", attr.device);
        host.resize(P.total);
        const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(host.data(), packed, P.total, hipMemcpyDeviceToHost) == hipSuccess;
        if (other) (void)hipSetDevice(cur_dev);
        if (!ok) { (void)hipGetLastError(); return fail(EMAVFI_E_LAUNCH, "packed_check: device synchronise / device-to-host copy failed"); }
        bytes = host.data();
    }
    BlobHeader got;
    memcpy(&got, bytes, sizeof got);
    const BlobHeader want = expected_header(P, dtype);
    if (memcmp(got.magic, want.magic, 8) != 0) return fail(EMAVFI_E_ARG, "packed_check: no EMAVFIPK header (a blob of a library older than 0.4.0, or not a blob)");
    if (got.version != want.version) return fail(EMAVFI_E_ARG, "packed_check: blob packed by library version %u, this is %u: re-pack", got.version, want.version);
    if (got.in_ch != want.in_ch || got.mid != want.mid || got.nb != want.nb)
        return fail(EMAVFI_E_ARG, "packed_check: blob is for EMA_VFI(%u, %u, %u), not (%d, %d, %d)", got.in_ch, got.mid, got.nb, in_channels, mid_channels, num_blocks);
    if (got.dtype != want.dtype) return fail(EMAVFI_E_ARG, "packed_check: blob was packed for dtype %u, asked for %d", got.dtype, dtype);
    if (got.layout_tag != want.layout_tag)
        return fail(EMAVFI_E_ARG, "packed_check: blob was packed under layout switches 0x%x, this process runs 0x%x (EMAVFI_CONV_MFMA16 / _RING / _WREG / _S2RING / EMAVFI_PACK_F16_CHAIN / EMAVFI_NO_FUSED_OFFSET)",
                    got.layout_tag, want.layout_tag);
    if (got.header_bytes != want.header_bytes || got.total_bytes != want.total_bytes)
        return fail(EMAVFI_E_ARG, "packed_check: blob size fields (%u, %llu) do not match this build's layout (%u, %llu)", got.header_bytes,
                    (unsigned long long)got.total_bytes, want.header_bytes, (unsigned long long)want.total_bytes);
    const uint64_t sum = blob_checksum_host(bytes + kBlobHeaderBytes, P.total - kBlobHeaderBytes);
    if (sum != got.checksum) return fail(EMAVFI_E_ARG, "packed_check: payload checksum %016llx, header says %016llx (corrupted blob)", (unsigned long long)sum, (unsigned long long)got.checksum);
    return EMAVFI_OK;
}

size_t emavfi_workspace_bytes(int in_channels, int mid_channels, int num_blocks, int B, int H, int W, int dtype)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) { fail(EMAVFI_E_UNSUPPORTED, "%s", P.why); return 0; }
    if (B < 1 || H < 1 || W < 1) { fail(EMAVFI_E_ARG, "workspace_bytes: B, H, W must be >= 1"); return 0; }
    Workspace ws{nullptr, 0, 0};
    FwdBuffers f;
    carve_forward(P, ws, f, B, H, W);
    return ws.used;
}

int emavfi_forward(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes, const float *frame1,
                   const float *frame2, float *out, void *workspace, size_t workspace_bytes, int B, int H, int W, int dtype,
                   float *const *taps, void *stream)
{
    Recorder rec;
    return forward_impl(in_channels, mid_channels, num_blocks, packed, packed_bytes, frame1, frame2, out, workspace, workspace_bytes, B, H, W,
                        dtype, taps, stream, rec);
}

int emavfi_forward_profiled(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes, const float *frame1,
                            const float *frame2, float *out, void *workspace, size_t workspace_bytes, int B, int H, int W,
                            int dtype, void *const *events, int n_events, void *stream)
{
    if (!events || n_events < 2) return fail(EMAVFI_E_ARG, "forward_profiled: events array required");
    Recorder rec;
    rec.events = events; rec.n_events = n_events;
    return forward_impl(in_channels, mid_channels, num_blocks, packed, packed_bytes, frame1, frame2, out, workspace, workspace_bytes, B, H, W,
                        dtype, nullptr, stream, rec);
}

int emavfi_forward_staged(int in_channels, int mid_channels, int num_blocks, const void *packed, size_t packed_bytes, const float *frame1,
                          const float *frame2, float *out, void *workspace, size_t workspace_bytes, int B, int H, int W, int dtype,
                          void *const *stage_events, void *const *events, int n_events, void *stream)
{
    Recorder rec;
    rec.stage_events = stage_events;
    if (events) { rec.events = events; rec.n_events = n_events; }
    return forward_impl(in_channels, mid_channels, num_blocks, packed, packed_bytes, frame1, frame2, out, workspace, workspace_bytes, B, H, W,
                        dtype, nullptr, stream, rec);
}

int emavfi_forward_launches(int in_channels, int mid_channels, int num_blocks, int B, int H, int W, int dtype,
                            char *names, size_t names_bytes, double *flops, double *bytes, int capacity)
{
    std::vector<LaunchRec> recs;
    Recorder rec;
    rec.recs = &recs; rec.dry = true;
    const int rc = forward_impl(in_channels, mid_channels, num_blocks, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, B, H, W,
                                dtype, nullptr, nullptr, rec);
    if (rc != EMAVFI_OK) return rc;
    const int n = (int)recs.size();
    if (capacity <= 0) return n;
    if (capacity < n) return fail(EMAVFI_E_ARG, "forward_launches: capacity %d < %d launches", capacity, n);
    size_t used = 0;
    for (int i = 0; i < n; ++i) {
        if (flops) flops[i] = recs[i].flops;
        if (bytes) bytes[i] = recs[i].bytes;
        if (names) {
            if (used + recs[i].name.size() + 2 > names_bytes) return fail(EMAVFI_E_ARG, "forward_launches: names buffer too small");
            memcpy(names + used, recs[i].name.c_str(), recs[i].name.size());
            used += recs[i].name.size();
            names[used++] = '\n';
            names[used] = 0;
        }
    }
    return n;
}

int emavfi_warp(const float *frame2, const float *flow, float *out, int B, int C, int H, int W, void *stream)
{
    if (!frame2 || !flow || !out) return fail(EMAVFI_E_ARG, "warp: null pointer");
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "warp: B, C, H, W must be >= 1");
    if ((size_t)H * W >= ((size_t)1 << 31)) return fail(EMAVFI_E_ARG, "warp: H*W too large");
    if (!aligned16(frame2) || !aligned16(flow) || !aligned16(out)) return fail(EMAVFI_E_ARG, "warp: pointers must be 16-byte aligned");
    EMAVFI_TRY(launch_warp_nchw(frame2, flow, out, B, C, H, W, (hipStream_t)stream), "warp");
    return EMAVFI_OK;
}

int emavfi_preprocess_u8(const unsigned char *frames_hwc, float *out_nchw, int B, int H, int W, int C, const float *mean,
                         const float *std, void *stream)
{
    if (!frames_hwc || !out_nchw || !mean || !std) return fail(EMAVFI_E_ARG, "preprocess_u8: null pointer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || C > 4) return fail(EMAVFI_E_ARG, "preprocess_u8: bad shape (C must be 1..4)");
    for (int c = 0; c < C; ++c)
        if (!(std[c] != 0.0f)) return fail(EMAVFI_E_ARG, "preprocess_u8: std[%d] must be non-zero", c);
    EMAVFI_TRY(launch_preprocess_u8(frames_hwc, out_nchw, B, H, W, C, mean, std, (hipStream_t)stream), "preprocess_u8");
    return EMAVFI_OK;
}

int emavfi_postprocess_u8(const float *frames_nchw, unsigned char *out_hwc, int B, int H, int W, int C, const double *mean,
                          const double *std, int denormalize, void *stream)
{
    if (!frames_nchw || !out_hwc || !mean || !std) return fail(EMAVFI_E_ARG, "postprocess_u8: null pointer");
    if (B < 1 || H < 1 || W < 1 || C < 1 || C > 4) return fail(EMAVFI_E_ARG, "postprocess_u8: bad shape (C must be 1..4)");
    EMAVFI_TRY(launch_postprocess_u8(frames_nchw, out_hwc, B, H, W, C, mean, std, denormalize ? 1 : 0, (hipStream_t)stream),
               "postprocess_u8");
    return EMAVFI_OK;
}

// ---- stage-level entries (diagnostics / parity tests of single operators) ----
static bool single_conv_layer(Layer &L, int Cin, int Cout, int stride, int esize, bool x3 = false)
{
    L = mk(0, Cout, Cin, stride);
    return conv_geometry(L, esize, read_layout_env(), x3);   // stage-level entry: packs and runs inside one call
}

size_t emavfi_conv3x3_workspace_bytes(int B, int Cin, int Cout, int H, int W, int stride, int dtype)
{
    Layer L;
    const bool x3 = dtype == EMAVFI_F32X3;
    const int e = dtype == EMAVFI_F32 ? 4 : 2;
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2) || !single_conv_layer(L, Cin, Cout, stride, e, x3)) {
        fail(EMAVFI_E_UNSUPPORTED, "conv3x3: no kernel instantiation for Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
        return 0;
    }
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    Workspace ws{nullptr, 0, 0};
    ws.take((size_t)B * H * W * L.cin_pad * e * (x3 ? 2 : 1));
    ws.take(L.w_bytes);
    ws.take((size_t)L.coutpad * sizeof(float));
    ws.take((size_t)B * Ho * Wo * rup(Cout, 16) * e * (x3 ? 2 : 1));
    ws.take(256);
    return ws.used;
}

int emavfi_conv3x3(const float *x, const float *weight, const float *bias, float *y, int B, int Cin, int Cout, int H, int W,
                   int stride, int act, int dtype, void *workspace, size_t workspace_bytes, void *stream)
{
    if (dtype != EMAVFI_F32 && dtype != EMAVFI_BF16 && dtype != EMAVFI_F16 && dtype != EMAVFI_F32X3) return fail(EMAVFI_E_ARG, "conv3x3: bad dtype %d", dtype);
    if (!x || !weight || !y || !workspace) return fail(EMAVFI_E_ARG, "conv3x3: null pointer");
    const bool x3 = dtype == EMAVFI_F32X3;   // the three-term f16 split: [hi | lo] halves in, fp32-accurate result out
    if (x3) dtype = EMAVFI_F16;
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2)) return fail(EMAVFI_E_ARG, "conv3x3: bad shape");
    if (act < EMAVFI_ACT_NONE || act > EMAVFI_ACT_TANH01) return fail(EMAVFI_E_ARG, "conv3x3: bad activation %d", act);
    if (act == EMAVFI_ACT_TANH01 && Cout > 4) return fail(EMAVFI_E_UNSUPPORTED, "conv3x3: TANH01 epilogue needs Cout <= 4");
    Plan P{};
    P.dtype = dtype; P.esize = dtype == EMAVFI_F32 ? 4 : 2;
    Layer L;
    if (!single_conv_layer(L, Cin, Cout, stride, P.esize, x3))
        return fail(EMAVFI_E_UNSUPPORTED, "conv3x3: no kernel instantiation for Cin=%d Cout=%d stride=%d", Cin, Cout, stride);
    P.x3 = x3;
    const int hw = x3 ? 2 : 1;
    // the kernels' DMA source offsets inside one sample are 32-bit (conv3x3.inl, conv_dma_src)
    if ((size_t)H * W * L.cin_pad * P.esize >= ((size_t)1 << 32)) return fail(EMAVFI_E_ARG, "conv3x3: one sample's input plane must be < 4 GiB");
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, ops = rup(Cout, 16);
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    void *xcl = ws.take((size_t)B * H * W * L.cin_pad * P.esize * hw);
    void *wp = ws.take(L.w_bytes);
    float *bp = (float *)ws.take((size_t)L.coutpad * sizeof(float));
    void *ycl = ws.take((size_t)B * Ho * Wo * ops * P.esize * hw);
    void *zpage = ws.take(256);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "conv3x3: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    L.w_off = (char *)wp - (char *)workspace;
    L.b_off = (char *)bp - (char *)workspace;
    PackDesc d{L.cout, L.cin_raw, 0, L.cin_take, L.ck, L.nchunk, L.nf, L.npass, 0, 0};
    d.mfma16 = L.mfma16 ? 1 : 0;
    d.ring = L.ring;
    d.x3 = x3 ? 1 : 0;
    if (hipMemsetAsync(zpage, 0, 256, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "conv3x3: zero page memset failed");
    EMAVFI_TRY(launch_pack_conv(weight, bias, wp, bp, d, dtype, s), "conv3x3 pack");
    EMAVFI_TRY(x3 ? launch_nchw_to_cl_x3(x, xcl, B, Cin, H, W, L.cin_pad, s) : launch_nchw_to_cl(x, xcl, B, Cin, H, W, L.cin_pad, dtype, s), "conv3x3 layout in");
    if (act == EMAVFI_ACT_TANH01) {
        EMAVFI_TRY(run_conv(P, L, workspace, xcl, L.cin_pad, H, W, nullptr, 0, 0, 0, EPI_PLANAR_TANH01, B, s, nullptr, y, Cout, zpage), "conv3x3");
    } else {
        EMAVFI_TRY(run_conv(P, L, workspace, xcl, L.cin_pad, H, W, ycl, ops, 0, ops, act == EMAVFI_ACT_RELU ? EPI_RELU : EPI_NONE, B, s, nullptr, nullptr, 0, zpage), "conv3x3");
        EMAVFI_TRY(x3 ? launch_cl_to_nchw_x3(ycl, y, B, Cout, Ho, Wo, ops, 0, s) : launch_cl_to_nchw(ycl, y, B, Cout, Ho, Wo, ops, 0, dtype, s), "conv3x3 layout out");
    }
    return EMAVFI_OK;
}

size_t emavfi_deform_conv2d_workspace_bytes(int B, int C, int O, int H, int W, int dtype)
{
    const int e = dtype == EMAVFI_F32 ? 4 : 2;
    Layer L = mk(0, O, C);
    if (B < 1 || C < 1 || O < 1 || H < 1 || W < 1 || !deform_geometry(L, e)) {
        fail(EMAVFI_E_UNSUPPORTED, "deform_conv2d: no kernel instantiation for C=%d O=%d", C, O);
        return 0;
    }
    if ((size_t)H * W >= ((size_t)1 << 24) || (size_t)H * W * L.ck * e >= ((size_t)1 << 32)) {
        fail(EMAVFI_E_ARG, "deform_conv2d: H*W must be < 2^24 and one sample's input plane < 4 GiB");
        return 0;
    }
    Workspace ws{nullptr, 0, 0};
    const size_t px = (size_t)B * H * W;
    ws.take(px * L.ck * e); ws.take(px * 32 * sizeof(float)); ws.take(L.w_bytes);
    ws.take((size_t)L.coutpad * sizeof(float)); ws.take(px * rup(O, 16) * e); ws.take(256);
    return ws.used;
}

int emavfi_deform_conv2d(const float *x, const float *offset, const float *mask, const float *weight, const float *bias,
                         float *y, int B, int C, int O, int H, int W, int dtype, void *workspace, size_t workspace_bytes,
                         void *stream)
{
    if (dtype != EMAVFI_F32 && dtype != EMAVFI_BF16 && dtype != EMAVFI_F16) return fail(EMAVFI_E_ARG, "deform_conv2d: bad dtype %d", dtype);
    if (!x || !offset || !mask || !weight || !y || !workspace) return fail(EMAVFI_E_ARG, "deform_conv2d: null pointer");
    if (B < 1 || C < 1 || O < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "deform_conv2d: bad shape");
    if ((size_t)H * W >= ((size_t)1 << 24)) return fail(EMAVFI_E_ARG, "deform_conv2d: H*W must be < 2^24");
    Plan P{};
    P.dtype = dtype; P.esize = dtype == EMAVFI_F32 ? 4 : 2;
    Layer L = mk(0, O, C);
    if (!deform_geometry(L, P.esize)) return fail(EMAVFI_E_UNSUPPORTED, "deform_conv2d: no kernel instantiation for C=%d O=%d", C, O);
    if ((size_t)H * W * L.ck * P.esize >= ((size_t)1 << 32))
        return fail(EMAVFI_E_ARG, "deform_conv2d: one sample's input plane must be < 4 GiB (32-bit byte offsets in the gather)");
    const size_t px = (size_t)B * H * W;
    const int ops = rup(O, 16);
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    void *xcl = ws.take(px * L.ck * P.esize);
    float *om = (float *)ws.take(px * 32 * sizeof(float));
    void *wp = ws.take(L.w_bytes);
    float *bp = (float *)ws.take((size_t)L.coutpad * sizeof(float));
    void *ycl = ws.take(px * ops * P.esize);
    void *zpage = ws.take(256);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "deform_conv2d: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    L.w_off = (char *)wp - (char *)workspace;
    L.b_off = (char *)bp - (char *)workspace;
    // the 16-bit LDS-window kernel contracts in f16 on chip: a bf16 call packs its bf16-rounded weights as f16 fragments
    if (dtype != EMAVFI_F32 && deform_pack3_shape(L.ck, L.nf, L.cin_take, L.cout)) L.pack3 = 1;
    const bool h_of_b = dtype == EMAVFI_BF16 && L.pack3 == 1;   // the LDS-window kernel contracts bf16-rounded weights stored as f16
    if (dtype == EMAVFI_F32 && deform_f32w_shape(L.ck, L.nf, L.cin_take, L.cout)) L.pack3 = 3;
    PackDesc d{L.cout, L.cin_raw, 0, L.cin_take, L.ck, 1, L.nf, 1, 0, h_of_b ? 1 : 0};
    d.pack3 = L.pack3;
    if (hipMemsetAsync(zpage, 0, 256, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "deform_conv2d: zero page memset failed");
    EMAVFI_TRY(launch_pack_conv(weight, bias, wp, bp, d, h_of_b ? (int)EMAVFI_F16 : dtype, s), "deform pack");
    EMAVFI_TRY(launch_nchw_to_cl(x, xcl, B, C, H, W, L.ck, dtype, s), "deform layout in");
    EMAVFI_TRY(launch_om_from_nchw(offset, mask, om, B, H, W, s), "deform offsets");
    EMAVFI_TRY(run_deform(P, L, workspace, xcl, L.ck, om, ycl, ops, ops, B, H, W, s, zpage), "deform_conv2d");
    EMAVFI_TRY(launch_cl_to_nchw(ycl, y, B, O, H, W, ops, 0, dtype, s), "deform layout out");
    return EMAVFI_OK;
}

// ---- ModulatedDeformConvPack.forward (ema_vfi.py:53-60) as ONE stage-level entry, routed through attention_block() - i.e. exactly as
// block i of emavfi_forward runs it: in the 16-bit modes at the reference width the one-launch kernel deform_pack3_kernel<T, FUSE_OFF = true>
// (offset_conv on the staged window, fast sigmoid, geometry, taps, fix-up), in fp32 conv3x3(EPI_OM) + the fp32 LDS-window DCN, under
// EMAVFI_AMP16 the fp16 offset_conv + fp32 DCN pair.  The model is EMA_VFI(3, C - 3, 1): the reference only ever builds this pack at
// mid_channels + 3 channels (ema_vfi.py:97).
static int mdcn_plan(Plan &P, int C, int dtype, int flags, bool &split, int &in_f16, int &out_f16)
{
    if (dtype == EMAVFI_F32X3) return fail(EMAVFI_E_UNSUPPORTED, "mdcn: EMAVFI_F32X3 has no stage entry of its own (its DCN is the exact fp32 one: use EMAVFI_F32)");
    if (C < 11 || !build_plan(P, 3, C - 3, 1, dtype))
        return fail(EMAVFI_E_UNSUPPORTED, "mdcn: C = %d is not mid_channels + 3 of a supported model (%s)", C, C < 11 ? "C < 11" : P.why);
    if (flags & ~(EMAVFI_MDCN_IN_F16 | EMAVFI_MDCN_OUT_F16 | EMAVFI_MDCN_SPLIT_TAIL)) return fail(EMAVFI_E_ARG, "mdcn: unknown flag bits 0x%x", flags);
    split = (flags & EMAVFI_MDCN_SPLIT_TAIL) != 0;
    in_f16 = (flags & EMAVFI_MDCN_IN_F16) ? 1 : 0;
    out_f16 = (flags & EMAVFI_MDCN_OUT_F16) ? 1 : 0;
    const bool one_launch = pack_fuses(P, 0) && P.dcn[0].pack3 != 0;
    if ((in_f16 || out_f16) && !(one_launch && P.dtype == EMAVFI_BF16 && !P.amp))
        return fail(EMAVFI_E_UNSUPPORTED, "mdcn: the f16 hand-off flags belong to the bf16 one-launch pack (C = 65..67)");
    if (split && !(pack_fuses(P, 0) && P.fpad - P.mid == 16)) return fail(EMAVFI_E_UNSUPPORTED, "mdcn: the split tail belongs to the one-launch pack");
    return EMAVFI_OK;
}

struct MdcnBuffers { void *blob, *xcl, *tail, *ycl; float *om, *xF, *yF; unsigned *census; };
static void mdcn_carve(const Plan &P, Workspace &ws, MdcnBuffers &m, int B, int H, int W, bool split)
{
    const size_t px = (size_t)B * H * W;
    m.blob = ws.take(P.total);
    m.xcl = ws.take(px * P.fps * P.esize);
    m.tail = split ? ws.take(px * 8 * P.esize) : nullptr;
    m.ycl = ws.take(px * P.fps * P.esize);
    m.om = (float *)ws.take(px * 32 * sizeof(float));
    m.xF = m.yF = nullptr;
    if (P.amp) {
        m.xF = (float *)ws.take(px * P.fpad * sizeof(float));
        m.yF = (float *)ws.take(px * P.fpad * sizeof(float));
    }
    m.census = (unsigned *)ws.take(kCensusBlock);
}

size_t emavfi_mdcn_workspace_bytes(int B, int C, int H, int W, int dtype, int flags)
{
    Plan P;
    bool split; int in_f16, out_f16;
    if (mdcn_plan(P, C, dtype, flags, split, in_f16, out_f16) != EMAVFI_OK) return 0;
    if (B < 1 || H < 1 || W < 1) { fail(EMAVFI_E_ARG, "mdcn: B, H, W must be >= 1"); return 0; }
    Workspace ws{nullptr, 0, 0};
    MdcnBuffers m;
    mdcn_carve(P, ws, m, B, H, W, split);
    return ws.used;
}

static int mdcn_impl(const float *x, const float *offset_weight, const float *offset_bias, const float *dcn_weight, const float *dcn_bias, float *y,
                     int B, int C, int H, int W, int dtype, int flags, void *workspace, size_t workspace_bytes, void *stream, Recorder &rec)
{
    Plan P;
    bool split; int in_f16, out_f16;
    if (const int rc = mdcn_plan(P, C, dtype, flags, split, in_f16, out_f16); rc != EMAVFI_OK) return rc;
    if (!x || !offset_weight || !offset_bias || !dcn_weight || !y || !workspace) return fail(EMAVFI_E_ARG, "mdcn: null pointer");
    if (B < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "mdcn: B, H, W must be >= 1");
    if (!aligned16(x) || !aligned16(y) || !aligned16(workspace)) return fail(EMAVFI_E_ARG, "mdcn: pointers must be 16-byte aligned");
    if ((size_t)B * H * W >= ((size_t)1 << 31) || (size_t)H * W >= ((size_t)1 << 24)) return fail(EMAVFI_E_ARG, "mdcn: B*H*W must be < 2^31 and H*W < 2^24");
    if ((size_t)H * W * P.fpad * (P.amp ? sizeof(float) : (size_t)P.esize) >= ((size_t)1 << 32)) return fail(EMAVFI_E_ARG, "mdcn: one sample's activation plane must be < 4 GiB");
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    MdcnBuffers m;
    mdcn_carve(P, ws, m, B, H, W, split);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "mdcn: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int kd = P.dtype, mid = P.mid;   // kernel storage type
    // pack the two layers exactly as emavfi_pack_weights packs attention_blocks.0 (both offset_conv copies where the plan has them)
    std::vector<const void *> params((size_t)emavfi_param_count(1), nullptr);
    params[P.off[0].param] = offset_weight; params[P.off[0].param + 1] = offset_bias;
    params[P.dcn[0].param] = dcn_weight; params[P.dcn[0].param + 1] = dcn_bias;
    if (hipMemsetAsync(m.blob, 0, P.total, s) != hipSuccess || hipMemsetAsync(m.census, 0, kCensusBlock, s) != hipSuccess)
        return fail(EMAVFI_E_LAUNCH, "mdcn: blob / census memset failed");
    EMAVFI_TRY(pack_layer(P.off[0], params.data(), m.blob, kd, s, P.amp), "mdcn pack offset_conv");
    if (P.has_offh) EMAVFI_TRY(pack_layer(P.offh[0], params.data(), m.blob, kd, s), "mdcn pack offset_conv (f16 fragments)");
    EMAVFI_TRY(pack_layer(P.dcn[0], params.data(), m.blob, kd, s), "mdcn pack dcn_v2");
    if (P.amp) EMAVFI_TRY(pack_layer(P.dcn32[0], params.data(), m.blob, EMAVFI_F32, s), "mdcn pack dcn_v2 (fp32 master weights)");
    if (P.amp) {
        EMAVFI_TRY(launch_nchw_to_cl(x, m.xF, B, C, H, W, P.fpad, EMAVFI_F32, s), "mdcn layout in (fp32)");
        EMAVFI_TRY(launch_nchw_to_cl(x, m.xcl, B, C, H, W, P.fps, EMAVFI_F16, s), "mdcn layout in (fp16 rounding)");
        if (const int rc = attention_block_amp(P, 0, m.blob, m.xcl, m.xF, m.yF, false, nullptr, m.om, B, H, W, s, rec); rc != EMAVFI_OK) return rc;
        EMAVFI_TRY(launch_cl_to_nchw(m.yF, y, B, C, H, W, P.fpad, 0, EMAVFI_F32, s), "mdcn layout out");
        return EMAVFI_OK;
    }
    const int xdt = in_f16 ? (int)EMAVFI_F16 : kd, ydt = out_f16 ? (int)EMAVFI_F16 : kd;
    if (split) {   // the first pack's input form: channels 0..mid-1 in the fusion pixels, channels mid.. in the compact tail buffer
        EMAVFI_TRY(launch_nchw_to_cl_sub(x, m.xcl, B, C, 0, mid, H, W, P.fps, xdt, s), "mdcn layout in");
        EMAVFI_TRY(launch_nchw_to_cl_sub(x, m.tail, B, C, mid, C - mid, H, W, kTailPs, xdt, s), "mdcn layout in (tail)");
    } else {
        EMAVFI_TRY(launch_nchw_to_cl(x, m.xcl, B, C, H, W, P.fps, xdt, s), "mdcn layout in");
    }
    if (const int rc = attention_block(P, 0, m.blob, m.xcl, m.ycl, m.om, m.tail, in_f16, out_f16, B, H, W, s, rec, m.census); rc != EMAVFI_OK) return rc;
    EMAVFI_TRY(launch_cl_to_nchw(m.ycl, y, B, C, H, W, P.fps, 0, ydt, s), "mdcn layout out");
    return EMAVFI_OK;
}

int emavfi_mdcn(const float *x, const float *offset_weight, const float *offset_bias, const float *dcn_weight, const float *dcn_bias, float *y,
                int B, int C, int H, int W, int dtype, int flags, void *workspace, size_t workspace_bytes, void *stream)
{
    Recorder rec;
    return mdcn_impl(x, offset_weight, offset_bias, dcn_weight, dcn_bias, y, B, C, H, W, dtype, flags, workspace, workspace_bytes, stream, rec);
}

int emavfi_mdcn_profiled(const float *x, const float *offset_weight, const float *offset_bias, const float *dcn_weight, const float *dcn_bias, float *y,
                         int B, int C, int H, int W, int dtype, int flags, void *workspace, size_t workspace_bytes, void *const *events, int n_events,
                         void *stream)
{
    if (!events || n_events < 2) return fail(EMAVFI_E_ARG, "mdcn_profiled: events array required");
    Recorder rec;
    rec.events = events; rec.n_events = n_events;
    return mdcn_impl(x, offset_weight, offset_bias, dcn_weight, dcn_bias, y, B, C, H, W, dtype, flags, workspace, workspace_bytes, stream, rec);
}

// ---- census of the one-launch packs (round 6): what deform_pack3_kernel itself counted while it ran - the (wave, tap)
// groups that took the fix-up, the samples outside the staged window, the largest |offset| - reduced over the launch's 64 atomic slots
// into out[block][4] (u64, DEVICE memory): {fix-up wave-taps, all wave-taps, samples outside, max |offset| as fp32 bits}.  All wave-taps
// is 0 for a block that did not run the one-launch kernel (fp32 / autocast modes, other widths): nothing was counted.
static unsigned long long pack3_wave_taps(int B, int H, int W) { return (unsigned long long)B * ((H + 15) / 16) * ((W + 15) / 16) * 4ull * 9ull; }

int emavfi_forward_census(int in_channels, int mid_channels, int num_blocks, int B, int H, int W, int dtype, const void *workspace,
                          size_t workspace_bytes, unsigned long long *out, void *stream)
{
    Plan P;
    if (!build_plan(P, in_channels, mid_channels, num_blocks, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s", P.why);
    if (B < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "forward_census: B, H, W must be >= 1");
    if (!workspace || !out) return fail(EMAVFI_E_ARG, "forward_census: null pointer");
    Workspace ws{(char *)const_cast<void *>(workspace), workspace_bytes, 0};
    FwdBuffers f;
    carve_forward(P, ws, f, B, H, W);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "forward_census: this forward's workspace has %zu bytes, got %zu", ws.used, workspace_bytes);
    unsigned long long totals[kMaxBlocks] = {};
    for (int i = 0; i < P.nb; ++i) totals[i] = pack_fuses(P, i) && P.dcn[i].pack3 == 1 ? pack3_wave_taps(B, H, W) : 0ull;
    EMAVFI_TRY(launch_census_reduce(f.census, out, P.nb, totals, (hipStream_t)stream), "forward_census");
    return EMAVFI_OK;
}

int emavfi_mdcn_census(int B, int C, int H, int W, int dtype, int flags, const void *workspace, size_t workspace_bytes, unsigned long long *out, void *stream)
{
    Plan P;
    bool split; int in_f16, out_f16;
    if (const int rc = mdcn_plan(P, C, dtype, flags, split, in_f16, out_f16); rc != EMAVFI_OK) return rc;
    if (B < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "mdcn_census: B, H, W must be >= 1");
    if (!workspace || !out) return fail(EMAVFI_E_ARG, "mdcn_census: null pointer");
    Workspace ws{(char *)const_cast<void *>(workspace), workspace_bytes, 0};
    MdcnBuffers m;
    mdcn_carve(P, ws, m, B, H, W, split);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "mdcn_census: this stage's workspace has %zu bytes, got %zu", ws.used, workspace_bytes);
    unsigned long long totals[kMaxBlocks] = {};
    totals[0] = pack_fuses(P, 0) && P.dcn[0].pack3 == 1 ? pack3_wave_taps(B, H, W) : 0ull;
    EMAVFI_TRY(launch_census_reduce(m.census, out, 1, totals, (hipStream_t)stream), "mdcn_census");
    return EMAVFI_OK;
}

// ---- context_encoding and reconstruction as stage-level entries (SURVEY 8b's proposed emavfi_context / emavfi_reconstruct): the model is
// EMA_VFI(3, mid_channels, 3) - the stage does not depend on num_blocks -, the layers are packed and run exactly as emavfi_forward
// packs and runs them (context_stage / reconstruction_stage above), the tensors cross the boundary as NCHW fp32.
static int stage_plan(Plan &P, int mid, int dtype, int B, int H, int W, const char *what)
{
    if (dtype == EMAVFI_F32X3) return fail(EMAVFI_E_UNSUPPORTED, "%s: EMAVFI_F32X3 has stage entries for conv3x3 only", what);
    if (!build_plan(P, 3, mid, 3, dtype)) return fail(EMAVFI_E_UNSUPPORTED, "%s: %s", what, P.why);
    if (B < 1 || H < 1 || W < 1) return fail(EMAVFI_E_ARG, "%s: B, H, W must be >= 1", what);
    if ((size_t)B * H * W >= ((size_t)1 << 31) || (size_t)H * W >= ((size_t)1 << 24) || (size_t)H * W * P.fpad * (P.amp ? sizeof(float) : (size_t)P.esize) >= ((size_t)1 << 32))
        return fail(EMAVFI_E_ARG, "%s: B*H*W must be < 2^31, H*W < 2^24 and one sample's activation plane < 4 GiB", what);
    return EMAVFI_OK;
}
struct StageBuffers { void *blob; float *zeros9; FwdBuffers f; };
static void stage_carve(const Plan &P, Workspace &ws, StageBuffers &b, int B, int H, int W)
{
    b.blob = ws.take(P.total);
    b.zeros9 = (float *)ws.take(((size_t)P.mid * 2 * P.mid * 9 + P.mid) * sizeof(float));   // a zero motion_estimation.0 for the context fold
    carve_forward(P, ws, b.f, B, H, W);
}

size_t emavfi_context_workspace_bytes(int B, int mid_channels, int H, int W, int dtype)
{
    Plan P;
    if (stage_plan(P, mid_channels, dtype, B, H, W, "context") != EMAVFI_OK) return 0;
    Workspace ws{nullptr, 0, 0};
    StageBuffers b;
    stage_carve(P, ws, b, B, H, W);
    return ws.used;
}

int emavfi_context(const float *feat, const float *const *params, float *ctx, int B, int mid_channels, int H, int W, int dtype,
                   void *workspace, size_t workspace_bytes, void *stream)
{
    Plan P;
    if (const int rc = stage_plan(P, mid_channels, dtype, B, H, W, "context"); rc != EMAVFI_OK) return rc;
    if (!feat || !params || !ctx || !workspace) return fail(EMAVFI_E_ARG, "context: null pointer");
    for (int i = 0; i < 8; ++i)
        if (!params[i]) return fail(EMAVFI_E_ARG, "context: params[%d] is null", i);
    if (!aligned16(feat) || !aligned16(workspace)) return fail(EMAVFI_E_ARG, "context: pointers must be 16-byte aligned");
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    StageBuffers b;
    stage_carve(P, ws, b, B, H, W);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "context: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int kd = P.dtype, mid = P.mid;
    std::vector<const void *> all((size_t)emavfi_param_count(3), nullptr);
    all[P.c0.param] = params[0]; all[P.c0.param + 1] = params[1];
    all[P.c1.param] = params[2]; all[P.c1.param + 1] = params[3];
    all[P.c2.param] = params[4]; all[P.c2.param + 1] = params[5];
    if (hipMemsetAsync(b.blob, 0, P.total, s) != hipSuccess || hipMemsetAsync(b.zeros9, 0, ((size_t)mid * 2 * mid * 9 + mid) * sizeof(float), s) != hipSuccess)
        return fail(EMAVFI_E_LAUNCH, "context: memset failed");
    EMAVFI_TRY(pack_layer(P.c0, all.data(), b.blob, kd, s, P.amp), "context pack 0");
    EMAVFI_TRY(pack_layer(P.c1, all.data(), b.blob, kd, s, P.amp), "context pack 1");
    EMAVFI_TRY(pack_layer(P.c2, all.data(), b.blob, kd, s, P.amp), "context pack 2");
    EMAVFI_TRY(launch_pack_ctx(params[6], params[7], b.zeros9, b.zeros9 + (size_t)mid * 2 * mid * 9, (float *)((char *)b.blob + P.ctx_off), mid, P.amp ? 1 : 0, s),
               "context pack linear");
    // `feat` as the forward stores it: channels 0..mid-1 of the fusion pixels, f16 in the bf16 model with the one-launch packs (Plan::feat16)
    EMAVFI_TRY(launch_nchw_to_cl(feat, b.f.fu0, B, mid, H, W, P.fps, P.feat16 ? (int)EMAVFI_F16 : kd, s), "context layout in");
    Recorder rec;
    if (const int rc = context_stage(P, b.blob, b.f.fu0, b.f, B, H, W, s, rec); rc != EMAVFI_OK) return rc;
    if (hipMemcpyAsync(ctx, b.f.ctx, (size_t)B * mid * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "context: copy out failed");
    return EMAVFI_OK;
}

size_t emavfi_reconstruct_workspace_bytes(int B, int mid_channels, int H, int W, int dtype)
{
    return emavfi_context_workspace_bytes(B, mid_channels, H, W, dtype);   // the same carving
}

int emavfi_reconstruct(const float *fused, const float *const *params, float *out, int B, int mid_channels, int H, int W, int dtype,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    Plan P;
    if (const int rc = stage_plan(P, mid_channels, dtype, B, H, W, "reconstruct"); rc != EMAVFI_OK) return rc;
    if (!fused || !params || !out || !workspace) return fail(EMAVFI_E_ARG, "reconstruct: null pointer");
    for (int i = 0; i < 6; ++i)
        if (!params[i]) return fail(EMAVFI_E_ARG, "reconstruct: params[%d] is null", i);
    if (!aligned16(fused) || !aligned16(out) || !aligned16(workspace)) return fail(EMAVFI_E_ARG, "reconstruct: pointers must be 16-byte aligned");
    Workspace ws{(char *)workspace, workspace_bytes, 0};
    StageBuffers b;
    stage_carve(P, ws, b, B, H, W);
    if (ws.used > workspace_bytes) return fail(EMAVFI_E_WORKSPACE, "reconstruct: workspace needs %zu bytes, got %zu", ws.used, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int kd = P.dtype, mid = P.mid;
    std::vector<const void *> all((size_t)emavfi_param_count(3), nullptr);
    all[P.r0.param] = params[0]; all[P.r0.param + 1] = params[1];
    all[P.r1.param] = params[2]; all[P.r1.param + 1] = params[3];
    all[P.r2.param] = params[4]; all[P.r2.param + 1] = params[5];
    if (hipMemsetAsync(b.blob, 0, P.total, s) != hipSuccess) return fail(EMAVFI_E_LAUNCH, "reconstruct: memset failed");
    EMAVFI_TRY(pack_layer(P.r0, all.data(), b.blob, kd, s, P.amp), "reconstruct pack 0");
    EMAVFI_TRY(pack_layer(P.r1, all.data(), b.blob, kd, s, P.amp), "reconstruct pack 1");
    EMAVFI_TRY(pack_layer(P.r2, all.data(), b.blob, kd, s, P.amp), "reconstruct pack 2");
    // the fusion tensor as the last attention block leaves it: mid + 3 channels in P.fps-channel pixels of the storage type
    EMAVFI_TRY(launch_nchw_to_cl(fused, b.f.fu0, B, mid + 3, H, W, P.fps, kd, s), "reconstruct layout in");
    Recorder rec;
    return reconstruction_stage(P, b.blob, b.f.fu0, b.f, out, B, H, W, s, rec);
}

}  // extern "C"
