#include <cstdlib>
#include "deform_lds.inl"
#ifndef EMAVFI_DEFORM_RPW
#define EMAVFI_DEFORM_RPW 2  // rows per wave; 1 (16 waves, <=128 VGPRs) spills and is 7x slower
#endif
// the reference width (mid_channels 64 -> 67 channels, k-groups to 80): LDS-staged window of 72 channels
static bool lds_shape(int ck, int nf, int cin_real) { return ck == 80 && nf == 3 && cin_real <= 72; }

bool deform_bf16_can_fuse_offset_conv(int ck, int nf, int cin_real, int off_ck, int off_nf)
{
    static const bool off = getenv("EMAVFI_NO_FUSED_OFFSET") != nullptr;  // A/B switch
    return !off && lds_shape(ck, nf, cin_real) && off_ck == ck && off_nf == 1;
}

int launch_deform_bf16(const DeformParams &p, hipStream_t s)
{
    if (lds_shape(p.ck, p.nf, p.cin_real)) {
        if (p.off_w) return launch_deform_lds<80, 3, 72, 2, EMAVFI_DEFORM_RPW, true>(p, s);
        return launch_deform_lds<80, 3, 72, 2, EMAVFI_DEFORM_RPW, false>(p, s);
    }
    if (p.off_w) return -1;  // the host only asks for fusion after deform_bf16_can_fuse_offset_conv()
    return launch_deform_any<bf16_t>(p, s);
}
