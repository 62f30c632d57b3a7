#include "deform_lds.inl"
#ifndef EMAVFI_DEFORM_RPW
#define EMAVFI_DEFORM_RPW 2  // rows per wave; 1 (16 waves, <=128 VGPRs) spills and is 7x slower
#endif
int launch_deform_bf16(const DeformParams &p, hipStream_t s)
{
    // the reference width (mid_channels 64 -> 67 channels, k-groups to 80): LDS-staged window of 72 channels
    if (p.ck == 80 && p.nf == 3 && p.cin_real <= 72) return launch_deform_lds<80, 3, 72, 2, EMAVFI_DEFORM_RPW>(p, s);
    return launch_deform_any<bf16_t>(p, s);
}
