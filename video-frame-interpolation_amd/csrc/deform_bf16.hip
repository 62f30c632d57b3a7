#include "deform_lds.inl"
int launch_deform_bf16(const DeformParams &p, hipStream_t s)
{
    // the reference width (mid_channels 64 -> 67 channels padded to 80): LDS-staged gather
    if (p.ck == 80 && p.nf == 3) return launch_deform_lds<80, 3, 2>(p, s);
    return launch_deform_any<bf16_t>(p, s);
}
