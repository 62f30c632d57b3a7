#include <cstdlib>
#include "deform_pack3.inl"

bool deform16_can_fuse_offset_conv(int ck, int nf, int cin_real, int off_ck, int off_nf)
{
    static const bool off = getenv("EMAVFI_NO_FUSED_OFFSET") != nullptr;  // A/B switch
    return !off && EMAVFI_PACK3 && ck == 80 && nf == 3 && cin_real > 64 && cin_real <= 67 && off_ck == ck && off_nf == 1;
}

int launch_deform_bf16(const DeformParams &p, hipStream_t s) { return launch_deform16<bf16_t>(p, s); }

