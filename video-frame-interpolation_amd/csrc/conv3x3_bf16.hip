#include "conv3x3.inl"
#include <cstdlib>
// bf16 launcher: full-resolution single-chunk layers go to the persistent, weights-resident kernel
// (CK, NF, waves): 64->64, 64->32 / 64->2, 67->27 of the mid_channels = 64 model.
int launch_conv3x3_bf16(const ConvParams &p, hipStream_t s)
{
    static const bool off = getenv("EMAVFI_NO_PERSISTENT_CONV") != nullptr;  // A/B switch for measurements
    if (!off && p.stride == 1 && p.nchunk == 1 && p.npass == 1) {
        // measured at B=8 x 720p (us per launch, tile-per-workgroup -> persistent): 64->64 670 -> 644,
        // 64->32 / 64->2 414 -> 370, 67->27 685 -> 557.  NOT used where it loses: 67->64 with 4 waves
        // (771 -> 915: one 4-wave workgroup per CU cannot overlap its own phases), 6->64, 32->3 (no gain).
        if (p.ck == 64 && p.nf == 2) return launch_conv_persist<bf16_t, 64, 2, 8>(p, s);
        if (p.ck == 64 && p.nf == 1) return launch_conv_persist<bf16_t, 64, 1, 8>(p, s);
        if (p.ck == 80 && p.nf == 1) return launch_conv_persist<bf16_t, 80, 1, 8>(p, s);
    }
    return launch_conv_any<bf16_t>(p, s);
}
