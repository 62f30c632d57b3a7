#include "conv3x3.inl"
#include <cstdlib>
int launch_conv3x3_bf16(const ConvParams &p, hipStream_t s)
{
    static const bool off = getenv("EMAVFI_NO_PERSISTENT_CONV") != nullptr;  // A/B switch for measurements
    return launch_conv16<bf16_t>(p, s, off);
}
int launch_conv_tail_bf16(const TailParams &p, hipStream_t s) { return launch_conv_tail<bf16_t>(p, s); }
