#include "conv3x3.inl"
#include "conv_first.inl"
#include "conv_ring_first.inl"
#include <cstdlib>
int launch_conv3x3_f16(const ConvParams &p, hipStream_t s)
{
    return launch_conv16<half_t>(p, s, (emavfi_switches() & SW_NO_PERSISTENT_CONV) != 0);   // EMAVFI_NO_PERSISTENT_CONV: A/B switch for measurements
}

int launch_conv_first_f16(const FirstParams &p, hipStream_t s) { return launch_conv_first_t<half_t>(p, s); }
int launch_conv_ringfirst_f16(const FirstParams &fp, const ConvParams &p, hipStream_t s) { return launch_conv_ringfirst_t<half_t>(fp, p, s); }
