#include "conv3x3.inl"
#include "conv_first.inl"
#include "conv_ring_first.inl"
#include <cstdlib>
int launch_conv3x3_bf16(const ConvParams &p, hipStream_t s)
{
    static const bool off = getenv("EMAVFI_NO_PERSISTENT_CONV") != nullptr;  // A/B switch for measurements
    return launch_conv16<bf16_t>(p, s, off);
}
#if EMAVFI_CONV_STAMPS
extern "C" int emavfi_debug_conv_stamps(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -2;
    unsigned long long z[8] = {0};
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), z, sizeof z) != hipSuccess) return -3;
    return 0;
}
#endif

int launch_conv_first_bf16(const FirstParams &p, hipStream_t s) { return launch_conv_first_t<bf16_t>(p, s); }
int launch_conv_ringfirst_bf16(const FirstParams &fp, const ConvParams &p, hipStream_t s) { return launch_conv_ringfirst_t<bf16_t>(fp, p, s); }
