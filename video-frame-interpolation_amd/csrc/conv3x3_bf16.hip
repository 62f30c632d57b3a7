#include "conv3x3.inl"
int launch_conv3x3_bf16(const ConvParams &p, hipStream_t s) { return launch_conv_any<bf16_t>(p, s); }
