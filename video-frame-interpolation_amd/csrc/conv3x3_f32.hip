#include "conv3x3.inl"
int launch_conv3x3_f32(const ConvParams &p, hipStream_t s) { return launch_conv_any<float>(p, s); }
