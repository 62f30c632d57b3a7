#include "deform.inl"
int launch_deform_bf16(const DeformParams &p, hipStream_t s) { return launch_deform_any<bf16_t>(p, s); }
