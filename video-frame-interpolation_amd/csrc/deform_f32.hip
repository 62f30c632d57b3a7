#include "deform.inl"
int launch_deform_f32(const DeformParams &p, hipStream_t s) { return launch_deform_any<float>(p, s); }
