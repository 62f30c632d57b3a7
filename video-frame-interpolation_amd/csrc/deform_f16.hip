#include "deform_pack3.inl"
int launch_deform_f16(const DeformParams &p, hipStream_t s) { return launch_deform16<half_t>(p, s); }
