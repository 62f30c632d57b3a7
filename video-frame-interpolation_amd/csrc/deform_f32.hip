#include "deform_f32w.inl"
int launch_deform_f32(const DeformParams &p, hipStream_t s)
{
    if (p.pack3 == 3) {   // weights in the deform_f32w.inl layout (host: deform_f32w_shape)
        if (!deform_f32w_shape(p.ck, p.nf, p.cin_real, p.cout_real) || p.off_w || p.x_tail) return -2;
        return launch_deform_f32w(p, s);
    }
    return launch_deform_any<float>(p, s);
}
