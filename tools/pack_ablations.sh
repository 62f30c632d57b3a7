# Timing-only ablations of the pack kernel's tap loop (round 5; the board is energy-bound under this kernel, tools/power_per_kernel.py:
# what a component costs in TIME is what it costs in joules).  Libraries: make TAG=_ablN EXTRA=-DEMAVFI_P3_ABL=N for N in the list below
# (bits: csrc/deform_pack3.inl).  Run from the repo root through gpurun:  bash tools/pack_ablations.sh "" abl1 abl2 ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/abl
for r in 1 2; do
  for tag in "$@"; do
    lib=$PWD/video-frame-interpolation_amd/emavfi/lib/libemavfi${tag:+_$tag}.so
    EMAVFI_LIB=$lib timeout -k 10 120 python tools/mode_kernels.py bf16 2> gpurun_out/abl/${tag:-prod}_$r.err | grep -E "deform|frames/s" | tr '\n' ' ' | sed "s/^/${tag:-prod} round $r: /"
    echo
  done
done
