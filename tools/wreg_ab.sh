# timing of conv_wreg.inl variants on ONE box: tools/wreg_ab.sh TAG ... ("" = product library); two interleaved rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wreg
names=""
for r in 1 2; do
  for tag in "$@"; do
    lib=$PWD/video-frame-interpolation_amd/emavfi/lib/libemavfi${tag:+_$tag}.so
    EMAVFI_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/wreg/${tag:-prod}_$r.json 2> gpurun_out/wreg/${tag:-prod}_$r.err || exit 1
    names="$names ${tag:-prod}_$r"
  done
done
python tools/ab_print.py gpurun_out/wreg $names --match=nf=8,pool
