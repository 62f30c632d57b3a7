# conv_wreg.inl: parity tests, then environment-switch A/B on ONE box (two interleaved rounds): tools/wreg_ab2.sh "ENV=.. ENV2=.." ...  ("" = defaults)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wreg
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -k "${WREG_TESTS:-pool_fused or conv3x3_matches_aten or context}" > gpurun_out/wreg/test.log 2>&1 || { tail -40 gpurun_out/wreg/test.log; exit 1; }
tail -2 gpurun_out/wreg/test.log
names=""
for r in 1 2; do
  i=0
  for envs in "$@"; do
    env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/wreg/v${i}_$r.json 2> gpurun_out/wreg/v${i}_$r.err || exit 1
    names="$names v${i}_$r"; i=$((i+1))
  done
done
python tools/ab_print.py gpurun_out/wreg $names --match=nf=8,pool,fold
