# conv_wreg.inl: parity tests, then library-variant A/B on ONE box (two interleaved rounds): tools/wreg_ab3.sh TAG ... ("" = product)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wreg
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -k "${WREG_TESTS:-pool_fused or conv3x3_matches_aten or context}" > gpurun_out/wreg/test.log 2>&1 || { tail -40 gpurun_out/wreg/test.log; exit 1; }
tail -2 gpurun_out/wreg/test.log
bash tools/wreg_ab.sh "$@"
