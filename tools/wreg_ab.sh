set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wreg
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3_matches_aten" > gpurun_out/wreg/test.log 2>&1 || { tail -30 gpurun_out/wreg/test.log; exit 1; }
tail -2 gpurun_out/wreg/test.log
for r in 1 2; do
EMAVFI_CONV_WREG=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/wreg/old_$r.json 2> gpurun_out/wreg/old_$r.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/wreg/new_$r.json 2> gpurun_out/wreg/new_$r.err
done
python tools/ab_print.py gpurun_out/wreg old_1 new_1 old_2 new_2 --match=nf=8,nf=4,pool
