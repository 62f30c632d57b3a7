#!/bin/bash
# fp32-mode profile artefacts (run from the repo root through gpurun):  tools/profile_fp32.sh OUTDIR
# the exact-fp32 forward at B = 8 x 720p and at BASELINE configs[1] (B = 16 x 256x256): rocprofv3 kernel stats + the two PMC traffic passes
out=$1
mkdir -p $out
export TMPDIR=/tmp
for cfg in "720 --batch 8 --height 720 --width 1280 --steps 6 --warmup 1" "256 --batch 16 --height 256 --width 256 --steps 20 --warmup 2"; do
  set -- $cfg; tag=$1; shift
  args="--dtype fp32 --no-extras $*"
  short=$(echo "$args" | sed 's/--steps [0-9]*/--steps 2/; s/--warmup [0-9]*/--warmup 1/')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 bench.py $args > $out/stats_$tag.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$tag -- python3 bench.py $short > $out/pmc_fetch_$tag.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$tag -- python3 bench.py $short > $out/pmc_write_$tag.log 2>&1 || exit 1
done
find $out -name "*.db" -delete
find $out -name "*_kernel_trace.csv" -size +8M -delete
ls -R $out | head -30
