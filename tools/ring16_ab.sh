#!/bin/bash
# tools/ring16_ab.sh OUT: parity, then the kernel's own time under rocprofv3 for both variants, two interleaved rounds
out=${1:-gpurun_out/ring16}; mkdir -p $out; export TMPDIR=/tmp
python3 tools/ring16_ab.py check 2>&1 | grep -v amdgpu.ids | tee $out/check.txt || exit 1
for r in 1 2; do for v in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/v${v}_$r -- python3 tools/ring16_ab.py run $v 20 > $out/v${v}_$r.log 2>&1 || { tail -5 $out/v${v}_$r.log; exit 1; }
  f=$(find $out/v${v}_$r -name "*kernel_stats.csv" | head -1)
  echo "variant $v round $r: $(grep -h 'ms per call' $out/v${v}_$r.log)"; grep -h "conv3x3_ring" $f | cut -d, -f1-4 | cut -c1-200
  find $out/v${v}_$r -name "*.db" -delete; find $out/v${v}_$r -name "*trace.csv" -delete
done; done
