#!/bin/bash
# Evidence for VERDICT r4 item 1 (run through gpurun from the repo root): tools/twostream_profile.sh OUTDIR
#   kernel traces (rocprofv3 --kernel-trace) of the pipelined forward in three forms + what overlapped (tools/twostream_overlap.py),
#   power / clock traces (tools/power_trace.py) of the same forms.
out=${1:-gpurun_out/r05_twostream_prof}
mkdir -p $out
export TMPDIR=/tmp
for v in 1:0:0:1 2:0:0:1 2:-1:0:1 2:0:1:0; do
  IFS=: read p st ow r2 <<< "$v"
  tag=p${p}_s${st}_w${ow}_r${r2}
  export EMAVFI_PIPELINE=$p EMAVFI_PIPELINE_STAGGER=$st EMAVFI_RING_ONE_WG=$ow EMAVFI_CONV_RING2=$r2
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-events > $out/trace_$tag.log 2>&1 || { tail -5 $out/trace_$tag.log; exit 1; }
  f=$(find $out/trace_$tag -name "*kernel_trace.csv" | head -1)
  echo "== pieces $p stagger $st ring-one-wg $ow ring2 $r2: $(grep -o '"value": [0-9.]*' $out/trace_$tag.log | head -1) frames/s under the tracer" > $out/overlap_$tag.txt
  python3 tools/twostream_overlap.py $f >> $out/overlap_$tag.txt 2>&1
  cat $out/overlap_$tag.txt
  timeout -k 10 120 python3 tools/power_trace.py bf16 5 > $out/power_$tag.txt 2>&1
  tail -5 $out/power_$tag.txt
  find $out/trace_$tag -name "*.db" -delete
  find $out/trace_$tag -name "*.csv" -size +4M -delete
done
