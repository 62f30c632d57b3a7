#!/bin/bash
# the pipelined (two-stream) forward against the one-sequence forward, same box, interleaved rounds.
#   tools/twostream_ab.sh OUTDIR [ROUNDS] [VARIANTS]     variant = pieces:stagger:onewg:ring2
#   pieces / stagger = EMAVFI_PIPELINE / EMAVFI_PIPELINE_STAGGER; onewg = EMAVFI_RING_ONE_WG (the persistent ring kernels launch one
#   workgroup per CU, leaving half of each CU's LDS to the other stream's pack kernel); ring2 = EMAVFI_CONV_RING2 (0: the 113 KiB
#   two-layer kernel, which cannot share a CU with a pack workgroup, runs as two ring launches)
out=${1:-gpurun_out/r05_twostream}; rounds=${2:-2}
variants=${3:-"1:0:0:1 2:0:0:1 2:-1:0:1 1:0:1:0 2:0:1:0 2:-1:1:0 4:0:1:0 2:0:0:0 1:0:0:0"}
mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in $variants; do
    IFS=: read p st ow r2 <<< "$v"
    tag=p${p}_s${st}_w${ow}_r${r2}
    EMAVFI_PIPELINE=$p EMAVFI_PIPELINE_STAGGER=$st EMAVFI_RING_ONE_WG=$ow EMAVFI_CONV_RING2=$r2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-events \
        > $out/${tag}_$r.json 2> $out/${tag}_$r.err || { echo "variant $v failed"; tail -5 $out/${tag}_$r.err; exit 1; }
  done
done
python - "$out" $variants <<'PY'
import json, sys, glob
out = sys.argv[1]
base = None
for v in sys.argv[2:]:
    p, st, ow, r2 = v.split(":")
    files = sorted(glob.glob(f"{out}/p{p}_s{st}_w{ow}_r{r2}_*.json"))
    vals = [json.load(open(f))["value"] for f in files]
    ms = [json.load(open(f))["ms_per_step"] for f in files]
    mean = sum(vals) / len(vals)
    if base is None:
        base = mean
    print(f"pieces {p:>2s} stagger {st:>2s} ring-one-wg {ow} ring2 {r2}: " + " ".join(f"{x:7.1f}" for x in vals) + f" frames/s  ({' '.join(f'{m:.3f}' for m in ms)} ms/step)  {100 * (mean / base - 1):+5.1f} %")
PY
