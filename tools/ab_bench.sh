#!/bin/bash
# A/B of library variants on ONE box: tools/ab_bench.sh OUTDIR TAG1 TAG2 ... ("" = product library); two interleaved rounds
out=$1; shift
mkdir -p $out
for round in 1 2; do
  for tag in "$@"; do
    lib=$PWD/video-frame-interpolation_amd/emavfi/lib/libemavfi${tag:+_$tag}.so
    EMAVFI_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extras > $out/ab_${tag:-prod}_$round.json 2> $out/ab_${tag:-prod}_$round.err || exit 1
    EMAVFI_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extras --dtype fp16 > $out/abh_${tag:-prod}_$round.json 2>> $out/ab_${tag:-prod}_$round.err || exit 1
  done
done
python - "$out" "$@" <<'PY'
import json, sys, glob
out = sys.argv[1]
for tag in sys.argv[2:]:
    name = tag or "prod"
    rows = []
    for f in sorted(glob.glob(f"{out}/ab_{name}_*.json")):
        d = json.load(open(f))
        k = {x["kernel"]: x["avg_us"] for x in d["kernels"]}
        dk = [v for n, v in k.items() if n.startswith("deform")][0]
        dh = json.load(open(f"{out}/abh_" + f.rsplit("/ab_", 1)[1]))
        kh = [x["avg_us"] for x in dh["kernels"] if x["kernel"].startswith("deform")][0]
        rows.append((d["value"], dk, k.get("conv3x3<bf16,ck=64,nf=2,s=1>", 0), sum(v for n, v in k.items() if n.endswith("s=2>")), dh["value"], kh))
    print(f"{name:8s}", " | ".join(f"bf16 {v:6.1f} fps pack {u:7.1f} conv64x64 {c:6.1f} stride2 {c1:6.1f} us; fp16 {h:6.1f} fps pack {kk:7.1f} us" for v, u, c, c1, h, kk in rows))
PY
