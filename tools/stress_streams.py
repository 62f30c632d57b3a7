#!/usr/bin/env python3
"""Forward beside heavy foreign work on another stream: which stages (taps) ever differ from the serial result?
usage: stress_streams.py [dtype=bf16] [forwards=600] [aggressor=mm|forward|none]      (EMAVFI_LIB selects a variant build)
The aggressor 'mm' is a torch bf16 GEMM (hipBLASLt/rocBLAS, MFMA-bound) - the strongest trigger found in round 2."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, lib, synth
DEV = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
aggr = sys.argv[3] if len(sys.argv) > 3 else "mm"
m = EMA_VFI(compute_dtype=dtype).to(DEV).eval()
m.load_state_dict(synth.synthetic_state_dict(seed=0))
xa = [t.to(DEV) for t in synth.synthetic_frames(51, 2, 192, 256, "natural")]
xb = [t.to(DEV) for t in synth.synthetic_frames(52, 1, 360, 640, "stress")]
A = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
with torch.no_grad():
    ref, taps = m(*xb, return_taps=True)
    ref = ref.clone(); taps = {k: v.clone() for k, v in taps.items()}
    m(*xa); A @ A
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    stage_bad, first_bad, out_bad = collections.Counter(), collections.Counter(), 0
    for it in range(n // 10):
        runs = []
        for _ in range(10):
            with torch.cuda.stream(s1):
                if aggr == "mm":
                    for _ in range(40):
                        A @ A
                elif aggr == "forward":
                    m(*xa)
            with torch.cuda.stream(s2):
                runs.append(m(*xb, return_taps=True))
        torch.cuda.synchronize()
        for o, tp in runs:
            first = True
            for k in tp:
                if k in taps and not torch.equal(tp[k], taps[k]):
                    stage_bad[k] += 1
                    if first:
                        first_bad[k] += 1
                        first = False
            out_bad += int(not torch.equal(o, ref))
    print(f"{dtype}, aggressor {aggr}: {out_bad} of {n} forwards differ from the serial result")
    print("  stages that differed (count):", dict(stage_bad))
    print("  first differing stage (count):", dict(first_bad))
