#!/usr/bin/env python3
"""Separate 'fresh workspace' from 'true concurrency' when two-stream results differ from the serial ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, lib, synth
DEV = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = EMA_VFI(compute_dtype=dtype).to(DEV).eval()
m.load_state_dict(synth.synthetic_state_dict(seed=0))
xa = [t.to(DEV) for t in synth.synthetic_frames(51, 2, 192, 256, "natural")]
xb = [t.to(DEV) for t in synth.synthetic_frames(52, 1, 360, 640, "stress")]

def diff(a, b, tag):
    d = (a - b).abs()
    n = int((d > 0).sum())
    print(f"  {tag}: {n} differing of {d.numel()}, max {d.max().item():.3e}", flush=True)
    if n:
        idx = (d > 0).nonzero()[:5].tolist()
        print("     first:", idx)

with torch.no_grad():
    ref_a, ref_b, taps_b = m(*xa).clone(), None, None
    ref_b, taps_b = m(*xb, return_taps=True)
    ref_b = ref_b.clone(); taps_b = {k: v.clone() for k, v in taps_b.items()}
    torch.cuda.synchronize()
    print("1) same input again on the default stream")
    diff(m(*xb), ref_b, "b")
    print("2) alone on a NEW stream (fresh workspace, nothing concurrent)")
    for t in range(3):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            o, tp = m(*xb, return_taps=True)
        torch.cuda.synchronize()
        diff(o, ref_b, f"b try {t}")
        for k in tp:
            if not torch.equal(tp[k], taps_b[k]):
                diff(tp[k], taps_b[k], "   first differing stage " + k)
                break
    print("3) fresh workspace poisoned with NaN bit patterns first")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m(*xb)
        torch.cuda.synchronize()
        key = (0, s.cuda_stream)
        lib._ws_cache[key].fill_(0xFF)
        o, tp = m(*xb, return_taps=True)
    torch.cuda.synchronize()
    diff(o, ref_b, "b poisoned")
    for k in tp:
        if not torch.equal(tp[k], taps_b[k]):
            diff(tp[k], taps_b[k], "   first differing stage " + k)
            break
    print("4) two streams interleaved (taps on b: first differing stage)")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    oa, ob = [], []
    for _ in range(12):
        with torch.cuda.stream(s1):
            oa.append(m(*xa))
        with torch.cuda.stream(s2):
            ob.append(m(*xb, return_taps=True))
    torch.cuda.synchronize()
    for i, o in enumerate(oa):
        if not torch.equal(o, ref_a):
            diff(o, ref_a, f"a[{i}]")
    for i, (o, tp) in enumerate(ob):
        if not torch.equal(o, ref_b):
            diff(o, ref_b, f"b[{i}]")
            for k in tp:
                if k in taps_b and not torch.equal(tp[k], taps_b[k]):
                    diff(tp[k], taps_b[k], "   first differing stage " + k)
                    break
    print("done")
