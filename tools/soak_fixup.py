#!/usr/bin/env python3
"""Soak of the pack kernel's fix-up hand-shake (round 6): the one-launch pack with offsets far beyond its window, repeated for ~40 s while a
second stream runs whole forwards (other kernels' waves on the same CUs), results compared bit for bit with the first run.  A wave that
waits for the hand-shake counter depends on the other three waves of ITS workgroup only; this run is there to show it under load."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch  # noqa: E402
from emavfi import EMA_VFI, lib, synth  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B, H, W = 4, 360, 640
x = torch.randn(B, 67, H, W, generator=g).bfloat16().float().to(dev)
ow = ((torch.rand(27, 67, 3, 3, generator=g) * 2 - 1) * 0.2).bfloat16().float().to(dev)
dw = (torch.randn(67, 67, 3, 3, generator=g) / 24).bfloat16().float().to(dev)
db = torch.zeros(67, device=dev)
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(synth.synthetic_state_dict(seed=0, offset_std=3.0, offset_bias=3.0), strict=True)
f1, f2 = synth.fast_frames(5, 2, 360, 640, device=dev)
side = torch.cuda.Stream(device=dev)
refs = {}
t0, n = time.time(), 0
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 40.0:
    for spread in (1.0, 3.0, 8.0):
        ob = ((torch.rand(27, generator=torch.Generator().manual_seed(int(spread))) * 2 - 1) * spread).to(dev)
        with torch.cuda.stream(side), torch.no_grad():
            out_f = model(f1, f2)
        y = lib.mdcn(x, ow, ob, dw, db, dtype="bf16")
        torch.cuda.synchronize()
        key = spread
        if key not in refs:
            refs[key] = (y.clone(), out_f.clone())
        else:
            assert torch.equal(y, refs[key][0]) and torch.equal(out_f, refs[key][1]), f"run {n}: results changed"
        n += 1
    if n % 60 == 0:
        print(f"{n} launches, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {n} pack launches with a concurrent forward, bit-identical", flush=True)
