#!/usr/bin/env python3
"""The streaming harness against the resident forward, same process, same box (also_stream_pcie >= 0.95 x value).
usage: stream_breakdown.py [pairs ...]   (default 64 256)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np, torch
from emavfi import EMA_VFI, FrameInterpolator, synth
dev = "cuda:0"
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(synth.synthetic_state_dict(seed=0))
a, b = synth.fast_frames(100, 8, 720, 1280, device=dev)
with torch.no_grad():
    for _ in range(5):
        model(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        model(a, b)
    torch.cuda.synchronize(); res = 160 / (time.perf_counter() - t0)
print(f"resident: {res:.1f} frames/s ({8e3 / res:.3f} ms per batch of 8)")
f1, _ = synth.synthetic_frames_u8(3, 1, 720, 1280, "natural")
for pairs in [int(v) for v in sys.argv[1:]] or [64, 256]:
    frames = [np.roll(f1[0], 3 * (i % 97), axis=1) for i in range(pairs + 1)]
    for quirks, zc in ((True, False), (False, False), (True, True)):
        fi = FrameInterpolator(model, interpolation_factor=1, batch_pairs=8, reference_quirks=quirks, copy_out=False, zero_copy=zc)
        sum(1 for _ in fi.run(frames[:17]))
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = sum(1 for _ in fi.run(frames))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = max(best, pairs / dt)
        print(f"harness, {pairs} pairs, reference_quirks={quirks} zero_copy={zc}: {best:.1f} interpolated frames/s = {100 * best / res:.1f} % of resident ({n} frames out)")
