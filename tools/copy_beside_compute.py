#!/usr/bin/env python3
"""How long do the harness's PCIe legs take BESIDE back-to-back forwards (round 5)?  For nine / eight 720p uint8 frames:
  (a) the zero-copy kernels (emavfi_preprocess_u8 / _postprocess_u8 reading / writing pinned host memory) on a high-priority side stream,
  (b) hipMemcpyAsync pinned <-> device on that stream (SDMA engines: no CU involved) + the same kernels on device-resident bytes,
each alone and while the caller's stream runs B = 8 x 720p bf16 forwards, and what each does to the forward's own rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np, torch
from emavfi import EMA_VFI, lib, synth
dev = torch.device("cuda:0")
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(synth.synthetic_state_dict(seed=0))
a, b = synth.fast_frames(100, 8, 720, 1280, device=dev)
h_in = torch.randint(0, 256, (9, 720, 1280, 3), dtype=torch.uint8).pin_memory()
h_out = torch.empty(8, 720, 1280, 3, dtype=torch.uint8).pin_memory()
d_in = torch.empty(9, 720, 1280, 3, dtype=torch.uint8, device=dev)
d_out = torch.empty(8, 720, 1280, 3, dtype=torch.uint8, device=dev)
x = torch.empty(9, 3, 720, 1280, device=dev)
y = torch.rand(8, 3, 720, 1280, device=dev)
side = torch.cuda.Stream(priority=-1)


def legs():
    return {
        "pre  zero-copy kernel": lambda: lib.preprocess_u8(h_in, device=dev, out=x),
        "post zero-copy kernel": lambda: lib.postprocess_u8(y, denormalize=True, out=h_out),
        "pre  memcpy H2D + device kernel": lambda: (d_in.copy_(h_in, non_blocking=True), lib.preprocess_u8(d_in, out=x)),
        "post device kernel + memcpy D2H": lambda: (lib.postprocess_u8(y, denormalize=True, out=d_out), h_out.copy_(d_out, non_blocking=True)),
    }


def run(busy, leg=None, n=12):
    """forwards/s on the main stream, and the leg's median latency on the side stream (enqueued once per forward)"""
    evs = []
    with torch.no_grad():
        for _ in range(3):
            model(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if busy:
                model(a, b)
            if leg is not None:
                with torch.cuda.stream(side):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side); leg(); e1.record(side)
                    evs.append((e0, e1))
            if not busy:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return (8 * n / dt if busy else None), (lat[len(lat) // 2] if lat else None)


base, _ = run(True)
print(f"forwards alone: {base:.1f} frames/s")
for name, leg in legs().items():
    _, alone = run(False, leg)
    fps, beside = run(True, leg)
    print(f"{name:34s}: alone {alone:7.3f} ms | beside forwards {beside:7.3f} ms, forwards {fps:.1f} frames/s ({100 * fps / base:.1f} %)")
