import os, sys, time
sys.path.insert(0, "video-frame-interpolation_amd")
import torch
from emavfi import EMA_VFI, synth
dev = "cuda:0"
sd = synth.synthetic_state_dict(seed=0)
for mode in ("amp16", "fp32"):
    m = EMA_VFI(compute_dtype=mode).to(dev).eval(); m.load_state_dict(sd)
    a, b = synth.fast_frames(300, 8, 720, 1280, device=dev)
    with torch.no_grad():
        for _ in range(2): m(a, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): m(a, b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"{mode}: {dt*1e3:.2f} ms per B=8 step = {8/dt:.1f} frames/s")
