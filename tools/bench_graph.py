"""Eager vs hipGraph replay of the forward on small, launch-bound configurations (20 launches per forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, synth

dev = "cuda:0"
sd = synth.synthetic_state_dict(seed=0)
for dt in ("bf16", "fp32"):
    m = EMA_VFI(compute_dtype=dt).to(dev).eval(); m.load_state_dict(sd)
    for B, H, W in ((1, 256, 256), (1, 720, 1280), (16, 256, 256), (8, 720, 1280)):
        f1, f2 = synth.fast_frames(5, B, H, W, device=dev)
        with torch.no_grad():
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3): m(f1, f2)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = m(f1, f2)
            def timeit(fn, n=50):
                for _ in range(5): fn()
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(n): fn()
                torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
            e = timeit(lambda: m(f1, f2)); r = timeit(g.replay)
        print(f"{dt} B={B} {W}x{H}: eager {e:.3f} ms, graph replay {r:.3f} ms ({B / r * 1e3:.0f} frames/s)")
