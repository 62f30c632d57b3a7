#!/usr/bin/env python3
"""fp32x3 against exact fp32 at B = 8 x 720p: frames/s and the per-kernel table (HIP events around every launch)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
import bench
from emavfi import synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = bench.Hip()
sd = synth.synthetic_state_dict(seed=0)
for mode in ("fp32x3", "fp32"):
    r = bench.profiled_mode(hip, sd, dev, mode, 8, 720, 1280, 5)
    print(mode, r["value"], "frames/s", r["ms_per_step"], "ms")
    for k in r["kernels"][:14]:
        print("   ", k)
