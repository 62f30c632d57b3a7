#!/usr/bin/env python3
"""Per-kernel table (HIP events around every launch) of one arithmetic mode at B = 8 x 720p: python tools/mode_kernels.py amp16|fp32|bf16|fp16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
import bench
from emavfi import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "amp16"
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
bench.PEAK.setdefault(mode, 2500.0)
r = bench.profiled_mode(bench.Hip(), synth.synthetic_state_dict(seed=0), dev, mode, 8, 720, 1280, 4)
print(f"{mode}: {r['value']} frames/s, {r['ms_per_step']} ms per step, sum of kernels {r['device_ms_per_step_sum_of_kernels']} ms")
import ctypes
from emavfi import lib
launches = lib.forward_launches(3, 64, 3, 8, 720, 1280, mode)
for k in r["kernels"]:
    print(f"  {k['kernel']:44s} x{k['launches_per_step']}  {k['avg_us']:9.1f} us  {100*k['share']:5.1f} %  {k['tflops']:8.1f} TFLOP/s  {k['gbs']:8.1f} GB/s")
