#!/usr/bin/env python3
"""Localise errors of the 16-bit deformable-conv kernel: per-row / per-column / per-channel error maps vs the fp32 kernel."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import lib
DEV = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for spread, H, W in ((0.0, 19, 41), (0.4, 19, 41), (2.0, 19, 41), (2.0, 40, 70), (12.0, 8, 32)):
    g = torch.Generator().manual_seed(67 * 7 + H)
    C = O = 67
    x = torch.randn(2, C, H, W, generator=g)
    off = torch.randn(2, 18, H, W, generator=g) * spread
    msk = torch.rand(2, 9, H, W, generator=g)
    w = torch.randn(O, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b = torch.randn(O, generator=g) * 0.1
    ref = lib.deform_conv2d(x.to(DEV), off.to(DEV), msk.to(DEV), w.to(DEV), b.to(DEV), dtype="fp32").cpu()
    got = lib.deform_conv2d(x.to(DEV), off.to(DEV), msk.to(DEV), w.to(DEV), b.to(DEV), dtype=dtype).cpu()
    e = (got - ref).abs()
    print(f"spread {spread} {H}x{W}: max err {e.max():.4f} (ref max {ref.abs().max():.3f}); bad (>0.03) {int((e > 0.03).sum())} of {e.numel()}")
    bad = (e > 0.03)
    if bad.any():
        print("  bad per batch:", bad.sum((1, 2, 3)).tolist())
        print("  bad per row  :", bad.sum((0, 1, 3)).tolist())
        print("  bad per col  :", bad.sum((0, 1, 2)).tolist())
        print("  bad per chan :", bad.sum((0, 2, 3)).tolist())
