#!/usr/bin/env python3
"""VERDICT r5 item 3, the timing half: what ONE 64 -> 64 layer costs as an exact fp32 contraction and as a three-term 16-bit split
(x_hi w_hi + x_lo w_hi + x_hi w_lo = one f16 convolution over 3 x 64 input channels), on the kernels that exist: run under
  rocprofv3 --kernel-trace --stats -d gpurun_out/fp32x3 -- python3 tools/fp32x3_layer_timing.py
and read the conv3x3_kernel rows (the stage entry's layout / packing kernels are separate rows).  B = 8 x 1280x720.
The split here runs the generic tile kernel (weights through LDS, 3 chunks); the fp32 layer the same kernel family in fp32."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch  # noqa: E402
from emavfi import lib  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
B, H, W = 8, 720, 1280
x = torch.randn(B, 64, H, W, generator=g).to(dev)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
b = torch.zeros(64, device=dev)
for _ in range(3):
    y32 = lib.conv3x3(x, w, b, dtype="fp32")
xh = x.half().float()
xl = (x - xh).half().float()
wh = w.half().float()
wl = (w - wh).half().float()
x3 = torch.cat([xh, xl, xh], 1)
w3 = torch.cat([wh, wh, wl], 1)
del xh, xl
for _ in range(3):
    y3 = lib.conv3x3(x3, w3, b, dtype="fp16")     # f16 OUTPUT (the existing epilogue): timing only
for _ in range(3):
    y16 = lib.conv3x3(x, w, b, dtype="fp16")
torch.cuda.synchronize()
print("fp32 vs f16 (one term):", (y32 - y16).abs().max().item(), " fp32 vs f16x3 (f16-rounded output):", (y32 - y3).abs().max().item(), " |y| max", y32.abs().max().item())
