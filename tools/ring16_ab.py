#!/usr/bin/env python3
"""EXPERIMENT (round 5): the 64 -> 64 LDS-ring convolution on v_mfma_f32_16x16x32 (csrc/conv_ring16.inl, EMAVFI_CONV_RING16=1) against the
shipped 32x32x16 kernel (csrc/conv_ring.inl), both through the stage entry emavfi_conv3x3 at B = 8 x 720p.
  python tools/ring16_ab.py check          parity of the two kernels (and of both against ATen at a small size)
  python tools/ring16_ab.py run 0|1 [N]    N calls of one variant (wrap in `rocprofv3 --kernel-trace --stats` for the kernel's own time)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
import torch.nn.functional as F
from emavfi import lib

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(dev)
b = (torch.randn(64, generator=g) * 0.1).to(dev)


def conv(x, variant, dtype):
    os.environ["EMAVFI_CONV_RING16"] = "1" if variant else "0"
    return lib.conv3x3(x, w, b, act=lib.ACT_RELU, dtype=dtype)


if sys.argv[1] == "check":
    for dtype, ulp in (("bf16", 2.0 ** -8), ("fp16", 2.0 ** -11)):
        for shape in ((2, 64, 37, 150), (1, 64, 5, 64), (1, 64, 1, 9), (3, 64, 130, 131)):
            x = torch.randn(*shape, generator=g).to(dev)
            a, c = conv(x, 0, dtype), conv(x, 1, dtype)
            ref = F.relu(F.conv2d(x.to(torch.bfloat16 if dtype == "bf16" else torch.float16).float().cpu(),
                                  w.to(torch.bfloat16 if dtype == "bf16" else torch.float16).float().cpu(), b.cpu(), padding=1))
            d = (a - c).abs()
            scale = max(1.0, ref.abs().max().item())
            print(f"{dtype} {shape}: ring16 vs ring max {d.max().item():.3e} ({(d > 0).float().mean().item() * 100:.2f} % differ), "
                  f"ring16 vs ATen {(c.cpu() - ref).abs().max().item():.3e}, ring vs ATen {(a.cpu() - ref).abs().max().item():.3e} (storage step ~{ulp * scale:.1e})")
            assert d.max().item() <= 2 * ulp * scale and (c.cpu() - ref).abs().max().item() <= 2 * ulp * scale
    print("ok")
else:
    variant, n = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 20
    x = torch.randn(8, 64, 720, 1280, generator=g).to(dev)
    for _ in range(3):
        conv(x, variant, "bf16")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        conv(x, variant, "bf16")
    torch.cuda.synchronize()
    print(f"variant {variant}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call (incl. the NCHW <-> channels-last passes)")
