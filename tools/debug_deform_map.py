"""Diagnostic: where (tile, row-in-tile, col, channel) do two bf16 deform runs differ?"""
import sys, os, math, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import lib
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
H, W = 720, 1280
x = torch.randn(1, 67, H, W, generator=g).to(dev)
off = (torch.randn(1, 18, H, W, generator=g) * 2).to(dev)
msk = torch.rand(1, 9, H, W, generator=g).to(dev)
w = (torch.randn(67, 67, 3, 3, generator=g) / math.sqrt(67 * 9)).to(dev)
b = torch.randn(67, generator=g).to(dev)
ref = lib.deform_conv2d(x, off, msk, w, b, dtype="fp32")
runs = [lib.deform_conv2d(x, off, msk, w, b, dtype="bf16").clone() for _ in range(4)]
torch.cuda.synchronize()
for i, r in enumerate(runs):
    bad = ((r - ref).abs() > 0.25)
    print(f"run {i}: elements >0.25 from fp32 result: {int(bad.sum())}; max err {(r-ref).abs().max().item():.3f}")
    if bad.any():
        idx = bad.nonzero().cpu()
        c, y, xx = idx[:, 1], idx[:, 2], idx[:, 3]
        tiles = collections.Counter(zip((y // 8).tolist(), (xx // 32).tolist()))
        print("   tiles hit:", len(tiles), "most common:", tiles.most_common(5))
        print("   row-in-tile hist:", torch.bincount(y % 8, minlength=8).tolist())
        print("   col-in-tile hist:", torch.bincount(xx % 32, minlength=32).tolist())
        print("   channel hist:", torch.bincount(c, minlength=67).tolist())
        pix = collections.Counter(zip(y.tolist(), xx.tolist()))
        print("   distinct pixels:", len(pix), "channels per bad pixel (top):", pix.most_common(3))
