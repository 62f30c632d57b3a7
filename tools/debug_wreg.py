import sys, math, torch
sys.path.insert(0, "video-frame-interpolation_amd")
from emavfi import lib
import torch.nn.functional as F
for (Cin, Cout, H, W, stride) in ((128, 256, 18, 34, 2), (256, 256, 9, 21, 1), (128, 256, 37, 131, 2)):
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(2, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    got = lib.conv3x3(x.cuda(), w.cuda(), b.cuda(), stride=stride, act=1, dtype="bf16").cpu()
    ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=1))
    err = (got - ref).abs()
    bad = err > 0.05 * ref.abs().max()
    print("case", Cin, Cout, H, W, stride, "bad frac", bad.float().mean().item(), "max err", err.max().item())
    print(" by row:", [round(bad[:, :, y].float().mean().item(), 3) for y in range(got.shape[2])])
    print(" by col (first 40):", [round(bad[:, :, :, xx].float().mean().item(), 2) for xx in range(min(40, got.shape[3]))])
    print(" by channel block of 32:", [round(bad[:, c:c + 32].float().mean().item(), 3) for c in range(0, 256, 32)])
    print(" by sample:", [round(bad[n].float().mean().item(), 3) for n in range(2)])
