"""Run the bf16 and fp16 forward N times at B=8 x 720p (3 fused deform packs each) and count outputs that
differ from the first run; also the standalone deform op at 720p in both dtypes."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, synth, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
sd = synth.synthetic_state_dict(seed=0)
f1, f2 = synth.fast_frames(3, 8, 720, 1280, device=dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 67, 720, 1280, generator=g).to(dev)
off = (torch.randn(1, 18, 720, 1280, generator=g) * 2).to(dev)
msk = torch.rand(1, 9, 720, 1280, generator=g).to(dev)
w = (torch.randn(67, 67, 3, 3, generator=g) / math.sqrt(603)).to(dev)
b = torch.randn(67, generator=g).to(dev)
for dt in ("bf16", "fp16"):
    m = EMA_VFI(compute_dtype=dt).to(dev).eval(); m.load_state_dict(sd)
    bad = 0
    with torch.no_grad():
        ref = m(f1, f2).clone()
        for i in range(n):
            out = m(f1, f2)
            bad += int((out != ref).sum())
    r0 = lib.deform_conv2d(x, off, msk, w, b, dtype=dt).clone()
    bad2 = 0
    for i in range(n):
        bad2 += int((lib.deform_conv2d(x, off, msk, w, b, dtype=dt) != r0).sum())
    torch.cuda.synchronize()
    print(f"   determinism {dt}: forward x{n} (B=8): {bad} differing elements; deform op x{n}: {bad2} differing elements")
