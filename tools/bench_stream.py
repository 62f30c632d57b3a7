"""PCIe-inclusive throughput of the streaming harness: uint8 720p host frames in, uint8 host frames out."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np, torch
from emavfi import EMA_VFI, FrameInterpolator, synth
dev = "cuda:0"
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(synth.synthetic_state_dict(seed=0))
f1, f2 = synth.synthetic_frames_u8(3, 1, 720, 1280, "natural")
frames = [np.roll(f1[0], 3 * i, axis=1) for i in range(65)]       # 64 pairs
for quirks, copy in ((True, True), (False, True), (False, False)):
    fi = FrameInterpolator(model, interpolation_factor=1, batch_pairs=8, reference_quirks=quirks, copy_out=copy)
    n = sum(1 for _ in fi.run(frames[:17]))                         # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = sum(1 for _ in fi.run(frames))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"reference_quirks={quirks} copy_out={copy}: {len(frames)-1} pairs -> {n} frames out in {dt*1e3:.1f} ms = "
          f"{(len(frames)-1)/dt:.1f} interpolated frames/s (host uint8 in/out, PCIe included, batch 8)")
