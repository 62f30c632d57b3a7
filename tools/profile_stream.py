"""Where the host time of the streaming harness goes (cProfile over one 64-pair run)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np, torch
from emavfi import EMA_VFI, FrameInterpolator, synth
dev = "cuda:0"
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(synth.synthetic_state_dict(seed=0))
f1, f2 = synth.synthetic_frames_u8(3, 1, 720, 1280, "natural")
frames = [np.roll(f1[0], 3 * i, axis=1) for i in range(65)]
fi = FrameInterpolator(model, interpolation_factor=1, batch_pairs=8, reference_quirks=False)
sum(1 for _ in fi.run(frames[:17]))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); n = sum(1 for _ in fi.run(frames)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print(f"{dt*1e3:.1f} ms total")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
