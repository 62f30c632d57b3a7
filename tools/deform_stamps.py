#!/usr/bin/env python3
"""Diagnostic: where a fused deformable-conv pack spends its wave cycles (in-kernel s_memtime stamps).
Needs the stamped build:  make -C video-frame-interpolation_amd/csrc TAG=_stamps EXTRA=-DEMAVFI_DEFORM_STAMPS=1
  EMAVFI_LIB=video-frame-interpolation_amd/emavfi/lib/libemavfi_stamps.so python tools/deform_stamps.py [bf16|fp16]
Shares only - a stamped build's run time is not the product's (its fences forbid overlaps)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch  # noqa: E402
from emavfi import EMA_VFI, lib, synth  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import numpy as np  # noqa: E402

ROWS = 16384
L = lib.load()
fn = L.emavfi_debug_deform_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
m = EMA_VFI(compute_dtype=dtype).to("cuda:0").eval()
m.load_state_dict(synth.synthetic_state_dict(seed=0))
f1, f2 = synth.fast_frames(100, 8, 720, 1280, device="cuda:0")
buf = np.zeros((ROWS, 8), dtype=np.uint64)
with torch.no_grad():
    for _ in range(2):
        m(f1, f2)
    assert fn(None, ROWS, 1) == 0
    m(f1, f2)   # each pack overwrites the rows: what is read back is the LAST pack of this forward
assert fn(buf.ctypes.data, ROWS, 0) == 0
sel = buf[buf[:, 6] == 1]
n_out = ((sel[:, 4] >> np.uint64(32)) & np.uint64(0xff)).astype(np.float64)      # taps of the wave with a lane outside the window, of 9
n_lanes = (sel[:, 4] >> np.uint64(40)).astype(np.float64)                          # samples outside, of 9 * 64
fixup = (sel[:, 2] >> np.uint64(32)).astype(np.float64)                            # cycles in the fix-up pass (round 6)
for col in (0, 1, 2, 3, 4, 5):   # (bits 32..63 carry the fix-up's detail: tools/fixup_stamps.py)
    sel[:, col] &= np.uint64(0xffffffff)
v = sel.astype(np.float64)
names = ["prologue (window DMA + barrier)", "offset_conv", "geometry + tail of all 9 taps (up front)",
         "gather + blend + MFMA steps (9 taps)", "epilogue stores (drained)", "total"]
print(f"{dtype}: {len(v)} waves sampled (median cycles per wave and tile)")
tot = np.median(v[:, 5])
for i, n in enumerate(names):
    print(f"  {n:42s} {np.median(v[:, i]):10.0f}  {100.0 * np.median(v[:, i]) / tot:5.1f} %   (p10 {np.percentile(v[:, i], 10):.0f}, p90 {np.percentile(v[:, i], 90):.0f})")
print(f"  fix-up: {n_out.mean():.3f} of 9 taps per wave have a lane outside the window ({100 * n_out.mean() / 9:.2f} %), "
      f"{n_lanes.mean():.2f} of 576 samples ({100 * n_lanes.mean() / 576:.3f} %); {fixup.mean():.0f} cycles per wave on average, "
      f"{fixup[n_out > 0].mean() if (n_out > 0).any() else 0:.0f} per wave that has any, "
      f"{fixup.sum() / max(1.0, n_out.sum()):.0f} per flagged (wave, tap)")
d = buf[buf[:, 6] == 1][:, 7]
parts = [((d >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(np.float64) * 4 for i in range(4)]
print("  prologue detail (median cycles): tile mapping + small loads + DMA issue %.0f, DMA landed after %.0f, convert %.0f, barrier wait %.0f"
      % tuple(np.median(x) for x in parts))
print(f"  per tap: steps {np.median(v[:, 3]) / 9:.0f} cycles")
