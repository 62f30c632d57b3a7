#!/usr/bin/env python3
"""Diagnostic: where a conv3x3_wreg_kernel wave (csrc/conv_wreg.inl) spends its time - s_memtime stamps at the seams.
Needs the stamped build:  make -C video-frame-interpolation_amd/csrc TAG=_stamps EXTRA=-DEMAVFI_DEFORM_STAMPS=1
  EMAVFI_LIB=video-frame-interpolation_amd/emavfi/lib/libemavfi_stamps.so python tools/wreg_stamps.py [bf16|fp16]
Runs emavfi_context alone at B = 8 x 720p (context_encoding.2 is the last stamped launch; its rows overwrite context_encoding.1's)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from emavfi import lib  # noqa: E402

dtype = "fp16" if "fp16" in sys.argv else "bf16"
ROWS = 16384
L = lib.load()
fn = L.emavfi_debug_deform_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
g = torch.Generator().manual_seed(0)
mid = 64
feat = torch.randn(8, mid, 720, 1280, generator=g).to("cuda:0")
params = []
for co, ci in ((2 * mid, mid), (4 * mid, 2 * mid), (4 * mid, 4 * mid)):
    params += [(torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).to("cuda:0"), (torch.randn(co, generator=g) * 0.1).to("cuda:0")]
params += [(torch.randn(mid, 4 * mid, generator=g) / 16).to("cuda:0"), torch.zeros(mid, device="cuda:0")]
S = 2 if "s2" in sys.argv else 1
if S == 2:   # context_encoding.1 alone, through the stage entry emavfi_conv3x3 (3600 workgroups as well)
    x = torch.randn(8, 2 * mid, 360, 640, generator=g).to("cuda:0")
    run = lambda: lib.conv3x3(x, params[2], params[3], stride=2, act=1, dtype=dtype)
else:
    run = lambda: lib.context(feat, params, dtype)
for _ in range(2):
    run()
assert fn(None, ROWS, 1) == 0
run()
buf = np.zeros((ROWS, 8), dtype=np.uint64)
assert fn(buf.ctypes.data, ROWS, 0) == 0
sel = buf[(buf[:, 4] == 1) & (buf[:, 6] == S)]
v = sel.astype(np.float64)
t0 = v[:, 0] - v[:, 0].min()
tot = v[:, 1] + v[:, 2] + v[:, 3]
print(f"{dtype}: {len(v)} waves of context_encoding.{3 - S} (s_memtime ticks = shader cycles)")
for i, n in ((1, "prologue (bias, DMA chunk 0, first weights, barrier)"), (2, "k loop (4 chunks x 36 steps x 8 MFMAs)"), (3, "epilogue")):
    print(f"  {n:55s} median {np.median(v[:, i]):8.0f}  p10 {np.percentile(v[:, i], 10):8.0f}  p90 {np.percentile(v[:, i], 90):8.0f}")
print(f"  wave total median {np.median(tot):.0f}; kernel span {t0.max() + tot[np.argmax(t0)]:.0f} ticks; start times: p50 {np.median(t0):.0f} max {t0.max():.0f}")
# occupancy picture: per CU (hw id), the sorted start times
hw = sel[:, 5]
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)) | (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(4)) | (((hw >> np.uint64(16)) & np.uint64(0xf)) << np.uint64(7))
ids, counts = np.unique(cu, return_counts=True)
print(f"  {len(ids)} distinct (cu, se, xcc-ish) ids; waves per id: min {counts.min()} median {np.median(counts):.0f} max {counts.max()}")
one = np.sort(t0[cu == ids[0]])
print("  start ticks of the waves of one CU:", " ".join(f"{x:.0f}" for x in one[:64]))
