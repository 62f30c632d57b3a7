#!/usr/bin/env python3
"""Where a wave of the LDS-ring convolution kernels spends a row step: s_memtime stamps at the seams.
Needs the stamped build:  make -C video-frame-interpolation_amd/csrc TAG=_stamps EXTRA=-DEMAVFI_DEFORM_STAMPS=1
  EMAVFI_LIB=video-frame-interpolation_amd/emavfi/lib/libemavfi_stamps.so python tools/ring_stamps.py [kind ...]
kind: ring (motion_estimation.0), tail (reconstruction.0), head (motion_estimation.1 + .2), ringtail (reconstruction.1 + .2),
ringfirst (cat + feat_ext_conv1 + conv_block_0).  One process per kind (the selector is read by the diagnostic build per launch)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
SEGS = {
    "ring": ["W  counted wait + barrier", "I  DMA issue + stores of row y-1", "M  bias + 36 MFMAs (operands 4 ahead)", "E  epilogue (ReLU, round, staging writes)", "-"],
    "tail": ["W  counted wait + barrier", "I  DMA issue + stores of row y-1", "M  bias + 3 tail + 36 MFMAs", "E  epilogue", "-"],
    "head": ["W  counted wait + barrier", "I  DMA issue", "M  bias + 36 MFMAs", "H+E  head's 6 reads + 6 MFMAs behind the contraction, main epilogue -> row ring in their shadow, head epilogue + store", "-"],
    "ringtail": ["W  counted wait + barrier", "I  DMA issue", "A  stage A's contraction (36 MFMAs 16x16x32 on four chains, 18 reads)", "H+E  head's 3 reads + 3 MFMAs behind it, stage A's epilogue -> row ring", "O  head epilogue (exp / rcp) + store"],
    "ringfirst": ["W  barrier", "I  frame loads + stores of row t-3", "B  stage B's contraction (36 MFMAs)", "A+E  stage A's 5 MFMAs behind it, B's epilogue in their shadow, A's epilogue -> row ring", "P  frame_put"],
}
KIND = {"ring": 10, "tail": 11, "head": 12, "ringtail": 14, "ringfirst": 15}


def one(kind):
    import numpy as np
    import torch
    from emavfi import EMA_VFI, lib, synth
    L = lib.load()
    fn = L.emavfi_debug_deform_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    dev = "cuda:0"
    m = EMA_VFI(compute_dtype="bf16").to(dev).eval()
    m.load_state_dict(synth.synthetic_state_dict(seed=0))
    a, b = synth.fast_frames(100, 8, 720, 1280, device=dev)
    ROWS = 16384
    with torch.no_grad():
        for _ in range(3):
            m(a, b)
        assert fn(None, ROWS, 1) == 0
        m(a, b)
    buf = np.zeros((ROWS, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, ROWS, 0) == 0
    v = buf[buf[:, 6] == KIND[kind]].astype(np.float64)
    if not len(v):
        print(f"{kind}: no stamped rows (is this the stamped build, and does the plan run this kernel?)")
        return
    steps = v[:, 7]
    per = v[:, :5] / steps[:, None]
    tot = v[:, 5]
    step_total = per[:, :4].sum(1) if kind in ("ring", "tail", "head") else per.sum(1)   # (head: segment 4 is a part of segment 1)
    print(f"== {kind}: {len(v)} waves, {np.median(steps):.0f} row steps per wave, kernel {np.median(tot):.0f} cycles per wave (s_memtime ticks = 100 MHz x? -> shares matter)")
    for i, name in enumerate(SEGS[kind]):
        if name == "-":
            continue
        print(f"   {name:52s} median {np.median(per[:, i]):7.1f}  p10 {np.percentile(per[:, i], 10):7.1f}  p90 {np.percentile(per[:, i], 90):7.1f}  "
              f"{100 * np.median(per[:, i]) / np.median(step_total):5.1f} % of a step")
    print(f"   step total (sum of segments) median {np.median(step_total):.1f}; steps x that = {100 * np.median(steps * step_total) / np.median(tot):.1f} % of the wave's kernel time")
    for w in range(4):   # per wave id (row = workgroup * 4 + wave)
        sel = (np.nonzero(buf[:, 6] == KIND[kind])[0] % 4) == w
        print(f"   wave {w}: " + "  ".join(f"{np.median(per[sel, i]):7.1f}" for i in range(5)))


if __name__ == "__main__":
    kinds = sys.argv[1:] or list(SEGS)
    if len(kinds) == 1 and os.environ.get("EMAVFI_STAMP_RING") == kinds[0]:
        one(kinds[0])
    else:
        for k in kinds:
            subprocess.run([sys.executable, os.path.abspath(__file__), k], env=dict(os.environ, EMAVFI_STAMP_RING=k), check=False)
