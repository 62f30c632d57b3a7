#!/usr/bin/env python3
"""Diagnostic (round 6): what the pack kernel's fix-up pass costs per flagged (wave, tap) at a given offset spread - in-kernel
s_memtime stamps of ONE ModulatedDeformConvPack (emavfi_mdcn on the forward's second block input, B = 8 x 720p, bf16, the offset
convolution rescaled exactly as bench.py's also_pack_vs_offset_spread does).  Needs the stamped build:
  make -C video-frame-interpolation_amd/csrc TAG=_stamps EXTRA=-DEMAVFI_DEFORM_STAMPS=1
  EMAVFI_LIB=video-frame-interpolation_amd/emavfi/lib/libemavfi_stamps.so python tools/fixup_stamps.py [spread ...]
Shares and per-item cycles only: the stamped build's fences forbid overlaps the product has."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from emavfi import EMA_VFI, lib, synth  # noqa: E402

spreads = [float(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8]
dev = torch.device("cuda", 0)
ROWS = 16384
L = lib.load()
fn = L.emavfi_debug_deform_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
sd = synth.synthetic_state_dict(seed=0)
B, H, W = 8, 720, 1280
model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
model.load_state_dict(sd, strict=True)
f1, f2 = synth.fast_frames(100, B, H, W, device=dev)
with torch.no_grad():
    _, taps = model(f1, f2, return_taps=True)
x = taps["fused_0"].clone()
del taps, f1, f2, model
torch.cuda.empty_cache()
ow, ob = sd["attention_blocks.1.offset_conv.weight"].to(dev), sd["attention_blocks.1.offset_conv.bias"].to(dev)
dw, db = sd["attention_blocks.1.dcn_v2.weight"].to(dev), sd["attention_blocks.1.dcn_v2.bias"].to(dev)
offch = torch.tensor(list(range(0, 9)) + list(range(18, 27)), device=dev)
sigma0 = lib.conv3x3(x[:1], ow, torch.zeros_like(ob), dtype="fp32")[:, offch].std().item()
for s_px in spreads:
    ow_s, ob_s = ow.clone(), ob.clone()
    ow_s[offch] *= (0.5 * s_px / sigma0)
    ob_s[offch] *= 0.5 * s_px
    lib.mdcn(x, ow_s, ob_s, dw, db, dtype="bf16")
    assert fn(None, ROWS, 1) == 0
    lib.mdcn(x, ow_s, ob_s, dw, db, dtype="bf16")
    buf = np.zeros((ROWS, 8), dtype=np.uint64)
    assert fn(buf.ctypes.data, ROWS, 0) == 0
    sel = buf[buf[:, 6] == 1]
    taps_out = ((sel[:, 4] >> np.uint64(32)) & np.uint64(0xff)).astype(np.float64)
    parked = (sel[:, 4] >> np.uint64(40)).astype(np.float64)
    fix = (sel[:, 2] >> np.uint64(32)).astype(np.float64)
    M32 = np.uint64(0xffffffff)
    total = (sel[:, 5] & M32).astype(np.float64)
    steps = (sel[:, 3] & M32).astype(np.float64)
    fx_wait, fx_issue = (sel[:, 0] >> np.uint64(32)).astype(np.float64), (sel[:, 1] >> np.uint64(32)).astype(np.float64)
    fx_land, fx_taps = (sel[:, 3] >> np.uint64(32)).astype(np.float64), (sel[:, 5] >> np.uint64(32)).astype(np.float64)
    any_ = taps_out > 0
    rounds = np.ceil(parked / 31.0)
    print(f"spread {s_px:g} px: {len(sel)} waves; tile {np.median(total):.0f} cycles (mean {total.mean():.0f}), main-loop tap {np.median(steps) / 9:.0f}; "
          f"flagged taps per wave {taps_out.mean():.3f} of 9 ({100 * taps_out.mean() / 9:.2f} %), waves with any {100 * any_.mean():.1f} %, "
          f"parked samples per flagged wave {parked[any_].mean() if any_.any() else 0:.1f} ({rounds[any_].mean() if any_.any() else 0:.2f} rounds); "
          f"fix-up {fix.mean():.0f} cycles per wave = {100 * fix.mean() / total.mean():.1f} % of the tile, "
          f"{fix[any_].mean() if any_.any() else 0:.0f} per wave that has any, {fix.sum() / max(1.0, taps_out.sum()):.0f} per flagged (wave, tap)")
    if any_.any():
        nr, nt = rounds[any_].sum(), taps_out[any_].sum()
        print(f"    per wave that has any: hand-shake wait {fx_wait[any_].mean():.0f}; descriptors + DMA issue {fx_issue[any_].mean():.0f} ({fx_issue[any_].sum() / nr:.0f} per round); "
              f"landing + conversion {fx_land[any_].mean():.0f} ({fx_land[any_].sum() / nr:.0f} per round); taps {fx_taps[any_].mean():.0f} ({fx_taps[any_].sum() / nt:.0f} per flagged tap "
              f"incl. round overhead)")
