#!/usr/bin/env python3
"""Phase sums of the 64->64 ping-pong kernel (diagnostic build: make -C .../csrc TAG=_cstamps EXTRA=-DEMAVFI_CONV_STAMPS=1).
usage: EMAVFI_LIB=.../libemavfi_cstamps.so python tools/conv_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import lib
L = lib.load()
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 64, 720, 1280, generator=g).cuda()
w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()
b = torch.zeros(64).cuda()
lib.conv3x3(x, w, b, dtype="bf16")
out = (ctypes.c_ulonglong * 8)()
assert L.emavfi_debug_conv_stamps(out, 1) == 0
for _ in range(5):
    lib.conv3x3(x, w, b, dtype="bf16")
assert L.emavfi_debug_conv_stamps(out, 0) == 0
c, s, gg, wt, slots, waves = [out[i] for i in range(6)]
tot = c + s + gg + wt
print(f"waves {waves}, slots per wave {slots / waves:.1f}; per slot and wave (10 ns ticks of s_memtime at 100 MHz): "
      f"contract {c / slots * 1:.1f}, store {s / slots:.1f}, stage issue {gg / slots:.1f}, wait+barrier {wt / slots:.1f}, sum {tot / slots:.1f}")
print(f"shares: contract {c / tot:.1%}, store {s / tot:.1%}, stage issue {gg / tot:.1%}, wait at barrier {wt / tot:.1%}")
