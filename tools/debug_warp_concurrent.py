#!/usr/bin/env python3
"""Does the tiled warp kernel give wrong pixels while another stream keeps the CUs busy with other kernels?
usage: debug_warp_concurrent.py [outer=150] [warps_per_outer=40] [aggressors=alone,forward]    (EMAVFI_LIB selects a variant build)
aggressors: alone forward conv16 conv32 deform16 deform32 mm ew warp"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, lib, synth
DEV = "cuda:0"
outer = int(sys.argv[1]) if len(sys.argv) > 1 else 150
inner = int(sys.argv[2]) if len(sys.argv) > 2 else 40
aggr = (sys.argv[3] if len(sys.argv) > 3 else "alone,forward").split(",")
m = EMA_VFI(compute_dtype="bf16").to(DEV).eval()
m.load_state_dict(synth.synthetic_state_dict(seed=0))
xa = [t.to(DEV) for t in synth.synthetic_frames(51, 2, 192, 256, "natural")]
xb = [t.to(DEV) for t in synth.synthetic_frames(52, 1, 360, 640, "stress")]
with torch.no_grad():
    _, taps = m(*xb, return_taps=True)
    flow, f2 = taps["flow"].float().contiguous().clone(), xb[1].contiguous()
    ref = lib.warp(f2, flow).clone()
    m(*xa)
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(1)
    cx = torch.randn(2, 32, 192, 256, generator=g).to(DEV)
    cw, cb = (torch.randn(32, 32, 3, 3, generator=g) * 0.05).to(DEV), torch.zeros(32, device=DEV)
    dx = torch.randn(2, 19, 96, 128, generator=g).to(DEV)
    doff, dmk = torch.randn(2, 18, 96, 128, generator=g).to(DEV), torch.rand(2, 9, 96, 128, generator=g).to(DEV)
    dw, db = (torch.randn(16, 19, 3, 3, generator=g) * 0.05).to(DEV), torch.zeros(16, device=DEV)
    A = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    E = torch.randn(8 << 20, device=DEV)
    wf2, wflow = torch.randn(2, 3, 192, 256, device=DEV), torch.randn(2, 2, 192, 256, device=DEV) * 3
    work = {
        "forward": lambda: m(*xa),
        "conv16": lambda: [lib.conv3x3(cx, cw, cb, dtype="bf16") for _ in range(3)],
        "conv32": lambda: [lib.conv3x3(cx, cw, cb, dtype="fp32") for _ in range(3)],
        "deform16": lambda: [lib.deform_conv2d(dx, doff, dmk, dw, db, dtype="bf16") for _ in range(3)],
        "deform32": lambda: [lib.deform_conv2d(dx, doff, dmk, dw, db, dtype="fp32") for _ in range(3)],
        "mm": lambda: [A @ A for _ in range(4)],
        "ew": lambda: [torch.sin(E) for _ in range(6)],
        "warp": lambda: [lib.warp(wf2, wflow) for _ in range(40)],
    }
    for w in work.values():
        w()
    torch.cuda.synchronize()
    for label in aggr:
        busy = label != "alone"
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        nbad = torch.zeros((), dtype=torch.int64, device=DEV)
        nel = torch.zeros((), dtype=torch.int64, device=DEV)
        first = None
        for it in range(outer):
            if busy:
                with torch.cuda.stream(s1):
                    work[label]()
            with torch.cuda.stream(s2):
                for _ in range(inner):
                    o = lib.warp(f2, flow)
                    ne = (o != ref)
                    c = ne.sum()
                    nbad += (c > 0)
                    nel += c
            if it % 50 == 49:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        print(f"{label}: {int(nbad)} of {outer * inner} warp launches wrong ({int(nel)} elements)", flush=True)
