#!/usr/bin/env python3
"""Board power and shader clock under ONE kernel of the step at a time (round 5: which kernels are at the power cap on their own?).
Needs the diagnostic build (make TAG=_stamps EXTRA=-DEMAVFI_DEFORM_STAMPS=1): its EMAVFI_DEBUG_REPEAT_PACK / _CONV = n repeat the
stage's kernel n times per call, so that the stage entry's layout conversions (HBM-bound, light) are < 2 % of the loop.
usage: EMAVFI_LIB=.../libemavfi_stamps.so power_per_kernel.py [seconds=4] [repeat=200]
Runs, back to back for `seconds` each and with rocm-smi polled from a thread (as tools/power_trace.py does for the whole forward):
  pack      one ModulatedDeformConvPack (emavfi_mdcn = the forward's attention block on the second block's real input, B = 8 x 720p, bf16)
  conv64    one 64 -> 64 3x3 convolution + ReLU (emavfi_conv3x3 -> conv3x3_ring_kernel, same size, bf16)
  forward   the whole forward
and prints per leg: launches/s, the kernel's time from the loop, median power and sclk."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch  # noqa: E402
from emavfi import EMA_VFI, lib, synth  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        c = json.loads(out)
        c = c[sorted(c)[0]]

        def num(*keys):
            for k, v in c.items():
                if all(x in k.lower() for x in keys):
                    try:
                        return float(str(v).replace("Mhz", "").strip("() "))
                    except ValueError:
                        pass
            return None
        return {"power_w": num("socket", "power") or num("average", "power"), "cap_w": num("max", "power"), "sclk_mhz": num("sclk", "speed")}
    except Exception:  # noqa: BLE001
        return None


def leg(name, fn, secs, per_call):
    fn(); torch.cuda.synchronize()
    samples, stop = [], []

    def poll():
        while not stop:
            s = smi()
            if s:
                s["t"] = time.time(); samples.append(s)
            time.sleep(0.1)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(per_call):
            fn()
        torch.cuda.synchronize(); n += per_call
    dt = time.time() - t0
    stop.append(1); th.join()
    load = [s for s in samples if t0 + 1.0 < s["t"] < t0 + dt]
    med = lambda k: sorted(s[k] for s in load if s.get(k) is not None)[len(load) // 2] if load else None   # noqa: E731
    print(f"{name:12s} {n:6d} calls in {dt:5.2f} s = {dt / n * 1e6:8.1f} us per call   power {med('power_w')} W of {med('cap_w')}   sclk {med('sclk_mhz')} MHz   ({len(load)} samples)", flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, H, W = 8, 720, 1280
    sd = synth.synthetic_state_dict(seed=0)
    model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
    model.load_state_dict(sd, strict=True)
    f1, f2 = synth.fast_frames(100, B, H, W, device=dev)
    with torch.no_grad():
        _, taps = model(f1, f2, return_taps=True)
    x = taps["fused_0"].clone()
    feat = x[:, :64].contiguous()
    del taps
    torch.cuda.empty_cache()
    g = lambda k: sd[k].to(dev)   # noqa: E731
    ow, ob = g("attention_blocks.1.offset_conv.weight"), g("attention_blocks.1.offset_conv.bias")
    dw, db = g("attention_blocks.1.dcn_v2.weight"), g("attention_blocks.1.dcn_v2.bias")
    cw, cb = g("feat_ext_blocks.1.0.weight") if "feat_ext_blocks.1.0.weight" in sd else None, None
    if cw is None:
        key = next(k for k in sd if k.endswith(".weight") and tuple(sd[k].shape) == (64, 64, 3, 3))
        cw, cb = g(key), g(key[:-6] + "bias")
    else:
        cb = g("feat_ext_blocks.1.0.bias")
    print("idle:", smi(), flush=True)
    with torch.no_grad():
        leg("forward", lambda: model(f1, f2), secs, 10)
        os.environ["EMAVFI_DEBUG_REPEAT_PACK"] = str(rep)
        leg(f"pack x{rep}", lambda: lib.mdcn(x, ow, ob, dw, db, dtype="bf16"), secs, 1)
        del os.environ["EMAVFI_DEBUG_REPEAT_PACK"]
        os.environ["EMAVFI_DEBUG_REPEAT_CONV"] = str(rep)
        leg(f"conv64 x{rep}", lambda: lib.conv3x3(feat, cw, cb, dtype="bf16", act=lib.ACT_RELU), secs, 1)
        del os.environ["EMAVFI_DEBUG_REPEAT_CONV"]


if __name__ == "__main__":
    main()
