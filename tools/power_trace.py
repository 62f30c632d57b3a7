#!/usr/bin/env python3
"""Board power and clocks while the forward runs back to back (is the step power-limited?).
usage: power_trace.py [dtype=bf16] [seconds=6] [EMAVFI_LIB selects a variant]
Polls `rocm-smi` (socket power, sclk, mclk, power cap) about every 0.15 s from a thread while the main thread enqueues
B=8 x 720p forwards; prints idle readings first, then min / median / max under load and the forward's rate."""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch
from emavfi import EMA_VFI, synth

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        def num(keys, cast=float):
            for k, v in c.items():
                if all(s in k.lower() for s in keys):
                    try:
                        return cast(str(v).replace("Mhz", "").replace("(", "").replace(")", "").strip())
                    except ValueError:
                        pass
            return None
        return {"power_w": num(("power", "socket")) or num(("average", "power")) or num(("current", "power")), "cap_w": num(("max", "power")),
                "sclk_mhz": num(("sclk", "clock")), "mclk_mhz": num(("mclk", "clock")), "raw": c}
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)}

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
dev = "cuda:0"
m = EMA_VFI(compute_dtype=dtype).to(dev).eval()
m.load_state_dict(synth.synthetic_state_dict(seed=0))
f1, f2 = (t.to(dev) for t in synth.synthetic_frames(1, 8, 720, 1280, "natural"))
with torch.no_grad():
    m(f1, f2); torch.cuda.synchronize()
idle = smi()
print("idle:", {k: v for k, v in idle.items() if k != "raw"})
if "raw" in idle:
    print("fields:", {k: v for k, v in idle["raw"].items()})
samples, stop = [], False
def poll():
    while not stop:
        s = smi(); s["t"] = time.time(); samples.append(s); time.sleep(0.1)
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
with torch.no_grad():
    while time.time() - t0 < secs:
        for _ in range(20):
            m(f1, f2)
        torch.cuda.synchronize(); n += 20
dt = time.time() - t0
stop = True; th.join()
print(f"{dtype}: {n} forwards in {dt:.2f} s = {dt / n * 1e3:.2f} ms per step, {n * 8 / dt:.1f} frames/s")
load = [s for s in samples if "error" not in s and s["t"] > t0 + 1.0 and s["t"] < t0 + dt]
for key in ("power_w", "sclk_mhz", "mclk_mhz", "cap_w"):
    v = sorted(s[key] for s in load if s.get(key) is not None)
    if v:
        print(f"  {key}: min {v[0]:.0f}  median {v[len(v) // 2]:.0f}  max {v[-1]:.0f}   ({len(v)} samples under load)")
