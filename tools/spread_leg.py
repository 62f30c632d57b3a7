#!/usr/bin/env python3
"""bench.py's also_pack_vs_offset_spread leg alone: python tools/spread_leg.py > profiles/r05_pack_vs_offset_spread.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from emavfi import synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
res = bench.pack_vs_offset_spread(bench.Hip(), synth.synthetic_state_dict(seed=0), dev, 8, 720, 1280)
print(json.dumps(res, indent=1))
