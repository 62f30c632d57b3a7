#!/usr/bin/env python3
"""tools/profile_fp32.sh output -> profiles/<tag>_fp32_{720,256}_kernel_stats.csv, <tag>_fp32_{720,256}_pmc_traffic_detail.json and the
stamped per-mode traffic files bench.py's `roofline_fp32*` objects read:  python tools/make_profiles_fp32.py gpurun_out/r4prof32 r04"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")
for size, name in (("720", "traffic_fp32_8x720x1280.json"), ("256", "traffic_fp32_16x256x256.json")):
    one = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)   # noqa: E731  (the newest: gpurun merges into gpurun_out/)
    shutil.copy(one(f"stats_{size}/*/*_kernel_stats.csv"), os.path.join(prof, f"{tag}_fp32_{size}_kernel_stats.csv"))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), one(f"pmc_fetch_{size}/*/*_counter_collection.csv"),
           one(f"pmc_write_{size}/*/*_counter_collection.csv"), prof + os.sep, f"{tag}_fp32_{size}", f"--install-as={name}"]
    print(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout)
