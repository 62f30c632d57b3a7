#!/usr/bin/env python3
"""Turn one tools/profile_round.sh output directory into the tracked artefacts under profiles/:

  python tools/make_profiles.py gpurun_out/r02/prof_x r02_final [--install]

  profiles/<tag>_bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `bench.py --steps 30 --warmup 2 --no-extras`
  profiles/<tag>_pmc_traffic_detail.json     HBM bytes per launch per kernel (FETCH_SIZE x2 + WRITE_SIZE, separate passes)
  profiles/<tag>_sq_counters.json            SQ counters (two passes) of the three heaviest kernels, with derived shares
  profiles/traffic.json (--install)          {kernel label: bytes}, stamped with the sha of the kernel sources
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(hits, key=os.path.getmtime)   # (gpurun MERGES a call's files into gpurun_out/: an earlier profile of the same OUT stays beside the new one)


shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), one("pmc_fetch/*/*_counter_collection.csv"),
       one("pmc_write/*/*_counter_collection.csv"), prof + os.sep, tag] + (["--install"] if "--install" in sys.argv else [])
print(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for part in ("pmc_sq1", "pmc_sq2"):
    seen = set()
    for row in csv.DictReader(open(one(f"{part}/*/*_counter_collection.csv"))):
        k = row["Kernel_Name"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if part == "pmc_sq1" and (k, row["Dispatch_Id"]) not in seen:
            seen.add((k, row["Dispatch_Id"]))
            calls[k] += 1
# the ten heaviest kernels of the library (round 5: the fused ring kernels - ringtail, ringfirst - were missing from the top-4 list; torch's own
# copy kernels are not ours)
top = [k for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)) if "rocclr" not in k and not k.startswith("void at::")][:10]
out = {"method": "rocprofv3 --kernel-trace --pmc <8 SQ counters> in two passes over `bench.py --steps 2 --warmup 1 --no-extras` "
                 "(B=8 x 720p, bf16); sums over all launches of the kernel in the run; SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are in "
                 "units of 4 cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES in cycles per SIMD", "kernels": {}}
for k in top:
    c = dict(acc[k])
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0
    c["launches"] = calls[k]
    c["derived"] = {"issuing_pct_of_wave_cycles": round(100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 1),
                    "parked_waitcnt_or_barrier_pct": round(100 * c.get("SQ_WAIT_ANY", 0) / wc, 1),
                    "issue_stalled_pct": round(100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 1),
                    "valu_pct": round(100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 1),
                    "mfma_busy_cycles_per_simd_per_launch": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, calls[k]) / 1024.0, 0),
                    "lds_bank_conflict_pct_of_lds_cycles": round(100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds, 1)}
    out["kernels"][k] = c
    print(k[:70], c["derived"])
json.dump(out, open(os.path.join(prof, f"{tag}_sq_counters.json"), "w"), indent=1)
