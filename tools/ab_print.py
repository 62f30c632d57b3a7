#!/usr/bin/env python3
"""Print value / step time / selected kernels of bench JSONs: ab_print.py DIR name1 name2 ... [--match substr,substr]"""
import json, sys
args = [a for a in sys.argv[1:] if not a.startswith("--match")]
match = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--match=")), ["ck=64,nf=1", "ck=64,nf=2", "deform", "tail"])
d0 = args[0]
for n in args[1:]:
    d = json.loads(open(f"{d0}/{n}.json").read().strip().splitlines()[-1])
    ks = {x["kernel"]: x["avg_us"] for x in d["kernels"]}
    print(n, d["value"], d["ms_per_step"], {k: v for k, v in ks.items() if any(m in k for m in match)})
