#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a pipelined (two-stream) bench run and says what actually overlapped
(VERDICT r4 item 1): per kernel family the launches, average duration and total; the sum of all kernel durations against the
wall-clock span they cover; the time during which kernels of two queues were in flight together, and how much of the pack
kernel's time was spent beside a convolution of the other stream.   usage: twostream_overlap.py TRACE.csv [skip_first_n_dispatches]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def col(r, *names):
    for n in names:
        if n in r:
            return r[n]
    raise KeyError(names)


ev = []
for r in rows:
    name = col(r, "Kernel_Name", "kernel_name")
    if name.startswith("void at::") or "elementwise" in name:
        continue
    ev.append((int(col(r, "Start_Timestamp", "start_timestamp")), int(col(r, "End_Timestamp", "end_timestamp")), col(r, "Queue_Id", "queue_id"), name))
ev.sort()
ev = ev[skip:]


def family(n):
    for key, fam in (("deform_pack3", "pack"), ("ring2", "conv ring2"), ("ringfirst", "conv ringfirst"), ("ringtail", "conv ringtail"), ("s2ring", "conv s2ring"),
                     ("conv3x3_ring", "conv ring"), ("wreg", "conv wreg"), ("warp", "warp"), ("pool", "small"), ("ctx_finish", "small"), ("blob_guard", "small")):
        if key in n:
            return fam
    return "other"


agg = {}
for s, e, q, n in ev:
    a = agg.setdefault(family(n), [0, 0])
    a[0] += 1
    a[1] += e - s
total = sum(a[1] for a in agg.values())
span = max(e for _, e, _, _ in ev) - min(s for s, _, _, _ in ev)
print(f"{len(ev)} kernel dispatches on queues {sorted(set(q for _, _, q, _ in ev))}")
for fam, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {fam:16s} {n:5d} launches  avg {t / n / 1e3:9.1f} us  total {t / 1e6:9.3f} ms  ({100 * t / total:5.1f} % of the sum)")
print(f"sum of kernel durations {total / 1e6:.3f} ms over a span of {span / 1e6:.3f} ms: ratio {total / span:.3f} (1.0 = no overlap, no gaps)")
# sweep: time with >= 2 kernels in flight, and pack time beside a conv of another queue
pts = []
for i, (s, e, q, n) in enumerate(ev):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active = set()
last = pts[0][0]
both = pack_beside_conv = pack_time = 0
for t, d, i in pts:
    dt = t - last
    if dt > 0 and active:
        qs = set(ev[j][2] for j in active)
        fams = [family(ev[j][3]) for j in active]
        if len(qs) >= 2:
            both += dt
        if "pack" in fams:
            pack_time += dt
            if any(f.startswith("conv") for f in fams):
                pack_beside_conv += dt
    last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
print(f"two queues in flight together: {both / 1e6:.3f} ms = {100 * both / span:.1f} % of the span")
print(f"a pack kernel in flight: {pack_time / 1e6:.3f} ms, of which beside a convolution kernel: {pack_beside_conv / 1e6:.3f} ms = {100 * pack_beside_conv / max(pack_time, 1):.1f} %")
