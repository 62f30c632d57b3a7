#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.

usage: pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix>
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced read and is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores and taken as is.  Kernel labels match bench.py."""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            d[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return d


def label(k):
    m = re.match(r"_Z17deform_lds_kernelI(DF16b|DF16_)Li(\d+)ELi(\d+)", k)
    if m:  # the LDS-window kernel (one launch per ModulatedDeformConvPack when its last template argument is true)
        return f"deform<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck={m.group(2)},nf={m.group(3)}>"
    m = re.match(r"_Z14conv3x3_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)EEv", k)
    if m:
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f32'},ck={m.group(2)},nf={m.group(3)},s={m.group(4)}>"
    m = re.match(r"_Z22conv3x3_persist_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi\d+EEv", k)
    if m:  # persistent variant serves the same layers (stride 1) as the tile-per-workgroup kernel
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f32'},ck={m.group(2)},nf={m.group(3)},s=1>"
    if "warp_tiled_kernel" in k:
        return "warp_tiled<3,nchw fp32> (emavfi_warp)" if "void>" in k else "warp_fused<bf16>"
    if "pack_input_kernel" in k:
        return "pack_input"
    if "pool_partial_kernel" in k:
        return "avg_pool_partial"
    return None


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f, key=lambda k: -sum(f[k])):
    lab = label(k)
    if not lab:
        continue
    fk, wk = sum(f[k]) / len(f[k]), sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    rd, wr = fk * 1024 * 2, wk * 1024
    out[lab] = {"hbm_bytes_per_launch": rd + wr, "fetch_bytes_corrected_x2": rd, "write_bytes": wr,
                "launches_sampled": len(f[k]), "raw_FETCH_SIZE_KiB": fk, "raw_WRITE_SIZE_KiB": wk}
    print(f"{lab:42s} n={len(f[k]):3d} read {rd / 1e6:9.1f} MB  write {wr / 1e6:8.1f} MB  total {(rd + wr) / 1e6:9.1f} MB")
json.dump({k: v["hbm_bytes_per_launch"] for k, v in out.items()}, open(sys.argv[3] + "traffic.json", "w"), indent=1)
json.dump({"method": __doc__, "kernels": out}, open(sys.argv[3] + "r01_final_pmc_traffic_detail.json", "w"), indent=1)
