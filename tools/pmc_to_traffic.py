#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.

usage: pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_dir/> <round tag, e.g. r02_final> [--install]
(--install also writes profiles/traffic.json, stamped with the sha of the kernel sources it was measured on: bench.py
refuses a file whose stamp does not match the sources it runs)
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced read and is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores and taken as is.  Kernel labels match bench.py."""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            k = row["Kernel_Name"]
            if "conv3x3_ring_kernel<" in k:   # garbled demangling (any instance with a `true` argument): tell them apart by their LDS size
                k += "|lds=" + row.get("LDS_Block_Size", "0")
            d[k].append(float(row["Counter_Value"]))
    return d


def label(k):
    m = re.match(r"_Z18deform_pack_kernelI(DF16b|DF16_)Li(\d+)ELi(\d+)ELb1", k)
    if m:  # the LDS-window kernel, one launch per ModulatedDeformConvPack (csrc/deform_pack.inl)
        return f"deform<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck=80,nf=3>"
    if "deform_pack3_kernel" in k:   # rocprofv3 prints the template demangled without its arguments: the bench runs the bf16 fused instance
        return "deform<bf16,ck=80,nf=3>"
    if "deform_f32w_kernel" in k:
        return "deform<f32,ck=80,nf=3>"
    m = re.match(r"_Z13deform_kernelI(DF16b|DF16_|f)Li(\d+)ELi(\d+)E", k)
    if m:
        return f"deform<{ {'DF16b': 'bf16', 'DF16_': 'f16', 'f': 'f32'}[m.group(1)] },ck={m.group(2)},nf={m.group(3)}>"
    m = re.match(r"_Z17conv_first_kernelI(DF16b|DF16_)", k)
    if m:
        return f"conv_first<{'bf16' if m.group(1) == 'DF16b' else 'f16'},6->64>"
    m = re.match(r"_Z25conv3x3_pingpong16_kernelI(DF16b|DF16_)Li64ELi2", k)
    if m:
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck=64,nf=2,s=1>"
    m = re.match(r"_Z24conv3x3_persist16_kernelI(DF16b|DF16_)Li64ELi1", k)
    if m:
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck=64,nf=1,s=1>"
    m = re.match(r"_Z19conv3x3_ring_kernelI(DF16b|DF16_)Lb(0|1)ELb(0|1)ELb[01]E", k)
    if m:  # 64 -> 64 layers / reconstruction.0 (67 -> 64) / motion_estimation.1 + .2: weights in registers, input rows through an LDS ring (csrc/conv_ring.inl)
        t = 'bf16' if m.group(1) == 'DF16b' else 'f16'
        if m.group(3) == '1':
            return f"conv3x3+head<{t},64->64->2>"
        return f"conv3x3<{t},ck={'80' if m.group(2) == '1' else '64'},nf=2,s=1>"
    if "conv3x3_ring_kernel<" in k:   # rocprofv3 garbles the demangling of instances with a `true` argument; the bench runs bf16;
        lds = int(k.rsplit("|lds=", 1)[1]) if "|lds=" in k else 0   # HEAD: two rings, 81 424 B; TAIL: 74 752 B
        return "conv3x3+head<bf16,64->64->2>" if lds > 78000 else "conv3x3<bf16,ck=80,nf=2,s=1>"
    m = re.match(r"_Z20conv3x3_ring2_kernelI(DF16b|DF16_)", k)
    if m:  # conv_block_1 + conv_block_2 in one launch (csrc/conv_ring2.inl)
        return f"conv3x3+conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},64->64->64>"
    if "conv3x3_ring2_kernel<" in k:   # (garbled demangling of the instance with ALT = true; the bench runs bf16)
        return "conv3x3+conv3x3<bf16,64->64->64>"
    m = re.match(r"_Z23conv3x3_ringtail_kernelI(DF16b|DF16_)", k)
    if m:  # reconstruction.1 + .2 in one launch (csrc/conv_ring_tail.inl)
        return f"conv3x3+tail<{'bf16' if m.group(1) == 'DF16b' else 'f16'},64->32->3>"
    m = re.match(r"_Z24conv3x3_ringfirst_kernelI(DF16b|DF16_)", k)
    if m:
        return f"conv_first+conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},6->64->64>"
    m = re.match(r"_Z21conv3x3_s2ring_kernelI(DF16b|DF16_)", k)
    if m:  # context_encoding.0: weights in registers, input rows through an LDS ring (csrc/conv3x3.inl)
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck=64,nf=4,s=2>"
    m = re.match(r"_Z19conv3x3_wreg_kernelI(DF16b|DF16_)Li(1|2)E", k)
    if m:  # context_encoding.1 / .2: all 256 output channels in one pass, weights streamed into registers (csrc/conv_wreg.inl)
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f16'},ck={'64' if m.group(2) == '1' else '32'},nf=8,s={m.group(2)}>"
    if "conv3x3_wreg_kernel<" in k:   # (garbled demangling of the bf16 stride-1 instance)
        return "conv3x3<bf16,ck=64,nf=8,s=1>"
    m = re.match(r"_Z14conv3x3_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)EEv", k)
    if m:
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f32'},ck={m.group(2)},nf={m.group(3)},s={m.group(4)}>"
    m = re.search(r"conv3x3_kernel<(float|__bf16|_Float16), (\d+), (\d+), (\d+)>", k)   # demangled form (instances without bool arguments)
    if m:
        return f"conv3x3<{ {'float': 'f32', '__bf16': 'bf16', '_Float16': 'f16'}[m.group(1)] },ck={m.group(2)},nf={m.group(3)},s={m.group(4)}>"
    m = re.match(r"_Z22conv3x3_persist_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi\d+EEv", k)
    if m:  # persistent variant serves the same layers (stride 1) as the tile-per-workgroup kernel
        return f"conv3x3<{'bf16' if m.group(1) == 'DF16b' else 'f32'},ck={m.group(2)},nf={m.group(3)},s=1>"
    if "warp_tiled_kernel" in k:
        return "warp_tiled<3,nchw fp32> (emavfi_warp)" if "void>" in k else ("warp_fused<f32>" if "float" in k else "warp_fused<bf16>")
    if "pack_input_kernel" in k:
        return "pack_input"
    if "pool_partial_kernel" in k:
        return "avg_pool_partial"
    return None


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
# several kernel instances can serve one bench label (e.g. the 64 -> 64 ring kernel and its instance that stores the other 16-bit
# type): pool their launches
fl, wl = collections.defaultdict(list), collections.defaultdict(list)
for k in f:
    lab = label(k)
    if lab:
        fl[lab] += f[k]
        wl[lab] += w.get(k, [])
out = {}
for lab in sorted(fl, key=lambda k: -sum(fl[k])):
    fk, wk = sum(fl[lab]) / len(fl[lab]), sum(wl[lab]) / max(1, len(wl[lab]))
    rd, wr = fk * 1024 * 2, wk * 1024
    out[lab] = {"hbm_bytes_per_launch": rd + wr, "fetch_bytes_corrected_x2": rd, "write_bytes": wr,
                "launches_sampled": len(fl[lab]), "raw_FETCH_SIZE_KiB": fk, "raw_WRITE_SIZE_KiB": wk}
    print(f"{lab:42s} n={len(fl[lab]):3d} read {rd / 1e6:9.1f} MB  write {wr / 1e6:8.1f} MB  total {(rd + wr) / 1e6:9.1f} MB")
tag = sys.argv[4] if len(sys.argv) > 4 else "rXX"
json.dump({"method": __doc__, "kernels": out}, open(sys.argv[3] + f"{tag}_pmc_traffic_detail.json", "w"), indent=1)
if "--install" in sys.argv or any(a.startswith("--install-as=") for a in sys.argv):
    import datetime, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    # kernel_source_sha() without importing torch: replicate bench.py's definition
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "video-frame-interpolation_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inl", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    t = {k: v["hbm_bytes_per_launch"] for k, v in out.items()}
    t["_kernel_source_sha"] = h.hexdigest()[:16]
    t["_measured"] = f"{tag}, {datetime.date.today().isoformat()}, profiles/{tag}_pmc_traffic_detail.json"
    name = "traffic.json"
    for a in sys.argv:
        if a.startswith("--install-as="):
            name = a.split("=", 1)[1]
    json.dump(t, open(os.path.join(root, "profiles", name), "w"), indent=1)
