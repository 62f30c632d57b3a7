#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of the code objects embedded in libemavfi.so (CPU box, no GPU needed).

  python tools/kernel_resources.py [substring ...]        # e.g. pack3 ring

Reads the AMDGPU metadata notes of every gfx950 code object in the library's .hip_fatbin section.  Occupancy per SIMD
follows MI355X_MICROARCH.md (register granule 8, min(8, 512 / alloc)); LDS occupancy = floor(160 KiB / LDS per workgroup)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    for i, a in enumerate(starts):
        part = os.path.join(tmp, f"b_{i}.bin")
        open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, f"d_{i}.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={part}", f"--output={co}"])
        yield co


def kernels(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):   # first key of a kernel record in llvm's ordering is .agpr_count
            pass
        if k == "name" and "symbol" not in cur and v.startswith("_Z") is False and not v.startswith("'"):
            continue
        cur[k] = v
        if k == "wavefront_size":   # last key of a kernel record
            if "symbol" in cur:
                yield dict(cur)
            cur = {}


def main():
    so = os.environ.get("EMAVFI_LIB", os.path.join(ROOT, "video-frame-interpolation_amd", "emavfi", "lib", "libemavfi.so"))
    pats = sys.argv[1:]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            for k in kernels(co):
                sym = k.get("symbol", "").strip("'").replace(".kd", "")
                name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
                if pats and not any(p in name for p in pats):
                    continue
                v, a = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0))
                alloc = (max(v + 0, 1) + 7) // 8 * 8
                lds = int(k.get("group_segment_fixed_size", 0))
                rows.append((name[:110], v, a, int(k.get("sgpr_count", 0)), lds, int(k.get("private_segment_fixed_size", 0)),
                             int(k.get("vgpr_spill_count", 0)), min(8, 512 // alloc)))
    print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds(static)':>11s} {'scratch':>8s} {'spill':>6s} {'waves/SIMD(regs)':>17s}")
    for r in sorted(rows):
        print(f"{r[0]:110s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:11d} {r[5]:8d} {r[6]:6d} {r[7]:17d}")


if __name__ == "__main__":
    main()
