#!/usr/bin/env python3
"""Disassembly of the kernels of libemavfi.so whose (mangled) name contains a substring: python tools/kernel_isa.py SUBSTR > out.s
(CPU box; the code objects are taken from the library's .hip_fatbin section as tests/test_cabi_cpu.py does)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects, LLVM, ROOT  # noqa: E402

so = os.environ.get("EMAVFI_LIB", os.path.join(ROOT, "video-frame-interpolation_amd", "emavfi", "lib", "libemavfi.so"))
pat = sys.argv[1]
with tempfile.TemporaryDirectory() as tmp:
    for co in code_objects(so, tmp):
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
        for name, body in re.findall(r"<(\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
            if pat in name and "s_endpgm" in body:
                print(f"; ===== {name}")
                print(body)
