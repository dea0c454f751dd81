#!/usr/bin/env python3
"""VGPRs / SGPRs / LDS / scratch of the kernels in a built library, read from the code object's notes (no rebuild):
    python scripts/kernel_regs.py [lib.so] [name pattern]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(REPO, "ppde_amd", "libppde_hip.so")
pat = [a for a in sys.argv[1:] if not a.endswith(".so")]
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    so = shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=d)
    obj = [p for p in os.listdir(d) if "gfx950" in p][0]
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(d, obj)], check=True, capture_output=True, text=True).stdout
blocks = notes.split("- .agpr_count:")[1:]
rows = []
for b in blocks:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), "0" if b.lstrip().split()[0] == "0" else b.lstrip().split()[0], g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    if all(p in n for p in pat):
        print(f"{n[:110]:110s} vgpr {r[1]:>3} agpr {r[2]:>3} sgpr {r[3]:>3} scratch {r[5]}")
