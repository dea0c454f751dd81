#!/usr/bin/env python3
"""Registers / scratch / occupancy of the library's kernels from `python -m ppde_amd.build --force --usage` output.

    python -m ppde_amd.build --force --usage > /tmp/usage.log 2>&1;  python scripts/kernel_usage.py /tmp/usage.log [pattern]
"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rx = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
rows = rx.findall(txt)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (_, v, ag, sc, occ, lds), d in zip(rows, names):
    if pat in d:
        print(f"{d[:100]:100s} VGPR {v:>3} AGPR {ag:>3} scratch {sc:>4} occ {occ}")
