#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU
needed).  python tools/resources.py ssd_policy_mfma.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, HIPCC_FLAGS  # noqa: E402

src = os.path.join(CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = [f for f in HIPCC_FLAGS if f != "-shared"] + [a for a in sys.argv[3:]]
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_res.o"],
                     stderr=subprocess.PIPE, stdout=subprocess.DEVNULL).stderr.decode()
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = re.sub(r"\(.*", "", k)
        print("%-60s VGPR %3d AGPR %3d SGPR %3d  scratch %4d  sgpr-spill %3d vgpr-spill %3d  occ %d" % (
            name[-60:], v.get("VGPRs", -1), v.get("AGPRs", 0), v.get("TotalSGPRs", -1), v.get("ScratchSize", 0), v.get("SGPRs Spill", 0),
            v.get("VGPRs Spill", 0), v.get("Occupancy", 0)))
