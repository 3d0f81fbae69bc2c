#!/usr/bin/env python3
"""Register / scratch table of the conv kernels: hipcc -Rpass-analysis=kernel-resource-usage, one line per instantiation.
Usage: python tools/resource_usage.py   (compiles vnect_amd/csrc/conv.hip for gfx950 into /tmp)"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "vnect_amd", "csrc", "conv.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/tmp/_ru.o"] + sys.argv[1:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-46s %5s %5s %7s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "sspill", "vspill"))
for k, v in rows.items():
    m = re.search(r"conv_stream_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)E", k)
    name = "stream<%s,%s,%s,%s,bf%s,prof%s>" % m.groups() if m else k[:46]
    print("%-46s %5d %5d %7d %6d %6d" % (name, v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1), v.get("SGPRs Spill", -1), v.get("VGPRs Spill", -1)))
