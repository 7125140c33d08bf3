"""Compact table of `make -C rttnw_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage): one line per kernel.
Usage: python profiles/resource_usage.py [substring ...]   (only kernels whose demangled name contains every substring)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "rttnw_amd", "csrc"), "resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:Function|Kernel) Name: (\S+)", line) or re.search(r"Name: (\S+) \[-Rpass", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void ", ""))
    if all(s in n for s in sys.argv[1:]):
        print("%-78s VGPR %-4s AGPR %-4s scratch %-5s occ %-3s LDS %s" % (n, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                                                  r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
