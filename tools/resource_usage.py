#!/usr/bin/env python3
"""profiles/<tag>_resource_usage.txt: registers, spills, scratch, occupancy and static LDS of every kernel of
libcup2d_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage (cross-compiles; no GPU needed).
usage: python tools/resource_usage.py r02"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "scratch"
src_dir = os.path.join(ROOT, "cup2d_amd", "csrc")
rows = []
for f in sorted(f for f in os.listdir(src_dir) if f.endswith(".hip")):
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", f, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], cwd=src_dir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cur = None
    for line in r.stdout.decode().splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*)$", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], stdout=subprocess.PIPE).stdout.decode().strip()
            name = name.replace("void cup2d::", "").replace("cup2d::", "")
            cur = {"name": name.split("(")[0], "file": f}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
out = os.path.join(ROOT, "profiles", "%s_resource_usage.txt" % tag)
with open(out, "w") as fo:
    fo.write("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage (ROCm 7.2), per kernel of libcup2d_hip.so\n")
    fo.write("# kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | SGPR spills | VGPR spills | static LDS B/WG\n")
    for r in rows:
        fo.write("%-52s | %3s | %3s | %3s | %3s | %s | %2s | %2s | %s\n" % (
            r["name"][:52], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("SGPRs Spill", "?"),
            r.get("VGPRs Spill", "?"), r.get("LDS Size [bytes/block]", "?")))
print("wrote", out, len(rows), "kernels")
