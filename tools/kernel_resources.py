#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
usage: kernel_resources.py <file.hip> [name-filter]   (run from the csrc directory of the tree to inspect; used for the r03-vs-HEAD codegen
comparison in profiles/r05_ab_r03_vs_head.txt)"""
import re, subprocess, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": dem.replace("(anonymous namespace)::", "")}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print("%-90s %5s %5s %5s %7s %4s %6s %6s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for r in rows:
    if filt and filt not in r["name"]: continue
    print("%-90s %5s %5s %5s %7s %4s %6s %6s %7s" % (r["name"][:90], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
