#!/usr/bin/env python
"""Compile one .hip file for gfx950 with -Rpass-analysis=kernel-resource-usage and print a
compact table: kernel, VGPRs, AGPRs, SGPRs, scratch, VGPR spills, occupancy, LDS."""
import re
import subprocess
import sys

src = sys.argv[1]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form"] + sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)[:60]
    print(f"{name:60s} vgpr {r.get('VGPRs', 0):4d} agpr {r.get('AGPRs', 0):4d} sgpr {r.get('TotalSGPRs', 0):4d} "
          f"scratch {r.get('ScratchSize', 0):4d} vspill {r.get('VGPRs Spill', 0):3d} occ {r.get('Occupancy', 0)}")
