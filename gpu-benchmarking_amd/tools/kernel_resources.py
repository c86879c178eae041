#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy of every kernel in one csrc/*.hip file (hipcc
-Rpass-analysis=kernel-resource-usage), optionally filtered by a substring of the demangled name.

    python3 tools/kernel_resources.py csrc/bwdtrans_hex.hip [filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function",
       "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for ln in err.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if flt in d:
        r = rows[n]
        short = re.sub(r"^void sf::", "", d).split("(")[0]
        print(f"{short:70s} vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', -1):3d} sgpr {r.get('TotalSGPRs', -1):3d} "
              f"scratch {r.get('ScratchSize', -1):5d} spill v{r.get('VGPRs Spill', -1)}/s{r.get('SGPRs Spill', -1)} "
              f"occ {r.get('Occupancy', -1)}")
