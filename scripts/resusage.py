#!/usr/bin/env python
"""kernel resource usage of one translation unit: scripts/resusage.py csrc/file.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:110]
        rows[cur] = {}
    else:
        rows[cur][k.split(" [")[0]] = v
for k, v in rows.items():
    print(f"{k}\n     " + "  ".join(f"{a}={b}" for a, b in v.items()))
