#!/usr/bin/env python3
"""Register / scratch use of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: kernel_regs.py polgen-rvc_amd/csrc/resblock.hip [name filter]"""
import re, subprocess, sys
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"rvcx::\(anonymous namespace\)::|void |\(rvcx::\w+\)", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if flt in name:
        print(f"{name[:90]:90s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} spill {r.get('VGPRs Spill'):>4s} "
              f"scratch {r.get('ScratchSize [bytes/lane]'):>4s} occ {r.get('Occupancy [waves/SIMD]')}")
