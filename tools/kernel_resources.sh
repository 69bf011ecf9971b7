#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [grep pattern]   -- VGPRs / AGPRs / scratch / LDS / occupancy of every kernel in a unit
cd "$(dirname "$0")/../midi-emotion_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 \
      -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "ScratchSize \[bytes/lane\]", "LDS Size \[bytes/block\]", "Occupancy \[waves/SIMD\]"):
        m = re.search(r"remark: .*?" + key + r": (\d+)", line)
        if m and cur is not None: cur[key.split()[0]] = m.group(1)
for r in rows:
    print("%-110s v %3s a %3s scratch %4s lds %6s occ %s" % (r["name"][:110], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("LDS"), r.get("Occupancy")))
' | { if [ -n "$2" ]; then grep -E "$2"; else cat; fi; }
