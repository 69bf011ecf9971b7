#!/bin/bash
# development aid: kernel order of the last train step of a short bench run (find stray copies / fills)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python $R/bench.py --steps 2 --warmup 1 --no_decode --no_extra --no_cpu_baseline --no_probe > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last adamw marks the end of the last step; previous adamw its start
idx = [i for i, n in enumerate(names) if "adamw" in n]
a, b = idx[-2] + 1, idx[-1] + 1
prev = None
for i in range(a, b):
    n = names[i]
    short = n.split("(")[0][-60:]
    if "copyBuffer" in n or "Fill" in n or "elementwise" in n:
        print("%4d  >>> %-50s  after: %s" % (i - a, short, names[i - 1].split("(")[0][-50:]))
PY
