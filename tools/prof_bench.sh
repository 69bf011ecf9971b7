#!/bin/bash
# rocprofv3 kernel stats of the train-step bench (run on the GPU box from the repo root); prints per-kernel time per step
R=${GRAFT_REPO_ROOT:-$(pwd)}
STEPS=${STEPS:-10}; WARM=${WARM:-3}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o p -- python $R/bench.py --steps $STEPS --warmup $WARM --no_decode --no_extra --no_cpu_baseline --no_probe > /tmp/prof_bench.log 2>&1
tail -1 /tmp/prof_bench.log | cut -c1-400
python - <<PY
import csv, glob
n = $STEPS + $WARM
for f in glob.glob("/tmp/prof_bench/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time per step: %.3f ms" % (tot / n / 1e6))
    for r in rows[:${TOP:-40}]:
        print("%-84s %5d/step %8.1f us avg %8.3f ms/step %5.1f%%" % (r["Name"][:84], int(r["Calls"]) // n, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / n / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
mkdir -p $R/gpurun_out/prof_bench && cp $(find /tmp/prof_bench -name "*kernel_stats.csv") $R/gpurun_out/prof_bench/ 2>/dev/null
