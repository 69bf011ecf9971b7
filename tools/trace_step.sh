#!/bin/bash
# development aid: every kernel launch of ONE train step (the last of a short bench run) in order, with its duration and the gap
# to its predecessor -- per-launch view that the per-name averages of --stats hide (e.g. one slow epilogue variant of a GEMM).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python $R/bench.py --steps 6 --warmup 3 --no_decode --no_extra --no_cpu_baseline --no_probe > /dev/null 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "adamw" in n]
a, b = idx[-2] + 1, idx[-1] + 1
def short(n):
    m = re.search(r"(gemm_nt256|gemm_tn256|tn256_reduce|rga_bwd_q|rga_bwd_kv|rga_bwd_e|rga_fwd64|resid_ln_fwd|resid_ln_bwd|adamw|sumsq|cast_transpose_multi|ce_bwd|ce_fwd|embed_bwd_gather|embed_bwd_cond|embed_bwd_heavy|embed_fwd|key_pad|gemm_nt_kernel|gemm_tn_bf16)", n)
    return m.group(1) if m else n.split("(")[0][-40:]
tot = 0
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = s - int(rows[i - 1]["End_Timestamp"])
    tot += e - s
    print("%3d %-22s %8.1f us  gap %6.1f  grid %s" % (i - a, short(names[i]), (e - s) / 1e3, gap / 1e3, rows[i].get("Grid_Size", "")))
print("step: %d launches, kernel time %.3f ms, span %.3f ms" % (b - a, tot / 1e6, (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6))
PY
