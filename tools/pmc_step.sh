#!/bin/bash
# MFMA / issue utilisation of every kernel family INSIDE the live train step (north_star: "rocprof ... MFMA utilisation against peak"):
# counters-only rocprofv3 passes (separate --pmc runs, kernel trace only) over a short train-only bench; per-kernel means through
# tools/rocpd_pmc.py.  usage (GPU box, repo root): bash tools/pmc_step.sh > gpurun_out/pmc_step.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_step
SHORT="--steps 3 --warmup 1 --no_cpu_baseline --no_probe --no_decode --no_extra"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_step/s$i -o p -- python $R/bench.py $SHORT > /tmp/pmc_step_$i.log 2>&1 || tail -3 /tmp/pmc_step_$i.log
done
python $R/tools/rocpd_pmc.py $(find /tmp/pmc_step -name "*.db")
