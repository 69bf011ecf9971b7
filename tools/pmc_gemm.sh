#!/bin/bash
# SQ counter passes over the GEMM micro-benchmark (run on the GPU box from the repo root); counters only, no trace domains
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_gemm
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_gemm/s$i -o p -- python $R/tools/bench_kernels.py --what gemm --iters 3 --only ${ONLY:-ffn2,dW2} > /tmp/pmc_gemm_$i.log 2>&1 || tail -3 /tmp/pmc_gemm_$i.log
done
python $R/tools/pmc_kernels.py /tmp/pmc_gemm ${KEYS:-gemm_nt256 gemm_tn256}
