#!/bin/bash
# memory-path counters of the attention micro-benchmark (counters only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_attn2
i=0
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TCC_CYCLE_sum TCP_GATE_EN1_sum SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_attn2/s$i -o p -- python $R/tools/bench_kernels.py --what attn --iters 3 > /tmp/pmc_attn2_$i.log 2>&1 || tail -3 /tmp/pmc_attn2_$i.log
done
python $R/tools/pmc_kernels.py /tmp/pmc_attn2 ${KEYS:-rga_}
