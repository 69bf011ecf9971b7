#!/bin/bash
# rocprofv3 kernel stats of the attention micro-benchmark (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_attn
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn -o p -- python $R/tools/bench_kernels.py --what attn --iters ${ITERS:-10} > /dev/null 2>&1
python $R/tools/kstats.py /tmp/prof_attn ${TOP:-8}
