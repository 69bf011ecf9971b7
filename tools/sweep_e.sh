#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 768 1024 1280 1536 1792 2048 2560 3072 4096; do
  rm -rf /tmp/pe; ME_EBLOCKS=$n timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o p -- python $R/tools/bench_kernels.py --what attn --iters 10 > /dev/null 2>&1
  echo -n "eblocks=$n  "; python $R/tools/kstats.py /tmp/pe 8 | grep bwd_e | awk '{print $(NF-1), $NF}'
done
