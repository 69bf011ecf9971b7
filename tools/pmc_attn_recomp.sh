#!/bin/bash
# HBM bytes per launch of the attention kernels with the query-owned backward reading the saved tiles back / rebuilding P
# (two PMC passes, counters only; tools/hbm_traffic.py applies the guide's gfx950 corrections)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa_f /tmp/pa_w
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pa_f -o p -- python $R/tools/ab_attn_recomp.py > /tmp/pa_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pa_w -o p -- python $R/tools/ab_attn_recomp.py > /tmp/pa_w.log 2>&1
python $R/tools/hbm_traffic.py $(find /tmp/pa_f -name "*.db" | head -1) $(find /tmp/pa_w -name "*.db" | head -1)
