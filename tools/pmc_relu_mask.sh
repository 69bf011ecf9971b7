#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm_f /tmp/pm_w
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_f -o p -- python $R/tools/pmc_relu_mask.py run > /tmp/pm_f.log 2>&1 || tail -3 /tmp/pm_f.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pm_w -o p -- python $R/tools/pmc_relu_mask.py run > /tmp/pm_w.log 2>&1 || tail -3 /tmp/pm_w.log
python $R/tools/pmc_relu_mask.py report $(find /tmp/pm_f -name "*.db" | head -1) $(find /tmp/pm_w -name "*.db" | head -1)
