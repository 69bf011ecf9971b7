#!/bin/bash
# End-of-milestone evidence, run on the GPU box from the repo root:  bash tools/round_profiles.sh r02_b
#   0. the DEFAULT bench.py command, un-profiled                            -> <tag>_bench.json
#   1. rocprofv3 --kernel-trace --stats over the C2 train steps ALONE (--no_probe --no_extra --no_decode) -> <tag>_step_kernel_stats.txt, step_trace.json,
#      and over the config-4 sub-measurement alone                           -> <tag>_config4_kernel_stats.txt
#      (round 2 traced the default command: attention rows mixed L = 1024 and L = 2048 launches, 40 % decode kernels)
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, counters only) over a short train-only bench -> <tag>_hbm_traffic.txt, hbm_traffic.json
#   3. kernel trace of 64 eager decode steps at t = 1024                   -> <tag>_decode_kernel_stats.txt
# Everything lands in gpurun_out/prof/ (copy what is to be judged into profiles/).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_*
timeout 900 python $R/bench.py > /tmp/rp_plain.log 2>&1
grep "^{\"metric\"" /tmp/rp_plain.log | tail -1 > $OUT/${TAG}_bench.json
# 1. the train steps ALONE (--no_probe: no instrumented steps, no warm replays of the NT calls -- round 5's *_bench_kernel_stats.txt
#    mixed 3 600 replay launches into the per-kernel averages): per-step time of every kernel family + profiles/step_trace.json
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_step -o p -- python $R/bench.py --no_extra --no_decode --no_cpu_baseline --no_probe --steps 20 --warmup 5 > /tmp/rp_step.log 2>&1
PROFILE_TAG=profiles/${TAG}_step_kernel_stats.txt python $R/tools/rocpd_stats.py $(find /tmp/rp_step -name "*.db" | head -1) 25 --json $OUT/step_trace.json > $OUT/${TAG}_step_kernel_stats.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_c4 -o p -- python $R/bench.py --only_config4 > /tmp/rp_c4.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/rp_c4 -name "*.db" | head -1) > $OUT/${TAG}_config4_kernel_stats.txt 2>&1
python $R/tools/make_roofline.py > /tmp/rp_roof.log 2>&1
SHORT="--steps 4 --warmup 1 --no_cpu_baseline --no_probe --no_decode --no_extra"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_f -o p -- python $R/bench.py $SHORT > /tmp/rp_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/rp_w -o p -- python $R/bench.py $SHORT > /tmp/rp_w.log 2>&1
PROFILE_TAG=profiles/${TAG}_hbm_traffic.txt python $R/tools/hbm_traffic.py $(find /tmp/rp_f -name "*.db" | head -1) $(find /tmp/rp_w -name "*.db" | head -1) --json $OUT/hbm_traffic.json > $OUT/${TAG}_hbm_traffic.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_dec -o p -- python $R/tools/prof_decode.py 1024 64 bf16 > /tmp/rp_dec.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/rp_dec -name "*.db" | head -1) 64 > $OUT/${TAG}_decode_kernel_stats.txt 2>&1
ls -la $OUT; head -12 $OUT/${TAG}_step_kernel_stats.txt; head -14 $OUT/${TAG}_hbm_traffic.txt; cut -c1-300 $OUT/${TAG}_bench.json
