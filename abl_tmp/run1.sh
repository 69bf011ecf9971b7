cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/p0 -o r -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --what attn > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/p0/r_results.db | grep -E "rga_" | awk '{print substr($1,20,20), $(NF-3)}'
