cd /tmp; export TMPDIR=/tmp
for n in 0 1 2 3 4; do
  if [ $n = 0 ]; then unset MIDIEMO_LIB; else export MIDIEMO_LIB=$GRAFT_REPO_ROOT/abl_tmp/abl$n.so; fi
  rocprofv3 --kernel-trace -d /tmp/p$n -o r -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --what attn > /dev/null 2>&1
  echo "ABL $n"; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/p$n/r_results.db | grep -E "rga_bwd_q|rga_fwd" | awk '{print substr($1,20,18), $(NF-3)}'
done
