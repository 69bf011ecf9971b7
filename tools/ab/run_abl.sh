export TMPDIR=/tmp
R=$PWD
for tag in "$@"; do
  export MIDIEMO_LIB=$R/ab/lib_$tag.so
  rm -rf /tmp/p_$tag
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/p_$tag -o r -- python $R/tools/bench_bwd_abl.py > /tmp/p_$tag.log 2>&1)
  echo "$tag $(grep 'rga_bwd total' /tmp/p_$tag.log) $(python tools/rocpd_stats.py $(find /tmp/p_$tag -name '*.db' | head -1) | grep -i 'rga_bwd_q' | cut -c64-100)"
done
