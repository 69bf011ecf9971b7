export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rga or attn or bwd" 2>&1 | tail -2
for tag in prev bf new prev bf new; do
  if [ $tag = new ]; then unset MIDIEMO_LIB; else export MIDIEMO_LIB=$R/ab/lib_$tag.so; fi
  rm -rf /tmp/p_$tag
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/p_$tag -o r -- python $R/tools/bench_bwd_abl.py > /tmp/p_$tag.log 2>&1)
  echo "$tag $(grep 'rga_bwd total' /tmp/p_$tag.log)"
  python tools/rocpd_stats.py $(find /tmp/p_$tag -name "*.db" | head -1) | grep -i "rga" | cut -c1-30,64-130
done
