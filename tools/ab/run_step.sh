# usage: run_step.sh tag... ; per-kernel table of a 6-step bench run for each lib (tag "new" = in-tree build)
export TMPDIR=/tmp
R=$PWD
for tag in "$@"; do
  if [ $tag = new ]; then unset MIDIEMO_LIB; else export MIDIEMO_LIB=$R/ab/lib_$tag.so; fi
  rm -rf /tmp/s_$tag
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/s_$tag -o r -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_probe > /tmp/s_$tag.log 2>&1)
  echo "== $tag $(tail -1 /tmp/s_$tag.log | cut -c100-200)"
  python tools/rocpd_stats.py $(find /tmp/s_$tag -name '*.db' | head -1) 8 | head -${TOP:-14} | cut -c1-40,62-130
done
