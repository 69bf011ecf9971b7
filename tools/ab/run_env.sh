# usage: run_env.sh VAR v1 v2 ... : per-kernel table for each value of the env var
export TMPDIR=/tmp
R=$PWD
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  rm -rf /tmp/s_$v
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/s_$v -o r -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_probe > /tmp/s_$v.log 2>&1)
  echo "== $VAR=$v $(tail -1 /tmp/s_$v.log | cut -c100-200)"
  python tools/rocpd_stats.py $(find /tmp/s_$v -name '*.db' | head -1) 8 | grep "${PAT:-rga}" | cut -c1-40,62-130
done
