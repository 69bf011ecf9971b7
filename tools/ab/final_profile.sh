# round-1 "f" evidence: kernel stats of the default bench command, bench line, SQ counters, HBM traffic
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r01_f
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no_cpu_baseline > /tmp/kt.log 2>&1)
python tools/rocpd_stats.py $(find /tmp/kt -name '*.db' | head -1) > $O/kernel_stats.txt 2>&1
find /tmp/kt -name '*stats*' | head -5 >> $O/kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hb_$C
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/hb_$C -o r -- python $R/bench.py --steps 4 --warmup 1 --no_cpu_baseline --no_probe > /tmp/hb_$C.log 2>&1)
done
python tools/hbm_traffic.py $(find /tmp/hb_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/hb_WRITE_SIZE -name '*.db' | head -1) --json $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
bash tools/ab/pmc_step.sh gemm_nt256 gemm_tn256 rga_fwd rga_bwd_q rga_bwd_kv rga_bwd_e resid_ln_bwd resid_ln_fwd ce_fwd > $O/counters.txt 2>&1
tail -1 $O/bench.json | cut -c1-400
