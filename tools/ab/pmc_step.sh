# SQ counters for all kernels of a short bench run: pmc_step.sh <match>
export TMPDIR=/tmp
R=$PWD
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="SQ_BUSY_CYCLES SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"
i=0; dbs=""
for P in "$P1" "$P2" "$P4"; do
  i=$((i+1)); d=/tmp/pmcs_$i; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_probe > $d.log 2>&1) || tail -3 $d.log
  dbs="$dbs $(find $d -name '*.db' | head -1)"
done
for m in "$@"; do python tools/rocpd_pmc.py $dbs --match $m; done
