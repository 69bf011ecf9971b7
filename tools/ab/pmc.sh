# usage: pmc.sh tag lib   -> prints counters of rga_bwd_q for that lib
export TMPDIR=/tmp
R=$PWD
tag=$1; export MIDIEMO_LIB=$2
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"
P4="SQ_BUSY_CYCLES SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
i=0; dbs=""
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); d=/tmp/pmc_${tag}_$i; rm -rf $d
  (cd /tmp && ITERS=3 rocprofv3 --kernel-trace --pmc $P -d $d -o r -- python $R/tools/bench_bwd_abl.py > $d.log 2>&1) || tail -5 $d.log
  dbs="$dbs $(find $d -name '*.db' | head -1)"
done
echo "=== $tag"
python tools/rocpd_pmc.py $dbs --match ${MATCH:-rga_bwd_q}
