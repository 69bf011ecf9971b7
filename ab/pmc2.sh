export TMPDIR=/tmp
R=$PWD
tag=$1; export MIDIEMO_LIB=$2
i=0; dbs=""
while read -r P; do
  i=$((i+1)); d=/tmp/pmc2_${tag}_$i; rm -rf $d
  (cd /tmp && ITERS=3 rocprofv3 --kernel-trace --pmc $P -d $d -o r -- python $R/tools/bench_bwd_abl.py > $d.log 2>&1) || { echo "pass $i failed: $P"; grep -i "error\|invalid\|unable" $d.log | head -3; }
  f=$(find $d -name '*.db' 2>/dev/null | head -1); [ -n "$f" ] && dbs="$dbs $f"
done <<'EOT'
TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_LATENCY TCP_GATE_EN1
TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
TCP_TCR_TCP_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TD_TD_BUSY
TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS TD_TC_STALL TCP_TD_TCP_STALL_CYCLES
EOT
echo "=== $tag"
python tools/rocpd_pmc.py $dbs --match ${MATCH:-rga_bwd}
