cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+\[?|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|LDSBankConflict|MfmaUtil|VALUBusy)\b" | sort -u | tr '\n' ' ' | head -c 6000
echo
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --what attn --iters 2 > /dev/null 2>&1
cp /tmp/pmc1/p_results.db $GRAFT_REPO_ROOT/gpurun_out/pmc1.db
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --what attn --iters 2 > /dev/null 2>&1
cp /tmp/pmc2/p_results.db $GRAFT_REPO_ROOT/gpurun_out/pmc2.db
