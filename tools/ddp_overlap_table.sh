#!/bin/bash
# one MI355X, ONE RCCL rank forced through the exchange machinery (MIDIEMO_BENCH_FORCE_DIST=1 + MIDIEMO_DDP_FORCE=1:
# asynchronous bucket all-reduces on RCCL's stream, comm windows, work-handle waits), the two-stream attention backward on / off
# (MIDIEMO_ATTN_BWD_OVERLAP) under every bucket policy: step time and the exposed wait of GradAllReducer.finish().
# usage: bash tools/ddp_overlap_table.sh > gpurun_out/r06_ddp_overlap.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIDIEMO_BENCH_FORCE_DIST=1 MIDIEMO_DDP_FORCE=1 MASTER_ADDR=127.0.0.1
echo "policy compress attn_bwd_overlap ms_per_step median_ms exposed_wait_ms_per_step exposed_frac"
P=29700
for rep in 1 2; do
for pol in window end eager; do
  for comp in "" bf16; do
    for ov in 1 0; do
      P=$((P+1))
      MASTER_PORT=$P MIDIEMO_DDP_POLICY=$pol MIDIEMO_DDP_COMPRESS=$comp MIDIEMO_ATTN_BWD_OVERLAP=$ov timeout 300 python $R/bench.py --gpus 1 --steps 30 --warmup 8 \
        --no_cpu_baseline --no_probe --no_decode --no_extra 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        d=json.loads(ln); dd=d['ddp']
        print('$pol', '${comp:-f32}', '$ov', d['ms_per_step'], d['median_ms_per_step'], dd.get('exposed_wait_ms_per_step'), dd.get('exposed_frac_of_step'))
"
    done
  done
done
done
