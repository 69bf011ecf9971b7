"""f16 tier without the low half of the residual stream (MIDIEMO_RESID_LO=0): parity sample (bench.tier_parity_sample) and the step time,
beside the default (hi + lo).  Question: does the f16 tier (10 mantissa bits + f32 statistics) still meet north_star's 1e-3 with a
single 16-bit residual array -- 1.6 GB less LayerNorm traffic per step?"""
import os, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
import bench
for lo in ("1", "0"):
    os.environ["MIDIEMO_RESID_LO"] = lo
    par = bench.tier_parity_sample()
    t16 = bench.tier_bench("fp16", bench.BATCH, bench.SEQ, steps=30, warmup=8)
    print("RESID_LO=%s parity" % lo, {k: par[k] for k in ("bf16_tier", "f16_tier")}, "f16 step ms", t16["ms_per_step"], "loss", t16["final_loss"], flush=True)
