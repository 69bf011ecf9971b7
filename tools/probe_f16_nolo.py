"""f16 tier without the low half of the residual stream (MIDIEMO_RESID_LO=0): parity sample (bench.tier_parity_sample) and the step time,
beside the default (hi + lo).  Question: does the f16 tier (10 mantissa bits + f32 statistics) still meet north_star's 1e-3 with a
single 16-bit residual array -- 1.6 GB less LayerNorm traffic per step?"""
import os, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
import bench
for lo in (sys.argv[1:] or ["1", "0"]):
    if lo.startswith("b"):
        os.environ["MIDIEMO_RESID_LO_BITS"] = lo[1:]; lo = "1"
    else:
        os.environ["MIDIEMO_RESID_LO_BITS"] = "16"
    os.environ["MIDIEMO_RESID_LO"] = lo
    par = bench.tier_parity_sample()
    t16 = bench.tier_bench("fp16", bench.BATCH, bench.SEQ, steps=30, warmup=8)
    import time
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW
    torch.manual_seed(0)
    model, _ = build_model(dict(bench.CFG, compute_dtype="bf16")); model = model.cuda().train(); model.seed_dropout(1000)
    opt = FusedAdamW(model, lr=2e-5, clip=1.0)
    bt = [bench.synthetic_batch(bench.CFG, bench.BATCH, bench.SEQ, 1234 + 7919 * i, "cuda") for i in range(4)]
    for i in range(8): model.loss_and_backward(*bt[i % 4]); opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30): model.loss_and_backward(*bt[i % 4]); opt.step()
    torch.cuda.synchronize(); bf = (time.perf_counter() - t0) / 30 * 1e3
    del model, opt
    print("RESID_LO=%s parity" % lo, {k: par[k] for k in ("bf16_tier", "f16_tier")}, "f16 step ms", t16["ms_per_step"], "bf16 step ms %.3f" % bf, flush=True)
