"""AB_WHAT=relu_mask (default) | attn_overlap | resid_lo.  Same-process, interleaved A/B of the whole train step (bench.py's workload) with and without the ReLU sign mask
(MIDIEMO_NO_RELU_MASK is read when a workspace is created: the workspace cache is dropped between the arms).
Median of per-step device times, ROUNDS rounds x STEPS steps per arm."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
import bench
from midiemo.models.build_model import build_model
from midiemo.optim import FusedAdamW
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model, _ = build_model(dict(bench.CFG, compute_dtype="bf16"))
model = model.to(dev).train()
model.seed_dropout(1000)
opt = FusedAdamW(model, lr=2e-5, clip=1.0)
batches = [bench.synthetic_batch(bench.CFG, bench.BATCH, bench.SEQ, 1234 + 7919 * i, dev) for i in range(4)]
def step(i):
    tok, cond, tgt = batches[i % 4]
    loss = model.loss_and_backward(tok, cond, tgt)
    opt.step()
    return loss
ROUNDS, STEPS = 6, 12
WHAT = os.environ.get("AB_WHAT", "relu_mask")
ARMS = ("gate", "mask") if WHAT == "relu_mask" else (("serial", "overlap") if WHAT == "attn_overlap" else ("lo16", "lo8"))
res = {a: [] for a in ARMS}
for r in range(ROUNDS):
    for arm in ARMS:
        if WHAT == "relu_mask":
            if arm == "gate": os.environ["MIDIEMO_NO_RELU_MASK"] = "1"
            else: os.environ.pop("MIDIEMO_NO_RELU_MASK", None)
            model._ws.clear()
        elif WHAT == "attn_overlap":
            model.attn_bwd_overlap = arm == "overlap"
        else:                                                   # AB_WHAT=resid_lo: the residual stream's low halves as 16-bit / 8-bit arrays
            model.resid_lo_bits = 8 if arm == "lo8" else 16
            model._ws.clear()
        for i in range(3): step(i)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)]
        evs[0].record()
        for i in range(STEPS):
            loss = step(i); evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        res[arm].append(ts[len(ts) // 2])
        print("round %d %s median %.3f ms  loss %.5f" % (r, arm, ts[len(ts) // 2], float(loss)), flush=True)
for arm, v in res.items(): print(arm, "median of rounds %.3f ms" % sorted(v)[len(v) // 2])
