"""In-step duration of every NT GEMM call of the train step (bench.py's workload), grouped by (N, K, write-out): HIP events
around each launch inside live steps -- the operands come from wherever the previous kernel left them, not from a replay's
warm caches.  Run once per MIDIEMO_NT_MAINLOOP setting (the switch is read when the library loads)."""
import os, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
import bench
from midiemo import ops
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
    model.loss_and_backward(tok, cond, tgt); opt.step()
for i in range(4): step(i)
rec = collections.defaultdict(list)
orig = {"gemm_nt": ops.gemm_nt, "gemm_nt_relu_mask": ops.gemm_nt_relu_mask}
def make(name):
    fn = orig[name]
    def nt(A, B, C, *rest, **kw):
        n = B.shape[0] if kw.get("N") is None else kw["N"]; k = A.shape[1] if kw.get("K") is None else kw["K"]
        tag = name[8:] + ("+bias" if kw.get("bias") is not None else "") + ("+add" if kw.get("add") is not None else "") + \
              ("+gate" if kw.get("gate") is not None else "") + ("+bwd" if kw.get("backward") else "") + (" f%d" % kw["flags"] if kw.get("flags") else "")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(A, B, C, *rest, **kw); e1.record()
        rec[(n, k, tag)].append((e0, e1))
    return nt
for name in orig: setattr(ops, name, make(name))
for i in range(8): step(i)
torch.cuda.synchronize()
tot = 0.0
for key in sorted(rec):
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in rec[key])
    per_step = len(us) // 8
    tot += us[len(us) // 2] * per_step
    print("N%5d K%5d %-22s x%2d  median %6.1f us  (min %6.1f)" % (key[0], key[1], key[2], per_step, us[len(us) // 2], us[0]))
print("sum of medians per step: %.3f ms   MIDIEMO_NT_MAINLOOP=%s" % (tot / 1e3, os.environ.get("MIDIEMO_NT_MAINLOOP", "default(3)")))
