"""development aid: resid_ln_bwd (dropout 0.1) under MIDIEMO_LNB_R / MIDIEMO_LNB_GRID; buffer sets rotate (nothing stays cached)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
T, d = 32768, 512
dt = torch.bfloat16
mk = lambda: torch.randn(T, d, device="cuda").to(dt)
stats = torch.rand(T, 2, device="cuda") + 0.5; gamma = torch.randn(d, device="cuda")
dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
sets = [(mk(), mk(), mk(), mk()) for _ in range(8)]
i = [0]
def run():
    dy, s, dx, da = sets[i[0] % len(sets)]; i[0] += 1
    ops.resid_ln_bwd(dy, s, stats, gamma, dx, da, dg, db, T, d, 0.1, 123, 3)
t = timeit(run)
print("R %s grid %s: resid_ln_bwd %.1f us (%.2f TB/s)" % (os.environ.get("MIDIEMO_LNB_R", "-"), os.environ.get("MIDIEMO_LNB_GRID", "-"), t, 4 * T * d * 2 / t / 1e6))
