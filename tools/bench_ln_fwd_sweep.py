"""development aid: resid_ln_fwd (hi + lo residual, dropout 0.1, as the train step calls it) under MIDIEMO_LN_R / MIDIEMO_LN_GRID"""
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
x, xl, a, y, yl, so = mk(), mk(), mk(), mk(), mk(), mk()
stats = torch.empty(T, 2, device="cuda"); gamma = torch.randn(d, device="cuda"); beta = torch.randn(d, device="cuda")
# rotate over several buffer sets so that nothing stays in the 256 MB Infinity Cache between iterations
sets = [(mk(), mk(), mk(), mk(), mk(), mk()) for _ in range(6)]
i = [0]
def run():
    x, xl, a, y, yl, so = sets[i[0] % len(sets)]; i[0] += 1
    ops.resid_ln_fwd(x, a, gamma, beta, y, so, stats, T, d, 1e-5, 0.1, 123, 3, x_lo=xl, y_lo=yl)
t = timeit(run)
print("R %s grid %s: resid_ln_fwd %.1f us (%.2f TB/s)" % (os.environ.get("MIDIEMO_LN_R", "-"), os.environ.get("MIDIEMO_LN_GRID", "-"), t, 6 * T * d * 2 / t / 1e6))
