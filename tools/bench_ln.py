import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
T, d = 32768, 512
dt = torch.bfloat16
dy = torch.randn(T, d, device="cuda").to(dt); s = torch.randn(T, d, device="cuda").to(dt)
stats = torch.rand(T, 2, device="cuda") + 0.5; gamma = torch.randn(d, device="cuda")
dx = torch.empty_like(dy); da = torch.empty_like(dy); dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
x = torch.randn(T, d, device="cuda").to(dt); a = torch.randn(T, d, device="cuda").to(dt); y = torch.empty_like(x); so = torch.empty_like(x)
beta = torch.randn(d, device="cuda")
t = timeit(lambda: ops.resid_ln_bwd(dy, s, stats, gamma, dx, da, dg, db, T, d, 0.1, 123, 3))
print("%s resid_ln_bwd %.1f us  (%.2f TB/s)" % (os.environ.get("TAG", ""), t, 4 * T * d * 2 / t / 1e6))
t = timeit(lambda: ops.resid_ln_fwd(x, a, gamma, beta, y, so, stats, T, d, 1e-5, 0.1, 123, 3))
print("%s resid_ln_fwd %.1f us  (%.2f TB/s)" % (os.environ.get("TAG", ""), t, 4 * T * d * 2 / t / 1e6))
t = timeit(lambda: ops.resid_ln_bwd(dy, s, stats, gamma, dx, da, dg, db, T, d, 0.0, 123, 3))
print("%s resid_ln_bwd p=0 %.1f us  (%.2f TB/s)" % (os.environ.get("TAG", ""), t, 4 * T * d * 2 / t / 1e6))
t = timeit(lambda: ops.resid_ln_fwd(x, a, gamma, beta, y, so, stats, T, d, 1e-5, 0.0, 123, 3))
print("%s resid_ln_fwd p=0 %.1f us  (%.2f TB/s)" % (os.environ.get("TAG", ""), t, 4 * T * d * 2 / t / 1e6))
c = torch.empty_like(dy)
t = timeit(lambda: c.copy_(dy))
print("torch copy 33.5 MB -> 33.5 MB %.1f us (%.2f TB/s r+w)" % (t, 2 * T * d * 2 / t / 1e6))
big = torch.randn(4 * T, d, device="cuda").to(dt); big2 = torch.empty_like(big)
t = timeit(lambda: big2.copy_(big))
print("torch copy 134 MB -> 134 MB %.1f us (%.2f TB/s r+w)" % (t, 2 * 4 * T * d * 2 / t / 1e6))
