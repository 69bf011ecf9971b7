import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
d = 512; dt = torch.bfloat16
for T in (256, 4096, 8192, 16384, 32768, 65536, 131072):
    dy = torch.randn(T, d, device="cuda").to(dt); s = torch.randn(T, d, device="cuda").to(dt)
    stats = torch.rand(T, 2, device="cuda") + 0.5; gamma = torch.randn(d, device="cuda")
    dx = torch.empty_like(dy); da = torch.empty_like(dy); dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    t = timeit(lambda: ops.resid_ln_bwd(dy, s, stats, gamma, dx, da, dg, db, T, d, 0.1, 123, 3))
    x = torch.randn(T, d, device="cuda").to(dt); a = torch.randn(T, d, device="cuda").to(dt); y = torch.empty_like(x); so = torch.empty_like(x); beta = torch.randn(d, device="cuda")
    t2 = timeit(lambda: ops.resid_ln_fwd(x, a, gamma, beta, y, so, stats, T, d, 1e-5, 0.1, 123, 3))
    print("T=%6d  ln_bwd %7.1f us   ln_fwd %7.1f us" % (T, t, t2))
