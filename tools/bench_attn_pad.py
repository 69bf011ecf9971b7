"""development aid: attention forward (training instantiation) + backward with trailing PAD keys as real batches have them"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = 32, 1024, 8, 64, 2048
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt); E = torch.randn(M, dh, device=dev).to(dt); Epk = ops.rga_pack_rel(E)
out = torch.empty(B, L, H, dh, device=dev, dtype=dt); lse = torch.empty(B, H, L, device=dev)
dout = torch.randn(B, L, H, dh, device=dev).to(dt); dqkv = torch.empty_like(qkv); dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
g = torch.Generator().manual_seed(3)
for name, frac_rows, max_pad in (("no PAD", 0.0, 0), ("25 % of the rows, up to 30 % PAD", 0.25, 0.3), ("every row, up to 50 % PAD", 1.0, 0.5)):
    kp = torch.zeros(B, L, dtype=torch.uint8)
    for b in range(B):
        if torch.rand(1, generator=g).item() < frac_rows:
            n = int(torch.rand(1, generator=g).item() * max_pad * L)
            if n: kp[b, -n:] = 1
    kpd = kp.to(dev) if frac_rows else None
    tf = timeit(lambda: ops.rga_fwd(qkv, Epk, kpd, out, lse, B, L, H, dh, M, PT=PT, MT=MT))
    tb = timeit(lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, L, H, dh, M))
    print("%-36s fwd %.1f us  bwd %.1f us" % (name, tf, tb))
