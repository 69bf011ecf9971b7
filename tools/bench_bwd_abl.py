import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=int(os.environ.get("ITERS", 10)), warm=int(os.environ.get("WARM", 3))):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = 32, 1024, 8, 64, 2048
Lp = L
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt); E = torch.randn(M, dh, device=dev).to(dt); Epk = ops.rga_pack_rel(E)
out = torch.randn(B, L, H, dh, device=dev).to(dt); lse = torch.randn(B, H, L, device=dev).abs() + 5
dout = torch.randn(B, L, H, dh, device=dev).to(dt); dqkv = torch.empty_like(qkv); dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev); kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
t = timeit(lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, Lp, H, dh, M))
print("%s rga_bwd total %.1f us" % (os.environ.get("TAG", ""), t))
