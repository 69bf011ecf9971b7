"""Do the latency-bound attention backward kernels overlap when two independent backward passes are issued on two
streams?  (development aid: potential of a horizontally fused bwd-q / kv launch)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = 32, 1024, 8, 64, 2048
def mk():
    qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt); E = torch.randn(M, dh, device=dev).to(dt); Epk = ops.rga_pack_rel(E)
    out = torch.empty(B, L, H, dh, device=dev, dtype=dt); lse = torch.empty(B, H, L, device=dev)
    kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
    PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
    ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT, MT=MT)
    dout = torch.randn(B, L, H, dh, device=dev).to(dt); dqkv = torch.empty_like(qkv); dE = torch.zeros(M, dh, device=dev)
    delta = torch.empty(B, H, L, device=dev); dGT = ops.rga_bwd_workspace(B, H, L, dt, dev)
    return lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dGT, B, L, L, H, dh, M)
a, b = mk(), mk()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def serial(): a(); b()
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a()
    with torch.cuda.stream(s2): b()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("two backward passes, one stream : %.1f us" % timeit(serial))
print("two backward passes, two streams: %.1f us" % timeit(par))
