"""A/B: attention backward with its three kernels in a row on one stream against the key-owned (dK, dV) and the E-row-owned
(dE) kernels side by side on two streams (me_rga_bwd_phases).  B32 L1024 H8 dh64 bf16; us per backward, interleaved rounds."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = int(os.environ.get("AB_B", 32)), int(os.environ.get("AB_L", 1024)), 8, 64, 2048
Lp = ((L + 31) // 32) * 32
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt)
E = torch.randn(M, dh, device=dev).to(dt)
Epk = ops.rga_pack_rel(E)
out = torch.empty(B, L, H, dh, device=dev, dtype=dt)
lse = torch.empty(B, H, L, device=dev)
dout = torch.randn(B, L, H, dh, device=dev).to(dt)
kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
delta = torch.empty(B, H, L, device=dev)
ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT, MT=MT)
res = {}
for ov in (False, True, 2):
    dqkv = torch.zeros_like(qkv); dE = torch.zeros(M, dh, device=dev)
    ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, Lp, H, dh, M, overlap=ov)
    torch.cuda.synchronize()
    res[ov] = (dqkv.clone(), dE.clone())
for ov in (True, 2):
    print("overlap=%s dqkv identical:" % ov, torch.equal(res[False][0].view(torch.int16), res[ov][0].view(torch.int16)),
          " dE max rel diff: %.2e" % float((res[False][1] - res[ov][1]).abs().max() / res[False][1].abs().max()))
dqkv = torch.zeros_like(qkv); dE = torch.zeros(M, dh, device=dev)
ts = {False: [], True: [], 2: []}
for r in range(8):
    for ov in (False, True, 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(6): ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, Lp, H, dh, M, overlap=ov)
        e1.record(); torch.cuda.synchronize()
        ts[ov].append(e0.elapsed_time(e1) / 6 * 1e3)
for ov in (False, True, 2): print("overlap=%s  median %.1f us   (rounds: %s)" % (ov, sorted(ts[ov])[4], " ".join("%.0f" % x for x in ts[ov])))
