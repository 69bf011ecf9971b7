"""Feasibility probe (round 4): can a weight-gradient GEMM with a SMALL per-CU footprint run beside the attention backward
(MFMA utilisation 11 %, bound by memory latency) on a second stream and hide under it?  The 256 x 256 persistent GEMMs own a
CU (160 KB of LDS, 8 waves x 244 registers), so they can only take turns with other kernels; the generic 128 x 128 kernels
(4 waves, ~37 KB of LDS) can share one.  Measures, at C2 sizes: attention backward alone, the generic TN kernel alone (K not a
multiple of 256 selects it), the 256-tile TN kernel alone, and attention backward + generic TN issued on two streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = 32, 1024, 8, 64, 2048
T = B * L
torch.manual_seed(0)
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt)
E = torch.randn(M, dh, device=dev).to(dt)
Epk = ops.rga_pack_rel(E)
out = torch.empty(B, L, H, dh, device=dev, dtype=dt)
lse = torch.empty(B, H, L, device=dev)
dout = torch.randn(B, L, H, dh, device=dev).to(dt)
dqkv = torch.empty_like(qkv)
dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dGT = ops.rga_bwd_workspace(B, H, L, dt, dev)
kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT, MT=MT)
attn = lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dGT, B, L, L, H, dh, M)

def tn_problem(N, K):
    dY = torch.randn(T, N, device=dev).to(dt); X = torch.randn(T, K, device=dev).to(dt)
    dW = torch.zeros(N, K, device=dev)
    nb = ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N, K, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev) if nb else None
    return lambda: ops.gemm_tn_acc(dY, X, dW, ws=ws), 2.0 * T * N * K
tn128, f128 = tn_problem(512, 2048 + 128)       # generic 128 x 128 transpose-read kernel (atomics)
tn256, f256 = tn_problem(512, 2048)             # persistent 256 x 256 kernel

def wall(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

ta, t128, t256 = wall(attn), wall(tn128), wall(tn256)
print("attention backward alone        %8.1f us" % ta)
print("TN generic 128-tile alone       %8.1f us  (%.0f TF)" % (t128, f128 / t128 / 1e6))
print("TN persistent 256-tile alone    %8.1f us  (%.0f TF)" % (t256, f256 / t256 / 1e6))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for n_tn in (1, 2, 3):
    def both():
        with torch.cuda.stream(sa): attn()
        with torch.cuda.stream(sb):
            for _ in range(n_tn): tn128()
    t = wall(both)
    print("attention bwd || %d x TN generic  %8.1f us   (serial sum %.1f; with the 256-tile kernel in line %.1f)" %
          (n_tn, t, ta + n_tn * t128, ta + n_tn * t256 * f128 / f256))
