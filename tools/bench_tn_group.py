"""A layer's four weight gradients as ONE grouped TN launch at C2 (T = 32768, bf16): us per launch (median over rounds) and TF/s.
A/B two builds by alternating MIDIEMO_LIB between processes, or run once."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
T, d, di = 32768, 512, 2048
r = lambda *s: torch.randn(*s, device=dev).to(dt)
shapes = [(3 * d, d), (d, d), (di, d), (d, di)]            # (N, K): dWqkv, dWo, dW1, dW2
items = []
for N, K in shapes:
    items.append((r(T, N), r(T, K), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev), N, K))
need = ops.workspace_bytes(ops.ME_WS_GEMM_TN_GROUP, T, ops.tn_group_tiles(shapes), 0, dt)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
fn = lambda: ops.gemm_tn_acc_group(items, T, dt, ws=ws)
for _ in range(3): fn()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
t = sorted(ts)[len(ts) // 2]
fl = sum(2.0 * T * N * K for N, K in shapes)
print("grouped TN launch (+ reduce): %.1f us  %.0f TF/s" % (t, fl / t / 1e6))
