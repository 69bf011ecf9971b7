"""development aid: weight-gradient GEMM with and without the fused bias column sums."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch
from midiemo import ops
from bench_kernels import timeit
T, dt, dev = 32768, torch.bfloat16, "cuda"
for (N_, K_, tag) in [(1536, 512, "dWqkv"), (512, 512, "dWo"), (2048, 512, "dW1"), (512, 2048, "dW2")]:
    A = torch.randn(T, N_, device=dev).to(dt)
    X = torch.randn(T, K_, device=dev).to(dt)
    dW = torch.zeros(N_, K_, device=dev)
    db = torch.zeros(N_, device=dev)
    need = ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N_, K_, dt)
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
    t1 = timeit(lambda: ops.gemm_tn_acc(A, X, dW, db, ws=ws), 20)
    t0 = timeit(lambda: ops.gemm_tn_acc(A, X, dW, None, ws=ws), 20)
    print("gemm_tn %-5s with bias %7.1f us   without %7.1f us" % (tag, t1, t0))
