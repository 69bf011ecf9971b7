"""NT GEMM time model probe: per-slab rate vs K and N (M = 32768, bf16).  slab = (t - 4 - 6.5 * tiles_per_cu) / (tiles_per_cu * K / 64)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev, dt = "cuda", torch.bfloat16
M = 32768
for N, K in [(512, 512), (512, 1024), (512, 2048), (512, 4096), (1024, 2048), (2048, 2048), (2048, 512), (2048, 1024)]:
    A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
    C = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
    t = timeit(lambda: ops.gemm_nt(A, B, C, bias=bias))
    tiles = (M // 256) * (N // 256) / 256.0
    slab = (t - 4 - 6.5 * tiles) / (tiles * K / 64)
    print("N %4d K %4d: %7.1f us %7.1f TF  tiles/CU %.0f  ~%.2f us per slab" % (N, K, t, 2.0 * M * N * K / t / 1e6, tiles, slab))
