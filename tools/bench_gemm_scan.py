"""NT / TN GEMM time against K (NT) and T (TN) at fixed output size: separates the per-launch fixed cost from the
per-slab rate.  Timed with rocprofv3-independent HIP events over back-to-back launches (launch gaps included)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dt = torch.bfloat16
M = 32768
for N in (512, 2048):
    for K in (64, 128, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
        C = torch.empty(M, N, device="cuda", dtype=dt); bias = torch.randn(N, device="cuda")
        t = timeit(lambda: ops.gemm_nt(A, B, C, bias=bias))
        print("NT M=%d N=%4d K=%4d  %7.1f us  (%d slabs per block)" % (M, N, K, t, (K // 64) * max(1, (M // 256) * (N // 256) // 256)))
for T in (2048, 4096, 8192, 16384, 32768):
    A = torch.randn(T, 512, device="cuda").to(dt); B = torch.randn(T, 512, device="cuda").to(dt)
    dW = torch.zeros(512, 512, device="cuda"); db = torch.zeros(512, device="cuda")
    t = timeit(lambda: ops.gemm_tn_acc(A, B, dW, db))
    print("TN T=%5d N=512 K=512  %7.1f us (kernel + reduce)" % (T, t))
