"""NT GEMM at the ffn2 shape (K = 2048) with padded operand row strides: is the 4 KB row stride the problem?"""
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
M, N, K = 32768, 512, 2048
for pa, pb in [(0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (8, 8)]:
    A = torch.randn(M, K + pa, device=dev).to(dt)[:, :K]
    B = torch.randn(N, K + pb, device=dev).to(dt)[:, :K]
    C = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    t = timeit(lambda: ops.gemm_nt(A, B, C, bias=bias))
    print("lda = K + %3d, ldb = K + %3d: %7.1f us  %7.1f TF" % (pa, pb, t, 2.0 * M * N * K / t / 1e6))
