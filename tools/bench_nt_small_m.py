"""development aid: NT GEMM at small M (the sliding-window forward: B = 4 x 1024 rows): 256-tile persistent kernel vs the generic 128-tile one (MIDIEMO_NO_NT256=1)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev, dt = "cuda", torch.bfloat16
for M in (1024, 2048, 4096, 8192, 16384):
    out = []
    for name, N, K in [("qkv", 1536, 512), ("proj", 512, 512), ("ffn1", 2048, 512), ("ffn2", 512, 2048), ("head", 1007, 512)]:
        A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
        ld = 1024 if N == 1007 else N
        C = torch.empty(M, ld, device=dev, dtype=dt); bias = torch.randn(N, device=dev)
        out.append("%s %.1f" % (name, timeit(lambda: ops.gemm_nt(A, B, C, bias=bias, N=N))))
    print(os.environ.get("MIDIEMO_NO_NT256", "0"), "M", M, " ".join(out))
