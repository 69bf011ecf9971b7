"""Every NT GEMM launch of one layer's forward + backward at C2 (M = 32768, bf16) with ITS epilogue (bias / ReLU / residual add /
ReLU gate), us per launch and TF/s: which launches sit far from their plain twin?"""
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
M, d, di = 32768, 512, 2048
r = lambda *s: torch.randn(*s, device=dev).to(dt)
cases = [
    ("fwd qkv      N1536 K512  bias", 1536, 512, dict(bias=True)),
    ("fwd Wo       N512  K512  bias", 512, 512, dict(bias=True)),
    ("fwd FFN_pre  N2048 K512  bias+relu", 2048, 512, dict(bias=True, relu=True)),
    ("fwd FFN_suf  N512  K2048 bias", 512, 2048, dict(bias=True)),
    ("bwd dC.W2T   N2048 K512  relu-gate", 2048, 512, dict(gate=True)),
    ("bwd dhid.W1T N512  K2048 +add", 512, 2048, dict(add=True)),
    ("bwd dC2.WoT  N512  K512  plain", 512, 512, dict()),
    ("bwd dqkv.WqkvT N512 K1536 +add", 512, 1536, dict(add=True)),
    ("    plain    N2048 K512", 2048, 512, dict()),
    ("    plain    N512  K2048", 512, 2048, dict()),
    ("    plain    N512  K1536", 512, 1536, dict()),
]
for name, N, K, e in cases:
    A, B, C = r(M, K), r(N, K), torch.empty(M, N, device=dev, dtype=dt)
    kw = {}
    if e.get("bias"): kw["bias"] = torch.randn(N, device=dev)
    if e.get("add"): kw["add"] = r(M, N)
    if e.get("gate"): kw["gate"] = r(M, N); kw["flags"] = ops.ME_EPI_RELU_BWD
    if e.get("relu"): kw["flags"] = ops.ME_EPI_RELU
    t = timeit(lambda: ops.gemm_nt(A, B, C, **kw))
    print("%-36s %7.1f us %7.1f TF" % (name, t, 2.0 * M * N * K / t / 1e6))
