"""development aid: full-batch NT GEMMs on 256 CUs (one stream) vs two half-batch chains on two streams with 128-CU
persistent grids (MIDIEMO_CU_RESERVE=128): do the epilogue write bursts of one chain hide under the other's main loops?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch
from midiemo import ops
dt, dev = torch.bfloat16, "cuda"
half = os.environ.get("MIDIEMO_CU_RESERVE") == "128"
T = 32768
shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048)]
def mk(M):
    return [(torch.randn(M, K, device=dev).to(dt), torch.randn(N, K, device=dev).to(dt), torch.empty(M, N, device=dev, dtype=dt),
             torch.randn(N, device=dev)) for (N, K) in shapes]
def chain(ts, reps):
    for _ in range(reps):
        for A, B, C, b in ts:
            ops.gemm_nt(A, B, C, bias=b)
reps = 10
if not half:
    ts = mk(T)
    chain(ts, 2); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(ts, reps); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("one stream, 256 CUs, M=%d: %.1f us per layer-forward GEMM set" % (T, (t1 - t0) / reps * 1e6))
else:
    ta, tb = mk(T // 2), mk(T // 2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def both(r):
        for _ in range(r):
            for i in range(len(shapes)):
                with torch.cuda.stream(sa):
                    A, B, C, b = ta[i]; ops.gemm_nt(A, B, C, bias=b)
                with torch.cuda.stream(sb):
                    A, B, C, b = tb[i]; ops.gemm_nt(A, B, C, bias=b)
    both(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); both(reps); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("two streams, 128 CUs each, M=%d each: %.1f us per layer-forward GEMM set" % (T // 2, (t1 - t0) / reps * 1e6))
    with torch.cuda.stream(sa):
        chain(ta, 2); torch.cuda.synchronize()
        t0 = time.perf_counter(); chain(ta, reps); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("one stream alone, 128 CUs, M=%d: %.1f us" % (T // 2, (t1 - t0) / reps * 1e6))
