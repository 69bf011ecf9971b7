"""Interleaved A/B timing of NT GEMM epilogue variants on one shape (round-robin over the variants, several rounds, median):
the matrix pipe is power limited and the clocks drift during a run, so back-to-back blocks of one variant are not comparable."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
M = 32768
r = lambda *s: torch.randn(*s, device=dev).to(dt)
def run(N, K, variants, rounds=12, iters=8):
    A, B, C = r(M, K), r(N, K), torch.empty(M, N, device=dev, dtype=dt)
    bias, addt, gatet = torch.randn(N, device=dev), r(M, N), r(M, N)
    kws = {"plain": {}, "bias": dict(bias=bias), "bias+relu": dict(bias=bias, flags=ops.ME_EPI_RELU), "relu": dict(flags=ops.ME_EPI_RELU),
           "add": dict(add=addt), "gate": dict(gate=gatet, flags=ops.ME_EPI_RELU_BWD)}
    ts = {v: [] for v in variants}
    for v in variants:
        ops.gemm_nt(A, B, C, **kws[v])
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.gemm_nt(A, B, C, **kws[v])
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / iters * 1e3)
    print("N %4d K %4d: " % (N, K) + "  ".join("%s %.1f" % (v, sorted(ts[v])[len(ts[v]) // 2]) for v in variants))
run(512, 2048, ["plain", "bias", "add"])
run(2048, 512, ["plain", "bias", "bias+relu", "relu", "gate"])
run(1536, 512, ["plain", "bias"])
run(512, 512, ["plain", "bias"])
run(512, 1536, ["plain", "add"])
