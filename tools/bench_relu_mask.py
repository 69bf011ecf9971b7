"""Interleaved A/B: FFN_pre forward and FFN_suf dgrad with the gate operand (me_gemm_nt) against the ReLU sign mask
(me_gemm_nt_relu_mask), headline shape M = 32768, d = 512, d_inner = 2048, bf16.  us per launch, median of 12 rounds x 8."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev, dt = "cuda", torch.bfloat16
M, N, K = 32768, 2048, 512
r = lambda *s: torch.randn(*s, device=dev).to(dt)
A, W1, bias = r(M, K), r(N, K), torch.randn(N, device=dev)
dC, W2T = r(M, K), r(N, K)
hid, hid2, d1, d2 = (torch.empty(M, N, device=dev, dtype=dt) for _ in range(4))
mask = torch.zeros(ops.workspace_bytes(ops.ME_WS_RELU_MASK, M, N, K, dt), dtype=torch.uint8, device=dev)
cases = {
    "fwd  bias+relu          ": lambda: ops.gemm_nt(A, W1, hid, bias=bias, flags=ops.ME_EPI_RELU),
    "fwd  bias+relu+mask out ": lambda: ops.gemm_nt_relu_mask(A, W1, hid2, mask, bias=bias),
    "bwd  gate = activations ": lambda: ops.gemm_nt(dC, W2T, d1, gate=hid, flags=ops.ME_EPI_RELU_BWD),
    "bwd  gate = sign mask   ": lambda: ops.gemm_nt_relu_mask(dC, W2T, d2, mask, backward=True),
    "     plain (no gate)    ": lambda: ops.gemm_nt(dC, W2T, d1),
}
for f in cases.values(): f()
torch.cuda.synchronize()
assert torch.equal(hid.view(torch.int16), hid2.view(torch.int16))
cases["bwd  gate = activations "](); cases["bwd  gate = sign mask   "](); torch.cuda.synchronize()
assert torch.equal(d1.view(torch.int16), d2.view(torch.int16))
ts = {k: [] for k in cases}
for _ in range(12):
    for k, f in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): f()
        e1.record(); torch.cuda.synchronize()
        ts[k].append(e0.elapsed_time(e1) / 8 * 1e3)
for k, v in ts.items(): print("%s %.1f us" % (k, sorted(v)[len(v) // 2]), flush=True)
