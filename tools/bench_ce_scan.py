"""Time of the cross-entropy kernels against the row count: separates fixed cost from streaming rate."""
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
V, ld = 1007, 1024
for T in (256, 4096, 16384, 32768, 65536, 131072):
    logits = torch.randn(T, ld, device="cuda"); tgt = torch.randint(2, V, (T,), device="cuda")
    lse = torch.empty(T, device="cuda"); acc = torch.zeros(2, device="cuda"); dl = torch.empty(T, ld, device="cuda", dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.ce_fwd(logits, tgt, lse, acc[0:1], acc[1:2], T, V, 0))
    t2 = timeit(lambda: ops.ce_bwd(logits, tgt, lse, dl, acc[1:2], 1.0, T, V, 0))
    g = torch.randn(T * 600, device="cuda"); out = torch.zeros(1, device="cuda")
    t3 = timeit(lambda: ops.sumsq(g, out)) if hasattr(ops, "sumsq") else float("nan")
    print("T=%6d  ce_fwd %7.1f us  ce_bwd %7.1f us   sumsq(%d MB) %7.1f us" % (T, t1, t2, T * 600 * 4 >> 20, t3))
