import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "midi-emotion_amd"))
import torch
from midiemo import ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
T, V, ld = 32768, 1007, 1024
LT = torch.bfloat16 if os.environ.get("LOGITS", "bf16") == "bf16" else torch.float32
logits = torch.randn(T, ld, device="cuda").to(LT); tgt = torch.randint(2, V, (T,), device="cuda")
lse = torch.empty(T, device="cuda"); acc = torch.zeros(2, device="cuda"); dl = torch.empty(T, ld, device="cuda", dtype=torch.bfloat16)
t = timeit(lambda: ops.ce_fwd(logits, tgt, lse, acc[0:1], acc[1:2], T, V, 0))
print("%s ce_fwd %.1f us (%.2f TB/s)" % (os.environ.get("TAG", ""), t, T * ld * logits.element_size() / t / 1e6))
t = timeit(lambda: ops.ce_bwd(logits, tgt, lse, dl, acc[1:2], 1.0, T, V, 0))
print("%s ce_bwd %.1f us (%.2f TB/s)" % (os.environ.get("TAG", ""), t, T * ld * (logits.element_size() + 2) / t / 1e6))
