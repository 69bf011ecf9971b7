"""development aid: embedding backward (table gradient + condition projection) at the headline shape, uniform and skewed token ids"""
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
B, L, d, dc, V = 32, 1024, 512, 128, 1007
dt, dev = torch.bfloat16, "cuda"
dout = torch.randn(B, L, d, device=dev).to(dt)
cond = torch.rand(B, 2, device=dev) * 2 - 1
g_emb = torch.zeros(V, d - dc, device=dev); g_cw = torch.zeros(dc, 2, device=dev); g_cb = torch.zeros(dc, device=dev)
g = torch.Generator().manual_seed(1)
uni = torch.randint(1, V, (B, L), generator=g).to(dev)
w = 1.0 / torch.arange(1, V, dtype=torch.float64) ** 1.1                    # Zipf-like: a few tokens thousands of times
skew = (torch.multinomial(w, B * L, replacement=True, generator=g) + 1).view(B, L).to(dev)
ws = ops.embed_bwd_ws(dev)
for name, tok in (("uniform", uni), ("skewed", skew)):
    for w in (None, ws):
        t = timeit(lambda: ops.embed_bwd(dout, tok, cond, g_emb, g_cw, g_cb, None, None, ops.ME_COND_CONCAT, B, L, d, dc, 0, 0.1, 5, ws=w))
        print("%s embed_bwd (all kernels) %s tokens, %s: %.1f us (max count %d)" % (os.environ.get("TAG", ""), name, "workspace" if w is not None else "no workspace", t, int(torch.bincount(tok.flatten()).max())))
