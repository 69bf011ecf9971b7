"""us per launch of the sampling kernel (me_sample_topk_topp: mask, log-softmax, temperature, sort, top-k, nucleus cut, draw),
B = 4 rows of V = 1007 logits -- generate()'s per-token sampling step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
dev = "cuda"
B, V = 4, 1007
torch.manual_seed(0)
logits = torch.randn(B, 1024, device=dev) * 3
temp = torch.full((B,), 1.2, device=dev)
u = torch.rand(B, device=dev)
out = torch.zeros(B, dtype=torch.int64, device=dev)
special = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
for tk, tp in ((0, 0.9), (40, 0.9), (0, 1.0)):
    for _ in range(5): ops.sample_topk_topp(logits, V, special, temp, tk, tp, u, out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.sample_topk_topp(logits, V, special, temp, tk, tp, u, out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("top_k %3d top_p %.2f: %.2f us per launch (graph of 50)  ids %s" % (tk, tp, e0.elapsed_time(e1) * 1e3 / 100, out.tolist()))
