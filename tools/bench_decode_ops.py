"""Per-kernel timing of the decode step (headline model, B = 4, bf16): every fused kernel launched back to back
N times between two events (graph-free; includes the ~1.5 us launch boundary).  usage: python tools/bench_decode_ops.py [t]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from midiemo import ops  # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = torch.bfloat16
dev = "cuda"
B, d, di, H, dh, V, M = 4, 512, 2048, 8, 64, 1007, 2048
ns = int(os.environ.get("MIDIEMO_DEC_NSPLIT", "8"))
r = lambda *s, dtype=dt: torch.randn(*s, device=dev).to(dtype)
Wqkv, Wo, W1, W2, Wf, E = r(3 * d, d), r(d, d), r(di, d), r(d, di), r(V, d), r(M, dh)
bq, bo, b1, b2, bf = r(3 * d, dtype=torch.float32), r(d, dtype=torch.float32), r(di, dtype=torch.float32), r(d, dtype=torch.float32), r(V, dtype=torch.float32)
g, be = r(d, dtype=torch.float32), r(d, dtype=torch.float32)
s_in, xres, s1, o1, s2 = (r(B, d, dtype=torch.float32) for _ in range(5))
q, hid = r(B, d), r(B, di)
kc, vc = r(B, H, M, dh), r(B, H, M, dh)
part = torch.zeros(B * H, ns, dh + 4, device=dev)
logits = torch.zeros(B, V, device=dev)
emb, pe = r(V, d - 128, dtype=torch.float32), r(M, d, dtype=torch.float32)
cw, cb = r(128, 2, dtype=torch.float32), r(128, dtype=torch.float32)
tok = torch.randint(2, V, (B, 1), device=dev)
cond = torch.rand(B, 2, device=dev)
hist = torch.zeros(B, M, dtype=torch.long, device=dev)
pos = torch.zeros(1, dtype=torch.int32, device=dev)
picked = torch.zeros(B, dtype=torch.long, device=dev)

OPS = {
    "embed_qkv": lambda: ops.dec_embed_qkv(tok, cond, emb, cw, cb, pe, 128, Wqkv, bq, xres, q, kc, vc, B, d, H, dh, M, t, None, dt),
    "ln_qkv": lambda: ops.dec_qkv(s_in, g, be, 1e-6, None, None, Wqkv, bq, xres, q, kc, vc, B, d, H, dh, M, t, None, dt),
    "attn": lambda: ops.dec_attn(q, kc, vc, E, None, 0, part, ns, B, H, dh, M, M, t, None, dt),
    "combine_wo": lambda: ops.dec_proj_resid(part, ns, H, dh, None, Wo, bo, xres, s1, B, d, d, dt),
    "ln_ffn1": lambda: ops.dec_ln_proj(s1, g, be, 1e-6, W1, b1, o1, hid, B, di, d, ops.ME_EPI_RELU, dt),
    "ffn2": lambda: ops.dec_proj_resid(None, 0, 0, 0, hid, W2, b2, o1, s2, B, d, di, dt),
    "ln_head": lambda: ops.dec_ln_proj(s2, g, be, 1e-6, Wf, bf, None, logits, B, V, d, ops.ME_EPI_OUT_F32, dt),
    "pick_commit": lambda: ops.greedy_pick_commit(logits, V, None, picked, hist, pos, B),
}
N = 50
for name, fn in OPS.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()                # N back-to-back launches replayed as one graph: device time, no host gaps
    with torch.cuda.graph(gr):
        for _ in range(N):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%-12s %7.2f us per launch (graph of %d back-to-back launches)" % (name, 1e3 * e0.elapsed_time(e1) / (4 * N), N))
