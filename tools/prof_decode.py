"""Decode step under rocprofv3: N eager KV-cached greedy steps of the headline model (B = 4) starting at context
length T0 (the cache content does not matter for timing).  usage: python tools/prof_decode.py [T0] [N] [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from midiemo import ops  # noqa: E402
from midiemo.decode import DecodeSession  # noqa: E402
from midiemo.models.build_model import build_model  # noqa: E402

T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cd = sys.argv[3] if len(sys.argv) > 3 else "bf16"
torch.manual_seed(0)
model, _ = build_model(dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
                            conditioning="continuous_concat", dropout=0.1, compute_dtype=cd))
model = model.cuda().eval()
cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
sess = DecodeSession(model, 4)
sess.t = T0
tok = torch.full((4,), 5, dtype=torch.long, device="cuda")
picked = torch.empty(4, dtype=torch.long, device="cuda")
with torch.no_grad():
    for _ in range(N):
        lg = sess.step(tok, cond)
        ops.greedy_pick(lg, 1007, None, picked, 4)
        tok = picked.clone()
torch.cuda.synchronize()
print("done", N, "steps from", T0)
