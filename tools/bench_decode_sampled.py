"""Sampled KV-cached decode (generate.py's default: temperature + repeat penalty + nucleus), B = 4 x 2048 tokens, bf16:
tokens/s and ms per token of DecodeSession.sample_run (one HIP graph per token)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from midiemo.decode import DecodeSession
from midiemo.models.build_model import build_model
from midiemo.vocab import get_maps, special_token_ids
torch.manual_seed(0)
model, _ = build_model(dict(bench.CFG, compute_dtype="bf16"))
model = model.cuda().eval()
B, N = 4, 2048
cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
specials = torch.tensor(special_token_ids(get_maps()), dtype=torch.int32, device="cuda")
tok0 = torch.full((B,), 1, dtype=torch.long, device="cuda")
V = model.vocab_size
is_ts = torch.zeros(V, dtype=torch.uint8); is_ts[200:300] = 1
u = torch.rand(N, B)
sess = DecodeSession(model, B)
with torch.no_grad():
    for rep in range(2):
        sess.reset()
        rc = torch.zeros(B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids = sess.sample_run(tok0, N, cond, specials, is_ts, rc, 1.2, 1.2, 0.05, 0, 0.9, u.cuda())
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        print("run %d: %.1f tokens/s, %.4f ms per token, ids checksum %d" % (rep, B * N / wall, 1e3 * wall / N, int(ids.sum())), flush=True)
