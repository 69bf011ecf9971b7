"""Decode benchmark (BASELINE configs[4]): KV-cached greedy decode, 4 (valence, arousal) pairs x 2048 tokens,
headline model (random init), one GPU.  Prints tokens/s and p50/p90 per-step latency (host wall per token)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from midiemo import ops  # noqa: E402
from midiemo.decode import DecodeSession  # noqa: E402
from midiemo.models.build_model import build_model  # noqa: E402
from midiemo.vocab import get_maps, special_token_ids  # noqa: E402


def main():
    gen_len = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    cd = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    torch.manual_seed(0)
    args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
                conditioning="continuous_concat", dropout=0.1, compute_dtype=cd)
    model, _ = build_model(args)
    model = model.cuda().eval()
    B = 4
    cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
    specials = torch.tensor(special_token_ids(get_maps()), dtype=torch.int32, device="cuda")
    sess = DecodeSession(model, B)
    tok = torch.full((B,), 1, dtype=torch.long, device="cuda")
    picked = torch.empty(B, dtype=torch.long, device="cuda")
    lat = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    with torch.no_grad():
        for i in range(gen_len):
            t0 = time.perf_counter()
            logits = sess.step(tok, cond)
            ops.greedy_pick(logits, 1007, specials, picked, B)
            tok = picked.clone()
            if i % 64 == 63:
                torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    lat = np.array(lat) * 1e3
    print("decode[%s] eager launches : B=%d gen_len=%d  %.1f tok/s  total %.2f s  per-step host ms p50 %.3f p90 %.3f" %
          (cd, B, gen_len, B * gen_len / total, total, np.percentile(lat, 50), np.percentile(lat, 90)))
    eager_ids = None
    # device-resident greedy loop replayed as one HIP graph per token
    sess2 = DecodeSession(model, B)
    tok0 = torch.full((B,), 1, dtype=torch.long, device="cuda")
    with torch.no_grad():
        sess2.greedy_run(tok0, 8, cond, specials)                 # capture + warm-up
        sess2.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_all = time.perf_counter()
        e0.record()
        ids = sess2.greedy_run(tok0, gen_len, cond, specials)
        e1.record()
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
    print("decode[%s] HIP-graph loop : B=%d gen_len=%d  %.1f tok/s  total %.3f s  per-step device ms %.4f  (ids checksum %d)" %
          (cd, B, gen_len, B * gen_len / total, total, e0.elapsed_time(e1) / gen_len, int(ids.sum().item())))


if __name__ == "__main__":
    main()
