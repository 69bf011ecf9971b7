"""me_dec_token (one persistent launch per token) against the per-stage launch chain on the headline model (6L d512 8H dh64
d_inner 2048, B = 4): logits of every step bit for bit (eager, teacher-forced with the chain's greedy ids), the graph-replayed
greedy streams, and the step time of both (events around graph replays).   python tools/check_dec_token.py [bf16 fp16 fp32] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
from midiemo.decode import DecodeSession            # noqa: E402
from midiemo.models.build_model import build_model  # noqa: E402


def session(model, B, token):
    os.environ["MIDIEMO_DEC_TOKEN"] = "1" if token else "0"
    s = DecodeSession(model, B)
    assert s.token_kernel == token, (s.token_kernel, token)
    return s


def main():
    dts = [a for a in sys.argv[1:] if not a.isdigit()] or ["bf16"]
    steps = int(next((a for a in sys.argv[1:] if a.isdigit()), 256))
    B = 4
    for cd in dts:
        torch.manual_seed(0)
        args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
                    conditioning="continuous_concat", compute_dtype=cd)
        model, _ = build_model(args)
        model = model.to("cuda").eval()
        cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
        start = torch.ones(B, dtype=torch.int64, device="cuda")
        a, b = session(model, B, False), session(model, B, True)
        # eager, teacher-forced: same input tokens, compare the logits of every step
        tok = start.clone()
        worst, nbad = 0.0, 0
        for t in range(steps):
            la = a.step(tok, cond).clone()
            lb = b.step(tok, cond).clone()
            if not torch.equal(la, lb):
                nbad += 1
                worst = max(worst, float((la - lb).abs().max()))
            tok = la.argmax(-1)
        b.check_token_status()
        kv_equal = all(torch.equal(x, y) for x, y in zip(a.kc + a.vc, b.kc + b.vc))
        print(f"{cd}: eager {steps} steps: logits differ in {nbad} steps (max abs {worst:.3e}); K/V caches equal: {kv_equal}", flush=True)
        # graph-replayed greedy streams + timing
        res = {}
        for name, tk in (("chain", False), ("token", True), ("chain2", False), ("token2", True)):
            s = session(model, B, tk)
            s.greedy_run(start, 8, cond=cond)                       # capture
            s.reset()
            torch.cuda.synchronize()
            ev = []
            t0 = time.perf_counter()
            ids = s.greedy_run(start, steps, cond=cond, step_events=ev)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            s.check_token_status()
            dts_ = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8, len(ev) - 1))
            res[name] = (ids.clone(), dts_[len(dts_) // 2], wall / steps * 1e3)
            print(f"{cd}: {name}: p50 {res[name][1]:.4f} ms/step, wall {res[name][2]:.4f} ms/step, launches/token {s.launches_per_token}", flush=True)
        print(f"{cd}: greedy ids equal: {torch.equal(res['chain'][0], res['token'][0])}; checksum {int(res['token'][0].sum())}", flush=True)


if __name__ == "__main__":
    main()
