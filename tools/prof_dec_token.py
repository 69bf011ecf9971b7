"""Where the time of me_dec_token goes: 100 MHz stamps of blocks 0 / 100 / 255 (a -DME_TOK_PROF build, tools/build_abl.sh tokprof
"-DME_TOK_PROF" with ONLY=me_decode_token; run with MIDIEMO_LIB=abl_tmp/lib_tokprof.so).  python tools/prof_dec_token.py [t] [dtype]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
from midiemo.decode import DecodeSession            # noqa: E402
from midiemo.models.build_model import build_model  # noqa: E402

NAMES = ["start", "requests issued", "s2 polled", "LN2 done", "qkv published", "q slice polled", "partial published", "group partials polled",
         "att published (S2 start)", "att polled", "s1 published (S3 start)", "s1 polled", "LN1 done", "hid published (S4 start)", "hid polled",
         "s2 published"]
NS = len(NAMES)


def main():
    t_at = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cd = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    B = 4
    torch.manual_seed(0)
    args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
                conditioning="continuous_concat", compute_dtype=cd)
    model, _ = build_model(args)
    model = model.to("cuda").eval()
    cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
    s = DecodeSession(model, B)
    assert s.token_kernel
    s._tok_ws = torch.zeros(s._tok_ws.numel() + (1 << 18), dtype=torch.uint8, device="cuda")
    start = torch.ones(B, dtype=torch.int64, device="cuda")
    s.greedy_run(start, t_at, cond=cond)
    torch.cuda.synchronize()
    s.check_token_status()
    off = 256 + 8 * (4 * 512 * 6 + 4 * 2048) + 8 * (4 * 8 * (1024 + 64 + 2))
    st = s._tok_ws[off:off + 8 * 1536].view(torch.int64).cpu().view(3, 512)
    for bi, blk in enumerate((0, 100, 255)):
        rows = st[bi, :16 * 7].view(7, 16)[:, :NS].double() / 100.0          # us
        t0 = rows[0, 0]
        print(f"block {blk}: token at t = {t_at - 1}: layer starts (us since token start): " + " ".join(f"{float(rows[l, 0] - t0):.2f}" for l in range(7)))
        mid = rows[1:6]                                                     # layers 1..5 (steady state)
        d = (mid[:, 1:] - mid[:, :-1]).mean(0)
        print("   phase (mean of layers 1-5)                us")
        for i in range(NS - 1):
            print(f"   {NAMES[i]:>28s} -> {NAMES[i + 1]:<28s} {float(d[i]):7.2f}")
        print(f"   layer total {float((mid[:, NS - 1] - mid[:, 0]).mean()):.2f} us")


if __name__ == "__main__":
    main()
