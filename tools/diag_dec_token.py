"""first differing quantity between me_dec_token and the launch chain (development aid): steps both sessions with random tokens and, at the
first step whose logits differ, compares the K / V rows of every layer at that position and the last layer's exchanged values."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
from midiemo.decode import DecodeSession
from midiemo.models.build_model import build_model

cd = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n_layer, H, d, di, dc = int(os.environ.get("NL", "3")), 2, 128, 256, 32
torch.manual_seed(5)
model, _ = build_model(dict(vocab_size=1007, n_layer=n_layer, n_head=H, d_model=d, d_inner=di, dropout=0.0, d_condition=dc,
                            conditioning="continuous_concat", compute_dtype=cd))
model = model.cuda().eval()
B, n = 4, 150
dh = d // H
cond = torch.rand(B, 2, device="cuda") * 2 - 1
toks = torch.randint(2, 1007, (n, B), device="cuda")
os.environ["MIDIEMO_DEC_TOKEN"] = "0"; a = DecodeSession(model, B)
os.environ["MIDIEMO_DEC_TOKEN"] = "1"; b = DecodeSession(model, B)
assert b.token_kernel and a.nsplit == b.nsplit
if os.environ.get("MIDIEMO_LIB"):
    b._tok_ws = torch.zeros(b._tok_ws.numel() + (1 << 19), dtype=torch.uint8, device="cuda")
ns = b.nsplit
with torch.no_grad():
    for i in range(n):
        la, lb = a.step(toks[i], cond).clone(), b.step(toks[i], cond).clone()
        if not torch.equal(la, lb):
            print("first differing step", i, "max abs", float((la - lb).abs().max()))
            for l in range(n_layer):
                print(" layer", l, "k_t equal", torch.equal(a.kc[l][:, :, i], b.kc[l][:, :, i]), "v_t equal", torch.equal(a.vc[l][:, :, i], b.vc[l][:, :, i]))
            ws = b._tok_ws
            rec = ws[256:].view(torch.int64)
            MR = 4
            def f32(off, cnt): return (rec[off:off + cnt] & 0xffffffff).to(torch.int32).view(torch.float32)
            o_s2, o_s1, o_att, o_qkv, o_hid = 0, MR * d, 2 * MR * d, 3 * MR * d, 6 * MR * d
            o_part = o_hid + MR * di
            s2 = f32(o_s2, B * d).view(B, d); s1 = f32(o_s1, B * d).view(B, d)
            print(" s1 differing elements per row:", (s1 != a.s1).sum(-1).tolist(), "cols of first bad row:", (s1 != a.s1)[(s1 != a.s1).any(-1).nonzero()[0, 0]].nonzero().flatten().tolist()[:12])
            print(" last layer s1 equal", torch.equal(s1, a.s1), float((s1 - a.s1).abs().max()), " s2 equal", torch.equal(s2, a.s2), float((s2 - a.s2).abs().max()))
            part = f32(o_part, B * H * ns * (dh + 2)).view(B * H, ns, dh + 2)
            pa = a.part.view(B * H, ns, dh + 4)
            print(" partial max equal", torch.equal(part[:, :, 0], pa[:, :, 0]), " sum equal", torch.equal(part[:, :, 1], pa[:, :, 1]),
                  " o equal", torch.equal(part[:, :, 2:], pa[:, :, 4:]))
            bad = (part[:, :, 2:] != pa[:, :, 4:]).any(-1) | (part[:, :, 0] != pa[:, :, 0]) | (part[:, :, 1] != pa[:, :, 1])
            print(" differing (row*H+head, split):", bad.nonzero().tolist()[:10])
            if cd == "fp32":
                from midiemo import ops
                eye = torch.eye(d, device="cuda"); zb = torch.zeros(d, device="cuda"); zr = torch.zeros(B, d, device="cuda"); out = torch.empty(B, d, device="cuda")
                ops.dec_proj_resid(a.part, ns, H, dh, None, eye, zb, zr, out, B, d, d, torch.float32)
                att = f32(o_att, B * d).view(B, d)
                ne = (att != out)
                print(" att (token records) vs chain combine through an identity projection: differing", int(ne.sum()), ne.nonzero().tolist()[:8],
                      [(float(att[i, j]), float(out[i, j])) for i, j in ne.nonzero().tolist()[:4]])
                print("  differing att columns of that row:", ne[ne.any(-1).nonzero()[0, 0]].nonzero().flatten().tolist())
                pa2 = a.part.view(B * H, ns, dh + 4)
                for i, j in ne.nonzero().tolist()[:2]:
                    mh_ = i * H + j // dh
                    print("  partials of (row, head)", i, j // dh, "max", pa2[mh_, :, 0].tolist(), "sum", pa2[mh_, :, 1].tolist(), "o", pa2[mh_, :, 4 + j % dh].tolist())
                if os.environ.get("TOKDBG"):
                    import numpy as np
                    poff = 256 + 8 * (o_part + 4 * 8 * (1024 + 64 + 2) + 2048)
                    dbg = ws[poff:poff + 256 * 128].view(torch.float32).view(256, 32).cpu().numpy()
                    i0, j0_ = ne.nonzero().tolist()[0]
                    mh_ = i0 * H + j0_ // dh
                    for blk in (mh_ * (ns - 1),):
                        r = dbg[blk]
                        print("  token block", blk, "\n   wn  ", [repr(x) for x in r[0:8]], "\n   max ", [repr(x) for x in r[8:16]], "\n   sum ", [repr(x) for x in r[16:24]], "\n   w0 w1 l inv msafe . . w7", [repr(x) for x in r[24:32]])
                    print("  final records: max", [repr(float(x)) for x in part[mh_, :, 0]], "sum", [repr(float(x)) for x in part[mh_, :, 1]], "o[0]", [repr(float(x)) for x in part[mh_, :, 2]])
                hid = f32(o_hid, B * di).view(B, di)
                print(" hid equal", torch.equal(hid, a.hid))
            break
    else:
        print("no difference in", n, "steps")
