"""How much does ONE bf16 weight moving to its neighbouring bf16 value change the gradients?  (DDP diagnosis, round 4:
free-running 2-rank and 1-rank trajectories separate by 1e-3 in the step-2 gradient after exactly one such flip.)
Headline model, B = 4 x L = 256; the weight enc_layers.5.FFN_suf.weight[i, j] is replaced by the next bf16 value;
gradient change per layer and tensor family for  (a) the HIP bf16 tier, (b) the HIP f32 tier (weight moved by the same
amount), (c) the oracle on the host under torch.autocast(bfloat16) -- the reference's own --amp arithmetic.
usage (GPU box): python tools/diag_sensitivity.py > gpurun_out/r04/diag_sensitivity.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "midi-emotion_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import ddp_worker as W  # noqa: E402
from oracle import ref_model as O  # noqa: E402

dev = torch.device("cuda", 0)
NAME = "enc_layers.5.FFN_suf.weight"


def batch():
    parts = [W.micro_batch(1, 0, r, torch.device("cpu")) for r in range(2)]
    return tuple(torch.cat([p[i] for p in parts]) for i in range(3))


def next_bf16(x):
    b = x.bfloat16()
    i = b.view(torch.int16) + 1               # sign-magnitude format: +1 on the bit pattern = one step away from zero
    return i.view(torch.bfloat16).float()


def per_tensor(ga, gb, names):
    out = {}
    for k in names:
        a, b = ga[k].double().flatten(), gb[k].double().flatten()
        out[k] = float((a - b).norm() / a.norm().clamp_min(1e-300))
    return out


def show(tag, d):
    print(tag)
    by_layer = {}
    for k, v in d.items():
        m = re.match(r"enc_layers\.(\d+)\.(.*)", k)
        if m:
            by_layer.setdefault(int(m.group(1)), {})[m.group(2)] = v
        else:
            print("    %-28s %.2e" % (k, v))
    for i in sorted(by_layer):
        print("    L%d  " % i + "  ".join("%s %.1e" % (k.replace("rga.", "").replace("weight", "w").replace("bias", "b").replace("layernorm", "ln"), v)
                                          for k, v in by_layer[i].items() if not k.endswith("Wk.bias")))


def hip_grads(dtype, sd):
    m = W.build(dtype, dev)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    x, c, y = batch()
    m.flat_grads.zero_()
    m.loss_and_backward(x.to(dev), c.to(dev), y.to(dev))
    m.link_grads()
    return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}


def main():
    W.use_big(True)
    m0 = W.build("bf16", dev)
    sd = {k: v.detach().cpu().clone() for k, v in m0.state_dict().items()}
    names = list(sd.keys())
    sd2 = {k: v.clone() for k, v in sd.items()}
    w = sd2[NAME]
    # pick the entry with the largest magnitude in row 0 (any entry will do) and move it one bf16 step
    j = int(w[0].abs().argmax())
    old = float(w[0, j])
    w[0, j] = next_bf16(w[0, j])
    print("%s[0, %d]: %.8g -> %.8g (one bf16 step of its rounded value)" % (NAME, j, old, float(w[0, j])))
    for dt in ("bf16", "fp32"):
        show("HIP %s tier: gradient change per tensor" % dt, per_tensor(hip_grads(dt, sd), hip_grads(dt, sd2), names))
        a, b = hip_grads(dt, sd), hip_grads(dt, sd)
        worst = max(per_tensor(a, b, names).values())
        print("    (same weights twice: worst tensor %.1e)" % worst)
    cfg = O.Cfg(1007, 6, 8, 512, 2048, d_condition=128, conditioning="continuous_concat")
    x, c, y = batch()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, _, Ga = O.loss_and_grads(cfg, {k: v.float() for k, v in sd.items()}, x, c, y)
        _, _, Gb = O.loss_and_grads(cfg, {k: v.float() for k, v in sd2.items()}, x, c, y)
    show("oracle under torch.autocast(bfloat16) on the host: gradient change per tensor",
         per_tensor({k: Ga[k].float() for k in names}, {k: Gb[k].float() for k in names}, names))


if __name__ == "__main__":
    main()
