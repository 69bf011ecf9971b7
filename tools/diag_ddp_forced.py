"""Second half of the DDP diagnosis.  tools/diag_ddp_bf16.py showed that the 2-rank bf16 trajectory is
reproduced by ONE rank that feeds the same 2-row pieces as separate micro-batches ("split"), and that the step-2 gradient
of those runs differs from the 4-row run's by 1e-3 although step 1 agrees to 8e-8.  This tool separates "different
parameters after step 1" from "different computation at the same parameters":
  A  run the unsplit reference for one step, keep p1 (f32 parameters after step 1) and g2
  B  a fresh model, load p1, compute the step-2 gradient with the SPLIT micro-batches -> compare with g2   (same params)
  C  run split for one step (its own p1'), compare p1' with p1: how many f32 entries differ, how many bf16 weights flip
  D  consistency of the prepared weights: every transposed bf16 copy must equal the natural copy's transpose
usage (GPU box): python tools/diag_ddp_forced.py > gpurun_out/r04/diag_ddp_forced.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "midi-emotion_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import ddp_worker as W  # noqa: E402
from midiemo.optim import FusedAdamW  # noqa: E402

dev = torch.device("cuda", 0)


def grad_of_step(model, step, split, accumulate=2):
    model.flat_grads.zero_()
    for micro in range(accumulate):
        parts = [W.micro_batch(step, micro, r, dev) for r in range(2)]
        x, c, y = (torch.cat([p[i] for p in parts]) for i in range(3))
        if split:
            for lo in (0, 2):
                model.loss_and_backward(x[lo:lo + 2], c[lo:lo + 2], y[lo:lo + 2], grad_scale=0.5 / accumulate)
        else:
            model.loss_and_backward(x, c, y, grad_scale=1.0 / accumulate)
    return model.flat_grads.detach().clone()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def prepared_consistent(model):
    bad = []
    for i, L in enumerate(model._prep["layers"]):
        for k in ("Wqkv", "Wo", "W1", "W2"):
            if not torch.equal(L[k].t().contiguous(), L[k + "T"]):
                bad.append((i, k, int((L[k].t() != L[k + "T"]).sum())))
    H = model._prep["head"]
    V = H["Wf"].shape[0]
    if not torch.equal(H["Wf"].t().contiguous(), H["WfT"][:, :V]):
        bad.append(("head", "Wf", int((H["Wf"].t() != H["WfT"][:, :V]).sum())))
    return bad


def main():
    W.use_big(True)
    runs = {}
    for split in (False, True):
        m = W.build("bf16", dev)
        opt = FusedAdamW(m, lr=2e-5, clip=1.0)
        g1 = grad_of_step(m, 0, split)
        opt.step(zero_grad=False)
        p1 = m.flat_params.detach().clone()
        g2 = grad_of_step(m, 1, split)
        runs[split] = dict(g1=g1, p1=p1, g2=g2, bad=prepared_consistent(m), model=m)
    a, b = runs[False], runs[True]
    print("g1 split vs unsplit: rel %.2e" % rel(b["g1"], a["g1"]))
    dp = (b["p1"] - a["p1"]).abs()
    print("p1 split vs unsplit: %d of %d f32 entries differ, max |dp| %.2e (lr 2e-5), %d entries differ by more than lr" %
          (int((dp > 0).sum()), dp.numel(), float(dp.max()), int((dp > 2e-5).sum())))
    flips = int((b["p1"].bfloat16() != a["p1"].bfloat16()).sum())
    print("bf16 images of p1: %d weights round differently" % flips)
    print("g2 free-running split vs unsplit: rel %.2e" % rel(b["g2"], a["g2"]))
    print("prepared weights consistent (natural vs transposed copies): unsplit %s, split %s" % (a["bad"] or "yes", b["bad"] or "yes"))
    # B: the split computation at the UNSPLIT run's parameters
    m = b["model"]
    with torch.no_grad():
        m.flat_params.copy_(a["p1"])
    m.mark_params_changed()
    g2f = grad_of_step(m, 1, True)
    print("g2 split AT THE UNSPLIT RUN'S PARAMETERS vs unsplit: rel %.2e   <- same parameters, different micro-batching" % rel(g2f, a["g2"]))
    # and the other way round
    m = a["model"]
    with torch.no_grad():
        m.flat_params.copy_(b["p1"])
    m.mark_params_changed()
    g2r = grad_of_step(m, 1, False)
    print("g2 unsplit at the split run's parameters vs split: rel %.2e" % rel(g2r, b["g2"]))
    # which entries of p1 flipped, by family
    import re
    fam = {}
    d16 = b["p1"].bfloat16() != a["p1"].bfloat16()
    for name, (o, n, _) in m._slices.items():
        c = int(d16[o:o + n].sum())
        if c:
            k = re.sub(r"enc_layers\.(\d+)\.", r"L\1.", name)
            fam[k] = fam.get(k, 0) + c
    print("bf16 flips by tensor:", fam)


if __name__ == "__main__":
    main()
