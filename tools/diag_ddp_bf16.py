"""where does the 2-rank bf16 trajectory of the headline model leave the 1-rank one?
Runs (all on cuda:0, 6L d512 8H, B = 2 x L = 256 per rank and micro-batch, accumulate 2, 3 steps):
  ddp2      two ranks through GradAllReducer (gloo), tests/ddp_worker.py
  ref       one rank on the concatenated batch (4 rows per micro-batch)
  variants  one rank with the rows of every micro-batch permuted, and one rank that feeds the four 2-row pieces
            as four micro-batches (same per-launch T as a rank of the 2-rank run)
and prints, per run against ref: gradient rel error of steps 1..3, update rel error after each step, and per parameter
family the number of entries whose 3-step update differs by more than lr.
usage (GPU box): python tools/diag_ddp_bf16.py [--dtype bf16] > gpurun_out/diag_ddp.txt"""
import argparse
import itertools
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "midi-emotion_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import ddp_worker as W  # noqa: E402
from midiemo.optim import FusedAdamW  # noqa: E402

LR = 2e-5


def one_rank(dtype, order=(0, 1, 2, 3), split=False, accumulate=2, freeze_wk_bias=False):
    """order: permutation of the 4 rows [rank0 row0, rank0 row1, rank1 row0, rank1 row1] of each micro-batch;
    split: feed rows (0,1) and (2,3) as separate micro-batches (grad_scale 1 / (2 accumulate))."""
    dev = torch.device("cuda", 0)
    model = W.build(dtype, dev)
    opt = FusedAdamW(model, lr=LR, clip=1.0)
    p0 = model.flat_params.detach().cpu().clone()
    gs, ps = [], []
    for step in range(W.STEPS):
        for micro in range(accumulate):
            parts = [W.micro_batch(step, micro, r, dev) for r in range(2)]
            x, c, y = (torch.cat([p[i] for p in parts]) for i in range(3))
            idx = torch.tensor(order, device=dev)
            x, c, y = x[idx], c[idx], y[idx]
            if split:
                for lo in (0, 2):
                    model.loss_and_backward(x[lo:lo + 2], c[lo:lo + 2], y[lo:lo + 2], grad_scale=0.5 / accumulate)
            else:
                model.loss_and_backward(x, c, y, grad_scale=1.0 / accumulate)
        if freeze_wk_bias:          # the parameter whose gradient is exactly 0 in exact arithmetic: take it out of the update
            for name, (o, n, _) in model._slices.items():
                if name.endswith("Wk.bias"):
                    model.flat_grads[o:o + n].zero_()
        gs.append(model.flat_grads.cpu().clone())
        opt.step()
        ps.append(model.flat_params.detach().cpu().clone())
    torch.cuda.synchronize()
    return dict(grads=gs, params_steps=ps, p0=p0, slices=dict(model._slices))


def two_ranks(dtype, policy="window", accumulate=2, port=29611):
    out = os.path.join(tempfile.mkdtemp(), "ddp.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIDIEMO_DDP_FORCE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py"),
                        "--policy", policy, "--accumulate", str(accumulate), "--backend", "gloo", "--compute_dtype", dtype,
                        "--out", out, "--big"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return torch.load(out)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def family(name):
    return re.sub(r"enc_layers\.\d+\.", "L.", name)


def report(tag, run, ref, keep):
    p0 = ref["p0"]
    line = [tag.ljust(28)]
    for s in range(W.STEPS):
        line.append("g%d %.2e" % (s + 1, rel(run["grads"][s][keep], ref["grads"][s][keep])))
    for s in range(W.STEPS):
        line.append("u%d %.2e" % (s + 1, rel((run["params_steps"][s] - p0)[keep], (ref["params_steps"][s] - p0)[keep])))
    print("  ".join(line))
    for s in range(1, W.STEPS):
        fam = {}
        for name, (o, n, _) in ref["slices"].items():
            f = fam.setdefault(family(name), [0.0, 0.0])
            f[0] += float((run["grads"][s][o:o + n].double() - ref["grads"][s][o:o + n].double()).pow(2).sum())
            f[1] += float(ref["grads"][s][o:o + n].double().pow(2).sum())
        print("    g%d rel per family  " % (s + 1) + "  ".join("%s %.1e" % (k.replace("L.", ""), (a / max(b, 1e-300)) ** 0.5)
                                                                 for k, (a, b) in sorted(fam.items())))
    du = (run["params_steps"][-1] - ref["params_steps"][-1]).abs()
    fam = {}
    for name, (o, n, _) in ref["slices"].items():
        if name.endswith("Wk.bias"):
            continue
        f = fam.setdefault(family(name), [0, 0])
        f[0] += int((du[o:o + n] > LR).sum())
        f[1] += n
    print("    entries with |d update| > lr  " + "  ".join("%s %d/%d" % (k, a, b) for k, (a, b) in sorted(fam.items()) if a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--freeze", action="store_true", help="only the Wk.bias experiment")
    ap.add_argument("--few", action="store_true", help="two permutations instead of six")
    a = ap.parse_args()
    W.use_big(True)
    ref = one_rank(a.dtype)
    keep = torch.ones(ref["p0"].numel(), dtype=torch.bool)
    for name, (o, n, _) in ref["slices"].items():
        if name.endswith("Wk.bias"):
            keep[o:o + n] = False
    print("dtype", a.dtype, "model", W.CFG, "B", W.B, "L", W.L)
    if a.freeze:
        print("Wk.bias taken out of the update in every run (its gradient is 0 in exact arithmetic; the runs step on rounding noise there)")
        ref = one_rank(a.dtype, freeze_wk_bias=True)
        report("1-rank split, Wk.bias frozen", one_rank(a.dtype, split=True, freeze_wk_bias=True), ref, keep)
        report("1-rank split rows (2,3,0,1), frozen", one_rank(a.dtype, order=(2, 3, 0, 1), split=True, freeze_wk_bias=True), ref, keep)
        report("1-rank rows (2,3,0,1), frozen", one_rank(a.dtype, order=(2, 3, 0, 1), freeze_wk_bias=True), ref, keep)
        return
    report("ref again (determinism)", one_rank(a.dtype), ref, keep)
    for pol in ("window", "end"):
        report("ddp2 %s" % pol, two_ranks(a.dtype, pol), ref, keep)
    report("1-rank split (T of a rank)", one_rank(a.dtype, split=True), ref, keep)
    perms = [(2, 3, 0, 1), (1, 0, 3, 2), (0, 2, 1, 3), (3, 2, 1, 0), (1, 2, 3, 0), (2, 0, 3, 1)]
    for p in perms[:2] if a.few else perms:
        report("1-rank rows %s" % (p,), one_rank(a.dtype, order=p), ref, keep)
        report("1-rank split rows %s" % (p,), one_rank(a.dtype, order=p, split=True), ref, keep)


if __name__ == "__main__":
    main()
