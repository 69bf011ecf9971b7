"""DDP diagnosis, part 3: the 2-rank bf16 run's step-3 gradient differs from the 1-rank gradient AT THE SAME f32
PARAMETERS by 2e-3 (steps 1, 2: 9e-8).  Compare the prepared (bf16) weights the ranks actually ran with against the ones
the 1-rank model derives from the same f32 parameters, and break the gradient difference down per layer.
usage (GPU box): python tools/diag_ddp_step3.py > gpurun_out/r04/diag_ddp_step3.txt"""
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
import test_ddp_gpu as TD  # noqa: E402

dev = torch.device("cuda", 0)


def main():
    W.use_big(True)
    out = os.path.join(tempfile.mkdtemp(), "ddp.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIDIEMO_DDP_FORCE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "tests", "ddp_worker.py"), "--policy", "window",
                        "--accumulate", "2", "--backend", "gloo", "--compute_dtype", "bf16", "--out", out, "--big", "--dump_prep"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = torch.load(out)
    model = W.build("bf16", dev)
    keep = TD._keep_mask(model)
    for s in range(W.STEPS):
        with torch.no_grad():
            model.flat_params.copy_(got["params_before"][s].to(dev))
        model.mark_params_changed()
        g = TD._grad_at(model, W, s, 2, 2)
        gsplit = TD._grad_at(model, W, s, 2, 2, split=True)
        print("step %d: ranks vs 1-rank at the same parameters: %.2e; 1-rank split feed vs 1-rank: %.2e; ranks vs split feed: %.2e" %
              (s + 1, TD.rel(got["grads"][s][keep], g[keep]), TD.rel(gsplit[keep], g[keep]), TD.rel(got["grads"][s][keep], gsplit[keep])))
        # prepared weights: the ranks' vs mine
        nd = 0
        mine = list(model._prep["layers"]) + [model._prep["head"]]
        for i, (a, b) in enumerate(zip(got["preps"][s], mine)):
            for k, v in a.items():
                if torch.is_tensor(b.get(k)) and b[k].shape == v.shape:
                    d = int((b[k].cpu() != v).sum())
                    if d:
                        nd += d
                        print("    prepared weight %s of layer %d: %d entries differ" % (k, i, d))
        print("    prepared weights differing between the ranks' run and the 1-rank model at the same f32 parameters: %d" % nd)
        fam = {}
        for name, (o, n, _) in model._slices.items():
            m = re.match(r"enc_layers\.(\d+)\.", name)
            k = "L" + m.group(1) if m else name.split(".")[0]
            f = fam.setdefault(k, [0.0, 0.0])
            f[0] += float((got["grads"][s][o:o + n].double() - g[o:o + n].double()).pow(2).sum())
            f[1] += float(g[o:o + n].double().pow(2).sum())
        print("    per bucket: " + "  ".join("%s %.1e" % (k, (a / max(b, 1e-300)) ** 0.5) for k, (a, b) in fam.items()))


if __name__ == "__main__":
    main()
