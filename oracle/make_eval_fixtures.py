"""Golden vectors for the evaluation aggregation -- runs ONLY in the build container.

The reference's utils.accuracy (utils.py:15-80) is imported and run on seeded batches with RAGGED trailing PAD; the
per-batch loss is torch's CrossEntropyLoss(ignore_index=0) as train.py:124,288-290 builds it; the weighting is the
statement sequence of Runner.evaluate (train.py:256-272) executed here line for line on those per-batch values (the
Runner itself cannot be imported: it needs the dataset and torch._six).  Writes tests/golden/f8_eval.npz (data only).
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_eval_fixtures.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/src"
if not os.path.isdir(REF):
    raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
sys.path.insert(0, REF)
from utils import accuracy as ref_accuracy        # noqa: E402  (reference)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "f8_eval.npz")
g = torch.Generator().manual_seed(808)
V, topk, pad_idx = 37, (1, 5), 0
ce = torch.nn.CrossEntropyLoss(ignore_index=pad_idx)
rec = {"V": V, "pad_idx": pad_idx}
n_elements_total, total_loss, total_accs = 0, 0.0, {k: 0.0 for k in topk}
shapes = [(3, 17), (2, 40), (4, 9), (1, 25)]
for i, (B, L) in enumerate(shapes):
    logits = torch.randn(B, L, V, generator=g)
    target = torch.randint(1, V, (B, L), generator=g)
    inp = torch.randint(1, V, (B, L), generator=g)
    for b in range(B):                                    # ragged trailing PAD, a different amount per row
        npad = int(torch.randint(0, L - 1, (1,), generator=g))
        if npad:
            target[b, L - npad:] = pad_idx
            inp[b, L - npad + 1:] = pad_idx
    # make a good share of the targets actual top-1 / top-5 hits so the accuracies are not ~ 1 / V
    flat = logits.view(-1, V)
    for r in range(0, flat.shape[0], 3):
        flat[r, target.view(-1)[r]] += 6.0
    loss = ce(logits.view(-1, V), target.view(-1))
    accuracies = ref_accuracy(logits, target, topk=topk, ignore_index=pad_idx)
    n_elements = inp.numel()                              # train.py:256
    total_loss += n_elements * loss.item()                # train.py:258
    for key, value in accuracies.items():                 # train.py:259-260
        total_accs[key] += n_elements * value
    n_elements_total += n_elements                        # train.py:261
    rec[f"b{i}_logits"], rec[f"b{i}_target"], rec[f"b{i}_input"] = logits.numpy(), target.numpy(), inp.numpy()
    rec[f"b{i}_loss"] = np.float64(loss.item())
    rec[f"b{i}_acc1"], rec[f"b{i}_acc5"] = np.float64(accuracies[1]), np.float64(accuracies[5])
rec["n_batches"] = len(shapes)
rec["avg_loss"] = np.float64(total_loss / n_elements_total)                     # train.py:271
rec["avg_acc1"] = np.float64(total_accs[1] / n_elements_total)                  # train.py:272
rec["avg_acc5"] = np.float64(total_accs[5] / n_elements_total)
np.savez_compressed(OUT, **rec)
print("wrote", OUT, {k: float(rec[k]) for k in ("avg_loss", "avg_acc1", "avg_acc5")})
