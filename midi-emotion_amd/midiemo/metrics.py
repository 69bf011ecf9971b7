"""Evaluation aggregation of the reference's Runner.evaluate (train.py:222-275) + utils.accuracy (utils.py:15-80).

Per batch: loss = CrossEntropyLoss(ignore_index=pad) (mean over the batch's non-PAD targets) and top-k accuracy =
hits / (#non-PAD targets of the batch); both are weighted by `input_.numel()` -- the number of INPUT positions of the
batch, padded ones included (train.py:256-262) -- and divided by the total of those weights (train.py:271-272).  With
ragged padding this is not the pooled accuracy over all valid targets; it is what the reference reports, so it is what
this does (pinned to the reference's own utils.accuracy in tests/golden/f8_eval.npz)."""
import torch


class EvalAccumulator:
    """acc = [sum n*loss, sum n*top1, sum n*top5, sum n] with n = input_.numel() of each batch (device f64: no host sync
    per batch; the four sums add across ranks)."""

    def __init__(self, device, topk=(1, 5)):
        self.topk = tuple(topk)
        self.acc = torch.zeros(2 + len(self.topk), device=device, dtype=torch.float64)

    def add(self, loss, logits, target, n_input, pad_idx):
        """loss: the batch's mean CE over non-PAD targets (device scalar); logits [.., V]; target [..]; n_input =
        input_.numel() of the batch."""
        y = target.reshape(-1)
        valid = y != pad_idx
        top = logits.reshape(-1, logits.size(-1)).topk(max(self.topk), dim=-1).indices
        hit = top == y.reshape(-1, 1)
        nv = valid.sum().double()                       # utils.accuracy divides by the batch's valid targets (nan if none)
        self.acc[0] += loss.double() * n_input
        for i, k in enumerate(self.topk):
            self.acc[1 + i] += (hit[:, :k].any(-1) & valid).sum().double() / nv * n_input
        self.acc[-1] += n_input

    def result(self):
        a = self.acc.tolist()
        den = a[-1]
        if den == 0:                    # no batch evaluated: the reference reports nan (train.py:267-269)
            return float("nan"), {k: float("nan") for k in self.topk}
        return a[0] / den, {k: a[1 + i] / den for i, k in enumerate(self.topk)}
