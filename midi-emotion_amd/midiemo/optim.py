"""FusedAdamW: global-norm clip + Adam(W) over the model's flat f32 buffers in two
kernel launches (me_sumsq, me_adamw_step), no host sync.

Semantics = reference train step tail (train.py:319-325): clip_grad_norm_(params,
clip) ; optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0).step() ;
zero_grad().  weight_decay > 0 gives decoupled (AdamW) decay."""
import torch

from . import ops


class FusedAdamW:
    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay, self.clip = lr, betas, eps, weight_decay, clip
        self.step_count = 0
        self._alloc()
        # mirrors torch.optim's param_groups just enough for the reference's LR warm-up code
        # (train.py:329-331 writes optimizer.param_groups[0]['lr'])
        self.param_groups = [{"lr": lr}]

    def _alloc(self):
        f = self.model.flat_params
        self.m = torch.zeros_like(f)
        self.v = torch.zeros_like(f)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=f.device)
        # ordered block sums: the clip coefficient is bit-reproducible, so data-parallel ranks (identical reduced
        # gradients) stay bit-identical through the update
        self._sumsq_ws = ops.sumsq_ws(f.device)

    def grad_norm(self):
        """Device scalar: global L2 norm of the (already all-reduced) flat gradient."""
        self.sumsq.zero_()
        ops.sumsq(self.model.flat_grads, self.sumsq, ws=self._sumsq_ws)
        return self.sumsq.sqrt()

    def step(self, grad_scale=1.0, zero_grad=True):
        m = self.model
        if self.m.device != m.flat_params.device or self.m.numel() != m.flat_params.numel():
            self._alloc()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        self.sumsq.zero_()
        if self.clip and self.clip > 0:
            try:
                ops.sumsq(m.flat_grads, self.sumsq, ws=self._sumsq_ws)
            except Exception:
                self._sumsq_ws = ops.sumsq_ws(m.flat_params.device)      # a failed launch may leave the ticket counter set
                raise
        ops.adamw_step(m.flat_params, m.flat_grads, self.m, self.v, self.sumsq, self.clip or 0.0, grad_scale, lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, zero_grad)
        m.mark_params_changed()

    def zero_grad(self):
        self.model.flat_grads.zero_()

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        """Own format ({"m", "v", "step", "lr"}) or a torch.optim.Adam state_dict as the reference writes it
        (train.py:403: {"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}): the per-parameter
        moments are copied into the flat buffers by NAME, position i of the torch state being the i-th key of the
        checkpoint ABI (= the reference model's parameters() order = this model's state_dict order)."""
        if "state" in sd and "param_groups" in sd:
            order = [n for n, _ in self.model.named_parameters()]
            if order != [k for k in self.model.state_dict().keys() if k in set(order)]:
                raise ValueError("model parameters() order differs from its state_dict order: a positional torch.optim "
                                 "state cannot be mapped to names")
            params = list(self.model.parameters())
            idx = [i for g in sd["param_groups"] for i in g["params"]]
            if len(idx) != len(params):
                raise ValueError("torch Adam state has %d parameters, the model %d" % (len(idx), len(params)))
            names = {id(p): n for n, p in self.model.named_parameters()}
            step = 0
            for i, p in zip(idx, params):
                st = sd["state"].get(i)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError("torch Adam state of parameter %d has shape %s, %s has %s" %
                                     (i, tuple(st["exp_avg"].shape), names[id(p)], tuple(p.shape)))
                self.model._pview(self.m, names[id(p)]).copy_(st["exp_avg"])
                self.model._pview(self.v, names[id(p)]).copy_(st["exp_avg_sq"])
                step = max(step, int(st["step"]))
            self.step_count = step
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.lr)
            return
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        self.param_groups[0]["lr"] = sd.get("lr", self.lr)
