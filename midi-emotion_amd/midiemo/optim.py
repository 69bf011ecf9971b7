"""FusedAdamW: global-norm clip + Adam(W) over the model's flat f32 buffers in two
kernel launches (me_sumsq, me_adamw_step), no host sync.

Semantics = reference train step tail (train.py:319-325): clip_grad_norm_(params,
clip) ; optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0).step() ;
zero_grad().  weight_decay > 0 gives decoupled (AdamW) decay."""
import torch

from . import _lib, ops


class LossScaler:
    """torch.cuda.amp.GradScaler of the reference's fp16 autocast path (train.py:101,108,317-324) for the f16 tier, with
    its state and its decisions ON THE DEVICE (me_scaler_step): no `.item()` per step.

        loss = model.loss_and_backward(x, c, y, loss_scale=scaler.scale_tensor)   # scaler.scale(loss).backward()
        opt.step(scaler=scaler)                        # unscale_ + clip_grad_norm_ + scaler.step(optimizer) + scaler.update()

    Same defaults as GradScaler (init 65536, growth 2.0 every 2000 finite steps, backoff 0.5); a step whose scaled gradient
    norm is inf / nan leaves parameters and Adam state untouched.  state_dict() has GradScaler's keys, so `scaler.pt` is
    interchangeable with the reference's."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.zeros(_lib.ME_SCALER_WORDS, dtype=torch.float32, device=device)
        self.state[_lib.ME_SCALER_SCALE] = float(init_scale)

    @property
    def scale_tensor(self):
        """f32 device scalar: the scale the next backward multiplies into dlogits"""
        return self.state[_lib.ME_SCALER_SCALE:_lib.ME_SCALER_SCALE + 1]

    def update(self, sumsq):
        """one launch between me_sumsq and me_adamw_step: found_inf, 1 / scale for this step, the next scale"""
        ops.scaler_step(self.state, sumsq, self.growth_factor, self.backoff_factor, self.growth_interval)

    # host-side views (each one synchronises: logging / checkpoints only)
    def get_scale(self):
        return float(self.state[_lib.ME_SCALER_SCALE])

    def steps_taken(self):
        return int(self.state[_lib.ME_SCALER_STEP])

    def steps_skipped(self):
        return int(self.state[_lib.ME_SCALER_SKIPPED])

    def set_steps_taken(self, n):
        self.state[_lib.ME_SCALER_STEP] = float(n)

    def state_dict(self):
        st = self.state.tolist()
        return {"scale": st[_lib.ME_SCALER_SCALE], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[_lib.ME_SCALER_TRACKER])}

    def load_state_dict(self, sd):
        if not sd:
            return                                   # GradScaler(enabled=False).state_dict() == {}
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state[_lib.ME_SCALER_SCALE] = float(sd["scale"])
        self.state[_lib.ME_SCALER_TRACKER] = float(sd.get("_growth_tracker", 0))



class FusedAdamW:
    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=1.0, scaler=None):
        self.model = model
        self.scaler = scaler             # LossScaler of the f16 tier (None: gradients are unscaled, every step is taken)
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

    def step(self, grad_scale=1.0, zero_grad=True, scaler=None):
        """scaler (LossScaler, f16 tier): the gradients carry its loss scale -- they are unscaled inside the update, a
        non-finite norm skips the update (gradients are still cleared) and the scale is adapted, all on the device;
        step_count then counts the ATTEMPTED steps, the bias corrections use the device-side count of steps taken."""
        m = self.model
        scaler = scaler if scaler is not None else self.scaler
        if self.m.device != m.flat_params.device or self.m.numel() != m.flat_params.numel():
            self._alloc()
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        self.sumsq.zero_()
        if (self.clip and self.clip > 0) or scaler is not None:
            try:
                ops.sumsq(m.flat_grads, self.sumsq, ws=self._sumsq_ws)
            except Exception:
                self._sumsq_ws = ops.sumsq_ws(m.flat_params.device)      # a failed launch may leave the ticket counter set
                raise
        if scaler is not None:
            scaler.update(self.sumsq)
        ops.adamw_step(m.flat_params, m.flat_grads, self.m, self.v, self.sumsq, self.clip or 0.0, grad_scale, lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, zero_grad,
                       scaler_state=scaler.state if scaler is not None else None)
        m.mark_params_changed()

    def zero_grad(self):
        self.model.flat_grads.zero_()

    def steps_taken(self):
        """optimiser steps that changed the parameters (with a LossScaler: attempted minus skipped; synchronises)"""
        return self.scaler.steps_taken() if self.scaler is not None else self.step_count

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.steps_taken(), "lr": self.param_groups[0]["lr"]}

    def _set_step(self, n):
        self.step_count = int(n)
        if self.scaler is not None:
            self.scaler.set_steps_taken(n)

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
            self._set_step(step)
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.lr)
            return
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self._set_step(int(sd["step"]))
        self.param_groups[0]["lr"] = sd.get("lr", self.lr)
