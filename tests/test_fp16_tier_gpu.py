"""f16 tier (round 6): the reference's own mixed precision is fp16 autocast + torch.cuda.amp.GradScaler
(/root/reference/src/train.py:101,108,281,317-324; generate.py:116).  The HIP engine's `compute_dtype="fp16"` tier stores
activations / weights in f16 (v_mfma_f32_32x32x16_f16, f32 accumulate) and keeps the scaler's state and decisions on the
device (me_scaler_step).  These tests pin the scaler against torch's GradScaler itself, the skipped-step semantics, and the
tier as a training path; the parity of its logits / gradients is in test_model_gpu.py ([fp16] parametrisations)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as O  # noqa: E402

DEV = "cuda"


def relerr(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def test_scaler_step_follows_torch_gradscaler():
    """me_scaler_step against torch.cuda.amp.GradScaler driven with the same finite / inf pattern: scale, growth tracker and
    the number of optimiser steps actually taken must agree after every update (growth every 4 finite steps here)."""
    from midiemo import _lib
    from midiemo.optim import LossScaler
    pattern = [0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0]         # 1 = non-finite gradient this step
    ref = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=4)
    p = torch.nn.Parameter(torch.zeros(4, device=DEV))
    sgd = torch.optim.SGD([p], lr=1.0)
    mine = LossScaler(DEV, init_scale=1024.0, growth_interval=4)
    ref.scale(torch.zeros(1, device=DEV))                  # GradScaler creates its device state on the first scale()
    taken = 0
    for i, bad in enumerate(pattern):
        p.grad = torch.full((4,), float("inf") if bad else 1.0, device=DEV)
        before = p.detach().clone()
        ref.unscale_(sgd)
        ref.step(sgd)
        ref.update()
        taken += int(not torch.equal(before, p.detach()))
        sumsq = torch.full((1,), float("nan") if (bad and i % 2) else (float("inf") if bad else 3.0), device=DEV)
        scale_used = mine.get_scale()
        mine.update(sumsq)
        st = mine.state.tolist()
        assert st[_lib.ME_SCALER_SCALE] == ref.get_scale(), (i, st, ref.get_scale())
        assert int(st[_lib.ME_SCALER_TRACKER]) == int(ref._growth_tracker.item()), (i, st)
        assert int(st[_lib.ME_SCALER_STEP]) == taken and int(st[_lib.ME_SCALER_FOUND_INF]) == bad, (i, st, taken)
        assert st[_lib.ME_SCALER_INV] == 1.0 / scale_used
    assert mine.steps_skipped() == sum(pattern)
    sd = mine.state_dict()
    rsd = ref.state_dict()
    assert set(sd) == set(rsd) and all(sd[k] == rsd[k] for k in sd), (sd, rsd)        # scaler.pt is interchangeable with the reference's
    other = LossScaler(DEV)
    other.load_state_dict(rsd)
    assert other.state_dict() == sd


def test_fp16_step_unscales_clips_and_skips_like_the_reference():
    """One fused step of the f16 tier = scaler.unscale_ + clip_grad_norm_(1.0) + scaler.step(Adam) + scaler.update()
    (train.py:320-324): against the oracle's Adam on the UNSCALED gradients; then a step whose gradients hold an inf must leave
    parameters and both moments bit-identical, clear the gradients, halve the scale and not count as a step."""
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW, LossScaler
    cfg = O.Cfg(1007, 2, 2, 128, 256, d_condition=32, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 5)
    model, _ = build_model(dict(vocab_size=1007, n_layer=2, n_head=2, d_model=128, d_inner=256, dropout=0.0, d_condition=32,
                                conditioning="continuous_concat", compute_dtype="fp16"))
    model.load_state_dict(P, strict=True)
    model = model.to(DEV).train()
    assert model.compute_dtype == torch.float16
    scaler = LossScaler(DEV)
    opt = FusedAdamW(model, lr=1e-3, clip=1.0, scaler=scaler)
    tok, cond, tgt = O.synthetic_batch(cfg, 3, 70, seed=9)
    Pc = {k: v.clone() for k, v in P.items()}
    M1 = {k: torch.zeros_like(v) for k, v in P.items()}
    M2 = {k: torch.zeros_like(v) for k, v in P.items()}
    for step in (1, 2):
        loss = model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV), loss_scale=scaler.scale_tensor)
        model.link_grads()
        G = {k: (p.grad.detach().cpu() / 65536.0) for k, p in model.named_parameters()}      # what unscale_ leaves
        assert all(torch.isfinite(g).all() for g in G.values())
        assert float(model.flat_grads.abs().max()) > 1.0                                        # the gradients really carry the scale
        O.adam_step(Pc, G, M1, M2, step, lr=1e-3)                                               # clip_grad_norm_(1.0) + Adam
        opt.step()
        assert float(model.flat_grads.abs().max()) == 0.0
        worst = max(relerr(p.detach() - P[k].to(DEV), Pc[k] - P[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
        assert worst < 2e-3, (step, worst)                                                    # updates of +-lr: sign flips of ~0 gradients only
        assert math.isfinite(float(loss))
    assert scaler.steps_taken() == 2 and scaler.steps_skipped() == 0 and scaler.get_scale() == 65536.0
    # ---- overflow: one inf in the (scaled) gradients
    before = (model.flat_params.clone(), opt.m.clone(), opt.v.clone())
    model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV), loss_scale=scaler.scale_tensor)
    model.flat_grads[12345] = float("inf")
    opt.step()
    assert torch.equal(model.flat_params, before[0]) and torch.equal(opt.m, before[1]) and torch.equal(opt.v, before[2])
    assert float(model.flat_grads.abs().max()) == 0.0
    assert scaler.steps_taken() == 2 and scaler.steps_skipped() == 1 and scaler.get_scale() == 32768.0
    assert opt.state_dict()["step"] == 2
    # ---- and the next step proceeds with the halved scale and the bias corrections of step 3
    model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV), loss_scale=scaler.scale_tensor)
    model.link_grads()
    G = {k: (p.grad.detach().cpu() / 32768.0) for k, p in model.named_parameters()}
    O.adam_step(Pc, G, M1, M2, 3, lr=1e-3)
    opt.step()
    worst = max(relerr(p.detach() - P[k].to(DEV), Pc[k] - P[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
    assert worst < 3e-3 and scaler.steps_taken() == 3, worst


def test_fp16_tier_learns_like_the_f32_tier():
    """The f16 tier as a TRAINING path (f16 hi + lo residual stream, f16 logits into the loss, dynamic loss scale): the
    learnable task of test_bf16_tier_learns_like_the_f32_tier, 150 Adam steps from the same weights and batches.  The scale starts at
    65536 like GradScaler; every step must have been taken (no overflow at this scale on this model)."""
    from midiemo.models.build_model import build_model  # noqa: F401
    from midiemo.optim import FusedAdamW, LossScaler
    from test_model_gpu import make_model
    cfg = O.Cfg(1007, 2, 4, 128, 256, d_condition=32, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 17)
    B, L = 16, 128

    def batch(step):
        g = torch.Generator().manual_seed(9000 + step)
        period = torch.randint(3, 9, (B, 1), generator=g)
        start = torch.randint(2, 900, (B, 1), generator=g)
        pos = torch.arange(L + 1)[None, :]
        tok = start + (pos % period) * 7
        cond = torch.stack([(period[:, 0].float() - 5.5) / 3.0, torch.zeros(B)], -1)
        return tok[:, :-1].contiguous().to(DEV), cond.to(DEV), tok[:, 1:].contiguous().to(DEV)

    traj = {}
    for cd in ("fp32", "fp16"):
        model = make_model(cfg, P, cd).train()
        scaler = LossScaler(DEV) if cd == "fp16" else None
        opt = FusedAdamW(model, lr=1e-3, clip=1.0, scaler=scaler)
        losses = []
        for step in range(150):
            x, c, y = batch(step)
            losses.append(model.loss_and_backward(x, c, y, loss_scale=scaler.scale_tensor if scaler else None))
            opt.step()
        traj[cd] = torch.stack(losses).cpu().numpy()
        if scaler:
            assert scaler.steps_taken() + scaler.steps_skipped() == 150 and scaler.steps_skipped() <= 2, scaler.state.tolist()
    a, b = traj["fp32"], traj["fp16"]
    assert a[0] > 6.5 and b[0] > 6.5
    assert a[-10:].mean() < 3.5 and b[-10:].mean() < 3.5
    assert abs(b[-10:].mean() - a[-10:].mean()) < 0.05 * a[-10:].mean() + 0.02
    assert np.abs(b[:20] - a[:20]).max() < 0.01                                 # early steps: the f16 trajectory hugs the f32 one (bf16: 0.03)


def test_fp16_full_size_step_is_finite_with_the_reference_scale():
    """BASELINE config 2 at its full size (B = 32 x L = 1024) in the f16 tier with GradScaler's initial scale 65536: no
    activation or scaled gradient leaves the f16 range (every gradient finite, step taken), and the unscaled gradient agrees
    with the bf16 tier's on the same batch and weights to the tiers' rounding (dropout off)."""
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW, LossScaler
    torch.manual_seed(0)
    args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
                conditioning="continuous_concat")
    g = torch.Generator().manual_seed(11)
    B, L = 32, 1024
    tok = torch.randint(2, 1007, (B, L + 1), generator=g)
    x, y = tok[:, :-1].contiguous().to(DEV), tok[:, 1:].contiguous().to(DEV)
    cond = (torch.rand(B, 2, generator=g) * 2 - 1).to(DEV)
    m16, _ = build_model(dict(args, compute_dtype="fp16"))
    m16 = m16.to(DEV).train()
    mbf, _ = build_model(dict(args, compute_dtype="bf16"))
    mbf.load_state_dict(m16.state_dict())
    mbf = mbf.to(DEV).train()
    scaler = LossScaler(DEV)
    opt = FusedAdamW(m16, lr=2e-5, clip=1.0, scaler=scaler)
    l16 = m16.loss_and_backward(x, cond, y, loss_scale=scaler.scale_tensor)
    lbf = mbf.loss_and_backward(x, cond, y)
    g16 = m16.flat_grads / 65536.0
    assert bool(torch.isfinite(m16.flat_grads).all())
    assert abs(float(l16) - float(lbf)) < 5e-3
    e = relerr(g16, mbf.flat_grads)
    assert e < 3e-2, e                                        # two 16-bit tiers against each other: bf16's rounding dominates
    opt.step()
    assert scaler.steps_taken() == 1 and scaler.steps_skipped() == 0
