"""Model-level parity (GPU): the HIP engine behind build_model()/nn.Module against
the oracle and the golden fixtures captured from the reference.

Tolerances: f32 tier -- logits rel-L2 <= 1e-4 (north_star gate is 1e-3), grads rel-L2 <= 2e-4;
bf16 tier -- DERIVED, not hand-picked: the HIP error against the fp32 reference must not exceed the reference's own
bf16-autocast error on the same batch (fixture F3 stores it for the headline model: 3.77e-3 on the stored rows; for
config 4 the oracle is run under torch.autocast(bfloat16) on the host next to its fp32 run)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as O  # noqa: E402

MODES = ["none", "discrete_token", "continuous_token", "continuous_concat"]
DEV = "cuda"


def relerr(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    n = float(ref.norm())
    if n < 1e-12:
        return float((got - ref).abs().max())
    return float((got - ref).norm() / n)


# ---- bf16 tier: bounds DERIVED from the reference arithmetic itself (no hand-picked numbers).  The oracle is run under
# torch.autocast(bfloat16) on the host next to its fp32 run on the SAME batch; what it loses there is what the reference
# would lose under its own --amp path (train.py:281).  The HIP path must stay within K x that loss, K recorded here:
#   logits  K = 1.25  (the HIP path keeps an ~f32 residual stream, rounds P to bf16 like autocast's matmul inputs)
#   loss    K = 2     (+ 1e-4 absolute: both errors are tiny differences of O(7) numbers)
#   grads   K = 1.5   (dS / dP are rounded to bf16 where autocast's backward keeps fp32 intermediates), floored at ONE bf16
#                      rounding unit 2^-9: a tensor whose autocast error is below the resolution of the storage type (fc.bias: the
#                      column sums of dlogits, which this path stores in bf16 and autocast keeps in fp32 -- 6.7e-4 vs 3.7e-4)
#                      is not held to less than that resolution.  Measured ratios go to gpurun_out/parity_report.txt
K_LOGITS, K_LOSS, K_GRADS, BF16_ULP = 1.25, 2.0, 1.5, 2.0 ** -9
# f16 tier (round 6): the reference's OWN mixed precision is fp16 autocast + GradScaler (train.py:101,108,281,317-324), so its
# yardstick is the oracle under torch.autocast(float16) with the loss scale of the step (GradScaler's initial 65536); same
# ratios K, rounding unit 2^-12.  North_star's absolute logits bound (1e-3 rel) is asserted on top wherever this tier runs.
AC_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16}
ULP = {"bf16": 2.0 ** -9, "fp16": 2.0 ** -12}
# gradient ratio of the f16 tier: 2.0.  Between its kernels the HIP path stores the residual-stream GRADIENT in 16 bits (one
# rounding per residual add) where autocast's backward carries it in f32; against the bf16 oracle (own error 4e-3 .. 4e-2 per
# tensor) that is invisible, against the f16 oracle (1.7e-3 on fc_condition.bias, the sum of that stream over all tokens) it shows:
# measured worst ratios 1.15 / 1.26 / 1.44 / 1.62 over the four modes (gpurun_out/parity_report.txt)
K_GRADS_TIER = {"bf16": K_GRADS, "fp16": 2.0}
LOSS_SCALE = {"bf16": 1.0, "fp16": 65536.0}
TIERS = ["fp32", "bf16", "fp16"]
NORTH_STAR_LOGITS = 1e-3


def fused_loss_and_grads(model, cd, *batch, **kw):
    """model.loss_and_backward in the tier's own way: the f16 tier backpropagates the scaled loss (device-resident scale, as
    optim.LossScaler passes it) and the gradients are unscaled here like the optimiser step does."""
    if cd != "fp16":
        return model.loss_and_backward(*batch, **kw)
    sc = torch.full((1,), LOSS_SCALE[cd], device=DEV)
    loss = model.loss_and_backward(*batch, loss_scale=sc, **kw)
    model.flat_grads.mul_(1.0 / LOSS_SCALE[cd])
    return loss


def autocast_forward_err(cfg, P, tok, cond, cd="bf16"):
    """rel-L2 of the oracle's 16-bit-autocast logits against its own fp32 logits on this batch."""
    P32 = {k: v.float() for k, v in P.items()}
    ref = O.forward(cfg, P32, tok, cond)
    with torch.autocast("cpu", dtype=AC_DTYPE[cd]):
        ac = O.forward(cfg, P32, tok, cond)
    ok = ~torch.isnan(ref)
    return relerr(ac.float()[ok], ref[ok])


def autocast_train_err(cfg, P, tok, cond, tgt, fn=None, cd="bf16"):
    """(logits rel-L2, |loss difference|, {parameter: gradient rel-L2}) of the oracle under 16-bit autocast (fp16: with the
    step's loss scale, GradScaler.scale / unscale_) vs its fp32 run."""
    fn = fn or O.loss_and_grads
    P32 = {k: v.float() for k, v in P.items()}
    cond32 = cond.float() if torch.is_tensor(cond) and cond.is_floating_point() else cond
    loss_ref, lg_ref, G = fn(cfg, P32, tok, cond32, tgt)
    kw = {"loss_scale": LOSS_SCALE[cd]} if cd == "fp16" else {}
    with torch.autocast("cpu", dtype=AC_DTYPE[cd]):
        loss_ac, lg_ac, G_ac = fn(cfg, P32, tok, cond32, tgt, **kw)
    ge = {k: relerr(G_ac[k].float(), G[k]) for k in G}
    le = relerr(lg_ac.float(), lg_ref) if lg_ref is not None else 0.0
    return le, abs(float(loss_ac) - float(loss_ref)), ge


def check_bf16_grads(model, G, ge_ac, what, cd="bf16"):
    """every parameter gradient within K_GRADS x the oracle's own autocast error for that tensor (+ 1e-6 for tensors whose
    gradient is rounding noise); returns the worst ratio for the report."""
    bad, worst = {}, (0.0, None)
    for k, p in model.named_parameters():
        if k.endswith("Wk.bias") or float(torch.as_tensor(G[k]).double().norm()) < 1e-9:
            continue
        eg, ea = relerr(p.grad, G[k]), ge_ac[k]
        if eg / max(ea, 1e-12) > worst[0]:
            worst = (eg / max(ea, 1e-12), k)
        if eg > max(K_GRADS_TIER[cd] * ea, ULP[cd]):
            bad[k] = (eg, ea)
    report("%s: worst gradient error / oracle autocast error = %.2f (%s)" % (what, worst[0], worst[1]))
    assert not bad, bad


def make_model(cfg, params, compute_dtype, dropout=0.0):
    from midiemo.models.build_model import build_model
    from midiemo.models.music_transformer import MusicTransformerContinuousToken, MusicTransformerMulti
    if cfg.max_seq == 2048:
        args = dict(vocab_size=cfg.vocab_size, n_layer=cfg.n_layer, n_head=cfg.n_head, d_model=cfg.d_model,
                    d_inner=cfg.d_inner, dropout=dropout, d_condition=cfg.d_condition if cfg.d_condition > 0 else -1,
                    conditioning=cfg.conditioning, compute_dtype=compute_dtype)
        model, _ = build_model(args)
    else:
        kw = dict(embedding_dim=cfg.d_model, d_inner=cfg.d_inner, vocab_size=cfg.vocab_size, num_layer=cfg.n_layer,
                  num_head=cfg.n_head, max_seq=cfg.max_seq, dropout=dropout, pad_token=0, compute_dtype=compute_dtype)
        if cfg.conditioning == "continuous_token":
            model = MusicTransformerContinuousToken(**kw)
        else:
            model = MusicTransformerMulti(d_condition=cfg.d_condition if cfg.d_condition > 0 else -1, **kw)
    missing = model.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(DEV)


def report(line):
    """Measured parity numbers, kept for DESIGN.md (gpurun_out/ is merged back from the GPU box)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")


def f1_cfg(mode, z):
    V, nl, nh, d, di, dc, M = [int(x) for x in z["cfg"]]
    return O.Cfg(V, nl, nh, d, di, d_condition=dc if dc > 0 else -1, conditioning=mode, max_seq=M)


def sub(a):
    f = np.asarray(a).reshape(-1)
    return f[::7] if f.size > 4096 else f


def test_state_dict_keys_match_reference_abi():
    cfg = O.Cfg(1007, 2, 2, 64, 128, d_condition=16, conditioning="continuous_concat")
    m = make_model(cfg, O.seeded_params(cfg, 1), "fp32")
    assert list(m.state_dict().keys()) == list(O.param_shapes(cfg).keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == O.param_shapes(cfg)[k]
    n = sum(p.numel() for p in m.parameters())
    assert n == sum(int(np.prod(s)) for s in O.param_shapes(cfg).values())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cd", TIERS)
def test_f1_logits_vs_golden(golden_dir, mode, cd):
    z = np.load(os.path.join(golden_dir, f"f1_{mode}.npz"))
    cfg = f1_cfg(mode, z)
    model = make_model(cfg, O.seeded_params(cfg, int(z["weight_seed"])), cd).eval()
    errs, lims = {}, {}
    P = O.seeded_params(cfg, int(z["weight_seed"]))
    for L in (1, 7, 33, 64):
        tok = torch.from_numpy(z[f"L{L}_tokens"]).to(DEV)
        cond = torch.from_numpy(z[f"L{L}_cond"]).to(DEV)
        with torch.no_grad():
            lg = model(tok, cond)
        errs[L] = relerr(lg, z[f"L{L}_logits"])
        lims[L] = 1e-4 if cd == "fp32" else K_LOGITS * autocast_forward_err(cfg, P, tok.cpu(), cond.cpu(), cd)
        if cd == "fp16":
            lims[L] = min(lims[L], NORTH_STAR_LOGITS)
    report("f1 %s logits rel-L2 vs reference, compute=%s: %s (bound %s)" %
           (mode, cd, {k: "%.2e" % v for k, v in errs.items()}, {k: "%.2e" % v for k, v in lims.items()}))
    assert all(errs[L] <= lims[L] for L in errs), (errs, lims)
    # PAD at position 0 -> the reference's NaN pattern
    with torch.no_grad():
        lg = model(torch.from_numpy(z["pad0_tokens"]).to(DEV), torch.from_numpy(z["pad0_cond"]).to(DEV))
    assert np.array_equal(torch.isnan(lg).cpu().numpy(), z["pad0_isnan"])


@pytest.mark.parametrize("mode", MODES)
def test_f1_autograd_grads_and_fused_adam_vs_golden(golden_dir, mode):
    """Reference usage pattern (train.py:288-325) through autograd, then the fused path; f32 tier."""
    z = np.load(os.path.join(golden_dir, f"f1_{mode}.npz"))
    cfg = f1_cfg(mode, z)
    P0 = O.seeded_params(cfg, int(z["weight_seed"]))
    model = make_model(cfg, P0, "fp32").train()
    V = cfg.vocab_size

    # --- autograd path, step-1 batch: p.grad vs golden grads
    tok = torch.from_numpy(z["opt1_tokens"]).to(DEV)
    cond = torch.from_numpy(z["opt1_cond"]).to(DEV)
    tgt = torch.from_numpy(z["opt1_target"]).to(DEV)
    lg = model(tok, cond)
    loss = torch.nn.functional.cross_entropy(lg.reshape(-1, V), tgt.reshape(-1), ignore_index=0)
    loss.backward()
    assert abs(loss.item() - float(z["opt1_loss"])) < 2e-5
    bad = {}
    for k, p in model.named_parameters():
        gn = float(z[f"gradnorm/{k}"])
        g = p.grad.detach().cpu().numpy()
        err = np.abs(sub(g) - z[f"grad/{k}"]).max()
        rms = gn / np.sqrt(max(g.size, 1))
        if err > 1e-2 * rms + 1e-7 or abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - gn) > 3e-4 * gn + 1e-8:
            bad[k] = (float(err), rms)
    assert not bad, bad
    model.zero_grad(set_to_none=True)

    # --- fused path: 3 steps of loss_and_backward + FusedAdamW vs golden normalised updates
    from midiemo.optim import FusedAdamW
    opt = FusedAdamW(model, lr=2e-5, clip=1.0)
    for step in (1, 2, 3):
        tok = torch.from_numpy(z[f"opt{step}_tokens"]).to(DEV)
        cond = torch.from_numpy(z[f"opt{step}_cond"]).to(DEV)
        tgt = torch.from_numpy(z[f"opt{step}_target"]).to(DEV)
        loss = model.loss_and_backward(tok, cond, tgt)
        assert abs(loss.item() - float(z[f"opt{step}_loss"])) < 3e-5, (step, loss.item())
        gn = opt.grad_norm().item()
        assert abs(gn - float(z[f"opt{step}_gradnorm"])) < 3e-4 * float(z[f"opt{step}_gradnorm"]), step
        opt.step()
        assert float(model.flat_grads.abs().max()) == 0.0
        if step in (1, 3):
            for k, p in model.named_parameters():
                if k.endswith("Wk.bias"):
                    continue        # exactly-zero gradient (softmax shift invariance); reference steps on fp noise
                upd = (p.detach().cpu().double() - P0[k].double()) / 2e-5
                err = np.abs(sub(upd.numpy()) - z[f"upd{step}/{k}"])
                tol = 3e-2 + 6e-3 * float(P0[k].abs().max())
                assert (err > tol).mean() <= 3e-2 and err.max() < 0.5, (k, step, err.max(), (err > tol).mean())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cd", TIERS)
def test_grads_vs_oracle_random_batch(mode, cd):
    """Every parameter gradient against the oracle's autograd (rel-L2 per tensor)."""
    V = 1017 if mode == "discrete_token" else 1007
    cfg = O.Cfg(V, 2, 2, 128, 256, d_condition=32, conditioning=mode)
    P = O.seeded_params(cfg, 5)
    model = make_model(cfg, P, cd).train()
    tok, cond, tgt = O.synthetic_batch(cfg, 3, 70, seed=9)
    tok[-1, -9:] = 0
    tgt[-1, -10:] = 0
    loss_ref, lg_ref, G = O.loss_and_grads(cfg, {k: v.double() for k, v in P.items()}, tok, cond.double(), tgt)
    loss = fused_loss_and_grads(model, cd, tok.to(DEV), cond.to(DEV), tgt.to(DEV))
    model.link_grads()
    if cd == "fp32":
        assert abs(loss.item() - loss_ref.item()) < 1e-4, (loss.item(), loss_ref.item())
        bad = {}
        for k, p in model.named_parameters():
            if k.endswith("Wk.bias"):
                assert float(p.grad.abs().max()) < 1e-6
                continue
            e = relerr(p.grad, G[k])
            if e > 2e-4:
                bad[k] = e
        assert not bad, bad
    else:
        # bf16: e.g. ReLU gates computed from bf16-rounded pre-activations flip for |x| ~ 1e-3 -- in the oracle's autocast run
        # just as here, which is why the bound is the oracle's own autocast error per tensor and not one number
        _, dl_ac, ge_ac = autocast_train_err(cfg, P, tok, cond, tgt, cd=cd)
        assert abs(loss.item() - loss_ref.item()) <= K_LOSS * dl_ac + 1e-4, (loss.item(), loss_ref.item(), dl_ac)
        check_bf16_grads(model, G, ge_ac, "random batch %s %s" % (mode, cd), cd)


def test_max_seq_discrete_token_config4_shape():
    """BASELINE config 4's shape: discrete_token (V = 1017, two bin tokens in front), L = max_seq = 2048 -- the largest
    sequence the model accepts.  Logits, loss and every gradient against the oracle (f32 tier), PAD tail included."""
    cfg = O.Cfg(1017, 2, 2, 128, 256, conditioning="discrete_token")
    P = O.seeded_params(cfg, 2)
    model = make_model(cfg, P, "fp32").train()
    tok, cond, tgt = O.synthetic_batch(cfg, 1, 2048, seed=3)
    tok[0, -37:] = 0
    tgt[0, -38:] = 0
    loss_ref, lg_ref, G = O.loss_and_grads(cfg, P, tok, cond, tgt)
    loss = model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV))
    model.link_grads()
    assert abs(loss.item() - loss_ref.item()) < 1e-4, (loss.item(), loss_ref.item())
    worst = max(relerr(p.grad, G[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
    assert worst < 2e-4, worst
    with torch.no_grad():
        lg = model.eval()(tok.to(DEV), cond.to(DEV))
    assert relerr(lg, lg_ref) < 1e-4
    report("config-4 shape (discrete_token, L=2048, f32): logits rel %.2e, worst grad rel %.2e" % (relerr(lg, lg_ref), worst))
    with pytest.raises((RuntimeError, ValueError)):
        model(torch.zeros(1, 2049, dtype=torch.long, device=DEV), cond.to(DEV))          # L > max_seq is rejected


def test_config4_bf16_discrete_token_L2048():
    """BASELINE config 4 in ITS OWN dtype and size: discrete_token (V = 1017, the two emotion-bin tokens in front),
    6 layers d512 8 heads d_inner 2048, L = 2048 = max_seq, bf16 storage.  Logits, loss and every parameter gradient
    against the oracle's fp32 run; thresholds = what the oracle itself loses under torch.autocast(bfloat16) on the same
    batch (logits: not above it; gradients: within 1.5x of it per tensor -- the HIP path rounds dS / dP to bf16 where
    autocast's backward keeps some fp32 intermediates).  Exercises the 256-tile bf16 GEMMs, Frag<bf16> transpose reads
    and 64 key tiles per query block at the longest sequence the model accepts."""
    cfg = O.Cfg(1017, 6, 8, 512, 2048, conditioning="discrete_token")
    P = O.seeded_params(cfg, 2)
    tok, cond, tgt = O.synthetic_batch(cfg, 1, 2048, seed=3)
    tok[0, 0], tok[0, 1] = 1007 + 3, 1012 + 1                          # valence / arousal bin tokens (loader.py:158-164)
    tgt[0, 0] = tok[0, 1]
    tok[0, -21:] = 0
    tgt[0, -22:] = 0
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    loss_ref, lg_ref, G = O.loss_and_grads(cfg, P, tok, cond, tgt)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss_ac, lg_ac, G_ac = O.loss_and_grads(cfg, P, tok, cond, tgt)
    ok_rows = torch.arange(2048 - 21)                                  # logits of PAD positions are compared as well below
    ac_logit = relerr(lg_ac.float(), lg_ref)
    model = make_model(cfg, P, "bf16").train()
    loss = model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV))
    model.link_grads()
    with torch.no_grad():
        lg = model.eval()(tok.to(DEV), cond.to(DEV))
    e = relerr(lg, lg_ref)
    report("config 4 (discrete_token V1017 6L d512 8H L2048, bf16): logits rel %.3e (oracle under bf16 autocast %.3e), "
           "loss %.5f vs %.5f" % (e, ac_logit, loss.item(), loss_ref.item()))
    assert e <= ac_logit, (e, ac_logit)
    assert abs(loss.item() - loss_ref.item()) <= max(2 * abs(loss_ac.item() - loss_ref.item()), 2e-3)
    bad, worst = {}, (0.0, None)
    for k, p in model.named_parameters():
        if k.endswith("Wk.bias"):
            continue
        eg, ea = relerr(p.grad, G[k]), relerr(G_ac[k], G[k])
        if eg / max(ea, 1e-12) > worst[0]:
            worst = (eg / max(ea, 1e-12), k)
        if eg > 1.5 * ea + 1e-6:
            bad[k] = (eg, ea)
    report("config 4 bf16 gradients: worst ratio to the oracle's autocast error %.2f (%s)" % worst)
    assert not bad, bad
    del ok_rows


def test_forward_follows_torch_optimizer_updates():
    """the INTEGRATION.md 2a pattern -- loss.backward() + a torch.optim optimiser -- updates the
    parameters through the nn.Parameter views, which does not bump the flat buffer's version counter.  The prepared
    (cast / transposed / packed) weights must be refreshed anyway: logits after the step must match the oracle run
    with the stepped parameters, in both tiers."""
    cfg = O.Cfg(1007, 2, 2, 128, 256, d_condition=32, conditioning="continuous_concat")
    for cd, tol in (("fp32", 1e-4), ("bf16", None)):
        P = O.seeded_params(cfg, 11)
        model = make_model(cfg, P, cd).train()
        tok, cond, tgt = O.synthetic_batch(cfg, 2, 64, seed=4)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        lg0 = model(tok.to(DEV), cond.to(DEV))
        loss = torch.nn.functional.cross_entropy(lg0.reshape(-1, 1007), tgt.to(DEV).reshape(-1), ignore_index=0)
        loss.backward()
        opt.step()
        with torch.no_grad():
            lg1 = model(tok.to(DEV), cond.to(DEV))
        P1 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        lg_ref = O.forward(cfg, P1, tok, cond)
        moved = relerr(lg1, lg0.detach())
        assert moved > 1e-2, moved                                  # lr 1e-2: the logits must move visibly
        if tol is None:
            tol = K_LOGITS * autocast_forward_err(cfg, P1, tok, cond)
        assert relerr(lg1, lg_ref) <= tol, (cd, relerr(lg1, lg_ref), tol)
        with torch.no_grad():                                      # p.copy_() / manual re-init through a parameter
            model.fc.weight.mul_(0.5)
            lg2 = model(tok.to(DEV), cond.to(DEV))
        P2 = dict(P1)
        P2["fc.weight"] = P1["fc.weight"] * 0.5
        tol2 = tol if cd == "fp32" else K_LOGITS * autocast_forward_err(cfg, P2, tok, cond)
        assert relerr(lg2, O.forward(cfg, P2, tok, cond)) <= tol2


def test_backward_survives_eval_forward_and_other_shapes():
    """an eval / no-grad forward or a forward of another (B, L) between forward and backward uses
    its own workspace and must not invalidate the pending backward; only a second grad-enabled forward of the SAME
    shape does (and raises)."""
    cfg = O.Cfg(1007, 2, 2, 128, 256, d_condition=32, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 12)
    model = make_model(cfg, P, "fp32").train()
    tok, cond, tgt = O.synthetic_batch(cfg, 2, 48, seed=5)
    tok2, cond2, _ = O.synthetic_batch(cfg, 3, 32, seed=6)
    _, _, G = O.loss_and_grads(cfg, P, tok, cond, tgt)
    lg = model(tok.to(DEV), cond.to(DEV))
    with torch.no_grad():
        model(tok.to(DEV), cond.to(DEV))                            # logging-style forward, same shape, no grad
    lgo = model(tok2.to(DEV), cond2.to(DEV))                        # another shape, grad enabled
    loss = torch.nn.functional.cross_entropy(lg.reshape(-1, 1007), tgt.to(DEV).reshape(-1), ignore_index=0)
    loss.backward()
    worst = max(relerr(p.grad, G[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
    assert worst < 2e-4, worst
    del lgo
    lg_a = model(tok.to(DEV), cond.to(DEV))
    model(tok.to(DEV), cond.to(DEV))                                # second grad-enabled forward of the same shape
    with pytest.raises(RuntimeError, match="one forward in flight"):
        lg_a.sum().backward()


@pytest.mark.parametrize("cd", TIERS)
def test_head_dim_48_like_published_checkpoints(cd):
    """The reference's published models are d768 / 16 heads (head dim 48).  Same geometry, small: d = 96, 2 heads;
    logits, loss, every gradient and a KV-cached decode step against the oracle."""
    cfg = O.Cfg(1007, 2, 2, 96, 192, d_condition=32, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 8)
    model = make_model(cfg, P, cd).train()
    tok, cond, tgt = O.synthetic_batch(cfg, 2, 75, seed=4)
    tok[1, -6:] = 0
    tgt[1, -7:] = 0
    loss_ref, lg_ref, G = O.loss_and_grads(cfg, {k: v.double() for k, v in P.items()}, tok, cond.double(), tgt)
    loss = fused_loss_and_grads(model, cd, tok.to(DEV), cond.to(DEV), tgt.to(DEV))
    model.link_grads()
    worst = max(relerr(p.grad, G[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
    if cd == "fp32":
        lim_lg, lim_step = 1e-4, 1e-4
        assert abs(loss.item() - loss_ref.item()) < 1e-4
        assert worst < 2e-4, worst
    else:
        le_ac, dl_ac, ge_ac = autocast_train_err(cfg, P, tok, cond, tgt, cd=cd)
        assert abs(loss.item() - loss_ref.item()) <= K_LOSS * dl_ac + 1e-4, (loss.item(), loss_ref.item(), dl_ac)
        check_bf16_grads(model, G, ge_ac, "head dim 48 %s" % cd, cd)
        lim_lg = K_LOGITS * le_ac
        lim_step = K_LOGITS * autocast_forward_err(cfg, P, tok[:, :12], cond, cd)     # the decode step sees the 12-token prefix
    model.eval()
    with torch.no_grad():
        lg = model(tok.to(DEV), cond.to(DEV))
        assert relerr(lg, lg_ref) <= lim_lg, (relerr(lg, lg_ref), lim_lg)
        from midiemo.decode import DecodeSession
        sess = DecodeSession(model, 2)
        for t in range(12):
            step_lg = sess.step(tok[:, t].to(DEV), cond.to(DEV))
        assert relerr(step_lg, lg_ref[:, 11]) <= lim_step * (1.0 if cd == "fp32" else 2.0), (relerr(step_lg, lg_ref[:, 11]), lim_step)
    report("head dim 48 (%s): logits rel %.2e, worst grad rel %.2e" % (cd, relerr(lg, lg_ref), worst))


def test_published_geometry_one_layer_f32():
    """One layer of the reference's default / published geometry (d768, 16 heads of 48, d_inner 3072, d_condition
    192, tgt_len-like ragged L): logits, loss and every gradient against the oracle in the exact-f32 tier."""
    cfg = O.Cfg(1007, 1, 16, 768, 3072, d_condition=192, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 13)
    model = make_model(cfg, P, "fp32").train()
    tok, cond, tgt = O.synthetic_batch(cfg, 2, 76, seed=6)
    loss_ref, lg_ref, G = O.loss_and_grads(cfg, P, tok, cond, tgt)
    loss = model.loss_and_backward(tok.to(DEV), cond.to(DEV), tgt.to(DEV))
    model.link_grads()
    assert abs(loss.item() - loss_ref.item()) < 1e-4, (loss.item(), loss_ref.item())
    worst = max(relerr(p.grad, G[k]) for k, p in model.named_parameters() if not k.endswith("Wk.bias"))
    assert worst < 2e-4, worst
    with torch.no_grad():
        lg = model.eval()(tok.to(DEV), cond.to(DEV))
    assert relerr(lg, lg_ref) < 1e-4
    report("published geometry, 1 layer (f32): logits rel %.2e, worst grad rel %.2e" % (relerr(lg, lg_ref), worst))


def test_f2_cfg1_logits_and_trajectory(golden_dir):
    """BASELINE config 1 (none, 2L d256 h4 di1024 L256 B2) through the HIP engine, f32 tier."""
    z = np.load(os.path.join(golden_dir, "f2_cfg1.npz"))
    cfg = O.Cfg(1007, 2, 4, 256, 1024, conditioning="none")
    model = make_model(cfg, O.seeded_params(cfg, int(z["weight_seed"])), "fp32")
    tok = torch.from_numpy(z["tokens"]).to(DEV)
    with torch.no_grad():
        lg = model.eval()(tok, None)
    e = relerr(lg[:, z["rows"]], z["logits_rows"])
    assert e < 1e-4, e
    from midiemo.optim import FusedAdamW
    model.train()
    opt = FusedAdamW(model, lr=float(z["traj_lr"]), clip=1.0)
    traj = []
    for step in range(20):
        inp, cond, tgt = O.synthetic_batch(cfg, 2, 256, seed=5000 + step)
        traj.append(model.loss_and_backward(inp.to(DEV), cond.to(DEV), tgt.to(DEV)).item())
        opt.step()
    np.testing.assert_allclose(np.array(traj), z["traj_loss"], rtol=0, atol=3e-3)


def test_bf16_tier_learns_like_the_f32_tier():
    """End-to-end sanity of the bf16 tier as a TRAINING path (hi + lo residual stream, bf16 logits into the loss, bf16
    weight copies refreshed after every optimiser step): on a learnable task -- next-token prediction of periodic
    sequences with a valence-dependent period -- 150 Adam steps must bring the loss far below log(V) and stay within a
    few percent of the exact-f32 tier's trajectory from the same initial weights and batches."""
    from midiemo.optim import FusedAdamW
    cfg = O.Cfg(1007, 2, 4, 128, 256, d_condition=32, conditioning="continuous_concat")
    P = O.seeded_params(cfg, 17)
    B, L = 16, 128

    def batch(step):
        g = torch.Generator().manual_seed(9000 + step)
        period = torch.randint(3, 9, (B, 1), generator=g)
        start = torch.randint(2, 900, (B, 1), generator=g)
        pos = torch.arange(L + 1)[None, :]
        tok = start + (pos % period) * 7                                     # periodic, period 3..8
        cond = torch.stack([(period[:, 0].float() - 5.5) / 3.0, torch.zeros(B)], -1)
        return tok[:, :-1].contiguous().to(DEV), cond.to(DEV), tok[:, 1:].contiguous().to(DEV)

    traj = {}
    for cd in ("fp32", "bf16"):
        model = make_model(cfg, P, cd).train()
        opt = FusedAdamW(model, lr=1e-3, clip=1.0)
        losses = []
        for step in range(150):
            x, c, y = batch(step)
            losses.append(model.loss_and_backward(x, c, y))
            opt.step()
        traj[cd] = torch.stack(losses).cpu().numpy()
    a, b = traj["fp32"], traj["bf16"]
    report("bf16 vs f32 training trajectory (150 steps): start %.3f / %.3f, end %.4f / %.4f, worst rel gap of the last 50 steps %.3f"
           % (a[0], b[0], a[-10:].mean(), b[-10:].mean(), float(np.abs(b[-50:] - a[-50:]).max() / a[-50:].mean())))
    assert a[0] > 6.5 and b[0] > 6.5                                         # ~log(1007) at random init
    assert a[-10:].mean() < 3.5 and b[-10:].mean() < 3.5                     # both tiers learn the task (the first period of a sequence is unpredictable)
    assert abs(b[-10:].mean() - a[-10:].mean()) < 0.05 * a[-10:].mean() + 0.02
    assert np.abs(b[:20] - a[:20]).max() < 0.03                              # early steps: same trajectory


@pytest.mark.parametrize("cd", TIERS)
def test_f3_headline_model_logits(golden_dir, cd):
    """cfg2 model (6L d512 h8 di2048 dc128), B=2, L=1024: logits + loss + per-tensor grad norms."""
    z = np.load(os.path.join(golden_dir, "f3_cfg2.npz"))
    cfg = O.Cfg(1007, 6, 8, 512, 2048, d_condition=128, conditioning="continuous_concat")
    model = make_model(cfg, O.seeded_params(cfg, int(z["weight_seed"])), cd).train()
    inp, cond, tgt = O.synthetic_batch(cfg, 2, 1024, seed=int(z["batch_seed"]))
    with torch.no_grad():
        lg = model(inp.to(DEV), cond.to(DEV))
    e = relerr(lg[:, z["rows"]], z["logits_rows"])
    ref_bf16 = float(z["autocast_bf16_rel_l2_rows"])       # the reference's own bf16-autocast error on these rows
    report("cfg2 (6L d512 h8 L1024 B2) logits rel-L2 vs reference fp32, compute=%s: %.3e "
           "(reference under bf16 autocast: %.3e)" % (cd, e, ref_bf16))
    # f16 tier: north_star's own bound -- logits within 1e-3 rel of the reference -- on the 16-bit tier that runs at the bench's speed
    assert e < {"fp32": 1e-4, "bf16": ref_bf16, "fp16": NORTH_STAR_LOGITS}[cd], (e, ref_bf16)
    if cd == "bf16":                                        # and it is no further from the autocast logits than fp32 is
        e_ac = relerr(lg[:, z["rows"]], z["autocast_bf16_logits_rows"])
        report("cfg2 bf16 logits vs the reference's bf16-autocast logits: %.3e" % e_ac)
        assert e_ac < 1.5 * ref_bf16, e_ac
    loss = fused_loss_and_grads(model, cd, inp.to(DEV), cond.to(DEV), tgt.to(DEV))
    # bf16 loss bound: a logit perturbation of relative size e moves the cross entropy by at most ~e * |logits| (rows of
    # O(1) logits): K_LOSS x the stored autocast logit error, against the hand-picked 5e-3 of round 2
    assert abs(loss.item() - float(z["loss"])) < {"fp32": 5e-5, "bf16": K_LOSS * ref_bf16, "fp16": K_LOSS * NORTH_STAR_LOGITS}[cd], (loss.item(), float(z["loss"]))
    model.link_grads()
    ge_ac = None
    if cd != "fp32":
        # per-tensor bound for the gradient NORMS the fixture stores: | ||g|| - ||g_ref|| | <= ||g - g_ref|| <= K_GRADS x the
        # oracle's own bf16-autocast error for that tensor on this batch (host run of the oracle, fp32 and autocast)
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
        _, _, ge_ac = autocast_train_err(cfg, O.seeded_params(cfg, int(z["weight_seed"])), inp, cond, tgt, cd=cd)
    bad = {}
    for k, p in model.named_parameters():
        if k.endswith("Wk.bias"):
            continue
        gn = float(p.grad.double().norm())
        ref = float(z[f"gradnorm/{k}"])
        lim = 1e-3 if cd == "fp32" else max(K_GRADS_TIER[cd] * ge_ac[k], ULP[cd])
        if abs(gn - ref) > lim * ref + 1e-9:
            bad[k] = (gn, ref, lim)
    assert not bad, bad


def test_dropout_training_runs_and_is_seeded():
    cfg = O.Cfg(1007, 2, 2, 128, 256, d_condition=32, conditioning="continuous_concat")
    model = make_model(cfg, O.seeded_params(cfg, 5), "bf16", dropout=0.1).train()
    tok, cond, tgt = O.synthetic_batch(cfg, 4, 64, seed=1)
    tok, cond, tgt = tok.to(DEV), cond.to(DEV), tgt.to(DEV)
    model.seed_dropout(7)
    l1 = model.loss_and_backward(tok, cond, tgt).item()
    g1 = model.flat_grads.clone()
    model.flat_grads.zero_()
    l2 = model.loss_and_backward(tok, cond, tgt).item()       # next dropout draw
    model.flat_grads.zero_()
    model.seed_dropout(7)
    l3 = model.loss_and_backward(tok, cond, tgt).item()
    # same (seed, counter) -> same masks; the loss sum uses float atomics, so allow 1e-5
    assert abs(l1 - l2) > 1e-4 and abs(l1 - l3) < 1e-5, (l1, l2, l3)
    assert torch.isfinite(g1).all()
    model.eval()
    with torch.no_grad():
        a = model(tok, cond)
        b = model(tok, cond)
    assert torch.equal(a, b)


def test_cpu_model_fails_loudly():
    from midiemo.models.music_transformer import MusicTransformerMulti
    m = MusicTransformerMulti(embedding_dim=64, d_inner=128, d_condition=-1, vocab_size=97, num_layer=1, num_head=2,
                              max_seq=64, dropout=0.0, pad_token=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randint(2, 97, (1, 8)), None)


def test_round5_paths_leave_the_gradients_bit_identical(monkeypatch):
    """The two train-step changes of round 5 are re-orderings / re-encodings, not new arithmetic: the ReLU sign mask (1 bit
    instead of the activations for the FFN_suf dX gate) and the attention backward's key-owned / E-row-owned kernels on two
    streams.  With both switched off the same batch must give the same loss and the same gradients: bit-identical for the
    projection / FFN weight matrices (T = 8192 rows: their products sum the token ranges in a fixed order), to f32 summation
    order for the tensors that are accumulated with atomics (LayerNorm, biases, embedding, the relative-position table).
    dropout on: the masks depend on (seed, counter) only."""
    cfg = O.Cfg(1007, 2, 4, 256, 1024, d_condition=64, conditioning="continuous_concat")
    tok, cond, tgt = O.synthetic_batch(cfg, 8, 1024, seed=3)
    tok, cond, tgt = tok.to(DEV), cond.to(DEV), tgt.to(DEV)
    res = {}
    for arm in ("new", "old"):
        if arm == "old":
            monkeypatch.setenv("MIDIEMO_NO_RELU_MASK", "1")
            monkeypatch.setenv("MIDIEMO_ATTN_BWD_OVERLAP", "0")
        model = make_model(cfg, O.seeded_params(cfg, 9), "bf16", dropout=0.1).train()
        model.seed_dropout(21)
        loss = model.loss_and_backward(tok, cond, tgt)
        ws = next(iter(model._ws.values()))
        assert (ws.layers[0].rmask is not None) == (arm == "new") and model.attn_bwd_overlap == (arm == "new")
        model.link_grads()
        res[arm] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters()})
    assert abs(res["new"][0] - res["old"][0]) < 1e-5
    exact = 0
    for k, g in res["new"][1].items():
        if g.dim() == 2 and k.endswith(".weight") and ("rga.W" in k or "rga.fc" in k or "FFN_" in k):
            assert torch.equal(g, res["old"][1][k]), k
            exact += 1
        elif not k.endswith("Wk.bias"):
            assert relerr(g, res["old"][1][k]) < 1e-5, k
    assert exact == 2 * 6


def test_full_size_c2_properties_bf16():
    """BASELINE config 2 at its full size (B = 32 x L = 1024, bf16), through properties that need no oracle run:
    batch independence (bit-exact), causality (bit-exact), gradient additivity over a batch split, and the
    engine's loss against a float64 cross-entropy of its own logits."""
    from midiemo.models.build_model import build_model
    torch.manual_seed(0)
    args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
                conditioning="continuous_concat", compute_dtype="bf16")
    model, _ = build_model(args)
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(11)
    B, L = 32, 1024
    tok = torch.randint(2, 1007, (B, L + 1), generator=g)
    tok[5, 900:] = 0                                            # one padded tail
    x, y = tok[:, :-1].contiguous().to(DEV), tok[:, 1:].contiguous().to(DEV)
    cond = (torch.rand(B, 2, generator=g) * 2 - 1).to(DEV)
    with torch.no_grad():
        full = model(x, cond).float()
        assert full.shape == (B, L, 1007) and bool(torch.isfinite(full).all())
        # batch independence: rows 4..7 alone give bit-identical logits
        part = model(x[4:8], cond[4:8]).float()
        assert torch.equal(part, full[4:8])
        # causality: changing the tokens from position 700 on leaves every earlier logit bit-identical
        x2 = x.clone()
        x2[:, 700:] = torch.randint(2, 1007, (B, L - 700), generator=g).to(DEV)
        alt = model(x2, cond).float()
        assert torch.equal(alt[:, :700], full[:, :700])
        assert not torch.equal(alt[:, 700:], full[:, 700:])
    # loss of the fused path == float64 CE of the engine's own logits (ignore_index = 0)
    model.zero_grad_flat() if hasattr(model, "zero_grad_flat") else model.flat_grads.zero_()
    loss = model.loss_and_backward(x, cond, y)
    ref = torch.nn.functional.cross_entropy(full.double().reshape(-1, 1007), y.reshape(-1), ignore_index=0)
    assert abs(float(loss) - float(ref)) < 2e-4 * abs(float(ref)), (float(loss), float(ref))
    g_full = model.flat_grads.clone()
    assert bool(torch.isfinite(g_full).all()) and float(g_full.norm()) > 0
    # additivity: mean-over-valid-targets gradients of the two half batches, recombined with their target counts
    n = [int((y[:16] != 0).sum()), int((y[16:] != 0).sum())]
    model.flat_grads.zero_()
    model.loss_and_backward(x[:16], cond[:16], y[:16], grad_scale=n[0] / (n[0] + n[1]))
    model.loss_and_backward(x[16:], cond[16:], y[16:], grad_scale=n[1] / (n[0] + n[1]))
    e = relerr(model.flat_grads, g_full)
    report("full-size C2 bf16: grad additivity rel %.2e, loss %.5f vs f64 CE of own logits %.5f" % (e, float(loss), float(ref)))
    assert e < 2e-3, e


@pytest.mark.parametrize("cd", TIERS)
def test_f6_regression_model_vs_reference_golden(golden_dir, cd):
    """MusicRegression (evaluation model, forward only) against outputs of the imported reference."""
    from midiemo.models.music_transformer import MusicRegression
    z = np.load(os.path.join(golden_dir, "f6_regression.npz"))
    V, N, H, d, di, M = [int(x) for x in z["cfg"]]
    if d // H not in (32, 48, 64):
        pytest.skip("fixture head dim")
    P = O.regression_seeded_params(O.regression_param_shapes(V, N, d, di, d // H, M), int(z["seed"][0]))
    model = MusicRegression(embedding_dim=d, d_inner=di, vocab_size=V, num_layer=N, num_head=H, max_seq=M, dropout=0.0,
                            pad_token=0, output_size=2, compute_dtype=cd)
    res = model.load_state_dict(P, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(DEV).eval()
    worst = 0.0
    for L in (1, 7, 33, 64):
        y = model(torch.from_numpy(z["tok_%d" % L]).to(DEV)).cpu()
        assert y.shape == (3, 2)
        worst = max(worst, float((y.double() - torch.from_numpy(z["y_%d" % L]).double()).abs().max()))
    report("F6 regression[%s]: max abs err of tanh outputs %.2e" % (cd, worst))
    assert worst < {"fp32": 2e-5, "bf16": 3e-2, "fp16": 4e-3}[cd], worst


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_regression_l1_loss_and_grads_vs_oracle(golden_dir, cd):
    """Training path of MusicRegression: L1 loss and every parameter gradient (bidirectional attention backward,
    head on position 0) against the oracle's autograd; then one fused Adam step moves the loss down."""
    from midiemo.models.music_transformer import MusicRegression
    from midiemo.optim import FusedAdamW
    V, N, H, d, di, M = 1008, 2, 2, 128, 256, 128
    shapes = O.regression_param_shapes(V, N, d, di, d // H, M)
    P = O.regression_seeded_params(shapes, 5)
    model = MusicRegression(embedding_dim=d, d_inner=di, vocab_size=V, num_layer=N, num_head=H, max_seq=M, dropout=0.0,
                            pad_token=0, output_size=2, compute_dtype=cd)
    model.load_state_dict(P, strict=True)
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(9)
    tok = torch.randint(1, V, (4, 97), generator=g)
    tgt = torch.rand(4, 2, generator=g) * 2 - 1
    cfg = O.Cfg(V, N, H, d, di, max_seq=M)
    loss_ref, _, G = O.regression_loss_and_grads(cfg, {k: v.double() for k, v in P.items()}, tok, tgt.double())
    model.flat_grads.zero_()
    loss = model.loss_and_backward(tok.to(DEV), tgt.to(DEV))
    model.link_grads()
    errs = {k: relerr(p.grad, G[k]) for k, p in model.named_parameters() if float(G[k].norm()) > 1e-9}
    worst = max(errs.values())
    report("regression grads[%s]: loss %.6f (oracle %.6f), worst grad rel %.2e (%s)" %
           (cd, float(loss), float(loss_ref), worst, max(errs, key=errs.get)))
    if cd == "fp32":
        assert abs(float(loss) - float(loss_ref)) < 1e-5
        assert worst < 3e-4, errs
    else:                                                  # bounds: the oracle's own bf16-autocast error on this batch
        P32 = {k: v.float() for k, v in P.items()}
        l32, _, G32 = O.regression_loss_and_grads(cfg, P32, tok, tgt)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lac, _, Gac = O.regression_loss_and_grads(cfg, P32, tok, tgt)
        assert abs(float(loss) - float(loss_ref)) <= K_LOSS * abs(float(lac) - float(l32)) + 1e-4
        bad = {k: (e, relerr(Gac[k].float(), G32[k])) for k, e in errs.items()
               if not k.endswith("Wk.bias") and e > max(K_GRADS * relerr(Gac[k].float(), G32[k]), BF16_ULP)}
        assert not bad, bad
    # the key / value biases get exactly-cancelling or tiny gradients like in the language model; everything else learns
    opt = FusedAdamW(model, lr=2e-5)                       # the first Adam step moves every weight by lr
    opt.step()
    model.flat_grads.zero_()
    loss2 = model.loss_and_backward(tok.to(DEV), tgt.to(DEV), backward=False)
    assert float(loss2) < float(loss)
