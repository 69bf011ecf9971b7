"""Golden-vector generator -- runs ONLY in the build container, where the
reference checkout is mounted read-only at /root/reference.

It imports the reference's own Python (src/models, src/generate.py), feeds it
seeded weights/inputs and writes inputs + expected outputs as small .npz files
under tests/golden/.  Nothing from the reference (source, bytecode) is copied:
fixtures are data only.  The GPU box never runs this script.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures.py
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
if not os.path.isdir(REF):
    raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
sys.modules["pretty_midi"] = MagicMock()
sys.modules["pypianoroll"] = MagicMock()
sys.path.insert(0, REF)

from models.music_multi import MusicTransformerMulti, RelativeGlobalAttention  # noqa: E402  (reference)
from models.music_continuous_token import MusicTransformerContinuousToken      # noqa: E402  (reference)
from models.build_model import build_model as ref_build_model                   # noqa: E402  (reference)
from data.data_processing import get_maps as ref_get_maps                       # noqa: E402  (reference)
import generate as ref_generate                                                 # noqa: E402  (reference)

from oracle import ref_model as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)

MODES = ["none", "discrete_token", "continuous_token", "continuous_concat"]


def ref_model_for(cfg: O.Cfg, params):
    if cfg.conditioning == "continuous_token":
        m = MusicTransformerContinuousToken(embedding_dim=cfg.d_model, d_inner=cfg.d_inner,
                                            vocab_size=cfg.vocab_size, num_layer=cfg.n_layer,
                                            num_head=cfg.n_head, max_seq=cfg.max_seq, dropout=0.0,
                                            pad_token=cfg.pad_token)
    else:
        m = MusicTransformerMulti(embedding_dim=cfg.d_model, d_inner=cfg.d_inner,
                                  d_condition=cfg.d_condition if cfg.d_condition > 0 else -1,
                                  vocab_size=cfg.vocab_size, num_layer=cfg.n_layer,
                                  num_head=cfg.n_head, max_seq=cfg.max_seq, dropout=0.0,
                                  pad_token=cfg.pad_token)
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def subsample(a: np.ndarray) -> np.ndarray:
    """Tensors > 4096 elements are stored as every 7th element (flat) to keep fixtures small."""
    f = a.reshape(-1)
    return f[::7].copy() if f.size > 4096 else f.copy()


def batch_with_pads(cfg, B, L, seed):
    inp, cond, tgt = O.synthetic_batch(cfg, B, L, seed)
    if inp.shape[1] >= 5:
        # last row: trailing PAD keys (and PAD targets), as filter_collate pads short samples
        npad = max(1, inp.shape[1] // 4)
        inp[-1, -npad:] = cfg.pad_token
        tgt[-1, -npad:] = cfg.pad_token
        tgt[-1, -npad - 1] = cfg.pad_token
    return inp, cond, tgt


# ---------------------------------------------------------------- F1: tiny, all modes
def make_f1():
    for mode in MODES:
        V = 107 if mode == "discrete_token" else 97
        cfg = O.Cfg(V, 2, 2, 64, 128, d_condition=16, conditioning=mode, max_seq=64)
        params = O.seeded_params(cfg, seed=11)
        model = ref_model_for(cfg, params)
        model.eval()
        rec = {"cfg": np.array([V, 2, 2, 64, 128, cfg.d_condition, 64]), "weight_seed": np.array(11)}
        for L in (1, 7, 33, 64):
            Lm = L if mode != "continuous_token" else max(L, 3)
            inp, cond, tgt = batch_with_pads(cfg, 3, Lm, seed=100 + L)
            with torch.no_grad():
                lg = model(inp, cond)
            loss = torch.nn.functional.cross_entropy(lg.reshape(-1, V), tgt.reshape(-1), ignore_index=0)
            rec[f"L{L}_tokens"] = inp.numpy()
            rec[f"L{L}_cond"] = cond.numpy()
            rec[f"L{L}_target"] = tgt.numpy()
            rec[f"L{L}_logits"] = lg.numpy().astype(np.float32)
            rec[f"L{L}_loss"] = np.array(loss.item(), dtype=np.float64)
        # PAD at position 0 -> fully masked first row -> NaN everywhere (reference behaviour)
        inp, cond, tgt = O.synthetic_batch(cfg, 2, 9, seed=7)
        inp[0, 0] = 0
        with torch.no_grad():
            lg = model(inp, cond)
        rec["pad0_tokens"] = inp.numpy()
        rec["pad0_cond"] = cond.numpy()
        rec["pad0_isnan"] = torch.isnan(lg).numpy()

        # grads + 3 optimiser steps (clip 1.0 + Adam lr 2e-5), batch L=33, train mode w/ dropout 0
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=2e-5)
        names = [k for k, _ in model.named_parameters()]
        prev = {k: v.detach().clone() for k, v in model.named_parameters()}
        for step in range(1, 4):
            inp, cond, tgt = batch_with_pads(cfg, 3, 33, seed=200 + step)
            rec[f"opt{step}_tokens"] = inp.numpy()
            rec[f"opt{step}_cond"] = cond.numpy()
            rec[f"opt{step}_target"] = tgt.numpy()
            lg = model(inp, cond)
            loss = torch.nn.CrossEntropyLoss(ignore_index=0)(lg.reshape(-1, V), tgt.reshape(-1))
            loss.backward()
            rec[f"opt{step}_loss"] = np.array(loss.item(), dtype=np.float64)
            if step == 1:
                for k, p in model.named_parameters():
                    g = p.grad.detach().numpy()
                    rec[f"grad/{k}"] = subsample(g)
                    rec[f"gradnorm/{k}"] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            rec[f"opt{step}_gradnorm"] = np.array(float(gn), dtype=np.float64)
            opt.step()
            model.zero_grad()
            if step in (1, 3):
                for k, p in model.named_parameters():
                    # normalised update (P_after - P_step0) / lr : O(1) numbers, meaningful in fp32
                    upd = (p.detach().double() - prev[k].double()) / 2e-5
                    rec[f"upd{step}/{k}"] = subsample(upd.numpy().astype(np.float32))
        rec["param_names"] = np.array(names)
        np.savez_compressed(os.path.join(OUT, f"f1_{mode}.npz"), **rec)
        print("F1", mode, "ok")


# ---------------------------------------------------------------- F2: BASELINE config 1 (CPU plumbing case)
def make_f2():
    cfg = O.Cfg(1007, 2, 4, 256, 1024, conditioning="none")
    params = O.seeded_params(cfg, seed=21)
    args = dict(vocab_size=1007, n_layer=2, n_head=4, d_model=256, d_inner=1024, dropout=0.0,
                d_condition=-1, conditioning="none")
    model, _ = ref_build_model(args)
    model.load_state_dict(params, strict=True)
    model.eval()
    inp, cond, tgt = O.synthetic_batch(cfg, 2, 256, seed=1234)
    with torch.no_grad():
        lg = model(inp, cond)
    rows = np.array([0, 1, 2, 100, 127, 128, 254, 255])
    rec = {"tokens": inp.numpy(), "target": tgt.numpy(), "rows": rows,
           "logits_rows": lg[:, rows].numpy().astype(np.float32),
           "loss": np.array(torch.nn.functional.cross_entropy(lg.reshape(-1, 1007), tgt.reshape(-1),
                                                              ignore_index=0).item()),
           "weight_seed": np.array(21)}
    # 20-step loss trajectory, lr 1e-3 so the curve is sensitive (clip 1.0, Adam), fresh batch per step
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    traj = []
    for step in range(20):
        inp, cond, tgt = O.synthetic_batch(cfg, 2, 256, seed=5000 + step)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(model(inp, cond).reshape(-1, 1007), tgt.reshape(-1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        model.zero_grad()
        traj.append(loss.item())
    rec["traj_lr"] = np.array(1e-3)
    rec["traj_loss"] = np.array(traj, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "f2_cfg1.npz"), **rec)
    print("F2 ok", traj[0], traj[-1])


# ---------------------------------------------------------------- F3: headline model shape (cfg2), B=2
def make_f3():
    cfg = O.Cfg(1007, 6, 8, 512, 2048, d_condition=128, conditioning="continuous_concat")
    params = O.seeded_params(cfg, seed=31)
    args = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0,
                d_condition=128, conditioning="continuous_concat")
    model, _ = ref_build_model(args)
    model.load_state_dict(params, strict=True)
    model.train()
    inp, cond, tgt = O.synthetic_batch(cfg, 2, 1024, seed=1234)
    lg = model(inp, cond)
    loss = torch.nn.CrossEntropyLoss(ignore_index=0)(lg.reshape(-1, 1007), tgt.reshape(-1))
    loss.backward()
    rs = np.random.RandomState(3)
    rows = np.sort(rs.choice(1024, 32, replace=False))
    rec = {"rows": rows, "logits_rows": lg.detach()[:, rows].numpy().astype(np.float32),
           "loss": np.array(loss.item()), "weight_seed": np.array(31), "batch_seed": np.array(1234)}
    # The reference's OWN reduced-precision error on this very batch: the same module under torch.autocast(bfloat16)
    # (train.py:281 wraps the model call in autocast; on this CPU-only container the bf16 CPU autocast is what can run).
    # The bf16 tier of the HIP path is gated on being at least this close to the fp32 logits.
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        lg_ac = model(inp, cond)
    lg_ac = lg_ac.float()
    ref32 = lg.detach().double()
    rec["autocast_bf16_logits_rows"] = lg_ac[:, rows].numpy().astype(np.float32)
    rec["autocast_bf16_rel_l2_all"] = np.array(float((lg_ac.double() - ref32).norm() / ref32.norm()))
    rec["autocast_bf16_rel_l2_rows"] = np.array(float((lg_ac.double()[:, rows] - ref32[:, rows]).norm() / ref32[:, rows].norm()))
    rec["autocast_bf16_max_abs"] = np.array(float((lg_ac.double() - ref32).abs().max()))
    print("F3 reference bf16-autocast rel-L2 vs its fp32 logits: all %.3e, stored rows %.3e, max abs %.3e" %
          (rec["autocast_bf16_rel_l2_all"], rec["autocast_bf16_rel_l2_rows"], rec["autocast_bf16_max_abs"]))
    for k, p in model.named_parameters():
        rec[f"gradnorm/{k}"] = np.array(np.sqrt((p.grad.double() ** 2).sum().item()))
    np.savez_compressed(os.path.join(OUT, "f3_cfg2.npz"), **rec)
    print("F3 ok", loss.item())


# ---------------------------------------------------------------- F4: greedy decode through the reference generate()
def decode_maps(mode):
    maps = ref_get_maps()
    if mode == "discrete_token":
        extra = sorted([f"<{c}{b}>" for c in "VA" for b in (-2, -1, 0, 1, 2)])   # data/loader.py:58-75
        lst = list(maps["idx2tuple"].values()) + extra
        maps["idx2tuple"] = {i: v for i, v in enumerate(lst)}
        maps["tuple2idx"] = {v: i for i, v in enumerate(lst)}
    return maps


def make_f4():
    conds = [[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]]          # train.py:361-366
    rec = {}
    for mode in MODES:
        V = 1017 if mode == "discrete_token" else 1007
        cfg = O.Cfg(V, 2, 2, 64, 128, d_condition=16, conditioning=mode)
        params = O.seeded_params(cfg, seed=41)
        args = dict(vocab_size=V, n_layer=2, n_head=2, d_model=64, d_inner=128, dropout=0.0,
                    d_condition=16 if mode == "continuous_concat" else -1, conditioning=mode)
        model, _ = ref_build_model(args)
        model.load_state_dict(params, strict=True)
        maps = decode_maps(mode)
        disc = None
        if mode == "discrete_token":
            bins = np.linspace(-1 - 1e-12, 1 + 1e-12, 6)
            disc = [[f"<V{np.searchsorted(bins, v, side='right') - 1 - 2}>",
                     f"<A{np.searchsorted(bins, a, side='right') - 1 - 2}>"] for v, a in conds]
            rec[f"{mode}_prefix"] = np.array([[maps["tuple2idx"][s] for s in d] for d in disc]).T
        for tag, gen_len, mil in (("noslide", 48, 48), ("slide", 40, 24)):
            captured = []
            orig = ref_generate.ind_tensor_to_str

            def spy(x, a, b, _c=captured, _o=orig):
                _c.append(x.clone().cpu().numpy())
                return _o(x, a, b)
            ref_generate.ind_tensor_to_str = spy
            try:
                ref_generate.generate(model, maps, torch.device("cpu"), "/tmp/none", mode,
                                      discrete_conditions=disc,
                                      continuous_conditions=None if mode == "none" else conds,
                                      max_input_len=mil, amp=False, gen_len=gen_len, top_k=1,
                                      debug=True, min_n_instruments=0,
                                      primers=[["<START>"]] * 4 if mode == "none" else [["<START>"]])
            finally:
                ref_generate.ind_tensor_to_str = orig
            ids = np.stack(captured, axis=1)                # [T, B]
            rec[f"{mode}_{tag}_ids"] = ids
            rec[f"{mode}_{tag}_cfg"] = np.array([gen_len, mil])
        print("F4", mode, "ok", ids[:6, 0])
    rec["conds"] = np.array(conds, dtype=np.float32)
    rec["weight_seed"] = np.array(41)
    np.savez_compressed(os.path.join(OUT, "f4_decode.npz"), **rec)


# ---------------------------------------------------------------- F4h: the same at the HEADLINE geometry (BASELINE config 2 / 5 model)
def _make_f4h_variant(mode, gen_len, fname, seed=43, min_margin=1e-4):
    """Greedy ids of the reference's own generate() on the 6-layer d512 8-head (dh 64) d_inner 2048 model of BASELINE configs
    2 / 4 / 5: 4 (valence, arousal) pairs x gen_len tokens, no slide (max_input_len = gen_len).  Weights =
    O.seeded_params(cfg, seed), regenerated by the test, not stored.  Also stored: the top-1 / top-2 logit margin of every step
    (from a second pass through the reference model over the generated sequences) -- the test's f32 criterion is bit-exact
    ids, and the margin says how much room that has."""
    conds = [[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]]          # train.py:361-366
    V = 1017 if mode == "discrete_token" else 1007
    dc = 128 if mode == "continuous_concat" else -1
    cfg = O.Cfg(V, 6, 8, 512, 2048, d_condition=dc, conditioning=mode)
    params = O.seeded_params(cfg, seed=seed)
    args = dict(vocab_size=V, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=dc, conditioning=mode)
    model, _ = ref_build_model(args)
    model.load_state_dict(params, strict=True)
    model.eval()
    maps = decode_maps(mode)
    disc, prefix = None, None
    if mode == "discrete_token":
        bins = np.linspace(-1 - 1e-12, 1 + 1e-12, 6)
        disc = [[f"<V{np.searchsorted(bins, v, side='right') - 1 - 2}>",
                 f"<A{np.searchsorted(bins, a, side='right') - 1 - 2}>"] for v, a in conds]
        prefix = np.array([[maps["tuple2idx"][s_] for s_ in d] for d in disc]).T          # [2, B]
    # generate() takes 2 off max_input_len for the token-conditioned modes (generate.py:76,81): + 2 keeps the window from sliding
    mil = gen_len + (2 if mode in ("discrete_token", "continuous_token") else 0)
    captured = []
    orig = ref_generate.ind_tensor_to_str

    def spy(x, a, b, _c=captured, _o=orig):
        _c.append(x.clone().cpu().numpy())
        return _o(x, a, b)
    ref_generate.ind_tensor_to_str = spy
    try:
        ref_generate.generate(model, maps, torch.device("cpu"), "/tmp/none", mode, discrete_conditions=disc,
                              continuous_conditions=None if mode == "none" else conds, max_input_len=mil, amp=False,
                              gen_len=gen_len, top_k=1, debug=True, min_n_instruments=0, primers=[["<START>"]])
    finally:
        ref_generate.ind_tensor_to_str = orig
    ids = np.stack(captured, axis=1)                        # [T, B], row 0 = <START>
    assert ids.shape == (gen_len, 4), ids.shape
    # margins: teacher-force the generated sequences through the reference model, mask what generate() masks
    inp = ids.T[:, :-1]
    if prefix is not None:                                  # discrete_token: the two bin tokens sit in front of every window (generate.py:105-107)
        inp = np.concatenate([prefix.T, inp], axis=1)
    with torch.no_grad():
        lg = model(torch.tensor(inp), torch.tensor(conds, dtype=torch.float32)).double()     # [B, T-1 (+2), V]
    if prefix is not None:
        lg = lg[:, 2:]
    for tok, idx in maps["tuple2idx"].items():             # generate.py:57,131-136: every symbol that starts with "<" is excluded
        if isinstance(tok, str) and tok[:1] == "<":
            lg[:, :, idx] = -float("inf")
    # margin of the GENERATED id over the best other candidate (teacher-forced logits differ from the step-by-step ones by f32
    # summation order, ~1e-6: a generated id that is not the teacher-forced arg-max sat on a tie and shows as a negative margin)
    gen = torch.tensor(ids[1:].T.astype(np.int64))[:, :, None]                  # [B, T-1, 1]
    chosen = lg.gather(2, gen)[:, :, 0]
    others = lg.scatter(2, gen, -float("inf")).max(dim=-1).values
    margin = (chosen - others).numpy().T                                        # [T-1, B]
    scale = float(lg[torch.isfinite(lg)].abs().max())
    print(fname, "seed", seed, ": ids", ids[:6, 0], "min margin %.3e (logit scale %.2f)" % (margin.min(), scale))
    if margin.min() < min_margin:
        return False
    rec = dict(ids=ids.astype(np.int16), conds=np.array(conds, dtype=np.float32), weight_seed=np.array(seed),
               margin=margin.astype(np.float32), logit_scale=np.array(scale), max_input_len=np.array(mil))
    if prefix is not None:
        rec["prefix"] = prefix.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, fname), **rec)
    return True


def make_f4h():
    """round 5: continuous_concat, 4 x 128 tokens"""
    _make_f4h_variant("continuous_concat", 128, "f4h_decode_cfg2.npz")


def make_f4h512():
    """round 6: continuous_concat, 4 x 512 tokens -- contexts past the first key-split chunk of the decode attention"""
    _make_f4h_variant("continuous_concat", 512, "f4h_decode_cfg2_512.npz")


def make_f4hd():
    """round 6: the discrete_token headline model (V = 1017, BASELINE config 4's model), 4 x 256 tokens"""
    # the weight seed is the first of 44, 45, ... whose greedy stream has no near-tie (margin >= 1e-4 = 100 x the f32 noise between
    # two evaluation orders of the same logits): a golden for BIT-exact ids must not hinge on a coin flip of the reference itself
    for seed in range(44, 60):
        if _make_f4h_variant("discrete_token", 256, "f4h_decode_cfg4_256.npz", seed=seed):
            break


# ---------------------------------------------------------------- F5: attention core in fp64 through the reference's skewing code
def make_f5():
    rs = np.random.RandomState(51)
    B, H, L, dh, M = 2, 2, 40, 32, 64
    rga = RelativeGlobalAttention(h=H, d=H * dh, max_seq=M).double()
    q = torch.tensor(rs.standard_normal((B, H, L, dh)), requires_grad=True)
    k = torch.tensor(rs.standard_normal((B, H, L, dh)), requires_grad=True)
    v = torch.tensor(rs.standard_normal((B, H, L, dh)), requires_grad=True)
    E = torch.tensor(rs.standard_normal((M, dh)), requires_grad=True)
    dO = torch.tensor(rs.standard_normal((B, H, L, dh)))
    pad = torch.zeros(B, L, dtype=torch.bool)
    pad[1, -6:] = True
    # the reference's own lines music_multi.py:211-232, driven with explicit q,k,v,E
    rga.len_k = L
    rga.len_q = L
    e = E[max(0, M - L):, :]
    QE = torch.einsum('bhld,md->bhlm', [q, e])
    QE = rga._qe_masking(QE)
    Srel = rga._skewing(QE)
    logits = (torch.matmul(q, k.permute(0, 1, 3, 2)) + Srel) / np.sqrt(dh)
    l = torch.arange(L)
    mask = (l[None, :] > l[:, None])[None] | pad[:, None, :]
    logits = logits + torch.zeros(B, 1, L, L, dtype=torch.float64).masked_fill(mask[:, None], float("-inf"))
    P = torch.softmax(logits, -1)
    Oo = torch.matmul(P, v)
    (Oo * dO).sum().backward()
    np.savez_compressed(os.path.join(OUT, "f5_attn_core.npz"),
                        q=q.detach().numpy(), k=k.detach().numpy(), v=v.detach().numpy(), E=E.detach().numpy(),
                        dO=dO.numpy(), pad=pad.numpy(), O=Oo.detach().numpy(),
                        lse=torch.logsumexp(logits, -1).detach().numpy(), srel=Srel.detach().numpy(),
                        dq=q.grad.numpy(), dk=k.grad.numpy(), dv=v.grad.numpy(), dE=E.grad.numpy())
    print("F5 ok")


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5"]
    for w in which:
        globals()["make_" + w]()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("fixtures total bytes:", tot)
