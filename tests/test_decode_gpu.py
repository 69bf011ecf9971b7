"""KV-cached greedy decode (GPU) against token streams captured from the reference's own
generate() (tests/golden/f4_decode.npz): bit-exact ids, all four conditioning modes, with and
without the sliding window; plus cache == full recompute."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ["none", "discrete_token", "continuous_token", "continuous_concat"]


def setup(mode, cd, golden_dir):
    import generate as G
    from midiemo.models.build_model import build_model
    from midiemo.vocab import get_maps
    z = np.load(os.path.join(golden_dir, "f4_decode.npz"))
    V = 1017 if mode == "discrete_token" else 1007
    cfg = O.Cfg(V, 2, 2, 64, 128, d_condition=16, conditioning=mode)
    args = dict(vocab_size=V, n_layer=2, n_head=2, d_model=64, d_inner=128, dropout=0.0,
                d_condition=16 if mode == "continuous_concat" else -1, conditioning=mode, compute_dtype=cd)
    model, _ = build_model(args)
    model.load_state_dict(O.seeded_params(cfg, int(z["weight_seed"])))
    maps = get_maps(n_emotion_bins=5 if mode == "discrete_token" else 0)
    conds = z["conds"].tolist()
    disc = None
    if mode == "discrete_token":
        bins = np.linspace(-1 - 1e-12, 1 + 1e-12, 6)
        disc = [[f"<V{np.searchsorted(bins, v, side='right') - 3}>", f"<A{np.searchsorted(bins, a, side='right') - 3}>"]
                for v, a in conds]
        pref = np.array([[maps["tuple2idx"][s] for s in d] for d in disc]).T
        assert np.array_equal(pref, z["discrete_token_prefix"])          # vocab restatement == reference maps
    return G, model.to("cuda"), maps, conds, disc, z


def run(G, model, maps, mode, conds, disc, gen_len, mil, use_cache, top_k=1):
    return G.generate(model, maps, torch.device("cuda"), "/tmp/none", mode, discrete_conditions=disc,
                      continuous_conditions=None if mode == "none" else conds, max_input_len=mil, amp=False,
                      gen_len=gen_len, top_k=top_k, debug=True, min_n_instruments=0,
                      primers=[["<START>"]] * 4 if mode == "none" else [["<START>"]], use_cache=use_cache,
                      return_ids=True).numpy()


@pytest.mark.parametrize("mode", MODES)
def test_greedy_ids_bit_exact_vs_reference(golden_dir, mode):
    G, model, maps, conds, disc, z = setup(mode, "fp32", golden_dir)
    for tag in ("noslide", "slide"):
        gen_len, mil = [int(x) for x in z[f"{mode}_{tag}_cfg"]]
        ids = run(G, model, maps, mode, conds, disc, gen_len, mil, use_cache=True)
        assert np.array_equal(ids, z[f"{mode}_{tag}_ids"]), (mode, tag)
        ids_nc = run(G, model, maps, mode, conds, disc, gen_len, mil, use_cache=False)
        assert np.array_equal(ids_nc, z[f"{mode}_{tag}_ids"]), (mode, tag, "no-cache")


F4H = {"cc128": ("f4h_decode_cfg2.npz", "continuous_concat"), "cc512": ("f4h_decode_cfg2_512.npz", "continuous_concat"),
       "dt256": ("f4h_decode_cfg4_256.npz", "discrete_token")}


@pytest.mark.parametrize("which", list(F4H))
def test_headline_geometry_greedy_ids_vs_reference(golden_dir, which):
    """The 6-layer d512 8-head (dh 64) d_inner 2048 model of BASELINE configs 2 / 4 / 5 against the REFERENCE at its own
    geometry: greedy tokens captured from the reference's generate() for the 4 standard (valence, arousal) pairs
    (oracle/make_fixtures.py make_f4h / make_f4h512 / make_f4hd; weights = O.seeded_params(cfg, seed), regenerated here):
    continuous_concat 4 x 128 (round 5), continuous_concat 4 x 512 (round 6: contexts past the first chunk of every key split
    of the decode attention, default nsplit) and the discrete_token headline model (V = 1017, two bin tokens in front) 4 x 256.
    f32 tier: ids bit-exact through the cached decode (DecodeSession under generate()) AND through the no-cache full
    recompute; the fixture's smallest margin is > 10 x the tier's logit error.  16-bit tiers (bf16, f16): the derived
    teacher-forcing criterion -- every step whose f32 top-2 gap exceeds twice K_LOGITS x the oracle's own autocast error at that
    step picks the reference's token; free-running agreement is printed (f16, the reference's own autocast dtype: expected far above bf16's)."""
    import generate as G
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    from midiemo.vocab import get_maps
    fname, mode = F4H[which]
    z = np.load(os.path.join(golden_dir, fname))
    ref_ids = z["ids"].astype(np.int64)                      # [T, 4]
    conds = z["conds"].tolist()
    V = 1017 if mode == "discrete_token" else 1007
    dc = 128 if mode == "continuous_concat" else -1
    cfg = O.Cfg(V, 6, 8, 512, 2048, d_condition=dc, conditioning=mode)
    P = O.seeded_params(cfg, int(z["weight_seed"]))
    maps = get_maps(n_emotion_bins=5 if mode == "discrete_token" else 0)
    disc, prefix = None, None
    if mode == "discrete_token":
        bins = np.linspace(-1 - 1e-12, 1 + 1e-12, 6)
        disc = [[f"<V{np.searchsorted(bins, v, side='right') - 3}>", f"<A{np.searchsorted(bins, a, side='right') - 3}>"] for v, a in conds]
        prefix = torch.tensor([[maps["tuple2idx"][s_] for s_ in d] for d in disc])      # [4, 2]
        assert np.array_equal(prefix.numpy().T, z["prefix"])
    models = {}
    for cd in ("fp32", "bf16", "fp16"):
        m, _ = build_model(dict(vocab_size=V, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=dc,
                                conditioning=mode, compute_dtype=cd))
        m.load_state_dict(P)
        models[cd] = m.to("cuda").eval()
    n = ref_ids.shape[0]
    mil = int(z["max_input_len"]) if "max_input_len" in z else n
    ids = run(G, models["fp32"], maps, mode, conds, disc, n, mil, use_cache=True)
    assert np.array_equal(ids, ref_ids), ("cached decode", np.argwhere(ids != ref_ids)[:4])
    ids_nc = run(G, models["fp32"], maps, mode, conds, disc, n, mil, use_cache=False)
    assert np.array_equal(ids_nc, ref_ids), ("full recompute", np.argwhere(ids_nc != ref_ids)[:4])
    # the f32 margin the bit-exactness rests on, measured on this run's own logits (teacher-forced through the cache)
    cond = torch.tensor(conds, dtype=torch.float32)
    toks = torch.from_numpy(ref_ids.T.copy())                # [4, T]
    feed = toks if prefix is None else torch.cat([prefix, toks], 1)      # discrete_token: the bin tokens occupy positions 0, 1
    sh = 0 if prefix is None else 2
    nv = 1007                                                # candidates: ids 2 .. 1006 (every "<...>" symbol is excluded, generate.py:57)
    lg32 = O.forward(cfg, P, feed, cond if mode.startswith("continuous") else torch.full((4, 2), float("nan")))
    c_dev = cond.cuda() if mode.startswith("continuous") else None

    def cached_logits(model):
        sess = DecodeSession(model, 4)
        out = []
        for t in range(feed.shape[1] - 1):
            lg = sess.step(feed[:, t].cuda(), c_dev).float().cpu()
            if t >= sh:
                out.append(lg[:, 2:nv])
        return out                                           # out[t] predicts toks[:, t + 1]
    worst = max(float((lg - lg32[:, t + sh, 2:nv]).abs().max()) for t, lg in enumerate(cached_logits(models["fp32"])))
    assert worst < 0.1 * float(z["margin"].min()), (worst, float(z["margin"].min()))
    print("headline-geometry decode %s, f32: 4 x %d greedy ids bit-exact vs the reference (cache and recompute); max |logit error| %.2e "
          "against a smallest margin of %.2e" % (which, n, worst, float(z["margin"].min())))
    K_LOGITS = 1.25
    for cd, acdt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        idb = run(G, models[cd], maps, mode, conds, disc, n, mil, use_cache=True)
        same = idb == ref_ids
        with torch.autocast("cpu", dtype=acdt):
            lgac = O.forward(cfg, P, feed, cond if mode.startswith("continuous") else torch.full((4, 2), float("nan"))).float()
        agree = stable = 0
        for t, lg in enumerate(cached_logits(models[cd])):
            r32 = lg32[:, t + sh, 2:nv]
            err_bound = K_LOGITS * (lgac[:, t + sh, 2:nv] - r32).abs().max(-1).values
            top2 = r32.topk(2, dim=-1)
            gap = top2.values[:, 0] - top2.values[:, 1]
            pick, want = lg.argmax(-1), top2.indices[:, 0]
            assert (want + 2 == toks[:, t + 1]).all()
            must = gap > 2 * err_bound
            assert (pick[must] == want[must]).all(), (cd, t, pick, want, gap, err_bound)
            agree += int((pick == want).sum())
            stable += int(must.sum())
        line = ("headline-geometry decode %s, %s: %d / %d tokens equal free-running; teacher-forced %d / %d steps pick the reference "
                "token, all %d steps with a gap above twice the derived bound agree" % (which, cd, int(same.sum()), same.size, agree, 4 * (n - 1), stable))
        print(line)
        try:
            with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass


@pytest.mark.parametrize("mode", ["continuous_concat", "discrete_token"])
def test_bf16_decode_agrees_with_reference_mostly(golden_dir, mode):
    """bf16 tier against the reference's greedy ids over ALL 48 tokens.  Greedy ids cannot be bit-exact in
    bf16 (a random-init model has top-2 logit gaps of ~1e-2, and one flipped token changes everything after it), so the
    free-running agreement is REPORTED, and what is ASSERTED is the derived statement: under teacher forcing with the
    reference's own tokens, every step whose fp32 top-2 gap exceeds twice the tier's logit error bound -- the oracle's own
    bf16-autocast error at that step x K_LOGITS -- must pick the reference's token."""
    G, model, maps, conds, disc, z = setup(mode, "bf16", golden_dir)
    gen_len, mil = [int(x) for x in z[f"{mode}_noslide_cfg"]]
    ref_ids = z[f"{mode}_noslide_ids"]                       # [steps, 4]
    ids = run(G, model, maps, mode, conds, disc, gen_len, mil, use_cache=True)
    assert ids.shape == ref_ids.shape
    assert ((ids >= 2) & (ids < 1007))[1:].all()            # specials are never generated
    same = ids == ref_ids
    first_div = [int(np.argmin(same[:, b])) if not same[:, b].all() else same.shape[0] for b in range(same.shape[1])]
    print("bf16 greedy vs reference ids (%s): %d / %d tokens equal free-running, first divergence per sample %s" %
          (mode, int(same.sum()), same.size, first_div))
    assert (ids[:2] == ref_ids[:2]).all()
    if mode != "continuous_concat":
        return
    # teacher forcing through the cached decode: logits of both tiers and of the oracle (fp32 / bf16 autocast) per step
    from midiemo.decode import DecodeSession
    K_LOGITS = 1.25                                          # same constant as tests/test_model_gpu.py
    cfg = O.Cfg(1007, 2, 2, 64, 128, d_condition=16, conditioning=mode)
    P = O.seeded_params(cfg, int(z["weight_seed"]))
    cond = torch.tensor(conds, dtype=torch.float32)
    toks = torch.from_numpy(ref_ids.T.copy())                # [4, steps]
    lg32 = O.forward(cfg, P, toks, cond)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lgac = O.forward(cfg, P, toks, cond).float()
    sess = DecodeSession(model, toks.shape[0])
    agree = stable = 0
    for t in range(toks.shape[1] - 1):
        lg = sess.step(toks[:, t].cuda(), cond.cuda()).float().cpu()[:, 2:1007]     # generate() never emits the specials
        r32 = lg32[:, t, 2:1007]
        err_bound = K_LOGITS * (lgac[:, t, 2:1007] - r32).abs().max(-1).values      # per sample, this step
        top2 = r32.topk(2, dim=-1)
        gap = top2.values[:, 0] - top2.values[:, 1]
        pick, want = lg.argmax(-1), top2.indices[:, 0]
        assert (want + 2 == toks[:, t + 1]).all()                                     # the oracle's arg-max IS the reference's next token
        must = gap > 2 * err_bound
        assert (pick[must] == want[must]).all(), (t, pick, want, gap, err_bound)
        agree += int((pick == want).sum())
        stable += int(must.sum())
    n = (toks.shape[1] - 1) * toks.shape[0]
    print("bf16 teacher-forced decode: %d / %d steps pick the reference token; %d steps have a top-2 gap above twice the "
          "derived logit error bound and all of them agree" % (agree, n, stable))


def test_sampling_path_runs_and_respects_exclusions(golden_dir):
    G, model, maps, conds, disc, z = setup("continuous_concat", "bf16", golden_dir)
    torch.manual_seed(3)
    ids = run(G, model, maps, "continuous_concat", conds, disc, 40, 24, use_cache=True, top_k=-1)
    assert ids.shape == (40, 4) and (ids[0] == 1).all()
    assert ((ids[1:] >= 2) & (ids[1:] < 1007)).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cd", ["fp32", "bf16", "fp16"])
def test_device_resident_sampling_loop_equals_eager_loop(golden_dir, mode, cd):
    """generate() with sampling: while the KV cache is valid the loop runs on the device (DecodeSession.sample_run: one
    HIP graph per token with me_sample_step computing the temperature / repeat penalty, drawing from the uniforms
    generate() prepared, and updating the repeat counters).  Same seed -> the same token stream as one Python
    iteration per token, bit for bit, including the hand-over to the sliding-window (full recompute) part and the
    repeat penalty (a peaked model: long runs of one- and two-choice steps)."""
    G, model, maps, conds, disc, z = setup(mode, cd, golden_dir)
    with torch.no_grad():                                # peaked logits: repeat counters grow, the penalty matters
        model.fc.weight.mul_(6.0)
        model.mark_params_changed()

    def go(device_loop, gen_len, mil, top_k, top_p):
        torch.manual_seed(11)
        return G.generate(model, maps, torch.device("cuda"), "/tmp/none", mode, discrete_conditions=disc,
                          continuous_conditions=None if mode == "none" else conds, max_input_len=mil, amp=False,
                          gen_len=gen_len, top_k=top_k, top_p=top_p, temperatures=[1.3, 0.9], penalty_coeff=0.5, debug=True,
                          min_n_instruments=0, primers=[["<START>"]] * 4 if mode == "none" else [["<START>"]],
                          use_cache=True, return_ids=True, device_loop=device_loop).numpy()

    for gen_len, mil, top_k, top_p in [(60, 1024, -1, 0.7), (50, 30, 8, 1.0), (33, 1024, 3, 0.9)]:
        a = go(True, gen_len, mil, top_k, top_p)
        b = go(False, gen_len, mil, top_k, top_p)
        assert a.shape == b.shape == (gen_len, 4)
        assert np.array_equal(a, b), (mode, cd, gen_len, mil, np.argwhere(a != b)[:4])
        assert ((a[1:] >= 2) & (a[1:] < 1007)).all()


def test_device_resident_sampling_loop_more_than_eight_sequences(golden_dir):
    """B = 10 > the 8 rows a decode kernel call takes: the step runs in two row chunks, the sampling kernel and the
    commit over all rows; the device loop must still equal the per-token loop."""
    G, model, maps, conds, disc, z = setup("continuous_concat", "fp32", golden_dir)
    conds10 = [[((3 * i) % 7 - 3) / 4.0, ((5 * i) % 9 - 4) / 5.0] for i in range(10)]

    def go(device_loop):
        torch.manual_seed(21)
        return G.generate(model, maps, torch.device("cuda"), "/tmp/none", "continuous_concat", continuous_conditions=conds10,
                          max_input_len=1024, amp=False, gen_len=24, top_k=-1, top_p=0.8, debug=True, min_n_instruments=0,
                          use_cache=True, return_ids=True, device_loop=device_loop).numpy()

    a, b = go(True), go(False)
    assert a.shape == (24, 10) and np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("conditioning", ["none", "continuous_concat"])
def test_device_resident_greedy_loop_matches_eager_steps(conditioning):
    """DecodeSession.greedy_run (device-side position, HIP-graph replay) == step() + greedy_pick token by token."""
    import torch
    from midiemo import ops
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    torch.manual_seed(5)
    dc = 32 if conditioning == "continuous_concat" else -1
    model, _ = build_model(dict(vocab_size=1007, n_layer=2, n_head=2, d_model=128, d_inner=256, dropout=0.0, d_condition=dc,
                                conditioning=conditioning, compute_dtype="fp32"))
    model = model.cuda().eval()
    B, n = 3, 40
    cond = torch.rand(B, 2, device="cuda") * 2 - 1
    special = torch.tensor([0, 1, 5], dtype=torch.int32, device="cuda")
    tok0 = torch.tensor([1, 7, 300], device="cuda")
    with torch.no_grad():
        ref = DecodeSession(model, B)
        tok, picked, want = tok0.clone(), torch.empty(B, dtype=torch.long, device="cuda"), []
        for _ in range(n):
            lg = ref.step(tok, cond)
            ops.greedy_pick(lg, 1007, special, picked, B)
            tok = picked.clone()
            want.append(tok.clone())
        want = torch.stack(want, 1)
        for use_graph in (False, True):
            sess = DecodeSession(model, B)
            got = sess.greedy_run(tok0, n, cond, special, use_graph=use_graph)
            assert sess.t == n
            assert torch.equal(got, want), (use_graph, got[:, :8], want[:, :8])
        # continue from where the graph loop stopped: positions keep counting
        more = sess.greedy_run(got[:, -1], 5, cond, special)
        lg = ref.step(want[:, -1], cond)
        ops.greedy_pick(lg, 1007, special, picked, B)
        assert torch.equal(more[:, 0], picked)


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("conditioning", ["none", "discrete_token", "continuous_concat"])
def test_fused_decode_stages_equal_the_separate_launches(conditioning, cd):
    """The fused decode stages (me_dec_embed_qkv_attn for the first layer -- round 4 --, me_dec_ln_qkv_attn for the others)
    against the separate launches they replace (me_dec_embed_qkv / me_dec_qkv + me_dec_attn) over 70 positions (more than one
    key-split chunk), sequences at different tokens.  Same operands rounded at the same places; what differs is f32 summation
    order (the fused stage keeps the newest key as a split of its own and sums the projection per head slice), so the bound
    is f32 rounding in the f32 tier (1e-5 of the logit scale) and ONE rounding unit of the stored type in the 16-bit tiers
    (2^-8 / 2^-11 rel-L2: a k / v / hidden element may round to the other neighbour); the caches must agree likewise."""
    import torch
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    torch.manual_seed(11)
    V = 1017 if conditioning == "discrete_token" else 1007
    dc = 32 if conditioning == "continuous_concat" else -1
    model, _ = build_model(dict(vocab_size=V, n_layer=3, n_head=2, d_model=128, d_inner=256, dropout=0.0, d_condition=dc,
                                conditioning=conditioning, compute_dtype=cd))
    model = model.cuda().eval()
    B, n = 4, 70
    cond = torch.rand(B, 2, device="cuda") * 2 - 1
    toks = torch.randint(2, 1007, (n, B), device="cuda")
    tol = {"fp32": 1e-5, "bf16": 2.0 ** -8, "fp16": 2.0 ** -11}[cd]
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    worst = 0.0
    with torch.no_grad():
        a, b = DecodeSession(model, B), DecodeSession(model, B)
        assert a.fused and a.launches_per_token == 4 * 3 + 2
        b.fused = False                                   # the round-2 form: one launch per piece
        for i in range(n):
            la, lb = a.step(toks[i], cond).clone(), b.step(toks[i], cond).clone()
            worst = max(worst, rel(la, lb))
            assert rel(la, lb) <= tol, (i, rel(la, lb))
        for l in range(3):
            assert rel(a.kc[l][:, :, :n], b.kc[l][:, :, :n]) <= tol and rel(a.vc[l][:, :, :n], b.vc[l][:, :, :n]) <= tol, l
    print("fused vs separate decode stages, %s %s: worst logits rel-L2 over %d steps %.2e (bound %.1e)" % (conditioning, cd, n, worst, tol))


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("conditioning,geom", [("continuous_concat", (3, 2, 128, 256, 32)), ("none", (2, 4, 128, 512, -1)),
                                               ("discrete_token", (2, 8, 512, 2048, -1))])
def test_token_kernel_equals_the_launch_chain_bit_for_bit(conditioning, geom, cd, monkeypatch):
    """me_dec_token (one persistent launch per token: the stages exchange self-validating 8-byte records instead of ending a
    kernel; opt-in, MIDIEMO_DEC_TOKEN=1) against the per-stage launch chain it restates: SAME arithmetic, rounding points and
    summation orders (csrc/me_decode_common.h), so logits of every step and the K / V caches must be EQUAL, not close -- over
    positions that cross several key-split chunks, with sequences at different tokens, for head dims 64 / 32 / 64, B = 3 and 4
    (rows beyond the batch), and through the graph-replayed greedy loop.  The error word of the workspace stays clear."""
    import torch
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    n_layer, n_head, d, di, dc = geom
    torch.manual_seed(5)
    V = 1017 if conditioning == "discrete_token" else 1007
    model, _ = build_model(dict(vocab_size=V, n_layer=n_layer, n_head=n_head, d_model=d, d_inner=di, dropout=0.0, d_condition=dc,
                                conditioning=conditioning, compute_dtype=cd))
    model = model.cuda().eval()
    for B, n in ((4, 150), (3, 40)):
        cond = torch.rand(B, 2, device="cuda") * 2 - 1
        toks = torch.randint(2, 1007, (n, B), device="cuda")
        with torch.no_grad():
            monkeypatch.setenv("MIDIEMO_DEC_TOKEN", "0")
            a = DecodeSession(model, B)
            monkeypatch.setenv("MIDIEMO_DEC_TOKEN", "1")
            b = DecodeSession(model, B)
            assert not a.token_kernel and b.token_kernel and b.launches_per_token == 2
            if a.nsplit != b.nsplit:                          # the chain with the token kernel's (power of two) key split
                monkeypatch.setenv("MIDIEMO_DEC_NSPLIT", str(b.nsplit))
                monkeypatch.setenv("MIDIEMO_DEC_TOKEN", "0")
                a = DecodeSession(model, B)
                monkeypatch.delenv("MIDIEMO_DEC_NSPLIT")
            for i in range(n):
                la, lb = a.step(toks[i], cond).clone(), b.step(toks[i], cond).clone()
                assert torch.equal(la, lb), (B, i, float((la - lb).abs().max()))
            for l in range(n_layer):
                assert torch.equal(a.kc[l], b.kc[l]) and torch.equal(a.vc[l], b.vc[l]), (B, l)
            b.check_token_status()
            # graph-replayed greedy loop from a fresh position 0
            a.reset(); b.reset()
            ia = a.greedy_run(toks[0], 64, cond=cond).clone()
            ib = b.greedy_run(toks[0], 64, cond=cond).clone()
            assert torch.equal(ia, ib)
            b.check_token_status()


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["bf16", "fp32"])
def test_token_kernel_with_fewer_blocks_than_projection_tiles(cd, monkeypatch):
    """me_dec_token on HALF the chip (128 blocks, 4 key splits): the q | k | v, FFN_pre and FFN_suf stages then have more column groups
    (192 / 256 / 256) than blocks and every block walks several of them -- the path a smaller part would take.  Still equal to the chain."""
    import torch
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    torch.manual_seed(9)
    model, _ = build_model(dict(vocab_size=1007, n_layer=2, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
                                conditioning="continuous_concat", compute_dtype=cd))
    model = model.cuda().eval()
    B, n = 4, 90
    cond = torch.rand(B, 2, device="cuda") * 2 - 1
    toks = torch.randint(2, 1007, (n, B), device="cuda")
    monkeypatch.setenv("MIDIEMO_DEC_NSPLIT", "4")
    with torch.no_grad():
        monkeypatch.setenv("MIDIEMO_DEC_TOKEN", "0")
        a = DecodeSession(model, B)
        monkeypatch.setenv("MIDIEMO_DEC_TOKEN", "1")
        monkeypatch.setenv("MIDIEMO_DEC_TOKEN_BLOCKS", "128")
        b = DecodeSession(model, B)
        assert b.token_kernel and a.nsplit == b.nsplit == 4 and b._tok_blocks == 128
        for i in range(n):
            la, lb = a.step(toks[i], cond).clone(), b.step(toks[i], cond).clone()
            assert torch.equal(la, lb), (i, float((la - lb).abs().max()))
        for l in range(2):
            assert torch.equal(a.kc[l], b.kc[l]) and torch.equal(a.vc[l], b.vc[l]), l
        b.check_token_status()


@pytest.mark.gpu
def test_reference_goldens_and_long_contexts_under_the_token_kernel():
    """The decode tests that pin the build to the REFERENCE (greedy ids of its generate(): toy models in all four modes, the headline
    geometry at 128 / 512 / 256 tokens) and the long-context self-consistency tests (cache == full recompute at t = 2046, 2048-token
    graph loop == eager steps, sampled loops, more than eight sequences) once more in a child process with MIDIEMO_DEC_TOKEN=1: every
    eligible DecodeSession then runs me_dec_token instead of the launch chain."""
    import subprocess
    env = dict(os.environ, MIDIEMO_DEC_TOKEN="1")
    k = "greedy_ids_bit_exact or headline_geometry or config5 or device_resident or bf16_decode_agrees or sampling_path"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k", k, "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


@pytest.mark.gpu
@pytest.mark.parametrize("top_k,top_p,V", [(-1, 0.7, 1007), (20, 1.0, 1007), (50, 0.9, 1007), (-1, 1.0, 1007), (3, 0.5, 1007),
                                             (-1, 0.8, 1500), (40, 1.0, 2048), (-1, 0.9, 3000), (-1, 1.0, 4096)])
def test_fused_sampling_tail_matches_torch_path(top_k, top_p, V):
    """me_sample_topk_topp == the torch restatement of generate.py:122-189 (same filtered distribution, same
    n_choices; the draw is the inverse CDF of that distribution at the supplied uniform).  V > 1024 runs the 2048- / 4096-wide
    instantiations of the sort (the reference samples any vocabulary, generate.py:152-183)."""
    import torch
    import torch.nn.functional as F
    from midiemo import ops
    g = torch.Generator().manual_seed(11 + max(top_k, 0))
    B, NP = 6, 1024 if V <= 1024 else (2048 if V <= 2048 else 4096)
    ld = NP
    logits = torch.zeros(B, ld)
    logits[:, :V] = torch.randn(B, V, generator=g) * 3
    logits[0, 5] = float("nan")
    logits[1, :] = logits[1, :] * 0.01                    # nearly flat row: many choices
    logits[2, 100] = 40.0                                 # peaked row: one choice
    special = torch.tensor([0, 1, 7, V - 1], dtype=torch.int32)
    temp = torch.tensor([1.2, 0.8, 1.0, 2.5, 1.2, 0.5])
    u = torch.rand(B, generator=g)
    dev = "cuda"
    out = torch.empty(B, dtype=torch.long, device=dev)
    nch = torch.empty(B, dtype=torch.int32, device=dev)
    dp = torch.empty(B, NP, device=dev)
    di = torch.empty(B, NP, dtype=torch.int32, device=dev)
    ops.sample_topk_topp(logits.to(dev), V, special.to(dev), temp.to(dev), top_k, top_p, u.to(dev), out, nch, dp, di)
    # torch path (generate.py:122-189)
    o = logits[:, :V].clone()
    o[o != o] = 0
    o[:, special.long()] = -float("inf")
    o = F.log_softmax(o, dim=-1) / temp[:, None]
    k_eff = V if (top_k <= 0 or top_k > V) else top_k
    o, top_inds = torch.topk(o, k_eff)
    if 0 < top_p < 1:
        cum = torch.cumsum(F.softmax(o, dim=-1), dim=-1)
        remove = cum > top_p
        remove[:, 0] = False
        o[remove] = -float("inf")
    probs = F.softmax(o, dim=-1)
    dp, di, out, nch = dp.cpu(), di.cpu(), out.cpu(), nch.cpu()
    assert torch.allclose(dp[:, :k_eff], probs, atol=2e-6, rtol=1e-5), float((dp[:, :k_eff] - probs).abs().max())
    assert float(dp[:, k_eff:].abs().max()) == 0.0 if k_eff < NP else True
    sup = probs > 0
    dense_k = torch.zeros(B, V).scatter_add_(1, di[:, :NP].long().clamp(max=V - 1), dp * (di < V))
    dense_t = torch.zeros(B, V).scatter_add_(1, top_inds, probs)
    assert torch.allclose(dense_k, dense_t, atol=2e-6, rtol=1e-5)       # same distribution over the vocabulary
    assert torch.equal(nch.long(), sup.sum(-1))
    cdf = torch.cumsum(dp.double(), -1)
    for b in range(B):
        pos = int(torch.searchsorted(cdf[b], torch.tensor(float(u[b]), dtype=torch.double), right=True))
        pos = min(pos, int(sup[b].sum()) - 1)
        cand = {int(di[b, pos])}
        if pos + 1 < NP and dp[b, pos + 1] > 0:
            cand.add(int(di[b, pos + 1]))                # f32 vs f64 prefix sums may differ by one slot at a boundary
        if pos > 0:
            cand.add(int(di[b, pos - 1]))
        assert int(out[b]) in cand and dp[b, (di[b] == int(out[b])).nonzero()[0, 0]] > 0
    assert int(out[2]) == 100                             # the peaked row always picks its peak


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at its real size: headline model (6L d512 8H), 4 (valence, arousal) pairs, 2048 positions = max_seq
# ----------------------------------------------------------------------------------------------------------------------
def _cfg5_model(cd):
    from midiemo.models.build_model import build_model
    torch.manual_seed(0)
    model, _ = build_model(dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
                                conditioning="continuous_concat", dropout=0.1, compute_dtype=cd))
    return model.cuda().eval()


COND5 = [[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]]           # train.py:361-366


def test_config5_cache_equals_full_recompute_at_length():
    """Cache vs full recompute on the 6L d512 model at t in {0, 511, 1023, 2046, 2047} (the last two: the final key split
    is ragged and t = max_seq - 1 is the clamp boundary of the device-side position).  f32 tier: the cached logits equal
    the full forward's last row to rounding.  bf16 tier: both are compared with the f32 full forward -- the cached path
    may not be further from it than 1.5x the bf16 full forward is (its own error, measured in the same test)."""
    from midiemo.decode import DecodeSession
    g = torch.Generator().manual_seed(11)
    toks = torch.randint(2, 1007, (4, 2048), generator=g).cuda()
    cond = torch.tensor(COND5, device="cuda")
    probes = [0, 511, 1023, 2046, 2047]
    got, full = {}, {}
    for cd in ("fp32", "bf16", "fp16"):
        model = _cfg5_model(cd)
        with torch.no_grad():
            sess = DecodeSession(model, 4)
            for t in range(2048):
                lg = sess.step(toks[:, t], cond)
                if t in probes:
                    got[cd, t] = lg.clone()
            for t in probes:
                full[cd, t] = model(toks[:, :t + 1], cond)[:, -1].clone()
        del sess, model
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for t in probes:
        e32 = rel(got["fp32", t], full["fp32", t])
        e_dec, e_full = rel(got["bf16", t], full["fp32", t]), rel(full["bf16", t], full["fp32", t])
        print("config 5 t=%d: f32 cache vs full %.2e; bf16 cache vs f32 %.2e (bf16 full forward vs f32 %.2e)" % (t, e32, e_dec, e_full))
        assert e32 < 1e-4, (t, e32)
        assert e_dec <= 1.5 * e_full + 5e-4, (t, e_dec, e_full)
        # f16 tier (the reference's autocast dtype): cache decode and full forward both inside north_star's 1e-3 of the f32 logits
        h_dec, h_full = rel(got["fp16", t], full["fp32", t]), rel(full["fp16", t], full["fp32", t])
        print("config 5 t=%d: f16 cache vs f32 %.2e (f16 full forward vs f32 %.2e)" % (t, h_dec, h_full))
        assert h_dec < 1e-3 and h_full < 1e-3 and h_dec <= 1.5 * h_full + 1e-4, (t, h_dec, h_full)


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_config5_full_length_greedy_graph_equals_eager(cd):
    """2048-token device-resident greedy loop (HIP-graph replay, position in device memory) == eager step() +
    greedy_pick ids on the headline model: same kernels, same arithmetic -> bit-identical ids, all 2048 positions."""
    from midiemo import ops
    from midiemo.decode import DecodeSession
    from midiemo.vocab import get_maps, special_token_ids
    model = _cfg5_model(cd)
    cond = torch.tensor(COND5, device="cuda")
    special = torch.tensor(special_token_ids(get_maps()), dtype=torch.int32, device="cuda")
    tok0 = torch.full((4,), 1, dtype=torch.long, device="cuda")           # <START>
    with torch.no_grad():
        ref = DecodeSession(model, 4)
        tok, picked, want = tok0.clone(), torch.empty(4, dtype=torch.long, device="cuda"), []
        for _ in range(2048):
            lg = ref.step(tok, cond)
            ops.greedy_pick(lg, 1007, special, picked, 4)
            tok = picked.clone()
            want.append(tok)
        want = torch.stack(want, 1)
        sess = DecodeSession(model, 4)
        got = sess.greedy_run(tok0, 2048, cond, special, use_graph=True)
    assert sess.t == 2048 and got.shape == (4, 2048)
    assert torch.equal(got, want), int((got != want).sum())
    with pytest.raises(RuntimeError):
        sess.greedy_run(got[:, -1], 1, cond, special)                      # position 2048 does not exist


@pytest.mark.parametrize("tag", ["k0p07", "k20p10", "k50p09"])
def test_sampling_kernel_matches_reference_distribution(golden_dir, tag):
    """SURVEY 8f #1, device half: for the logits and temperatures of every step of a reference generate() run (fixture
    f7: captured with spies on torch.topk / torch.multinomial, oracle/make_host_fixtures.py) me_sample_topk_topp must
    produce the reference's filtered distribution -- same surviving ids, same probabilities, same number of choices."""
    from midiemo import ops
    z = np.load(os.path.join(golden_dir, "f7_host.npz"))
    top_k, top_p, steps, B, k_eff = z[f"samp_{tag}_cfg"]
    top_k, steps, B, k_eff = int(top_k), int(steps), int(B), int(k_eff)
    V = z[f"samp_{tag}_logits"].shape[-1]
    special = torch.from_numpy(z["special_ids"]).cuda()
    out = torch.empty(B, dtype=torch.long, device="cuda")
    nch = torch.empty(B, dtype=torch.int32, device="cuda")
    dp = torch.empty(B, 1024, device="cuda")
    di = torch.empty(B, 1024, dtype=torch.int32, device="cuda")
    g = torch.Generator().manual_seed(1)
    for s in range(steps):
        lg = torch.zeros(B, 1024)
        lg[:, :V] = torch.from_numpy(z[f"samp_{tag}_logits"][s])
        temp = torch.from_numpy(z[f"samp_{tag}_temp"][s]).float()
        u = torch.rand(B, generator=g)
        ops.sample_topk_topp(lg.cuda(), V, special, temp.cuda(), top_k, float(top_p), u.cuda(), out, nch, dp, di)
        probs = torch.from_numpy(z[f"samp_{tag}_probs"][s])                      # [B, k_eff], sorted descending
        inds = torch.from_numpy(z[f"samp_{tag}_inds"][s]).long()
        dense_ref = torch.zeros(B, V).scatter_add_(1, inds, probs)
        dpc, dic = dp.cpu(), di.cpu()
        dense_got = torch.zeros(B, V).scatter_add_(1, dic.long().clamp(max=V - 1), dpc * (dic < V))
        assert torch.allclose(dense_got, dense_ref, atol=3e-6, rtol=2e-4), (s, float((dense_got - dense_ref).abs().max()))
        assert torch.equal(nch.cpu().long(), (probs > 0).sum(-1)), s
        assert bool(((dense_ref.gather(1, out.cpu()[:, None]) > 0).all())), s     # the draw lands on a surviving id


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["continuous_concat", "continuous_token", "none"])
def test_window_forward_graph_equals_eager_forward(golden_dir, cd, mode):
    """WindowForward (the sliding-window regime of generate(): one captured HIP graph per token) returns exactly the last-
    position logits of the eager model(x, cond) call, for successive different windows of the same shape, after a shape
    change (re-capture), and after a parameter update (prepared weights refreshed outside the graph)."""
    from midiemo.decode import WindowForward
    G, model, maps, conds, disc, z = setup(mode, cd, golden_dir)
    model.eval()
    win = WindowForward(model)
    g = torch.Generator().manual_seed(21)
    cond = None if mode == "none" else torch.tensor(conds, dtype=torch.float32, device="cuda")
    B = 4
    for L in (40, 40, 40, 57, 40):
        x = torch.randint(2, 1007, (B, L), generator=g).cuda()
        with torch.no_grad():
            ref = model(x, cond)[:, -1, :].clone()
        got = win.last_logits(x, cond)
        assert torch.equal(got, ref), (cd, mode, L)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    model.mark_params_changed()
    x = torch.randint(2, 1007, (B, 40), generator=g).cuda()
    with torch.no_grad():
        ref = model(x, cond)[:, -1, :].clone()
    assert torch.equal(win.last_logits(x, cond), ref)
