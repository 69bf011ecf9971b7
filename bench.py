"""bench.py -- headline benchmark: MIDI tokens/s of one full training step
(forward + CE + backward + gradient all-reduce + global-norm clip + AdamW, dropout on)
of the emotion-conditioned Music Transformer on MI355X.

Workload (BASELINE.json configs[1]): continuous_concat, 6 layers, d_model 512,
8 heads, d_inner 2048, d_condition 128, V 1007, seq 1024, batch 32 per GPU, bf16 storage /
f32 accumulate, synthetic tokens in [2, V), weights random-init.  Weak scaling: 32 seq/GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = tokens / wall time of the K timed steps (barrier + synchronize on both
sides, max over ranks); `median_ms_per_step` = median of the per-step HIP-event durations inside that region.
`decode` (N = 1 only) = BASELINE config 5: KV-cached greedy decode, 4 (valence, arousal) pairs x 2048 tokens, tok/s,
p50 / p90 device step latency and the HBM roofline fraction of a step at context 1024.  `hbm_kernels` = GB/s of every
HBM-bound kernel of the train step from in-run HIP events.  `extra.config4` = discrete_token V1017 L2048 B16 train step.
`roofline` is the dominant kernel family
(the NT GEMM gemm_nt256_kernel<bf16>: every nn.Linear forward and dX product): algorithmic FLOPs of its launches /
their HIP-event durations INSIDE live train steps (events on the launch stream around every launch, in instrumented steps
after the timed region, minus the cost of an empty event pair measured at the same place); the committed rocprofv3 step trace
of the same sources is quoted beside it (`rocprof_step_trace`), and a warm back-to-back replay only as `warm_replay_*`.  `cpu_baseline` is the oracle (oracle/ref_model.py, a port) timed on the
host cores on a bounded sample (B = 2, same model / seq, dropout 0.1 on, 1 warm-up + >= 3 timed steps),
rank 0, N=1 only.  `extra.fp16_tier` = the same workload on the f16 tier (the reference's own autocast dtype, with its GradScaler
on the device: the 16-bit tier inside north_star's 1e-3 logits bound); `extra.fp32_tier` = the exact-f32 tier.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "midi-emotion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
           conditioning="continuous_concat", dropout=0.1)
SEQ, BATCH = 1024, 32
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def train_flop_per_token(c, L):
    """SURVEY 8d official figure: causal-discounted, 2*MAC, train = 3 x fwd."""
    d, di, V = c["d_model"], c["d_inner"], c["vocab_size"]
    per_layer = 8 * d * d + 4 * d * di + 3 * (2 * d * (L + 1) // 2)
    fwd = c["n_layer"] * per_layer + 2 * d * V
    return 3 * fwd


def synthetic_batch(c, B, L, seed, device):
    """tokens randint(2, V, (B, L+1)), input = tok[:, :-1], target = tok[:, 1:], cond U(-1,1)."""
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(2, c["vocab_size"], (B, L + 1), generator=g)
    cond = torch.rand(B, 2, generator=g) * 2 - 1
    return tok[:, :-1].contiguous().to(device), cond.to(device), tok[:, 1:].contiguous().to(device)


def cpu_baseline(c, L, budget_s=30.0):
    """The oracle's full train step on the host cores, as BASELINE.md section 3 plans it: forward + CE + autograd backward +
    global-norm clip + Adam, f32, dropout 0.1 ON, the synthetic batch of SURVEY 8d (seed 1234), B = 2 at L = 1024, one
    warm-up step and >= 3 timed steps (median), every host core this process may use.  `cores` = the thread count torch
    really ran with."""
    from oracle import ref_model as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # BASELINE.md section 3 plans "all host cores"; measured on the GPU box (256 hardware threads, 2 x EPYC 9575F): with 256
    # threads ONE step of this small-op-bound graph takes 218 s (9.4 tokens/s: gpurun_out/r5e), with 32 threads ~0.5 s --
    # so the pool is capped at 32 and the line says both numbers.
    torch.set_num_threads(max(1, min(avail, 32)))
    ncores = torch.get_num_threads()
    cfg = O.Cfg(c["vocab_size"], c["n_layer"], c["n_head"], c["d_model"], c["d_inner"],
                d_condition=c["d_condition"], conditioning=c["conditioning"])
    P = O.seeded_params(cfg, 0)
    M1 = {k: torch.zeros_like(v) for k, v in P.items()}
    M2 = {k: torch.zeros_like(v) for k, v in P.items()}
    B = 2 if L <= 1024 else 1
    inp, cond, tgt = O.synthetic_batch(cfg, B, L, seed=1234)
    torch.manual_seed(0)                                    # dropout masks
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        _, _, G = O.loss_and_grads(cfg, P, inp, cond, tgt, dropout=c["dropout"])
        O.adam_step(P, G, M1, M2, step + 1, lr=2e-5, clip=1.0)
        times.append(time.perf_counter() - t0)
        step += 1
        if step >= 6 or (step >= 4 and time.perf_counter() - t_start > budget_s):
            break
    timed = sorted(times[1:])                               # step 0 = warm-up
    med = timed[len(timed) // 2]
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(B * L / med, 1), "unit": "tokens/s", "cores": ncores, "cores_available": avail, "kind": "port", "cpu": cpu_model,
            "sample": "oracle train step (fwd+CE+bwd+clip+Adam, f32, dropout %.1f on), B=%d L=%d, 1 warm-up + %d timed steps, "
                      "median %.2f s (min %.2f s)" % (c["dropout"], B, L, len(timed), med, timed[0])}


def tier_bench(cd, B, L, steps=5, warmup=2):
    """The SAME C2 workload on another tier of the engine: "fp16" (the reference's own autocast dtype + its GradScaler on the
    device: the 16-bit tier that meets north_star's "logits within 1e-3 rel") or "fp32" (exact-f32 MFMA): tokens/s and ms per
    step of the full train step."""
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW, LossScaler
    torch.manual_seed(0)
    model, _ = build_model(dict(CFG, compute_dtype=cd))
    model = model.cuda().train()
    model.seed_dropout(1000)
    scaler = LossScaler("cuda") if cd == "fp16" else None
    opt = FusedAdamW(model, lr=2e-5, clip=1.0, scaler=scaler)
    ls = scaler.scale_tensor if scaler is not None else None
    batches = [synthetic_batch(CFG, B, L, 1234 + 7919 * i, "cuda") for i in range(4)]
    for i in range(warmup):
        model.loss_and_backward(*batches[i % 4], loss_scale=ls)
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = model.loss_and_backward(*batches[(warmup + i) % 4], loss_scale=ls)
        opt.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    res = {"workload": "the headline C2 workload (batch %d, seq %d, dropout 0.1, fwd+CE+bwd+clip+AdamW) on the %s tier" %
                       (B, L, {"fp16": "f16 (v_mfma_f32_32x32x16_f16, dynamic loss scale)", "fp32": "exact-f32 MFMA"}[cd]),
           "dtype": {"fp16": "f16", "fp32": "f32"}[cd],
           "tokens_per_s": round(B * L * steps / el, 1), "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "warmup": warmup,
           "final_loss": round(float(loss.item()), 4),
           "step_tflops_algorithmic": round(B * L * steps / el * train_flop_per_token(CFG, L) / 1e12, 2)}
    if cd == "fp32":
        res["peak_f32_mfma_tflops"] = 157.3
    if scaler is not None:
        res["loss_scaler"] = {"scale": scaler.get_scale(), "steps_taken": scaler.steps_taken(), "steps_skipped": scaler.steps_skipped(),
                              "semantics": "torch.cuda.amp.GradScaler defaults (65536, x2 / 2000 finite steps, x0.5 + skipped step on inf / nan), state on the device"}
    del opt, model
    torch.cuda.empty_cache()
    return res


def tier_parity_sample():
    """Logits rel-L2 of every tier against the oracle on a bounded sample (B = 1, L = 256 of the headline model, seeded weights,
    dropout off on both sides: seconds on the host) -- north_star's bound is 1e-3."""
    from oracle import ref_model as O
    from midiemo.models.build_model import build_model
    c = CFG
    cfg = O.Cfg(c["vocab_size"], c["n_layer"], c["n_head"], c["d_model"], c["d_inner"], d_condition=c["d_condition"],
                conditioning=c["conditioning"])
    P = O.seeded_params(cfg, 31)
    tok, cond, _ = O.synthetic_batch(cfg, 1, 256, seed=1234)
    with torch.no_grad():
        ref = O.forward(cfg, P, tok, cond).double()
        out = {}
        for cd in ("fp32", "bf16", "fp16"):
            m, _ = build_model(dict(CFG, dropout=0.0, compute_dtype=cd))
            m.load_state_dict(P)
            m = m.cuda().eval()
            lg = m(tok.cuda(), cond.cuda()).double().cpu()
            out[cd] = float((lg - ref).norm() / ref.norm())
            del m
    torch.cuda.empty_cache()
    return {"f32_tier": float("%.3e" % out["fp32"]), "bf16_tier": float("%.3e" % out["bf16"]), "f16_tier": float("%.3e" % out["fp16"]),
            "sample": "seeded weights (seed 31), B=1 L=256, dropout off", "north_star_bound": 1e-3}


def source_hash():
    """sha256 over the kernel sources: ties profiles/hbm_traffic.json (a PMC pass) to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "midi-emotion_amd", "csrc")
    for n in sorted(os.listdir(cs)):
        if n.endswith((".hip", ".h")):
            h.update(open(os.path.join(cs, n), "rb").read())
    return h.hexdigest()


def decode_bytes(c, t, B, elt=2):
    """algorithmic HBM bytes of one decode step at context length t (weights once + K/V/E rows read + k/v written)."""
    d, di, V, N, H = c["d_model"], c["d_inner"], c["vocab_size"], c["n_layer"], c["n_head"]
    dh = d // H
    weights = elt * (N * (3 * d * d + d * d + 2 * d * di) + V * d)
    return weights + N * t * dh * elt + B * N * 2 * t * d * elt + B * N * 2 * d * elt


def decode_bench(cd, gen_len=2048):
    """BASELINE config 5: KV-cached greedy decode (generate.py --topk 1), B = 4 conditions, gen_len tokens, one GPU."""
    from midiemo.decode import DecodeSession
    from midiemo.models.build_model import build_model
    from midiemo.vocab import get_maps, special_token_ids
    torch.manual_seed(0)
    model, _ = build_model(dict(CFG, compute_dtype=cd))
    model = model.cuda().eval()
    B = 4
    cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")      # train.py:361-366
    specials = torch.tensor(special_token_ids(get_maps()), dtype=torch.int32, device="cuda")
    tok0 = torch.full((B,), 1, dtype=torch.long, device="cuda")                                   # <START>
    sess = DecodeSession(model, B)
    with torch.no_grad():
        sess.greedy_run(tok0, 8, cond, specials)                 # graph capture + warm-up
        sess.reset()
        torch.cuda.synchronize()
        evs = []
        e0 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        ids = sess.greedy_run(tok0, gen_len, cond, specials, step_events=evs)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    lat = [a.elapsed_time(b) for a, b in zip([e0] + evs[:-1], evs)]      # ms per step on the device
    lat_sorted = sorted(lat)
    pct = lambda q: lat_sorted[min(len(lat_sorted) - 1, int(q * len(lat_sorted)))]
    mid = lat[gen_len // 2 - 32: gen_len // 2 + 32] if gen_len >= 128 else lat
    t_mid = gen_len // 2
    l_mid = sorted(mid)[len(mid) // 2] * 1e-3
    elt = 4 if cd == "fp32" else 2
    by = decode_bytes(CFG, t_mid, B, elt)
    n_launch = sess.launches_per_token
    del sess, model
    torch.cuda.empty_cache()
    return {"dtype": cd, "batch": B, "gen_len": gen_len, "tokens_per_s": round(B * gen_len / wall, 1),
            "step_ms_p50": round(pct(0.5), 4), "step_ms_p90": round(pct(0.9), 4), "step_ms_mean_wall": round(1e3 * wall / gen_len, 4),
            "launches_per_step": n_launch, "replay": "one HIP graph per token, position in device memory",
            "roofline": {"bound": "hbm", "context": t_mid, "bytes_per_step": by, "step_ms": round(l_mid * 1e3, 4),
                         "achieved": round(by / l_mid / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(by / l_mid / 1e9 / PEAK_HBM_GBS, 4)},
            "ids_checksum": int(ids.sum().item())}


def decode_slide_bench(cd="bf16", n_tok=48, window=1024):
    """The sliding-window regime of the reference's DEFAULT generate flags (generate.py:259-285: --max_input_len 1024,
    --gen_len 2048): past 1024 tokens the window moves, every remaining token shifts its absolute position
    (generate.py:101-103, music_multi.py:163), the K/V cache is invalid and each new token costs a full forward over the
    window -- what generate() of this build does there too (exact, SURVEY hard part 5).  Timed: n_tok tokens of that
    regime, B = 4, greedy pick, exactly the statements of generate()'s reference path."""
    from midiemo import ops
    from midiemo.models.build_model import build_model
    from midiemo.vocab import get_maps, special_token_ids
    torch.manual_seed(0)
    model, _ = build_model(dict(CFG, compute_dtype=cd))
    model = model.cuda().eval()
    B, V = 4, CFG["vocab_size"]
    cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
    specials = torch.tensor(special_token_ids(get_maps()), dtype=torch.int32, device="cuda")
    g = torch.Generator().manual_seed(5)
    song = torch.randint(2, V, (window + 8, B), generator=g).cuda()            # a window that is already full
    picked = torch.empty(B, dtype=torch.long, device="cuda")
    from midiemo.decode import WindowForward
    win = WindowForward(model)
    def one():
        nonlocal song
        inp = song[-window:]
        out = win.last_logits(inp.t().contiguous(), cond)                       # generate()'s sliding path: the window forward as one HIP graph
        ops.greedy_pick(out.contiguous(), V, specials, picked, B)
        song = torch.cat((song, picked.clone()[None, :]), 0)
    with torch.no_grad():
        for _ in range(4):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_tok):
            one()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    del model
    torch.cuda.empty_cache()
    return {"dtype": cd, "batch": B, "window": window, "tokens_timed": n_tok, "tokens_per_s": round(B * n_tok / wall, 1),
            "ms_per_token_step": round(1e3 * wall / n_tok, 3),
            "note": "full forward over the 1024-token window per new token (the reference's sliding window invalidates every cached position), replayed as one HIP graph"}


def decode_cpu_baseline(budget_s=12.0):
    """The oracle's restatement of the reference's decode loop (full-window recompute per token, no cache) on the host
    cores, on a bounded prefix: B = 4, as many tokens as fit the budget (the cost per token grows with the prefix)."""
    from oracle import ref_model as O
    c = CFG
    cfg = O.Cfg(c["vocab_size"], c["n_layer"], c["n_head"], c["d_model"], c["d_inner"], d_condition=c["d_condition"],
                conditioning=c["conditioning"])
    P = O.seeded_params(cfg, 0)
    conds = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]])
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for n_try in (8, 16, 32, 64):
            t1 = time.perf_counter()
            O.greedy_decode(cfg, P, conds, n_try, 2048)
            dt, n = time.perf_counter() - t1, n_try
            if time.perf_counter() - t0 + 2.5 * dt > budget_s:
                break
    return {"value": round(4 * n / dt, 2), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle greedy decode, full recompute per token (generate.py:99-119), f32, B=4, first %d tokens" % n}


class OpProbe:
    """HIP-event timing of the HBM-bound kernels of a train step (SURVEY 8d asks for their GB/s): every launch of the
    wrapped ops is bracketed by events on the launch stream; bytes = the algorithmic bytes of the call."""

    def __init__(self, ops, model, B, L):
        self.ops, self.rec, self.orig = ops, {}, {}
        d, V = model.embedding_dim, model.head_size
        T = B * L
        es = 4 if model.compute_dtype == torch.float32 else 2
        lo = es if (model.resid_lo and es == 2) else 0
        n = model.flat_params.numel()
        ldv = ((V + 63) // 64) * 64
        self.bytes = {
            "resid_ln_fwd": T * d * (es + lo + es + es + lo + es) + T * 8,        # x(+lo), a in; y(+lo), s out; stats
            "resid_ln_bwd": T * d * 4 * es + T * 8,                               # dy, s in; dx, da out
            "ce_fwd": T * ldv * es + T * 12,                 # logits in the compute type
            "ce_bwd": T * ldv * es + T * ldv * es + T * 12,
            "embed_fwd": T * d * (es + lo) + T * (d - model.d_condition) * 4 + T * d * 4,   # out(+lo); table rows; PE
            "embed_bwd": T * d * es,
            "sumsq": n * 4,
            "adamw_step": n * 4 * 7,                                               # p, g, m, v in; p, m, v (+ zeroed g) out
        }

    def __enter__(self):
        for name in self.bytes:
            orig = getattr(self.ops, name)
            self.orig[name] = orig

            def wrapped(*a, _o=orig, _n=name, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _o(*a, **k)
                e1.record()
                self.rec.setdefault(_n, []).append((e0, e1))
                return r
            setattr(self.ops, name, wrapped)
        return self

    def __exit__(self, *a):
        for name, orig in self.orig.items():
            setattr(self.ops, name, orig)

    def table(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.rec.items():
            us = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
            med = us[len(us) // 2]
            out[name] = {"launches": len(evs), "us": round(med, 2), "MB": round(self.bytes[name] / 1e6, 1),
                         "GBps": round(self.bytes[name] / med / 1e3, 1), "frac_of_8TBps": round(self.bytes[name] / med / 1e3 / PEAK_HBM_GBS, 3)}
        return out


class GemmProbe:
    """HIP-event timing of every NT-GEMM launch (me_gemm_nt and me_gemm_nt_relu_mask; events recorded on the launch stream).
    At the bench shapes all of them run the 256 x 256 tile kernel (gemm_nt256_kernel<T, ...>).  What an event pair adds to the
    kernel it brackets is calibrated by replay(): the same launches timed with per-launch pairs and with ONE pair around the whole
    back-to-back sequence; the difference per launch (~2 us: most of an event packet's latency hides under the kernel it follows;
    an EMPTY pair costs 6 us and over-corrects -- round 6 tried it: 56.6 us against rocprofv3's 60.7) is subtracted from the in-step figure."""

    def __init__(self, ops, steps=0, pool=1200):
        self.ops, self.rec, self.calls, self.steps = ops, [], [], steps
        self.orig = {"gemm_nt": ops.gemm_nt, "gemm_nt_relu_mask": ops.gemm_nt_relu_mask}
        # events are created BEFORE the instrumented steps: hipEventCreate inside the loop (three per launch) made the host the
        # bottleneck of those steps, and an e0 recorded on an idle stream then times the host's launch gap, not the kernel
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]

    def _ev(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        def make(name):
            fn = self.orig[name]

            def nt(A, B, C, *rest, **kw):
                m = A.shape[0] if kw.get("M") is None else kw["M"]
                k = A.shape[1] if kw.get("K") is None else kw["K"]
                n = B.shape[0] if kw.get("N") is None else kw["N"]
                e0, e1 = self._ev(), self._ev()
                e0.record()
                fn(A, B, C, *rest, **kw)
                e1.record()
                self.rec.append((2.0 * m * n * k, e0, e1, (m * k + n * k) * A.element_size() + m * n * C.element_size()))
                self.calls.append((fn, (A, B, C) + tuple(rest), dict(kw)))
            return nt
        for name in self.orig:
            setattr(self.ops, name, make(name))
        return self

    def __exit__(self, *a):
        for name, fn in self.orig.items():
            setattr(self.ops, name, fn)

    def summary(self):
        """(FLOPs, ms, launches) of ONE step: the first instrumented step is dropped (the stream restarts after the timed region's
        synchronisation), every launch position of the step takes the MEDIAN of its durations over the remaining steps."""
        torch.cuda.synchronize()
        per = len(self.rec) // max(1, self.steps)
        keep = range(1, self.steps) if self.steps > 1 else range(self.steps)
        flops = sum(r[0] for r in self.rec[:per])
        ms = 0.0
        for j in range(per):
            d = sorted(self.rec[st * per + j][1].elapsed_time(self.rec[st * per + j][2]) for st in keep)
            ms += d[len(d) // 2]
        self.alg_bytes = sum(r[3] for r in self.rec[:per]) / max(1, per)
        return flops, ms, per

    def replay(self, reps=24, rounds=3):
        """Duration of every distinct NT call of ONE step from a back-to-back replay: `reps` launches of the call (same
        operands, same write-out path) between TWO events, best of `rounds` -- per-launch event pairs add 2-6 us of event
        overhead to every launch.  A WARM number: after the first launch the operands sit in L2 / Infinity Cache, which the live
        step's activation operands do not -- reported as `warm_replay_*` only, never as roofline.achieved.
        Returns (sum of per-call average ms over one step's calls, calls)."""
        calls = self.calls[:len(self.calls) // max(1, self.steps)] if self.steps else self.calls
        total, over = 0.0, []
        for (fn, a, kw) in calls:
            best = None
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn(*a, **kw)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / reps
                best = t if best is None else min(best, t)
            total += best
            # the same launches with one event pair EACH (what the in-step probe does): the excess per launch is the pair's cost
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in evs:
                e0.record()
                fn(*a, **kw)
                e1.record()
            torch.cuda.synchronize()
            pair = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[reps // 2]
            over.append(pair - best)
        self.pair_overhead_ms = max(0.0, sorted(over)[len(over) // 2])            # median over the step's calls
        return total, len(calls)


def config4_bench(steps=8, warmup=3):
    """BASELINE config 4 as a sub-measurement: discrete_token (V = 1017), same 6L d512 model, L = 2048, B = 16, bf16,
    full train step; synthetic tokens with the two emotion-bin tokens in front."""
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW
    c = dict(CFG, vocab_size=1017, conditioning="discrete_token", d_condition=-1)
    torch.manual_seed(0)
    model, _ = build_model(dict(c, compute_dtype="bf16"))
    model = model.cuda().train()
    opt = FusedAdamW(model, lr=2e-5, clip=1.0)
    B, L = 16, 2048
    g = torch.Generator().manual_seed(4321)
    tok = torch.randint(2, 1007, (B, L + 1), generator=g)
    tok[:, 0] = torch.randint(1007, 1012, (B,), generator=g)
    tok[:, 1] = torch.randint(1012, 1017, (B,), generator=g)
    x, y = tok[:, :-1].contiguous().cuda(), tok[:, 1:].contiguous().cuda()
    for _ in range(warmup):
        model.loss_and_backward(x, None, y)
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = model.loss_and_backward(x, None, y)
        opt.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d, di, V = c["d_model"], c["d_inner"], 1017
    fpt = 3 * (c["n_layer"] * (8 * d * d + 4 * d * di + 3 * (2 * d * (L + 1) // 2)) + 2 * d * V)
    tps = B * L * steps / el
    res = {"workload": "discrete_token V1017 6L d512 8H, seq 2048, batch 16, bf16, fwd+CE+bwd+clip+AdamW, dropout 0.1",
           "tokens_per_s": round(tps, 1), "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
           "step_tflops_algorithmic": round(tps * fpt / 1e12, 2), "final_loss": round(float(loss.item()), 4)}
    del model, opt
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 timed steps after >= 10 warm-up (the driver passes its own)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH, help="sequences per GPU (weak scaling)")
    ap.add_argument("--seq", type=int, default=SEQ)
    ap.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_probe", action="store_true")
    ap.add_argument("--no_decode", action="store_true")
    ap.add_argument("--no_extra", action="store_true")
    ap.add_argument("--only_config4", action="store_true", help="run only the config-4 sub-measurement (profiling passes)")
    args = ap.parse_args()
    if args.only_config4:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        torch.cuda.set_device(0)
        print(json.dumps({"extra": {"config4": config4_bench()}}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hooks (single-GPU boxes): MIDIEMO_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # MIDIEMO_DIST_BACKEND=gloo replaces RCCL, so the multi-rank control flow can be exercised without N GPUs
    if os.environ.get("MIDIEMO_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("MIDIEMO_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MIDIEMO_BENCH_FORCE_DIST=1 (test hook): take the distributed code path (process group, bucket all-reduce hooks,
    # barrier, MAX over ranks) even with one rank, so RCCL itself is exercised on a single-GPU box
    dist_on = world > 1 or bool(os.environ.get("MIDIEMO_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from midiemo import ops
    from midiemo.ddp import GradAllReducer, broadcast_params
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW, LossScaler

    torch.manual_seed(0)                                   # identical random-init weights on every rank
    margs = dict(CFG, compute_dtype=args.compute_dtype)
    model, _ = build_model(margs)
    model = model.to(dev).train()
    broadcast_params(model.flat_params)
    model.seed_dropout(1000 + rank)
    scaler = LossScaler(dev) if args.compute_dtype == "fp16" else None      # GradScaler of the reference's fp16 path (train.py:108)
    opt = FusedAdamW(model, lr=2e-5, clip=1.0, scaler=scaler)             # == Adam(lr) + clip_grad_norm_(1.0), train.py:182,321
    reducer = GradAllReducer(lambda: model.flat_grads, model.bucket_ranges())
    B, L = args.batch, args.seq
    batches = [synthetic_batch(CFG, B, L, 1234 + rank + 7919 * i, dev) for i in range(4)]

    def step(i):
        tok, cond, tgt = batches[i % len(batches)]
        loss = model.loss_and_backward(tok, cond, tgt, bucket_hook=reducer.hook if dist_on else None,
                                       loss_scale=scaler.scale_tensor if scaler is not None else None)
        reducer.finish()
        opt.step(grad_scale=reducer.grad_scale)
        return loss

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    # MIDIEMO_DDP_POLICY=auto: its probe phases (2 x AUTO_PROBE steps, one host sync at the decision) must not fall into the timed
    # region whatever --warmup says: keep stepping, untimed, until the reducer has decided
    extra_warm = 0
    while dist_on and getattr(reducer, "_auto", None) is not None and extra_warm < 64:
        step(args.warmup + extra_warm)
        extra_warm += 1
    fence()
    reducer.timing = dist_on               # two event records per step around the bucket waits (exposed communication)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    e0.record()
    loss = None
    for i in range(args.steps):
        loss = step(args.warmup + i)
        step_ev[i].record()                    # one timing event per step (no sync): median of the per-step device times
    e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    per_step = sorted(a.elapsed_time(b) for a, b in zip([e0] + step_ev[:-1], step_ev))
    median_ms = per_step[len(per_step) // 2]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.item())
    exposed = sorted(reducer.exposed_ms()) if dist_on else []
    reducer.timing = False
    if dist_on and exposed:
        # slowest rank's view: MAX over ranks of the mean exposed wait
        tw = torch.tensor([sum(exposed) / len(exposed), exposed[len(exposed) // 2]], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        exposed_mean, exposed_p50 = float(tw[0]), float(tw[1])

    ddp_state = (reducer.policy, reducer.decision)
    probe, hbm_table, gp, op_probe = None, None, None, None
    if not args.no_probe:
        # every rank runs the instrumented steps (they contain the gradient all-reduce); only rank 0 times its GEMMs
        if rank == 0:
            NPROBE = 7
            with GemmProbe(ops, steps=NPROBE) as gp:
                torch.cuda.synchronize()
                t_i = time.perf_counter()
                for i in range(NPROBE):
                    step(i)
                probe = gp.summary()
                gp.instrumented_step_ms = 1e3 * (time.perf_counter() - t_i) / NPROBE
            replay_ms, replay_calls = gp.replay()          # the workspace buffers of the last step are still alive
            with OpProbe(ops, model, B, L) as op_probe:
                for i in range(3):
                    step(i)
                hbm_table = op_probe.table()
        else:
            for i in range(7):
                step(i)
            for i in range(3):
                step(i)
    fence()

    if rank == 0:
        tokens = world * B * L * args.steps
        tps = tokens / elapsed
        fpt = train_flop_per_token(CFG, L)
        out = {
            "metric": "MIDI tokens/sec training (B32 seq1024 d512 6L)", "value": round(tps, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "median_ms_per_step": round(median_ms, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.compute_dtype, "data": "synthetic",
            "config": {"workload": "continuous_concat 6L d512 8H d_inner2048 d_cond128 V1007, seq %d, batch %d/GPU, "
                                   "fwd+CE+bwd+clip+AdamW, dropout 0.1" % (L, B),
                       "global_batch": world * B, "seq_len": L, "parallelism": "dp%d" % world,
                       "final_loss": round(final_loss, 4), "gpu_event_ms_per_step": round(e0.elapsed_time(e1) / args.steps, 3)},
            "step_tflops_algorithmic": round(tps * fpt / 1e12 / world, 2),
        }
        if probe is not None:
            flops, ms, n = probe
            # HBM bytes/launch of the same kernel: PMC counters need their own rocprofv3 pass (guide section
            # "HBM traffic"), so the committed summary of that pass over this exact command is quoted here --
            # only for the workload it was collected on.
            traffic, traffic_note = None, "no PMC pass committed for this workload"
            tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if args.compute_dtype == "bf16" and (B, L) == (32, 1024) and os.path.exists(tj):
                tjd = json.load(open(tj))
                # launch-weighted mean over the family's instantiations
                ks = [tjd[n_] for n_ in tjd if n_.startswith("gemm_nt256_kernel")]
                if tjd.get("_source_sha256") != source_hash():
                    traffic_note = "profiles/hbm_traffic.json was measured on other kernel sources (hash mismatch): not quoted"
                elif ks:
                    nl = sum(k["launches"] for k in ks)
                    traffic = int(round(sum((k["read_MB"] + k["write_MB"]) * k["launches"] for k in ks) / nl * 1e6))
                    traffic_note = "HBM bytes per launch, mean over the family's %d traced launches (rocprofv3 PMC passes FETCH_SIZE x2 + WRITE_SIZE, %s)" % (nl, tjd.get("_profile", "profiles/"))
            # achieved = algorithmic FLOPs of the launches of 3 LIVE steps / (their HIP-event durations - the calibrated cost of the
            # event pair around each of them): the kernel as the timed region runs it (cold activation operands)
            peak = PEAK_BF16_TFLOPS if args.compute_dtype != "fp32" else 157.3
            ms_live = ms - n * gp.pair_overhead_ms               # one step: `n` launches, medians over the instrumented steps
            ach = flops / (ms_live * 1e-3) / 1e12
            kname = {"bf16": "gemm_nt256_kernel<bf16>", "fp16": "gemm_nt256_kernel<f16>", "fp32": "gemm_nt_kernel<float>"}[args.compute_dtype]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_unit": traffic_note,
                               "algorithmic_bytes_per_launch": int(gp.alg_bytes), "launches_per_step": n,
                               "avg_launch_us": round(1000.0 * ms_live / n, 2),
                               "gemm_nt_ms_per_step": round(ms_live, 3),
                               "timing": "HIP events on the launch stream around every NT launch of %d live train steps after one discarded (%d launches per "
                                         "step, median per launch position), minus the calibrated cost of an event pair around a launch (%.2f us: per-launch "
                                         "pairs vs one pair around the same back-to-back launches)" % (gp.steps - 1, n, 1000.0 * gp.pair_overhead_ms),
                               "raw_event_avg_launch_us": round(1000.0 * ms / n, 2),
                               "instrumented_step_ms": round(gp.instrumented_step_ms, 3),     # must stay device-bound: ~ ms_per_step + the event packets
                               "warm_replay_avg_launch_us": round(1000.0 * replay_ms / replay_calls, 2),
                               "warm_replay_frac": round(flops / (replay_ms * 1e-3) / 1e12 / peak, 4),
                               "warm_replay_note": "24 back-to-back launches of each call on L2 / Infinity-Cache-warm operands, best of 3: an upper bound, not the step",
                               "step_frac_of_peak": round(tps * fpt / 1e12 / world / PEAK_BF16_TFLOPS, 4)}
            # the committed rocprofv3 kernel trace of the train steps alone (tools/round_profiles.sh -> profiles/step_trace.json),
            # quoted only for the sources it was measured on: the profile's average launch duration of the same family
            sj = os.path.join(ROOT, "profiles", "step_trace.json")
            if args.compute_dtype == "bf16" and (B, L) == (32, 1024) and os.path.exists(sj):
                sjd = json.load(open(sj))
                if sjd.get("_source_sha256") == source_hash() and "gemm_nt" in sjd.get("families", {}):
                    fam = sjd["families"]["gemm_nt"]
                    out["roofline"]["rocprof_step_trace"] = {
                        "avg_launch_us": fam["avg_us"], "ms_per_step": fam["ms_per_step"], "launches_per_step": fam["launches_per_step"],
                        "frac": round(flops / (fam["ms_per_step"] * 1e-3) / 1e12 / peak, 4), "profile": sjd.get("_profile")}
                else:
                    out["roofline"]["rocprof_step_trace"] = "profiles/step_trace.json was measured on other kernel sources (hash mismatch): not quoted"
        # multi-GPU knobs of this run (SCALE runs are only interpretable with them): the overlap policy of the bucket
        # all-reduces (midiemo/ddp.py) and the CUs the persistent GEMM grids leave to RCCL (default 0: a reserve makes
        # EVERY 256-tile launch take a second tile round -- measured on one GPU: qkv 59.6 -> 70.8 us, proj 27.1 -> 43.9 us
        # with 64 CUs reserved -- so none is applied automatically)
        out["ddp"] = {"policy": os.environ.get("MIDIEMO_DDP_POLICY", "window"), "policy_in_force": ddp_state[0],
                      "compress": "bf16" if reducer is not None and reducer.compress else None,       # MIDIEMO_DDP_COMPRESS=bf16: 41.2 MB instead of 82.4 MB per step
                      "policy_decision": ddp_state[1],          # "auto": the two measured spans and the choice (midiemo/ddp.py)
                      "cu_reserve": int(os.environ.get("MIDIEMO_CU_RESERVE", "0") or 0), "world": world,
                      "backend": backend if dist_on else None}
        if dist_on and exposed:
            # time per step the compute stream sat in GradAllReducer.finish() waiting for bucket all-reduces (HIP events
            # around the waits, max over ranks): communication NOT hidden behind the backward.  The first thing to read in
            # a SCALE run: value(N) / (N value(1)) ~ 1 - exposed / step when nothing else changes.
            out["ddp"].update(exposed_wait_ms_per_step=round(exposed_mean, 4), exposed_wait_ms_p50=round(exposed_p50, 4),
                              exposed_frac_of_step=round(exposed_mean / (1000.0 * elapsed / args.steps), 4),
                              grad_bytes_per_step=int(model.flat_grads.numel() * (2 if reducer.compress else 4)))
        if hbm_table is not None:
            out["hbm_kernels"] = hbm_table
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(CFG, L)
        if world == 1 and not args.no_extra:
            out["extra"] = {"config4": config4_bench()}
            if args.compute_dtype == "bf16":
                model = opt = reducer = gp = op_probe = None        # (the probes hold the step's workspace tensors)
                torch.cuda.empty_cache()
                # the 16-bit tier inside north_star's tolerance, with the headline run's own step count, beside the headline number
                out["extra"]["fp16_tier"] = tier_bench("fp16", B, L, steps=args.steps, warmup=args.warmup)
                out["extra"]["fp16_tier"]["ms_per_step_vs_bf16"] = round(out["extra"]["fp16_tier"]["ms_per_step"] / (1000.0 * elapsed / args.steps), 4)
                out["extra"]["fp32_tier"] = tier_bench("fp32", B, L)
                par = tier_parity_sample()
                out["extra"]["logits_rel_l2_vs_oracle"] = par
                out["extra"]["fp16_tier"]["logits_rel_l2_vs_oracle"] = par["f16_tier"]
                out["extra"]["fp32_tier"]["logits_rel_l2_vs_oracle"] = par
        if world == 1 and not args.no_decode:
            model = opt = None
            torch.cuda.empty_cache()
            dec = decode_bench("bf16")
            dec["fp32"] = {k: v for k, v in decode_bench("fp32").items() if k in ("tokens_per_s", "step_ms_p50", "step_ms_p90", "roofline")}
            dec["fp16"] = {k: v for k, v in decode_bench("fp16").items() if k in ("tokens_per_s", "step_ms_p50", "step_ms_p90", "roofline", "ids_checksum")}
            if not args.no_cpu_baseline:
                dec["cpu_baseline"] = decode_cpu_baseline()
            dec["slide"] = decode_slide_bench("bf16")
            # the opt-in per-token persistent kernel (me_dec_token, MIDIEMO_DEC_TOKEN=1): same token stream, measured beside the default
            prev = os.environ.get("MIDIEMO_DEC_TOKEN")
            os.environ["MIDIEMO_DEC_TOKEN"] = "1"
            try:
                tk = decode_bench("bf16")
                dec["token_kernel"] = {k: tk[k] for k in ("tokens_per_s", "step_ms_p50", "step_ms_p90", "launches_per_step", "ids_checksum")}
                dec["token_kernel"]["note"] = "one persistent launch per token (stages exchange tagged records); bit-identical to the launch chain, opt-in"
            except Exception as e:                               # an extra: its failure must not take the bench line with it
                dec["token_kernel"] = {"error": repr(e)[:300]}
            finally:
                if prev is None:
                    os.environ.pop("MIDIEMO_DEC_TOKEN", None)
                else:
                    os.environ["MIDIEMO_DEC_TOKEN"] = prev
            out["decode"] = dec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
