"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol declared in
include/midiemo.h, host-side module logic (state_dict ABI, flat packing, factory), and the
N>1 gradient-exchange path on 2 gloo processes."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_model as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from midiemo import _lib
    from midiemo.build import build
    build()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "midiemo.h")).read()
    declared = set(re.findall(r"\b(?:int|size_t)\s+(me_\w+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.me_abi_version() == _lib.ABI_VERSION
    # prototypes: same number of parameters in the header and in the binding
    for name in declared:
        m = re.search(r"\b(?:int|size_t)\s+%s\s*\(([^;]*?)\)\s*;" % name, hdr, re.S)
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(_lib.SIGNATURES[name]), name


def test_workspace_sizes_are_host_side_arithmetic():
    """me_workspace_bytes is pure host code (no device needed): the caller-owned buffers of the entry points added in round 5.
    ReLU sign mask: 1 bit per element, rows rounded up to 256; 0 = the shape / dtype keeps the gate operand."""
    from midiemo import _lib
    lib = _lib.load()
    ws = lambda op, M, N, K, dt: int(lib.me_workspace_bytes(op, M, N, K, dt))
    assert ws(_lib.ME_WS_RELU_MASK, 32768, 2048, 512, _lib.ME_BF16) == 32768 * 2048 // 8
    assert ws(_lib.ME_WS_RELU_MASK, 777, 1024, 512, _lib.ME_BF16) == 1024 * 1024 // 8        # rows -> 1024
    assert ws(_lib.ME_WS_RELU_MASK, 32768, 1007, 512, _lib.ME_BF16) == 0                     # N % 64
    assert ws(_lib.ME_WS_RELU_MASK, 32768, 2048, 512, _lib.ME_F32) == 0                      # exact tier keeps the activations
    assert ws(_lib.ME_WS_RELU_MASK, 128, 2048, 512, _lib.ME_BF16) == 0                       # below the 256-tile kernel
    # attention tiles: packed causal triangle of 32 x 32 tiles per (batch, head)
    nq = 1024 // 32
    assert ws(_lib.ME_WS_RGA_PT, 256, 1024, 1, _lib.ME_BF16) == 256 * (nq * (nq + 1) // 2) * 1024 * 2
    assert ws(_lib.ME_WS_RGA_DGT, 256, 1024, 0, _lib.ME_BF16) == 256 * (nq * (nq + 1) // 2) * 1024 * 2
    # me_dec_token (round 6): control block + 8-byte records of 4 rows: s2, s1, att [d], qkv [3d], hid [d_inner], partials
    assert ws(_lib.ME_WS_DEC_TOKEN, 4, 2048, 512, _lib.ME_BF16) == 256 + 8 * (4 * 512 * 6 + 4 * 2048 + 4 * 8 * (512 + 32 + 2))
    assert ws(_lib.ME_WS_DEC_TOKEN, 5, 2048, 512, _lib.ME_BF16) == 0                         # at most ME_DEC_TOKEN_ROWS sequences


@pytest.mark.parametrize("mode", ["none", "discrete_token", "continuous_token", "continuous_concat"])
def test_build_model_contract(mode):
    from midiemo.models.build_model import build_model
    V = 1017 if mode == "discrete_token" else 1007
    args = dict(vocab_size=V, n_layer=2, n_head=2, d_model=64, d_inner=128, dropout=0.1,
                d_condition=16 if mode == "continuous_concat" else -1, conditioning=mode)
    model, ret = build_model(args)
    assert ret is args and ret["regression"] is False                  # build_model.py:26-27 mutates/returns args
    cfg = O.Cfg(V, 2, 2, 64, 128, d_condition=16, conditioning=mode)
    shapes = O.param_shapes(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())                      # checkpoint ABI
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert model.max_seq == 2048 and model.pad_token == 0              # build_model.py:22-23
    # load_config_dict path (generate.py:341) + overwrite_dropout
    cfgd = dict(args, overwrite_dropout=True, dropout=0.25)
    m2, _ = build_model(None, load_config_dict=cfgd)
    assert m2.dropout_p == 0.25
    # weights round-trip through load_state_dict into the flat buffer
    P = O.seeded_params(cfg, 3)
    model.load_state_dict(P)
    for k, v in model.state_dict().items():
        assert torch.equal(v, P[k])
    lo, hi = model.bucket_ranges()[0][0], model.bucket_ranges()[-1][1]
    assert lo == 0 and hi == model.flat_params.numel()
    o, n, shp = model._slices["fc.weight"]
    assert torch.equal(model.flat_params[o:o + n].view(shp), P["fc.weight"])
    # Wq|Wk|Wv adjacency = fused [3d, d] projection matrix
    o, _, _ = model._slices["enc_layers.1.rga.Wq.weight"]
    fused = model.flat_params[o:o + 3 * 64 * 64].view(192, 64)
    assert torch.equal(fused[64:128], P["enc_layers.1.rga.Wk.weight"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.randint(2, 90, (1, 8)), None)


def test_reference_init_distributions():
    from midiemo.models.build_model import build_model
    torch.manual_seed(0)
    args = dict(vocab_size=1007, n_layer=1, n_head=8, d_model=512, d_inner=2048, dropout=0.1, d_condition=128,
                conditioning="continuous_concat")
    m, _ = build_model(args)
    sd = m.state_dict()
    assert sd["embedding.weight"].shape == (1007, 384)
    assert float(sd["embedding.weight"].abs().max()) <= 0.1 and float(sd["fc.bias"].abs().max()) == 0.0
    assert abs(float(sd["enc_layers.0.rga.E"].std()) - 1.0) < 0.02                     # randn (music_multi.py:185)
    assert float(sd["enc_layers.0.FFN_pre.weight"].abs().max()) <= 1 / np.sqrt(512) + 1e-6
    assert sum(p.numel() for p in m.parameters()) == 386688 + 384 + 3283456 + 515584 + 1007


def test_unsupported_configs_raise():
    from midiemo.models.build_model import build_model
    m, _ = build_model(dict(vocab_size=1007, n_layer=1, n_head=16, d_model=768, d_inner=3072, dropout=0.1,
                            d_condition=-1, conditioning="none"))       # dh = 48 (published checkpoints): supported
    assert m.dh == 48
    with pytest.raises(ValueError, match="head dim"):
        build_model(dict(vocab_size=1007, n_layer=1, n_head=16, d_model=640, d_inner=1024, dropout=0.1,
                         d_condition=-1, conditioning="none"))          # dh = 40: no kernel instantiation
    # regression=True builds the evaluation model: reference state_dict keys; like every model here it needs a HIP device
    r, _ = build_model(dict(vocab_size=1008, n_layer=1, n_head=8, d_model=512, d_inner=2048, dropout=0.1,
                            d_condition=-1, conditioning="none", regression=True))
    assert type(r).__name__ == "MusicRegression" and r.head_size == 2 and not r.causal
    keys = set(r.state_dict().keys())
    assert {"fc.0.weight", "fc.0.bias", "embedding.weight", "enc_layers.0.rga.E"} <= keys and "fc.weight" not in keys
    with pytest.raises(RuntimeError, match="HIP device"):
        r.loss_and_backward(torch.ones(1, 8, dtype=torch.long), torch.zeros(1, 2))


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "midi-emotion_amd"))
from midiemo.ddp import GradAllReducer, broadcast_params
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
ranges = [(0, 40), (40, 1000), (1000, 1960), (1960, 2000)]
torch.manual_seed(100 + rank)
flat = torch.randn(2000)
params = torch.randn(2000)
broadcast_params(params)
ref = [torch.empty(2000) for _ in range(world)]
dist.all_gather(ref, flat.clone())
base = flat.clone()
for policy in ("window", "eager", "end"):
    flat.copy_(base)
    red = GradAllReducer(lambda: flat, ranges, policy=policy)
    assert red.policy == policy
    red.hook(3)                   # backward order: head, [window, layer] x 2, embedding
    for b in (2, 1):
        red.hook(-1)              # comm window: parked buckets are launched here under "window"
        if policy == "window":
            assert not red._pending
        red.hook(b)
    red.hook(0)
    n_before = len(red._works)
    assert n_before == {"window": 2, "eager": 4, "end": 0}[policy], (policy, n_before)
    red.finish()
    assert torch.allclose(flat, sum(ref), atol=1e-6), policy
    assert abs(red.grad_scale - 1.0 / world) < 1e-12
# "auto": AUTO_PROBE steps under "end", AUTO_PROBE under "window", then ONE decision every rank agrees on; every step of
# the probe phases and after still produces the exact sum
flat.copy_(base)
red = GradAllReducer(lambda: flat, ranges, policy="auto")
seen = []
for s_ in range(2 * GradAllReducer.AUTO_PROBE + 2):
    flat.copy_(base)
    seen.append(red.policy)
    red.hook(3)
    for b in (2, 1):
        red.hook(-1)
        red.hook(b)
    red.hook(0)
    red.finish()
    assert torch.allclose(flat, sum(ref), atol=1e-6), ("auto", s_)
n = GradAllReducer.AUTO_PROBE
assert seen[:2 * n] == ["end"] * n + ["window"] * n, seen
assert red.decision is not None and red.decision["chosen"] in ("end", "window") and seen[-1] == red.decision["chosen"]
chosen = [None] * world
dist.all_gather_object(chosen, red.decision["chosen"])
assert len(set(chosen)) == 1, chosen
p0 = [torch.empty(2000) for _ in range(world)]
dist.all_gather(p0, params)
assert all(torch.equal(p0[0], q) for q in p0)
try:
    red.hook(1); red.hook(1)
    raise SystemExit("double reduce not detected")
except RuntimeError:
    pass
print("rank", rank, "ok")
dist.destroy_process_group()
'''


WORKER_SEQ = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "midi-emotion_amd"))
from midiemo.ddp import GradAllReducer
from midiemo.models.music_transformer import MusicTransformerHIP, MusicTransformerMulti
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
# the HEADLINE model's own bucket layout (6 layers: 8 buckets over 20.6 M gradient elements) and the EXACT hook sequence its
# backward emits (MusicTransformerHIP.backward_hook_sequence; tests/test_ddp_gpu.py asserts the engine follows it)
torch.manual_seed(0)
m = MusicTransformerMulti(embedding_dim=512, d_inner=2048, d_condition=128, vocab_size=1007, num_layer=6, num_head=8, max_seq=2048,
                          dropout=0.1, pad_token=0)
ranges, n = m.bucket_ranges(), m.flat_grads.numel()
assert len(ranges) == 8 and ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
seq = MusicTransformerHIP.backward_hook_sequence(6)
assert seq == [7, -1, 6, -1, 5, -1, 4, -1, 3, -1, 2, -1, 1, 0]
g = torch.Generator().manual_seed(7 + rank)
base = torch.randint(-8, 9, (n,), generator=g).float()          # small integers: every partial sum is exact in f32 AND in bf16
ref = [torch.empty(n) for _ in range(world)]
dist.all_gather(ref, base)
total = sum(ref)
for compress in ("", "bf16"):
    for policy in ("window", "eager", "end", "auto"):
        flat = base.clone()
        red = GradAllReducer(lambda: flat, ranges, policy=policy, compress=compress)
        launched = []
        orig = red._launch
        def spy(lo, hi, _o=orig, _l=launched):
            _l.append((lo, hi)); _o(lo, hi)
        red._launch = spy
        for step in range(2 * GradAllReducer.AUTO_PROBE + 1 if policy == "auto" else 1):
            flat.copy_(base)
            del launched[:]
            pol = red.policy                # ("auto": the policy in force for THIS step; finish() may switch it for the next)
            for h in seq:
                red.hook(h)
            red.finish()
            # every element reduced exactly once: a second all-reduce of a range would give world * total there, none base
            assert torch.equal(flat, total), (compress, policy, step, int((flat != total).sum()))
            cover = sorted(launched)
            assert cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), (policy, cover)
            if pol == "eager":             # one collective per bucket, in the order the backward completes them
                assert launched == [tuple(ranges[b]) for b in seq if b >= 0], (policy, launched)
            elif pol == "window":          # parked buckets leave at the next window (adjacent ones merged); only the last two wait for finish()
                assert launched[0] == tuple(ranges[7]) and launched[-1] == (ranges[0][0], ranges[1][1]) and len(launched) == 7, launched
            elif pol == "end":
                assert launched == [(0, n)], launched
        if policy == "auto":
            assert red.decision is not None
print("rank", rank, "ok")
dist.destroy_process_group()
'''


def test_reducer_replays_the_engines_hook_sequence_every_policy_and_compression(tmp_path):
    """GradAllReducer driven with the EXACT bucket_hook sequence the 6-layer engine emits, over the
    headline model's real bucket ranges, under all four policies, plain and with bf16-compressed buckets: every gradient
    element is reduced exactly once (integer-valued gradients: sums exact in f32 and bf16, so equality is exact), the launches
    tile the flat buffer without gap or overlap and leave in bucket order.  Two gloo ranks on the CPU."""
    script = tmp_path / "wseq.py"
    script.write_text(WORKER_SEQ % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_gradient_allreduce_two_gloo_processes(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29513", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_gradient_allreduce_four_gloo_processes(tmp_path):
    """The same worker with FOUR ranks (bucket order, run-merging in _flush and the "auto" decision with more than two
    ranks)."""
    script = tmp_path / "w4.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 4


def test_midi_writer_round_trip_and_reference_semantics():
    """midiemo.midi_writer == data/data_processing_reverse.py:12-53 (tuples_to_mid) written out as a Standard MIDI
    File with pretty_midi's default timing (120 bpm, 220 ticks per quarter: 1 s = 440 ticks); read back and compared."""
    from midiemo.midi_writer import PROGRAMS, VELOCITIES, read_midi, symbols_to_midi_bytes, symbols_to_notes
    from midiemo.vocab import get_maps, ind_list_to_str
    maps = get_maps()
    t2i = maps["tuple2idx"]
    ev = maps["event2idx"]
    ids = [t2i["<START>"], t2i[(ev["ON_PIANO"], 60)], t2i[(ev["TIMESHIFT"], 400)], t2i[(ev["ON_DRUMS"], 36)],
           t2i[(ev["TIMESHIFT"], 104)], t2i[(ev["OFF_PIANO"], 60)], t2i[(ev["OFF_GUITAR"], 50)],      # OFF without ON: ignored
           t2i[(ev["ON_STRINGS"], 72)], t2i[(ev["TIMESHIFT"], 1000)], t2i[(ev["OFF_DRUMS"], 36)],
           t2i[(ev["OFF_STRINGS"], 72)], t2i[(ev["ON_BASS"], 40)], t2i["<PAD>"]]                       # dangling ON: no note
    symbols = ind_list_to_str(ids, maps)
    notes = symbols_to_notes(symbols)
    assert notes["PIANO"] == [(0.0, 0.504, 60)] and notes["GUITAR"] == [] and notes["BASS"] == []
    assert notes["DRUMS"] == [(0.4, 1.504, 36)] and notes["STRINGS"] == [(0.504, 1.504, 72)]
    data = symbols_to_midi_bytes(symbols)
    res, tempo, tracks = read_midi(data)
    assert (res, tempo) == (220, 500000) and set(tracks) == {k.lower() for k in PROGRAMS}
    assert tracks["drums"]["channel"] == 9 and tracks["piano"]["channel"] == 0
    assert len({t["channel"] for t in tracks.values()}) == 5
    for name, (program, _) in PROGRAMS.items():
        assert tracks[name.lower()]["program"] == program
    tk = lambda s: int(round(s * 440))
    assert tracks["piano"]["notes"] == [(0, tk(0.504), 60, VELOCITIES["PIANO"])]
    assert tracks["drums"]["notes"] == [(tk(0.4), tk(1.504), 36, VELOCITIES["DRUMS"])]
    assert tracks["strings"]["notes"] == [(tk(0.504), tk(1.504), 72, VELOCITIES["STRINGS"])]
    assert tracks["guitar"]["notes"] == [] and tracks["bass"]["notes"] == []


def test_lr_schedules_follow_torch_schedulers():
    """train.py's LRSchedule (reference train.py:128-139,326-333,433-434): `cyclic` and `dev_perf` must produce exactly the
    learning rates of torch's CyclicLR / ReduceLROnPlateau (the classes the reference constructs), `constant` must leave a
    restored lr alone, warm-up is linear in the step."""
    sys.path.insert(0, ROOT)
    import train

    class Opt:
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    # cyclic: base = lr_min (config.py:145-146 sets lr = lr_min), warm-up 3 steps, then one scheduler.step() per step
    a = train.parse_args(["--scheduler", "cyclic", "--lr_min", "1e-5", "--lr_max", "1e-3", "--warmup_step", "3"])
    assert a.lr == 1e-5
    opt = Opt(a.lr)
    s = train.LRSchedule(a, opt)
    ref_opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=a.lr)
    ref = torch.optim.lr_scheduler.CyclicLR(ref_opt, a.lr_min, a.lr_max, cycle_momentum=False)
    got, want = [], []
    for step in range(1, 40):
        s.on_step(step)
        got.append(opt.param_groups[0]["lr"])
        if step <= a.warmup_step:
            want.append(a.lr * step / a.warmup_step)
        else:
            ref_opt.step()
            ref.step()
            want.append(ref_opt.param_groups[0]["lr"])
    assert got == pytest.approx(want, rel=1e-12)
    assert got[-1] > got[5] > a.lr_min                       # the triangle is rising (step_size_up = 2000)

    # restart: a run interrupted after n updates and resumed must continue the uninterrupted lr sequence.
    # The train loop calls on_step(step - 1) after update `step`, so n completed updates = on_step(0) .. on_step(n - 1).
    for warm in (3, 0):
        a = train.parse_args(["--scheduler", "cyclic", "--lr_min", "1e-5", "--lr_max", "1e-3", "--warmup_step", str(warm)])
        opt = Opt(a.lr)
        s = train.LRSchedule(a, opt)
        full = []
        for upd in range(1, 31):
            s.on_step(upd - 1)
            full.append(opt.param_groups[0]["lr"])
        for n in (2, 3, 4, 5, 17):
            opt2 = Opt(full[n - 1])                          # the checkpointed optimiser carries the last written lr
            s2 = train.LRSchedule(a, opt2, start_step=n)
            rest = []
            for upd in range(n + 1, 31):
                s2.on_step(upd - 1)
                rest.append(opt2.param_groups[0]["lr"])
            assert rest == pytest.approx(full[n:], rel=1e-12), (warm, n)

    # dev_perf: stepped with the validation loss at evaluations only
    a = train.parse_args(["--scheduler", "dev_perf", "--lr", "1e-3", "--decay_rate", "0.5", "--patience", "1", "--lr_min", "2e-4"])
    opt = Opt(a.lr)
    s = train.LRSchedule(a, opt)
    ref_opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=a.lr)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, factor=0.5, patience=1, min_lr=2e-4)
    for step, v in enumerate([3.0, 2.5, 2.6, 2.7, 2.8, 2.9, 3.0, 3.1, 3.2], 1):
        s.on_step(step)                                      # no per-step effect
        s.on_eval(v)
        ref.step(v)
        assert opt.param_groups[0]["lr"] == ref_opt.param_groups[0]["lr"]
    assert opt.param_groups[0]["lr"] == pytest.approx(2.5e-4) or opt.param_groups[0]["lr"] == pytest.approx(2e-4)

    # constant: a restored learning rate survives (only --overwrite_lr changes it)
    a = train.parse_args(["--scheduler", "constant", "--lr", "2e-5"])
    opt = Opt(7e-6)
    s = train.LRSchedule(a, opt)
    for step in range(1, 5):
        s.on_step(step)
    assert opt.param_groups[0]["lr"] == 7e-6
    # cosine / inv_sqrt closed forms with warm-up
    a = train.parse_args(["--scheduler", "cosine", "--lr", "1e-3", "--warmup_step", "10", "--max_step", "100"])
    opt = Opt(a.lr)
    s = train.LRSchedule(a, opt)
    s.on_step(5)
    assert opt.param_groups[0]["lr"] == pytest.approx(5e-4)
    s.on_step(100)
    assert opt.param_groups[0]["lr"] == pytest.approx(0.0, abs=1e-12)


def test_midi_notes_match_reference_tuples_to_mid(golden_dir):
    """SURVEY 8f #3, pinned: fixture f7 holds what the reference's tuples_to_mid (data_processing_reverse.py:12-53)
    produced for seeded token streams -- every pretty_midi.Instrument(program, is_drum, name) and
    Note(velocity, pitch, start, end) it constructed (oracle/make_host_fixtures.py).  The build's writer must derive the
    same notes, programs and velocities from the same ids; the ids -> symbol strings step is pinned by a checksum."""
    from midiemo.midi_writer import PROGRAMS, VELOCITIES, symbols_to_midi_bytes, symbols_to_notes
    from midiemo.vocab import get_maps, ind_list_to_str
    z = np.load(os.path.join(golden_dir, "f7_host.npz"))
    maps = get_maps()
    for si in (0, 1):
        ids = z[f"midi{si}_ids"].tolist()
        symbols = ind_list_to_str(ids, maps)
        assert sum((i + 1) * len(s) for i, s in enumerate(symbols)) == int(z[f"midi{si}_symbols_crc"])
        notes = symbols_to_notes(symbols)
        assert set(notes) == set(PROGRAMS)
        total = 0
        for name, (program, is_drum) in PROGRAMS.items():
            meta = z[f"midi{si}_{name.lower()}_meta"]
            assert (int(meta[0]), bool(meta[1])) == (program, is_drum), name
            ref = z[f"midi{si}_{name.lower()}_notes"]               # rows: velocity, pitch, start, end
            got = notes[name]
            assert len(got) == len(ref), (name, len(got), len(ref))
            for (s0, e0, p0), r in zip(got, ref):
                assert (VELOCITIES[name], p0) == (int(r[0]), int(r[1])) and abs(s0 - r[2]) < 1e-9 and abs(e0 - r[3]) < 1e-9
            total += len(got)
        assert total > 10 * (si + 1)
        assert len(symbols_to_midi_bytes(symbols)) > 100          # the container itself: test_midi_writer_round_trip_...


def test_sampling_temperature_and_repeat_penalty_match_reference(golden_dir):
    """SURVEY 8f #1, host half: replaying the reference run captured in f7 (tokens drawn and number of surviving choices
    of every step), generate.sampling_temperature / update_repeat_counts must give the per-row temperatures the
    reference used at every step (note vs rest temperature after a TIMESHIFT, repeat penalty; generate.py:138-163,186-189)."""
    sys.path.insert(0, ROOT)
    import generate as G
    from midiemo.vocab import get_maps, timeshift_token_mask
    z = np.load(os.path.join(golden_dir, "f7_host.npz"))
    maps = get_maps()
    is_ts = torch.tensor(timeshift_token_mask(maps), dtype=torch.bool)
    for tag in ("k0p07", "k20p10", "k50p09"):
        steps, B = int(z[f"samp_{tag}_cfg"][2]), int(z[f"samp_{tag}_cfg"][3])
        prev = torch.full((B,), maps["tuple2idx"]["<START>"], dtype=torch.long)
        rc = torch.zeros(B)
        seen_penalty = False
        for s in range(steps):
            temp = G.sampling_temperature(prev, rc, is_ts, 1.2, 0.9, 0.5)
            np.testing.assert_allclose(temp.numpy(), z[f"samp_{tag}_temp"][s], rtol=2e-5)
            seen_penalty |= bool((temp > 1.2 + 1e-6).any())
            n_choices = torch.from_numpy((z[f"samp_{tag}_probs"][s] > 0).sum(-1))
            rc = G.update_repeat_counts(rc, n_choices)
            prev = torch.from_numpy(z[f"samp_{tag}_tokens"][s]).long()
        if tag == "k0p07":
            assert seen_penalty                              # the peaked rows drive the repeat counter past 3


def test_eval_aggregation_matches_reference_accuracy(golden_dir):
    """Runner.evaluate weights every batch's loss and utils.accuracy (top-1 / top-5 over the batch's non-PAD
    targets) by input_.numel() (train.py:256-272).  Fixture f8: the reference's own utils.accuracy on seeded batches with
    ragged trailing PAD + the aggregate those statements give; midiemo.metrics.EvalAccumulator (what train.py's evaluate()
    runs) must reproduce the per-batch numbers and the aggregate."""
    from midiemo.metrics import EvalAccumulator
    z = np.load(os.path.join(golden_dir, "f8_eval.npz"))
    pad, V = int(z["pad_idx"]), int(z["V"])
    ev = EvalAccumulator("cpu")
    ce = torch.nn.CrossEntropyLoss(ignore_index=pad)
    for i in range(int(z["n_batches"])):
        lg, tg, inp = (torch.from_numpy(z[f"b{i}_{n}"]) for n in ("logits", "target", "input"))
        loss = ce(lg.view(-1, V), tg.view(-1))
        assert float(loss) == pytest.approx(float(z[f"b{i}_loss"]), rel=1e-6)
        one = EvalAccumulator("cpu")
        one.add(loss, lg, tg, inp.numel(), pad)
        _, accs = one.result()
        assert accs[1] == pytest.approx(float(z[f"b{i}_acc1"]), rel=1e-6) and accs[5] == pytest.approx(float(z[f"b{i}_acc5"]), rel=1e-6)   # the reference divides in float32
        ev.add(loss, lg, tg, inp.numel(), pad)
    loss, accs = ev.result()
    assert loss == pytest.approx(float(z["avg_loss"]), rel=1e-6)
    assert accs[1] == pytest.approx(float(z["avg_acc1"]), rel=1e-6) and accs[5] == pytest.approx(float(z["avg_acc5"]), rel=1e-6)
    # and it is NOT the pooled accuracy over all valid targets (what round 2 computed): the batches are ragged
    hits = tot = 0
    for i in range(int(z["n_batches"])):
        lg, tg = torch.from_numpy(z[f"b{i}_logits"]), torch.from_numpy(z[f"b{i}_target"])
        v = tg.view(-1) != pad
        hits += int(((lg.view(-1, V).argmax(-1) == tg.view(-1)) & v).sum())
        tot += int(v.sum())
    assert abs(hits / tot - accs[1]) > 1e-4


def test_fused_adam_loads_reference_optimizer_state_by_name():
    """FusedAdamW.load_state_dict takes the reference's optimizer.pt (a torch.optim.Adam state_dict, train.py:403),
    which carries no names -- only positions in the order of the REFERENCE model's parameters().  That order is the
    checkpoint-ABI key order (O.param_shapes = the reference's state_dict order, pinned in test_build_model_contract);
    many tensors share a shape (Wq / Wk / Wv / fc, the LayerNorm vectors), so a permutation would load silently: every
    moment must land under its own NAME."""
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW
    args = dict(vocab_size=97, n_layer=2, n_head=2, d_model=64, d_inner=128, dropout=0.0, d_condition=16,
                conditioning="continuous_concat")
    model, _ = build_model(args)
    cfg = O.Cfg(97, 2, 2, 64, 128, d_condition=16, conditioning="continuous_concat")
    names = list(O.param_shapes(cfg).keys())                      # reference registration order
    ref_params = [torch.nn.Parameter(torch.zeros(O.param_shapes(cfg)[n])) for n in names]
    ref_opt = torch.optim.Adam(ref_params, lr=3e-4)
    g = torch.Generator().manual_seed(4)
    for p in ref_params:                                          # one step with distinct per-tensor gradients
        p.grad = torch.randn(p.shape, generator=g)
    ref_opt.step()
    sd = ref_opt.state_dict()
    opt = FusedAdamW(model, lr=1e-3)
    opt.load_state_dict(sd)
    assert opt.step_count == 1 and opt.param_groups[0]["lr"] == 3e-4
    for i, n in enumerate(names):
        assert torch.equal(model._pview(opt.m, n), sd["state"][i]["exp_avg"]), n
        assert torch.equal(model._pview(opt.v, n), sd["state"][i]["exp_avg_sq"]), n
    # a model whose parameters() order differs from the reference's must be refused, not loaded by position
    assert [n for n, _ in model.named_parameters()] == names


def test_train_cli_accepts_every_reference_flag(capsys):
    """Every flag of the reference's config.py:7-112 parses (a reference command line must not die in argparse): the 13 that
    round 4 lacked are honoured by the loaders (--overfit, --bar_start_prob, --max_transpose, --n_samples, --no_pad,
    --overwrite_dropout), accepted and reported like the reference's own dead flags (--n_bars, --eval_tgt_len,
    --arousal_feature, --find_lr), or answered with a notice (--no_cuda, --reset_scaler); --regression_dir exits with the
    reason (it needs the MIDI -> token direction)."""
    import train
    a = train.parse_args(["--overfit", "--bar_start_prob", "0.9", "--max_transpose", "2", "--n_samples", "10", "--n_bars", "4",
                          "--no_pad", "--eval_tgt_len", "100", "--overwrite_dropout", "--arousal_feature", "tempo", "--find_lr",
                          "--no_cuda", "--reset_scaler", "--num_workers", "6"])
    assert a.overfit and a.bar_start_prob == 0.9 and a.max_transpose == 2 and a.n_samples == 10 and a.no_pad and a.overwrite_dropout
    assert a.num_workers == 0                     # config.py:132-133: --overfit / --debug run without loader workers
    out = capsys.readouterr().out
    for flag in ("--no_cuda", "--reset_scaler", "--find_lr", "--n_bars", "--eval_tgt_len", "--arousal_feature"):
        assert flag in out, (flag, out)
    assert train.parse_args([]).num_workers == 8  # config.py:98 default
    with pytest.raises(SystemExit) as e:
        train.parse_args(["--regression_dir", "gen"])
    assert "MIDI" in str(e.value)
    with pytest.raises(SystemExit):
        train.parse_args(["--full_dataset"])      # config.py:123-124: LPD-full has NaN features
    assert train.parse_args(["--full_dataset", "--conditioning", "none"]).full_dataset
