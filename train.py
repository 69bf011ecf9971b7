"""train.py -- training driver for the emotion-conditioned Music Transformer on MI355X.

Keeps the reference's CLI surface (src/config.py flag names/defaults) and train-loop
semantics (src/train.py:294-333: CrossEntropyLoss(ignore_index=pad), gradient
accumulation, global-norm clip, Adam(lr), LR warm-up, periodic log/checkpoint/eval) and
checkpoint files (model.pt / optimizer.pt / stats.pt / model_config.pt / mappings.pt,
train.py:180,397-407) -- but the step itself runs on the HIP engine:

    loss = model.loss_and_backward(x, cond, y)      # fwd + CE + bwd, flat f32 grads
    reducer.finish()                                # RCCL all-reduce buckets (overlapped with bwd)
    opt.step()                                      # fused clip + AdamW (weight_decay 0 == Adam)

Data: the Lakh/Spotify pipeline (src/data, src/create_dataset) is outside the accelerated
path and its dataset is not shipped; `--synthetic` (default) draws the synthetic batches
defined in SURVEY 8d.  One process per GPU: launch with
    python -m torch.distributed.run --nproc-per-node N train.py ...
Deviations from the reference, on purpose: the default 16-bit tier is bf16 (no GradScaler; BASELINE.json's headline
dtype) -- `--compute_dtype fp16` runs the reference's own autocast dtype with its GradScaler semantics on the device
(midiemo.optim.LossScaler: init 65536, x2 every 2000 finite steps, x0.5 and a skipped step on inf / nan; `scaler.pt` in the
checkpoint like train.py:403-404, `--reset_scaler` like train.py:207);
`loss.item()` is only read every --log_step (the reference syncs every step, train.py:308);
schedulers `cosine`/`inv_sqrt` are implemented (the reference never constructs them,
train.py:129 tests for '--'); `cyclic` and `dev_perf` drive torch's own CyclicLR / ReduceLROnPlateau on a shadow
optimiser and copy its learning rate into the fused optimiser (train.py:132-139).  The reference also calls
`scheduler.step()` without a metric on every step past the warm-up (train.py:333), which raises for ReduceLROnPlateau;
here `dev_perf` is stepped with the validation loss at every evaluation only (train.py:433-434), the part that works.
A restart writes into a fresh time-stamped work_dir (never into the checkpoint it resumes from) and carries
performance.csv over, like the reference (train.py:173,118).
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "midi-emotion_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Generates emotion-based symbolic music (MI355X engine)")
    # ---- reference flags (config.py:7-113), same names and defaults
    p.add_argument("--conditioning", type=str, default="continuous_concat",
                   choices=["none", "discrete_token", "continuous_token", "continuous_concat"])
    p.add_argument("--n_layer", type=int, default=20)
    p.add_argument("--n_head", type=int, default=16)
    p.add_argument("--d_model", type=int, default=768)
    p.add_argument("--d_condition", type=int, default=192)
    p.add_argument("--d_inner", type=int, default=768 * 4)
    p.add_argument("--tgt_len", type=int, default=1216)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=2e-5)
    p.add_argument("--scheduler", default="constant", choices=["cosine", "inv_sqrt", "dev_perf", "constant", "cyclic"])
    p.add_argument("--lr_min", type=float, default=5e-6, help="minimum learning rate (cyclic base / dev_perf floor)")
    p.add_argument("--lr_max", type=float, default=5e-3, help="maximum learning rate for the cyclic scheduler")
    p.add_argument("--decay_rate", type=float, default=0.5, help="ReduceLROnPlateau factor (dev_perf)")
    p.add_argument("--patience", type=int, default=10, help="ReduceLROnPlateau patience in evaluations (dev_perf)")
    p.add_argument("--warmup_step", type=int, default=0)
    p.add_argument("--gen_step", type=int, default=8000, help="generation interval (train.py:335-373)")
    p.add_argument("--gen_len", type=int, default=2048)
    p.add_argument("--max_gen_input_len", type=int, default=-1)
    p.add_argument("--temp_note", type=float, default=1.2)
    p.add_argument("--temp_rest", type=float, default=1.2)
    p.add_argument("--clip", type=float, default=1.0)
    p.add_argument("--batch_size", type=int, default=4, help="sequences per GPU")
    p.add_argument("--accumulate_step", type=int, default=1)
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--log_step", type=int, default=1000)
    p.add_argument("--eval_step", type=int, default=8000)
    p.add_argument("--max_eval_step", type=int, default=1000)
    p.add_argument("--work_dir", default="../output", type=str)
    p.add_argument("--restart_dir", type=str, default=None)
    p.add_argument("--debug", action="store_true", help="do not write files")
    p.add_argument("--max_step", type=int, default=1000000000)
    p.add_argument("--n_emotion_bins", type=int, default=5)
    p.add_argument("--no_amp", action="store_true", help="exact-f32 engine tier instead of a 16-bit one")
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp16"],
                   help="16-bit tier used unless --no_amp: bf16 (default) or fp16 = the reference's autocast dtype, with loss scaling")
    p.add_argument("--overwrite_lr", action="store_true")
    p.add_argument("--regression", action="store_true")
    # ---- additions
    p.add_argument("--synthetic", action="store_true", default=True, help="synthetic token batches (SURVEY 8d); the default")
    # real data (config.py:10-12, train.py:45-93): used when --feature_file is given
    p.add_argument("--data_folder", type=str, default="../data_files/lpd_5/lpd_5_full_transposable")
    p.add_argument("--feature_file", type=str, default=None,
                   help="CSV of per-song features (the reference hard-codes ../data_files/features/pianoroll/"
                        "full_dataset_features_summarized.csv); giving it switches from synthetic to real batches")
    p.add_argument("--full_dataset", action="store_true", help="also train on songs without emotion labels")
    p.add_argument("--always_use_discrete_condition", action="store_true")
    p.add_argument("--num_workers", type=int, default=8)
    # reference flags of the loaders (data/loader.py:24-31 defaults).  The reference parses them (config.py:92-103) and then
    # constructs its Loader WITHOUT them (train.py:64-68: only regression / always_use_discrete_condition are passed; only
    # LoaderExhaustive gets max_samples = --n_samples, train.py:61-63).  DELIBERATE DEVIATION: here --overfit / --no_pad /
    # --bar_start_prob / --max_transpose / a positive --n_samples reach the train and test Loader (defaults unchanged, so a
    # default command line draws the reference's samples: tests/test_data_cpu.py); the exhaustive evaluation mirrors the
    # reference exactly, including its max_samples = -1 default, i.e. data[:-1]: the LAST test song is not evaluated
    p.add_argument("--overfit", action="store_true", help="work on a single sample (debug data folder, no workers)")
    p.add_argument("--bar_start_prob", type=float, default=0.5, help="probability of a training sample starting at a bar")
    p.add_argument("--max_transpose", type=int, default=3, help="maximum transposition in semitones")
    p.add_argument("--n_samples", type=int, default=-1, help="limit the number of songs (faster debugging)")
    p.add_argument("--no_pad", action="store_true", help="do not pad short sequences (they are filtered by the collate)")
    p.add_argument("--overwrite_dropout", action="store_true", help="on restart: replace the checkpoint's dropout by --dropout")
    # reference flags that the reference itself never reads after parsing (config.py:33-38,48-50,90-91): accepted for
    # command-line compatibility, reported when set
    p.add_argument("--n_bars", type=int, default=-1)
    p.add_argument("--eval_tgt_len", type=int, default=-1)
    p.add_argument("--arousal_feature", type=str, default="note_density", choices=["tempo", "note_density"])
    p.add_argument("--find_lr", action="store_true")
    # reference flags without a counterpart on this engine: accepted, answered with a notice
    p.add_argument("--no_cuda", action="store_true", help="(reference: run on the CPU) -- this engine has no CPU path")
    p.add_argument("--reset_scaler", action="store_true", help="on restart: do not load scaler.pt (train.py:207); fp16 tier only")
    p.add_argument("--regression_dir", type=str, default=None,
                   help="(reference: regress emotions of a folder of generated MIDI files, train.py:70-73) -- needs the "
                        "MIDI -> token direction, which is outside this build (DESIGN section 7)")
    p.add_argument("--exhaustive_eval", action="store_true",
                   help="evaluate on every chunk of every test song and exit (config.py:106, train.py:448-461); needs "
                        "--feature_file; --data_folder is then the root that holds maps.pt and lpd_5_full_transposable/")
    p.add_argument("--weight_decay", type=float, default=0.0, help="decoupled decay; 0 == reference Adam")
    args = p.parse_args(argv)
    if args.regression_dir is not None:
        raise SystemExit("--regression_dir: emotion regression over generated MIDI files needs the reference's MIDI -> token "
                         "pipeline (pretty_midi / pypianoroll), which this build does not contain; --regression on the "
                         "dataset (with --feature_file) is supported")
    for flag, msg in (("no_cuda", "there is no CPU path: training runs on the HIP engine (cuda:LOCAL_RANK)"),
                      ("reset_scaler", "" if (args.compute_dtype == "fp16" and not args.no_amp) else
                       "bf16 / f32 tiers use no GradScaler: nothing to reset"),
                      ("find_lr", "the reference parses --find_lr, never runs a finder and only sets debug = True (config.py:135-136); same here")):
        if getattr(args, flag) and msg:
            print(f"[train.py] --{flag}: {msg}")
    if args.find_lr:
        args.debug = True                                      # config.py:135-136
    for flag, default in (("n_bars", -1), ("eval_tgt_len", -1), ("arousal_feature", "note_density")):
        if getattr(args, flag) != default:
            print(f"[train.py] --{flag}: parsed and unused by the reference (config.py), unused here")
    if args.full_dataset and (args.conditioning not in ("discrete_token", "none") or args.regression):
        raise SystemExit("--full_dataset: LPD-full has NaN features (config.py:123-124): conditioning must be discrete_token or none")
    if args.debug or args.overfit:
        args.num_workers = 0                                   # config.py:132-133
    if args.conditioning != "continuous_concat":
        args.d_condition = -1                                  # config.py:120-121
    if args.scheduler == "cyclic":
        args.lr = args.lr_min                                  # config.py:145-146
    if args.exhaustive_eval:
        if not args.feature_file:
            raise SystemExit("--exhaustive_eval needs --feature_file (there is nothing exhaustive about synthetic batches)")
        args.max_eval_step = 0                                 # config.py:123: the whole test set
    if args.regression:
        args.n_layer = 8                                       # config.py:128-130
        args.d_condition = -1
        print("Using 8 layers for regression")
    return args


def synthetic_batch(args, V, B, L, seed, device):
    """SURVEY 8d synthetic inputs for every conditioning mode (regression: <CLS> + tokens, the (valence, arousal) target
    travels in the condition slot as in data/loader.py:166-168,181-182)."""
    g = torch.Generator().manual_seed(seed)
    if args.regression:
        tok = torch.randint(2, V - 1, (B, L), generator=g)
        tok[:, 0] = V - 1                                      # <CLS> is the last symbol of the loader's vocabulary
        cond = torch.rand(B, 2, generator=g) * 2 - 1
        return tok.to(device), cond.to(device), None
    if args.conditioning == "continuous_token":
        tok = torch.randint(2, V, (B, L - 1), generator=g)
        inp, tgt = tok[:, :-1], torch.nn.functional.pad(tok[:, 1:], (2, 0), value=0)   # loader.py:55-57,184-187
    else:
        tok = torch.randint(2, V, (B, L + 1), generator=g)
        if args.conditioning == "discrete_token":
            tok[:, 0] = torch.randint(1007, 1007 + args.n_emotion_bins, (B,), generator=g)
            tok[:, 1] = torch.randint(1007 + args.n_emotion_bins, 1007 + 2 * args.n_emotion_bins, (B,), generator=g)
        inp, tgt = tok[:, :-1], tok[:, 1:]
    if args.conditioning in ("continuous_token", "continuous_concat"):
        cond = torch.rand(B, 2, generator=g) * 2 - 1
    else:
        cond = torch.full((B, 2), float("nan"))
    return inp.contiguous().to(device), cond.to(device), tgt.contiguous().to(device)


def lr_at(args, step):
    """Closed-form schedules: warm-up (train.py:327-331) then cosine / inv_sqrt."""
    if args.warmup_step > 0 and step <= args.warmup_step:
        return args.lr * step / args.warmup_step
    if args.scheduler == "cosine":
        return 0.5 * args.lr * (1 + math.cos(math.pi * min(1.0, step / max(1, args.max_step))))
    return args.lr / math.sqrt(max(1.0, step / max(1, args.warmup_step)))      # inv_sqrt


class LRSchedule:
    """Learning-rate policy of the run (train.py:128-139,326-333,433-434), applied to FusedAdamW.param_groups[0]['lr'].
    constant: the optimiser's lr is never touched (a restored checkpoint keeps its lr unless --overwrite_lr);
    cosine / inv_sqrt: closed form (lr_at); cyclic / dev_perf: torch's CyclicLR / ReduceLROnPlateau run on a shadow
    SGD optimiser with one parameter, so the sequence of learning rates is torch's own."""

    def __init__(self, args, opt, start_step=0):
        self.args, self.opt, self.shadow, self.sched = args, opt, None, None
        if args.scheduler in ("cyclic", "dev_perf"):
            self.shadow = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=opt.param_groups[0]["lr"])
            if args.scheduler == "cyclic":
                self.sched = torch.optim.lr_scheduler.CyclicLR(self.shadow, args.lr_min, args.lr_max, cycle_momentum=False)
            else:
                self.sched = torch.optim.lr_scheduler.ReduceLROnPlateau(self.shadow, factor=args.decay_rate,
                                                                        patience=args.patience, min_lr=args.lr_min)

        if args.scheduler == "cyclic":
            # restart: put the triangle where the interrupted run left it.  A run that completed start_step updates has
            # called on_step(0) .. on_step(start_step - 1); the scheduler stepped for those arguments that were not warm-up
            # steps -- counted with on_step's own predicate (one step ahead before).
            # CyclicLR's only state is its call count (last_epoch) and its lr a function of it: jump there instead of
            # replaying start_step iterations.
            n_warm = min(start_step, args.warmup_step + 1) if args.warmup_step > 0 else 0
            n = start_step - n_warm
            if n > 0:
                self.sched.last_epoch = n - 1
                self.shadow.step()
                self.sched.step()

    def _in_warmup(self, step):
        return self.args.warmup_step > 0 and step <= self.args.warmup_step

    def _copy(self):
        self.opt.param_groups[0]["lr"] = self.shadow.param_groups[0]["lr"]

    def on_step(self, step):
        a = self.args
        if a.scheduler == "constant":
            return
        if a.scheduler in ("cosine", "inv_sqrt"):
            self.opt.param_groups[0]["lr"] = lr_at(a, step)
        elif self._in_warmup(step):
            self.opt.param_groups[0]["lr"] = a.lr * step / a.warmup_step
        elif a.scheduler == "cyclic":
            self.shadow.step()                       # keeps torch's "optimizer.step() before scheduler.step()" order
            self.sched.step()
            self._copy()

    def on_eval(self, val_loss):
        if self.args.scheduler == "dev_perf":
            self.sched.step(val_loss)
            self._copy()


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py: the MI355X engine needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from midiemo.ddp import GradAllReducer, broadcast_params
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW, LossScaler
    from midiemo.vocab import get_maps

    torch.manual_seed(args.seed if args.seed > 0 else 0)
    train_loader = test_loader = None
    if args.feature_file:
        # real data: feature table -> splits -> per-song bar files (train.py:45-93); one shard of the songs per rank
        import random as _random
        import numpy as _np
        from midiemo.data import Loader, filter_collate, preprocess_features
        if args.seed > 0:
            _random.seed(args.seed + rank)
            _np.random.seed(args.seed + rank)
        n_bins = args.n_emotion_bins if args.conditioning == "discrete_token" and not args.regression else None
        train_feats, test_feats = preprocess_features(args.feature_file, n_bins=n_bins,
                                                      conditional=args.conditioning != "none" or args.regression,
                                                      use_labeled_only=not args.full_dataset)
        kw = dict(always_use_discrete_condition=args.always_use_discrete_condition, regression=args.regression,
                  pad=not args.no_pad, overfit=args.overfit, max_samples=args.n_samples if args.n_samples > 0 else None)
        if args.exhaustive_eval:                               # train.py:58-63: max_samples=args.n_samples as is (-1 -> data[:-1])
            from midiemo.data import LoaderExhaustive
            test_ds = LoaderExhaustive(args.data_folder, test_feats, args.tgt_len, args.conditioning, **dict(kw, max_samples=args.n_samples))
            train_ds = test_ds                                 # never iterated: the run evaluates and exits
        else:
            kw.update(bar_start_prob=args.bar_start_prob, max_transpose=args.max_transpose)
            train_ds = Loader(args.data_folder, train_feats, args.tgt_len, args.conditioning, **kw)
            test_ds = Loader(args.data_folder, test_feats, args.tgt_len, args.conditioning, **kw)
        if args.conditioning == "discrete_token" and not args.exhaustive_eval:
            # the reference takes the vocabulary from the test split alone (train.py:76-80), which silently assumes that
            # every emotion bin occurs there; use the union of both splits so that small collections train too
            syms = sorted({s[k] for ds in (train_ds, test_ds) for s in ds.data for k in ("valence", "arousal")})
            train_ds.set_extra_tokens(syms)
            test_ds.set_extra_tokens(syms)
        maps = test_ds.get_maps()

        def make_loader(ds, shuffle):
            sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=shuffle) if world > 1 else None
            return torch.utils.data.DataLoader(ds, args.batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                               num_workers=args.num_workers, collate_fn=filter_collate, pin_memory=True,
                                               drop_last=True)
        train_loader, test_loader = make_loader(train_ds, not args.debug), make_loader(test_ds, False)
        if rank == 0:
            print(f"Data loader lengths\nTrain: {len(train_ds)}\nTest: {len(test_ds)}")
    else:
        maps = get_maps(n_emotion_bins=args.n_emotion_bins if args.conditioning == "discrete_token" else 0)
        if args.regression:                                   # data/loader.py:77-79: <CLS> joins the vocabulary
            maps["tuple2idx"]["<CLS>"] = len(maps["idx2tuple"])
            maps["idx2tuple"][len(maps["idx2tuple"])] = "<CLS>"
    V = len(maps["tuple2idx"])
    pad_idx = maps["tuple2idx"]["<PAD>"]
    config = dict(vars(args), vocab_size=V, compute_dtype="fp32" if args.no_amp else args.compute_dtype)
    work_dir = os.path.join(args.work_dir, ("DEBUG_" if args.debug else "") + time.strftime("%Y%m%d-%H%M%S"))
    base_dir, n_try = work_dir, 0
    while os.path.exists(work_dir):                           # one-second names: never reuse (or restart into) an existing run
        n_try += 1
        work_dir = "%s-%d" % (base_dir, n_try)
    restart = os.path.join(args.work_dir, args.restart_dir) if args.restart_dir else None
    if restart:
        config = torch.load(os.path.join(restart, "model_config.pt"))
        # the kernels index the embedding / CE tables with the token ids unchecked: a checkpoint whose vocabulary or
        # conditioning differs from what this command line builds would read out of bounds -- refuse instead
        want = {"vocab_size": V, "conditioning": args.conditioning, "regression": bool(args.regression)}
        bad = {k: (config.get(k), v) for k, v in want.items() if k in config and config.get(k) != v}
        if bad:
            raise SystemExit("restart: model_config.pt disagrees with the command line (saved, requested): %s" % bad)
        # train.py:175.  build_model() reads ONLY the saved config once load_config_dict is given (the reference's own
        # --overwrite_dropout therefore never fires: its flag is looked up in model_config.pt); here the command line's
        # flag and rate are merged into the loaded config so that the documented behaviour is the actual one
        if args.overwrite_dropout:
            config = dict(config, overwrite_dropout=True, dropout=args.dropout)
        if not args.no_amp and config.get("compute_dtype") in ("bf16", "fp16"):
            config = dict(config, compute_dtype=args.compute_dtype)      # the 16-bit tier is a run-time choice, not a model property
        model, _ = build_model(vars(args), load_config_dict=config)
        model.load_state_dict(torch.load(os.path.join(restart, "model.pt"), map_location="cpu"))
        # work_dir stays the fresh time-stamped directory (train.py:173-180 writes there, never into restart_dir)
    else:
        model, config = build_model(config)
    model = model.to(device).train()
    broadcast_params(model.flat_params)
    model.mark_params_changed()
    model.seed_dropout((args.seed if args.seed > 0 else 0) * 1000 + rank)
    # f16 tier: the reference's GradScaler (train.py:108), state and decisions on the device
    scaler = LossScaler(device) if model.compute_dtype == torch.float16 else None
    opt = FusedAdamW(model, lr=args.lr, clip=args.clip, weight_decay=args.weight_decay, scaler=scaler)
    stats = {"step": 0, "hour": 0.0, "epoch": 0, "sample": 0}
    if restart:
        # optimizer.pt and stats.pt are restored independently, each tolerating absence (train.py:187-211)
        try:
            opt.load_state_dict(torch.load(os.path.join(restart, "optimizer.pt"), map_location=device))
        except FileNotFoundError:
            print("Optimizer was not saved. Start from scratch.")
        except Exception as e:
            print("optimizer state not restored (%s: %s); moments start from zero" % (type(e).__name__, e))
        try:
            stats = dict(stats, **torch.load(os.path.join(restart, "stats.pt")))
        except Exception as e:
            print("stats.pt not restored (%s); step / hour / epoch start from zero" % type(e).__name__)
        scaler_fp = os.path.join(restart, "scaler.pt")
        if scaler is not None and os.path.exists(scaler_fp) and not args.reset_scaler:      # train.py:207-209
            try:
                scaler.load_state_dict(torch.load(scaler_fp))
            except Exception as e:
                print("scaler.pt not restored (%s); the loss scale starts from 65536" % type(e).__name__)
        if args.overwrite_lr:
            opt.param_groups[0]["lr"] = args.lr
        if rank == 0 and not args.debug and os.path.exists(os.path.join(restart, "performance.csv")):
            os.makedirs(work_dir, exist_ok=True)                # train.py:173,118: the resumed run continues the table
            import shutil
            shutil.copy(os.path.join(restart, "performance.csv"), os.path.join(work_dir, "performance.csv"))
    sched = LRSchedule(args, opt, start_step=int(stats.get("step", 0)))
    reducer = GradAllReducer(lambda: model.flat_grads, model.bucket_ranges())
    if rank == 0 and not args.debug:
        os.makedirs(work_dir, exist_ok=True)
        torch.save(config, os.path.join(work_dir, "model_config.pt"))      # train.py:180
        torch.save(maps, os.path.join(work_dir, "mappings.pt"))            # train.py:114
    n_params = sum(p.numel() for p in model.parameters())
    if rank == 0:
        print(f"#params = {n_params}  world = {world}  dtype = {config['compute_dtype']}  work_dir = {work_dir}")

    B, L = args.batch_size, args.tgt_len
    step = stats["step"]
    micro = 0
    loss_acc = torch.zeros((), device=device)
    n_acc = 0
    t0 = time.time()
    tok_per_micro = world * B * L

    def to_device(batch):
        """collated (input, condition, target) -> device tensors; None when every sample of the batch was rejected."""
        if not batch or not isinstance(batch[0], torch.Tensor) or batch[0].numel() == 0:
            return None
        x, c = batch[0], batch[1]
        y = batch[2] if len(batch) > 2 and isinstance(batch[2], torch.Tensor) else None       # regression: no token target
        return (x.to(device, non_blocking=True), c.to(device, non_blocking=True),
                y.to(device, non_blocking=True) if y is not None else None)

    def real_batches(loader, epochs_counter=None):
        while True:
            if hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(stats["epoch"])
            n = 0
            for batch in loader:
                b = to_device(batch)
                if b is not None:
                    n += 1
                    yield b
            if epochs_counter is not None:
                stats["epoch"] += 1
            if n == 0:
                raise SystemExit("the data loader produced no usable sample (check --data_folder / min_n_instruments)")

    train_iter = real_batches(train_loader, True) if train_loader is not None else None

    def evaluate():
        """The reference's Runner.evaluate (train.py:222-275): per-batch CE loss and utils.accuracy (top-1 / top-5 over the
        batch's non-PAD targets), each weighted by input_.numel() -- midiemo.metrics.EvalAccumulator, pinned to the
        reference's own utils.accuracy by tests/golden/f8_eval.npz."""
        from midiemo.metrics import EvalAccumulator
        model.eval()
        ev = EvalAccumulator(device)
        acc = torch.zeros(4, device=device, dtype=torch.float64)          # regression: L1 sums, #sequences
        n = min(args.max_eval_step, 8) if test_loader is None else (args.max_eval_step if args.max_eval_step > 0 else 1 << 60)
        test_iter = iter(test_loader) if test_loader is not None else None
        with torch.no_grad():
            for i in range(n):
                if test_iter is None:
                    x, c, y = synthetic_batch(args, V, B, L, 10_000_019 + i * 31 + rank, device)
                else:
                    b = to_device(next(test_iter, None))
                    if b is None:
                        break
                    x, c, y = b
                if args.regression:                     # train.py:246-254: clamp, L1 per dimension
                    pred = model(x).clamp(-1.0, 1.0)
                    nb = float(pred.shape[0])
                    acc[0] += (pred - c).abs().mean().double() * nb
                    acc[1] += (pred[:, 0] - c[:, 0]).abs().mean().double() * nb
                    acc[2] += (pred[:, 1] - c[:, 1]).abs().mean().double() * nb
                    acc[3] += nb
                    continue
                loss, logits = model.loss_and_backward(x, c, y, backward=False, return_logits=True)
                ev.add(loss, logits, y, x.numel(), pad_idx)
        model.train()
        if not args.regression:
            if world > 1:
                dist.all_reduce(ev.acc)
            return ev.result()
        if world > 1:
            dist.all_reduce(acc)
        a = acc.tolist()
        den = max(a[3], 1.0)
        return a[0] / den, {1: a[1] / den, 5: a[2] / den}

    # performance.csv: same columns as the reference (train.py:116-118), one row per log / eval event
    import csv
    perf_cols = ["epoch", "step", "hour", "lr", "trn_loss", "val_loss", "val_l1_v", "val_l1_a"]
    perf_path = os.path.join(work_dir, "performance.csv")

    def perf_row(**kw):
        if rank != 0 or args.debug:
            return
        new_file = not os.path.exists(perf_path)
        with open(perf_path, "a", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=perf_cols)
            if new_file:
                w.writeheader()
            w.writerow({k: kw.get(k, float("nan")) for k in perf_cols})

    def generate_samples(step_no):
        """In-training sample generation with the fixed condition set (train.py:335-373)."""
        from generate import generate
        max_input_len = args.max_gen_input_len if args.max_gen_input_len > 0 else args.tgt_len
        primers, disc, cont = [["<START>"]], None, None
        if args.conditioning == "none":
            primers = [["<START>"] for _ in range(4)]
        elif args.conditioning == "discrete_token":
            disc = [["<V-2>", "<A-2>"], ["<V-2>", "<A2>"], ["<V2>", "<A-2>"], ["<V2>", "<A2>"]]
        else:
            cont = [[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]]
        was_training = model.training
        with torch.no_grad():
            generate(model, maps, device, os.path.join(work_dir, "generations", "training"), args.conditioning,
                     debug=args.debug, verbose=False, amp=not args.no_amp, discrete_conditions=disc,
                     continuous_conditions=cont, min_n_instruments=1, gen_len=args.gen_len, max_input_len=max_input_len,
                     step=str(step_no), primers=primers, temperatures=[args.temp_note, args.temp_rest])
        model.train(was_training)

    if args.exhaustive_eval:                                   # train.py:448-461
        v, accs = evaluate()
        if rank == 0:
            if args.regression:
                print("Exhaustive evaluation | Loss: {:7.4f}, l1_v: {:7.4f}, l1_a: {:7.4f}".format(v, accs[1], accs[5]))
            else:
                print("Exhaustive evaluation | Loss: {:7.4f}, ppl: {:5.2f}, top1: {:7.4f}, top5: {:7.4f}".format(
                    v, math.exp(min(v, 20)), accs[1], accs[5]))
        if world > 1:
            dist.destroy_process_group()
        return
    try:
        while step < args.max_step:
            if train_iter is None:
                x, c, y = synthetic_batch(args, V, B, L, 1234 + rank + 7919 * (step * args.accumulate_step + micro), device)
            else:
                x, c, y = next(train_iter)
            last = (micro + 1) % args.accumulate_step == 0
            # every micro-batch contributes grad/accumulate_step (train.py:309); buckets are exchanged on the last one
            hook = reducer.hook if (last and world > 1) else None
            ls = scaler.scale_tensor if scaler is not None else None     # scaler.scale(loss).backward(), train.py:317
            if args.regression:                                 # train.py:282-284: L1Loss(model(input), condition)
                loss = model.loss_and_backward(x, c, grad_scale=1.0 / args.accumulate_step, bucket_hook=hook, loss_scale=ls)
            else:
                loss = model.loss_and_backward(x, c, y, grad_scale=1.0 / args.accumulate_step, bucket_hook=hook, loss_scale=ls)
            loss_acc += loss.detach()
            n_acc += 1
            micro += 1
            if not last:
                continue
            micro = 0
            reducer.finish()
            step += 1
            # the reference steps the optimiser FIRST and only then writes the warm-up / scheduler learning rate, with its
            # 0-based step counter (train.py:319-333,438): update n (1-based) runs with the rate written after update n - 1,
            # i.e. the initial rate, then lr * 0 / warmup, lr * 1 / warmup, ...  Same sequence here.
            opt.step(grad_scale=reducer.grad_scale)
            sched.on_step(step - 1)
            if step % args.gen_step == 0 and not args.regression and rank == 0:
                generate_samples(step)
            if step % args.log_step == 0 or step == args.max_step:
                cur = float(loss_acc.item()) / max(n_acc, 1)                 # the only host sync
                el = time.time() - t0
                if rank == 0:
                    print("| step {:>8d} | lr {:.3e} | ms/batch {:7.2f} | tok/s {:10.0f} | loss {:7.4f} | ppl {:9.3f}".format(
                        step, opt.param_groups[0]["lr"], 1000 * el / max(n_acc, 1), n_acc * tok_per_micro / el, cur,
                        math.exp(min(cur, 20))))
                    perf_row(epoch=stats["epoch"], step=step, hour=stats["hour"] + el / 3600, lr=opt.param_groups[0]["lr"],
                             trn_loss=cur)
                    if not args.debug:
                        stats.update(step=step, hour=stats["hour"] + el / 3600)
                        torch.save(model.state_dict(), os.path.join(work_dir, "model.pt"))
                        torch.save(opt.state_dict(), os.path.join(work_dir, "optimizer.pt"))
                        if scaler is not None:                   # train.py:403-404
                            torch.save(scaler.state_dict(), os.path.join(work_dir, "scaler.pt"))
                        torch.save(stats, os.path.join(work_dir, "stats.pt"))
                loss_acc.zero_()
                n_acc = 0
                t0 = time.time()
            if step % args.eval_step == 0:
                v, accs = evaluate()
                sched.on_eval(v)
                if args.regression:
                    if rank == 0:
                        print("| eval at step {:>8d} | valid loss {:7.4f} | l1_v {:.4f} | l1_a {:.4f}".format(step, v, accs[1], accs[5]))
                    perf_row(epoch=stats["epoch"], step=step, hour=stats["hour"], lr=opt.param_groups[0]["lr"], val_loss=v,
                             val_l1_v=accs[1], val_l1_a=accs[5])
                    continue
                if rank == 0:
                    print("| eval at step {:>8d} | valid loss {:7.4f} | ppl {:9.3f} | top-1 {:.4f} | top-5 {:.4f}".format(
                        step, v, math.exp(min(v, 20)), accs[1], accs[5]))
                perf_row(epoch=stats["epoch"], step=step, hour=stats["hour"], lr=opt.param_groups[0]["lr"], val_loss=v)
    except KeyboardInterrupt:
        print("Exiting from training early")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
