"""Worker of tests/test_ddp_gpu.py: one data-parallel rank running REAL train steps of the HIP engine
(loss_and_backward with bucket_hook=GradAllReducer.hook -> finish -> FusedAdamW.step(grad_scale=1/world)), every rank on
cuda:0 (1-GPU boxes).  Rank 0 writes, for every step, the parameters the step started from, the averaged flat gradient
the optimiser was given and the parameters after the update (+ Adam's m / v at the end)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "midi-emotion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = dict(vocab_size=1007, n_layer=2, n_head=2, d_model=128, d_inner=256, dropout=0.0, d_condition=32,
           conditioning="continuous_concat")
B, L, STEPS = 2, 96, 3
# --big: the headline model (BASELINE configs 2 / 3: 6 layers, d 512, 8 heads) at B = 2 x L = 256 per rank
BIG = dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, dropout=0.0, d_condition=128,
           conditioning="continuous_concat")


SMALL = CFG


def use_big(big=True):
    global CFG, L
    CFG, L = (BIG, 256) if big else (SMALL, 96)


def micro_batch(step, micro, rank, device):
    g = torch.Generator().manual_seed(9000 + 131 * step + 17 * micro + rank)
    tok = torch.randint(2, CFG["vocab_size"], (B, L + 1), generator=g)
    cond = torch.rand(B, 2, generator=g) * 2 - 1
    return tok[:, :-1].contiguous().to(device), cond.to(device), tok[:, 1:].contiguous().to(device)


def build(compute_dtype, device):
    from midiemo.models.build_model import build_model
    torch.manual_seed(7)
    model, _ = build_model(dict(CFG, compute_dtype=compute_dtype))
    return model.to(device).train()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", required=True)
    ap.add_argument("--accumulate", type=int, default=1)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--compute_dtype", default="fp32")
    ap.add_argument("--out", required=True)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--compress", default="", help="bf16: bf16-compressed buckets (GradAllReducer compress)")
    ap.add_argument("--inject_inf_step", type=int, default=-1,
                    help="f16 tier: rank 1 alone writes an inf into its local gradient at this step -- every rank must skip that update together")
    ap.add_argument("--dump_prep", action="store_true", help="diagnosis: also save the prepared (cast) weights every step ran with")
    a = ap.parse_args()
    if a.big:
        use_big()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(a.backend)
    from midiemo.ddp import GradAllReducer, broadcast_params
    from midiemo.optim import FusedAdamW, LossScaler
    model = build(a.compute_dtype, dev)
    scaler = LossScaler(dev) if a.compute_dtype == "fp16" else None
    if rank != 0:
        with torch.no_grad():
            model.flat_params.add_(0.01 * rank)             # the broadcast has to repair this
        model.mark_params_changed()
    broadcast_params(model.flat_params)
    model.mark_params_changed()
    opt = FusedAdamW(model, lr=2e-5, clip=1.0, scaler=scaler)          # same as the reference run in test_ddp_gpu.py
    red = GradAllReducer(lambda: model.flat_grads, model.bucket_ranges(), policy=a.policy, compress=a.compress)
    # the bucket_hook sequence the engine really emits must be the documented one (the CPU reducer test replays exactly it)
    seen_hooks = []
    red_hook = red.hook

    def hook(b):
        seen_hooks.append(b)
        red_hook(b)
    g1 = None
    gs, pb, pa, preps = [], [], [], []
    for step in range(STEPS):
        pb.append(model.flat_params.detach().cpu().clone())          # the parameters this step's gradient is taken at
        for micro in range(a.accumulate):
            x, c, y = micro_batch(step, micro, rank, dev)
            last = micro + 1 == a.accumulate
            if last and step == a.inject_inf_step and rank == 1:
                model.flat_grads[777] = float("inf")         # a local overflow on ONE rank: the all-reduce spreads it, all ranks skip
            model.loss_and_backward(x, c, y, grad_scale=1.0 / a.accumulate, bucket_hook=hook if last else None,
                                    loss_scale=scaler.scale_tensor if scaler is not None else None)
            if a.dump_prep and micro == 0 and rank == 0:
                preps.append([{k: v.detach().cpu().clone() for k, v in L.items() if torch.is_tensor(v)} for L in model._prep["layers"]] +
                             [{k: v.detach().cpu().clone() for k, v in model._prep["head"].items()}])
        red.finish()
        assert seen_hooks == type(model).backward_hook_sequence(model.num_layer), seen_hooks
        del seen_hooks[:]
        if step == 0:
            g1 = (model.flat_grads * red.grad_scale).clone()
        gs.append((model.flat_grads * red.grad_scale).cpu())         # averaged gradient as the optimiser sees it
        opt.step(grad_scale=red.grad_scale)
        pa.append(model.flat_params.detach().cpu().clone())
    torch.cuda.synchronize()
    # the invariant data-parallel training rests on (SURVEY 8e "identical optimizer state evolution on every rank"): every
    # rank holds BIT-identical parameters and Adam moments after every update -- same reduced gradients, and an optimiser
    # whose clip coefficient does not depend on block arrival order (me_sumsq's ordered block sums)
    checks = [("params", model.flat_params.detach()), ("m", opt.m), ("v", opt.v)]
    if scaler is not None:
        checks.append(("scaler state", scaler.state))       # scale, tracker, steps taken / skipped: decided identically everywhere
    for name, t in checks:
        ref = t.clone()
        dist.broadcast(ref, src=0)
        ndiff = int((ref != t).sum())
        assert ndiff == 0, "rank %d: %d entries of %s differ from rank 0 after %d steps" % (rank, ndiff, name, STEPS)
    if rank == 0:
        torch.save({"g1": g1.cpu(), "params": model.flat_params.detach().cpu().clone(), "grads": gs, "params_before": pb,
                    "params_steps": pa, "m": opt.m.cpu(), "v": opt.v.cpu(), "preps": preps,
                    "scaler": scaler.state.cpu() if scaler is not None else None}, a.out)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "done")


if __name__ == "__main__":
    main()
