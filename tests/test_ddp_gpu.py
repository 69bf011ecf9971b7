"""SURVEY 8e / VERDICT r1 item 1c: the data-parallel path through the REAL train step.  Two ranks on ONE GPU run
3 steps of loss_and_backward(bucket_hook=GradAllReducer.hook) -> finish() -> FusedAdamW.step(grad_scale=1/world) under
each overlap policy (and with gradient accumulation) and must reproduce the 1-rank trajectory on the concatenated
batch: averaged flat gradient of step 1 and parameters after 3 steps.  (The gloo test in test_host_cpu.py only feeds
the reducer synthetic vectors; this one checks that _backward_impl hands over FINAL gradients bucket by bucket.)"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def single_rank_reference(world, accumulate, compute_dtype, big=False, rank_order=None):
    import ddp_worker as W
    from midiemo.optim import FusedAdamW
    W.use_big(big)
    dev = torch.device("cuda", 0)
    model = W.build(compute_dtype, dev)
    opt = FusedAdamW(model, lr=2e-5, clip=1.0)
    p0 = model.flat_params.detach().cpu().clone()
    # d loss / d Wk.bias is exactly zero (softmax shift invariance): both runs step on rounding noise there and Adam
    # normalises noise to +-lr, so those entries are excluded from the trajectory comparison
    keep = torch.ones(p0.numel(), dtype=torch.bool)
    for name, (o, n, _) in model._slices.items():
        if name.endswith("Wk.bias"):
            keep[o:o + n] = False
    g1 = None
    for step in range(W.STEPS):
        for micro in range(accumulate):
            parts = [W.micro_batch(step, micro, r, dev) for r in (rank_order or range(world))]
            x, c, y = (torch.cat([p[i] for p in parts]) for i in range(3))
            model.loss_and_backward(x, c, y, grad_scale=1.0 / accumulate)
        if step == 0:
            g1 = model.flat_grads.clone()
        opt.step()
    torch.cuda.synchronize()
    return g1.cpu(), model.flat_params.detach().cpu().clone(), p0, keep


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run_workers(tmp_path, policy, accumulate, backend, compute_dtype, port, nproc=2, big=False):
    out = str(tmp_path / f"ddp_{policy}_{accumulate}_{backend}_{nproc}_{int(big)}.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MIDIEMO_DDP_FORCE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "ddp_worker.py"), "--policy", policy, "--accumulate", str(accumulate),
                        "--backend", backend, "--compute_dtype", compute_dtype, "--out", out] + (["--big"] if big else []),
                       capture_output=True, text=True, env=env, timeout=600)
    return r, out


@pytest.mark.parametrize("policy,accumulate", [("window", 1), ("eager", 1), ("end", 1), ("window", 2)])
def test_two_ranks_one_gpu_match_single_rank(tmp_path, policy, accumulate):
    r, out = run_workers(tmp_path, policy, accumulate, "gloo", "fp32", 29541 + accumulate + len(policy))
    assert r.returncode == 0 and r.stdout.count("done") == 2, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    g1, params, p0, keep = single_rank_reference(2, accumulate, "fp32")
    eg, ep = rel(got["g1"][keep], g1[keep]), rel(got["params"][keep], params[keep])
    eu = rel((got["params"] - p0)[keep], (params - p0)[keep])
    print("ddp %s acc=%d: grad rel %.2e, params-after-3-steps rel %.2e, 3-step update rel %.2e" % (policy, accumulate, eg, ep, eu))
    assert eg <= 1e-5, eg                  # f32 tier: same sums in a different order
    assert ep <= 1e-4, ep                  # lr 2e-5: dominated by the few noise-sign entries (each 2 lr per step)
    assert eu <= 2e-2, eu                  # Adam turns rounding noise of near-zero gradients into +-lr: a few entries differ


def test_two_ranks_bf16_headline_model_accumulate(tmp_path):
    """the tier and the model the benchmark times -- bf16 storage, hi + lo residual stream, bf16 weight refresh
    after every reduced step, 6 layers d512 8 heads -- with world = 2 and --accumulate 2 through the model's backward
    (grouped weight-gradient launches hand their buckets over at the end of each layer).  Bound DERIVED, not picked: the
    2-rank run differs from the 1-rank run on the concatenated batch only by f32 summation order (each rank sums its own
    rows, the all-reduce adds the two), the same kind of difference two 1-rank bf16 runs show when the rows of the batch
    are fed in the other order; the 2-rank deviation must stay within 4 x that."""
    r, out = run_workers(tmp_path, "window", 2, "gloo", "bf16", 29583, big=True)
    assert r.returncode == 0 and r.stdout.count("done") == 2, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    g1, params, p0, keep = single_rank_reference(2, 2, "bf16", big=True)
    g1p, paramsp, _, _ = single_rank_reference(2, 2, "bf16", big=True, rank_order=(1, 0))
    eg, eu = rel(got["g1"][keep], g1[keep]), rel((got["params"] - p0)[keep], (params - p0)[keep])
    bg, bu = rel(g1p[keep], g1[keep]), rel((paramsp - p0)[keep], (params - p0)[keep])
    print("ddp bf16 6L d512 acc=2: grad rel %.2e (row-order noise of two 1-rank runs %.2e), 3-step update rel %.2e (%.2e)" %
          (eg, bg, eu, bu))
    assert eg <= 4 * bg + 1e-7, (eg, bg)
    assert eu <= 4 * bu + 1e-7, (eu, bu)


def test_two_ranks_one_gpu_rccl_backend(tmp_path):
    """The same through RCCL (backend "nccl").  Two ranks on one device is not a configuration RCCL promises to
    support: if the communicator refuses it the test is skipped (the driver's multi-GPU run covers RCCL)."""
    r, out = run_workers(tmp_path, "window", 1, "nccl", "fp32", 29561)
    if r.returncode != 0:
        msg = r.stdout + r.stderr
        if "Duplicate GPU detected" in msg or "ncclInvalidUsage" in msg:
            # measured on the MI355X boxes (RCCL 2.26.6): "Duplicate GPU detected : rank 1 and rank 0 both on CUDA device"
            pytest.skip("RCCL refuses two ranks on one device (ncclInvalidUsage: Duplicate GPU detected)")
        assert False, msg[-4000:]
    got = torch.load(out)
    g1, params, p0, keep = single_rank_reference(2, 1, "fp32")
    assert rel(got["g1"][keep], g1[keep]) <= 1e-5 and rel(got["params"][keep], params[keep]) <= 1e-4


@pytest.mark.parametrize("policy", ["window", "eager", "end"])
def test_one_rank_through_rccl(tmp_path, policy):
    """The RCCL code path itself (backend "nccl": communicator on the device, asynchronous all-reduce of every bucket on
    RCCL's stream, the comm-window / eager / end policies waiting on the work handles, 1 / world folded into the optimiser)
    with the one rank a single-GPU box allows: the trajectory must equal the plain single-process run."""
    r, out = run_workers(tmp_path, policy, 1, "nccl", "fp32", 29571 + len(policy), nproc=1)
    assert r.returncode == 0 and r.stdout.count("done") == 1, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    g1, params, p0, keep = single_rank_reference(1, 1, "fp32")
    assert rel(got["g1"][keep], g1[keep]) <= 1e-6 and rel(got["params"][keep], params[keep]) <= 1e-6


def test_bench_one_rank_under_the_launcher_rccl():
    """bench.py under the driver's launch line with one rank and the real backend: init_process_group("nccl") on the
    device, RCCL barrier and MAX all-reduce around the timed region."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MIDIEMO_BENCH_FORCE_DIST="1", MIDIEMO_DDP_FORCE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no_decode", "--no_extra", "--no_cpu_baseline", "--no_probe"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 32 and d["value"] > 0


def test_bench_two_ranks_on_one_device():
    """bench.py's N > 1 control flow (rank setup from the launcher's environment, per-rank batches, barrier + max over
    ranks, rank 0 prints ONE JSON line with the whole-job aggregate) under the driver's own launch line, with both ranks
    on cuda:0 and gloo instead of RCCL (RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, MIDIEMO_BENCH_ONE_DEVICE="1", MIDIEMO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no_decode", "--no_extra", "--no_cpu_baseline", "--no_probe"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 64 * 1024 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
