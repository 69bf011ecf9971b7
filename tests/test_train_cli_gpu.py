"""train.py CLI (GPU): BASELINE config 1 plumbing (none, 2L d256 h4 di1024 seq256 batch2) through the
HIP engine, loss decreases; checkpoint files use the reference's names and reload."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_train_cli_config1_and_checkpoints(tmp_path, capsys):
    import train
    argv = ["--conditioning", "none", "--n_layer", "2", "--d_model", "256", "--n_head", "4", "--d_inner", "1024",
            "--tgt_len", "256", "--batch_size", "2", "--lr", "1e-3", "--max_step", "30", "--log_step", "10",
            "--eval_step", "30", "--work_dir", str(tmp_path), "--dropout", "0.1", "--seed", "1"]
    train.main(argv)
    out = capsys.readouterr().out
    losses = [float(l.split("| loss")[1].split("|")[0]) for l in out.splitlines() if "| loss" in l]
    assert len(losses) == 3 and losses[-1] < losses[0] - 0.05, losses
    assert "valid loss" in out and "top-1" in out and "top-5" in out
    ev = [l for l in out.splitlines() if "top-1" in l][0]
    top1, top5 = float(ev.split("top-1")[1].split("|")[0]), float(ev.split("top-5")[1])
    assert 0.0 <= top1 <= top5 <= 1.0
    run = [d for d in os.listdir(tmp_path)][0]
    files = set(os.listdir(tmp_path / run))
    assert {"model.pt", "optimizer.pt", "stats.pt", "model_config.pt", "mappings.pt", "performance.csv"} <= files
    import csv
    rows = list(csv.DictReader(open(tmp_path / run / "performance.csv")))
    assert list(rows[0].keys()) == ["epoch", "step", "hour", "lr", "trn_loss", "val_loss", "val_l1_v", "val_l1_a"]
    assert [int(r["step"]) for r in rows] == [10, 20, 30, 30]           # three log rows + one eval row (train.py:389,426)
    assert rows[-1]["trn_loss"] == "nan" and float(rows[-1]["val_loss"]) > 0
    sd = torch.load(tmp_path / run / "model.pt")
    assert "enc_layers.1.rga.E" in sd and sd["fc.weight"].shape == (1007, 256)
    assert torch.load(tmp_path / run / "stats.pt")["step"] == 30
    # resume (train.py:156-216 semantics): continues from step 30
    train.main(argv[:-6] + ["--work_dir", str(tmp_path), "--restart_dir", run, "--max_step", "40", "--log_step", "10",
                            "--eval_step", "1000", "--conditioning", "none"])
    out = capsys.readouterr().out
    assert "step       40" in out


def test_train_cli_fp16_tier_scaler_checkpoint_and_overwrite_dropout(tmp_path, capsys):
    """--compute_dtype fp16 = the reference's autocast dtype with its GradScaler (train.py:108,317-324): the loss decreases, the
    checkpoint holds scaler.pt with GradScaler's keys (train.py:403-404), a restart loads it (train.py:207-209) unless
    --reset_scaler, and --overwrite_dropout on a restart really changes the model's rate."""
    import train
    argv = ["--conditioning", "continuous_concat", "--n_layer", "2", "--d_model", "128", "--n_head", "2", "--d_inner", "256",
            "--d_condition", "32", "--tgt_len", "128", "--batch_size", "4", "--lr", "1e-3", "--max_step", "30", "--log_step", "10",
            "--eval_step", "1000", "--work_dir", str(tmp_path), "--dropout", "0.1", "--seed", "1", "--compute_dtype", "fp16",
            "--accumulate_step", "2"]
    train.main(argv)
    out = capsys.readouterr().out
    assert "dtype = fp16" in out
    losses = [float(l.split("| loss")[1].split("|")[0]) for l in out.splitlines() if "| loss" in l]
    assert len(losses) == 3 and losses[-1] < losses[0] - 0.05, losses
    run = sorted(os.listdir(tmp_path))[0]
    sd = torch.load(tmp_path / run / "scaler.pt")
    assert set(sd) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert sd["scale"] in (65536.0, 32768.0, 16384.0) and sd["growth_interval"] == 2000
    assert torch.load(tmp_path / run / "optimizer.pt")["step"] == 30 - {65536.0: 0, 32768.0: 1, 16384.0: 2}[sd["scale"]]
    # a hand-edited scale must come back on restart, and not with --reset_scaler
    sd["scale"], sd["_growth_tracker"] = 4096.0, 7
    torch.save(sd, tmp_path / run / "scaler.pt")
    rest = ["--conditioning", "continuous_concat", "--n_layer", "2", "--d_model", "128", "--n_head", "2", "--d_inner", "256",
            "--d_condition", "32", "--tgt_len", "128", "--batch_size", "4", "--lr", "1e-3", "--log_step", "5", "--eval_step", "1000",
            "--work_dir", str(tmp_path), "--restart_dir", run, "--max_step", "35", "--compute_dtype", "fp16", "--seed", "1"]
    train.main(rest + ["--dropout", "0.3", "--overwrite_dropout"])
    out = capsys.readouterr().out
    assert "Dropout rate changed to 0.3" in out and "step       35" in out
    runs = sorted(os.listdir(tmp_path))
    sd2 = torch.load(tmp_path / runs[-1] / "scaler.pt")
    assert sd2["scale"] == 4096.0 and sd2["_growth_tracker"] == 7 + 5, sd2
    train.main(rest + ["--reset_scaler"])
    capsys.readouterr()
    runs3 = [r for r in sorted(os.listdir(tmp_path)) if r not in runs]
    assert torch.load(tmp_path / runs3[-1] / "scaler.pt")["scale"] == 65536.0


def test_train_cli_grad_accumulation_matches_big_batch():
    """accumulate_step=2 with batch 2 == one step with the same 4 sequences (dropout 0, f32 tier)."""
    import train
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW
    a = train.parse_args(["--conditioning", "continuous_concat", "--n_layer", "1", "--d_model", "128", "--n_head", "2",
                          "--d_inner", "256", "--d_condition", "32", "--tgt_len", "64", "--dropout", "0"])
    cfg = dict(vars(a), vocab_size=1007, compute_dtype="fp32")
    torch.manual_seed(0)
    m1, _ = build_model(dict(cfg))
    m1 = m1.cuda().train()
    torch.manual_seed(0)
    m2, _ = build_model(dict(cfg))
    m2 = m2.cuda().train()
    assert torch.equal(m1.flat_params, m2.flat_params)
    xa = train.synthetic_batch(a, 1007, 2, 64, 1, "cuda")
    xb = train.synthetic_batch(a, 1007, 2, 64, 2, "cuda")
    m1.loss_and_backward(*xa, grad_scale=0.5)
    m1.loss_and_backward(*xb, grad_scale=0.5)
    big = tuple(torch.cat([u, v], 0) for u, v in zip(xa, xb))
    m2.loss_and_backward(*big)
    err = float((m1.flat_grads - m2.flat_grads).norm() / m2.flat_grads.norm())
    assert err < 1e-5, err
    FusedAdamW(m1, lr=1e-3).step()
    FusedAdamW(m2, lr=1e-3).step()
    assert float((m1.flat_params - m2.flat_params).abs().max()) < 1e-5


def _make_song_collection(root, n_songs=12, seed=0):
    """A tiny collection in the reference's on-disk format: <root>/songs/*.pt ({"bars": [int16 [n, 2]]}), <root>/maps.pt
    and a feature table with the columns data/preprocess_features.py reads."""
    import numpy as np
    from midiemo import data as D, vocab
    rng = np.random.default_rng(seed)
    maps = vocab.get_maps()
    maps["transposable_event_inds"] = D.transposable_event_inds(maps)
    os.makedirs(os.path.join(root, "songs"))
    torch.save(maps, os.path.join(root, "maps.pt"))
    ev = maps["event2idx"]
    ins = ["DRUMS", "GUITAR", "BASS", "PIANO", "STRINGS"]
    rows = []
    for s in range(n_songs):
        bars = []
        for _ in range(24):
            r = []
            for _ in range(40):
                if rng.random() < 0.3:
                    r.append((ev["TIMESHIFT"], int(rng.integers(1, 126)) * 8))
                else:
                    r.append((ev["%s_%s" % ("ON" if rng.random() < 0.5 else "OFF", ins[int(rng.integers(0, 5))])],
                              int(rng.integers(30, 100))))
            bars.append(torch.tensor(r, dtype=torch.int16))
        torch.save({"bars": bars}, os.path.join(root, "songs", "s%02d.pt" % s))
        rows.append("s%02d,%.4f,%.4f,5,True" % (s, 0.1 + 0.8 * rng.random(), 1.0 + rng.random()))
    with open(os.path.join(root, "features.csv"), "w") as fh:
        fh.write("file,valence,note_density_per_instrument,n_instruments,is_matched\n" + "\n".join(rows) + "\n")
    return os.path.join(root, "songs"), os.path.join(root, "features.csv")


@pytest.mark.parametrize("mode", ["continuous_concat", "discrete_token", "continuous_token"])
def test_train_cli_real_data_path(tmp_path, capsys, mode):
    """--feature_file switches train.py to the reference's data path (feature table -> Loader -> filter_collate)."""
    import train
    folder, csv_file = _make_song_collection(str(tmp_path / "lpd"))
    argv = ["--conditioning", mode, "--n_layer", "1", "--d_model", "128", "--n_head", "2", "--d_inner", "256",
            "--d_condition", "32", "--tgt_len", "128", "--batch_size", "4", "--lr", "1e-3", "--max_step", "12", "--log_step", "6",
            "--eval_step", "12", "--max_eval_step", "2", "--work_dir", str(tmp_path / "out"), "--seed", "3", "--num_workers", "0",
            "--data_folder", folder, "--feature_file", csv_file]
    train.main(argv)
    out = capsys.readouterr().out
    assert "Data loader lengths" in out
    losses = [float(l.split("| loss")[1].split("|")[0]) for l in out.splitlines() if "| loss" in l]
    assert len(losses) == 2 and all(0 < v < 8 for v in losses) and losses[1] < losses[0], losses
    assert "valid loss" in out
    run = os.listdir(tmp_path / "out")[0]
    maps = torch.load(tmp_path / "out" / run / "mappings.pt", weights_only=False)
    assert len(maps["tuple2idx"]) == (1017 if mode == "discrete_token" else 1007)      # 12 songs cover all 2 x 5 bins


def test_train_cli_regression(tmp_path, capsys):
    """--regression trains the evaluation model (8 layers as in config.py:128-130) with the L1 loss and logs the
    reference's val_l1_v / val_l1_a columns."""
    import csv
    import train
    argv = ["--regression", "--conditioning", "none", "--d_model", "128", "--n_head", "2", "--d_inner", "256",
            "--tgt_len", "96", "--batch_size", "8", "--lr", "2e-4", "--max_step", "40", "--log_step", "20", "--eval_step", "40",
            "--work_dir", str(tmp_path), "--dropout", "0.0", "--seed", "2"]
    train.main(argv)
    out = capsys.readouterr().out
    assert "Using 8 layers for regression" in out
    losses = [float(l.split("| loss")[1].split("|")[0]) for l in out.splitlines() if "| loss" in l]
    assert len(losses) == 2 and all(0 < v < 1.2 for v in losses), losses
    assert "l1_v" in out and "l1_a" in out
    run = os.listdir(tmp_path)[0]
    rows = list(csv.DictReader(open(tmp_path / run / "performance.csv")))
    assert 0 < float(rows[-1]["val_l1_v"]) < 1.5 and 0 < float(rows[-1]["val_l1_a"]) < 1.5
    sd = torch.load(tmp_path / run / "model.pt")
    assert "fc.0.weight" in sd and sd["fc.0.weight"].shape == (2, 128) and "enc_layers.7.rga.E" in sd
    assert sd["embedding.weight"].shape[0] == 1008                      # 1007 tokens + <CLS>


def test_train_cli_exhaustive_eval(tmp_path, capsys):
    """--exhaustive_eval: restore a checkpoint, evaluate every chunk of every test song (LoaderExhaustive) and exit."""
    import train
    folder, csv_file = _make_song_collection(str(tmp_path / "lpd"), n_songs=40)
    base = ["--conditioning", "continuous_concat", "--n_layer", "1", "--d_model", "128", "--n_head", "2", "--d_inner", "256",
            "--d_condition", "32", "--tgt_len", "128", "--batch_size", "4", "--lr", "1e-3", "--work_dir", str(tmp_path / "out"),
            "--seed", "3", "--num_workers", "0", "--feature_file", csv_file]
    train.main(base + ["--data_folder", folder, "--max_step", "6", "--log_step", "6", "--eval_step", "1000"])
    run = os.listdir(tmp_path / "out")[0]
    capsys.readouterr()
    # the exhaustive loader's own convention: <root>/maps.pt and <root>/lpd_5_full_transposable/*.pt
    root = tmp_path / "ex"
    root.mkdir()
    os.symlink(folder, str(root / "lpd_5_full_transposable"))
    os.symlink(os.path.join(os.path.dirname(folder), "maps.pt"), str(root / "maps.pt"))
    train.main(base + ["--data_folder", str(root), "--exhaustive_eval", "--restart_dir", run])
    out = capsys.readouterr().out
    line = [l for l in out.splitlines() if l.startswith("Exhaustive evaluation")]
    assert len(line) == 1 and "top1" in line[0] and "| step" not in out, out
    assert 0 < float(line[0].split("Loss:")[1].split(",")[0]) < 8


def test_restart_writes_a_fresh_dir_and_accepts_reference_optimizer_state(tmp_path, capsys):
    """a restart must not overwrite the checkpoint it resumes from; stats.pt and optimizer.pt are
    restored independently; an optimizer.pt in the REFERENCE's format (torch.optim.Adam state_dict, train.py:403) is
    converted into the fused optimiser's flat moments; a checkpoint built for another vocabulary is refused."""
    import time
    import train
    from midiemo.models.build_model import build_model
    from midiemo.optim import FusedAdamW
    argv = ["--conditioning", "none", "--n_layer", "1", "--d_model", "64", "--n_head", "2", "--d_inner", "128",
            "--tgt_len", "64", "--batch_size", "2", "--lr", "1e-3", "--log_step", "5", "--eval_step", "1000",
            "--gen_step", "1000", "--work_dir", str(tmp_path), "--dropout", "0.0", "--seed", "1"]
    train.main(argv + ["--max_step", "5"])
    run = os.listdir(tmp_path)[0]
    before = {f: os.path.getmtime(tmp_path / run / f) for f in os.listdir(tmp_path / run)}
    # replace optimizer.pt by a reference-format torch.optim.Adam state_dict with recognisable moments
    cfg = torch.load(tmp_path / run / "model_config.pt")
    model, _ = build_model(None, load_config_dict=cfg)
    tadam = torch.optim.Adam(model.parameters(), lr=3e-4)
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, 0.01 * (i + 1))
    tadam.step()
    torch.save(tadam.state_dict(), tmp_path / run / "optimizer.pt")
    model = model.cuda()
    opt = FusedAdamW(model)
    opt.load_state_dict(torch.load(tmp_path / run / "optimizer.pt", map_location="cuda"))
    assert opt.step_count == 1 and opt.param_groups[0]["lr"] == 3e-4
    for i, (n, p) in enumerate(model.named_parameters()):
        assert torch.allclose(model._pview(opt.m, n), torch.full_like(p, 0.1 * 0.01 * (i + 1)), rtol=1e-5), n
        assert torch.allclose(model._pview(opt.v, n), torch.full_like(p, 0.001 * (0.01 * (i + 1)) ** 2), rtol=1e-5), n
    before["optimizer.pt"] = os.path.getmtime(tmp_path / run / "optimizer.pt")
    capsys.readouterr()
    time.sleep(1.1)                                         # work_dir names have one-second resolution
    train.main(argv + ["--max_step", "10", "--restart_dir", run])
    out = capsys.readouterr().out
    assert "step       10" in out and "not restored" not in out
    runs = sorted(os.listdir(tmp_path))
    assert len(runs) == 2
    new = [r for r in runs if r != run][0]
    assert {f: os.path.getmtime(tmp_path / run / f) for f in before} == before          # the source checkpoint is untouched
    assert torch.load(tmp_path / new / "stats.pt")["step"] == 10
    import csv
    rows = list(csv.DictReader(open(tmp_path / new / "performance.csv")))
    assert [int(r["step"]) for r in rows] == [5, 10]                                     # the table continues
    # constant scheduler: the restored lr (3e-4 from the torch state) is kept, --lr is not re-imposed
    assert float(rows[-1]["lr"]) == pytest.approx(3e-4)
    with pytest.raises(SystemExit, match="disagrees"):
        train.main(argv[:1] + ["discrete_token"] + argv[2:] + ["--max_step", "12", "--restart_dir", run])


def test_in_training_sample_generation(tmp_path, capsys):
    """--gen_step (train.py:335-373): every gen_step steps rank 0 generates the four fixed conditions into
    <work_dir>/generations/training with the KV-cached decoder, then training continues."""
    import train
    argv = ["--conditioning", "continuous_concat", "--n_layer", "1", "--d_model", "64", "--n_head", "2", "--d_inner", "128",
            "--d_condition", "16", "--tgt_len", "64", "--batch_size", "2", "--lr", "1e-3", "--max_step", "6", "--log_step", "3",
            "--eval_step", "1000", "--gen_step", "3", "--gen_len", "24", "--work_dir", str(tmp_path), "--seed", "2"]
    train.main(argv)
    out = capsys.readouterr().out
    assert "step        6" in out
    run = os.listdir(tmp_path)[0]
    gen = tmp_path / run / "generations" / "training"
    files = os.listdir(gen) if os.path.isdir(gen) else []
    mids = [f for f in files if f.endswith(".mid")]
    # 2 generation rounds (steps 3 and 6) x 4 conditions; a sample without any instrument is reported, not saved
    assert len(mids) + out.count("not saving") == 8, (files, out[-500:])
    assert all(f.startswith(("3_", "6_")) for f in mids) and mids, mids
