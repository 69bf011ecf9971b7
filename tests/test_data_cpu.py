"""Real-data input path (midiemo/data.py) against vectors captured from the imported reference
(oracle/make_loader_fixtures.py): feature-table splits, every Loader sample of seeded runs in all conditioning modes
(bit-exact token ids, conditions, targets, rejected samples), the None-filtering collate, and the vectorised token
helpers against their per-row definitions."""
import io
import json
import math
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
from midiemo import data as D      # noqa: E402
from midiemo import vocab          # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "loader_fixture.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(FIX, allow_pickle=False)


@pytest.fixture(scope="module")
def song_dir(fx, tmp_path_factory):
    root = tmp_path_factory.mktemp("lpd")
    folder = root / "songs"
    folder.mkdir()
    maps = vocab.get_maps()
    maps["transposable_event_inds"] = D.transposable_event_inds(maps)
    torch.save(maps, str(root / "maps.pt"))
    for i, name in enumerate(fx["song_names"].tolist()):
        flat, lens = torch.from_numpy(fx["bars_%02d" % i]), fx["barlen_%02d" % i].tolist()
        torch.save({"bars": list(torch.split(flat, lens))}, str(folder / (name + ".pt")))
    return str(folder)


def same(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (math.isnan(a) and math.isnan(b)) or a == pytest.approx(b, rel=1e-12, abs=1e-15)
    return a == b


def test_preprocess_features_matches_reference(fx, tmp_path):
    csv = tmp_path / "features.csv"
    csv.write_bytes(fx["features_csv"].tobytes())
    ref = json.loads(fx["preprocess_json"].tobytes().decode())
    assert set(ref) == {"bins5", "cont", "bins4_full", "uncond"}
    for tag, case in ref.items():
        got = D.preprocess_features(str(csv), **case["kw"])
        for split, want in zip(got, (case["train"], case["test"])):
            assert len(split) == len(want) and len(want) > 0, tag
            for g, w in zip(split, want):
                assert set(g) == set(w), (tag, g, w)
                for k in w:
                    gv = None if (isinstance(g[k], float) and math.isnan(g[k]) and w[k] is None) else g[k]
                    assert same(gv, w[k]), (tag, k, g, w)


def test_loader_samples_bit_exact_vs_reference(fx, song_dir):
    meta = json.loads(fx["meta_json"].tobytes().decode())
    meta = {k: v for k, v in meta.items() if not k.startswith("ex_")}
    assert len(meta) == 8
    n_rejected = 0
    for tag, m in meta.items():
        ds = D.Loader(song_dir, m["data"], **m["kw"])
        assert ds.get_vocab_len() == m["vocab"] and len(ds) == m["len"], tag        # the missing file is filtered out
        random.seed(123)
        np.random.seed(456)
        j = 0
        for rep in range(3):
            for idx in range(len(ds)):
                x, c, y = ds[idx]
                none = fx["%s_none_%03d" % (tag, j)].tolist()
                assert [x is None, c is None, y is None] == none, (tag, j)
                n_rejected += none[0]
                if x is not None:
                    assert x.dtype == torch.int64 and np.array_equal(x.numpy(), fx["%s_x_%03d" % (tag, j)]), (tag, j)
                    assert np.array_equal(c.numpy(), fx["%s_c_%03d" % (tag, j)], equal_nan=True), (tag, j)
                if y is not None:
                    assert np.array_equal(y.numpy(), fx["%s_y_%03d" % (tag, j)]), (tag, j)
                j += 1
        assert j == m["n"]
    assert n_rejected > 0                       # the fixture exercises the (None, None, None) path


def test_filter_collate_drops_rejected_samples(fx, song_dir):
    meta = json.loads(fx["meta_json"].tobytes().decode())["concat"]
    ds = D.Loader(song_dir, meta["data"], **meta["kw"])
    random.seed(123)
    np.random.seed(456)
    items = [ds[i] for i in range(len(ds))]
    assert any(it[0] is None for it in items)
    x, c, y = D.filter_collate(items[:10])
    assert np.array_equal(x.numpy(), fx["collate_x"]) and np.array_equal(y.numpy(), fx["collate_y"])
    assert np.array_equal(c.numpy(), fx["collate_c"], equal_nan=True)
    assert D.filter_collate([(None, None, None)] * 3) == [[], [], []]
    assert D.filter_collate([None, None]) == []
    out = D.filter_collate([{"a": 1, "b": np.float32(2.0)}, None, {"a": 3, "b": np.float32(4.0)}])
    assert out["a"].tolist() == [1, 3] and out["b"].dtype == torch.float32


def test_vectorised_token_helpers_match_their_definitions():
    maps = vocab.get_maps()
    tr = D.transposable_event_inds(maps)
    assert sorted(maps["idx2event"][i] for i in tr) == sorted(
        "%s_%s" % (o, i) for o in ("ON", "OFF") for i in ("GUITAR", "BASS", "PIANO", "STRINGS"))
    g = torch.Generator().manual_seed(3)
    ev = torch.randint(0, 11, (500,), generator=g)
    val = torch.where(ev == 10, torch.randint(1, 126, (500,), generator=g) * 8, torch.randint(21, 109, (500,), generator=g))
    x = torch.stack([ev, val], 1).to(torch.int16)
    for n in (-3, -1, 0, 2, 3):
        want = x.clone()
        for i in range(want.size(0)):                        # per-row definition (data_processing.py:224-230)
            if want[i, 0].item() in tr and 21 <= want[i, 1].item() + n <= 108:
                want[i, 1] += n
        assert torch.equal(D.transpose(x.clone(), n, tr), want)
    want = torch.tensor([maps["tuple2idx"][tuple(r.tolist())] for r in x], dtype=torch.int16)
    assert torch.equal(D.tensor_to_ind_tensor(x, maps["tuple2idx"]), want)
    with pytest.raises(KeyError):
        D.tensor_to_ind_tensor(torch.tensor([[10, 7]], dtype=torch.int16), maps["tuple2idx"])     # 7 ms is not a timeshift step
    for k in range(1, 6):
        names = ["DRUMS", "GUITAR", "BASS", "PIANO", "STRINGS"][:k]
        rows = [[maps["event2idx"]["ON_" + nm], 60] for nm in names] + [[maps["event2idx"]["TIMESHIFT"], 8]]
        assert D.count_instruments(torch.tensor(rows, dtype=torch.int16), maps) == k


def test_exhaustive_loader_bit_exact_vs_reference(fx, song_dir, tmp_path):
    """LoaderExhaustive (the --exhaustive_eval data path): every chunk of every song, all modes, against the reference."""
    meta = {k: v for k, v in json.loads(fx["meta_json"].tobytes().decode()).items() if k.startswith("ex_")}
    assert len(meta) == 6
    root = tmp_path / "ex"
    (root / "lpd_5_full_transposable").mkdir(parents=True)
    maps = vocab.get_maps()
    maps["transposable_event_inds"] = D.transposable_event_inds(maps)
    torch.save(maps, str(root / "maps.pt"))
    for name in fx["song_names"].tolist():
        os.symlink(os.path.join(song_dir, name + ".pt"), str(root / "lpd_5_full_transposable" / (name + ".pt")))
    for tag, m in meta.items():
        ds = D.LoaderExhaustive(str(root), m["data"], **m["kw"])
        assert len(ds) == m["n"] and ds.get_vocab_len() == m["vocab"], tag
        items = [ds[i] for i in range(len(ds))]
        assert np.array_equal(np.stack([x.numpy() for x, _, _ in items]), fx[tag + "_x"]), tag
        assert np.array_equal(np.stack([c.numpy() for _, c, _ in items]), fx[tag + "_c"], equal_nan=True), tag
        if tag + "_y" in fx.files:
            assert np.array_equal(np.stack([y.numpy() for _, _, y in items]), fx[tag + "_y"]), tag
        else:
            assert all(y is None for _, _, y in items)
