"""Golden vectors for the real-data input path -- runs ONLY in the build container, where the
reference checkout is mounted read-only at /root/reference.

Builds a small synthetic song collection in the reference's on-disk format (per-song `.pt` with a list of int16 bar
tensors [n, 2] = (event index, value), `maps.pt` next to the folder) and a synthetic feature table, runs the
reference's own preprocess_features / Loader / filter_collate on them with seeded python / numpy generators and stores
inputs and outputs as data in tests/golden/loader_fixture.npz.  Nothing of the reference (source, bytecode) is copied.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_loader_fixtures.py
"""
import json
import os
import random
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
if not os.path.isdir(REF):
    raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
sys.modules["pretty_midi"] = MagicMock()
sys.modules["pypianoroll"] = MagicMock()
six = types.ModuleType("torch._six")                      # removed from torch 2: collate.py only needs string_classes
six.string_classes = (str, bytes)
sys.modules["torch._six"] = six
sys.path.insert(0, REF)
from data.data_processing import get_maps as ref_get_maps         # noqa: E402  (reference)
from data.loader import Loader as RefLoader                        # noqa: E402  (reference)
sys.modules.setdefault("tqdm", __import__("tqdm"))
from data.loader_exhaustive import LoaderExhaustive as RefExhaustive   # noqa: E402  (reference)
from data.collate import filter_collate as ref_collate             # noqa: E402  (reference)
from data.preprocess_features import preprocess_features as ref_preprocess   # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden", "loader_fixture.npz")


def make_song(rng, maps, n_bars, instruments):
    """Random but well-formed bars: ON/OFF events of the chosen instruments with TIMESHIFTs between them."""
    ev = maps["event2idx"]
    bars = []
    for _ in range(n_bars):
        rows = []
        for _ in range(int(rng.integers(6, 40))):
            if rng.random() < 0.35:
                rows.append((ev["TIMESHIFT"], int(rng.integers(1, 126)) * 8))
            else:
                ins = instruments[int(rng.integers(0, len(instruments)))]
                rows.append((ev["%s_%s" % ("ON" if rng.random() < 0.5 else "OFF", ins)], int(rng.integers(21, 109))))
        bars.append(torch.tensor(rows, dtype=torch.int16))
    return bars


def main():
    rng = np.random.default_rng(7)
    maps = ref_get_maps()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        folder = os.path.join(tmp, "songs")
        os.makedirs(folder)
        torch.save(maps, os.path.join(tmp, "maps.pt"))
        all_ins = ["DRUMS", "GUITAR", "BASS", "PIANO", "STRINGS"]
        names, nbars = [], []
        for i in range(10):
            k = [5, 4, 3, 2, 5, 3, 4, 5, 1, 3][i]           # songs 3 and 8 have too few instruments: rejected samples
            bars = make_song(rng, maps, int(rng.integers(3, 30)), all_ins[:k])
            name = "song_%02d" % i
            torch.save({"bars": bars}, os.path.join(folder, name + ".pt"))
            names.append(name); nbars.append(len(bars))
            out["bars_%02d" % i] = np.concatenate([b.numpy() for b in bars], 0)
            out["barlen_%02d" % i] = np.array([b.shape[0] for b in bars], np.int32)
        # ---- feature table
        import pandas as pd
        n = 60
        df = pd.DataFrame({
            "file": ["f_%03d" % i for i in rng.permutation(n)],
            "valence": np.where(rng.random(n) < 0.25, np.nan, rng.random(n)),
            "note_density_per_instrument": rng.gamma(2.0, 1.5, n),
            "n_instruments": rng.integers(1, 6, n),
            "tempo": rng.random(n) * 100,
        })
        df.loc[3, "valence"] = 0.0                              # dropped: zero valence
        df.loc[5, "note_density_per_instrument"] = 80.0         # dropped: outlier
        df["is_matched"] = ~df["valence"].isna()
        csv = os.path.join(tmp, "features.csv")
        df.to_csv(csv, index=False)
        out["features_csv"] = np.frombuffer(open(csv, "rb").read(), dtype=np.uint8)
        pre = {}
        for tag, kw in {"bins5": dict(n_bins=5), "cont": dict(n_bins=None), "bins4_full": dict(n_bins=4, use_labeled_only=False),
                        "uncond": dict(n_bins=None, conditional=False)}.items():
            tr, te = ref_preprocess(csv, **kw)
            pre[tag] = {"kw": kw, "train": tr, "test": te}
        out["preprocess_json"] = np.frombuffer(json.dumps(pre, default=lambda o: None if o != o else o).encode(), dtype=np.uint8)

        # ---- loader cases
        val = ["<V-2>", "<V-1>", "<V0>", "<V1>", "<V2>"]
        aro = ["<A-2>", "<A-1>", "<A0>", "<A1>", "<A2>"]
        cases = {
            "none": dict(conditioning="none", input_len=64),
            "concat": dict(conditioning="continuous_concat", input_len=48),
            "ctoken": dict(conditioning="continuous_token", input_len=40),
            "dtoken": dict(conditioning="discrete_token", input_len=56),
            "dtoken_always": dict(conditioning="discrete_token", input_len=24, always_use_discrete_condition=True),
            "long": dict(conditioning="none", input_len=1024),
            "nostart_notranspose": dict(conditioning="continuous_concat", input_len=32, use_start_token=False, max_transpose=0,
                                        bar_start_prob=0.2, min_n_instruments=2),
            "regression": dict(conditioning="continuous_concat", input_len=32, regression=True),
        }
        meta = {}
        for tag, kw in cases.items():
            disc = kw["conditioning"] == "discrete_token"
            data = [{"file": nm, "valence": val[i % 5] if disc else float(np.round(-0.9 + 0.2 * i, 3)),
                     "arousal": aro[(2 * i) % 5] if disc else float(np.round(0.8 - 0.17 * i, 3))} for i, nm in enumerate(names)]
            data.append({"file": "missing_song", "valence": val[0] if disc else 0.0, "arousal": aro[0] if disc else 0.0})
            ds = RefLoader(folder, data, **kw)
            random.seed(123); np.random.seed(456)
            items = []
            for rep in range(3):
                for idx in range(len(ds)):
                    items.append(ds[idx])
            meta[tag] = {"kw": kw, "data": data, "n": len(items), "vocab": ds.get_vocab_len(), "len": len(ds)}
            for j, (x, c, y) in enumerate(items):
                out["%s_x_%03d" % (tag, j)] = np.zeros(0, np.int64) if x is None else x.numpy()
                out["%s_c_%03d" % (tag, j)] = np.zeros(0, np.float32) if c is None else c.numpy()
                out["%s_y_%03d" % (tag, j)] = np.zeros(0, np.int64) if y is None else y.numpy()
                out["%s_none_%03d" % (tag, j)] = np.array([x is None, c is None, y is None])
            if tag == "concat":
                b = ref_collate(items[:10])
                out["collate_x"], out["collate_c"], out["collate_y"] = b[0].numpy(), b[1].numpy(), b[2].numpy()
        # ---- exhaustive loader: its own path convention (maps.pt inside the folder, songs in lpd_5_full_transposable/)
        exroot = os.path.join(tmp, "ex")
        os.makedirs(os.path.join(exroot, "lpd_5_full_transposable"))
        torch.save(maps, os.path.join(exroot, "maps.pt"))
        for nm in names:
            os.symlink(os.path.join(folder, nm + ".pt"), os.path.join(exroot, "lpd_5_full_transposable", nm + ".pt"))
        ex_cases = {
            "ex_none": dict(conditioning="none", input_len=64),
            "ex_concat": dict(conditioning="continuous_concat", input_len=48),
            "ex_ctoken": dict(conditioning="continuous_token", input_len=40),
            "ex_dtoken": dict(conditioning="discrete_token", input_len=56),
            "ex_dtoken_always": dict(conditioning="discrete_token", input_len=24, always_use_discrete_condition=True),
            "ex_regression": dict(conditioning="none", input_len=32, regression=True),
        }
        for tag, kw in ex_cases.items():
            disc = kw["conditioning"] == "discrete_token"
            data = [{"file": nm, "valence": val[i % 5] if disc else float(np.round(-0.9 + 0.2 * i, 3)),
                     "arousal": aro[(2 * i) % 5] if disc else float(np.round(0.8 - 0.17 * i, 3))} for i, nm in enumerate(names)]
            ds = RefExhaustive(exroot, data, **kw)
            items = [ds[i] for i in range(len(ds))]
            meta[tag] = {"kw": kw, "data": data, "n": len(items), "vocab": ds.get_vocab_len()}
            out[tag + "_x"] = np.stack([x.numpy() for x, _, _ in items])
            out[tag + "_c"] = np.stack([c.numpy() for _, c, _ in items])
            if items[0][2] is not None:
                out[tag + "_y"] = np.stack([y.numpy() for _, _, y in items])
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        out["song_names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
