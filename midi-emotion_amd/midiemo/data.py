"""Real-data input path of the reference: feature table -> splits, per-song bar files -> training
samples, None-filtering collate.  Host-side integer work, no GPU code; kept call-compatible so that a user who owns
the Lakh-Spotify files can point train.py at them.

Restates, with the reference's random-number call order preserved (python `random` and `numpy.random`, so a seeded
run draws the same crops as the reference):
  * preprocess_features  -- data/preprocess_features.py:4-106 (filter, IQR outliers, [-1, 1] scaling, quantile bins,
                            matched/unmatched split, last 5 % of the matched files as test split)
  * Loader               -- data/loader.py:15-195 (random bar window with at least `min_n_instruments`, transposition,
                            <START> / arbitrary offset, emotion-token prefix or continuous condition, trim, pad, shift)
  * LoaderExhaustive     -- data/loader_exhaustive.py:14-173 (whole songs in consecutive chunks, for --exhaustive_eval)
  * filter_collate       -- data/collate.py:37-82 (default collate that drops None samples)
  * transpose, tensor_to_ind_tensor, count_instruments -- data/data_processing.py:224-246, utils.py:143-148, done on
    whole tensors instead of python loops over tokens (same results: tests/test_data_cpu.py pins them to vectors
    captured from the imported reference).
The regression/<CLS> variant of the loader is included; the MusicRegression model itself is out of scope."""
import collections
import os
import random

import numpy as np
import torch

from . import vocab

MIN_PITCH, MAX_PITCH = 21, 108


# ----------------------------------------------------------------------------- token helpers
def transposable_event_inds(maps):
    """Event indices whose pitch may be shifted: every ON_/OFF_ event except the drums (data_processing.py:196-199)."""
    if "transposable_event_inds" in maps:
        return list(maps["transposable_event_inds"])
    return [i for s, i in maps["event2idx"].items() if s != "TIMESHIFT" and not s.endswith("_DRUMS")]


def transpose(x, n, transposable, min_pitch=MIN_PITCH, max_pitch=MAX_PITCH):
    """In place: rows (event, pitch) of transposable events move by n semitones when the result stays in range."""
    if n == 0 or x.numel() == 0:
        return x
    ev, val = x[:, 0].long(), x[:, 1].long()
    ok = torch.isin(ev, torch.as_tensor(list(transposable), dtype=torch.long)) & (val + n <= max_pitch) & (val + n >= min_pitch)
    x[:, 1] += ok.to(x.dtype) * n
    return x


class TupleIndex:
    """(event, value) -> token index as a table lookup (data_processing.py:231-246 does a dict lookup per row)."""

    def __init__(self, tuple2idx):
        pairs = [(k, v) for k, v in tuple2idx.items() if isinstance(k, tuple)]
        ne = 1 + max(k[0] for k, _ in pairs)
        nv = 1 + max(k[1] for k, _ in pairs)
        self.table = torch.full((ne, nv), -1, dtype=torch.int16)
        for (e, v), i in pairs:
            self.table[e, v] = i

    def __call__(self, x):
        ev, val = x[:, 0].long(), x[:, 1].long()
        bad = (ev < 0) | (ev >= self.table.shape[0]) | (val < 0) | (val >= self.table.shape[1])
        if bool(bad.any()):
            raise KeyError(tuple(x[int(bad.nonzero()[0])].tolist()))
        out = self.table[ev, val]
        if bool((out < 0).any()):
            raise KeyError(tuple(x[int((out < 0).nonzero()[0])].tolist()))
        return out


def tensor_to_ind_tensor(x, tuple2idx):
    return TupleIndex(tuple2idx)(x)


def count_instruments(bars, maps):
    """Number of distinct instruments among the note events of a [n, 2] token tensor (utils.py:143-148 on the
    string form: 'ON_PIANO_60' has three fields, 'TIMESHIFT_8' two)."""
    names = maps["idx2event"]
    seen = set()
    for e in torch.unique(bars[:, 0]).tolist():
        parts = names[int(e)].split("_")
        if len(parts) == 2:                     # ON_<INS> / OFF_<INS>; with the pitch appended that is 3 fields
            seen.add(parts[1])
    return len(seen)


# ----------------------------------------------------------------------------- dataset
class Loader:
    """Map-style dataset: idx -> (input [input_len] int64, condition [2] f32, target [input_len] int64 or None), or
    (None, None, None) when no window with enough instruments was found in `n_try` draws."""

    def __init__(self, data_folder, data, input_len, conditioning, save_input_dir=None, pad=True,
                 use_start_token=True, use_end_token=False, max_transpose=3, n_try=5,
                 bar_start_prob=0.5, debug=False, overfit=False, regression=False,
                 max_samples=None, min_n_instruments=3, use_cls_token=True,
                 always_use_discrete_condition=False):
        self.data_folder = data_folder
        self.bar_start_prob = bar_start_prob
        self.save_input_dir = save_input_dir
        self.input_len = input_len
        self.n_try = n_try
        self.min_n_instruments = min_n_instruments
        self.overfit = overfit
        self.one_sample = None
        self.transpose_options = list(range(-max_transpose, max_transpose + 1))
        self.conditioning = conditioning
        self.regression = regression
        self.use_cls_token = use_cls_token
        self.always_use_discrete_condition = always_use_discrete_condition
        self.pad_token = "<PAD>" if pad else None
        self.start_token = "<START>" if use_start_token else None
        self.end_token = "<END>" if use_end_token else None
        self.cls_token = "<CLS>"

        present = set(os.listdir(self.data_folder))
        self.data = [s for s in data if s["file"] + ".pt" in present]

        maps_root = data_folder + "_debug" if (debug or overfit) else data_folder      # loader.py:44-52
        maps_file = os.path.join(os.path.abspath(maps_root + "/.."), "maps.pt")
        self.maps = torch.load(maps_file, weights_only=False) if os.path.exists(maps_file) else vocab.get_maps()
        self.maps.setdefault("transposable_event_inds", transposable_event_inds(self.maps))

        extra = []
        if conditioning == "continuous_token":
            self.input_len -= 2                       # two condition vectors are prepended by the model
        elif conditioning == "discrete_token":
            extra = sorted({s[label] for s in self.data for label in ("valence", "arousal")})
        if regression and use_cls_token:
            extra.append(self.cls_token)
        self._base_syms = list(self.maps["idx2tuple"].values())
        self.set_extra_tokens(extra)
        if max_samples is not None and not debug and not overfit:
            self.data = self.data[:max_samples]
        self.n_bars = max(round(input_len / 256 * 4), 1)      # about 4x the needed bars; trimmed later

    def set_extra_tokens(self, extra):
        """Vocabulary = base tokens + `extra` (emotion-bin symbols, <CLS>), appended in the given order.  The reference
        derives `extra` from the songs of the split at hand (loader.py:58-75) and takes the model's vocabulary from the
        TEST split (train.py:76-80); callers that want both splits to agree on small collections pass the union."""
        syms = self._base_syms + list(extra)
        self.maps["idx2tuple"] = dict(enumerate(syms))
        self.maps["tuple2idx"] = {s: i for i, s in enumerate(syms)}
        self._index = TupleIndex(self.maps["tuple2idx"])

    def get_vocab_len(self):
        return len(self.maps["tuple2idx"])

    def get_maps(self):
        return self.maps

    def get_pad_idx(self):
        return self.maps["tuple2idx"][self.pad_token]

    def __len__(self):
        return len(self.data)

    def _tok(self, sym):
        return torch.tensor([self.maps["tuple2idx"][sym]], dtype=torch.int16)

    def __getitem__(self, idx):
        if self.overfit and self.one_sample is not None:
            return tuple(self.one_sample)
        item = torch.load(os.path.join(self.data_folder, self.data[idx]["file"] + ".pt"), weights_only=False)
        all_bars = item["bars"]

        bars, n_ins, tries = None, 0, 0
        while tries < self.n_try and n_ins < self.min_n_instruments:
            first = random.randint(0, max(0, len(all_bars) - self.n_bars - 1))
            window = all_bars[first:min(len(all_bars), first + self.n_bars)]
            if len(window):
                bars = torch.cat([torch.as_tensor(b) for b in window], dim=0)
                n_ins = count_instruments(bars, self.maps)
            else:
                n_ins = 0
            tries += 1
        if n_ins < self.min_n_instruments:
            return None, None, None

        if self.transpose_options:
            bars = transpose(bars, random.choice(self.transpose_options), self.maps["transposable_event_inds"])
        bars = self._index(bars)

        r = np.random.uniform()
        at_bar = not (r > self.bar_start_prob and bars.size(0) > self.input_len)
        if at_bar:
            if self.start_token is not None:
                bars = torch.cat((self._tok(self.start_token), bars), dim=0)
        else:
            start = np.random.randint(0, bars.size(0) - self.input_len)
            bars = bars[start:start + self.input_len + 1]
        if self.regression and self.use_cls_token:
            bars = torch.cat((self._tok(self.cls_token), bars), dim=0)

        condition = torch.tensor([np.nan, np.nan], dtype=torch.float32)
        if self.conditioning == "discrete_token" and (at_bar or self.always_use_discrete_condition):
            s = self.data[idx]
            bars = torch.cat((self._tok(s["valence"]), self._tok(s["arousal"]), bars), dim=0)
        elif self.conditioning in ("continuous_token", "continuous_concat") or self.regression:
            condition = torch.tensor([self.data[idx]["valence"], self.data[idx]["arousal"]], dtype=torch.float32)

        bars = bars[:self.input_len + 1]                              # + 1: the shifted target
        if self.pad_token is not None and bars.shape[0] < self.input_len + 1:
            bars = torch.nn.functional.pad(bars, (0, self.input_len + 1 - bars.shape[0]), value=self.get_pad_idx())
        bars = bars.long()
        input_ = bars[:-1]
        if self.regression:
            target = None
        else:
            target = bars[1:]
            if self.conditioning == "continuous_token":               # the model sequence is 2 positions longer
                target = torch.nn.functional.pad(target, (condition.size(0), 0), value=self.get_pad_idx())
        if self.overfit:
            self.one_sample = [input_, condition, target]
        return input_, condition, target


class LoaderExhaustive:
    """Every song of the split cut into consecutive chunks, no randomness (data/loader_exhaustive.py:14-173, used by
    `--exhaustive_eval`).  Restated as is, including its own path convention: `maps.pt` INSIDE `data_folder` and the song
    files in `data_folder/lpd_5_full_transposable/`; the discrete-condition tokens and <CLS> are outside the chunk
    length; a trailing partial chunk is dropped."""

    def __init__(self, data_folder, data, input_len, conditioning, save_input_dir=None, pad=True,
                 use_start_token=True, use_end_token=False, always_use_discrete_condition=False,
                 debug=False, overfit=False, regression=False, max_samples=None, use_cls_token=True):
        self.data_folder = data_folder
        self.save_input_dir = save_input_dir
        self.input_len = input_len
        self.overfit = overfit
        self.one_sample = None
        self.conditioning = conditioning
        self.regression = regression
        if debug or overfit:
            data_folder = data_folder + "_debug"
        self.maps = torch.load(os.path.join(data_folder, "maps.pt"), weights_only=False)
        self.pad_token = "<PAD>" if pad else None
        self.start_token = "<START>" if use_start_token else None
        self.end_token = "<END>" if use_end_token else None
        self.cls_token = "<CLS>"

        extra = []
        if conditioning == "continuous_token":
            self.input_len -= 2
        elif conditioning == "discrete_token":
            self.input_len -= 2
            extra = sorted({s[label] for s in data for label in ("valence", "arousal")})
        if regression and use_cls_token:
            extra.append(self.cls_token)
            self.input_len -= 1
        chunk_len = self.input_len if regression else self.input_len + 1        # + 1: the shifted target
        if extra:
            syms = list(self.maps["idx2tuple"].values()) + extra
            self.maps["idx2tuple"] = dict(enumerate(syms))
            self.maps["tuple2idx"] = {s: i for i, s in enumerate(syms)}
        if max_samples is not None and not debug and not overfit:
            data = data[:max_samples]
        index = TupleIndex(self.maps["tuple2idx"])
        t2i = self.maps["tuple2idx"]
        tok = lambda sym: torch.tensor([t2i[sym]], dtype=torch.int16)

        chunks = []
        for rec in data:
            item = torch.load(os.path.join(data_folder, "lpd_5_full_transposable", rec["file"] + ".pt"), weights_only=False)
            if conditioning in ("continuous_token", "continuous_concat") or regression:
                condition = torch.tensor([rec["valence"], rec["arousal"]], dtype=torch.float32)
            else:
                condition = torch.tensor([np.nan, np.nan], dtype=torch.float32)
            song = index(torch.cat([torch.as_tensor(b) for b in item["bars"]], 0))
            if self.start_token is not None:
                song = torch.cat((tok(self.start_token), song), 0)
            cond_tokens = None
            if conditioning == "discrete_token":
                cond_tokens = torch.cat((tok(rec["valence"]), tok(rec["arousal"])), 0)
                if not always_use_discrete_condition:
                    song = torch.cat((cond_tokens, song), 0)           # once, in front of the song
            parts = list(torch.split(song, chunk_len))
            if parts[-1].size(0) != chunk_len:
                parts.pop(-1)
            if regression and use_cls_token:
                parts = [torch.cat((tok(self.cls_token), x), 0) for x in parts]
            if conditioning == "discrete_token" and always_use_discrete_condition:
                parts = [torch.cat((cond_tokens, x), 0) for x in parts]     # in front of every chunk
            chunks += [(x, condition) for x in parts]
        self.data = chunks

    def get_vocab_len(self):
        return len(self.maps["tuple2idx"])

    def get_maps(self):
        return self.maps

    def get_pad_idx(self):
        return self.maps["tuple2idx"][self.pad_token]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        chunk, condition = self.data[idx]
        chunk = chunk.long()
        if self.regression:
            return chunk, condition, None
        target = chunk[1:]
        if self.conditioning == "continuous_token":
            target = torch.nn.functional.pad(target, (condition.size(0), 0), value=self.get_pad_idx())
        return chunk[:-1], condition, target


# ----------------------------------------------------------------------------- collate
def filter_collate(batch):
    """torch's classic default collate, minus the samples that are None (a Loader item is a tuple of three; a tuple
    whose first field is None is dropped as a whole by the recursion on its columns)."""
    if isinstance(batch, (list, tuple)):
        batch = [b for b in batch if b is not None]
    if not batch:
        return batch
    first = batch[0]
    if isinstance(first, torch.Tensor):
        return torch.stack(batch, 0)
    if type(first).__module__ == "numpy" and type(first).__name__ not in ("str_", "string_"):
        if isinstance(first, np.ndarray):
            if first.dtype.kind in "SaUO":
                raise TypeError("batch must contain tensors, numbers, dicts or lists; found %s" % first.dtype)
            return filter_collate([torch.from_numpy(b) for b in batch])
        if first.shape == ():
            return torch.as_tensor(np.asarray(batch))
    if isinstance(first, float):
        return torch.tensor(batch, dtype=torch.float64)
    if isinstance(first, int):
        return torch.tensor(batch)
    if isinstance(first, (str, bytes)):
        return batch
    if isinstance(first, collections.abc.Mapping):
        return {k: filter_collate([d[k] for d in batch]) for k in first}
    if isinstance(first, tuple) and hasattr(first, "_fields"):
        return type(first)(*(filter_collate(col) for col in zip(*batch)))
    if isinstance(first, collections.abc.Sequence):
        return [filter_collate(col) for col in zip(*batch)]
    raise TypeError("batch must contain tensors, numbers, dicts or lists; found %s" % type(first))


# ----------------------------------------------------------------------------- feature table
def emotion_token_labels(n_bins, label):
    return vocab.emotion_symbols(n_bins, label[0].upper())


def preprocess_features(feature_file, n_bins=None, min_n_instruments=3, test_ratio=0.05, outlier_range=1.5,
                        conditional=True, use_labeled_only=True):
    """CSV (file, valence, note_density_per_instrument, n_instruments, is_matched, ...) -> [train, test] lists of
    {"file", "valence", "arousal"} records."""
    import pandas as pd
    df = pd.read_csv(feature_file).rename(columns={"note_density_per_instrument": "arousal"})
    original_columns = list(df.columns)
    labels = ["valence", "arousal"]
    df = df[(df["n_instruments"] >= min_n_instruments) & (df["valence"] != 0)]

    drop = []
    for lab in labels:                                    # Tukey fences, both features measured before any row is dropped
        q1, q3 = df[lab].quantile(0.25), df[lab].quantile(0.75)
        lo, hi = q1 - outlier_range * (q3 - q1), q3 + outlier_range * (q3 - q1)
        drop += df.index[(df[lab] < lo) | (df[lab] > hi)].tolist()
    df = df.drop(drop)
    for lab in labels:
        lo, hi = df[lab].min(), df[lab].max()
        df[lab] = (df[lab] - lo) / (hi - lo) * 2 - 1

    if n_bins is not None:
        for lab in labels:
            names = emotion_token_labels(n_bins, lab) + [None]            # last slot: NaN (unlabelled) rows
            edges = [df[lab].quantile(q) for q in np.linspace(0, 1, n_bins + 1)]
            edges[-1] += 1e-6
            df[lab] = [names[i - 1] for i in np.digitize(df[lab].to_numpy(), edges)]
    else:
        df = df.where(pd.notnull(df), None)

    matched = df[df["is_matched"]].sort_values("file").reset_index(drop=True)
    unmatched = df[~df["is_matched"]]
    n_test = round(len(matched) * test_ratio)
    test = matched.loc[len(matched) - n_test:len(matched)]
    train = matched.loc[:len(matched) - n_test]            # label slice: inclusive, shares one row with `test` as in the reference
    if not use_labeled_only:
        train = pd.concat([train, unmatched]).sort_values("file").reset_index(drop=True)

    for lab in labels:
        test = test[~test[lab].isnull()]
        if use_labeled_only:
            train = train[~train[lab].isnull()]
    unused = [c for c in original_columns if c not in ("file", "valence", "arousal")]
    if not conditional:
        unused += ["valence", "arousal"]
    return [s.drop(columns=unused, errors="ignore").to_dict("records") for s in (train, test)]
