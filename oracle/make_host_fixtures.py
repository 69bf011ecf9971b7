"""Golden vectors for the host-side tail of generation -- runs ONLY in the build container.

  * token -> note lists: the reference's tuples_to_mid (data/data_processing_reverse.py:12-53) is run on seeded token
    streams with a recording stand-in for pretty_midi (the package is not installed here; only the constructor calls
    Instrument(program, is_drum, name) / Note(velocity, pitch, start, end) are captured -- that IS what the function
    computes; writing the .mid container is pretty_midi's job and is covered by midiemo.midi_writer's own read-back test);
  * the filtered sampling distribution: the reference's generate() (generate.py:92-189) is run with a stub model that
    returns seeded logits, torch.topk / torch.multinomial are spied on, and for every step the temperature-scaled,
    top-k / nucleus-filtered probabilities, their vocabulary ids, and the drawn ids are stored.

Writes tests/golden/f7_host.npz (data only).   PYTHONDONTWRITEBYTECODE=1 python oracle/make_host_fixtures.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
if not os.path.isdir(REF):
    raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")

# ---- recording stand-in for pretty_midi -------------------------------------------------------------------------------
pm = types.ModuleType("pretty_midi")


class Instrument:
    def __init__(self, program, is_drum=False, name=""):
        self.program, self.is_drum, self.name, self.notes = program, is_drum, name, []


class Note:
    def __init__(self, velocity, pitch, start, end):
        self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end


class PrettyMIDI:
    def __init__(self, *a, **k):
        self.instruments = []

    def write(self, path):
        pass


pm.Instrument, pm.Note, pm.PrettyMIDI = Instrument, Note, PrettyMIDI
sys.modules["pretty_midi"] = pm
from unittest.mock import MagicMock  # noqa: E402
sys.modules["pypianoroll"] = MagicMock()
sys.path.insert(0, REF)
from data.data_processing import get_maps as ref_get_maps                       # noqa: E402  (reference)
from data.data_processing_reverse import ind_tensor_to_mid, ind_tensor_to_str  # noqa: E402  (reference)
import generate as ref_generate                                                 # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden", "f7_host.npz")
rec = {}

# ---- (a) token streams -> note lists --------------------------------------------------------------------------------
maps = ref_get_maps()
V = len(maps["tuple2idx"])
rs = np.random.RandomState(7)
INSTR = ["DRUMS", "PIANO", "GUITAR", "BASS", "STRINGS"]
for si, n_tok in enumerate((60, 400)):
    # streams rich in ON -> TIMESHIFT -> OFF patterns: uniform ids almost never close a note
    ids = []
    open_notes = []
    t2i, ev = maps["tuple2idx"], maps["event2idx"]
    while len(ids) < n_tok:
        r = rs.rand()
        if r < 0.4:
            ins, pitch = INSTR[rs.randint(5)], int(rs.randint(30, 90))
            key = (ev["ON_" + ins], pitch)
            if key in t2i:
                ids.append(t2i[key]); open_notes.append((ins, pitch))
        elif r < 0.7:
            shifts = [k for k in t2i if isinstance(k, tuple) and maps["idx2event"][k[0]] == "TIMESHIFT"]
            ids.append(t2i[shifts[rs.randint(len(shifts))]])
        elif r < 0.95 and open_notes:
            ins, pitch = open_notes.pop(rs.randint(len(open_notes)))
            ids.append(t2i[(ev["OFF_" + ins], pitch)])
        else:
            ids.append(int(rs.randint(0, V)))                 # anything, specials and dangling OFFs included
    ids = np.array(ids[:n_tok], dtype=np.int64)
    mid = ind_tensor_to_mid(torch.from_numpy(ids), maps["idx2tuple"], maps["idx2event"])
    rec[f"midi{si}_ids"] = ids
    for tr in mid.instruments:
        rec[f"midi{si}_{tr.name}_meta"] = np.array([tr.program, int(tr.is_drum)], dtype=np.int64)
        rec[f"midi{si}_{tr.name}_notes"] = np.array([[n.velocity, n.pitch, n.start, n.end] for n in tr.notes], dtype=np.float64).reshape(-1, 4)
    rec[f"midi{si}_symbols_crc"] = np.array(sum((i + 1) * len(s) for i, s in enumerate(
        ind_tensor_to_str(torch.from_numpy(ids), maps["idx2tuple"], maps["idx2event"]))))
print("midi fixtures:", {k: v.shape for k, v in rec.items() if k.endswith("_notes")})

# ---- (b) sampling tail --------------------------------------------------------------------------------------------------


class StubModel(torch.nn.Module):
    """Returns seeded logits for the last position; everything before is ignored by generate()."""

    def __init__(self, logits):
        super().__init__()
        self.logits, self.calls = logits, 0

    def forward(self, x, cond):
        out = torch.zeros(x.shape[0], x.shape[1], self.logits.shape[-1])
        out[:, -1] = self.logits[self.calls]
        self.calls += 1
        return out


for tag, top_k, top_p in (("k0p07", -1, 0.7), ("k20p10", 20, 1.0), ("k50p09", 50, 0.9)):
    steps, B = 12, 4
    g = torch.Generator().manual_seed(100 + max(top_k, 0))
    logits = torch.randn(steps, B, V, generator=g) * 3
    logits[2, 1] *= 0.01                                      # nearly flat row
    logits[3:10, 2, 300] = 40.0                               # peaked row for 7 steps: one choice -> the repeat counter passes 3
                                                              # and the penalty max(0, log((n + 1) / 4) * coeff) becomes positive
    logits[6, 0, 17] = float("nan")
    cap = {"topk_vals": [], "topk_inds": [], "probs": [], "drawn": []}
    orig_topk, orig_mult = torch.topk, torch.multinomial

    def spy_topk(inp, k, *a, **kw):
        r = orig_topk(inp, k, *a, **kw)
        cap["topk_vals"].append(r[0].clone()); cap["topk_inds"].append(r[1].clone())
        return r

    def spy_mult(p, n, *a, **kw):
        r = orig_mult(p, n, *a, **kw)
        cap["probs"].append(p.clone()); cap["drawn"].append(r.clone())
        return r

    ref_generate.torch.topk, ref_generate.torch.multinomial = spy_topk, spy_mult
    torch.manual_seed(5)
    try:
        ref_generate.generate(StubModel(logits.clone()), maps, torch.device("cpu"), "/tmp/none", "continuous_concat",
                              continuous_conditions=[[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], max_input_len=64,
                              amp=False, gen_len=steps, temperatures=[1.2, 0.9], top_k=top_k, top_p=top_p, debug=True,
                              min_n_instruments=0, penalty_coeff=0.5)
    finally:
        ref_generate.torch.topk, ref_generate.torch.multinomial = orig_topk, orig_mult
    k_eff = cap["probs"][0].shape[1]
    rec[f"samp_{tag}_cfg"] = np.array([top_k, top_p, steps, B, k_eff], dtype=np.float64)
    rec[f"samp_{tag}_logits"] = logits.numpy()
    rec[f"samp_{tag}_scaled_topk"] = torch.stack(cap["topk_vals"]).numpy()        # log_softmax / temperature, sorted
    rec[f"samp_{tag}_inds"] = torch.stack(cap["topk_inds"]).numpy().astype(np.int32)
    rec[f"samp_{tag}_probs"] = torch.stack(cap["probs"]).numpy()
    rec[f"samp_{tag}_drawn"] = torch.stack(cap["drawn"]).squeeze(-1).numpy().astype(np.int32)
    # effective per-row temperature of every step (note / rest temperature + repeat penalty, generate.py:138-163):
    # scaled = log_softmax(clean logits) / temp  ->  temp = (2nd largest log-prob) / (2nd largest scaled value): the largest
    # log-prob of a peaked row is 0
    clean = logits.clone()
    clean[clean != clean] = 0
    clean[:, :, [maps["tuple2idx"][s_] for s_ in maps["tuple2idx"] if isinstance(s_, str) and s_[0] == "<"]] = -float("inf")
    lsm2 = torch.log_softmax(clean, -1).topk(2, dim=-1).values[:, :, 1]
    rec[f"samp_{tag}_temp"] = (lsm2 / torch.stack(cap["topk_vals"])[:, :, 1]).numpy().astype(np.float32)
    rec[f"samp_{tag}_tokens"] = torch.stack([i_.gather(1, d_) for i_, d_ in zip(cap["topk_inds"], cap["drawn"])]).squeeze(-1).numpy().astype(np.int32)
    print(tag, "k_eff", k_eff, "choices/row", (rec[f"samp_{tag}_probs"] > 0).sum(-1).mean())

rec["special_ids"] = np.array(sorted(maps["tuple2idx"][s] for s in maps["tuple2idx"] if isinstance(s, str) and s[0] == "<"), dtype=np.int32)
np.savez_compressed(OUT, **rec)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
