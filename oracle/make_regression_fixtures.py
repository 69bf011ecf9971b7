"""Golden vectors for the evaluation model MusicRegression -- runs ONLY in the build container.
Imports the reference's models/music_regression.py, feeds it seeded weights / tokens and stores inputs + outputs in
tests/golden/f6_regression.npz (data only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_regression_fixtures.py
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
if not os.path.isdir(REF):
    raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
sys.modules["pretty_midi"] = MagicMock()
sys.modules["pypianoroll"] = MagicMock()
sys.path.insert(0, REF)
from models.music_regression import MusicRegression          # noqa: E402  (reference)
from models.build_model import build_model as ref_build      # noqa: E402  (reference)
from oracle import ref_model as O                             # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "f6_regression.npz")


def main():
    out = {}
    # tiny model, constructed directly (max_seq = 64 keeps E small), and the cfg the reference builds for --regression
    V, N, H, d, di, M = 1008, 2, 2, 64, 128, 64            # head dim 32 (the HIP kernels cover 32, 48, 64)
    torch.manual_seed(3)
    ref = MusicRegression(embedding_dim=d, d_inner=di, vocab_size=V, num_layer=N, num_head=H, max_seq=M, dropout=0.0,
                          pad_token=0, output_size=2).eval()
    shapes = O.regression_param_shapes(V, N, d, di, d // H, M)
    sd = ref.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes, "state_dict layout differs from the restatement"
    P = O.regression_seeded_params(shapes, 17)
    g = torch.Generator().manual_seed(18)
    ref.load_state_dict(P)
    out["cfg"] = np.array([V, N, H, d, di, M])
    out["seed"] = np.array([17])
    cfg = O.Cfg(V, N, H, d, di, max_seq=M)
    for L in (1, 7, 33, 64):
        tok = torch.randint(1, V, (3, L), generator=g)
        with torch.no_grad():
            y = ref(tok)
            mine = O.regression_forward(cfg, {k: v.double() for k, v in P.items()}, tok)
        err = float((mine - y.double()).abs().max())
        assert err < 2e-5, (L, err)
        out["tok_%d" % L] = tok.numpy()
        out["y_%d" % L] = y.numpy()
        print("L=%2d  oracle vs reference max abs %.2e" % (L, err))
    # build_model(regression=True) of the reference: which class / kwargs it produces
    args = dict(vocab_size=1008, n_layer=2, n_head=2, d_model=64, d_inner=128, dropout=0.0, d_condition=-1,
                conditioning="none", regression=True, output_size=2)
    try:
        m, _ = ref_build(dict(args))
        out["build_class"] = np.array([type(m).__name__])
        out["build_keys"] = np.array(sorted(m.state_dict().keys()))
    except Exception as e:                                   # noqa: BLE001
        out["build_class"] = np.array(["error: %r" % (e,)])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", out["build_class"])


if __name__ == "__main__":
    main()
