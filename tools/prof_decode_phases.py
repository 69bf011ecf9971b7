"""Phase table of the decode kernels: needs the tools-only build
    ONLY=me_decode tools/build_abl.sh decprof "-DME_DEC_PROF"
in which lane 0 of every block stamps s_memtime at its phase boundaries.  Runs eager KV-cached steps of the headline model
(B = 4) at context T0 and prints, per launch of ONE token, the median over the blocks of each phase (shader-clock cycles ->
microseconds with the clock measured from the first to the last stamp of the step against the host timer)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MIDIEMO_LIB", os.path.join(ROOT, "abl_tmp", "lib_decprof.so"))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd")); sys.path.insert(0, ROOT)
import torch
from midiemo import _lib, ops
from midiemo.decode import DecodeSession
from midiemo.models.build_model import build_model
T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = _lib.load()
L.me_debug_dec_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L.me_debug_dec_prof_read.restype = ctypes.c_int
torch.manual_seed(0)
model, _ = build_model(dict(vocab_size=1007, n_layer=6, n_head=8, d_model=512, d_inner=2048, d_condition=128,
                            conditioning="continuous_concat", dropout=0.1, compute_dtype="bf16"))
model = model.cuda().eval()
cond = torch.tensor([[-0.8, -0.8], [-0.8, 0.8], [0.8, -0.8], [0.8, 0.8]], device="cuda")
sess = DecodeSession(model, 4); sess.t = T0
tok = torch.full((4,), 5, dtype=torch.long, device="cuda")
SL, BL, ST = 256, 320, 8
buf = np.zeros((SL, BL, ST), dtype=np.uint64); kinds = np.zeros(SL, dtype=np.int32)
with torch.no_grad():
    for _ in range(6): sess.step(tok, cond)                       # warm-up
    L.me_debug_dec_prof_read(None, None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): sess.step(tok, cond)
    e1.record(); torch.cuda.synchronize()
n = L.me_debug_dec_prof_read(buf.ctypes.data_as(ctypes.c_void_p), kinds.ctypes.data_as(ctypes.c_void_p), 0)
per_tok = n // 8
print("launches recorded %d (%d per token, the greedy pick is not instrumented); 8 eager steps took %.1f us each" % (n, per_tok, e0.elapsed_time(e1) * 1e3 / 8))
names = {0 * 4 + 3: "LN -> head (f32 logits)", 0 * 4 + 2: "LN1 -> FFN_pre + ReLU", 2 * 4 + 1: "combine -> Wo + resid", 3 * 4 + 1: "FFN_suf + resid (K split)",
         100: "LN2 -> qkv -> attention splits", 101: "embed -> qkv -> attention splits"}
ph_gemv = ["issue weight loads", "prologue (input rows -> LDS, incl. its memory round trip)", "first FMAs (= weights landed)", "further chunks", "lane reduce (+ K-split LDS)"]
ph_attn = ["issue weight loads", "LayerNorm / embedding row (+ round trip)", "q (k, v) projection", "scores pass (K, E round trip)", "exp / P.V pass"]
last = slice((8 - 1) * per_tok, 8 * per_tok)                      # the last of the 8 tokens
tok_t0 = None
for s_ in range(last.start, last.stop):
    kind, nb = int(kinds[s_]) & 255, int(kinds[s_]) >> 8
    st = buf[s_, :min(nb, BL)].astype(np.int64)
    ok = st[:, 0] > 0
    st = st[ok]
    if tok_t0 is None: tok_t0 = st[:, 0].min()
    nph = 5
    d = np.diff(st[:, :nph + 1], axis=1).astype(np.float64)
    d[d < 0] = np.nan                                             # blocks that left early (empty splits, role blocks)
    med = np.nanmedian(d, axis=0)
    span = (np.nanmax(st[:, :nph + 1]) - st[:, 0].min())
    start_skew = st[:, 0].max() - st[:, 0].min()
    labels = ph_attn if kind >= 100 else ph_gemv
    print("%-34s blocks %3d  start +%6d cyc  skew %5d  span %6d | " % (names.get(kind, str(kind)), nb, st[:, 0].min() - tok_t0, start_skew, span) +
          "  ".join("%s %d" % (labels[i].split(" (")[0][:22], med[i]) for i in range(nph)))
print("(cycles of the shader clock; 100 cycles ~ 0.045 us at 2.2 GHz)")
