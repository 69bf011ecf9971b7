"""Interleaved timing of ablation variants of the ping-pong NT main loop (development aid; needs a library built with
-DME_NT_ABL: tools/build_abl.sh ntabl "-DME_NT_ABL").  usage: ab_nt_vars.py <lib> "ml:var,ml:var,..." [N K]..."""
import ctypes, os, shutil, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
from midiemo import _lib
src = sys.argv[1]
cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(3, len(sys.argv) - 1, 2)] or [(512, 2048), (1536, 512)]
tmp = tempfile.mkdtemp()
dev, dt = "cuda", torch.bfloat16
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def call(L, A, B, C, bias=None):
    M, K = A.shape; N = B.shape[0]
    rc = L.me_gemm_nt(ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), ptr(bias), None, 0, None, 0, M, N, K, 0, _lib.ME_BF16, st())
    assert rc == 0, rc
r = lambda *s: torch.randn(*s, device=dev).to(dt)
libs = []
for i, (ml, var) in enumerate(cfgs):
    dst = os.path.join(tmp, "lib_%d.so" % i)
    shutil.copy(src, dst)
    os.environ["MIDIEMO_NT_MAINLOOP"] = str(ml); os.environ["MIDIEMO_NT_VAR"] = str(var)
    L = ctypes.CDLL(dst)
    L.me_gemm_nt.argtypes = _lib.SIGNATURES["me_gemm_nt"]; L.me_gemm_nt.restype = ctypes.c_int
    call(L, r(256, 64), r(256, 64), torch.empty(256, 256, device=dev, dtype=dt))
    libs.append(L)
torch.cuda.synchronize()
M = 32768
ZERO = bool(os.environ.get("AB_ZERO"))
for (N, K) in shapes:
    A, B, C = r(M, K), r(N, K), torch.empty(M, N, device=dev, dtype=dt)
    if ZERO: A.zero_(); B.zero_()
    ts = [[] for _ in libs]
    for L in libs: call(L, A, B, C)
    torch.cuda.synchronize()
    for _ in range(10):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): call(L, A, B, C)
            e1.record(); torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1) / 8 * 1e3)
    print("N %4d K %4d: " % (N, K) + "  ".join("%d:%d %.1f" % (cfgs[i][0], cfgs[i][1], sorted(ts[i])[len(ts[i]) // 2]) for i in range(len(libs))), flush=True)
