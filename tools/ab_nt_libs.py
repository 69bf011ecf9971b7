import ctypes, os, sys
sys.path.insert(0, "/root/repo/midi-emotion_amd"); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "midi-emotion_amd"))
import torch
from midiemo import _lib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ML = int(os.environ.get("AB_LIB_ML", "2"))
cfgs = [("old", "midi-emotion_amd/midiemo/libmidiemo_hip.so", 0)] + ([("4w", "midi-emotion_amd/midiemo/libmidiemo_hip.so", 2)] if ML == 2 else []) + [(n, "abl_tmp/lib_%s.so" % n, ML) for n in sys.argv[1:]]
import shutil, tempfile
tmp = tempfile.mkdtemp(); libs = []
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def call(L, A, B, C, bias=None, add=None, gate=None, flags=0):
    M, K = A.shape; N = B.shape[0]
    rc = L.me_gemm_nt(ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), ptr(bias), ptr(add), add.stride(0) if add is not None else 0, ptr(gate), gate.stride(0) if gate is not None else 0, M, N, K, flags, _lib.ME_BF16, st())
    assert rc == 0, rc
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
for i, (n, path, ml) in enumerate(cfgs):
    dst = os.path.join(tmp, "l%d.so" % i); shutil.copy(os.path.join(R, path), dst)
    os.environ["MIDIEMO_NT_MAINLOOP"] = str(ml)
    L = ctypes.CDLL(dst); L.me_gemm_nt.argtypes = _lib.SIGNATURES["me_gemm_nt"]; L.me_gemm_nt.restype = ctypes.c_int
    call(L, r(256, 128), r(256, 128), torch.empty(256, 256, device="cuda", dtype=torch.bfloat16)); libs.append(L)
torch.cuda.synchronize()
M = 32768
for (N, K, what) in ((512, 2048, "bias"), (512, 2048, "add"), (2048, 512, "bias"), (2048, 512, "gate"), (1536, 512, "bias"), (512, 512, "bias"), (512, 1536, "add")):
    A, B, C = r(M, K), r(N, K), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias, addt, gatet = torch.randn(N, device="cuda"), r(M, N), r(M, N)
    kw = {"bias": dict(bias=bias), "add": dict(add=addt), "gate": dict(gate=gatet, flags=4)}[what]
    ts = [[] for _ in libs]
    for L in libs: call(L, A, B, C, **kw)
    torch.cuda.synchronize()
    for _ in range(10):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): call(L, A, B, C, **kw)
            e1.record(); torch.cuda.synchronize(); ts[i].append(e0.elapsed_time(e1) / 8 * 1e3)
    print("N %4d K %4d %-5s " % (N, K, what) + "  ".join("%s %.1f" % (cfgs[i][0], sorted(ts[i])[5]) for i in range(len(libs))), flush=True)
