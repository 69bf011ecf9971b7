"""Same-process A/B of the two main loops of the 256-tile NT GEMM (MIDIEMO_NT_MAINLOOP = 0 register-staged single phase,
1 ping-pong + direct-to-LDS feed, 2 hand-scheduled 4-wave loop): private copies of the library, each initialised under its own setting; results must be
BIT-identical, timings are interleaved (round-robin, median) because the matrix pipe is power limited and the clocks drift."""
import ctypes, os, shutil, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "midi-emotion_amd"))
import torch
from midiemo import _lib
src = os.environ.get("MIDIEMO_LIB") or os.path.join(HERE, "..", "midi-emotion_amd", "midiemo", "libmidiemo_hip.so")
tmp = tempfile.mkdtemp()
libs = {}
MLS = tuple(int(x) for x in os.environ.get("AB_ML", "0,1,2,3").split(","))
for ml in MLS:
    dst = os.path.join(tmp, "lib_ml%d.so" % ml)
    shutil.copy(src, dst)
    os.environ["MIDIEMO_NT_MAINLOOP"] = str(ml)
    L = ctypes.CDLL(dst)
    L.me_gemm_nt.argtypes = _lib.SIGNATURES["me_gemm_nt"]
    L.me_gemm_nt.restype = ctypes.c_int
    libs[ml] = L
dev, dt = "cuda", torch.bfloat16
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def ptr(t): return ctypes.c_void_p(t.data_ptr()) if t is not None else None
def call(L, A, B, C, bias=None, add=None, gate=None, flags=0):
    M, K = A.shape; N = B.shape[0]
    rc = L.me_gemm_nt(ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), ptr(bias), ptr(add), add.stride(0) if add is not None else 0,
                      ptr(gate), gate.stride(0) if gate is not None else 0, M, N, K, flags, _lib.ME_BF16, st())
    assert rc == 0, rc
    if not warmed.get(id(L)):          # the first call of a copy reads the environment
        warmed[id(L)] = True
warmed = {}
r = lambda *s: torch.randn(*s, device=dev).to(dt)
# first call of each copy under its own setting
for ml in MLS:
    os.environ["MIDIEMO_NT_MAINLOOP"] = str(ml)
    call(libs[ml], r(256, 64), r(256, 64), torch.empty(256, 256, device=dev, dtype=dt))
torch.cuda.synchronize()

def check(M, N, K, what, f32=False):
    A, B = r(M, K), r(N, K)
    bias, addt, gatet = torch.randn(N, device=dev), r(M, N), r(M, N)
    kws = {"plain": {}, "bias": dict(bias=bias), "bias+relu": dict(bias=bias, flags=1), "add": dict(add=addt), "add+bias": dict(add=addt, bias=bias),
           "gate": dict(gate=gatet, flags=4)}[what]
    kws = dict(kws)
    if f32: kws["flags"] = kws.get("flags", 0) | 2
    outs = []
    for ml in MLS:
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else dt)
        call(libs[ml], A, B, C, **kws)
        outs.append(C)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0].view(torch.int32 if f32 else torch.int16), o.view(torch.int32 if f32 else torch.int16)) for o in outs[1:])
    ref = A.float() @ B.float().t()
    base = outs[0].float()
    print("check M%6d N%5d K%5d %-9s f32=%d  bit-identical=%s  finite=%s" % (M, N, K, what, f32, same, bool(all(torch.isfinite(o.float()).all() for o in outs))), flush=True)
    return same

def bench(M, N, K, what, rounds=12, iters=8):
    A, B, C = r(M, K), r(N, K), torch.empty(M, N, device=dev, dtype=dt)
    bias, addt, gatet = torch.randn(N, device=dev), r(M, N), r(M, N)
    kws = {"plain": {}, "bias": dict(bias=bias), "bias+relu": dict(bias=bias, flags=1), "add": dict(add=addt), "gate": dict(gate=gatet, flags=4)}[what]
    ts = {ml: [] for ml in MLS}
    for ml in MLS: call(libs[ml], A, B, C, **kws)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for ml in MLS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call(libs[ml], A, B, C, **kws)
            e1.record(); torch.cuda.synchronize()
            ts[ml].append(e0.elapsed_time(e1) / iters * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    print("bench M%6d N%5d K%5d %-9s  " % (M, N, K, what) + "  ".join("ml%d %.1f us (%+.1f %%)" % (ml, med(ts[ml]), 100 * (med(ts[ml]) / med(ts[MLS[0]]) - 1)) for ml in MLS) +
          "  best %.0f TF/s" % (2e-6 * M * N * K / min(med(ts[ml]) for ml in MLS)), flush=True)

ok = True
if "--nocheck" not in sys.argv:
    for (M, N, K) in ((256, 256, 64), (256, 256, 128), (256, 256, 512), (512, 512, 192), (300, 260, 64), (1000, 700, 320), (4096, 1536, 512), (32768, 512, 2048),
                      (32768, 1007, 512), (777, 1007, 512), (32768, 2048, 512)):
        for what in ("plain", "bias+relu", "add", "add+bias", "gate"):
            ok &= check(M, N, K, what)
        ok &= check(M, N, K, "bias", f32=True)
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH", flush=True)
if "--nobench" not in sys.argv:
    M = 32768
    for (N, K, whats) in ((512, 2048, ("plain", "bias", "add", "gate")), (2048, 512, ("plain", "bias+relu", "gate")), (1536, 512, ("bias",)), (512, 512, ("bias",)),
                          (512, 1536, ("add",)), (1007, 512, ("bias",))):
        for w in whats: bench(M, N, K, w)
    bench(16384, 512, 2048, "bias"); bench(16384, 1024, 1024, "bias"); bench(32768, 512, 1024, "bias")
