"""development aid: attention backward / TN / NT GEMM under a CU-masked stream (hipExtStreamCreateWithCUMask): how do the memory-bound
and the MFMA-bound kernels scale with the number of CUs?  usage: bench_cumask.py <hex 32-bit pattern repeated over 8 words> ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "midi-emotion_amd"))
import torch
from midiemo import ops
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(pattern):
    if isinstance(pattern, int): pattern = [pattern] * 8
    words = (ctypes.c_uint32 * 8)(*pattern)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
def timeit(fn, st, iters=10, warm=3):
    with torch.cuda.stream(st):
        for _ in range(warm): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev, dt = "cuda", torch.bfloat16
B, L, H, dh, M = 32, 1024, 8, 64, 2048
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt); E = torch.randn(M, dh, device=dev).to(dt); Epk = ops.rga_pack_rel(E)
out = torch.randn(B, L, H, dh, device=dev).to(dt); lse = torch.randn(B, H, L, device=dev).abs() + 5
dout = torch.randn(B, L, H, dh, device=dev).to(dt); dqkv = torch.empty_like(qkv); dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
T = B * L
A = torch.randn(T, 512, device=dev).to(dt); W = torch.randn(2048, 512, device=dev).to(dt); C = torch.empty(T, 2048, device=dev, dtype=dt)
bias = torch.randn(2048, device=dev)
x = torch.randn(T, 512, device=dev).to(dt); a2 = torch.randn(T, 512, device=dev).to(dt); y = torch.empty_like(x); so = torch.empty_like(x)
stats = torch.empty(T, 2, device=dev); gamma = torch.randn(512, device=dev); beta = torch.randn(512, device=dev)
def parse(a):
    if a.startswith("low"):                      # lowN: the N lowest bits of the 256-bit mask
        n = int(a[3:]); return [(0xffffffff if n >= 32 * (i + 1) else ((1 << max(0, n - 32 * i)) - 1)) for i in range(8)]
    if a.startswith("w"):                        # wXXXXXXXX,...: explicit words
        return [int(x, 16) for x in a[1:].split(",")]
    return [int(a, 16)] * 8
for pat in [[0xffffffff] * 8] + [parse(a) for a in sys.argv[1:]]:
    st = masked_stream(pat)
    ncu = sum(bin(w).count("1") for w in pat)
    t_b = timeit(lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, L, H, dh, M), st)
    t_g = timeit(lambda: ops.gemm_nt(A, W, C, bias=bias), st)
    t_l = timeit(lambda: ops.resid_ln_fwd(x, a2, gamma, beta, y, so, stats, T, 512, 1e-5, 0.1, 123, 3), st)
    print("mask %s (%3d bits): rga_bwd %.1f us, gemm_nt ffn1 %.1f us, resid_ln_fwd %.1f us" % (",".join("%08x" % w for w in pat), ncu, t_b, t_g, t_l))
