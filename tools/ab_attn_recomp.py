"""A/B of the query-owned attention backward kernel: probability tiles read back (default) vs rebuilt from q / k / E / lse
(me_rga_bwd_phases with phases = 1 and PT = MT = NULL: the FlashAttention-2 shape); both arms in one process, interleaved.
usage: python tools/ab_attn_recomp.py [B H L]      (under rocprofv3 --pmc: the two instantiations show up as two kernels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch  # noqa: E402
from midiemo import ops  # noqa: E402

B, H, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 8, 1024)
dh, M = 64, 2048
dt = torch.bfloat16
torch.manual_seed(0)
dev = "cuda"
qkv = (torch.randn(B, L, 3, H, dh, device=dev) * 0.7).to(dt)
E = torch.randn(M, dh, device=dev).to(dt)
Epk = ops.rga_pack_rel(E)
out = torch.empty(B, L, H, dh, dtype=dt, device=dev)
lse = torch.empty(B, H, L, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
ops.rga_fwd(qkv, Epk, None, out, lse, B, L, H, dh, M, PT=PT, MT=MT)
dout = torch.randn(B, L, H, dh, device=dev).to(dt)
dqkv = torch.zeros_like(qkv)
dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev)
dGT = ops.rga_bwd_workspace(B, H, L, dt, dev)
Lp = ((L + 31) // 32) * 32


def phase(bits, recomp=False):
    ops.check(ops.lib().me_rga_bwd_phases(qkv.data_ptr(), Epk.data_ptr(), None, out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dqkv.data_ptr(),
                                          dE.data_ptr(), delta.data_ptr(), None if recomp else PT.data_ptr(), None if recomp else MT.data_ptr(),
                                          dGT.data_ptr(), B, L, Lp, H, dh, M, 1, bits, ops._code(dt), torch.cuda.current_stream().cuda_stream), "phases")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


res = {"readback": [], "recompute": []}
for _ in range(3):
    res["readback"].append(timeit(lambda: phase(1)))
    res["recompute"].append(timeit(lambda: phase(1, True)))
tkv = timeit(lambda: phase(2))
te = timeit(lambda: phase(4))
tf_train = timeit(lambda: ops.rga_fwd(qkv, Epk, None, out, lse, B, L, H, dh, M, PT=PT, MT=MT))
tf_inf = timeit(lambda: ops.rga_fwd(qkv, Epk, None, out, lse, B, L, H, dh, M))
phase(1)
torch.cuda.synchronize()
dq0 = dqkv[:, :, 0].float().clone()
phase(1, True)
torch.cuda.synchronize()
dq1 = dqkv[:, :, 0].float()
print("B%d H%d L%d (warm back-to-back launches): bwd-q reading the saved tiles back %s us, rebuilding P %s us; kv %.1f us, dE %.1f us; "
      "fwd storing P %.1f us, fwd without %.1f us; dq recomputed vs read-back rel-L2 %.2e" %
      (B, H, L, " ".join("%.1f" % t for t in res["readback"]), " ".join("%.1f" % t for t in res["recompute"]), tkv, te, tf_train, tf_inf,
       float((dq1 - dq0).norm() / dq0.norm())))
