"""development aid: per-phase s_memtime sums of rga_bwd_q_kernel (library built with -DME_PROFQ: tools/abl64.sh profq "-DME_PROFQ ..." me_attn)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch
from midiemo import ops, _lib
lib = _lib.load()
B, L, H, dh, M = 32, 1024, 8, 64, 2048
dt, dev = torch.bfloat16, "cuda"
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt); E = torch.randn(M, dh, device=dev).to(dt); Epk = ops.rga_pack_rel(E)
out = torch.randn(B, L, H, dh, device=dev).to(dt); lse = torch.randn(B, H, L, device=dev).abs() + 5
dout = torch.randn(B, L, H, dh, device=dev).to(dt); dqkv = torch.empty_like(qkv); dE = torch.zeros(M, dh, device=dev)
delta = torch.empty(B, H, L, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
names = ["prologue (delta, ring zero, first tiles)", "E^T loads issued, V frags, dP MFMAs", "dS (P x (dP - delta)), P reload issued",
         "ring scatter + K^T.dS MFMAs", "ring read + E^T MFMAs (waits for E^T)", "dG^T tile: tr reads + stores", "K/V sstore + gload",
         "barrier", "epilogue (dq stores)", "-", "pro: dO/O/lse loads issued", "pro: ring zero", "pro: tile + P loads issued", "pro: delta (first wait)",
         "pro: tile 0 -> LDS + block barrier"]
buf = (ctypes.c_ulonglong * 16)()
run = lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, L, H, dh, M)
for _ in range(3): run()
torch.cuda.synchronize()
lib.me_profq_read(buf, 1)
run(); torch.cuda.synchronize()
lib.me_profq_read(buf, 1)
n = buf[15]; tot = sum(buf[i] for i in range(15))
print("waves", n, "mean ticks per wave", tot / n, "(s_memtime ticks: 100 MHz)")
for i, nm in enumerate(names):
    print("  %-46s %10.1f ticks/wave  %5.1f %%" % (nm, buf[i] / n, 100.0 * buf[i] / tot))
