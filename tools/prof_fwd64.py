"""development aid: per-phase s_memtime sums of rga_fwd64_kernel (library built with -DME_PROF)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch
from midiemo import ops, _lib
lib = _lib.load()
B, L, H, dh, M = 32, 1024, 8, 64, 2048
dt, dev = torch.bfloat16, "cuda"
qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt)
E = torch.randn(M, dh, device=dev).to(dt)
Epk = ops.rga_pack_rel(E)
out = torch.empty(B, L, H, dh, device=dev, dtype=dt)
lse = torch.empty(B, H, L, device=dev)
kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
names = ["pad scan", "prologue", "main: sstore+gload+MFMA S/G", "main: E loads + ring trips + adds", "main: softmax",
         "main: P stores + PV", "main: barrier", "gen: work", "gen: barrier", "epilogue"]
buf = (ctypes.c_ulonglong * 16)()
for train in (False, True):
    for _ in range(3):
        ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT if train else None, MT=MT if train else None)
    torch.cuda.synchronize()
    lib.me_prof_read(buf, 1)
    ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT if train else None, MT=MT if train else None)
    torch.cuda.synchronize()
    lib.me_prof_read(buf, 1)
    n = buf[15]
    tot = sum(buf[i] for i in range(10))
    print("train" if train else "infer", "waves", n, "mean ticks per wave", tot / n)
    for i, nm in enumerate(names):
        print("  %-36s %10.1f ticks/wave  %5.1f %%" % (nm, buf[i] / n, 100.0 * buf[i] / tot))
