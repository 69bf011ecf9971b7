"""Kernel microbenchmarks at the headline shapes (run under rocprofv3 --kernel-trace and
summarise with tools/rocpd_stats.py, or standalone: prints HIP-event timings)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "midi-emotion_amd"))
import torch  # noqa: E402
from midiemo import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="attn,gemm")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--L", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="", help="comma list of GEMM shape tags to run (default all)")
    a = ap.parse_args()
    dev, dt = "cuda", torch.bfloat16
    B, L, H, dh, M, d, di = a.B, a.L, 8, 64, 2048, 512, 2048
    T = B * L
    torch.manual_seed(0)
    if "attn" in a.what:
        Lp = ((L + 31) // 32) * 32
        qkv = torch.randn(B, L, 3, H, dh, device=dev).to(dt)
        E = torch.randn(M, dh, device=dev).to(dt)
        Epk = ops.rga_pack_rel(E)
        out = torch.empty(B, L, H, dh, device=dev, dtype=dt)
        lse = torch.empty(B, H, L, device=dev)
        dout = torch.randn(B, L, H, dh, device=dev).to(dt)
        dqkv = torch.empty_like(qkv)
        dE = torch.zeros(M, dh, device=dev)
        delta = torch.empty(B, H, L, device=dev)
        PT, MT = ops.rga_saved_buffers(B, H, L, dt, dev)
        dST = ops.rga_bwd_workspace(B, H, L, dt, dev)
        kp = torch.zeros(B, L, dtype=torch.uint8, device=dev)
        flop = 3 * 2 * B * H * dh * L * (L + 1) / 2
        t = timeit(lambda: ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M), a.iters)
        print("rga_fwd (infer)  %9.1f us  %7.1f TF (causal-discounted 3 contractions)" % (t, flop / t / 1e6))
        t = timeit(lambda: ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT, MT=MT), a.iters)
        print("rga_fwd (train)  %9.1f us  %7.1f TF" % (t, flop / t / 1e6))
        t = timeit(lambda: ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dST, B, L, Lp, H, dh, M),
                   a.iters)
        print("rga_bwd (3 krn)  %9.1f us  %7.1f TF (2x fwd flops)" % (t, 2 * flop / t / 1e6))
    if "gemm" in a.what:
        for (M_, N_, K_, tag) in [(T, 1536, 512, "qkv"), (T, 512, 512, "proj"), (T, 2048, 512, "ffn1"),
                                   (T, 512, 2048, "ffn2"), (T, 1007, 512, "head")]:
            if a.only and tag not in a.only.split(","):
                continue
            A = torch.randn(M_, K_, device=dev).to(dt)
            Bm = torch.randn(N_, K_, device=dev).to(dt)
            ldc = (N_ + 63) // 64 * 64                      # the model pads the logits rows to 64 columns (16-byte row stores)
            C = torch.empty(M_, ldc, device=dev, dtype=dt)[:, :N_]
            bias = torch.randn(N_, device=dev)
            t = timeit(lambda: ops.gemm_nt(A, Bm, C, bias=bias), a.iters)
            print("gemm_nt %-5s M%d N%d K%d %9.1f us  %7.1f TF" % (tag, M_, N_, K_, t, 2.0 * M_ * N_ * K_ / t / 1e6))
            if "blaslt" in a.what:      # yardstick only: the vendor library on the same shape (no bias)
                Bt = Bm.t()
                Cb = torch.empty(M_, N_, device=dev, dtype=dt)
                t = timeit(lambda: torch.matmul(A, Bt, out=Cb), a.iters)
                print("   torch.matmul (hipBLASLt)      %9.1f us  %7.1f TF" % (t, 2.0 * M_ * N_ * K_ / t / 1e6))
        for (N_, K_, tag) in [(1536, 512, "dWqkv"), (512, 512, "dWo"), (2048, 512, "dW1"), (512, 2048, "dW2")]:
            if a.only and tag not in a.only.split(","):
                continue
            A = torch.randn(T, N_, device=dev).to(dt)
            X = torch.randn(T, K_, device=dev).to(dt)
            dW = torch.zeros(N_, K_, device=dev)
            db = torch.zeros(N_, device=dev)
            need = ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N_, K_, dt)
            ws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
            t = timeit(lambda: ops.gemm_tn_acc(A, X, dW, db, ws=ws), a.iters)
            print("gemm_tn %-5s T%d N%d K%d %9.1f us  %7.1f TF" % (tag, T, N_, K_, t, 2.0 * T * N_ * K_ / t / 1e6))
            if "blaslt" in a.what:
                At = A.t()
                Cw = torch.empty(N_, K_, device=dev, dtype=dt)
                t = timeit(lambda: torch.matmul(At, X, out=Cw), a.iters)
                print("   torch.matmul (hipBLASLt)      %9.1f us  %7.1f TF" % (t, 2.0 * T * N_ * K_ / t / 1e6))


if __name__ == "__main__":
    main()
