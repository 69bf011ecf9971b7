"""Kernel-level parity tests (GPU): every C-ABI entry point against the oracle /
plain fp64 torch math on the same seeded inputs.  f32 tier: tight tolerances;
bf16 tier: error relative to the fp64 result bounded by bf16 rounding of the
inputs (stated per test)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_model as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from midiemo import ops as _ops
    _ops.lib()
    return _ops


DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
DT16 = [torch.bfloat16, torch.float16]        # the two 16-bit storage tiers: same kernels, different MFMA opcode / conversions


def store_err(ref, dtype=torch.bfloat16):
    """rel-L2 error of rounding the exact (f64) result to the storage type ONCE -- what a torch bf16 op with f32
    accumulation commits on the same inputs, and the floor of any kernel that stores T."""
    r = torch.as_tensor(ref).detach().double().cpu()
    n = float(r.norm())
    return float((r.to(dtype).double() - r).norm() / n) if n > 1e-12 else 0.0


K_STORE = 2.0     # a bf16 result may sit one rounding away from the rounded exact value (f32 accumulation order, fused adds)


def tol(dtype, f32, *stored):
    """f32 tier: the given absolute bound.  bf16 tier: DERIVED, not picked -- K_STORE x the sum of the single-rounding errors
    of every tensor the kernel keeps in bf16 between the f64-exact inputs and the checked output (`stored`: their exact
    values; the inputs themselves are rounded before the reference is formed)."""
    if dtype == torch.float32:
        return f32
    assert stored, "bf16 bound needs the exact value(s) of the stored tensor(s)"
    # + the f32 accumulation of the products: 16-bit x 16-bit products are exact in f32, their sums are not (f16's 22-bit products
    # lose more per add than bf16's 16-bit ones: measured 2e-6 on a K = 512 projection with nothing stored in between)
    return K_STORE * sum(store_err(t, dtype) for t in stored) + (1e-6 if dtype == torch.bfloat16 else 5e-6)


def relerr(got, ref):
    """||got-ref|| / ||ref||; an exactly-zero reference is compared on an absolute 1e-4 scale."""
    got = got.double().cpu()
    ref = ref.double().cpu()
    n = float(ref.norm())
    if n < 1e-12:
        return float((got - ref).abs().max()) / 1e-4 * 1e-6
    return float((got - ref).norm() / n)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


# ------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 64), (77, 1007, 512), (3, 40, 8), (300, 130, 2048)])
def test_gemm_nt_plain_and_bias(ops, dtype, M, N, K):
    A = rnd(M, K, seed=1).to(dtype)
    B = rnd(N, K, seed=2).to(dtype)
    bias = rnd(N, seed=3).float()
    ref = A.double() @ B.double().t() + bias.double()
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    C = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(Ad, Bd, C, bias=bd)
    assert relerr(C, ref) < tol(dtype, 1e-5, ref), (relerr(C, ref))
    C32 = torch.full((M, N + 3), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_nt(Ad, Bd, C32, bias=bd, flags=ops.ME_EPI_OUT_F32)
    assert relerr(C32[:, :N], ref) < 1e-5 * (1 if dtype == torch.float32 else 1), relerr(C32[:, :N], ref)
    assert torch.isnan(C32[:, N:]).all()


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 1536, 512), (700, 1007, 512), (1024, 512, 2048), (260, 200, 128), (2050, 2048, 512)])
def test_gemm_nt_256_tile_path(ops, dtype, M, N, K):
    """16-bit shapes that take the 256x256 kernel (K % 64 == 0), incl. ragged M/N and all epilogues."""
    A = rnd(M, K, seed=31).to(dtype)
    B = rnd(N, K, seed=32).to(dtype)
    bias = rnd(N, seed=33).float()
    add = rnd(M, N, seed=34).to(dtype)
    gate = rnd(M, N, seed=35).to(dtype)
    base = A.double() @ B.double().t()
    Ad, Bd = A.to(DEV), B.to(DEV)
    ld = ((N + 15) // 16) * 16
    C = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(Ad, Bd, C, bias=bias.to(DEV), flags=ops.ME_EPI_RELU, N=N)
    assert relerr(C[:, :N], torch.relu(base + bias.double())) < tol(dtype, 0, torch.relu(base + bias.double()))
    assert torch.isnan(C[:, N:]).all()
    addp = torch.zeros(M, ld, dtype=dtype, device=DEV)
    addp[:, :N] = add.to(DEV)
    ops.gemm_nt(Ad, Bd, C, add=addp, N=N)
    assert relerr(C[:, :N], base + add.double()) < tol(dtype, 0, base + add.double())
    assert torch.isnan(C[:, N:]).all()
    # round 4: a 16-byte aligned residual operand is read row-contiguously and transposed into the accumulators' layout through
    # the staging buffer; the sum is still formed in f32 before the one rounding, so the result must be BIT-identical to the
    # element-wise path, which an 8-byte-aligned view of the same values selects
    widea = torch.zeros(M, ld + 16, dtype=dtype, device=DEV)
    widea[:, 4:4 + N] = add.to(DEV)
    add8 = widea[:, 4:4 + ld]
    assert add8.data_ptr() % 16 == 8
    Ca = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(Ad, Bd, Ca, add=add8, N=N)
    assert torch.equal(Ca[:, :N], C[:, :N]) and torch.isnan(Ca[:, N:]).all()
    gatep = torch.zeros(M, ld, dtype=dtype, device=DEV)
    gatep[:, :N] = gate.to(DEV)
    ops.gemm_nt(Ad, Bd, C, gate=gatep, flags=ops.ME_EPI_RELU_BWD, N=N)
    assert relerr(C[:, :N], base * (gate.double() > 0)) < tol(dtype, 0, base * (gate.double() > 0))
    assert torch.isnan(C[:, N:]).all()
    # round 4: with 16-byte aligned gate rows and N % 8 == 0 the gate is applied to the staged (already rounded) rows, read
    # row-contiguously; a select commutes with the rounding, so the result must be BIT-identical to the element-wise path,
    # which an 8-byte-aligned view of the same values selects
    wide = torch.zeros(M, ld + 16, dtype=dtype, device=DEV)
    wide[:, 4:4 + N] = gate.to(DEV)
    gate8 = wide[:, 4:4 + ld]
    assert gate8.data_ptr() % 16 == 8
    C2 = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(Ad, Bd, C2, gate=gate8, flags=ops.ME_EPI_RELU_BWD, N=N)
    assert torch.equal(C2[:, :N], C[:, :N]) and torch.isnan(C2[:, N:]).all()
    C32 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm_nt(Ad, Bd, C32, bias=bias.to(DEV), flags=ops.ME_EPI_OUT_F32)
    assert relerr(C32, base + bias.double()) < 1e-5
    # f32 output with a residual operand and a gate: the element-wise write-out path of the f32-out instantiation
    ops.gemm_nt(Ad, Bd, C32, bias=bias.to(DEV), add=addp, gate=gatep, flags=ops.ME_EPI_OUT_F32 | ops.ME_EPI_RELU_BWD, N=N)
    assert relerr(C32, (base + bias.double() + add.double()) * (gate.double() > 0)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_epilogues(ops, dtype):
    M, N, K = 200, 136, 96
    A = rnd(M, K, seed=4).to(dtype)
    B = rnd(N, K, seed=5).to(dtype)
    bias = rnd(N, seed=6).float()
    add = rnd(M, N, seed=7).to(dtype)
    gate = rnd(M, N, seed=8).to(dtype)
    base = A.double() @ B.double().t()
    Ad, Bd = A.to(DEV), B.to(DEV)
    C = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.gemm_nt(Ad, Bd, C, bias=bias.to(DEV), flags=ops.ME_EPI_RELU)
    assert relerr(C, torch.relu(base + bias.double())) < tol(dtype, 1e-5, torch.relu(base + bias.double()))
    ops.gemm_nt(Ad, Bd, C, add=add.to(DEV))
    assert relerr(C, base + add.double()) < tol(dtype, 1e-5, base + add.double())
    ops.gemm_nt(Ad, Bd, C, gate=gate.to(DEV), flags=ops.ME_EPI_RELU_BWD)
    assert relerr(C, base * (gate.double() > 0)) < tol(dtype, 1e-5, base * (gate.double() > 0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,N,K", [(256, 128, 128), (1000, 136, 72), (21, 1007, 64), (4096, 96, 256), (4096, 512, 512),
                                   (2048, 1007, 256), (8192, 200, 1536),
                                   (5000, 256, 512), (2101, 512, 256), (32768, 1536, 512), (4100, 1007, 512)])   # 256-tile kernel, ragged ranges
def test_gemm_tn_acc(ops, dtype, T, N, K):
    ch = 4 if dtype == torch.float32 else 8
    ldA = ((N + ch - 1) // ch) * ch + ch
    if (T, N, K) == (4100, 1007, 512):
        ldA = 1024                      # rows readable up to the next multiple of 256: 256-tile kernel with N % 256 != 0
    A = torch.zeros(T, ldA, dtype=dtype)
    A[:, :N] = rnd(T, N, seed=9).to(dtype)
    B = rnd(T, K, seed=10).to(dtype)
    dW0 = rnd(N, K, seed=11).float()
    db0 = rnd(N, seed=12).float()
    ref = dW0.double() + A[:, :N].double().t() @ B.double()
    refb = db0.double() + A[:, :N].double().sum(0)
    dW = dW0.clone().to(DEV)
    db = db0.clone().to(DEV)
    need = ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N, K, dtype)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV) if need else None
    ops.gemm_tn_acc(A.to(DEV), B.to(DEV), dW, db, T=T, N=N, K=K, ws=ws)
    assert relerr(dW, ref) < 2e-5, relerr(dW, ref)   # inputs already rounded, f32 accumulate and f32 output in both tiers
    assert relerr(db, refb) < 2e-5, relerr(db, refb)


@pytest.mark.parametrize("dt16", DT16)
def test_gemm_tn_acc_caller_workspace(ops, dt16):
    """SURVEY 8b ownership: the partial-tile workspace is the caller's (me_workspace_bytes).  With it the summation
    order is fixed -> bit-identical results across launches that reuse the buffer; without it the kernel falls back to
    f32 atomics (same value up to rounding order); a buffer that is too small is an error, not a silent fallback."""
    T, N, K = 8192, 512, 256
    A = rnd(T, N, seed=21).to(dt16).to(DEV)
    B = rnd(T, K, seed=22).to(dt16).to(DEV)
    need = ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N, K, dt16)
    assert need > 0 and ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, N, K, torch.float32) == 0
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    got = [torch.zeros(N, K, device=DEV) for _ in range(3)]
    for g in got:
        ops.gemm_tn_acc(A, B, g, None, T=T, N=N, K=K, ws=ws)
    for g in got[1:]:
        assert torch.equal(g, got[0])                    # fixed summation order: bit-identical
    atom = torch.zeros(N, K, device=DEV)
    ops.gemm_tn_acc(A, B, atom, None, T=T, N=N, K=K, ws=None)
    ref = A.double().t() @ B.double()
    assert relerr(got[0], ref) < 2e-5 and relerr(atom, ref) < 2e-5
    with pytest.raises(RuntimeError, match="ME_ERR_WORKSPACE"):
        ops.gemm_tn_acc(A, B, atom, None, T=T, N=N, K=K, ws=ws[:need // 2])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,shapes", [(8192, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]),    # a layer's four weight gradients
                                      (4100, [(256, 256), (1007, 512)]),                             # ragged token ranges, N % 256 != 0
                                      (4096, [(96, 256), (512, 512)]),                               # one item not readable to 256 columns: separate launches
                                      (300, [(128, 128), (64, 72)])])                                # small shapes: separate launches
def test_gemm_tn_acc_group(ops, dtype, T, shapes):
    """me_gemm_tn_acc_group: several dW += dY^T X products over one token dimension in one launch == the separate calls
    (same values up to the summation order, which stays fixed: two runs are bit-identical)."""
    ch = 4 if dtype == torch.float32 else 8
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        ldA = 1024 if N == 1007 else ((N + ch - 1) // ch) * ch
        A = torch.zeros(T, ldA, dtype=dtype)
        A[:, :N] = rnd(T, N, seed=40 + i).to(dtype)
        X = rnd(T, K, seed=50 + i).to(dtype)
        dW0, db0 = rnd(N, K, seed=60 + i).float(), rnd(N, seed=70 + i).float()
        refs.append((dW0.double() + A[:, :N].double().t() @ X.double(), db0.double() + A[:, :N].double().sum(0)))
        items.append([A.to(DEV), X.to(DEV), dW0, db0 if i != 1 else None, N, K])
    need = max([ops.workspace_bytes(ops.ME_WS_GEMM_TN_GROUP, T, ops.tn_group_tiles([(n, k) for n, k in shapes if k % 256 == 0]), 0, dtype)] +
               [ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, n, k, dtype) for n, k in shapes])
    ws = torch.empty(need, dtype=torch.uint8, device=DEV) if need else None
    runs = []
    for _ in range(2):
        cur = [(A, X, dW0.clone().to(DEV), db0.clone().to(DEV) if db0 is not None else None, N, K) for A, X, dW0, db0, N, K in items]
        ops.gemm_tn_acc_group(cur, T, dtype, ws=ws)
        runs.append(cur)
    for (A, X, dW, db, N, K), (rW, rb) in zip(runs[0], refs):
        assert relerr(dW, rW) < 2e-5, (N, K, relerr(dW, rW))
        if db is not None:
            assert relerr(db, rb) < 2e-5, (N, K, relerr(db, rb))
    grouped = dtype != torch.float32 and T >= 2048 and all(k % 256 == 0 and (n % 256 == 0 or n == 1007) for n, k in shapes)
    if ws is not None and grouped:                       # the grouped kernel: fixed summation order
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a[2], b[2])
            if a[3] is not None:
                assert torch.equal(a[3], b[3])               # the bias column sums go through the workspace as well


@pytest.mark.parametrize("dt", DT16)
def test_gemm_full_size_replication_property(ops, dt):
    """The train step's GEMM shapes at FULL size (T = 32768 tokens, bf16) through a size-independent property: the token
    dimension is a 256-row block replicated 128 times.  NT: every 256-row block of C must be bit-identical to the 256-row
    product (tile walk, XCD renumbering, persistent-tile bookkeeping).  TN (grouped launch of a layer's four products):
    dW of the full problem = 128 x dW of one block up to f32 summation order, and two runs are bit-identical."""
    T0, R = 256, 128
    T = T0 * R
    shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048)]
    for N, K in shapes:
        A0 = rnd(T0, K, seed=N + K).to(dt).to(DEV)
        W = rnd(N, K, seed=N + 2 * K).to(dt).to(DEV)
        bias = rnd(N, seed=5).float().to(DEV)
        C0 = torch.empty(T0, N, dtype=dt, device=DEV)
        ops.gemm_nt(A0, W, C0, bias=bias)
        C = torch.empty(T, N, dtype=dt, device=DEV)
        ops.gemm_nt(A0.repeat(R, 1), W, C, bias=bias)
        assert torch.equal(C.view(R, T0, N), C0.expand(R, T0, N)), (N, K)
        ref = (A0.double() @ W.double().t() + bias.double())
        assert relerr(C0, ref) < tol(dt, 1e-6, ref)
    items_small, items_full = [], []
    for N, K in shapes:
        dY0 = rnd(T0, N, seed=3 * N + K).to(dt).to(DEV)
        X0 = rnd(T0, K, seed=N + 7 * K).to(dt).to(DEV)
        items_small.append((dY0, X0))
        items_full.append((dY0.repeat(R, 1), X0.repeat(R, 1)))
    def run(items, Tn):
        outs = [(torch.zeros(a.shape[1], b.shape[1], device=DEV), torch.zeros(a.shape[1], device=DEV)) for a, b in items]
        need = max([ops.workspace_bytes(ops.ME_WS_GEMM_TN_GROUP, Tn, ops.tn_group_tiles(shapes), 0, dt)] +
                   [ops.workspace_bytes(ops.ME_WS_GEMM_TN, Tn, n, k, dt) for n, k in shapes])
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
        ops.gemm_tn_acc_group([(a, b, dW, db, a.shape[1], b.shape[1]) for (a, b), (dW, db) in zip(items, outs)], Tn, dt, ws=ws)
        return outs
    full1, full2, small = run(items_full, T), run(items_full, T), run(items_small, T0)
    for (dW1, db1), (dW2, db2), (dWs, dbs), (N, K) in zip(full1, full2, small, shapes):
        assert torch.equal(dW1, dW2) and torch.equal(db1, db2), (N, K)          # fixed summation order
        assert relerr(dW1, R * dWs) < 1e-5 and relerr(db1, R * dbs) < 1e-5, (N, K)


@pytest.mark.parametrize("dt16", DT16)
def test_resid_ln_fwd_hi_lo_residual_stream(ops, dt16):
    """bf16 tier: the residual stream travels as hi + lo (me_resid_ln_fwd x_lo / y_lo).  y must be exactly
    bf16(LN(x_hi + x_lo + a)) computed in f32, and y + y_lo must carry ~16 mantissa bits of it."""
    rows, d = 300, 512
    x = rnd(rows, d, seed=51).float() * 3
    a = rnd(rows, d, seed=52).to(dt16)
    gamma, beta = rnd(d, seed=53).float(), rnd(d, seed=54).float()
    x_hi = x.to(dt16)
    x_lo = (x - x_hi.float()).to(dt16)
    xs = x_hi.double() + x_lo.double() + a.double()
    ref = torch.nn.functional.layer_norm(xs, (d,), gamma.double(), beta.double(), 1e-6)
    y = torch.empty(rows, d, dtype=dt16, device=DEV)
    y_lo = torch.empty_like(y)
    ops.resid_ln_fwd(x_hi.to(DEV), a.to(DEV), gamma.to(DEV), beta.to(DEV), y, None, None, rows, d, 1e-6, 0.0, 0, 1,
                     x_lo=x_lo.to(DEV), y_lo=y_lo)
    assert relerr(y, ref) < 3e-3                                        # one bf16 rounding
    assert relerr(y.double() + y_lo.double(), ref) < 2e-5               # hi + lo: f32-class
    assert (y.float().cpu() - ref.float().to(dt16).float()).abs().max() <= (2.0 ** -6 if dt16 == torch.bfloat16 else 2.0 ** -9)   # ulp-level agreement with T(ref)
    y2 = torch.empty_like(y)                                            # without lo the input rounding is visible
    ops.resid_ln_fwd(x_hi.to(DEV), a.to(DEV), gamma.to(DEV), beta.to(DEV), y2, None, None, rows, d, 1e-6, 0.0, 0, 1)
    ref_hi = torch.nn.functional.layer_norm(x_hi.double() + a.double(), (d,), gamma.double(), beta.double(), 1e-6)
    assert relerr(y2, ref_hi) < 3e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_cast_transpose(ops, dtype):
    src = rnd(70, 1007, seed=13).float().to(DEV)
    dst = torch.zeros(70, 1008, dtype=dtype, device=DEV)
    dstT = torch.zeros(1007, 72, dtype=dtype, device=DEV)
    ops.cast_transpose(src, dst, dstT, dtype)
    assert torch.equal(dst[:, :1007], src.to(dtype))
    assert torch.equal(dstT[:, :70], src.to(dtype).t())
    assert (dst[:, 1007:] == 0).all() and (dstT[:, 70:] == 0).all()


def _ln_ref(s, gamma, beta, dtype):
    x = torch.nn.functional.layer_norm(s.double(), (s.shape[1],), gamma.double(), beta.double(), 1e-6)
    return x, x.float().to(dtype).double()                      # f32-class LN output, and the T-rounded GEMV operand


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Mr", [1, 4, 7])
def test_dec_ln_proj(ops, dtype, Mr):
    """me_dec_ln_proj: LayerNorm prologue + projection (+ ReLU, T output) / (f32 logits), x_out = the f32 LN rows."""
    N, K = 1007, 512
    s_in = rnd(Mr, K, seed=14).float() * 2
    gamma, beta = rnd(K, seed=17).float() + 1.5, rnd(K, seed=18).float()
    W = rnd(N, K, seed=15).to(dtype)
    bias = rnd(N, seed=16).float()
    x, xr = _ln_ref(s_in, gamma, beta, dtype)
    ref = xr @ W.double().t() + bias.double()
    y = torch.full((Mr, N), 9.0, dtype=torch.float32, device=DEV)
    xo = torch.zeros(Mr, K, dtype=torch.float32, device=DEV)
    ops.dec_ln_proj(s_in.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, W.to(DEV), bias.to(DEV), xo, y, Mr, N, K,
                    ops.ME_EPI_OUT_F32, dtype)
    assert relerr(y, ref) < tol(dtype, 1e-5, xr), relerr(y, ref)     # bf16: the LN rows are rounded to bf16 before the projection (f32 output)
    assert relerr(xo, x) < 1e-5
    yt = torch.zeros(Mr, N + 8, dtype=dtype, device=DEV)
    ops.dec_ln_proj(s_in.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, W.to(DEV), bias.to(DEV), None, yt, Mr, N, K,
                    ops.ME_EPI_RELU, dtype)
    assert relerr(yt[:, :N], torch.relu(ref)) < tol(dtype, 1e-5, torch.relu(ref), xr)
    assert (yt[:, N:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Mr,N,K", [(4, 512, 2048), (3, 96, 768), (8, 768, 3072)])
def test_dec_proj_resid_plain(ops, dtype, Mr, N, K):
    """me_dec_proj_resid without attention partials: out = resid + bias + x_T . W^T (FFN_suf + residual)."""
    x = rnd(Mr, K, seed=24).to(dtype)
    W = rnd(N, K, seed=25).to(dtype)
    bias, resid = rnd(N, seed=26).float(), rnd(Mr, N, seed=27).float()
    ref = resid.double() + bias.double() + x.double() @ W.double().t()
    out = torch.zeros(Mr, N, dtype=torch.float32, device=DEV)
    ops.dec_proj_resid(None, 0, 0, 0, x.to(DEV), W.to(DEV), bias.to(DEV), resid.to(DEV), out, Mr, N, K, dtype)
    assert relerr(out, ref) < 2e-5, relerr(out, ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_dec_qkv_ln_prologue_and_cache_append(ops, dtype):
    """me_dec_qkv: (LayerNorm | hi + lo) -> q|k|v projection; q to its buffer, k / v appended at position t of the
    [Mr, H, Mc, dh] caches (host t and device t_dev), nothing else in the caches touched."""
    Mr, H, dh, Mc = 3, 4, 32, 40
    d = H * dh
    s_in = rnd(Mr, d, seed=31).float() * 2
    gamma, beta = rnd(d, seed=32).float() + 1.5, rnd(d, seed=33).float()
    W = rnd(3 * d, d, seed=34).to(dtype)
    bias = rnd(3 * d, seed=35).float()
    x, xr = _ln_ref(s_in, gamma, beta, dtype)
    ref = (xr @ W.double().t() + bias.double()).float().to(dtype)
    for t, use_dev in ((5, False), (17, True)):
        kc = torch.full((Mr, H, Mc, dh), 7.0, dtype=dtype, device=DEV)
        vc = torch.full((Mr, H, Mc, dh), -7.0, dtype=dtype, device=DEV)
        q = torch.zeros(Mr, d, dtype=dtype, device=DEV)
        xo = torch.zeros(Mr, d, dtype=torch.float32, device=DEV)
        t_dev = torch.tensor([t], dtype=torch.int32, device=DEV) if use_dev else None
        ops.dec_qkv(s_in.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, None, None, W.to(DEV), bias.to(DEV), xo, q, kc, vc, Mr, d, H,
                    dh, Mc, 0 if use_dev else t, t_dev, dtype)
        assert relerr(q, ref[:, :d]) < tol(dtype, 1e-5, ref[:, :d], xr)
        assert relerr(kc[:, :, t].reshape(Mr, d), ref[:, d:2 * d]) < tol(dtype, 1e-5, ref[:, d:2 * d], xr)
        assert relerr(vc[:, :, t].reshape(Mr, d), ref[:, 2 * d:]) < tol(dtype, 1e-5, ref[:, 2 * d:], xr)
        keep = torch.ones(Mc, dtype=torch.bool)
        keep[t] = False
        assert (kc[:, :, keep] == 7.0).all() and (vc[:, :, keep] == -7.0).all()
        assert relerr(xo, x) < 1e-5
    # first layer: x = hi + lo
    xf = rnd(Mr, d, seed=36).float()
    hi = xf.to(dtype)
    lo = (xf - hi.float()).to(dtype)
    ref2 = ((hi.double() + lo.double()).float().to(dtype).double() @ W.double().t() + bias.double())
    ops.dec_qkv(None, None, None, 1e-6, hi.to(DEV), lo.to(DEV) if dtype != torch.float32 else None, W.to(DEV), bias.to(DEV), xo, q,
                kc, vc, Mr, d, H, dh, Mc, 3, None, dtype)
    assert relerr(q, ref2[:, :d]) < tol(dtype, 1e-5, ref2[:, :d])
    assert relerr(xo, hi.double() + (lo.double() if dtype != torch.float32 else 0)) < 1e-6



# ------------------------------------------------------------------ residual + LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(37, 64), (256, 512), (5, 256), (301, 512), (1, 512), (8195, 256)])   # odd row counts: the two-rows-per-wave tails
def test_resid_ln_fwd_bwd(ops, dtype, rows, d):
    x = rnd(rows, d, seed=17).to(dtype)
    a = rnd(rows, d, seed=18).to(dtype)
    gamma = (1 + 0.1 * rnd(d, seed=19)).float()
    beta = (0.1 * rnd(d, seed=20)).float()
    dy = rnd(rows, d, seed=21).to(dtype)
    xs = x.double().requires_grad_(True)
    as_ = a.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = beta.double().requires_grad_(True)
    yref = O.layer_norm(xs + as_, g64, b64, 1e-6)
    (yref * dy.double()).sum().backward()

    y = torch.empty(rows, d, dtype=dtype, device=DEV)
    s = torch.empty_like(y)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=DEV)
    ops.resid_ln_fwd(x.to(DEV), a.to(DEV), gamma.to(DEV), beta.to(DEV), y, s, stats, rows, d, 1e-6, 0.0, 0, 1)
    assert relerr(y, yref.detach()) < tol(dtype, 2e-6, yref)
    assert relerr(s, (x.double() + a.double())) < tol(dtype, 1e-7, x.double() + a.double())
    dx = torch.empty_like(y)
    da = torch.empty_like(y)
    dg = torch.zeros(d, dtype=torch.float32, device=DEV)
    db = torch.zeros(d, dtype=torch.float32, device=DEV)
    ops.resid_ln_bwd(dy.to(DEV), s, stats, gamma.to(DEV), dx, da, dg, db, rows, d, 0.0, 0, 1)
    assert relerr(dx, xs.grad) < tol(dtype, 1e-5, xs.grad, x.double() + a.double())      # bf16: dx stored, the saved sum s stored
    assert torch.equal(dx, da)
    assert relerr(dg, g64.grad) < tol(dtype, 1e-5, x.double() + a.double())
    assert relerr(db, b64.grad) < tol(dtype, 1e-5, x.double() + a.double())


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_mask_consistency(ops, dtype):
    """Dropout: keep-rate ~ 1-p, forward mask == backward mask, inverted scaling."""
    rows, d, p = 512, 512, 0.1
    x = torch.zeros(rows, d, dtype=dtype, device=DEV)
    a = torch.ones(rows, d, dtype=dtype, device=DEV)
    gamma = torch.ones(d, device=DEV)
    beta = torch.zeros(d, device=DEV)
    y = torch.empty_like(x)
    s = torch.empty_like(x)
    stats = torch.empty(rows, 2, device=DEV)
    ops.resid_ln_fwd(x, a, gamma, beta, y, s, stats, rows, d, 1e-6, p, 1234, 3)
    keep = (s.float() != 0)
    rate = keep.float().mean().item()
    assert abs(rate - 0.9) < 0.005, rate
    assert torch.allclose(s.float()[keep], torch.tensor(1 / 0.9, device=DEV), rtol=1e-2)
    s2 = torch.empty_like(x)
    ops.resid_ln_fwd(x, a, gamma, beta, y, s2, stats, rows, d, 1e-6, p, 1234, 4)
    assert (s2 != s).any()                          # different site -> different mask
    ops.resid_ln_fwd(x, a, gamma, beta, y, s2, stats, rows, d, 1e-6, p, 1234, 3)
    assert torch.equal(s2, s)                       # same (seed, site) -> same mask
    dx = torch.empty_like(x)
    da = torch.empty_like(x)
    dg = torch.zeros(d, device=DEV)
    db = torch.zeros(d, device=DEV)
    dy = torch.randn(rows, d, device=DEV).to(dtype)
    ops.resid_ln_bwd(dy, s, stats, gamma, dx, da, dg, db, rows, d, p, 1234, 3)
    assert torch.equal(da.float() != 0, keep & (dx.float() != 0))
    sel = keep & (dx.float().abs() > 1e-3)
    assert torch.allclose(da.float()[sel], dx.float()[sel] / 0.9, rtol=2e-2)


# ------------------------------------------------------------------ cross entropy
@pytest.mark.parametrize("dtype", DTYPES)
def test_ce_fwd_bwd(ops, dtype):
    rows, V, ld = 97, 1007, 1008
    lg = rnd(rows, V, seed=22, scale=3.0).float()
    tgt = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(23))
    tgt[::5] = 0
    l64 = lg.double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(l64, tgt, ignore_index=0)
    loss.backward()
    lgd = torch.zeros(rows, ld, device=DEV)
    lgd[:, :V] = lg.to(DEV)
    row_lse = torch.empty(rows, device=DEV)
    acc = torch.zeros(2, device=DEV)
    ops.ce_fwd(lgd, tgt.to(DEV), row_lse, acc[0:1], acc[1:2], rows, V, 0)
    nvalid = int((tgt != 0).sum())
    assert abs(acc[1].item() - nvalid) < 1e-3
    assert abs(acc[0].item() / nvalid - loss.item()) < 2e-5 * abs(loss.item())
    assert relerr(row_lse, torch.logsumexp(lg.double(), -1)) < 1e-6
    dl = torch.full((rows, ld), float("nan"), dtype=dtype, device=DEV)
    ops.ce_bwd(lgd, tgt.to(DEV), row_lse, dl, acc[1:2], 1.0, rows, V, 0)
    assert relerr(dl[:, :V], l64.grad) < tol(dtype, 1e-5, l64.grad)
    assert (dl[:, V:] == 0).all()


def test_rowwise_kernels_full_size_replication_property(ops):
    """LayerNorm forward / backward and the loss head at FULL size (T = 32768 rows, bf16, d = 512, V = 1007) through row
    independence: the rows are a 256-row block replicated 128 times.  Per-row outputs of every replica must be bit-identical
    to the 256-row launch (which the tests above check against f64); sums over rows (d gamma, d beta, loss, bias gradient)
    must be 128 x the small launch's up to f32 summation order; dlogits scale with 1 / n_valid = an exact power of two."""
    dt = torch.bfloat16
    T0, R, d, V, ld = 256, 128, 512, 1007, 1024
    T = T0 * R
    x0, xl0, a0 = (rnd(T0, d, seed=90 + i).to(dt).to(DEV) for i in range(3))
    gamma, beta = (1 + 0.1 * rnd(d, seed=93)).float().to(DEV), (0.1 * rnd(d, seed=94)).float().to(DEV)
    def ln(x, xl, a, rows):
        y, yl, s_ = (torch.empty(rows, d, dtype=dt, device=DEV) for _ in range(3))
        st = torch.empty(rows, 2, device=DEV)
        ops.resid_ln_fwd(x, a, gamma, beta, y, s_, st, rows, d, 1e-6, 0.0, 0, 1, x_lo=xl, y_lo=yl)
        return y, yl, s_, st
    small = ln(x0, xl0, a0, T0)
    full = ln(x0.repeat(R, 1), xl0.repeat(R, 1), a0.repeat(R, 1), T)
    for f, s_ in zip(full, small):
        assert torch.equal(f.view(R, T0, -1), s_.expand(R, *s_.shape))
    dy0 = rnd(T0, d, seed=95).to(dt).to(DEV)
    def lnb(dy, s_, st, rows):
        dx, da = torch.empty(rows, d, dtype=dt, device=DEV), torch.empty(rows, d, dtype=dt, device=DEV)
        dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        ops.resid_ln_bwd(dy, s_, st, gamma, dx, da, dg, db, rows, d, 0.0, 0, 1)
        return dx, dg, db
    bs = lnb(dy0, small[2], small[3], T0)
    bf = lnb(dy0.repeat(R, 1), full[2], full[3], T)
    assert torch.equal(bf[0].view(R, T0, d), bs[0].expand(R, T0, d))
    assert relerr(bf[1], R * bs[1]) < 1e-5 and relerr(bf[2], R * bs[2]) < 1e-5
    # loss head: bf16 logits [rows, ld], ignore_index 0
    lg0 = torch.zeros(T0, ld, dtype=dt, device=DEV)
    lg0[:, :V] = rnd(T0, V, seed=96, scale=4.0).to(dt).to(DEV)
    tgt0 = torch.randint(0, V, (T0,), generator=torch.Generator().manual_seed(97))
    tgt0[::6] = 0
    def ce(lg, tgt, rows):
        row_lse = torch.empty(rows, device=DEV)
        acc = torch.zeros(2, device=DEV)
        ops.ce_fwd(lg, tgt, row_lse, acc[0:1], acc[1:2], rows, V, 0)
        dl = torch.empty(rows, ld, dtype=dt, device=DEV)
        dbias = torch.zeros(ld, device=DEV)
        ops.ce_bwd(lg, tgt, row_lse, dl, acc[1:2], 1.0, rows, V, 0, dbias=dbias if ops.ce_bwd_fuses_dbias(lg, dl) else None)
        return row_lse, acc, dl, dbias
    cs = ce(lg0, tgt0.to(DEV), T0)
    cf = ce(lg0.repeat(R, 1), tgt0.repeat(R).to(DEV), T)
    assert torch.equal(cf[0].view(R, T0), cs[0].expand(R, T0))
    assert abs(cf[1][1].item() - R * cs[1][1].item()) < 0.5 and abs(cf[1][0].item() / (R * cs[1][0].item()) - 1) < 1e-5
    assert torch.equal((cf[2].float() * R).view(R, T0, ld), cs[2].float().expand(R, T0, ld))      # 1 / n_valid: exact factor 2^-7
    assert relerr(cf[3][:V], cs[3][:V]) < 1e-5                                                  # sum of 128 x (1 / 128) rows


@pytest.mark.parametrize("V,ld", [(1007, 1024), (1017, 1024), (97, 128), (2500, 2560)])
@pytest.mark.parametrize("dt16", DT16)
def test_ce_bf16_logits(ops, V, ld, dt16):
    """bf16 tier: the head GEMM writes bf16 logits (me_ce_fwd / me_ce_bwd with logits_dtype = ME_BF16).  The kernels
    must give exactly the cross-entropy of those rounded logits (f32 arithmetic): compared with fp64 CE of the same
    bf16 values; padding columns [V, ld) hold garbage and must not leak in."""
    rows = 203
    lg = rnd(rows, V, seed=31, scale=4.0).to(dt16)
    tgt = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(32))
    tgt[::7] = 0
    l64 = lg.double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(l64, tgt, ignore_index=0)
    loss.backward()
    lgd = torch.full((rows, ld), 1e4, dtype=dt16, device=DEV)        # huge padding: would dominate the lse
    lgd[:, :V] = lg.to(DEV)
    row_lse = torch.empty(rows, device=DEV)
    acc = torch.zeros(2, device=DEV)
    ops.ce_fwd(lgd, tgt.to(DEV), row_lse, acc[0:1], acc[1:2], rows, V, 0)
    nvalid = int((tgt != 0).sum())
    assert abs(acc[1].item() - nvalid) < 1e-3
    assert abs(acc[0].item() / nvalid - loss.item()) < 2e-5 * abs(loss.item())
    assert relerr(row_lse, torch.logsumexp(lg.double(), -1)) < 1e-6
    dl = torch.full((rows, ld), float("nan"), dtype=dt16, device=DEV)
    # (x 4096 through the device-resident loss scale: f16 dlogits of a 1 / n_valid mean would sit in the subnormal range)
    lsc = torch.full((1,), 4096.0, device=DEV)
    ops.ce_bwd(lgd, tgt.to(DEV), row_lse, dl, acc[1:2], 1.0, rows, V, 0, loss_scale=lsc)
    assert relerr(dl[:, :V], l64.grad * 4096.0) < tol(dt16, 0, l64.grad * 4096.0)
    assert (dl[:, V:] == 0).all()
    # fused head-bias gradient: f32 column sums of dlogits BEFORE the rounding to bf16 (accumulated into dbias)
    if ld <= 2048:
        assert ops.ce_bwd_fuses_dbias(lgd, dl)
        db = torch.zeros(V, device=DEV)
        dl2 = torch.full_like(dl, float("nan"))
        ops.ce_bwd(lgd, tgt.to(DEV), row_lse, dl2, acc[1:2], 1.0, rows, V, 0, dbias=db, loss_scale=lsc)
        if dt16 == torch.bfloat16:
            assert torch.equal(dl2, dl)
        else:      # f16: hipcc fuses (x * scale -> f16) into ONE rounding (v_fma_mixlo_f16) in the variant without the column sums and
            # rounds twice (v_mul_f32, v_cvt_pk_f16_f32) in the one with them: neighbours at most, and only on double-rounding ties
            assert int((dl2.view(torch.int16).int() - dl.view(torch.int16).int()).abs().max()) <= 1
            assert float((dl2 != dl).float().mean()) < 1e-3
        assert relerr(db, 4096.0 * l64.grad.sum(0)) < 2e-5                           # f32 exact, not the 16-bit column sums
        ops.ce_bwd(lgd, tgt.to(DEV), row_lse, dl2, acc[1:2], 1.0, rows, V, 0, dbias=db, loss_scale=lsc)   # accumulates (+=)
        assert relerr(db, 2 * 4096.0 * l64.grad.sum(0)) < 2e-5
        if dt16 == torch.bfloat16:
            assert relerr(dl[:, :V].double().sum(0).cpu(), 4096.0 * l64.grad.sum(0)) > 1e-4   # ... which is what it replaces
    else:
        assert not ops.ce_bwd_fuses_dbias(lgd, dl)
        with pytest.raises(RuntimeError, match="ME_ERR_BAD_SHAPE"):
            ops.ce_bwd(lgd, tgt.to(DEV), row_lse, dl, acc[1:2], 1.0, rows, V, 0, dbias=torch.zeros(V, device=DEV))


# ------------------------------------------------------------------ optimiser
def test_sumsq_and_adamw(ops):
    n = 100003
    rs = np.random.RandomState(5)
    P = {"w": torch.from_numpy(rs.standard_normal(n)).double()}
    G = {"w": torch.from_numpy(rs.standard_normal(n) * 0.01).double()}
    M1 = {"w": torch.zeros(n, dtype=torch.float64)}
    M2 = {"w": torch.zeros(n, dtype=torch.float64)}
    p = P["w"].float().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        G["w"] = torch.from_numpy(rs.standard_normal(n) * (2.0 if step == 2 else 0.001)).double()
        g = G["w"].float().to(DEV)
        ss = torch.zeros(1, device=DEV)
        ops.sumsq(g, ss)
        assert abs(ss.item() - float((G["w"] ** 2).sum())) < 1e-4 * float((G["w"] ** 2).sum())
        O.adam_step(P, G, M1, M2, step, lr=1e-3, clip=1.0)
        ops.adamw_step(p, g, m, v, ss, 1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, True)
        assert (g == 0).all()
        assert float((p.double().cpu() - P["w"]).abs().max()) < 2e-6, step


def test_sumsq_ordered_is_bit_reproducible(ops):
    """me_sumsq with the caller's workspace adds the block sums in index order: the result must not depend on which block
    finishes when (it sets the clip coefficient, and data-parallel ranks must come out of the update bit-identical --
    round-4 DDP diagnosis).  20.6 M elements = the headline model's flat gradient; the same call repeated while other
    work (a second stream hammering the memory system) perturbs the arrival order; the workspace is reused (the library
    leaves its ticket zeroed); the accumulate contract (*out +=) holds."""
    n = 20604399
    g = torch.randn(n, device=DEV) * 1e-3
    ws = ops.sumsq_ws(DEV)
    ref = None
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=DEV)
    for it in range(24):
        out = torch.zeros(1, device=DEV)
        if it % 2:
            with torch.cuda.stream(side):
                junk.normal_()                                  # competes for CUs / HBM: different block arrival order
        ops.sumsq(g, out, ws=ws)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
            exact = float((g.double() ** 2).sum())
            assert abs(float(ref) - exact) <= 1e-5 * exact
        assert torch.equal(out, ref), (it, float(out), float(ref))
    assert int(ws.view(torch.int32)[0]) == 0                    # ticket counter left zeroed
    # the slots of the REUSED workspace must never be served stale (another XCD's L2 may still hold the previous call's
    # lines): different data on every call, each checked against its own f64 sum
    for it in range(12):
        gi = g * (1.0 + 0.37 * it)
        out = torch.zeros(1, device=DEV)
        if it % 3 == 1:
            with torch.cuda.stream(side):
                junk.normal_()
        ops.sumsq(gi, out, ws=ws)
        torch.cuda.synchronize()
        exact = float((gi.double() ** 2).sum())
        assert abs(float(out) - exact) <= 2e-6 * exact, (it, float(out), exact)
    out = torch.full((1,), 2.0, device=DEV)
    ops.sumsq(g, out, ws=ws)
    assert abs(float(out) - 2.0 - float(ref)) <= 1e-6 * float(ref) + 1e-6
    with pytest.raises(RuntimeError, match="ME_ERR_WORKSPACE"):
        ops.sumsq(g, out, ws=torch.zeros(16, dtype=torch.uint8, device=DEV))


# ------------------------------------------------------------------ embedding prologue
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", ["none", "continuous_concat", "continuous_token"])
def test_embed_fwd_bwd(ops, dtype, mode):
    V, d, dc, B, L = 97, 64, 16, 3, 21
    cfg = O.Cfg(V, 1, 2, d, 128, d_condition=dc, conditioning=mode, max_seq=64)
    P = {k: v.double() for k, v in O.seeded_params(cfg, 3).items()}
    tok, cond, _ = O.synthetic_batch(cfg, B, L, 7)
    tok[1, -3:] = 0
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.embed(cfg, Pg, tok, cond.double())
    dy = rnd(*ref.shape, seed=24)
    (ref * dy).sum().backward()
    Ltok = tok.shape[1]
    Lm = ref.shape[1]
    code = {"none": ops.ME_COND_NONE, "continuous_concat": ops.ME_COND_CONCAT, "continuous_token": ops.ME_COND_TOKEN}[mode]
    f = lambda k: P[k].float().contiguous().to(DEV) if k in P else None
    pe = O.sinusoid_pe(64, d).float().to(DEV)
    cw0 = f("fc_condition.weight") if mode == "continuous_concat" else f("fc_condition.0.weight")
    cb0 = f("fc_condition.bias") if mode == "continuous_concat" else f("fc_condition.0.bias")
    cw1, cb1 = f("fc_condition.1.weight"), f("fc_condition.1.bias")
    out = torch.empty(B, Lm, d, dtype=dtype, device=DEV)
    ops.embed_fwd(out, tok.to(DEV), cond.to(DEV), f("embedding.weight"), cw0, cb0, cw1, cb1, pe, code, B, Ltok, d,
                  cfg.d_condition, 0.0, 0)
    assert relerr(out, ref.detach()) < tol(dtype, 1e-6, ref)
    zl = lambda t: torch.zeros_like(t) if t is not None else None
    g_emb, g_cw0, g_cb0, g_cw1, g_cb1 = zl(f("embedding.weight")), zl(cw0), zl(cb0), zl(cw1), zl(cb1)
    ops.embed_bwd(dy.to(dtype).to(DEV), tok.to(DEV), cond.to(DEV), g_emb, g_cw0, g_cb0, g_cw1, g_cb1, code, B, Ltok, d,
                  cfg.d_condition, 0, 0.0, 0)
    gref = Pg["embedding.weight"].grad.clone()
    gref[0] = 0
    assert relerr(g_emb, gref) < tol(dtype, 1e-5, dy)          # f32 sums of the bf16-rounded dy rows
    assert (g_emb[0] == 0).all()
    if mode == "continuous_concat":
        assert relerr(g_cw0, Pg["fc_condition.weight"].grad) < tol(dtype, 1e-5, dy)
        assert relerr(g_cb0, Pg["fc_condition.bias"].grad) < tol(dtype, 1e-5, dy)
    if mode == "continuous_token":
        for i, (gw, gb) in enumerate(((g_cw0, g_cb0), (g_cw1, g_cb1))):
            assert relerr(gw, Pg[f"fc_condition.{i}.weight"].grad) < tol(dtype, 1e-5, dy)
            assert relerr(gb, Pg[f"fc_condition.{i}.bias"].grad) < tol(dtype, 1e-5, dy)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("p_top", [0.45, 0.85])
def test_embed_bwd_frequent_tokens(ops, dtype, use_ws, p_top):
    """Token ids as skewed as real MIDI streams (one id on ~45 % of the positions, another on 12 %, PAD in between): the
    table gradient must equal the index_add of the dy rows whether the frequent rows are summed by their own block (no
    workspace: rounds of 2048 positions) or spread over 64 blocks by the helper launch (workspace); the workspace comes
    back zeroed and serves the next call."""
    V, d, B, L = 211, 256, 8, 1500
    g = torch.Generator().manual_seed(12)
    tok = torch.randint(1, V, (B, L), generator=g)
    u = torch.rand(B, L, generator=g)
    tok[u < p_top] = 17                                     # 0.85: more than half of every scanned slice (list overflow paths)
    tok[(u >= p_top) & (u < p_top + 0.12)] = 101
    tok[:, -7:] = 0
    dy = rnd(B, L, d, seed=33).to(dtype)
    ref = torch.zeros(V, d, dtype=torch.float64)
    ref.index_add_(0, tok.flatten(), dy.double().view(-1, d) * (d ** 0.5))
    ref[0] = 0
    ws = ops.embed_bwd_ws(DEV) if use_ws else None
    for rep_ in range(2):                                   # the second call reuses the workspace the first one left behind
        g_emb = torch.zeros(V, d, device=DEV)
        ops.embed_bwd(dy.to(DEV), tok.to(DEV), None, g_emb, None, None, None, None, ops.ME_COND_NONE, B, L, d, 0, 0, 0.0, 0, ws=ws)
        assert relerr(g_emb, ref) < tol(dtype, 1e-5, dy), (use_ws, rep_)
        assert (g_emb[0] == 0).all()
        if ws is not None:
            torch.cuda.synchronize()
            assert int(ws.view(torch.int32)[:2].abs().sum()) == 0
    # continuous_token layout: two condition slots in front of every sequence (row index = b * (L + 2) + l + 2)
    dy2 = rnd(B, L + 2, d, seed=34).to(dtype)
    ref2 = torch.zeros(V, d, dtype=torch.float64)
    ref2.index_add_(0, tok.flatten(), dy2[:, 2:].double().reshape(-1, d) * (d ** 0.5))
    ref2[0] = 0
    cond = torch.rand(B, 2, generator=g) * 2 - 1
    g_emb = torch.zeros(V, d, device=DEV)
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    ops.embed_bwd(dy2.to(DEV), tok.to(DEV), cond.to(DEV), g_emb, z(d, 1), z(d), z(d, 1), z(d), ops.ME_COND_TOKEN, B, L, d, 0, 0, 0.0, 0, ws=ws)
    assert relerr(g_emb, ref2) < tol(dtype, 1e-5, dy2), ("continuous_token", use_ws)


def test_key_pad_mask(ops):
    tok = torch.tensor([[5, 0, 7, 0], [0, 1, 2, 3]])
    kp = torch.empty(2, 6, dtype=torch.uint8, device=DEV)
    ops.key_pad_mask(kp, tok.to(DEV), 2, 4, 2, 0)
    assert kp.cpu().tolist() == [[0, 0, 0, 1, 0, 1], [0, 0, 1, 0, 0, 0]]


# ------------------------------------------------------------------ relative global attention
def attn_case(B, H, L, dh, M, seed, pad_rows=True):
    q = rnd(B, H, L, dh, seed=seed)
    k = rnd(B, H, L, dh, seed=seed + 1)
    v = rnd(B, H, L, dh, seed=seed + 2)
    E = rnd(M, dh, seed=seed + 3)
    dO = rnd(B, H, L, dh, seed=seed + 4)
    pad = torch.zeros(B, L, dtype=torch.bool)
    if pad_rows and L > 8:
        pad[-1, -(L // 5):] = True
        if B > 1:
            pad[0, 3] = True
    return q, k, v, E, dO, pad


def run_attn(ops, dtype, q, k, v, E, dO, pad, backward=True, causal=True):
    B, H, L, dh = q.shape
    M = E.shape[0]
    Lp = ((L + 31) // 32) * 32
    to_tok = lambda t: t.permute(0, 2, 1, 3)          # [B,L,H,dh]
    qkv = torch.stack([to_tok(q), to_tok(k), to_tok(v)], dim=2).contiguous().to(dtype).to(DEV)   # [B,L,3,H,dh]
    Ed = E.to(dtype).to(DEV).contiguous()
    kp = pad.to(torch.uint8).to(DEV) if pad is not None else None
    out = torch.full((B, L, H, dh), float("nan"), dtype=dtype, device=DEV)
    lse = torch.empty(B, H, L, dtype=torch.float32, device=DEV)
    Epk = ops.rga_pack_rel(Ed)
    PT = MT = None
    if backward:                                        # training mode: the forward leaves its probability tiles + running maxima
        PT, MT = ops.rga_saved_buffers(B, H, L, dtype, DEV, causal=causal)
        PT.fill_(float("nan"))                          # no initialisation contract: every tile read was written before
        MT.fill_(float("nan"))
        out_inf = torch.full_like(out, float("nan"))
        lse_inf = torch.empty_like(lse)
        ops.rga_fwd(qkv, Epk, kp, out_inf, lse_inf, B, L, H, dh, M, causal=causal)       # inference mode: same outputs
    ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, causal=causal, PT=PT, MT=MT)
    if backward:
        assert torch.equal(out.nan_to_num(7.0), out_inf.nan_to_num(7.0)) and torch.equal(lse.nan_to_num(7.0), lse_inf.nan_to_num(7.0))
    res = {"O": out.permute(0, 2, 1, 3).float().cpu(), "lse": lse.cpu()}
    if backward:
        dout = to_tok(dO).contiguous().to(dtype).to(DEV)
        dqkv = torch.full_like(qkv, float("nan"))
        dE = torch.zeros(M, dh, dtype=torch.float32, device=DEV)
        delta = torch.empty(B, H, L, dtype=torch.float32, device=DEV)
        dGT = ops.rga_bwd_workspace(B, H, L, dtype, DEV)
        dGT.fill_(float("nan"))
        ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dGT, B, L, Lp, H, dh, M, causal=causal)
        g = dqkv.float().cpu().permute(2, 0, 3, 1, 4)   # [3,B,H,L,dh]
        res.update(dq=g[0], dk=g[1], dv=g[2], dE=dE.cpu())
    return res


def ref_attn(q, k, v, E, dO, pad, dtype, causal=True):
    """fp64 oracle on inputs rounded to the storage dtype."""
    r = lambda t: t.to(dtype).double().requires_grad_(True)
    q, k, v, E = r(q), r(k), r(v), r(E)
    o, lse = O.rga_attention_core(q, k, v, E, pad, causal)
    (o * dO.to(dtype).double()).sum().backward()
    return {"O": o.detach(), "lse": lse.detach(), "dq": q.grad, "dk": k.grad, "dv": v.grad, "dE": E.grad}


K_ATTN = 1.5      # the kernels round P and dS to bf16 once more than autocast's matmuls do on some paths (measured ratios < 1 mostly)


def attn_bounds(dtype, f32, q, k, v, E, dO, pad, causal=True, backward=True):
    """Per-output bound.  f32 tier: the given number.  bf16 tier: K_ATTN x what the ORACLE's own arithmetic loses on these
    inputs under torch.autocast(bfloat16) (bf16 matmuls, f32 softmax -- the reference's --amp path, train.py:281) against
    its f64 run, + 1e-4 -- derived per case instead of one hand-picked 1.5e-2."""
    names = ("O", "lse", "dq", "dk", "dv", "dE") if backward else ("O", "lse")
    if dtype == torch.float32:
        return {n: f32 for n in names}
    ref = ref_attn(q, k, v, E, dO, pad, dtype, causal) if backward else None
    r = lambda t: t.to(dtype).float().requires_grad_(backward)
    q32, k32, v32, E32 = r(q), r(k), r(v), r(E)
    with torch.autocast("cpu", dtype=dtype):
        o, lse = O.rga_attention_core(q32, k32, v32, E32, pad, causal)
    if backward:
        (o.float() * dO.to(dtype).float()).sum().backward()
        ac = {"O": o.detach().float(), "lse": lse.detach().float(), "dq": q32.grad, "dk": k32.grad, "dv": v32.grad, "dE": E32.grad}
    else:
        rr = lambda t: t.to(dtype).double()
        o64, lse64 = O.rga_attention_core(rr(q), rr(k), rr(v), rr(E), pad, causal)
        ref = {"O": o64, "lse": lse64}
        ac = {"O": o.detach().float(), "lse": lse.detach().float()}
    out = {}
    for n in names:
        a_, r_ = ac[n].double(), ref[n].double()
        ok = ~(torch.isnan(a_) | torch.isnan(r_))
        out[n] = K_ATTN * relerr(a_[ok], r_[ok]) + 1e-4
    return out


def attn_ok(errs, lims):
    return all(errs[n] <= lims[n] for n in errs)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rga_golden_f5(ops, dtype, golden_dir):
    z = np.load(os.path.join(golden_dir, "f5_attn_core.npz"))
    q, k, v, E, dO = (torch.from_numpy(z[n]) for n in ("q", "k", "v", "E", "dO"))
    pad = torch.from_numpy(z["pad"])
    got = run_attn(ops, dtype, q, k, v, E, dO, pad)
    if dtype == torch.float32:
        errs = {n: relerr(got[n], torch.from_numpy(z[n])) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
        assert all(e < 2e-5 for e in errs.values()), errs
    else:
        ref = ref_attn(q, k, v, E, dO, pad, dtype)
        errs = {n: relerr(got[n], ref[n]) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
        lims = attn_bounds(dtype, 0, q, k, v, E, dO, pad)
        assert attn_ok(errs, lims), (errs, lims)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,L,dh,M", [(2, 2, 1, 32, 64), (1, 3, 7, 64, 64), (2, 2, 33, 32, 64), (2, 4, 64, 64, 64),
                                         (1, 2, 130, 64, 256), (2, 2, 256, 64, 2048), (1, 1, 300, 32, 512),
                                         (2, 3, 70, 48, 128), (1, 2, 260, 48, 2048), (2, 2, 1, 48, 64)])   # dh 48: published checkpoints
def test_rga_fwd_bwd_shapes(ops, dtype, B, H, L, dh, M):
    q, k, v, E, dO, pad = attn_case(B, H, L, dh, M, seed=100 + L)
    got = run_attn(ops, dtype, q, k, v, E, dO, pad)
    ref = ref_attn(q, k, v, E, dO, pad, dtype)
    errs = {n: relerr(got[n], ref[n]) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
    lims = attn_bounds(dtype, 3e-5, q, k, v, E, dO, pad)
    assert attn_ok(errs, lims), (errs, lims)
    # rows of E that can never be reached (relative distance >= L) get exactly zero gradient
    if M > L:
        assert (got["dE"][: M - L] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L,M", [(512, 2048), (448, 512), (200, 256)])
def test_rga_long_unpadded_and_mixed_rows(ops, dtype, L, M):
    """Sequences long enough for the mask-free MAIN steps of the 64-key-step kernels (bf16, dh 64): batch row 0 has no
    padded key (MAIN steps + the diagonal / ragged tail), batch row 1 has padded keys (general path only)."""
    q, k, v, E, dO, _ = attn_case(2, 2, L, 64, M, seed=300 + L, pad_rows=False)
    pad = torch.zeros(2, L, dtype=torch.bool)
    pad[1, 5] = True
    pad[1, -(L // 7):] = True
    got = run_attn(ops, dtype, q, k, v, E, dO, pad)
    ref = ref_attn(q, k, v, E, dO, pad, dtype)
    errs = {n: relerr(got[n], ref[n]) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
    lims = attn_bounds(dtype, 3e-5, q, k, v, E, dO, pad)
    assert attn_ok(errs, lims), (errs, lims)
    got0 = run_attn(ops, dtype, q, k, v, E, dO, None)                  # no pad mask at all
    nopad = torch.zeros(2, L, dtype=torch.bool)
    ref0 = ref_attn(q, k, v, E, dO, nopad, dtype)
    errs = {n: relerr(got0[n], ref0[n]) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
    lims = attn_bounds(dtype, 3e-5, q, k, v, E, dO, nopad)
    assert attn_ok(errs, lims), (errs, lims)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rga_full_size_replication_property(ops, dtype):
    """BASELINE config 2 at FULL size (B = 32, H = 8, L = 1024, dh = 64, M = 1024) without a full-size oracle run: the batch is a
    2-sequence case (oracle-checked by the tests above at this L) replicated 16 times, one sequence with trailing PAD.
    Every (sequence, head) is computed independently, so O, lse, dq, dk, dv of every replica must be BIT-identical to the
    2-sequence launch of the same kernels (grid mapping, workspace indexing and tile bookkeeping at the full grid), and dE --
    a sum over the batch -- must be 16 x the small launch's dE up to f32 summation order."""
    B0, R, H, L, dh, M = 2, 16, 8, 1024, 64, 1024
    q, k, v, E, dO, _ = attn_case(B0, H, L, dh, M, seed=77, pad_rows=False)
    pad = torch.zeros(B0, L, dtype=torch.bool)
    pad[1, -100:] = True
    small = run_attn(ops, dtype, q, k, v, E, dO, pad)
    rep = lambda t: t.repeat(R, *([1] * (t.dim() - 1)))
    full = run_attn(ops, dtype, rep(q), rep(k), rep(v), E, rep(dO), rep(pad))
    for n in ("O", "lse", "dq", "dk", "dv"):
        a, b = full[n].nan_to_num(7.0), rep(small[n]).nan_to_num(7.0)
        assert torch.equal(a, b), (n, (a - b).abs().max().item())
    assert relerr(full["dE"], R * small["dE"]) < 1e-5, relerr(full["dE"], R * small["dE"])
    ok = ~pad[:, None, :, None].expand(B0, H, L, dh)                        # spot check of the small launch against the oracle's f64 run
    ref = ref_attn(q[:, :1], k[:, :1], v[:, :1], E, dO[:, :1], pad, dtype)
    lim = 3e-5 if dtype == torch.float32 else 2e-2
    assert relerr(small["O"][:, :1][ok[:, :1]], ref["O"][ok[:, :1]]) < lim


def test_rga_fully_masked_row_is_nan(ops):
    """PAD at key 0: query 0 has no valid key -> NaN (reference behaviour, SURVEY hard part 6);
    later queries with leading masked keys must stay finite."""
    q, k, v, E, dO, _ = attn_case(1, 1, 40, 32, 64, seed=5, pad_rows=False)
    pad = torch.zeros(1, 40, dtype=torch.bool)
    pad[0, 0] = True
    got = run_attn(ops, torch.float32, q, k, v, E, dO, pad, backward=False)
    ref, _ = O.rga_attention_core(q, k, v, E, pad)
    assert torch.isnan(got["O"][0, 0, 0]).all()
    assert relerr(got["O"][0, 0, 1:], ref[0, 0, 1:]) < 2e-5
    # whole first key tile masked, later keys valid
    pad = torch.zeros(1, 40, dtype=torch.bool)
    pad[0, :34] = True
    got = run_attn(ops, torch.float32, q, k, v, E, dO, pad, backward=False)
    ref, _ = O.rga_attention_core(q, k, v, E, pad)
    assert torch.isnan(got["O"][0, 0, :34]).all()
    assert relerr(got["O"][0, 0, 34:], ref[0, 0, 34:]) < 2e-5


def test_rga_prefix_invariance(ops):
    """logits at position t do not depend on later tokens nor on L."""
    q, k, v, E, dO, _ = attn_case(1, 2, 96, 64, 2048, seed=9, pad_rows=False)
    full = run_attn(ops, torch.float32, q, k, v, E, dO, None, backward=False)["O"]
    part = run_attn(ops, torch.float32, q[:, :, :50], k[:, :, :50], v[:, :, :50], E, dO[:, :, :50], None, backward=False)["O"]
    assert relerr(full[:, :, :50], part) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dh,nsplit", [(64, 1), (64, 8), (48, 3), (32, 5)])
def test_dec_attn_matches_full(ops, dtype, dh, nsplit):
    """me_dec_attn (key-split partials) + the combine prologue of me_dec_proj_resid == row t of the full attention,
    for every t (short ranges, empty splits, ragged last split).  The combine is read back through an identity
    projection (W = I, no bias, zero residual)."""
    B, H, L, M = 3, 2, 45, 2048
    q, k, v, E, dO, pad = attn_case(B, H, L, dh, M, seed=77, pad_rows=False)
    ref = ref_attn(q, k, v, E, dO, None, dtype)["O"]            # [B,H,L,dh]
    lim_o = attn_bounds(dtype, 2e-5, q, k, v, E, dO, None, backward=False)["O"]
    Ed = E.to(dtype).to(DEV).contiguous()
    Mc = 64
    d = H * dh
    kc = torch.zeros(B, H, Mc, dh, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    part = torch.zeros(B * H, nsplit, dh + 4, dtype=torch.float32, device=DEV)      # ME_DEC_PART_REC(dh)
    eye = torch.eye(d, dtype=dtype, device=DEV)
    zero = torch.zeros(B, d, dtype=torch.float32, device=DEV)
    out = torch.empty(B, d, dtype=torch.float32, device=DEV)
    t_dev = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in range(L):
        kc[:, :, t] = k[:, :, t].to(dtype).to(DEV)
        vc[:, :, t] = v[:, :, t].to(dtype).to(DEV)
        qt = q[:, :, t].reshape(B, d).contiguous().to(dtype).to(DEV)
        if t % 2:
            ops.dec_attn(qt, kc, vc, Ed, None, 0, part, nsplit, B, H, dh, M, Mc, t, None, dtype)
        else:                                                   # position from device memory (graph replay path)
            t_dev.fill_(t)
            ops.dec_attn(qt, kc, vc, Ed, None, 0, part, nsplit, B, H, dh, M, Mc, 0, t_dev, dtype)
        ops.dec_proj_resid(part, nsplit, H, dh, None, eye, None, zero, out, B, d, d, dtype)
        e = relerr(out.view(B, H, dh), ref[:, :, t])
        assert e <= (lim_o if t >= 8 else 4 * lim_o), (t, e, lim_o)      # one row of very few keys: 4 x the whole-tensor figure


def test_dec_attn_pad_keys_and_fully_masked_row(ops):
    """Padded keys are excluded; a query whose every key is padded yields NaN like the reference's softmax."""
    B, H, dh, M, Mc, t, ns = 2, 2, 64, 2048, 32, 11, 4
    q, k, v, E, dO, _ = attn_case(B, H, t + 1, dh, M, seed=5, pad_rows=False)
    pad = torch.zeros(B, t + 1, dtype=torch.uint8)
    pad[0, 3:6] = 1
    pad[1, :] = 1
    ref = ref_attn(q, k, v, E, dO, pad.bool(), torch.float32)["O"]
    kc = torch.zeros(B, H, Mc, dh, device=DEV)
    vc = torch.zeros_like(kc)
    kc[:, :, :t + 1] = k.float().to(DEV)
    vc[:, :, :t + 1] = v.float().to(DEV)
    d = H * dh
    part = torch.zeros(B * H, ns, dh + 4, device=DEV)      # ME_DEC_PART_REC(dh)
    out = torch.empty(B, d, device=DEV)
    ops.dec_attn(q[:, :, t].reshape(B, d).contiguous().float().to(DEV), kc, vc, E.float().to(DEV), pad.to(DEV), t + 1, part, ns, B, H,
                 dh, M, Mc, t, None, torch.float32)
    ops.dec_proj_resid(part, ns, H, dh, None, torch.eye(d, device=DEV), None, torch.zeros(B, d, device=DEV), out, B, d, d,
                       torch.float32)
    assert relerr(out[0].view(H, dh), ref[0, :, t]) < 2e-5
    assert torch.isnan(out[1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dh", [32, 48, 64])
def test_rel_pack_layout_and_refresh_path(ops, dtype, dh):
    """me_rga_pack_rel against its definition (include/midiemo.h) and against the multi-tensor refresh (mode 1), bit-exact."""
    M = 96
    g = torch.Generator().manual_seed(5)
    E32 = torch.randn(M, dh, generator=g)
    Ed = E32.to(dtype).to(DEV)
    pk = ops.rga_pack_rel(Ed).cpu()
    KA, DB = dh // 16, (dh + 31) // 32
    ref = torch.zeros(M // 32, KA + 2 * DB, 64, 8, dtype=dtype)
    Eh = Ed.cpu()
    for eb in range(M // 32):
        for lane in range(64):
            a, h = lane & 31, lane >> 5
            for kk in range(KA):
                ref[eb, kk, lane] = Eh[eb * 32 + a, kk * 16 + h * 8: kk * 16 + h * 8 + 8]
            for i in range(DB):
                for t in range(2):
                    if i * 32 + a < dh:
                        ref[eb, KA + 2 * i + t, lane] = Eh[eb * 32 + 16 * t + 8 * h: eb * 32 + 16 * t + 8 * h + 8, i * 32 + a]
    assert torch.equal(pk.view_as(ref), ref)
    # refresh path: f32 master -> (cast copy, packed images) in one multi-tensor launch
    src = E32.to(DEV).contiguous()
    dst = torch.zeros(M, dh, dtype=dtype, device=DEV) if dtype != torch.float32 else None
    pk2 = torch.zeros(ops.rel_pack_numel(M, dh), dtype=dtype, device=DEV)
    desc, n, tiles = ops.make_ct_desc([(src, dst, pk2, 1)], DEV)
    ops.cast_transpose_multi(desc, n, tiles, dtype)
    assert torch.equal(pk2.cpu(), pk)
    if dst is not None:
        assert torch.equal(dst.cpu(), Eh)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_cast_transpose_multi_shapes(ops, dtype):
    """One launch over tensors of every kind the walk has to handle: vector path (multiples of 64), ragged edges
    (1007 rows, 77 columns), odd column counts (element path), padded leading dimensions, copy-only and transpose-only
    descriptors, a tensor smaller than one supertile.  Bit-exact against torch's cast / transpose."""
    g = torch.Generator().manual_seed(17)
    shapes = [(512, 512), (1007, 512), (512, 2048), (130, 77), (3, 5), (64, 200), (96, 64), (1, 512)]
    items, want = [], []
    for i, (r, c) in enumerate(shapes):
        src = torch.randn(r, c, generator=g).to(DEV)
        ld = c + (8 if i % 2 else 0)
        ldT = (r + 7) // 8 * 8 + (8 if i % 3 == 0 else 0)
        dst = torch.full((r, ld), 7.0, dtype=dtype, device=DEV)[:, :c] if i != 5 else None
        dstT = torch.full((c, ldT), 7.0, dtype=dtype, device=DEV)[:, :r] if i != 6 else None
        if i == 3:                                   # odd leading dimension as well: every access takes the element path
            dstT = torch.full((c, r + 1), 7.0, dtype=dtype, device=DEV)[:, :r]
        items.append((src, dst, dstT))
        want.append((src.to(dtype), src.t().to(dtype)))
    desc, n, tiles = ops.make_ct_desc(items, DEV)
    ops.cast_transpose_multi(desc, n, tiles, dtype)
    torch.cuda.synchronize()
    for (src, dst, dstT), (w, wT), shp in zip(items, want, shapes):
        if dst is not None:
            assert torch.equal(dst, w), shp
            assert (dst.as_strided((dst.shape[0], dst.stride(0)), (dst.stride(0), 1))[:, dst.shape[1]:] == 7).all(), shp   # padding untouched
        if dstT is not None:
            assert torch.equal(dstT, wT), shp
            assert (dstT.as_strided((dstT.shape[0], dstT.stride(0)), (dstT.stride(0), 1))[:, dstT.shape[1]:] == 7).all(), shp


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,L,dh,M", [(2, 2, 1, 32, 64), (1, 3, 7, 64, 64), (2, 2, 33, 32, 64), (1, 2, 130, 64, 256),
                                         (2, 2, 256, 64, 2048), (1, 1, 300, 48, 512), (2, 4, 520, 64, 2048)])
def test_rga_fwd_bidirectional(ops, dtype, B, H, L, dh, M):
    """causal = 0 (MusicRegression, mask = None): every key attended, relative term only on / below the diagonal;
    with and without padded keys; against the fp64 oracle."""
    q, k, v, E, dO, pad = attn_case(B, H, L, dh, M, seed=300 + L)
    for use_pad in (None, pad):
        got = run_attn(ops, dtype, q, k, v, E, dO, use_pad, backward=False, causal=False)
        r = lambda t: t.to(dtype).double()
        o, lse = O.rga_attention_core(r(q), r(k), r(v), r(E), use_pad, causal=False)
        errs = {"O": relerr(got["O"], o), "lse": relerr(got["lse"], lse)}
        lims = attn_bounds(dtype, 3e-5, q, k, v, E, dO, use_pad, causal=False, backward=False)
        assert attn_ok(errs, lims), (errs, lims, use_pad is not None)
    # and it really differs from the causal result once there is more than one key
    if L > 1:
        c = run_attn(ops, dtype, q, k, v, E, dO, None, backward=False, causal=True)
        assert relerr(c["O"], o) > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,L,dh,M", [(2, 2, 1, 32, 64), (1, 3, 7, 64, 64), (2, 2, 33, 32, 64), (1, 2, 130, 64, 256),
                                         (2, 2, 256, 64, 2048), (1, 1, 300, 48, 512), (1, 2, 520, 64, 2048)])
def test_rga_bwd_bidirectional(ops, dtype, B, H, L, dh, M):
    """Backward of the bidirectional attention (causal = 0): dq, dk, dv, dE against the fp64 oracle's autograd, with
    and without padded keys; the P^T / dS^T workspaces start as garbage (every tile is written before it is read)."""
    q, k, v, E, dO, pad = attn_case(B, H, L, dh, M, seed=400 + L)
    for use_pad in (None, pad):
        got = run_attn(ops, dtype, q, k, v, E, dO, use_pad, causal=False)
        ref = ref_attn(q, k, v, E, dO, use_pad, dtype, causal=False)
        errs = {n: relerr(got[n], ref[n]) for n in ("O", "lse", "dq", "dk", "dv", "dE")}
        lims = attn_bounds(dtype, 3e-5, q, k, v, E, dO, use_pad, causal=False)
        assert attn_ok(errs, lims), (errs, lims, use_pad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,B,H,L,dh", [(torch.bfloat16, 4, 8, 1024, 64), (torch.float16, 2, 8, 1024, 64), (torch.bfloat16, 2, 2, 300, 64), (torch.float32, 2, 2, 200, 32)])
def test_rga_bwd_phases_on_two_streams_equal_the_single_call(ops, dtype, B, H, L, dh):
    """me_rga_bwd_phases (round 5): the backward kernel by kernel.  (1) key-owned and E-row-owned kernels side by side on two
    streams, (2) delta as its own launch and the key-owned kernel beside the query-owned one -- dQ / dK / dV must be BIT-identical
    to me_rga_bwd (every element has one writer and the same arithmetic), dE equal up to the order of its f32 atomics."""
    M = 2048
    Lp = ((L + 31) // 32) * 32
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B, L, 3, H, dh, generator=g) * 0.7).to(dtype).to(DEV)
    E = (torch.randn(M, dh, generator=g) * 0.5).to(dtype).to(DEV)
    dout = torch.randn(B, L, H, dh, generator=g).to(dtype).to(DEV)
    kp = torch.zeros(B, L, dtype=torch.uint8, device=DEV)
    kp[0, L - 7:] = 1
    Epk = ops.rga_pack_rel(E)
    out = torch.empty(B, L, H, dh, device=DEV, dtype=dtype)
    lse = torch.empty(B, H, L, device=DEV)
    PT, MT = ops.rga_saved_buffers(B, H, L, dtype, DEV)
    dGT = ops.rga_bwd_workspace(B, H, L, dtype, DEV)
    ops.rga_fwd(qkv, Epk, kp, out, lse, B, L, H, dh, M, PT=PT, MT=MT)
    res = {}
    for ov in (False, True, 2):
        dqkv = torch.full_like(qkv, float("nan"))
        dE = torch.zeros(M, dh, device=DEV)
        delta = torch.full((B, H, L), float("nan"), device=DEV)
        ops.rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta, PT, MT, dGT, B, L, Lp, H, dh, M, overlap=ov)
        torch.cuda.synchronize()
        res[ov] = (dqkv, dE, delta)
    it = torch.int16 if dtype != torch.float32 else torch.int32
    for ov in (True, 2):
        assert torch.equal(res[ov][0].view(it), res[False][0].view(it)), ov
        assert torch.equal(res[ov][2], res[False][2]), ov                      # delta: same lanes, same order of the multiply-adds
        assert float((res[ov][1] - res[False][1]).abs().max()) <= 2e-5 * float(res[False][1].abs().max()), ov
    assert torch.isfinite(res[False][0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (777, 1024, 512), (2050, 2048, 512), (1024, 192, 128), (32768, 2048, 512)])
@pytest.mark.parametrize("dtype", DT16)
def test_gemm_nt_relu_mask_equals_the_gate_operand(ops, M, N, K, dtype):
    """me_gemm_nt_relu_mask (round 5): the FFN_pre forward leaves the ReLU's sign pattern as a bit mask and the FFN_suf dgrad
    applies it -- both launches must agree BIT for bit with me_gemm_nt (ME_EPI_RELU / gate = the activations + ME_EPI_RELU_BWD),
    ragged M included; the padding columns of C stay untouched; shapes the 256-tile kernel does not serve answer 0 bytes."""
    nbytes = ops.workspace_bytes(ops.ME_WS_RELU_MASK, M, N, K, dtype)
    if N % 64:
        assert nbytes == 0
        return
    assert nbytes == ((M + 255) // 256) * 32 * (N // 64) * 64
    A = rnd(M, K, seed=41).to(dtype).to(DEV)
    W1 = rnd(N, K, seed=42).to(dtype).to(DEV)
    bias = rnd(N, seed=43).float().to(DEV)
    ld = N + 16
    hid_ref = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(A, W1, hid_ref, bias=bias, flags=ops.ME_EPI_RELU, N=N)
    hid = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    mask = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=DEV)
    ops.gemm_nt_relu_mask(A, W1, hid, mask, bias=bias, N=N)
    assert torch.equal(hid[:, :N].view(torch.int16), hid_ref[:, :N].view(torch.int16)) and torch.isnan(hid[:, N:]).all()
    assert 0.2 < float((hid_ref[:, :N] > 0).float().mean()) < 0.8           # the mask has both values to get wrong
    # backward: dhid = (dC . W2T^T) where hid > 0
    dC = rnd(M, K, seed=44).to(dtype).to(DEV)
    W2T = rnd(N, K, seed=45).to(dtype).to(DEV)
    d_ref = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt(dC, W2T, d_ref, gate=hid_ref, flags=ops.ME_EPI_RELU_BWD, N=N)
    d = torch.full((M, ld), float("nan"), dtype=dtype, device=DEV)
    ops.gemm_nt_relu_mask(dC, W2T, d, mask, N=N, backward=True)
    assert torch.equal(d[:, :N].view(torch.int16), d_ref[:, :N].view(torch.int16)) and torch.isnan(d[:, N:]).all()
    dref = (dC.double() @ W2T.double().t()).cpu() * (hid_ref[:, :N] > 0).cpu().double()
    assert relerr(d[:, :N], dref) < tol(dtype, 0, dref)


@pytest.mark.gpu
def test_gemm_nt_relu_mask_refuses_what_it_does_not_serve(ops):
    dtype = torch.bfloat16
    assert ops.workspace_bytes(ops.ME_WS_RELU_MASK, 4096, 1007, 512, dtype) == 0            # N % 64
    assert ops.workspace_bytes(ops.ME_WS_RELU_MASK, 128, 2048, 512, dtype) == 0             # below the 256-tile kernel's shapes
    assert ops.workspace_bytes(ops.ME_WS_RELU_MASK, 4096, 2048, 512, torch.float32) == 0    # f32 tier keeps the gate
    A, W = rnd(128, 512, seed=1).to(dtype).to(DEV), rnd(2048, 512, seed=2).to(dtype).to(DEV)
    C = torch.empty(128, 2048, dtype=dtype, device=DEV)
    mask = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError):
        ops.gemm_nt_relu_mask(A, W, C, mask)


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["bf16", "fp16"])
@pytest.mark.parametrize("d", [512, 768])
def test_resid_ln_and_embed_with_8_bit_low_halves(cd, d):
    """ME_LO8: the residual stream's low half as ONE byte per element, q = round((v - hi) / (ulp(hi) / 256)).  The high halves, the
    pre-norm sums and the statistics must EQUAL those of the 16-bit form wherever the inputs agree, and hi + decode(q) must carry the
    stream to 2^-(MB + 8) of its magnitude (MB = 7 / 10 stored mantissa bits): checked against an f64 LayerNorm on the decoded stream,
    for the one-chunk-per-lane kernel (d = 512) and the generic one (d = 768), and for the embedding prologue."""
    import torch
    from midiemo import ops
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[cd]
    mb = {"bf16": 7, "fp16": 10}[cd]
    torch.manual_seed(3)
    rows = 777
    dev = "cuda"

    def decode(hi, q):                                  # the kernel's lo8_dec in torch: step = 2^(exponent(hi) - MB - 8)
        e = torch.floor(torch.log2(hi.double().abs().clamp_min(1e-300)))
        return hi.double() + (q.view(torch.int8).double() * torch.pow(2.0, e - mb - 8)) * (hi != 0)

    xf = torch.randn(rows, d, device=dev, dtype=torch.float64) * 1.7
    x_hi = xf.to(dt)
    # the exact 8-bit low half of xf
    e = torch.floor(torch.log2(x_hi.double().abs().clamp_min(1e-300)))
    q = torch.round((xf - x_hi.double()) / torch.pow(2.0, e - mb - 8)).clamp(-127, 127).to(torch.int8).view(torch.uint8)
    x_dec = decode(x_hi, q)
    a = (torch.randn(rows, d, device=dev) * 0.8).to(dt)
    gamma, beta = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev) * 0.1
    y, s_out, st = torch.empty(rows, d, device=dev, dtype=dt), torch.empty(rows, d, device=dev, dtype=dt), torch.empty(rows, 2, device=dev)
    y_q = torch.zeros(rows, d, device=dev, dtype=torch.uint8)
    ops.resid_ln_fwd(x_hi, a, gamma, beta, y, s_out, st, rows, d, 1e-6, 0.0, 0, 1, x_lo=q, y_lo=y_q)
    s_ref = x_dec + a.double()
    mu, var = s_ref.mean(-1, keepdim=True), s_ref.var(-1, unbiased=False, keepdim=True)
    y_ref = (s_ref - mu) / torch.sqrt(var + 1e-6) * gamma.double() + beta.double()
    y_dec = decode(y, y_q)
    ulp = 2.0 ** -(mb + 8)
    # the decoded output stream follows the f64 LayerNorm to the f32 arithmetic of the kernel + one step of the 8-bit code
    err = ((y_dec - y_ref).abs() / y_ref.abs().clamp_min(0.25)).max().item()
    assert err < 4 * ulp + 3e-6, (err, ulp)
    # and the hi half alone is the rounding of that stream (what the next GEMM reads)
    assert ((y.double() - y_ref).abs() / y_ref.abs().clamp_min(0.25)).max().item() < 2.0 ** -(mb + 1) * 1.01 + 3e-6
    # same call with 16-bit low halves holding the SAME stream (x_lo16 = the decoded low half, exactly representable): equal hi / sums / statistics
    # (bf16 only: a code step 2^(E - 18) of small f16 values lies below f16's own denormal step)
    if cd == "bf16":
        x_lo16 = (x_dec - x_hi.double()).to(dt)
        assert torch.equal((x_hi.double() + x_lo16.double()), x_dec)
        y2, s2, st2, ylo2 = torch.empty_like(y), torch.empty_like(s_out), torch.empty_like(st), torch.empty_like(y)
        ops.resid_ln_fwd(x_hi, a, gamma, beta, y2, s2, st2, rows, d, 1e-6, 0.0, 0, 1, x_lo=x_lo16, y_lo=ylo2)
        assert torch.equal(y, y2) and torch.equal(s_out, s2) and torch.equal(st, st2)
    if d == 512:
        # embedding prologue: hi equal to the 16-bit form, decoded stream within one code step of hi + lo16
        B, L, V = 3, 37, 1007
        tok = torch.randint(1, V, (B, L), device=dev)
        emb = (torch.rand(V, d, device=dev) - 0.5) * 0.2
        pe = torch.randn(64, d, device=dev)
        o8, o16, q8, l16 = torch.empty(B * L, d, device=dev, dtype=dt), torch.empty(B * L, d, device=dev, dtype=dt), \
            torch.zeros(B * L, d, device=dev, dtype=torch.uint8), torch.empty(B * L, d, device=dev, dtype=dt)
        ops.embed_fwd(o8, tok, None, emb, None, None, None, None, pe, ops.ME_COND_NONE, B, L, d, 0, 0.0, 0, out_lo=q8)
        ops.embed_fwd(o16, tok, None, emb, None, None, None, None, pe, ops.ME_COND_NONE, B, L, d, 0, 0.0, 0, out_lo=l16)
        assert torch.equal(o8, o16)
        full = o16.double() + l16.double()
        assert ((decode(o8, q8) - full).abs() / full.abs().clamp_min(0.25)).max().item() < 2 * ulp
