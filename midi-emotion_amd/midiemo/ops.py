"""Tensor-level wrappers over the C-ABI (one Python function per entry point).

Every wrapper takes torch CUDA tensors, passes raw device pointers and the
current HIP stream, and raises RuntimeError on any non-zero status.  Nothing
here computes on the host."""
import ctypes

import torch

from . import _lib
from ._lib import (ME_BF16, ME_F16, ME_COND_CONCAT, ME_COND_NONE, ME_COND_TOKEN, ME_EPI_OUT_F32, ME_EPI_RELU,
                   ME_EPI_RELU_BWD, ME_F32, ME_TN_MAX_GROUP, ME_WS_EMBED_BWD, ME_WS_SUMSQ, ME_WS_GEMM_TN, ME_WS_GEMM_TN_GROUP, ME_WS_RGA_DGT, ME_WS_RGA_MT,
                   ME_WS_RGA_PT, ME_WS_RELU_MASK, ME_WS_DEC_TOKEN, check)

DTYPE_CODE = {torch.float32: ME_F32, torch.bfloat16: ME_BF16, torch.float16: ME_F16}


def lib():
    return _lib.load()


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("midiemo ops need device (HIP) tensors; got a %s tensor -- there is no CPU path" % t.device)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _code(dtype):
    try:
        return DTYPE_CODE[dtype]
    except KeyError:
        raise RuntimeError("unsupported compute dtype %s (float32, bfloat16 or float16)" % dtype)


def cast_transpose(src, dst, dstT, dtype):
    rows, cols = src.shape
    check(lib().me_cast_transpose(_ptr(src), rows, cols, _ptr(dst), dst.stride(0) if dst is not None else 0,
                                  _ptr(dstT), dstT.stride(0) if dstT is not None else 0, _code(dtype), _stream()),
          "me_cast_transpose")


def cast_transpose_multi(desc_dev, n_tensors, total_tiles, dtype):
    """desc_dev: uint8 device tensor holding n_tensors packed me_ct_desc structs (see make_ct_desc)."""
    check(lib().me_cast_transpose_multi(_ptr(desc_dev), n_tensors, total_tiles, _code(dtype), _stream()),
          "me_cast_transpose_multi")


def make_ct_desc(items, device):
    """items: list of (src f32 [rows, cols], dst or None, dstT or None[, mode]) -> (desc uint8 device tensor, n,
    total_tiles).  Layout = struct me_ct_desc {const float* src; void* dst; void* dstT; int32 rows, cols, ld_dst,
    ld_dstT, tile_begin, mode;} (48 bytes)."""
    import struct
    blob, tiles = b"", 0
    for it in items:
        src, dst, dstT = it[:3]
        mode = it[3] if len(it) > 3 else 0           # 1 (ME_CT_PACK_REL): dstT = packed relative table (1-D)
        rows, cols = src.shape
        blob += struct.pack("<QQQiiiiii", src.data_ptr(), dst.data_ptr() if dst is not None else 0,
                            dstT.data_ptr() if dstT is not None else 0, rows, cols,
                            dst.stride(0) if dst is not None else 0,
                            dstT.stride(0) if (dstT is not None and mode == 0) else 0, tiles, mode)
        tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    return t, len(items), tiles


def _lo_flag(*los):
    """ME_LO8 when the low halves are byte arrays (uint8 [rows, d]), 0 for the 16-bit form."""
    los = [t for t in los if t is not None]
    if any(t.dtype == torch.uint8 for t in los):
        if not all(t.dtype == torch.uint8 for t in los):
            raise RuntimeError("low halves of the residual stream: 8-bit and 16-bit arrays mixed in one call")
        return _lib.ME_LO8
    return 0


def embed_fwd(out, tokens, cond, emb, cw0, cb0, cw1, cb1, pe, mode, B, Ltok, d, dc, p, seed, pos_dev=None, out_lo=None):
    check(lib().me_embed_fwd(_ptr(out), _ptr(out_lo), _code(out.dtype) | _lo_flag(out_lo), _ptr(tokens), _ptr(cond), _ptr(emb), _ptr(cw0), _ptr(cb0),
                             _ptr(cw1), _ptr(cb1), _ptr(pe), _ptr(pos_dev), mode, B, Ltok, d, dc, float(p), int(seed), _stream()),
          "me_embed_fwd")


def embed_bwd(dout, tokens, cond, g_emb, g_cw0, g_cb0, g_cw1, g_cb1, mode, B, Ltok, d, dc, pad_token, p, seed, ws=None):
    """ws: zero-initialised uint8/int32 device buffer of workspace_bytes(ME_WS_EMBED_BWD) bytes.  Tokens with more than 192
    occurrences in the batch are published there by the gather kernel and summed by the second launch in slices of about
    128 occurrences spread over the chip, instead of by one block.  A call that returns 0 leaves the buffer zeroed; after a
    call that RAISED the caller must zero it again (or drop it) before the next use."""
    vocab = g_emb.shape[0]
    check(lib().me_embed_bwd(_ptr(dout), _code(dout.dtype), _ptr(tokens), _ptr(cond), _ptr(g_emb), _ptr(g_cw0),
                             _ptr(g_cb0), _ptr(g_cw1), _ptr(g_cb1), mode, B, Ltok, d, dc, vocab, pad_token, float(p),
                             int(seed), _ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, _stream()), "me_embed_bwd")


def embed_bwd_ws(device):
    """the (zeroed) frequent-token workspace of embed_bwd"""
    return torch.zeros(workspace_bytes(ME_WS_EMBED_BWD, 0, 0, 0, torch.bfloat16), dtype=torch.uint8, device=device)


def key_pad_mask(key_pad, tokens, B, Ltok, shift, pad_token):
    check(lib().me_key_pad_mask(_ptr(key_pad), _ptr(tokens), B, Ltok, shift, pad_token, _stream()), "me_key_pad_mask")


def gemm_nt(A, B, C, bias=None, add=None, gate=None, M=None, N=None, K=None, flags=0, dtype=None):
    """C[M,N] = A[M,K] . B[N,K]^T (+bias)(relu)(+add)(relu-gate).  2-D row-major views (stride(1) == 1)."""
    dtype = dtype or A.dtype
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    check(lib().me_gemm_nt(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C), C.stride(0), _ptr(bias),
                           _ptr(add), add.stride(0) if add is not None else 0,
                           _ptr(gate), gate.stride(0) if gate is not None else 0,
                           M, N, K, flags, _code(dtype), _stream()), "me_gemm_nt")


def gemm_nt_relu_mask(A, B, C, mask, bias=None, M=None, N=None, K=None, backward=False, dtype=None):
    """forward: C = relu(A . B^T + bias), mask <- sign pattern; backward: C = mask ? A . B^T : 0.  mask: uint8 buffer of
    workspace_bytes(ME_WS_RELU_MASK, M, N, K, dtype) bytes (0 bytes = shape not served: use gemm_nt with the gate operand)."""
    dtype = dtype or A.dtype
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    check(lib().me_gemm_nt_relu_mask(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C), C.stride(0), _ptr(bias), _ptr(mask),
                                     M, N, K, 1 if backward else 0, _code(dtype), _stream()), "me_gemm_nt_relu_mask")


def workspace_bytes(op, M, N, K, dtype):
    """me_workspace_bytes: scratch bytes the entry point `op` needs for this shape (0 = none)."""
    return int(lib().me_workspace_bytes(int(op), int(M), int(N), int(K), _code(dtype)))


def gemm_tn_acc(A, B, dW, dbias=None, T=None, N=None, K=None, dtype=None, ws=None):
    """dW[N,K] += A[T,N]^T . B[T,K] ; dbias[N] += colsum(A).  ws: caller-owned uint8 workspace of at least
    workspace_bytes(ME_WS_GEMM_TN, T, N, K, dtype) bytes (deterministic summation); None -> f32 atomics."""
    dtype = dtype or A.dtype
    T = A.shape[0] if T is None else T
    N = A.shape[1] if N is None else N
    K = B.shape[1] if K is None else K
    check(lib().me_gemm_tn_acc(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(dW), dW.stride(0), _ptr(dbias),
                               T, N, K, _ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0,
                               _code(dtype), _stream()), "me_gemm_tn_acc")


def gemm_tn_acc_group(items, T, dtype, ws=None):
    """items: list of (dY [T, N], X [T, K], dW [N, K] f32, dbias [N] f32 or None, N, K) sharing the token dimension T --
    one grouped launch (me_gemm_tn_acc_group); ws: uint8 workspace of workspace_bytes(ME_WS_GEMM_TN_GROUP, T, tiles, 0)."""
    if not 0 < len(items) <= ME_TN_MAX_GROUP:
        raise RuntimeError("gemm_tn_acc_group takes 1..%d products" % ME_TN_MAX_GROUP)
    arr = (_lib.TnItem * len(items))()
    for it, (A, B, dW, db, N, K) in zip(arr, items):
        it.A, it.lda, it.B, it.ldb = _ptr(A), A.stride(0), _ptr(B), B.stride(0)
        it.dW, it.lddw, it.dbias, it.N, it.K = _ptr(dW), dW.stride(0), _ptr(db), N, K
    check(lib().me_gemm_tn_acc_group(arr, len(items), T, _ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0,
                                     _code(dtype), _stream()), "me_gemm_tn_acc_group")


def tn_group_tiles(shapes):
    """total 256 x 256 tiles of a list of (N, K) weight-gradient shapes (argument N of ME_WS_GEMM_TN_GROUP)."""
    return sum(((n + 255) // 256) * (k // 256) for n, k in shapes)


def rel_pack_numel(M, dh):
    """Elements of the packed relative table (me_rga_pack_rel): per 32-row block dh/16 row images + 2*ceil(dh/32)
    transposed images of 512 elements."""
    return (M // 32) * (dh // 16 + 2 * ((dh + 31) // 32)) * 512


def rga_pack_rel(E, out=None):
    """E [M, dh] (compute dtype) -> packed fragment images read by rga_fwd / rga_bwd."""
    M, dh = E.shape
    if out is None:
        out = torch.empty(rel_pack_numel(M, dh), dtype=E.dtype, device=E.device)
    check(lib().me_rga_pack_rel(_ptr(E), _ptr(out), M, dh, _code(E.dtype), _stream()), "me_rga_pack_rel")
    return out


def rga_fwd(qkv, Epk, key_pad, out, lse, B, L, H, dh, M, causal=True, PT=None, MT=None):
    """Epk = rga_pack_rel(E).  causal=False: bidirectional attention of the regression model.  PT / MT (rga_saved_buffers):
    training mode -- the probability tiles and running maxima me_rga_bwd reads."""
    check(lib().me_rga_fwd(_ptr(qkv), _ptr(Epk), _ptr(key_pad), _ptr(out), _ptr(lse), _ptr(PT), _ptr(MT), B, L, H, dh, M,
                           1 if causal else 0, _code(qkv.dtype), _stream()), "me_rga_fwd")


def rga_saved_buffers(B, H, L, dtype, device, causal=True):
    """(PT, MT): what a training-mode me_rga_fwd leaves for me_rga_bwd of the same layer (sizes: me_workspace_bytes)."""
    Lp = ((L + 31) // 32) * 32
    es = torch.empty(0, dtype=dtype).element_size()
    n_pt = workspace_bytes(ME_WS_RGA_PT, B * H, Lp, 1 if causal else 0, dtype) // es
    n_mt = workspace_bytes(ME_WS_RGA_MT, B * H, Lp, 0, dtype) // 4
    return torch.empty(n_pt, dtype=dtype, device=device), torch.empty(n_mt, dtype=torch.float32, device=device)


def rga_bwd_workspace(B, H, L, dtype, device):
    """dGT: uninitialised tile workspace of me_rga_bwd (shared by all layers), sized by me_workspace_bytes."""
    Lp = ((L + 31) // 32) * 32
    es = torch.empty(0, dtype=dtype).element_size()
    return torch.empty(workspace_bytes(ME_WS_RGA_DGT, B * H, Lp, 0, dtype) // es, dtype=dtype, device=device)


_side = {}


def _side_stream(device):
    """second stream + two events per device for rga_bwd(overlap=True) (created once; the library itself owns no stream)."""
    key = torch.device(device).index or 0
    if key not in _side:
        _side[key] = (torch.cuda.Stream(device=device), torch.cuda.Event(), torch.cuda.Event())
    return _side[key]


def rga_bwd(qkv, Epk, out, lse, dout, dqkv, dE, delta_ws, PT, MT, dGT, B, L, Lp, H, dh, M, causal=True, overlap=False):
    """overlap=True: the key-owned kernel (dK, dV) stays on the current stream, the E-row-owned kernel (dE) runs beside it on a
    second stream (me_rga_bwd_phases; both only depend on the query-owned kernel); the current stream then waits for it.
    Lifetime: the side stream reads dGT / qkv and accumulates into dE without the caching allocator knowing -- the caller keeps
    those tensors alive until the current stream has passed this call (the model's workspaces and flat gradients are; a caller
    that frees them right after the call must synchronise or record_stream() them first)."""
    es = PT.element_size()
    if (PT.numel() * es < workspace_bytes(ME_WS_RGA_PT, B * H, Lp, 1 if causal else 0, qkv.dtype) or
            MT.numel() * 4 < workspace_bytes(ME_WS_RGA_MT, B * H, Lp, 0, qkv.dtype) or
            dGT.numel() * es < workspace_bytes(ME_WS_RGA_DGT, B * H, Lp, 0, qkv.dtype)):
        raise RuntimeError("rga_bwd: PT / MT / dGT smaller than me_workspace_bytes")
    if not overlap:
        check(lib().me_rga_bwd(_ptr(qkv), _ptr(Epk), _ptr(out), _ptr(lse), _ptr(dout), _ptr(dqkv), _ptr(dE), _ptr(delta_ws),
                               _ptr(PT), _ptr(MT), _ptr(dGT), B, L, Lp, H, dh, M, 1 if causal else 0, _code(qkv.dtype), _stream()),
              "me_rga_bwd")
        return
    side, ev_q, ev_e = _side_stream(qkv.device)
    main = torch.cuda.current_stream(qkv.device)

    def phase(bits, stream):
        check(lib().me_rga_bwd_phases(_ptr(qkv), _ptr(Epk), _ptr(out), _ptr(lse), _ptr(dout), _ptr(dqkv), _ptr(dE), _ptr(delta_ws),
                                      _ptr(PT), _ptr(MT), _ptr(dGT), B, L, Lp, H, dh, M, 1 if causal else 0, bits, _code(qkv.dtype),
                                      ctypes.c_void_p(stream.cuda_stream)), "me_rga_bwd_phases")
    phase(1, main)
    ev_q.record(main)
    side.wait_event(ev_q)
    phase(4, side)                  # dE first: the short streaming kernel gets its blocks in beside the long one
    ev_e.record(side)
    phase(2, main)
    main.wait_event(ev_e)


def resid_ln_fwd(x, a, gamma, beta, y, s_out, stats, rows, d, eps, p, seed, site, x_lo=None, y_lo=None):
    check(lib().me_resid_ln_fwd(_ptr(x), _ptr(x_lo), _ptr(a), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y_lo), _ptr(s_out), _ptr(stats), rows, d,
                                float(eps), float(p), int(seed), int(site), _code(x.dtype) | _lo_flag(x_lo, y_lo), _stream()), "me_resid_ln_fwd")


def resid_ln_bwd(dy, s, stats, gamma, dx, da, dgamma, dbeta, rows, d, p, seed, site):
    check(lib().me_resid_ln_bwd(_ptr(dy), _ptr(s), _ptr(stats), _ptr(gamma), _ptr(dx), _ptr(da), _ptr(dgamma),
                                _ptr(dbeta), rows, d, float(p), int(seed), int(site), _code(dy.dtype), _stream()),
          "me_resid_ln_bwd")


def ce_fwd(logits, target, row_lse, loss_sum, n_valid, rows, V, ignore_index):
    check(lib().me_ce_fwd(_ptr(logits), logits.stride(0), _ptr(target), _ptr(row_lse), _ptr(loss_sum), _ptr(n_valid),
                          rows, V, ignore_index, _code(logits.dtype), _stream()), "me_ce_fwd")


def ce_bwd_fuses_dbias(logits, dlogits):
    """can me_ce_bwd produce the head's bias gradient itself (f32 column sums before the 16-bit rounding)?"""
    return (logits.dtype in (torch.bfloat16, torch.float16) and dlogits.dtype == logits.dtype and logits.stride(0) % 8 == 0 and
            dlogits.stride(0) % 8 == 0 and logits.stride(0) >= dlogits.stride(0) and dlogits.stride(0) <= 2048)


def ce_bwd(logits, target, row_lse, dlogits, n_valid, extra_scale, rows, V, ignore_index, dbias=None, loss_scale=None):
    """loss_scale: f32 device scalar (the f16 tier's dynamic loss scale, LossScaler.scale_tensor) multiplied into dlogits."""
    check(lib().me_ce_bwd(_ptr(logits), logits.stride(0), _ptr(target), _ptr(row_lse), _ptr(dlogits),
                          dlogits.stride(0), _ptr(n_valid), float(extra_scale), _ptr(loss_scale), _ptr(dbias), rows, V, ignore_index,
                          _code(logits.dtype), _code(dlogits.dtype), _stream()), "me_ce_bwd")


def sumsq(g, out, ws=None):
    """out[0] += sum g^2.  ws: zeroed uint8 buffer of workspace_bytes(ME_WS_SUMSQ) bytes (sumsq_ws): block sums are added
    in a fixed order -- bit-reproducible, identical on every data-parallel rank; None: atomics in arrival order."""
    check(lib().me_sumsq(_ptr(g), g.numel(), _ptr(out), _ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0,
                         _stream()), "me_sumsq")


def sumsq_ws(device):
    return torch.zeros(workspace_bytes(ME_WS_SUMSQ, 0, 0, 0, torch.float32), dtype=torch.uint8, device=device)


def scaler_step(state, sumsq_t, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    """GradScaler.unscale_/step/update bookkeeping on the device (me_scaler_step): state f32 [ME_SCALER_WORDS]."""
    check(lib().me_scaler_step(_ptr(state), _ptr(sumsq_t), float(growth_factor), float(backoff_factor), int(growth_interval),
                               _stream()), "me_scaler_step")


def adamw_step(p, g, m, v, sumsq_t, clip, grad_scale, lr, beta1, beta2, eps, weight_decay, step, zero_grad, scaler_state=None):
    """scaler_state: the state me_scaler_step just updated (f16 tier): unscale, skip on inf / nan, device-side step count."""
    bc1 = 1.0 - beta1 ** max(step, 1)
    bc2 = 1.0 - beta2 ** max(step, 1)
    check(lib().me_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(sumsq_t), float(clip),
                              float(grad_scale), float(lr), float(beta1), float(beta2), float(eps),
                              float(weight_decay), float(bc1), float(bc2), int(bool(zero_grad)), _ptr(scaler_state), _stream()),
          "me_adamw_step")


def dec_ln_qkv_attn(s_in, gamma, beta, eps, Wqkv, bqkv, x_out, kcache, vcache, E, key_pad, ld_pad, part, nsplit, Mr, d, H, dh, M,
                    Mc, t, t_dev, dtype):
    """fused LayerNorm -> q|k|v of one head -> cache append -> key-split attention partials (me_dec_ln_qkv_attn)."""
    check(lib().me_dec_ln_qkv_attn(_ptr(s_in), _ptr(gamma), _ptr(beta), float(eps), _ptr(Wqkv), _ptr(bqkv), _ptr(x_out),
                                   _ptr(kcache), _ptr(vcache), _ptr(E), _ptr(key_pad), ld_pad, _ptr(part), nsplit, Mr, d, H, dh, M,
                                   Mc, int(t), _ptr(t_dev), _code(dtype), _stream()), "me_dec_ln_qkv_attn")


def dec_qkv(s_in, gamma, beta, eps, x_hi, x_lo, Wqkv, bqkv, x_out, q_out, kcache, vcache, Mr, d, H, dh, Mc, t, t_dev, dtype):
    check(lib().me_dec_qkv(_ptr(s_in), _ptr(gamma), _ptr(beta), float(eps), _ptr(x_hi), _ptr(x_lo), _ptr(Wqkv), _ptr(bqkv),
                           _ptr(x_out), _ptr(q_out), _ptr(kcache), _ptr(vcache), Mr, d, H, dh, Mc, int(t), _ptr(t_dev),
                           _code(dtype), _stream()), "me_dec_qkv")


def dec_embed_qkv(tokens, cond, emb, cw, cb, pe, d_cond, Wqkv, bqkv, x_out, q_out, kcache, vcache, Mr, d, H, dh, Mc, t, t_dev, dtype):
    check(lib().me_dec_embed_qkv(_ptr(tokens), _ptr(cond), _ptr(emb), _ptr(cw), _ptr(cb), _ptr(pe), int(d_cond), _ptr(Wqkv),
                                 _ptr(bqkv), _ptr(x_out), _ptr(q_out), _ptr(kcache), _ptr(vcache), Mr, d, H, dh, Mc, int(t),
                                 _ptr(t_dev), _code(dtype), _stream()), "me_dec_embed_qkv")


def dec_embed_qkv_attn(tokens, cond, emb, cw, cb, pe, d_cond, Wqkv, bqkv, x_out, kcache, vcache, E, key_pad, ld_pad, part, nsplit,
                       Mr, d, H, dh, M, Mc, t, t_dev, dtype):
    """first layer: embedding row -> q|k|v of one head -> cache append -> key-split attention partials (me_dec_embed_qkv_attn)."""
    check(lib().me_dec_embed_qkv_attn(_ptr(tokens), _ptr(cond), _ptr(emb), _ptr(cw), _ptr(cb), _ptr(pe), int(d_cond), _ptr(Wqkv),
                                      _ptr(bqkv), _ptr(x_out), _ptr(kcache), _ptr(vcache), _ptr(E), _ptr(key_pad), ld_pad, _ptr(part),
                                      nsplit, Mr, d, H, dh, M, Mc, int(t), _ptr(t_dev), _code(dtype), _stream()), "me_dec_embed_qkv_attn")


def dec_attn(q, kcache, vcache, E, key_pad, ld_pad, part, nsplit, Mr, H, dh, M, Mc, t, t_dev, dtype):
    check(lib().me_dec_attn(_ptr(q), _ptr(kcache), _ptr(vcache), _ptr(E), _ptr(key_pad), ld_pad, _ptr(part), nsplit, Mr, H,
                            dh, M, Mc, int(t), _ptr(t_dev), _code(dtype), _stream()), "me_dec_attn")


def dec_proj_resid(part, nsplit, H, dh, x_T, W, bias, resid, out, Mr, N, K, dtype):
    check(lib().me_dec_proj_resid(_ptr(part), nsplit, H, dh, _ptr(x_T), x_T.stride(0) if x_T is not None else 0, _ptr(W),
                                  W.stride(0), _ptr(bias), _ptr(resid), _ptr(out), Mr, N, K, _code(dtype), _stream()),
          "me_dec_proj_resid")


def dec_ln_proj(s_in, gamma, beta, eps, W, bias, x_out, y, Mr, N, K, flags, dtype):
    check(lib().me_dec_ln_proj(_ptr(s_in), _ptr(gamma), _ptr(beta), float(eps), _ptr(W), W.stride(0), _ptr(bias), _ptr(x_out),
                               _ptr(y), y.stride(0), Mr, N, K, int(flags), _code(dtype), _stream()), "me_dec_ln_proj")

def dec_token_blocks(dh, d, d_inner, dtype):
    """co-resident blocks me_dec_token gets on this device (0: shape / type not served)."""
    return int(lib().me_dec_token_blocks(int(dh), int(d), int(d_inner), _code(dtype)))


def dec_token_table(layers, device):
    """device copy of an array of me_dec_layer built from dicts of tensors (keys = the struct's fields)."""
    import numpy as np
    from ._lib import DecLayer
    arr = (DecLayer * len(layers))()
    for i, L in enumerate(layers):
        for name, _ in DecLayer._fields_:
            setattr(arr[i], name, L[name].data_ptr())
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def dec_token(tokens, cond, emb, cw, cb, pe, d_cond, table, n_layer, Wf, bf, V, logits, ws, nsplit, Mr, d, d_inner, H, dh, M, Mc, t,
              t_dev, eps, blocks, dtype):
    """one decode position through all layers and the head in ONE persistent launch (me_dec_token)."""
    check(lib().me_dec_token(_ptr(tokens), _ptr(cond), _ptr(emb), _ptr(cw), _ptr(cb), _ptr(pe), int(d_cond), _ptr(table), int(n_layer),
                             _ptr(Wf), Wf.stride(0), _ptr(bf), int(V), _ptr(logits), logits.stride(0), _ptr(ws), ws.numel() * ws.element_size(),
                             int(nsplit), int(Mr), int(d), int(d_inner), int(H), int(dh), int(M), int(Mc), int(t), _ptr(t_dev), float(eps),
                             int(blocks), _code(dtype), _stream()), "me_dec_token")



def sample_topk_topp(logits, V, special, temp, top_k, top_p, u, out_ids, n_choices=None, dbg_p=None, dbg_i=None):
    B = out_ids.numel()
    check(lib().me_sample_topk_topp(_ptr(logits), logits.stride(0), V, _ptr(special),
                                    special.numel() if special is not None else 0, _ptr(temp), int(top_k), float(top_p),
                                    _ptr(u), _ptr(out_ids), _ptr(n_choices), _ptr(dbg_p), _ptr(dbg_i), B, _stream()),
          "me_sample_topk_topp")


def sample_step(logits, V, special, prev_tok, is_timeshift, repeat_counts, temp_note, temp_rest, penalty_coeff, top_k, top_p,
                u_table, pos, pos0, out_ids, n_choices=None):
    B = out_ids.numel()
    check(lib().me_sample_step(_ptr(logits), logits.stride(0), V, _ptr(special),
                               special.numel() if special is not None else 0, _ptr(prev_tok), _ptr(is_timeshift),
                               _ptr(repeat_counts), float(temp_note), float(temp_rest), float(penalty_coeff), int(top_k),
                               float(top_p), _ptr(u_table), u_table.stride(0), _ptr(pos), int(pos0), _ptr(out_ids),
                               _ptr(n_choices), B, _stream()), "me_sample_step")


def decode_commit(tok, history, pos, B):
    check(lib().me_decode_commit(_ptr(tok), _ptr(history), history.stride(0), _ptr(pos), B, _stream()), "me_decode_commit")


def greedy_pick_commit(logits, V, special, out_ids, history, pos, B):
    check(lib().me_greedy_pick_commit(_ptr(logits), logits.stride(0), V, _ptr(special),
                                      special.numel() if special is not None else 0, _ptr(out_ids), _ptr(history),
                                      history.stride(0), _ptr(pos), B, _stream()), "me_greedy_pick_commit")


def greedy_pick(logits, V, special, out_ids, B):
    check(lib().me_greedy_pick(_ptr(logits), logits.stride(0), V, _ptr(special),
                               special.numel() if special is not None else 0, _ptr(out_ids), B, _stream()),
          "me_greedy_pick")
