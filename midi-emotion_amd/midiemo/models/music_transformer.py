"""Emotion-conditioned Music Transformer on the HIP engine.

Mirrors the reference module contract (models/music_multi.py:41-108,
models/music_continuous_token.py:32-105): same constructor kwargs, same
state_dict keys/shapes, `forward(x, condition) -> logits [B, L(+2), V]`,
`.train()/.eval()/.to()/.parameters()`; usable under autograd
(`loss.backward()` fills `p.grad`).  Unlike the reference, the compute is not a
graph of PyTorch ops: forward and backward are explicit kernel sequences over
the C-ABI of libmidiemo_hip.so (see include/midiemo.h), activations live in
preallocated workspaces, and all parameters/gradients live in two flat f32
buffers (one fused optimiser pass, contiguous all-reduce buckets).

Extras used by the build's own train.py / bench.py / generate.py:
  * `loss_and_backward(...)`  fused forward + CE + backward into the flat grads
  * `flat_params / flat_grads / bucket_ranges()`  for FusedAdamW and DDP
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn

from .. import ops

# compute tiers: "bf16" (BASELINE's headline dtype), "fp16" (the reference's own autocast dtype, train.py:281 -- needs the loss
# scale: optim.LossScaler), "fp32" (exact-f32 MFMA, the parity gate)
_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16, "f16": torch.float16,
           "fp32": torch.float32, "float32": torch.float32}


def _round_up(x, m):
    return ((x + m - 1) // m) * m


class _Lin(nn.Module):
    """Parameter holder with nn.Linear's names/shapes/initialiser (compute happens in the engine)."""

    def __init__(self, in_f, out_f):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.empty(out_f))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_f)
        nn.init.uniform_(self.bias, -bound, bound)


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Emb(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d))


class _RGA(nn.Module):
    """RelativeGlobalAttention parameters (music_multi.py:180-185)."""

    def __init__(self, d, h, max_seq):
        super().__init__()
        self.Wq = _Lin(d, d)
        self.Wk = _Lin(d, d)
        self.Wv = _Lin(d, d)
        self.fc = _Lin(d, d)
        self.E = nn.Parameter(torch.randn(max_seq, d // h))


class _EncoderLayer(nn.Module):
    """EncoderLayer parameters (music_multi.py:111-124)."""

    def __init__(self, d, d_inner, h, max_seq):
        super().__init__()
        self.rga = _RGA(d, h, max_seq)
        self.FFN_pre = _Lin(d, d_inner)
        self.FFN_suf = _Lin(d_inner, d)
        self.layernorm1 = _LN(d)
        self.layernorm2 = _LN(d)


def sinusoid_table(max_seq, d):
    """music_multi.py:137-147, evaluated in float64 like the reference's Python floats."""
    p = np.arange(max_seq, dtype=np.float64)[:, None]
    i = np.arange(d, dtype=np.float64)[None, :]
    par = np.mod(i, 2.0)
    ang = p * np.exp(-math.log(10000.0) * i / d) * np.exp(math.log(10000.0) / d * par) + 0.5 * math.pi * par
    return torch.from_numpy(np.sin(ang)).float()


class _Workspace:
    pass


class MusicTransformerHIP(nn.Module):
    LN_EPS = 1e-6
    _logits_f32 = False         # workspace logits of the loss path in f32 regardless of the compute type (MusicRegression)

    def __init__(self, embedding_dim=None, d_inner=None, d_condition=-1, vocab_size=None, num_layer=None,
                 num_head=None, max_seq=2048, dropout=0.1, pad_token=0, token_conditioning=False,
                 compute_dtype="bf16", head_size=None, causal=True):
        super().__init__()
        self.head_size = vocab_size if head_size is None else head_size     # output features of the head (vocab; 2 for regression)
        self.causal = bool(causal)                                           # False: MusicRegression's bidirectional attention
        self.max_seq = max_seq
        self.num_layer = num_layer
        self.num_head = num_head
        self.embedding_dim = embedding_dim
        self.d_inner = d_inner
        self.vocab_size = vocab_size
        self.pad_token = pad_token
        self.token_conditioning = bool(token_conditioning)
        d_condition = 0 if (d_condition is None or d_condition < 0 or token_conditioning) else d_condition
        self.d_condition = d_condition
        self.dropout_p = float(dropout)
        self.dh = embedding_dim // num_head
        if embedding_dim % num_head or self.dh not in (32, 48, 64):
            raise ValueError("head dim %d unsupported by the HIP attention kernels (32, 48 or 64)" % self.dh)
        if max_seq % 32:
            raise ValueError("max_seq must be a multiple of 32")
        self.compute_dtype = _DTYPES[compute_dtype] if isinstance(compute_dtype, str) else compute_dtype
        # residual stream carried as bf16 hi + lo (f32-class precision, like autocast's fp32 stream); 0 = round it to bf16
        self.resid_lo = os.environ.get("MIDIEMO_RESID_LO", "1") != "0"
        # 1 (default): both residual arrays of a layer (the layer stream h and LayerNorm1's output o1) carry a low half;
        # 2 / 3 (probes, tools/probe_f16_nolo.py): only h / only o1
        self._resid_lo_mode = int(os.environ.get("MIDIEMO_RESID_LO", "1") or 1)
        # the low halves as one BYTE per element (ME_LO8: 15 mantissa bits of the bf16 stream in 3 bytes) or as a second 16-bit array.
        # bf16 tier: 8 bits since round 6 (-0.06 ms per step same-process; logits 3.476e-3 against 3.487e-3, every derived bound holds);
        # f16 tier: 16 bits (with 8, the condition projection's bias gradient sits at 2.12 x the oracle's fp16-autocast error, bound 2.0).
        # MIDIEMO_RESID_LO_BITS=8|16 overrides both.
        self._resid_lo_bits_env = int(os.environ.get("MIDIEMO_RESID_LO_BITS", "0"))
        # attention backward: the key-owned (dK, dV) and the E-row-owned (dE) kernels side by side on two streams (both only
        # depend on the query-owned kernel; same results); measured -9 us (L = 1024) / -35 us (L = 2048) per layer
        # Under an initialised process group (RCCL's own streams alive) the second stream measured no gain under the `window` /
        # `eager` bucket policies and +0.1 ms per step under `end` (profiles/r06_ddp_overlap.txt): default = on for a single
        # process, off once torch.distributed is initialised; MIDIEMO_ATTN_BWD_OVERLAP=0|1 forces either.
        self._attn_bwd_overlap_env = os.environ.get("MIDIEMO_ATTN_BWD_OVERLAP")

        self.embedding = _Emb(vocab_size, embedding_dim - d_condition)
        if self.token_conditioning:
            self.fc_condition = nn.ModuleList([_Lin(1, embedding_dim) for _ in range(2)])
        elif d_condition > 0:
            self.fc_condition = _Lin(2, d_condition)
        self.enc_layers = nn.ModuleList([_EncoderLayer(embedding_dim, d_inner, num_head, max_seq)
                                         for _ in range(num_layer)])
        self.fc = self._make_head(embedding_dim, self.head_size)
        self.init_weights()

        self._step_seed = 0x5EED
        self._fwd_count = 0
        self._ws = {}
        self._prep = None
        self._ct_desc = None
        self._prep_version = None
        self._dirty = True
        self._pe = None
        self._packing = False
        self._pack()

    @property
    def attn_bwd_overlap(self):
        if self._attn_bwd_overlap_env is not None:
            return self._attn_bwd_overlap_env != "0"
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized())

    @attn_bwd_overlap.setter
    def attn_bwd_overlap(self, on):
        self._attn_bwd_overlap_env = "1" if on else "0"

    # ------------------------------------------------------------------ head (overridden by MusicRegression: Sequential(Linear, Tanh))
    _HEAD_W, _HEAD_B = "fc.weight", "fc.bias"

    def _make_head(self, d, n_out):
        return _Lin(d, n_out)

    def _head_linear(self):
        return self.fc

    # ------------------------------------------------------------------ init / packing
    def init_weights(self):
        """music_multi.py:75-82 / music_continuous_token.py:68-75."""
        r = 0.1
        with torch.no_grad():
            self.embedding.weight.uniform_(-r, r)
            self._head_linear().bias.zero_()
            self._head_linear().weight.uniform_(-r, r)
            if self.token_conditioning:
                for m in self.fc_condition:
                    m.weight.uniform_(-r, r)
                    m.bias.zero_()
            elif self.d_condition > 0:
                self.fc_condition.bias.zero_()
                self.fc_condition.weight.uniform_(-r, r)

    def _param_order(self):
        """(name, param) in flat-buffer order; Wq/Wk/Wv weights (and biases) are adjacent so the
        fused QKV projection reads one [3d, d] matrix.  Groups = DDP buckets."""
        groups = []
        g = [("embedding.weight", self.embedding.weight)]
        if self.token_conditioning:
            for i, m in enumerate(self.fc_condition):
                g += [(f"fc_condition.{i}.weight", m.weight), (f"fc_condition.{i}.bias", m.bias)]
        elif self.d_condition > 0:
            g += [("fc_condition.weight", self.fc_condition.weight), ("fc_condition.bias", self.fc_condition.bias)]
        groups.append(g)
        for i, l in enumerate(self.enc_layers):
            p = f"enc_layers.{i}."
            groups.append([
                (p + "rga.E", l.rga.E),
                (p + "rga.Wq.weight", l.rga.Wq.weight), (p + "rga.Wk.weight", l.rga.Wk.weight),
                (p + "rga.Wv.weight", l.rga.Wv.weight),
                (p + "rga.Wq.bias", l.rga.Wq.bias), (p + "rga.Wk.bias", l.rga.Wk.bias), (p + "rga.Wv.bias", l.rga.Wv.bias),
                (p + "rga.fc.weight", l.rga.fc.weight), (p + "rga.fc.bias", l.rga.fc.bias),
                (p + "FFN_pre.weight", l.FFN_pre.weight), (p + "FFN_pre.bias", l.FFN_pre.bias),
                (p + "FFN_suf.weight", l.FFN_suf.weight), (p + "FFN_suf.bias", l.FFN_suf.bias),
                (p + "layernorm1.weight", l.layernorm1.weight), (p + "layernorm1.bias", l.layernorm1.bias),
                (p + "layernorm2.weight", l.layernorm2.weight), (p + "layernorm2.bias", l.layernorm2.bias),
            ])
        groups.append([(self._HEAD_W, self._head_linear().weight), (self._HEAD_B, self._head_linear().bias)])
        return groups

    def _pack(self):
        """(Re)build the flat f32 parameter/gradient buffers on the parameters' current device and
        turn every nn.Parameter into a view of the flat buffer."""
        groups = self._param_order()
        dev = groups[0][0][1].device
        off = 0
        slices, buckets = {}, []
        for g in groups:
            start = off
            for name, p in g:
                n = p.numel()
                slices[name] = (off, n, tuple(p.shape))
                off = _round_up(off + n, 4)
            buckets.append((start, off))
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for g in groups:
                for name, p in g:
                    o, n, shp = slices[name]
                    flat[o:o + n].copy_(p.detach().reshape(-1).to(torch.float32))
                    p.data = flat[o:o + n].view(shp)
                    p.grad = None
        self._flat, self._gflat, self._slices, self._buckets = flat, gflat, slices, buckets
        self._ws = {}
        self._prep = None
        self._ct_desc = None
        self._dirty = True
        self._pe = sinusoid_table(self.max_seq, self.embedding_dim).to(dev)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        if not self._packing:
            self._packing = True
            try:
                self._pack()
            finally:
                self._packing = False
        return self

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._dirty = True
        return r

    # ------------------------------------------------------------------ flat views for optimiser / DDP
    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        return self._gflat

    def bucket_ranges(self):
        """[(lo, hi)] element ranges of the flat buffers: embedding(+cond), layer 0..N-1, head."""
        return list(self._buckets)

    def _pview(self, buf, name):
        o, n, shp = self._slices[name]
        return buf[o:o + n].view(shp)

    def link_grads(self):
        """Make every p.grad a view of the flat gradient buffer (used by tests / external optimisers)."""
        for g in self._param_order():
            for name, p in g:
                p.grad = self._pview(self._gflat, name)

    def mark_params_changed(self):
        self._dirty = True

    # ------------------------------------------------------------------ prepared (cast / transposed) weights
    def _param_versions(self):
        """Sum of the version counters of the flat buffer and of every nn.Parameter view.  `p.data = flat[...]` views
        do NOT share flat's counter, so an in-place update made through a parameter (torch.optim step, p.copy_(),
        EMA, manual re-init) bumps only p._version: both have to be watched."""
        v = self._flat._version
        for g in self._param_order():
            for _, p in g:
                v += p._version
        return v

    def _refresh_weights(self):
        ver = self._param_versions()
        if self._prep is not None and not self._dirty and ver == self._prep_version:
            return
        dt = self.compute_dtype
        dev = self._flat.device
        d, di, V, dh, M = self.embedding_dim, self.d_inner, self.head_size, self.dh, self.max_seq
        if self._prep is None:
            def buf(r, c, ld=None):
                return torch.zeros(r, ld or c, dtype=dt, device=dev)
            layers = []
            for _ in range(self.num_layer):
                L = {}
                if dt != torch.float32:
                    L.update(Wqkv=buf(3 * d, d), Wo=buf(d, d), W1=buf(di, d), W2=buf(d, di), E=buf(M, dh))
                L.update(WqkvT=buf(d, 3 * d), WoT=buf(d, d), W1T=buf(d, di), W2T=buf(di, d),
                         Epk=torch.zeros(ops.rel_pack_numel(M, dh), dtype=dt, device=dev))
                layers.append(L)
            head = {"WfT": buf(d, V, _round_up(V, 64))}
            if dt != torch.float32:
                head["Wf"] = buf(V, d)
            self._prep = {"layers": layers, "head": head}
        f = self._flat
        if getattr(self, "_ct_desc", None) is None or self._ct_desc_ptr != f.data_ptr():
            # one descriptor table for all prepared weights: the refresh is a single multi-tensor launch
            items = []
            for i, L in enumerate(self._prep["layers"]):
                p = f"enc_layers.{i}."
                o, _, _ = self._slices[p + "rga.Wq.weight"]
                wqkv = f[o:o + 3 * d * d].view(3 * d, d)
                srcs = {"Wqkv": wqkv, "Wo": self._pview(f, p + "rga.fc.weight"), "W1": self._pview(f, p + "FFN_pre.weight"),
                        "W2": self._pview(f, p + "FFN_suf.weight"), "E": self._pview(f, p + "rga.E")}
                for k, src in srcs.items():
                    if k == "E":      # relative table: natural cast copy (decode) + packed fragment images (fwd / bwd)
                        items.append((src, L.get(k) if dt != torch.float32 else None, L["Epk"], 1))
                    else:
                        items.append((src, L.get(k) if dt != torch.float32 else None, L[k + "T"]))
                    if dt == torch.float32:
                        L[k] = src
                ob, _, _ = self._slices[p + "rga.Wq.bias"]
                L["bqkv"] = f[ob:ob + 3 * d]
            H = self._prep["head"]
            src = self._pview(f, self._HEAD_W)
            items.append((src, H.get("Wf") if dt != torch.float32 else None, H["WfT"]))
            if dt == torch.float32:
                H["Wf"] = src
            self._ct_desc = ops.make_ct_desc(items, dev)
            self._ct_desc_ptr = f.data_ptr()
        ops.cast_transpose_multi(self._ct_desc[0], self._ct_desc[1], self._ct_desc[2], dt)
        self._prep_version = ver
        self._dirty = False

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, B, Lm, save):
        key = (B, Lm, bool(save))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        if len(self._ws) > 6:
            self._ws.clear()
        dt, dev = self.compute_dtype, self._flat.device
        d, di, H, V, N = self.embedding_dim, self.d_inner, self.num_head, self.head_size, self.num_layer
        T = B * Lm
        Lp = _round_up(Lm, 32)
        e = lambda *s, dtype=dt: torch.empty(*s, dtype=dtype, device=dev)
        ws = _Workspace()
        ws.Lp = Lp
        ws.key_pad = e(B, Lm, dtype=torch.uint8)
        nl = N if save else 1
        ws.h = [e(T, d) for _ in range(N + 1 if save else 2)]
        # bf16 tier: low-order half of the residual stream (h, o1), so that the stream itself is ~f32 like the
        # reference's under autocast while the GEMMs still read the bf16 "hi" tensors (me_resid_ln_fwd x_lo / y_lo)
        lo = self.resid_lo and dt != torch.float32
        ws.hlo = [(e(T, d, dtype=torch.uint8) if self.resid_lo_bits == 8 else e(T, d)) if lo else None for _ in range(2)]
        ws.layers = []
        for _ in range(nl):
            L = _Workspace()
            L.qkv, L.att, L.o1, L.hid = e(T, 3 * d), e(T, d), e(T, d), e(T, di)
            L.lse = e(B, H, Lm, dtype=torch.float32)
            if save:
                L.s1, L.s2 = e(T, d), e(T, d)
                L.st1, L.st2 = e(T, 2, dtype=torch.float32), e(T, 2, dtype=torch.float32)
                # what the attention forward leaves for its backward: unnormalised probability tiles + running maxima
                L.PT, L.MT = ops.rga_saved_buffers(B, H, Lm, dt, dev, causal=self.causal)
                # ReLU sign mask of the FFN (1 bit per hidden unit: the FFN_suf dgrad reads it instead of the activations);
                # 0 bytes = this shape / dtype keeps the gate operand
                nmask = 0 if os.environ.get("MIDIEMO_NO_RELU_MASK") else ops.workspace_bytes(ops.ME_WS_RELU_MASK, T, di, d, dt)
                L.rmask = e(nmask, dtype=torch.uint8) if nmask else None
            else:
                L.s1 = L.s2 = L.st1 = L.st2 = L.PT = L.MT = L.rmask = None
            ws.layers.append(L)
        ws.tmp = e(T, d)
        if save:
            ldv = _round_up(V, 64)
            # the training loss reads the logits in the compute type (bf16 tier: what the reference's autocast F.linear
            # produces; the 134 MB f32 logits tensor of round 1 is gone); MusicRegression's tanh head keeps f32
            ws.logits = e(T, ldv, dtype=torch.float32 if self._logits_f32 else dt)
            ws.row_lse = e(T, dtype=torch.float32)
            ws.dlogits = torch.zeros(T, ldv, dtype=dt, device=dev)
            ws.acc = torch.zeros(2, dtype=torch.float32, device=dev)      # loss_sum, n_valid
            ws.dA, ws.dB, ws.dC, ws.dC2 = e(T, d), e(T, d), e(T, d), e(T, d)
            ws.dhid, ws.dqkv = e(T, di), e(T, 3 * d)
            ws.delta = e(B, H, Lm, dtype=torch.float32)
            # tiles of the skewed dS (dG^T) of the layer being differentiated (me_workspace_bytes; no initialisation
            # contract: every tile is written before it is read)
            ws.dGT = ops.rga_bwd_workspace(B, H, Lm, dt, dev)
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ engine: forward
    def _mode(self):
        if self.token_conditioning:
            return ops.ME_COND_TOKEN
        return ops.ME_COND_CONCAT if self.d_condition > 0 else ops.ME_COND_NONE

    def _cond_params(self, buf):
        if self.token_conditioning:
            return (self._pview(buf, "fc_condition.0.weight"), self._pview(buf, "fc_condition.0.bias"),
                    self._pview(buf, "fc_condition.1.weight"), self._pview(buf, "fc_condition.1.bias"))
        if self.d_condition > 0:
            return (self._pview(buf, "fc_condition.weight"), self._pview(buf, "fc_condition.bias"), None, None)
        return (None, None, None, None)

    def _check_inputs(self, tokens, cond):
        if not self._flat.is_cuda:
            raise RuntimeError("MusicTransformerHIP runs only on a HIP device (model.to('cuda')); "
                               "there is no CPU fallback")
        if tokens.dim() != 2:
            raise ValueError("tokens must be [batch, seq]")
        tokens = tokens.to(device=self._flat.device, dtype=torch.int64).contiguous()
        B, Ltok = tokens.shape
        shift = 2 if self.token_conditioning else 0
        if Ltok + shift > self.max_seq:
            raise RuntimeError("sequence length %d exceeds max_seq %d" % (Ltok + shift, self.max_seq))
        if cond is None:
            cond = torch.full((B, 2), float("nan"))
        cond = cond.to(device=self._flat.device, dtype=torch.float32).contiguous()
        return tokens, cond, B, Ltok, Ltok + shift

    @property
    def resid_lo_bits(self):
        if getattr(self, "_resid_lo_bits_set", 0):
            return self._resid_lo_bits_set
        if self._resid_lo_bits_env in (8, 16):
            return self._resid_lo_bits_env
        return 8 if self.compute_dtype == torch.bfloat16 else 16

    @resid_lo_bits.setter
    def resid_lo_bits(self, bits):                      # (tools/ab_step.py flips it between arms; drop the workspaces afterwards)
        self._resid_lo_bits_set = int(bits)

    def _forward_impl(self, tokens, cond, B, Ltok, Lm, save, p_drop, seed, logits_out):
        dt = self.compute_dtype
        d, di, H, dh, V, N, M = (self.embedding_dim, self.d_inner, self.num_head, self.dh, self.head_size,
                                 self.num_layer, self.max_seq)
        T = B * Lm
        ws = self._workspace(B, Lm, save)
        ws.stamp = getattr(ws, "stamp", 0) + 1     # a saved-activation workspace is overwritten by every forward that uses it
        self._refresh_weights()
        f = self._flat
        shift = Lm - Ltok
        ops.key_pad_mask(ws.key_pad, tokens, B, Ltok, shift, self.pad_token)
        cw0, cb0, cw1, cb1 = self._cond_params(f)
        ops.embed_fwd(ws.h[0], tokens, cond, self._pview(f, "embedding.weight"), cw0, cb0, cw1, cb1, self._pe,
                      self._mode(), B, Ltok, d, self.d_condition, p_drop, seed, out_lo=ws.hlo[0] if self._resid_lo_mode != 3 else None)
        nh = len(ws.h)
        hlo_h = ws.hlo[0] if self._resid_lo_mode != 3 else None           # low half of the layer stream
        hlo_o = ws.hlo[1] if self._resid_lo_mode != 2 else None           # low half of LayerNorm1's output
        for i in range(N):
            W = self._prep["layers"][i]
            Lw = ws.layers[i if save else 0]
            x = ws.h[i % nh if not save else i]
            y = ws.h[(i + 1) % nh if not save else i + 1]
            p = f"enc_layers.{i}."
            ops.gemm_nt(x, W["Wqkv"], Lw.qkv, bias=W["bqkv"], M=T, N=3 * d, K=d, dtype=dt)
            ops.rga_fwd(Lw.qkv, W["Epk"], ws.key_pad if self.causal else None, Lw.att, Lw.lse, B, Lm, H, dh, M,
                        causal=self.causal, PT=Lw.PT, MT=Lw.MT)   # bidirectional (mask=None in the reference): no pad mask either
            ops.gemm_nt(Lw.att, W["Wo"], ws.tmp, bias=self._pview(f, p + "rga.fc.bias"), M=T, N=d, K=d, dtype=dt)
            ops.resid_ln_fwd(x, ws.tmp, self._pview(f, p + "layernorm1.weight"), self._pview(f, p + "layernorm1.bias"),
                             Lw.o1, Lw.s1, Lw.st1, T, d, self.LN_EPS, p_drop, seed, 1 + 2 * i, x_lo=hlo_h, y_lo=hlo_o)
            if Lw.rmask is not None:
                ops.gemm_nt_relu_mask(Lw.o1, W["W1"], Lw.hid, Lw.rmask, bias=self._pview(f, p + "FFN_pre.bias"), M=T, N=di, K=d, dtype=dt)
            else:
                ops.gemm_nt(Lw.o1, W["W1"], Lw.hid, bias=self._pview(f, p + "FFN_pre.bias"), M=T, N=di, K=d,
                            flags=ops.ME_EPI_RELU, dtype=dt)
            ops.gemm_nt(Lw.hid, W["W2"], ws.tmp, bias=self._pview(f, p + "FFN_suf.bias"), M=T, N=d, K=di, dtype=dt)
            ops.resid_ln_fwd(Lw.o1, ws.tmp, self._pview(f, p + "layernorm2.weight"), self._pview(f, p + "layernorm2.bias"),
                             y, Lw.s2, Lw.st2, T, d, self.LN_EPS, p_drop, seed, 2 + 2 * i, x_lo=hlo_o, y_lo=hlo_h)
        hN = ws.h[N % nh if not save else N]
        out = logits_out if logits_out is not None else ws.logits
        ops.gemm_nt(hN, self._prep["head"]["Wf"], out, bias=self._pview(f, self._HEAD_B), M=T, N=V, K=d,
                    flags=ops.ME_EPI_OUT_F32 if out.dtype == torch.float32 else 0, dtype=dt)
        return ws

    # ------------------------------------------------------------------ engine: backward
    @staticmethod
    def backward_hook_sequence(num_layer):
        """The bucket_hook arguments in the order _backward_impl issues them (bucket ids of bucket_ranges(): 0 = embedding,
        i + 1 = layer i, num_layer + 1 = head; -1 = a comm window opens: the next layer's attention backward follows).  The
        reducer tests replay exactly this sequence on the CPU; tests/test_ddp_gpu.py asserts the engine emits it."""
        seq = [num_layer + 1]
        for i in reversed(range(num_layer)):
            seq += [-1, i + 1]
        seq.append(0)
        return seq

    def _backward_impl(self, ws, tokens, cond, B, Ltok, Lm, p_drop, seed, gflat, bucket_hook=None, head_bias_done=False):
        """dlogits in ws.dlogits (T, padded ld) -> accumulates every parameter gradient into gflat."""
        dt = self.compute_dtype
        d, di, H, dh, V, N, M = (self.embedding_dim, self.d_inner, self.num_head, self.dh, self.head_size,
                                 self.num_layer, self.max_seq)
        T = B * Lm
        f = self._flat
        ldv = ws.dlogits.shape[1]
        head = self._prep["head"]
        gv = lambda name: self._pview(gflat, name)
        hN = ws.h[N]
        # Weight gradients (dY^T X) run in line on the caller's stream: forking them onto a side stream measured
        # +0.5 % (every kernel of the chain already fills all 256 CUs) and an asynchronous partial-tile reduce
        # measured 1.5 % slower -- both experiments were removed with the library-owned workspace (round 2).
        tnws = self._tn_workspace(T)

        # The four weight gradients of a layer are deferred to the end of the layer's backward and run as ONE grouped
        # launch (me_gemm_tn_acc_group): the token split is #CUs / 48 tiles = 5 ranges instead of 16-64 per product, a
        # quarter of the partial-tile traffic, two launches instead of eight.  (dY buffers dC / dhid / dC2 / dqkv and the
        # saved activations they pair with stay valid until the next layer's backward starts.)
        pending = []

        def wgrad(dY, X, gW, gb, N, K):
            pending.append((dY, X, gW, gb, N, K))

        def flush_wgrads():
            if pending:
                ops.gemm_tn_acc_group(pending, T, dt, ws=tnws)
                del pending[:]

        # (the head's product stays a launch of its own: grouped with the last layer it makes 56 tiles = 4 token ranges on
        # 224 of the 256 CUs -- measured no faster than 48 tiles x 5 ranges on 240 CUs plus the small head launch)
        wgrad(ws.dlogits, hN, gv(self._HEAD_W), None if head_bias_done else gv(self._HEAD_B), N=V, K=d)
        flush_wgrads()
        ops.gemm_nt(ws.dlogits, head["WfT"], ws.dA, M=T, N=d, K=ldv, dtype=dt)
        if bucket_hook:
            bucket_hook(N + 1)
        dy = ws.dA
        for i in reversed(range(N)):
            W = self._prep["layers"][i]
            Lw = ws.layers[i]
            p = f"enc_layers.{i}."
            x = ws.h[i]
            # LN2 + FFN
            ops.resid_ln_bwd(dy, Lw.s2, Lw.st2, self._pview(f, p + "layernorm2.weight"), ws.dB, ws.dC,
                             gv(p + "layernorm2.weight"), gv(p + "layernorm2.bias"), T, d, p_drop, seed, 2 + 2 * i)
            wgrad(ws.dC, Lw.hid, gv(p + "FFN_suf.weight"), gv(p + "FFN_suf.bias"), N=d, K=di)
            if Lw.rmask is not None:
                ops.gemm_nt_relu_mask(ws.dC, W["W2T"], ws.dhid, Lw.rmask, M=T, N=di, K=d, backward=True, dtype=dt)
            else:
                ops.gemm_nt(ws.dC, W["W2T"], ws.dhid, gate=Lw.hid, M=T, N=di, K=d, flags=ops.ME_EPI_RELU_BWD, dtype=dt)
            wgrad(ws.dhid, Lw.o1, gv(p + "FFN_pre.weight"), gv(p + "FFN_pre.bias"), N=di, K=d)
            ops.gemm_nt(ws.dhid, W["W1T"], ws.dA, add=ws.dB, M=T, N=d, K=di, dtype=dt)          # d(o1) total
            # LN1 + attention
            ops.resid_ln_bwd(ws.dA, Lw.s1, Lw.st1, self._pview(f, p + "layernorm1.weight"), ws.dB, ws.dC2,
                             gv(p + "layernorm1.weight"), gv(p + "layernorm1.bias"), T, d, p_drop, seed, 1 + 2 * i)
            wgrad(ws.dC2, Lw.att, gv(p + "rga.fc.weight"), gv(p + "rga.fc.bias"), N=d, K=d)
            ops.gemm_nt(ws.dC2, W["WoT"], ws.dA, M=T, N=d, K=d, dtype=dt)                        # d(att)
            if bucket_hook:
                bucket_hook(-1)            # comm window: ~0.5 ms of attention-backward kernels follow (ddp.GradAllReducer)
            ops.rga_bwd(Lw.qkv, W["Epk"], Lw.att, Lw.lse, ws.dA, ws.dqkv, gv(p + "rga.E"), ws.delta, Lw.PT, Lw.MT, ws.dGT,
                        B, Lm, ws.Lp, H, dh, M, causal=self.causal, overlap=self.attn_bwd_overlap)
            o, _, _ = self._slices[p + "rga.Wq.weight"]
            ob, _, _ = self._slices[p + "rga.Wq.bias"]
            wgrad(ws.dqkv, x, gflat[o:o + 3 * d * d].view(3 * d, d), gflat[ob:ob + 3 * d], N=3 * d, K=d)
            ops.gemm_nt(ws.dqkv, W["WqkvT"], ws.dA, add=ws.dB, M=T, N=d, K=3 * d, dtype=dt)     # d(x) total
            dy = ws.dA
            flush_wgrads()
            if bucket_hook:
                bucket_hook(i + 1)
        g = self._cond_params(gflat)
        if getattr(self, "_emb_ws", None) is None or self._emb_ws.device != dy.device:
            self._emb_ws = ops.embed_bwd_ws(dy.device)                 # zeroed once; the library leaves it zeroed
        try:
            ops.embed_bwd(dy, tokens, cond, gv("embedding.weight"), g[0], g[1], g[2], g[3], self._mode(), B, Ltok, d,
                          self.d_condition, self.pad_token, p_drop, seed, ws=self._emb_ws)
        except Exception:
            # a launch that failed between the gather and the frequent-token kernel can leave (token, count) entries
            # behind; the "library leaves it zeroed" contract only holds for calls that returned 0
            self._emb_ws = None
            raise
        flush_wgrads()                                     # optimizer / next forward see every gradient
        if bucket_hook:
            bucket_hook(0)

    def _tn_workspace(self, T):
        """Caller-owned partial-tile workspace of the weight-gradient GEMMs (me_workspace_bytes):
        one buffer sized for the largest (T, N, K) of the step, reused by every launch on the stream."""
        d, di, V = self.embedding_dim, self.d_inner, self.head_size
        shapes = [(3 * d, d), (d, d), (di, d), (d, di), (V, d)]
        need = max(ops.workspace_bytes(ops.ME_WS_GEMM_TN, T, n, k, self.compute_dtype) for n, k in shapes)
        if all(k % 256 == 0 for _, k in shapes):          # the grouped launch of a layer's four products (+ the head's)
            for grp in (shapes[:4], shapes):
                need = max(need, ops.workspace_bytes(ops.ME_WS_GEMM_TN_GROUP, T, ops.tn_group_tiles(grp), 0, self.compute_dtype))
        if need == 0:
            return None
        buf = getattr(self, "_tnws", None)
        if buf is None or buf.numel() < need or buf.device != self._flat.device:
            buf = self._tnws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
        return buf

    # ------------------------------------------------------------------ public API
    def _next_seed(self):
        self._fwd_count += 1
        return (self._step_seed * 0x9E3779B97F4A7C15 + self._fwd_count * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def seed_dropout(self, seed):
        self._step_seed = int(seed) & 0xFFFFFFFF
        self._fwd_count = 0

    def forward(self, x, condition=None):
        """model(x, condition) -> logits f32 [B, L(+2), V]   (music_multi.py:84-108)."""
        tokens, cond, B, Ltok, Lm = self._check_inputs(x, condition)
        V = self.head_size
        p_drop = self.dropout_p if self.training else 0.0
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if not need_grad:
            out = torch.empty(B * Lm, V, dtype=torch.float32, device=self._flat.device)
            self._forward_impl(tokens, cond, B, Ltok, Lm, False, p_drop, self._next_seed(), out)
            return out.view(B, Lm, V)
        params = [p for g in self._param_order() for _, p in g]
        return _EngineFn.apply(self, tokens, cond, (B, Ltok, Lm), p_drop, self._next_seed(), *params)

    def loss_and_backward(self, x, condition, target, grad_scale=1.0, bucket_hook=None, backward=True,
                          return_logits=False, loss_scale=None):
        """Fused train-step front half: forward, CrossEntropyLoss(ignore_index=pad) (mean over
        non-pad targets), backward into `flat_grads` (+=).  Returns the (unscaled) loss as a device scalar
        (no host sync).  Replaces Runner.forward_pass + loss.backward() (train.py:276-292,317).
        loss_scale: f32 device scalar multiplied into the backward (optim.LossScaler.scale_tensor -- the f16 tier's
        `scaler.scale(loss).backward()`; the optimiser step divides it out again)."""
        tokens, cond, B, Ltok, Lm = self._check_inputs(x, condition)
        target = target.to(device=self._flat.device, dtype=torch.int64).contiguous().view(-1)
        T, V = B * Lm, self.head_size
        if target.numel() != T:
            raise ValueError("target has %d elements, model output has %d positions" % (target.numel(), T))
        p_drop = self.dropout_p if self.training else 0.0
        seed = self._next_seed()
        ws = self._forward_impl(tokens, cond, B, Ltok, Lm, True, p_drop, seed, None)
        ws.acc.zero_()
        ops.ce_fwd(ws.logits, target, ws.row_lse, ws.acc[0:1], ws.acc[1:2], T, V, self.pad_token)
        loss = ws.acc[0] / ws.acc[1]
        if backward:
            fuse_db = ops.ce_bwd_fuses_dbias(ws.logits, ws.dlogits)     # bf16 tier: head bias gradient from the f32 dlogits
            ops.ce_bwd(ws.logits, target, ws.row_lse, ws.dlogits, ws.acc[1:2], grad_scale, T, V, self.pad_token,
                       dbias=self._pview(self._gflat, self._HEAD_B) if fuse_db else None, loss_scale=loss_scale)
            self._backward_impl(ws, tokens, cond, B, Ltok, Lm, p_drop, seed, self._gflat, bucket_hook, head_bias_done=fuse_db)
        if return_logits:                                   # f32 [B, Lm, V] copy of the workspace logits
            return loss, ws.logits[:, :V].float().view(B, Lm, V)
        return loss


class _EngineFn(torch.autograd.Function):
    """Bridges the explicit engine into autograd so the reference's
    `loss = CE(model(x, c)); loss.backward()` pattern works unchanged (train.py:288-317)."""

    @staticmethod
    def forward(ctx, model, tokens, cond, dims, p_drop, seed, *params):
        B, Ltok, Lm = dims
        V = model.head_size
        out = torch.empty(B * Lm, V, dtype=torch.float32, device=model._flat.device)
        ws = model._forward_impl(tokens, cond, B, Ltok, Lm, True, p_drop, seed, out)
        ctx.model, ctx.ws, ctx.args = model, ws, (tokens, cond, B, Ltok, Lm, p_drop, seed)
        ctx.stamp = ws.stamp
        return out.view(B, Lm, V)

    @staticmethod
    def backward(ctx, dlogits):
        model, ws = ctx.model, ctx.ws
        tokens, cond, B, Ltok, Lm, p_drop, seed = ctx.args
        if ctx.stamp != ws.stamp:
            raise RuntimeError("backward() after another grad-enabled forward() of the same (batch, length): the "
                               "activations of the HIP engine live in one workspace per shape (one forward in flight "
                               "per shape; no-grad / eval forwards and other shapes use their own workspaces)")
        V = model.head_size
        ws.dlogits[:, :V].copy_(dlogits.reshape(B * Lm, V))
        gbuf = torch.zeros_like(model._gflat)
        model._backward_impl(ws, tokens, cond, B, Ltok, Lm, p_drop, seed, gbuf)
        grads = [model._pview(gbuf, name) for g in model._param_order() for name, _ in g]
        return (None, None, None, None, None, None) + tuple(grads)


class MusicTransformerMulti(MusicTransformerHIP):
    """none / discrete_token / continuous_concat  (models/music_multi.py:41-73 kwargs)."""

    def __init__(self, embedding_dim=None, d_inner=None, d_condition=None, vocab_size=None, num_layer=None,
                 num_head=None, max_seq=None, dropout=None, pad_token=None, compute_dtype="bf16"):
        super().__init__(embedding_dim=embedding_dim, d_inner=d_inner, d_condition=d_condition,
                         vocab_size=vocab_size, num_layer=num_layer, num_head=num_head, max_seq=max_seq,
                         dropout=dropout, pad_token=pad_token, token_conditioning=False,
                         compute_dtype=compute_dtype)


class MusicTransformerContinuousToken(MusicTransformerHIP):
    """continuous_token  (models/music_continuous_token.py:32-66 kwargs)."""

    def __init__(self, embedding_dim=None, d_inner=None, vocab_size=None, num_layer=None, num_head=None,
                 max_seq=None, dropout=None, pad_token=None, has_start_token=True, n_conditions=2,
                 compute_dtype="bf16"):
        if n_conditions != 2:
            raise ValueError("n_conditions must be 2 (valence, arousal)")
        super().__init__(embedding_dim=embedding_dim, d_inner=d_inner, d_condition=-1, vocab_size=vocab_size,
                         num_layer=num_layer, num_head=num_head, max_seq=max_seq, dropout=dropout,
                         pad_token=pad_token, token_conditioning=True, compute_dtype=compute_dtype)
        self.has_start_token = has_start_token
        self.n_conditions = n_conditions


class MusicRegression(MusicTransformerHIP):
    """Evaluation model of the reference (models/music_regression.py:34-92): token embedding * sqrt(d) + PE, the same
    post-LN encoder layers with BIDIRECTIONAL relative attention (`no_mask=True`: mask=None, nothing masked, relative
    term only for key <= q), and tanh(Linear(d, output_size)) of position 0 (the <CLS> token the loader prepends).
    Same constructor kwargs and `state_dict` keys (`fc.0.weight` / `fc.0.bias`).  forward() is inference (no autograd
    graph); training goes through loss_and_backward() (L1 loss, bidirectional attention backward).  With
    `no_mask=False` the reference applies the causal + pad mask of the language model, which is the base class."""
    _HEAD_W, _HEAD_B = "fc.0.weight", "fc.0.bias"
    _logits_f32 = True

    def __init__(self, embedding_dim=None, d_inner=None, vocab_size=None, num_layer=None, num_head=None,
                 max_seq=None, dropout=None, pad_token=None, output_size=None, d_condition=-1, no_mask=True,
                 compute_dtype="bf16"):
        assert d_condition is None or d_condition <= 0                       # music_regression.py:41
        self.output_size = output_size
        self.no_mask = no_mask
        super().__init__(embedding_dim=embedding_dim, d_inner=d_inner, d_condition=-1, vocab_size=vocab_size,
                         num_layer=num_layer, num_head=num_head, max_seq=max_seq, dropout=dropout, pad_token=pad_token,
                         token_conditioning=False, compute_dtype=compute_dtype, head_size=output_size,
                         causal=not no_mask)

    def _make_head(self, d, n_out):
        return nn.Sequential(_Lin(d, n_out), nn.Tanh())

    def _head_linear(self):
        return self.fc[0]

    def init_weights(self):
        """music_regression.py:72-74: only the embedding is re-initialised."""
        with torch.no_grad():
            self.embedding.weight.uniform_(-0.1, 0.1)

    def forward(self, x):
        """tokens [B, L] -> tanh(head(x[:, 0])) f32 [B, output_size]."""
        tokens, cond, B, Ltok, Lm = self._check_inputs(x, None)
        p_drop = self.dropout_p if self.training else 0.0
        out = torch.empty(B * Lm, self.head_size, dtype=torch.float32, device=self._flat.device)
        with torch.no_grad():
            self._forward_impl(tokens, cond, B, Ltok, Lm, False, p_drop, self._next_seed(), out)
            return torch.tanh(out.view(B, Lm, self.head_size)[:, 0, :]).clone()

    def loss_and_backward(self, x, target, grad_scale=1.0, bucket_hook=None, backward=True, loss_scale=None):
        """L1Loss(tanh(head(x[:, 0])), target) (train.py:282-284: `self.l1_loss(output, condition)`, mean over B x
        output_size) and its backward into `flat_grads` (+=).  target: [B, output_size] (valence, arousal)."""
        tokens, cond, B, Ltok, Lm = self._check_inputs(x, None)
        n_out = self.head_size
        target = torch.as_tensor(target).to(device=self._flat.device, dtype=torch.float32).reshape(B, n_out)
        p_drop = self.dropout_p if self.training else 0.0
        seed = self._next_seed()
        ws = self._forward_impl(tokens, cond, B, Ltok, Lm, True, p_drop, seed, None)
        ldv = ws.logits.shape[1]
        y = torch.tanh(ws.logits.view(B, Lm, ldv)[:, 0, :n_out])
        diff = y - target
        loss = diff.abs().mean()
        if backward:
            dz = torch.sign(diff) * (1.0 - y * y) * (float(grad_scale) / diff.numel())
            if loss_scale is not None:
                dz = dz * loss_scale
            ws.dlogits.zero_()                                   # only position 0 of every sequence feeds the head
            ws.dlogits.view(B, Lm, ldv)[:, 0, :n_out] = dz.to(ws.dlogits.dtype)
            self._backward_impl(ws, tokens, cond, B, Ltok, Lm, p_drop, seed, self._gflat, bucket_hook)
        return loss
