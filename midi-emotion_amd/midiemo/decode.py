"""KV-cached incremental decode for generate().

The reference recomputes the whole window for every new token
(generate.py:99-119: O(T^2) per step).  Because logits at position t do not
depend on later tokens nor on the sequence length (prefix invariance, SURVEY 8a
A15 -- pinned by tests), the model call can be made incremental: per layer the
new token's k, v are appended to a cache [B, H, max_seq, dh] and a single-query
relative-global-attention step runs against it (me_rga_decode_step; the relative
row needed for key j is E[M-1-(t-j)]).

Small-batch projections use the weight-streaming me_gemv_small kernel (decode
is HBM/latency bound: the bf16 weights, 41 MB at the headline model, are read
once per step).  Everything stays on the device; one step issues ~8 kernels
per layer on the current stream.

Cache validity: the cache is only valid while absolute positions are stable.
Once the reference's sliding window starts dropping the oldest token every
position shifts by one (generate.py:101-103 + music_multi.py:163) and
`continuous_concat` with a per-step varying condition changes every cached
embedding; in both cases the caller falls back to full recompute (exact
reference semantics) -- see generate.py.
"""
import torch

from . import ops


class DecodeSession:
    def __init__(self, model, batch_size):
        if model.training:
            raise RuntimeError("DecodeSession needs model.eval()")
        self.m = model
        self.B = int(batch_size)
        m = model
        dev, dt = m.flat_params.device, m.compute_dtype
        d, di, H, dh, V = m.embedding_dim, m.d_inner, m.num_head, m.dh, m.vocab_size
        B = self.B
        e = lambda *s, dtype=dt: torch.empty(*s, dtype=dtype, device=dev)
        self.kc = [torch.zeros(B, H, m.max_seq, dh, dtype=dt, device=dev) for _ in range(m.num_layer)]
        self.vc = [torch.zeros(B, H, m.max_seq, dh, dtype=dt, device=dev) for _ in range(m.num_layer)]
        self.x, self.y = e(B, d), e(B, d)
        self.qkv, self.att, self.tmp, self.o1, self.hid = e(B, 3 * d), e(B, d), e(B, d), e(B, d), e(B, di)
        self.logits = e(B, V, dtype=torch.float32)
        self.t = 0                      # next model position to be written
        self._pos_dev = None            # device-side position: set while a step is issued for graph capture / replay
        self._graph = None
        m._refresh_weights()

    def reset(self):
        self.t = 0

    # -------------------------------------------------------------- projections
    def _proj(self, x, W, bias, y, N, K, flags=0):
        if self.B <= 8:
            ops.gemv_small(x, W, bias, y, self.B, N, K, flags=flags, dtype=self.m.compute_dtype)
        else:
            ops.gemm_nt(x, W, y, bias=bias, M=self.B, N=N, K=K, flags=flags, dtype=self.m.compute_dtype)

    def _layers_and_head(self):
        m, B, t = self.m, self.B, self.t
        d, di, H, dh, V, M = m.embedding_dim, m.d_inner, m.num_head, m.dh, m.vocab_size, m.max_seq
        f = m.flat_params
        x, y = self.x, self.y
        for i in range(m.num_layer):
            W = m._prep["layers"][i]
            p = f"enc_layers.{i}."
            self._proj(x, W["Wqkv"], W["bqkv"], self.qkv, 3 * d, d)
            ops.rga_decode_step(self.qkv, self.kc[i], self.vc[i], W["E"], None, 0, self.att, B, H, dh, M, M, t,
                                t_dev=self._pos_dev)
            self._proj(self.att, W["Wo"], m._pview(f, p + "rga.fc.bias"), self.tmp, d, d)
            ops.resid_ln_fwd(x, self.tmp, m._pview(f, p + "layernorm1.weight"), m._pview(f, p + "layernorm1.bias"),
                             self.o1, None, None, B, d, m.LN_EPS, 0.0, 0, 0)
            self._proj(self.o1, W["W1"], m._pview(f, p + "FFN_pre.bias"), self.hid, di, d, flags=ops.ME_EPI_RELU)
            self._proj(self.hid, W["W2"], m._pview(f, p + "FFN_suf.bias"), self.tmp, d, di)
            ops.resid_ln_fwd(self.o1, self.tmp, m._pview(f, p + "layernorm2.weight"), m._pview(f, p + "layernorm2.bias"),
                             y, None, None, B, d, m.LN_EPS, 0.0, 0, 0)
            x, y = y, x
        self._proj(x, m._prep["head"]["Wf"], m._pview(f, "fc.bias"), self.logits, V, d, flags=ops.ME_EPI_OUT_F32)
        if self._pos_dev is None:
            self.t += 1
        return self.logits

    # -------------------------------------------------------------- public steps
    def prefill_condition_slots(self, cond):
        """continuous_token: model positions 0 and 1 are the two condition vectors
        (music_continuous_token.py:92-97)."""
        m = self.m
        if not m.token_conditioning:
            return
        if self.t != 0:
            raise RuntimeError("condition slots must be written first")
        cond = cond.to(device=m.flat_params.device, dtype=torch.float32).contiguous()
        f = m.flat_params
        cw0, cb0, cw1, cb1 = m._cond_params(f)
        d = m.embedding_dim
        both = torch.empty(self.B, 2, d, dtype=m.compute_dtype, device=f.device)
        dummy = torch.zeros(self.B, 1, dtype=torch.int64, device=f.device)
        ops.embed_fwd(both, dummy, cond, m._pview(f, "embedding.weight"), cw0, cb0, cw1, cb1, m._pe,
                      ops.ME_COND_TOKEN, self.B, 0, d, 0, 0.0, 0)
        for s in range(2):
            self.x.copy_(both[:, s])
            self._layers_and_head()

    def step(self, tokens, cond=None):
        """Feed one token per sequence (int64 [B]) at the next position; returns logits f32 [B, V]
        (a view of an internal buffer, valid until the next step)."""
        m = self.m
        if self.t >= m.max_seq:
            raise RuntimeError("decode position %d exceeds max_seq %d" % (self.t, m.max_seq))
        f = m.flat_params
        tokens = tokens.to(device=f.device, dtype=torch.int64).reshape(self.B, 1).contiguous()
        d = m.embedding_dim
        return self._embed_and_run(tokens, cond, m._pe[self.t:], None)

    def _embed_and_run(self, tokens, cond, pe_t, pos_dev):
        m = self.m
        f = m.flat_params
        d = m.embedding_dim
        if m.d_condition > 0:
            cond = cond.to(device=f.device, dtype=torch.float32).contiguous()
            cw0, cb0, _, _ = m._cond_params(f)
            ops.embed_fwd(self.x, tokens, cond, m._pview(f, "embedding.weight"), cw0, cb0, None, None, pe_t,
                          ops.ME_COND_CONCAT, self.B, 1, d, m.d_condition, 0.0, 0, pos_dev=pos_dev)
        else:
            ops.embed_fwd(self.x, tokens, None, m._pview(f, "embedding.weight"), None, None, None, None, pe_t,
                          ops.ME_COND_NONE, self.B, 1, d, 0, 0.0, 0, pos_dev=pos_dev)
        return self._layers_and_head()

    # -------------------------------------------------------------- device-resident greedy loop
    def greedy_run(self, tokens, n_steps, cond=None, special=None, use_graph=True):
        """Feed `tokens` (int64 [B]) at the next position, then keep feeding the arg-max token back for n_steps
        steps in total, entirely on the device (generate.py:99-189 with top_k = 1): the position lives in device
        memory (me_embed_fwd pos_dev, me_rga_decode_step t_dev), me_greedy_pick writes the next input token and
        me_decode_commit appends it to a history buffer and advances the position.  The step is captured once as a
        HIP graph and replayed, so a token costs one graph launch instead of ~45 kernel launches from Python.
        Returns the generated ids int64 [B, n_steps] (on the device)."""
        m = self.m
        dev = m.flat_params.device
        n_steps = int(n_steps)
        if self.t + n_steps > m.max_seq:
            raise RuntimeError("decode positions %d..%d exceed max_seq %d" % (self.t, self.t + n_steps, m.max_seq))
        if not hasattr(self, "_tok"):
            self._tok = torch.zeros(self.B, 1, dtype=torch.int64, device=dev)
            self._hist = torch.zeros(self.B, m.max_seq, dtype=torch.int64, device=dev)
            self._pos = torch.zeros(1, dtype=torch.int32, device=dev)
            self._cond = torch.zeros(self.B, 2, dtype=torch.float32, device=dev)
            self._special = None
        self._tok.copy_(tokens.to(device=dev, dtype=torch.int64).reshape(self.B, 1))
        self._pos.fill_(self.t)
        if m.d_condition > 0:
            self._cond.copy_(cond.to(device=dev, dtype=torch.float32))
        sp = None if special is None else special.to(device=dev, dtype=torch.int32).contiguous()
        if (sp is None) != (self._special is None) or (sp is not None and not torch.equal(sp, self._special)):
            self._special, self._graph = sp, None          # the special-id list is baked into the captured step

        def one_step():
            self._pos_dev = self._pos
            try:
                self._embed_and_run(self._tok, self._cond if m.d_condition > 0 else None, m._pe, self._pos)
                ops.greedy_pick(self.logits, m.vocab_size, self._special, self._tok, self.B)
                ops.decode_commit(self._tok, self._hist, self._pos, self.B)
            finally:
                self._pos_dev = None

        t0 = self.t
        done = 0
        if use_graph and self._graph is None and n_steps > 2:
            one_step()                                      # warm-up outside the capture (lazy allocations, attributes)
            done = 1
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step()
            self._graph = g                                 # capturing does not execute
        for _ in range(n_steps - done):
            if use_graph and self._graph is not None:
                self._graph.replay()
            else:
                one_step()
        self.t = t0 + n_steps
        return self._hist[:, t0:t0 + n_steps]
