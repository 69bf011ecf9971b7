"""KV-cached incremental decode for generate().

The reference recomputes the whole window for every new token
(generate.py:99-119: O(T^2) per step).  Because logits at position t do not
depend on later tokens nor on the sequence length (prefix invariance, SURVEY 8a
A15 -- pinned by tests), the model call can be made incremental: per layer the
new token's k, v are appended to a cache [B, H, max_seq, dh] and a single-query
relative-global-attention step runs against it (the relative row needed for key
j is E[M-1-(t-j)]).

A step is HBM / latency bound (41 MB of bf16 weights + the K / V cache are
streamed once, the arithmetic is negligible), so what counts is the number of
dependent kernels and how widely each one spreads its stream.  Per layer five
fused kernels of csrc/me_decode.hip: {LayerNorm2 of the previous layer + q|k|v
projection + cache append}, {key-split attention}, {combine + Wo + residual},
{LayerNorm1 + FFN_pre + ReLU}, {FFN_suf + residual}; then {LayerNorm2 +
vocabulary head}.  The residual stream stays f32 on the device ([B, d] buffers).

Cache validity: the cache is only valid while absolute positions are stable.
Once the reference's sliding window starts dropping the oldest token every
position shifts by one (generate.py:101-103 + music_multi.py:163) and
`continuous_concat` with a per-step varying condition changes every cached
embedding; in both cases the caller falls back to full recompute (exact
reference semantics) -- see generate.py.
"""
import os

import torch

from . import ops

ROWS = 8            # sequences per kernel call (me_dec_*: Mr <= 8); larger batches run in row chunks


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
        f32 = torch.float32
        self.kc = [torch.zeros(B, H, m.max_seq, dh, dtype=dt, device=dev) for _ in range(m.num_layer)]
        self.vc = [torch.zeros(B, H, m.max_seq, dh, dtype=dt, device=dev) for _ in range(m.num_layer)]
        self.xlo = dt != f32                                   # bf16 tier: condition slots are fed as hi + lo rows
        self.q, self.hid = e(B, d), e(B, di)
        self.xres, self.o1res = e(B, d, dtype=f32), e(B, d, dtype=f32)     # f32 residual inputs (LayerNorm outputs)
        self.s1, self.s2 = e(B, d, dtype=f32), e(B, d, dtype=f32)           # f32 pre-norm sums
        # key splits of the attention: B*H*nsplit blocks should cover the chip (256 CUs)
        self.nsplit = int(os.environ.get("MIDIEMO_DEC_NSPLIT", "0")) or max(1, min(8, 256 // max(1, min(B, ROWS) * H)))
        # me_dec_attn keeps one split's scores in LDS: at most 2048 keys per split
        self.nsplit = min(8, max(self.nsplit, -(-m.max_seq // 2048)))
        if -(-m.max_seq // self.nsplit) > 2048:
            raise RuntimeError("max_seq %d exceeds the cached decode's %d keys (8 splits of 2048)" % (m.max_seq, 8 * 2048))
        self.part = e(B * H, self.nsplit, dh + 4, dtype=f32)       # ME_DEC_PART_REC(dh): (max, sum, 0, 0, o[dh])
        # fused qkv + attention stage (me_dec_ln_qkv_attn): the last split is the new key, so it needs >= 2 splits
        # (the kernel's own preconditions, me_dec_ln_qkv_attn / me_dec_embed_qkv_attn: d % 8 == 0, d <= 1024, dh in {32, 48, 64};
        # any other shape takes the separate me_dec_qkv + me_dec_attn launches)
        self.fused = os.environ.get("MIDIEMO_DEC_UNFUSED", "0") in ("", "0") and d <= 1024 and d % 8 == 0 and dh in (32, 48, 64) and \
            self.nsplit >= 2 and -(-m.max_seq // (self.nsplit - 1)) <= 2048
        self.logits = e(B, V, dtype=f32)
        # One persistent launch per token (me_dec_token, round 6) instead of 4 per layer + head: the stages exchange self-validating
        # records, no kernel boundaries.  Serves token-fed steps of up to 4 sequences; nsplit a power of two.  Bit-identical to the
        # per-stage launches (tests) but not faster on this part (0.163 vs 0.154 ms per token, profiles/r06_decode_token.txt:
        # six exchanges of ~2.2 us per layer against four kernel boundaries): opt-in, MIDIEMO_DEC_TOKEN=1.
        self.token_kernel = False
        self._tok_table = None
        if self.fused and os.environ.get("MIDIEMO_DEC_TOKEN", "0") not in ("", "0") and B <= ops._lib.ME_DEC_TOKEN_ROWS and \
                m.num_layer <= ops._lib.ME_DEC_MAX_LAYERS and di % 8 == 0 and (m.d_condition <= 0 or (m.d_condition % 4 == 0 and m.d_condition < d)):
            ns = 8 if self.nsplit >= 8 else (4 if self.nsplit >= 4 else 2)
            blocks = ops.dec_token_blocks(dh, d, di, dt)
            self._tok_blocks = int(os.environ.get("MIDIEMO_DEC_TOKEN_BLOCKS", "0"))       # 0 = one block per CU (tests: fewer)
            if self._tok_blocks > 0:
                blocks = min(blocks, self._tok_blocks)
            if dh % ns == 0 and (dh // ns) % 2 == 0 and B * H * ns <= blocks and -(-m.max_seq // (ns - 1)) <= 2048:
                if ns != self.nsplit:
                    self.nsplit = ns
                    self.part = e(B * H, self.nsplit, dh + 4, dtype=f32)
                nbytes = ops.workspace_bytes(ops.ME_WS_DEC_TOKEN, B, di, d, dt)
                self._tok_ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)      # zero before the first use (epoch, records)
                self.token_kernel = nbytes > 0 and self._tok_ws.data_ptr() % 256 == 0
        self.t = 0                      # next model position to be written
        self._pos_dev = None            # device-side position: set while a step is issued for graph capture / replay
        self._graph = None
        m._refresh_weights()

    def reset(self):
        self.t = 0

    def _layers_and_head(self, tokens=None, cond=None, x_hi=None, x_lo=None):
        """One position through all layers and the head.  The first layer's input is either the embedding of `tokens`
        (int64 [B, 1], computed inside the first kernel: me_dec_embed_qkv) or the rows x_hi (+ x_lo) [B, d] in T."""
        m, B, t = self.m, self.B, self.t
        d, di, H, dh, V, M = m.embedding_dim, m.d_inner, m.num_head, m.dh, m.vocab_size, m.max_seq
        f, dt, eps, ns = m.flat_params, m.compute_dtype, m.LN_EPS, self.nsplit
        pv = lambda name: m._pview(f, name)
        if tokens is not None and m.d_condition > 0:
            cw, cb = pv("fc_condition.weight"), pv("fc_condition.bias")
        else:
            cw = cb = None
        if tokens is not None and self.token_kernel:
            key = (id(m._prep), f.data_ptr())
            if self._tok_table is None or self._tok_key != key:
                rows = []
                for i in range(m.num_layer):
                    W, p = m._prep["layers"][i], f"enc_layers.{i}."
                    rows.append(dict(Wqkv=W["Wqkv"], bqkv=W["bqkv"], Wo=W["Wo"], bo=pv(p + "rga.fc.bias"), W1=W["W1"], b1=pv(p + "FFN_pre.bias"),
                                     W2=W["W2"], b2=pv(p + "FFN_suf.bias"), ln1_g=pv(p + "layernorm1.weight"), ln1_b=pv(p + "layernorm1.bias"),
                                     ln2_g=pv(p + "layernorm2.weight"), ln2_b=pv(p + "layernorm2.bias"), E=W["E"], kcache=self.kc[i], vcache=self.vc[i]))
                self._tok_table, self._tok_key = ops.dec_token_table(rows, f.device), key
            ops.dec_token(tokens, cond if cw is not None else None, pv("embedding.weight"), cw, cb, m._pe, m.d_condition, self._tok_table,
                          m.num_layer, m._prep["head"]["Wf"], pv(m._HEAD_B), V, self.logits, self._tok_ws, ns, B, d, di, H, dh, M, M, t,
                          self._pos_dev, eps, getattr(self, "_tok_blocks", 0), dt)
            if self._pos_dev is None:
                self.t += 1
            return self.logits
        for r0 in range(0, B, ROWS):                          # row chunks (Mr <= 8 per kernel call)
            r1 = min(B, r0 + ROWS)
            Mr = r1 - r0
            part = self.part[r0 * H:r1 * H]
            fused0 = False
            for i in range(m.num_layer):
                W = m._prep["layers"][i]
                p = f"enc_layers.{i}."
                if i == 0 and tokens is not None and self.fused and (m.d_condition <= 0 or (m.d_condition % 4 == 0 and m.d_condition < d)):
                    # first layer, fused like the others (round 4): embedding row -> q|k|v of a head -> cache append ->
                    # attention partials in ONE launch (me_dec_embed_qkv_attn): 26 launches per token at 6 layers
                    ops.dec_embed_qkv_attn(tokens[r0:r1], cond[r0:r1] if cw is not None else None, pv("embedding.weight"), cw, cb,
                                           m._pe, m.d_condition, W["Wqkv"], W["bqkv"], self.xres[r0:r1], self.kc[i][r0:r1],
                                           self.vc[i][r0:r1], W["E"], None, 0, part, ns, Mr, d, H, dh, M, M, t, self._pos_dev, dt)
                    fused0 = True
                elif i == 0 and tokens is not None:
                    ops.dec_embed_qkv(tokens[r0:r1], cond[r0:r1] if cw is not None else None, pv("embedding.weight"), cw, cb,
                                      m._pe, m.d_condition, W["Wqkv"], W["bqkv"], self.xres[r0:r1], self.q[r0:r1],
                                      self.kc[i][r0:r1], self.vc[i][r0:r1], Mr, d, H, dh, M, t, self._pos_dev, dt)
                elif i == 0:
                    ops.dec_qkv(None, None, None, eps, x_hi[r0:r1], x_lo[r0:r1] if x_lo is not None else None, W["Wqkv"],
                                W["bqkv"], self.xres[r0:r1], self.q[r0:r1], self.kc[i][r0:r1], self.vc[i][r0:r1], Mr, d, H, dh,
                                M, t, self._pos_dev, dt)
                elif self.fused:
                    # layers >= 1: LayerNorm2 of the previous layer -> q|k|v of a head -> cache append -> attention partials
                    # in ONE launch (me_dec_ln_qkv_attn): 27 instead of 32 launches per token at 6 layers
                    pp = f"enc_layers.{i - 1}."
                    ops.dec_ln_qkv_attn(self.s2[r0:r1], pv(pp + "layernorm2.weight"), pv(pp + "layernorm2.bias"), eps,
                                        W["Wqkv"], W["bqkv"], self.xres[r0:r1], self.kc[i][r0:r1], self.vc[i][r0:r1], W["E"],
                                        None, 0, part, ns, Mr, d, H, dh, M, M, t, self._pos_dev, dt)
                else:
                    pp = f"enc_layers.{i - 1}."
                    ops.dec_qkv(self.s2[r0:r1], pv(pp + "layernorm2.weight"), pv(pp + "layernorm2.bias"), eps, None, None,
                                W["Wqkv"], W["bqkv"], self.xres[r0:r1], self.q[r0:r1], self.kc[i][r0:r1], self.vc[i][r0:r1],
                                Mr, d, H, dh, M, t, self._pos_dev, dt)
                if (i == 0 and not fused0) or not self.fused:
                    ops.dec_attn(self.q[r0:r1], self.kc[i][r0:r1], self.vc[i][r0:r1], W["E"], None, 0, part, ns, Mr, H, dh, M,
                                 M, t, self._pos_dev, dt)
                ops.dec_proj_resid(part, ns, H, dh, None, W["Wo"], pv(p + "rga.fc.bias"), self.xres[r0:r1], self.s1[r0:r1],
                                   Mr, d, d, dt)
                ops.dec_ln_proj(self.s1[r0:r1], pv(p + "layernorm1.weight"), pv(p + "layernorm1.bias"), eps, W["W1"],
                                pv(p + "FFN_pre.bias"), self.o1res[r0:r1], self.hid[r0:r1], Mr, di, d, ops.ME_EPI_RELU, dt)
                ops.dec_proj_resid(None, 0, 0, 0, self.hid[r0:r1], W["W2"], pv(p + "FFN_suf.bias"), self.o1res[r0:r1],
                                   self.s2[r0:r1], Mr, d, di, dt)
            pl = f"enc_layers.{m.num_layer - 1}."
            ops.dec_ln_proj(self.s2[r0:r1], pv(pl + "layernorm2.weight"), pv(pl + "layernorm2.bias"), eps,
                            m._prep["head"]["Wf"], pv(m._HEAD_B), None, self.logits[r0:r1], Mr, V, d, ops.ME_EPI_OUT_F32, dt)
        if self._pos_dev is None:
            self.t += 1
        return self.logits

    # -------------------------------------------------------------- public steps
    @property
    def launches_per_token(self):
        if self.token_kernel:
            return 2                                                # the token + pick / commit
        return (4 if self.fused else 5) * self.m.num_layer + 2     # + head, + pick / commit

    def check_token_status(self):
        """me_dec_token's error word (a poll ran into its bound: results undefined).  Synchronises; call at a sync point."""
        if self.token_kernel and int(self._tok_ws[8:12].view(torch.int32).item()) != 0:
            raise RuntimeError("me_dec_token: a block gave up polling its inputs (error word set); decode results are undefined")

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
        both_lo = torch.empty_like(both) if self.xlo else None
        dummy = torch.zeros(self.B, 1, dtype=torch.int64, device=f.device)
        ops.embed_fwd(both, dummy, cond, m._pview(f, "embedding.weight"), cw0, cb0, cw1, cb1, m._pe,
                      ops.ME_COND_TOKEN, self.B, 0, d, 0, 0.0, 0, out_lo=both_lo)
        for s in range(2):
            self._layers_and_head(x_hi=both[:, s].contiguous(), x_lo=both_lo[:, s].contiguous() if both_lo is not None else None)

    def step(self, tokens, cond=None):
        """Feed one token per sequence (int64 [B]) at the next position; returns logits f32 [B, V]
        (a view of an internal buffer, valid until the next step)."""
        m = self.m
        if self.t >= m.max_seq:
            raise RuntimeError("decode position %d exceeds max_seq %d" % (self.t, m.max_seq))
        f = m.flat_params
        tokens = tokens.to(device=f.device, dtype=torch.int64).reshape(self.B, 1).contiguous()
        return self._embed_and_run(tokens, cond)

    def _embed_and_run(self, tokens, cond):
        m = self.m
        if m.d_condition > 0:
            cond = cond.to(device=m.flat_params.device, dtype=torch.float32).contiguous()
        return self._layers_and_head(tokens=tokens, cond=cond)

    # -------------------------------------------------------------- device-resident greedy loop
    def greedy_run(self, tokens, n_steps, cond=None, special=None, use_graph=True, step_events=None):
        """Feed `tokens` (int64 [B]) at the next position, then keep feeding the arg-max token back for n_steps
        steps in total, entirely on the device (generate.py:99-189 with top_k = 1): the position lives in device
        memory (me_embed_fwd pos_dev, the t_dev argument of me_dec_qkv / me_dec_attn), me_greedy_pick writes the next input token and
        me_decode_commit appends it to a history buffer and advances the position.  The step is captured once as a
        HIP graph and replayed, so a token costs one graph launch instead of ~35 kernel launches from Python.
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
                self._embed_and_run(self._tok, self._cond if m.d_condition > 0 else None)
                ops.greedy_pick_commit(self.logits, m.vocab_size, self._special, self._tok, self._hist, self._pos, self.B)
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
            if step_events is not None:                     # per-step device latency (bench.py): one timing event per step
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                step_events.append(ev)
        self.t = t0 + n_steps
        self.check_token_status()                           # (opt-in token kernel only: reads its error word, synchronises)
        return self._hist[:, t0:t0 + n_steps]

    def sample_run(self, tokens, n_steps, cond, special, is_timeshift, repeat_counts, temp_note, temp_rest,
                   penalty_coeff, top_k, top_p, uniforms, use_graph=True):
        """The sampling loop of generate() (generate.py:99-189) on the device: feed `tokens` (int64 [B]) at the next
        position, sample the next token with me_sample_step (NaN / specials mask, log-softmax, the per-row temperature
        with its repeat penalty, top-k, nucleus cut, inverse-CDF draw at uniforms[step], repeat-counter update), append
        it and feed it back, n_steps times.  One HIP graph per token (33 launches) instead of ~40 eager launches and a
        dozen small torch ops from Python.  `uniforms` float32 [n_steps, B] are drawn by the caller (so the random
        stream is the caller's generator's); `repeat_counts` float32 [B] is updated in place.
        Returns the sampled ids int64 [B, n_steps] (on the device)."""
        m = self.m
        dev = m.flat_params.device
        n_steps = int(n_steps)
        if m.vocab_size > 4096:
            raise RuntimeError("sample_run: the sampling kernel sorts at most 4096 logits (vocab %d)" % m.vocab_size)
        if self.t + n_steps > m.max_seq:
            raise RuntimeError("decode positions %d..%d exceed max_seq %d" % (self.t, self.t + n_steps, m.max_seq))
        if uniforms.shape[0] < n_steps or uniforms.shape[1] != self.B or uniforms.dtype != torch.float32:
            raise ValueError("uniforms must be float32 [>= n_steps, B]")
        if not hasattr(self, "_tok"):
            self._tok = torch.zeros(self.B, 1, dtype=torch.int64, device=dev)
            self._hist = torch.zeros(self.B, m.max_seq, dtype=torch.int64, device=dev)
            self._pos = torch.zeros(1, dtype=torch.int32, device=dev)
            self._cond = torch.zeros(self.B, 2, dtype=torch.float32, device=dev)
            self._special = None
        if not hasattr(self, "_s_u"):
            self._s_u = torch.zeros(m.max_seq, self.B, dtype=torch.float32, device=dev)
            self._s_rc = torch.zeros(self.B, dtype=torch.float32, device=dev)
            self._s_ts = torch.zeros(m.vocab_size, dtype=torch.uint8, device=dev)
            self._s_graph, self._s_key = None, None
        self._tok.copy_(tokens.to(device=dev, dtype=torch.int64).reshape(self.B, 1))
        self._pos.fill_(self.t)
        if m.d_condition > 0:
            self._cond.copy_(cond.to(device=dev, dtype=torch.float32))
        self._s_u[:n_steps].copy_(uniforms[:n_steps])
        self._s_rc.copy_(repeat_counts.to(device=dev, dtype=torch.float32))
        self._s_ts.copy_(is_timeshift.to(device=dev, dtype=torch.uint8))
        sp = None if special is None else special.to(device=dev, dtype=torch.int32).contiguous()
        # everything that is baked into the captured step: scalars, the special-id list and the position origin
        key = (float(temp_note), float(temp_rest), float(penalty_coeff), int(top_k), float(top_p), self.t,
               None if sp is None else tuple(sp.tolist()))
        if key != self._s_key:
            self._s_key, self._s_graph, self._s_special = key, None, sp
        t0 = self.t

        def one_step():
            self._pos_dev = self._pos
            try:
                self._embed_and_run(self._tok, self._cond if m.d_condition > 0 else None)
                # block b reads the fed token of row b first and writes the sampled one last: the input buffer is reused
                ops.sample_step(self.logits, m.vocab_size, self._s_special, self._tok, self._s_ts, self._s_rc, temp_note,
                                temp_rest, penalty_coeff, top_k, top_p, self._s_u, self._pos, t0, self._tok)
                ops.decode_commit(self._tok, self._hist, self._pos, self.B)         # history[pos] = next; ++pos
            finally:
                self._pos_dev = None

        done = 0
        if use_graph and self._s_graph is None and n_steps > 2:
            one_step()                                      # warm-up outside the capture
            done = 1
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step()
            self._s_graph = g
        for _ in range(n_steps - done):
            if use_graph and self._s_graph is not None:
                self._s_graph.replay()
            else:
                one_step()
        self.t = t0 + n_steps
        self.check_token_status()
        repeat_counts.copy_(self._s_rc.to(repeat_counts.device))
        return self._hist[:, t0:t0 + n_steps]


class WindowForward:
    """The sliding-window regime of generate() (generate.py:99-119 once the sequence is longer than --max_input_len): every
    new token shifts the absolute position of the whole window, the K/V cache is invalid and the reference -- and this
    build -- run a full forward over the window per token.  The window has a FIXED shape there, so the ~55 launches of the
    forward are captured once into a HIP graph and replayed per token (the eager loop is bound by the host: 0.89 ms per
    token at B = 4 x 1024 against the forward's GPU time).  Same kernels, same arguments, same results as `model(x, cond)`."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self._graph = None

    @torch.no_grad()
    def last_logits(self, tokens, cond=None):
        """tokens [B, L] (any integer dtype), cond [B, 2] or None -> f32 logits of the LAST position [B, V]."""
        m = self.model
        if m.training:
            raise RuntimeError("WindowForward is an inference path (model.eval())")
        tokens, cond_c, B, Ltok, Lm = m._check_inputs(tokens, cond)
        V = m.head_size
        m._refresh_weights()                                 # host-side version check; never part of the captured graph
        # The captured launches carry raw device addresses of the model's workspace of this shape and of its prepared
        # weights: both objects are part of the key AND referenced from here, so that model._workspace() dropping its
        # cache (more than 6 shapes) or a rebuilt _prep (dtype change) can neither free the memory under a replay nor go
        # unnoticed.
        ws_now = m._ws.get((B, Lm, False))
        key = (B, Ltok, Lm, m._flat.data_ptr(), m.compute_dtype, id(m._prep), id(ws_now))
        if key != self._key or ws_now is None:
            self._graph = None
            self._tok = tokens.clone()
            self._cond = cond_c.clone()
            self._out = torch.empty(B * Lm, V, dtype=torch.float32, device=tokens.device)
            # warm-up: the workspace exists afterwards
            self._ws_ref = m._forward_impl(self._tok, self._cond, B, Ltok, Lm, False, 0.0, 0, self._out)
            self._prep_ref = m._prep
            self._key = (B, Ltok, Lm, m._flat.data_ptr(), m.compute_dtype, id(m._prep), id(self._ws_ref))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                m._forward_impl(self._tok, self._cond, B, Ltok, Lm, False, 0.0, 0, self._out)
            self._graph = g
        self._tok.copy_(tokens)
        self._cond.copy_(cond_c)
        self._graph.replay()
        return self._out.view(B, Lm, V)[:, -1, :].clone()
