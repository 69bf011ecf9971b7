/* midiemo.h -- C-ABI of libmidiemo_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of serkansulun/midi-emotion: the
 * emotion-conditioned Music-Transformer block stack (train step + cached
 * decode).  The reference has no native layer: its "FFI" for this path is the
 * set of PyTorch op call sites inside src/models/music_multi.py,
 * src/models/music_continuous_token.py and src/train.py.  Every entry point
 * below names the reference call sites (file:line under /root/reference/src)
 * it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE addresses owned by the caller (PyTorch tensors);
 *     the library is stateless: it never allocates, frees or synchronises, keeps no
 *     mutable state between calls (see "Process-wide switches" below for the read-once
 *     settings) and never retains a pointer past the call.  Scratch memory
 *     is caller-owned: me_workspace_bytes() says how much an entry point needs;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work;
 *   - `dtype` selects the activation/weight storage type T of the call:
 *       ME_F32  : exact-f32 MFMA (v_mfma_f32_32x32x2_f32), parity tier
 *       ME_BF16 : bf16 storage, f32 accumulate (v_mfma_f32_32x32x16_bf16)
 *       ME_F16  : f16 storage, f32 accumulate (v_mfma_f32_32x32x16_f16; same rate and bytes as bf16, 10 mantissa bits) -- the
 *                 reference's own mixed precision (torch.cuda.amp.autocast + GradScaler, train.py:101,108,281,317-324;
 *                 generate.py:116); gradients need the loss scale (me_scaler_step, me_ce_bwd, me_adamw_step)
 *     master parameters, gradients, optimiser state, statistics are always f32;
 *   - return value: ME_OK or a negative ME_ERR_* code; nothing throws/aborts.
 *
 * Process-wide switches.  The only state the library keeps is read-only after first use: a per-device cache of the CU
 * count / "LDS limit raised" flags, and these environment variables, each read ONCE (getenv at first use) -- development
 * and measurement aids, never needed for correct results:
 *   MIDIEMO_CU_RESERVE=n   persistent GEMM grids use (#CUs - n) blocks (CUs left to a concurrent RCCL kernel; default 0)
 *   MIDIEMO_NO_NT256=1     16-bit NT GEMMs run the generic 128 x 128 kernel instead of the persistent 256 x 256 one
 *   MIDIEMO_NO_TN256=1     likewise for the weight-gradient (TN) GEMMs
 *   MIDIEMO_DEC_CW=1|2|4, MIDIEMO_DEC_KS=0|1   column / K-split geometry of the decode GEMV kernels
 *   MIDIEMO_DEBUG=1        print the HIP error string when a launch fails
 * (host side, midiemo/decode.py: MIDIEMO_DEC_TOKEN=1 runs a decode token as ONE launch, me_dec_token, instead of 4 per layer)
 */
#ifndef MIDIEMO_H
#define MIDIEMO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_ABI_VERSION 24
#define ME_DEC_PART_REC(dh) ((dh) + 4)      /* floats per attention partial of the decode step (me_dec_attn) */
#define ME_SUMSQ_WS_BYTES 1040   /* me_sumsq workspace: ticket counter + 256 block sums + padding */

enum { ME_F32 = 0, ME_BF16 = 1, ME_F16 = 2 };

enum {
    ME_OK = 0,
    ME_ERR_BAD_DTYPE = -1,
    ME_ERR_BAD_SHAPE = -2,   /* unsupported head dim, K not chunk-aligned, L > max_seq, ... */
    ME_ERR_ALIGNMENT = -3,   /* pointer / leading dimension not 16-byte aligned */
    ME_ERR_LAUNCH = -4,      /* hipGetLastError() after launch */
    ME_ERR_NULL = -5,
    ME_ERR_WORKSPACE = -6    /* caller workspace too small / misaligned (see me_workspace_bytes) */
};

/* conditioning modes of the embedding prologue */
enum { ME_COND_NONE = 0, ME_COND_CONCAT = 1, ME_COND_TOKEN = 2 };

/* epilogue flags of me_gemm_nt */
enum {
    ME_EPI_RELU = 1,        /* y = max(y, 0) after bias             (F.relu, music_multi.py:131) */
    ME_EPI_OUT_F32 = 2,     /* C is float regardless of dtype       (logits for the CE head)     */
    ME_EPI_RELU_BWD = 4     /* y = (gate > 0) ? y : 0 ; gate is T   (autograd of F.relu)         */
};

#define ME_LO8 0x100        /* OR-ed into `dtype` of me_resid_ln_fwd / me_embed_fwd: 8-bit low halves of the residual stream */
int me_abi_version(void);

/* ---- parameter preparation ------------------------------------------------
 * f32 master [rows][cols] -> T copy (dst, ld_dst) and/or T transposed copy
 * (dstT [cols][rows], ld_dstT).  Either destination may be NULL.
 * Replaces: the implicit weight casts of torch.cuda.amp.autocast (train.py:281). */
int me_cast_transpose(const float* src, int rows, int cols, void* dst, int ld_dst,
                      void* dstT, int ld_dstT, int dtype, void* stream);

/* The same for many tensors in ONE launch.  desc_dev: device array of n_tensors descriptors, tile_begin = number
 * of 32x32 tiles of all earlier tensors (ascending), total_tiles = sum over tensors of ceil(rows/32)*ceil(cols/32);
 * dst / dstT as above (either may be NULL). */
typedef struct me_ct_desc {
    const float* src;
    void* dst;
    void* dstT;
    int32_t rows, cols, ld_dst, ld_dstT, tile_begin;
    int32_t mode;   /* 0: dstT = transposed copy; 1 (ME_CT_PACK_REL): src is a relative table E [M][dh] and dstT receives
                     * its packed fragment images (the layout of me_rga_pack_rel; ld_dstT unused) */
} me_ct_desc;
#define ME_CT_PACK_REL 1
int me_cast_transpose_multi(const me_ct_desc* desc_dev, int n_tensors, int total_tiles, int dtype, void* stream);

/* ---- embedding prologue ---------------------------------------------------
 * out[B, Lm, d] (T).  Lm = Ltok (+2 for ME_COND_TOKEN).
 *   NONE  : out = emb[tok]*sqrt(d) + PE                       (music_multi.py:91-92,101)
 *   CONCAT: out = cat(emb[tok]*sqrt(d-dc), Wc.cond+bc) + PE   (music_multi.py:94-101)
 *   TOKEN : out = cat_seq([W0*v+b0, W1*a+b1], emb[tok]*sqrt(d)) + PE
 *                                                  (music_continuous_token.py:80-100)
 * followed by inverted dropout(p) with the counter-based mask (seed, site 0)
 * (music_multi.py:102).  emb is the f32 master table [V][d-dc]; pe is f32 [>=Lm][d].
 * pos_dev (may be NULL): device int32, added to every row's position for the PE lookup -- the decode step reads
 * its position from device memory so that the whole step is replayable as a HIP graph.
 * out_lo (T, may be NULL): receives the low-order part v - float(T(v)) of every element: the residual stream of the
 * bf16 tier is carried as hi + lo (see me_resid_ln_fwd). */
int me_embed_fwd(void* out, void* out_lo, int dtype, const int64_t* tokens, const float* cond,
                 const float* emb, const float* cw0, const float* cb0,
                 const float* cw1, const float* cb1, const float* pe, const int32_t* pos_dev,
                 int mode, int B, int Ltok, int d_model, int d_cond,
                 float p_drop, uint64_t seed, void* stream);

/* Gradient of the prologue: accumulates (+=) into the f32 gradient tensors.
 * Rows of g_emb for token == pad_token receive nothing (padding_idx,
 * music_multi.py:57-59).  vocab = number of rows of the table: one block per row collects the positions holding
 * that token and sums their rows (no atomics on the table); 0 selects plain global atomics.
 * ws (may be NULL): me_workspace_bytes(ME_WS_EMBED_BWD, 0, 0, 0, dtype) = 1024 bytes, 16-byte aligned, ZERO before the
 * first call; the library leaves it zeroed, so one buffer serves every later call on the same stream.  With it, tokens
 * that occur more than 192 times in the batch (real MIDI streams: time shifts, frequent notes) are cut into work items of
 * ~128 occurrences and spread over the chip by a second launch instead of being summed by a single block; results are the
 * same up to f32 summation order. */
int me_embed_bwd(const void* dout, int dtype, const int64_t* tokens, const float* cond,
                 float* g_emb, float* g_cw0, float* g_cb0, float* g_cw1, float* g_cb1,
                 int mode, int B, int Ltok, int d_model, int d_cond, int vocab, int pad_token,
                 float p_drop, uint64_t seed, void* ws, size_t ws_bytes, void* stream);

/* ---- key padding mask -----------------------------------------------------
 * key_pad[B, Lm] (uint8, 1 = masked key) from tokens == pad_token; the `shift`
 * leading slots (2 for ME_COND_TOKEN) are never pad.
 * Replaces generate_mask's pad part (music_multi.py:25-38); the causal part is a
 * predicate inside the attention kernels and is never materialised. */
int me_key_pad_mask(uint8_t* key_pad, const int64_t* tokens, int B, int Ltok, int shift,
                    int pad_token, void* stream);

/* ---- GEMM  C[M,N] = A[M,K] . B[N,K]^T  (+bias[N]) (+add[M,N]) ------------------
 * A, B are T with the contraction dimension contiguous (lda, ldb in elements,
 * multiples of 16 bytes).  C is T, or float with ME_EPI_OUT_F32.  bias is f32 or
 * NULL; add (T, ld = ldadd) is added after bias or NULL; gate (T, ld = ldgate) is
 * the ReLU-backward gate or NULL.
 * Replaces nn.Linear forward (music_multi.py:106,131-132,196-209,237) and, with
 * pre-transposed weights, the dX = dY.W products of its autograd. */
int me_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
               const float* bias, const void* add, int ldadd, const void* gate, int ldgate,
               int M, int N, int K, int flags, int dtype, void* stream);

/* ---- FFN_pre forward / FFN_suf dgrad with the ReLU's sign pattern as a bit mask ------------------
 * dir 0:  C = relu(A . B^T + bias)  and  mask <- (C > 0)          (FFN_pre + F.relu, music_multi.py:129-131)
 * dir 1:  C = mask ? (A . B^T) : 0                                  (autograd of F.relu over FFN_suf's dX; bias ignored)
 * Same results, bit for bit, as me_gemm_nt with ME_EPI_RELU / with gate = the activations and ME_EPI_RELU_BWD; the
 * backward launch then reads 1 bit per element instead of the bf16 activations (134 MB per layer at the headline shape).
 * mask: caller-owned, me_workspace_bytes(ME_WS_RELU_MASK, M, N, K, dtype) bytes, 16-byte aligned, opaque (1 bit per element
 * in the order the kernel's write-out touches them: 1 KB per 128 rows x 64 columns, rows rounded up to 256); a size of
 * 0 means the shape / dtype is not served (16-bit types, N % 64 == 0, the shapes the 256-tile kernel takes) -- callers then
 * keep the gate operand; calling anyway returns ME_ERR_BAD_SHAPE.  C is T, ldc % 8 == 0, 16-byte aligned. */
int me_gemm_nt_relu_mask(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                         void* mask, int M, int N, int K, int dir, int dtype, void* stream);

/* ---- workspace sizes (caller supplies every workspace) -----------------
 * Bytes of scratch the entry point `op` needs for a call of the given shape and dtype (an upper bound;
 * 0 = none).  ME_WS_GEMM_TN: me_gemm_tn_acc with (M, N, K) = (T, N, K). */
enum {
    ME_WS_GEMM_TN = 1,   /* me_gemm_tn_acc partial tiles: (M, N, K) = (T, N, K) */
    ME_WS_RGA_PT = 2,    /* me_rga_fwd / me_rga_bwd PT: (M, N, K) = (B*H, Lp, causal) */
    ME_WS_RGA_DGT = 3,   /* me_rga_bwd dGT workspace:   (M, N, K) = (B*H, Lp, unused) */
    ME_WS_RGA_MT = 4,    /* me_rga_fwd / me_rga_bwd MT: (M, N, K) = (B*H, Lp, unused) */
    ME_WS_GEMM_TN_GROUP = 5, /* me_gemm_tn_acc_group: (M, N, K) = (T, sum over the items of ceil(N/256) * (K/256), unused) */
    ME_WS_EMBED_BWD = 6,     /* me_embed_bwd frequent-token list: 1024 bytes, zero before the first use */
    ME_WS_SUMSQ = 7,         /* me_sumsq ordered block sums: ME_SUMSQ_WS_BYTES, zero before the first use */
    ME_WS_RELU_MASK = 8,     /* me_gemm_nt_relu_mask sign mask: (M, N, K) = the product's; 0 = shape / dtype not served, use the gate */
    ME_WS_DEC_TOKEN = 9      /* me_dec_token exchange records: (M, N, K) = (Mr, d_inner, d); zero before the first use */
};
size_t me_workspace_bytes(int op, int M, int N, int K, int dtype);

/* ---- GEMM  dW[N,K] += A[T,N]^T . B[T,K] ---------------------------------------
 * A = dY (T, lda), B = X (T, ldb), dW f32 (lddw).  If dbias != NULL also
 * dbias[N] += column sums of A.  Replaces the weight/bias gradients of nn.Linear.
 * The token dimension is split over the CUs.  ws (16-byte aligned, ws_bytes >=
 * me_workspace_bytes(ME_WS_GEMM_TN, T, N, K, dtype)) receives the partial tiles and the partial bias column sums, which
 * are then summed in a FIXED order (bit-reproducible dW and dbias); it may be reused by the next call on the same stream.  ws = NULL: the partial tiles
 * are accumulated with f32 atomics instead (order-dependent rounding, slower); a workspace that is too small is an
 * error (ME_ERR_WORKSPACE), never a silent fallback. */
int me_gemm_tn_acc(const void* A, int lda, const void* B, int ldb, float* dW, int lddw,
                   float* dbias, int T, int N, int K, void* ws, size_t ws_bytes, int dtype, void* stream);

/* Several weight gradients over the SAME token dimension in one launch (the four nn.Linear weight gradients of an
 * EncoderLayer -- for the last layer also the vocabulary head's --, music_multi.py:126-135,196-237: autograd of FFN_suf, FFN_pre, rga.fc and the fused Wq|Wk|Wv).  Same
 * results as n_items calls of me_gemm_tn_acc up to the summation order (still fixed, bit-reproducible); the token
 * dimension is split #CUs / (tiles of ALL items) ways instead of once per product, which at the headline shapes cuts the
 * partial-tile traffic to a quarter and replaces eight launches by two.  `items` is a HOST array (read during the call
 * only).  16-bit types, every N readable up to a multiple of 256 columns (lda), K % 256 == 0, T >= 2048 run the grouped kernel
 * (workspace: me_workspace_bytes(ME_WS_GEMM_TN_GROUP, T, total tiles, 0, dtype)); anything else is executed as
 * separate me_gemm_tn_acc calls with the same workspace. */
#define ME_TN_MAX_GROUP 5
typedef struct me_tn_item {
    const void* A; int lda;      /* dY [T, N] */
    const void* B; int ldb;      /* X  [T, K] */
    float* dW; int lddw;         /* [N, K] f32, accumulated into */
    float* dbias;                /* [N] f32 or NULL */
    int N, K;
} me_tn_item;
int me_gemm_tn_acc_group(const me_tn_item* items, int n_items, int T, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ---- relative global attention ---------------------------------------------
 * qkv  : T [B, L, 3, H, dh]  (token-major output of the fused QKV projection; the kernels read
 *        q / k / v tiles straight from it -- no head-major permute copies)
 * Epk  : the layer's relative table E (T [M, dh]) packed by me_rga_pack_rel (below)
 * key_pad : uint8 [B, L] or NULL
 * out  : T [B, L, H, dh]   lse : f32 [B, H, L]
 *   logits[l,j] = (q_l.k_j + q_l.E[M-1-(l-j)]) / sqrt(dh),  j<=l and key j not pad
 * Replaces music_multi.py:196-235 (head split/permute, einsum QE, _qe_masking, _skewing, QK^T,
 * mask, softmax, PV, head merge).  dh in {32, 48, 64}; M % 32 == 0; L <= M.  Epk = me_rga_pack_rel(E).
 * causal = 1: the language model (generate_mask: key <= q and not padded).  causal = 0: the bidirectional attention of
 * MusicRegression (models/music_regression.py:79, mask = None): all keys, relative term only for key <= q (the
 * reference's skewing leaves zeros above the diagonal); me_rga_bwd takes the same flag. */
int me_rga_fwd(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, void* PT, float* MT,
               int B, int L, int H, int dh, int M, int causal, int dtype, void* stream);
/* PT / MT (both NULL for inference, both given for training): the forward additionally leaves what the backward needs
 * instead of recomputing the softmax --
 *   PT : T, per (batch, head) the 32 x 32 tiles (32 queries x 32 keys, stored in the forward's register-image order:
 *        opaque to the caller) of the UNNORMALISED probabilities p = exp((s - m_t)) taken against the running row
 *        maximum m_t at that key tile: the packed lower triangle (key tile, query tile >= key tile) for causal = 1,
 *        the full square for causal = 0        (bytes: me_workspace_bytes(ME_WS_RGA_PT, B*H, Lp, causal));
 *   MT : f32 [B*H][Lp/32 key tiles][Lp] the running maxima m_t (raw logit units)   (ME_WS_RGA_MT).
 * P = p * exp(m_t / sqrt(dh) - lse).  Lp = L rounded up to 32.  No initialisation needed; kept until me_rga_bwd of the
 * same layer has run. */

/* Packs the relative table E (T [M][dh], music_multi.py:191 self.E) into the fragment images me_rga_fwd / me_rga_bwd
 * read: per block of 32 rows, dh/16 images of the rows themselves (operand of Q.E^T) followed by 2*ceil(dh/32) images
 * of the transposed block (operand of dQ += dG.E); every image is 64 lanes x 8 elements, i.e. one coalesced 1 KB load
 * per wave instruction instead of 32 cache lines.  Epk: T [(M/32) * (dh/16 + 2*ceil(dh/32)) * 512], 16-byte aligned.
 * Call it whenever E changes (the model does it inside its multi-tensor weight refresh, me_ct_desc.mode = 1). */
int me_rga_pack_rel(const void* E, void* Epk, int M, int dh, int dtype, void* stream);

/* Backward of me_rga_fwd.  dout: T [B,L,H,dh].  Writes dqkv (T, same layout as qkv), accumulates (+=) dE f32 [M, dh]
 * (natural layout).  PT / MT: what me_rga_fwd left for this layer (read only).  Workspaces (caller-owned, 16-byte
 * aligned, contents need NOT be initialised): delta f32 [B,H,L];
 *   dGT : T, per (batch, head) the tiles (query tile qt, step t <= qt) of the skewed dS (32 rows of the relative table
 *         x 32 queries, stored as the dE kernel's operand fragment images: opaque): what dE is contracted from
 *         (ME_WS_RGA_DGT).
 * Neither S nor dS is stored or recomputed from Q.K^T: dS = P o (V dO^T - delta) / sqrt(dh) with P from PT / MT.
 * causal: as in me_rga_fwd (autograd of music_multi.py:211-235 for 1, of music_regression.py's mask = None attention for 0). */
int me_rga_bwd(const void* qkv, const void* Epk, const void* out, const float* lse, const void* dout,
               void* dqkv, float* dE, float* delta_ws, const void* PT, const float* MT, void* dGT,
               int B, int L, int Lp, int H, int dh, int M, int causal, int dtype, void* stream);

/* The same backward, kernel by kernel.  phases: bit 0 = query-owned kernel (dQ part of dqkv, delta_ws, dGT), bit 1 =
 * key-owned kernel (dK, dV parts of dqkv; needs delta_ws), bit 2 = E-row-owned kernel (dE += ; needs dGT).  Within one
 * call the selected kernels run in the order 0, 1, 2.  The kernels of bits 1 and 2 do not depend on each other: a caller
 * that owns two streams may enqueue them side by side (its own events order them; the library never synchronises).
 * phases = 7 on one stream is me_rga_bwd. */
int me_rga_bwd_phases(const void* qkv, const void* Epk, const void* out, const float* lse, const void* dout,
                      void* dqkv, float* dE, float* delta_ws, const void* PT, const float* MT, void* dGT,
                      int B, int L, int Lp, int H, int dh, int M, int causal, int phases, int dtype, void* stream);

/* ---- residual + dropout + LayerNorm (post-LN, eps) ---------------------------
 *   s = x + dropout(a) ;  y = LN(s) * gamma + beta          (music_multi.py:128-129,133-134)
 * x, a, y, s_out are T [rows, d]; s_out (pre-norm sum, needed by backward) and
 * stats (f32 [rows][2] = mean, rstd) may be NULL for inference.
 * Residual stream precision (bf16 tier): under torch.autocast the reference keeps the residual stream and
 * LayerNorm in fp32 and only rounds the Linear inputs to bf16 (train.py:281).  x_lo / y_lo (T, may be NULL) carry
 * the low-order half of that stream: the residual input is x + x_lo, y is the bf16 operand of the next GEMM
 * (= bf16(LN output), exactly what autocast feeds the Linear) and y_lo = bf16(LN output - y).
 * dtype | ME_LO8 (16-bit tiers; also me_embed_fwd): x_lo / y_lo / out_lo are BYTE arrays [rows][d] -- the low half as
 * q = round((v - hi) / (ulp(hi) / 256)), |q| <= 127: hi + q ulp / 256 carries 15 (bf16) / 18 (f16) mantissa bits in 3 instead of
 * 4 bytes per element of the residual stream (round 6: the two 16-bit low halves were 0.22 ms of the 8.4 ms step). */
int me_resid_ln_fwd(const void* x, const void* x_lo, const void* a, const float* gamma, const float* beta,
                    void* y, void* y_lo, void* s_out, float* stats, int rows, int d, float eps,
                    float p_drop, uint64_t seed, uint32_t site, int dtype, void* stream);

/* Backward: given dy (T), s (T), stats, gamma:
 *   ds = LN'(dy) ; dx = ds (T) ; da = dropout'(ds) (T) ; dgamma,dbeta += (f32). */
int me_resid_ln_bwd(const void* dy, const void* s, const float* stats, const float* gamma,
                    void* dx, void* da, float* dgamma, float* dbeta, int rows, int d,
                    float p_drop, uint64_t seed, uint32_t site, int dtype, void* stream);

/* ---- cross-entropy head ------------------------------------------------------
 * logits [rows, ld] (V valid columns) in logits_dtype: ME_F32, or the 16-bit tier's own type (ME_BF16 / ME_F16 == dtype) -- the
 * 16-bit tiers' head GEMM writes T logits (what the reference's autocast F.linear produces; no 4-byte logits tensor exists
 * then); target int64 [rows].
 * The loss arithmetic is f32 either way.
 * me_ce_fwd:  row_lse[r] = logsumexp(logits[r, :V]);
 *             *loss_sum += sum over target != ignore of (row_lse - logit[target]);
 *             *n_valid  += count(target != ignore)          (both f32 device scalars)
 * me_ce_bwd:  dlogits (T [rows, ld_d]) = (exp(logit - row_lse) - onehot) * (target != ignore)
 *             * extra_scale * (loss_scale_dev ? *loss_scale_dev : 1) / *n_valid ; columns V..ld_d-1 are written as 0.
 *             loss_scale_dev (f32 device scalar, may be NULL): the dynamic loss scale of the f16 tier = state[ME_SCALER_SCALE]
 *             of me_scaler_step (GradScaler.scale(loss), train.py:317) -- read on the device, so a step never syncs.
 *             dbias (f32 [V] or NULL): dbias[j] += sum over rows of the f32 dlogits[:, j] BEFORE the rounding to T -- the
 *             vocabulary head's bias gradient (music_multi.py:71,106), which would otherwise be summed from the bf16
 *             dlogits by me_gemm_tn_acc (16-bit logits and dlogits, ld_d <= 2048 only; anything else: ME_ERR_BAD_SHAPE).
 * Replaces CrossEntropyLoss(ignore_index=pad) + its autograd (train.py:124,288-290). */
int me_ce_fwd(const void* logits, int ld, const int64_t* target, float* row_lse,
              float* loss_sum, float* n_valid, int rows, int V, int ignore_index, int logits_dtype, void* stream);
int me_ce_bwd(const void* logits, int ld, const int64_t* target, const float* row_lse,
              void* dlogits, int ld_d, const float* n_valid, float extra_scale, const float* loss_scale_dev, float* dbias,
              int rows, int V, int ignore_index, int logits_dtype, int dtype, void* stream);

/* ---- optimiser: global-norm clip + Adam(W) -----------------------------------
 * me_sumsq: *out += sum g[i]^2   (zero *out first; multiple calls accumulate).  ws (may be NULL): caller-owned scratch of
 *   me_workspace_bytes(ME_WS_SUMSQ, 0, 0, 0, dtype) = ME_SUMSQ_WS_BYTES bytes, 4-byte aligned, ZERO before the first use
 *   (a call that returns 0 leaves it ready for the next): the block sums are then added in a fixed order and the result
 *   is bit-reproducible -- which is what keeps the parameters of data-parallel ranks bit-identical after the clip
 *   (they hold identical reduced gradients; SURVEY 8e "identical optimizer state evolution on every rank").  NULL: the
 *   block sums are added with atomics in arrival order (last-bit differences from call to call).
 * me_adamw_step: coef = min(1, clip/(sqrt(*sumsq)+1e-6)) (clip <= 0: coef = 1);
 *   g' = g*coef*grad_scale ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ;
 *   p = p*(1 - lr*wd) - (lr/bias_corr1) * m / (sqrt(v)/sqrt(bias_corr2) + eps)
 * with bias_corr{1,2} = 1 - beta^step computed by the caller.  weight_decay = 0 is
 * exactly torch.optim.Adam.  If zero_grad != 0 the gradient is zeroed in the same pass.
 * scaler_state (may be NULL; f16 tier): the state me_scaler_step just updated -- the gradients (and *sumsq) carry the loss
 *   scale: g' additionally * state[ME_SCALER_INV]; if state[ME_SCALER_FOUND_INF] != 0 nothing is updated (the gradients are
 *   still zeroed): GradScaler.step skipping optimizer.step (train.py:322); bias_corr{1,2} are then recomputed on the device
 *   from state[ME_SCALER_STEP] (the count of steps actually taken), the by-value ones are ignored.
 * Replaces clip_grad_norm_ + optim.Adam.step + zero_grad (train.py:320-325).
 *
 * me_scaler_step: torch.cuda.amp.GradScaler (train.py:101,108,317-324: scale(loss), unscale_, step, update) on the device,
 *   one launch after me_sumsq and before me_adamw_step, no host sync.  state: f32 [ME_SCALER_WORDS], caller-owned:
 *     [ME_SCALER_SCALE]     the loss scale the NEXT backward multiplies into dlogits (initialise to 65536 = GradScaler's init_scale)
 *     [ME_SCALER_INV]       1 / (the scale the gradients just summed were produced with)      (written)
 *     [ME_SCALER_TRACKER]   consecutive finite steps since the last change of the scale      (initialise 0)
 *     [ME_SCALER_STEP]      optimiser steps actually taken                                     (initialise 0 or the resumed count)
 *     [ME_SCALER_FOUND_INF] 1 if *sumsq (squared norm of the SCALED gradients) is inf / nan    (written)
 *     [ME_SCALER_SKIPPED]   number of skipped steps so far                                     (initialise 0)
 *   found_inf: scale *= backoff_factor, tracker = 0, skipped += 1; else step += 1, tracker += 1 and, at growth_interval,
 *   scale *= growth_factor (never to inf), tracker = 0 -- GradScaler.update() with its defaults 2.0 / 0.5 / 2000. */
enum { ME_SCALER_SCALE = 0, ME_SCALER_INV = 1, ME_SCALER_TRACKER = 2, ME_SCALER_STEP = 3, ME_SCALER_FOUND_INF = 4,
       ME_SCALER_SKIPPED = 5, ME_SCALER_WORDS = 8 };
int me_scaler_step(float* state, const float* sumsq, float growth_factor, float backoff_factor, int growth_interval, void* stream);
int me_sumsq(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream);
int me_adamw_step(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq,
                  float clip, float grad_scale, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float bias_corr1, float bias_corr2, int zero_grad,
                  const float* scaler_state, void* stream);

/* ---- KV-cached decode step (generate.py:92-122 with the model call made incremental) --------------------
 * One new position `t` for each of Mr <= 8 sequences; the reference recomputes the whole window for every token
 * (generate.py:99-119).  Five kernels per layer (me_dec_qkv, me_dec_attn, me_dec_proj_resid, me_dec_ln_proj,
 * me_dec_proj_resid) + me_dec_ln_proj for the head; every position-dependent quantity can come from device memory
 * (t_dev, may be NULL) so that the whole step is replayable as one HIP graph.  The residual stream (pre-norm sums,
 * LayerNorm outputs) is f32 [Mr][d]; projection operands are rounded to T exactly where the training forward rounds
 * them; q, k, v and the caches are T.  kcache / vcache: T [Mr][H][Mc][dh] (position-major per head).
 *
 * me_dec_qkv: x = LayerNorm(s_in; gamma, beta, eps) (s_in f32 [Mr][d] = the previous layer's second pre-norm sum,
 *   music_multi.py:133-134), or x = x_hi + x_lo (T [Mr][d], x_lo may be NULL) when s_in == NULL (first layer: the
 *   embedding).  x is written to x_out (f32, residual of the attention block); q | k | v = T(x).Wqkv^T + bqkv
 *   (music_multi.py:196-209); q -> q_out T [Mr][d], k / v -> the caches at position t (t_dev overrides t). */
int me_dec_qkv(const float* s_in, const float* gamma, const float* beta, float eps, const void* x_hi, const void* x_lo,
               const void* Wqkv, const float* bqkv, float* x_out, void* q_out, void* kcache, void* vcache,
               int Mr, int d, int H, int dh, int Mc, int t, const int32_t* t_dev, int dtype, void* stream);

/* me_dec_embed_qkv: me_dec_qkv for the first layer with the embedding prologue of me_embed_fwd folded in (one
 *   position, modes NONE / CONCAT; for ME_COND_TOKEN the two condition slots go through me_dec_qkv(x_hi, x_lo)):
 *   x = emb[token] * sqrt(d - d_cond) | (cw . cond + cb) + pe[t]   (music_multi.py:89-101), f32, no dropout.
 *   tokens int64 [Mr]; cond f32 [Mr][2]; emb f32 [V][d - d_cond]; cw f32 [d_cond][2], cb f32 [d_cond] (d_cond <= 0:
 *   unused); pe f32 [>= t + 1][d]. */
int me_dec_embed_qkv(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb,
                     const float* pe, int d_cond, const void* Wqkv, const float* bqkv, float* x_out, void* q_out,
                     void* kcache, void* vcache, int Mr, int d, int H, int dh, int Mc, int t, const int32_t* t_dev,
                     int dtype, void* stream);

/* me_dec_attn: for every (sequence, head) and each of nsplit key ranges of [0, t]:
 *   s_j = q.(K[j] + E[M-1-(t-j)]) / sqrt(dh)  (pad keys masked; music_multi.py:211-231 for a single query row),
 *   part[seq*H + head][split] = (max_j s_j, sum_j exp(s_j - max), 0, 0, sum_j exp(s_j - max) V[j])  -- f32 [ME_DEC_PART_REC(dh)
 *   = dh + 4]: the P.V part starts 16-byte aligned (round 5: the combine reads it with 16-byte loads).
 * E: T [M][dh] natural layout.  nsplit <= 8.  grid = Mr*H x nsplit blocks; one key per 8-lane group, 16-byte coalesced K / V / E
 * reads.  The splits are combined by the prologue of me_dec_proj_resid. */
int me_dec_attn(const void* q, const void* kcache, const void* vcache, const void* E, const uint8_t* key_pad, int ld_pad,
                float* part, int nsplit, int Mr, int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype,
                void* stream);

/* Fused decode stage: LayerNorm(s_in row) -> q | k | v projection of one head -> cache append at position t -> key-split
 * attention partials, one block per (row, head, split): me_dec_qkv(s_in, ...) + me_dec_attn in ONE launch (a head's
 * attention needs only that head's q and the keys 0..t-1 already cached, so the two launches had no real seam; splits
 * 0..nsplit-2 share the cached keys, partial nsplit-1 is the new key t, computed -- with k_t / v_t and their cache
 * append -- by two extra blocks per (row, head)).  part / nsplit as me_dec_attn (read by me_dec_proj_resid); x_out (f32
 * [Mr, d] or NULL) receives the LayerNorm rows.  d <= 1024, 2 <= nsplit <= 8.  s_in: f32 [Mr, d].
 * Replaces, per layer >= 1 of a cached decode step, music_multi.py:133-134 (layernorm2 of the previous layer) and
 * :196-232 for the one new position (generate.py:116-119). */
int me_dec_ln_qkv_attn(const float* s_in, const float* gamma, const float* beta, float eps, const void* Wqkv, const float* bqkv,
                       float* x_out, void* kcache, void* vcache, const void* E, const uint8_t* key_pad, int ld_pad, float* part,
                       int nsplit, int Mr, int d, int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype,
                       void* stream);

/* The same stage for the FIRST layer: the block's input row is the embedding of the fed token (arguments and arithmetic
 * of me_dec_embed_qkv: token row * sqrt(d - d_cond) | condition projection, + the sinusoid row of position t, f32) instead
 * of a LayerNorm; me_dec_embed_qkv + me_dec_attn in ONE launch.  x_out receives the f32 embedding rows (the residual input
 * of the layer).  Replaces music_multi.py:89-101 + :196-232 for the one new position (generate.py:116-119). */
int me_dec_embed_qkv_attn(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb,
                          const float* pe, int d_cond, const void* Wqkv, const float* bqkv, float* x_out, void* kcache,
                          void* vcache, const void* E, const uint8_t* key_pad, int ld_pad, float* part, int nsplit, int Mr, int d,
                          int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype, void* stream);

/* me_dec_proj_resid: out f32 [Mr][N] = resid f32 [Mr][N] + bias + T(x).W^T with
 *   x = softmax-combine of the attention partials (part != NULL; K = H*dh; replaces the head merge + self.fc of
 *       music_multi.py:233-237 and the residual add of :128), or
 *   x = x_T (T [Mr][ldx]; part == NULL: FFN_suf + residual, music_multi.py:132-133). */
int me_dec_proj_resid(const float* part, int nsplit, int H, int dh, const void* x_T, int ldx, const void* W, int ldw,
                      const float* bias, const float* resid, float* out, int Mr, int N, int K, int dtype, void* stream);

/* me_dec_ln_proj: x = LayerNorm(s_in; gamma, beta, eps) -> x_out (f32, may be NULL); y = T(x).W^T + bias,
 *   flags & ME_EPI_RELU: ReLU (FFN_pre, music_multi.py:129-131); flags & ME_EPI_OUT_F32: y is f32 [Mr][ldy] (the
 *   vocabulary head, music_multi.py:106), else T [Mr][ldy]. */
int me_dec_ln_proj(const float* s_in, const float* gamma, const float* beta, float eps, const void* W, int ldw,
                   const float* bias, float* x_out, void* y, int ldy, int Mr, int N, int K, int flags, int dtype,
                   void* stream);

/* ---- One decode token as ONE persistent launch (round 6) ---------------------------------------------------------
 * me_dec_token runs what the launch chain {me_dec_embed_qkv_attn | me_dec_ln_qkv_attn, me_dec_proj_resid, me_dec_ln_proj,
 * me_dec_proj_resid} x n_layer + me_dec_ln_proj (head) runs -- the model call of generate.py:116-119 for one new position --
 * inside a single kernel of one block per CU.  The stages keep their all-to-all seams, but a seam is no longer a kernel
 * boundary + a cold weight round trip (1.7 + ~2.5 us): every value that crosses a seam travels as an 8-byte record
 * {payload, tag} written and polled with agent-scope atomic stores / loads (no counter, no fence, no L2 write-back: the
 * record validates itself), ~2.2 us per exchange between 256 blocks (tools/ubench_ll_exchange.hip), and each block requests
 * its stage's weight rows and cache rows BEFORE it polls.  Arithmetic, rounding points and summation orders are those of the
 * per-stage launches (csrc/me_decode_common.h); tests compare the two paths bit for bit.
 *
 * layers: DEVICE array of n_layer <= ME_DEC_MAX_LAYERS entries (the kernel walks it; every pointer inside 16-byte aligned,
 *     none NULL -- the library cannot check device memory).
 * ws: caller-owned, me_workspace_bytes(ME_WS_DEC_TOKEN, Mr, d_inner, d, dtype) bytes, 256-byte aligned, ZERO before the first
 *     use and private to one stream (it holds the exchange records, the launch epoch the tags are derived from and an
 *     error word: the uint32 at byte offset 8 is non-zero after a poll ran into its bound (~0.3 s) -- the results of that token
 *     and of every later one are undefined; a caller checks it when it next synchronises).
 * nsplit in {2, 4, 8}, dh % nsplit == 0, (dh / nsplit) even, Mr * H * nsplit <= blocks; Mr <= 4; d <= 1024; d % 8 == 0;
 * d_cond as me_dec_embed_qkv.  logits: f32 [Mr][ld_logits].  blocks: 0 = one per CU. */
#define ME_DEC_MAX_LAYERS 16
typedef struct me_dec_layer {
    const void* Wqkv;  const float* bqkv;      /* T [3d][d], f32 [3d]          (music_multi.py:196-209) */
    const void* Wo;    const float* bo;        /* T [d][d], f32 [d]            (:233-237) */
    const void* W1;    const float* b1;        /* T [d_inner][d], f32 [d_inner] (FFN_pre, :129-131) */
    const void* W2;    const float* b2;        /* T [d][d_inner], f32 [d]      (FFN_suf, :132-133) */
    const float* ln1_g; const float* ln1_b;    /* f32 [d]                      (:128) */
    const float* ln2_g; const float* ln2_b;    /* f32 [d]                      (:133-134) */
    const void* E;                             /* T [M][dh] natural layout     (:185, 240-243) */
    void* kcache;      void* vcache;           /* T [Mr][H][Mc][dh] */
} me_dec_layer;
int me_dec_token(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb, const float* pe,
                 int d_cond, const me_dec_layer* layers, int n_layer, const void* Wf, int ldwf, const float* bf, int V,
                 float* logits, int ld_logits, void* ws, size_t ws_bytes, int nsplit, int Mr, int d, int d_inner, int H, int dh,
                 int M, int Mc, int t, const int32_t* t_dev, float eps, int blocks, int dtype, void* stream);
/* Largest number of co-resident blocks the device gives me_dec_token for this dtype / dh (0: shape not served). */
int me_dec_token_blocks(int dh, int d, int d_inner, int dtype);

/* Greedy pick for generate(top_k=1): logits f32 [B, ld]; NaN -> 0, ids in
 * special[0..n_special) -> -inf, argmax -> out_ids[B] (generate.py:122-136,166-183). */
int me_greedy_pick(const float* logits, int ld, int V, const int32_t* special, int n_special,
                   int64_t* out_ids, int B, void* stream);

/* Sampling tail of generate() for one step (generate.py:122-189), one launch for the whole batch, V <= 4096
 * (one block per row sorts NP = 1024 / 2048 / 4096 (value, id) pairs in LDS, NP = the smallest of the three >= V):
 * NaN -> 0, ids in special[] -> -inf, log_softmax, / temp[b], keep the top_k largest (all if top_k <= 0), nucleus
 * cut at top_p (0 < top_p < 1; the first entry always stays), renormalise, draw by inverse CDF from the caller's
 * uniform u[b] in [0,1) -> out_ids[b]; n_choices[b] (may be NULL) = number of entries with probability > 0
 * (drives the repeat-penalty counter, generate.py:186-189).  dbg_p / dbg_i (may be NULL, f32 / int32 [B][NP]):
 * the final sorted probabilities and their vocabulary ids.  The reference draws with torch.multinomial, whose
 * random stream cannot be reproduced; the distribution is identical (tested), the draw is inverse-CDF. */
int me_sample_topk_topp(const float* logits, int ld, int V, const int32_t* special, int n_special,
                        const float* temp, int top_k, float top_p, const float* u, int64_t* out_ids,
                        int32_t* n_choices, float* dbg_p, int32_t* dbg_i, int B, void* stream);

/* One sampled decode step for a device-resident generation loop (HIP-graph replay): me_sample_topk_topp with the
 * per-row temperature of generate.py:138-163 computed on the device from the token that was just fed (prev_tok,
 * is_timeshift [V] bytes, repeat_counts [B] f32 in / out: updated as generate.py:186-189 does), and the uniform taken
 * from row (*pos - pos0) of u_table [steps, u_ld] (drawn in advance by the caller, so the stream of random numbers is
 * the caller's generator's).  The f32 operations of the temperature are the ones of the torch expression, unfused. */
int me_sample_step(const float* logits, int ld, int V, const int32_t* special, int n_special, const int64_t* prev_tok,
                   const uint8_t* is_timeshift, float* repeat_counts, float temp_note, float temp_rest, float penalty_coeff,
                   int top_k, float top_p, const float* u_table, int u_ld, const int32_t* pos, int pos0, int64_t* out_ids,
                   int32_t* n_choices, int B, void* stream);

/* Device-side decode bookkeeping: history[b][*pos] = tok[b] (int64 [B][ld_hist]); *pos += 1.
 * With me_embed_fwd(pos_dev) and the t_dev arguments of me_dec_* a greedy decode step has no host-side state
 * (the token loop of generate.py:99-189 for top_k = 1) and can be captured once and replayed. */
int me_decode_commit(const int64_t* tok, int64_t* history, int ld_hist, int32_t* pos, int B, void* stream);

/* me_greedy_pick followed by me_decode_commit of the picked ids in one launch (one block, a wave per sequence):
 * out_ids[b] = pick ; history[b][*pos] = pick ; *pos += 1. */
int me_greedy_pick_commit(const float* logits, int ld, int V, const int32_t* special, int n_special, int64_t* out_ids,
                          int64_t* history, int ld_hist, int32_t* pos, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIDIEMO_H */
