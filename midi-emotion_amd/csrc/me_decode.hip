// KV-cached decode step of the emotion-conditioned Music Transformer (gfx950): generate.py:92-122 with the model
// call made incremental (one new position per sequence, Mr <= 8 sequences per call).
//
// The step is HBM / latency bound: the layer weights (bf16: 6.3 MB per layer at the headline model) and the K / V
// cache of every (sequence, head) are streamed once; the arithmetic is negligible.  What decides the step time is
// the number of dependent kernels (a boundary costs 1.2-1.9 us on this part, a grid barrier 4-7 us: the guide's
// verdict is to cut at every all-to-all seam and replay the launches as one HIP graph) and whether every kernel
// spreads its stream over many CUs with 16-byte coalesced loads.  Five kernels per layer:
//
//   me_dec_qkv        LayerNorm2 of the previous layer (prologue, recomputed by every block: 4 x 512 values)
//                     -> q | k | v projection -> q to a buffer, k / v appended to the caches at position t
//   me_dec_attn       one block per (sequence, head, key split): scores q.(K[j] + E[M-1-(t-j)]) / sqrt(dh) with one
//                     key per 8-lane group (a 64-lane load instruction covers 8 complete 128-byte cache rows),
//                     block softmax over its key range, partial (max, sum, P.V) per split
//   me_dec_proj_resid combine of the split partials (prologue) -> Wo projection + bias + residual   (pre-norm sum s1)
//   me_dec_ln_proj    LayerNorm1 (prologue) -> FFN_pre + bias + ReLU
//   me_dec_proj_resid FFN_suf + bias + residual                                                     (pre-norm sum s2)
// and after the last layer me_dec_ln_proj again: LayerNorm2 -> vocabulary head (f32 logits).
//
// Precision: the residual stream, LayerNorm and the softmax stay in f32 (tiny tensors); the operands of the
// projections are rounded to the compute type T exactly where the training forward rounds them (LN output, attention
// output, FFN hidden), q / k / v and the caches are T.
#include "me_decode_common.h"

namespace {

enum { PRO_LN = 0, PRO_HILO = 1, PRO_ATTN = 2, PRO_PLAIN = 3, PRO_EMBED = 4 };
enum { EPI_QKV = 0, EPI_RESID = 1, EPI_T = 2, EPI_F32 = 3 };

struct DecArgs {
    // prologue
    const float* s_in;      // PRO_LN: pre-norm sum f32 [Mr][K]
    const float* gamma;
    const float* beta;
    float eps;
    const void* x_hi;       // PRO_HILO: T [Mr][K] (+ x_lo, may be NULL);  PRO_PLAIN: T [Mr][ldx]
    const void* x_lo;
    int ldx;
    const float* part;      // PRO_ATTN: f32 [Mr*H][nsplit][dh + 4] = (max, sum, 0, 0, o[dh]): ME_DEC_PART_REC(dh), o 16-byte aligned
    int nsplit, H, dh;
    const int64_t* tokens;  // PRO_EMBED: one token per row; cond f32 [Mr][2]; emb f32 [V][K - dc]; cw f32 [dc][2]; cb f32 [dc]; pe f32 [>= t + 1][K]
    const float* cond;
    const float* emb;
    const float* cw;
    const float* cb;
    const float* pe;
    int dc;
    float* x_out;           // f32 [Mr][K] (may be NULL): the prologue's unrounded result (residual of a later kernel)
    // projection
    const void* W;          // T [N][ldw]
    int ldw;
    const float* bias;      // f32 [N] or NULL
    int Mr, N, K;
    // epilogue
    int relu;
    void* y;                // EPI_T: T [Mr][ldy];  EPI_F32 / EPI_RESID: f32 [Mr][ldy];  EPI_QKV: q T [Mr][d]
    int ldy;
    const float* resid;     // EPI_RESID: f32 [Mr][N]
    void* kcache;           // EPI_QKV: T [Mr][H][Mc][dh]
    void* vcache;
    int Mc, t;
    const int32_t* t_dev;
};


// ---------------------------------------------------------------------------------------------------------
// y = epilogue( T(prologue(...)) . W^T + bias ).  Block = 4 waves, wave = CW output columns, lanes walk the contraction
// dimension in 16-byte chunks (K = 512 bf16: one chunk per lane and column), MR rows share every weight chunk.
// Latency is the whole cost of such a kernel, so the weight chunks (they depend on nothing) are requested FIRST, the
// prologue runs under their flight, and every prologue issues all of its own loads before it consumes any
// (measured with per-element load -> use loops: 34 us for the FFN_suf projection, 16 us for the combine prologue).
// Tried and dropped: warming the L2 with the NEXT projection's weights from inside each launch (one word per cache line,
// same block -> XCD residue): 0.169 ms per step against 0.159 without -- the weights are served by the Infinity Cache
// anyway and the extra loads only lengthen every kernel's tail.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int PRO, int EPI, int MR, int CW, bool KS>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const DecArgs a) {
    constexpr int CH = ET<T>::CH;
    constexpr int PF = KS ? 2 : 4;                                      // chunk positions per lane requested before the prologue
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [MR][K]: the projection's input rows (T-rounded values)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = a.K, Mr = a.Mr;
    const int nch_all = K / CH;
    const int kq = KS ? (nch_all + 3) / 4 : 0;                          // chunks per wave quarter
    const int ch_lo = KS ? wid * kq : 0;
    const int nch = KS ? max(0, min(kq, nch_all - ch_lo)) : nch_all;    // chunk positions of this wave: [ch_lo, ch_lo + nch)
    const T* W = reinterpret_cast<const T*>(a.W) + (size_t)ch_lo * CH;
    const float* xw = xs + ch_lo * CH;                                  // this wave's slice of every input row
    const int n0 = KS ? blockIdx.x * CW : (blockIdx.x * 4 + wid) * CW;

    // ---- the prologue's OWN loads go out first (round 5, tools/prof_decode_phases.py): loads return in order per wave, so a
    //      prologue whose few input loads queue behind 4-16 cold weight loads waits for the weights' round trip before it
    //      can start (measured: 2.0-2.8 us of a 3.9 us kernel sat in the prologue, the first FMAs after it took 0.45 us: the
    //      weights had long landed).  Input first, weights second: the prologue computes under the weights' flight.
    f32x4_t pre_ln[4];                                                   // PRO_LN: row m = wid
    chunk16 pre_h[4], pre_l[4];                                          // PRO_HILO / PRO_PLAIN: the thread's first four chunks
    float pre_ms[DEC_NSMAX], pre_ls[DEC_NSMAX];                          // PRO_ATTN: (max, sum) of (row, head) tid; o of the first two groups
    f32x4_t pre_o[2][DEC_NSMAX];
    if constexpr (PRO == PRO_LN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane * 4 + 256 * i;
            pre_ln[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (wid < MR && wid < Mr && k < K) pre_ln[i] = *reinterpret_cast<const f32x4_t*>(a.s_in + (size_t)wid * K + k);
        }
    } else if constexpr (PRO == PRO_HILO || PRO == PRO_PLAIN) {
        const T* hi = reinterpret_cast<const T*>(a.x_hi);
        const T* lo = reinterpret_cast<const T*>(a.x_lo);
        const int total = MR * nch_all;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(tid + 256 * u, total - 1), m = min(i / nch_all, Mr - 1), c = i % nch_all;
            pre_h[u] = ld_chunk(hi + (size_t)m * a.ldx + c * CH);
            if (PRO == PRO_HILO && lo) pre_l[u] = ld_chunk(lo + (size_t)m * a.ldx + c * CH);
        }
    } else if constexpr (PRO == PRO_ATTN) {
        const int dh = a.dh, ns = a.nsplit, rec = ME_DEC_PART_REC(dh), nmh = Mr * a.H;
        const float* p = a.part + (size_t)min(tid, nmh - 1) * ns * rec;
#pragma unroll
        for (int s_ = 0; s_ < DEC_NSMAX; ++s_) {
            const float2 msl = *reinterpret_cast<const float2*>(p + min(s_, ns - 1) * rec);       // (max, sum): one 8-byte load
            pre_ms[s_] = msl.x;
            pre_ls[s_] = msl.y;
        }
        const int ng = MR * K / 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int gi = min(tid + 256 * u, ng - 1), m = min(gi / (K / 4), Mr - 1), k = (gi % (K / 4)) * 4;
            const float* po = a.part + (size_t)(m * a.H + k / dh) * ns * rec + 4 + k % dh;
#pragma unroll
            for (int s_ = 0; s_ < DEC_NSMAX; ++s_) pre_o[u][s_] = *reinterpret_cast<const f32x4_t*>(po + min(s_, ns - 1) * rec);
        }
    }
    asm volatile("" ::: "memory");                                       // keep the issue order: input loads, then weights

    // ---- weight chunks of this wave's columns: in flight during the whole prologue
    chunk16 wp[PF][CW];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int ch = lane + 64 * u, chc = ch < nch ? ch : 0;
        if (64 * u < nch) {                                             // wave-uniform: no loads for positions past the row
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int n = min(n0 + c, a.N - 1);                     // clamped: results of columns >= N are never stored
                wp[u][c] = ld_w(W + (size_t)n * a.ldw + (size_t)chc * CH);
            }
        }
    }

    // ---- bias / residual of the output this lane will own (see the reduction below): requested now, not after the
    //      reduction (one dependent load less at the tail: 0.163 -> 0.159 ms per step)
    constexpr int NV = CW * MR;
    const int oidx = (lane & 15) + (NV / 4) * (lane >> 4);
    const bool owner = (lane & 15) < NV / 4 && n0 + oidx / MR < a.N && oidx % MR < Mr && (!KS || wid == 0);
    float bias_v = 0.f, resid_v = 0.f;
    if (owner) {
        if (a.bias) bias_v = a.bias[n0 + oidx / MR];
        if constexpr (EPI == EPI_RESID) resid_v = a.resid[(size_t)(oidx % MR) * a.N + n0 + oidx / MR];
    }

    // ---- prologue: input rows into LDS
    if constexpr (PRO == PRO_LN) {
        // one wave per row, the row stays in registers (K <= 1024): mean, centred variance, normalise
        constexpr int NV = 4;
        for (int m = wid; m < MR; m += 4) {
            f32x4_t v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = lane * 4 + 256 * i;
                v[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (m == wid) v[i] = pre_ln[i];                          // requested before the weights
                else if (m < Mr && k < K) v[i] = *reinterpret_cast<const f32x4_t*>(a.s_in + (size_t)m * K + k);
            }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            const float mean = wave_sum(sum) / K;
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (lane * 4 + 256 * i < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d_ = v[i][e] - mean; vs += d_ * d_; }
                }
            }
            const float rstd = rsqrtf(wave_sum(vs) / K + a.eps);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = lane * 4 + 256 * i;
                if (k < K) {
                    f32x4_t o = {0.f, 0.f, 0.f, 0.f}, r = o;
                    if (m < Mr) {
                        const f32x4_t g = *reinterpret_cast<const f32x4_t*>(a.gamma + k);
                        const f32x4_t be = *reinterpret_cast<const f32x4_t*>(a.beta + k);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (v[i][e] - mean) * rstd * g[e] + be[e]; r[e] = round_to<T>(o[e]); }
                        if (a.x_out && blockIdx.x == 0) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = o;
                    }
                    *reinterpret_cast<f32x4_t*>(&xs[m * K + k]) = r;
                }
            }
        }
    } else if constexpr (PRO == PRO_HILO || PRO == PRO_PLAIN) {
        // 16-byte chunks, four per thread in flight
        const T* hi = reinterpret_cast<const T*>(a.x_hi);
        const T* lo = reinterpret_cast<const T*>(a.x_lo);
        const int total = MR * nch_all;
        for (int i0 = tid; i0 < total; i0 += 256 * 4) {
            chunk16 h4[4], l4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 == tid) { h4[u] = pre_h[u]; l4[u] = pre_l[u]; continue; }      // requested before the weights
                const int i = min(i0 + 256 * u, total - 1), m = min(i / nch_all, Mr - 1), c = i % nch_all;
                h4[u] = ld_chunk(hi + (size_t)m * a.ldx + c * CH);
                if (PRO == PRO_HILO && lo) l4[u] = ld_chunk(lo + (size_t)m * a.ldx + c * CH);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                if (i < total) {
                    const int m = i / nch_all, c = i % nch_all;
                    float v[CH], r[CH];
                    chunk_to_f32<T>(h4[u], v);
                    if (PRO == PRO_HILO && lo) {
                        float l[CH];
                        chunk_to_f32<T>(l4[u], l);
#pragma unroll
                        for (int e = 0; e < CH; ++e) v[e] += l[e];
                    }
#pragma unroll
                    for (int e = 0; e < CH; ++e) { if (m >= Mr) v[e] = 0.f; r[e] = round_to<T>(v[e]); }
#pragma unroll
                    for (int q4 = 0; q4 < CH / 4; ++q4) {
                        *reinterpret_cast<f32x4_t*>(&xs[m * K + c * CH + 4 * q4]) = (f32x4_t){r[4 * q4], r[4 * q4 + 1], r[4 * q4 + 2], r[4 * q4 + 3]};
                        if (a.x_out && blockIdx.x == 0 && m < Mr)
                            *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + c * CH + 4 * q4) = (f32x4_t){v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
                    }
                }
            }
        }
    } else if constexpr (PRO == PRO_EMBED) {
        // embedding prologue of the first layer (music_multi.py:89-101 for one position): token row * sqrt(d - dc) |
        // condition projection, + sinusoid row of position t; f32 throughout (no hi / lo round trip through memory)
        const int dc = a.dc, de = K - dc;
        const int t = a.t_dev ? min(*a.t_dev, a.Mc - 1) : a.t;
        const float sq = sqrtf((float)de);
        for (int i = tid; i < MR * (K / 4); i += 256) {
            const int m = i / (K / 4), k = (i % (K / 4)) * 4;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f}, r = v;
            if (m < Mr) {
                const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(a.pe + (size_t)t * K + k);
                if (k < de) {
                    const f32x4_t e4 = *reinterpret_cast<const f32x4_t*>(a.emb + (size_t)a.tokens[m] * de + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = e4[e] * sq + p4[e];
                } else {
                    const float c0 = a.cond[m * 2], c1 = a.cond[m * 2 + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = a.cw[(k - de + e) * 2] * c0 + a.cw[(k - de + e) * 2 + 1] * c1 + a.cb[k - de + e] + p4[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = round_to<T>(v[e]);
                if (a.x_out && blockIdx.x == 0) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = v;
            }
            *reinterpret_cast<f32x4_t*>(&xs[m * K + k]) = r;
        }
    } else {    // PRO_ATTN: softmax-combine the key-split partials of every (row, head)
        const int dh = a.dh, ns = a.nsplit, rec = ME_DEC_PART_REC(dh), nmh = Mr * a.H;
        float* wn = xs + MR * K;                                        // [nmh][DEC_NSMAX] normalised split weights
        for (int mh = tid; mh < nmh; mh += 256) {                       // Mr * H may exceed the block (8 rows x > 32 heads)
            const float* p = a.part + (size_t)mh * ns * rec;
            float ms[DEC_NSMAX], ls[DEC_NSMAX];
#pragma unroll
            for (int s_ = 0; s_ < DEC_NSMAX; ++s_) {
                if (mh == tid) { ms[s_] = pre_ms[s_]; ls[s_] = pre_ls[s_]; continue; }   // requested before the weights
                const float2 msl = *reinterpret_cast<const float2*>(p + min(s_, ns - 1) * rec);
                ms[s_] = msl.x;
                ls[s_] = msl.y;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int s_ = 0; s_ < DEC_NSMAX; ++s_) if (s_ < ns) mx = fmaxf(mx, ms[s_]);
            const float msafe = mx == -INFINITY ? 0.f : mx;
            float l = 0.f, w[DEC_NSMAX];
#pragma unroll
            for (int s_ = 0; s_ < DEC_NSMAX; ++s_) {
                w[s_] = s_ < ns ? ET<T>::fexp(ms[s_] - msafe) : 0.f;    // exp(-inf) = 0 for empty / fully masked splits
                l = mul_add_unfused(w[s_], ls[s_], l);
            }
            const float inv = 1.f / l;                                  // every key masked: 0 * inf = NaN like the reference's softmax
#pragma unroll
            for (int s_ = 0; s_ < DEC_NSMAX; ++s_) wn[mh * DEC_NSMAX + s_] = w[s_] * inv;
        }
        __syncthreads();
        // 4 consecutive head-dim elements per thread and pass, two passes in flight: 2 x DEC_NSMAX independent 16-byte loads
        const int ng = MR * K / 4;
        for (int g0 = tid; g0 < ng; g0 += 512) {
            f32x4_t o[2][DEC_NSMAX];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (g0 == tid) {                                         // requested before the weights, together with (max, sum)
#pragma unroll
                    for (int s_ = 0; s_ < DEC_NSMAX; ++s_) o[u][s_] = pre_o[u][s_];
                    continue;
                }
                const int gi = min(g0 + 256 * u, ng - 1), m = min(gi / (K / 4), Mr - 1), k = (gi % (K / 4)) * 4;
                const float* p = a.part + (size_t)(m * a.H + k / dh) * ns * rec + 4 + k % dh;
#pragma unroll
                for (int s_ = 0; s_ < DEC_NSMAX; ++s_) o[u][s_] = *reinterpret_cast<const f32x4_t*>(p + min(s_, ns - 1) * rec);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gi = g0 + 256 * u;
                if (gi < ng) {
                    const int m = gi / (K / 4), k = (gi % (K / 4)) * 4;
                    f32x4_t v = {0.f, 0.f, 0.f, 0.f}, r = v;
                    if (m < Mr) {
                        const int mh = m * a.H + k / dh;
#pragma unroll
                        for (int s_ = 0; s_ < DEC_NSMAX; ++s_) {
                            const float w = s_ < ns ? wn[mh * DEC_NSMAX + s_] : 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = s_ < ns ? fmaf(w, o[u][s_][e], v[e]) : v[e];
                        }
                        if (a.x_out && blockIdx.x == 0) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = v;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = round_to<T>(v[e]);
                    *reinterpret_cast<f32x4_t*>(&xs[m * K + k]) = r;
                }
            }
        }
    }
    __syncthreads();

    // ---- projection
    if (n0 >= a.N) return;
    float acc[CW][MR];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int ch = lane + 64 * u;
        if (64 * u < nch) dec_fma_chunks<T, MR, CW>(acc, wp[u], ch < nch, xw, K, ch < nch ? ch : 0);
    }
    constexpr int U = 2;                                    // further chunk positions: two in flight
    for (int ch0 = lane + 64 * PF; ch0 < nch; ch0 += 64 * U) {
        chunk16 w[U][CW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + 64 * u, chc = ch < nch ? ch : ch0;
#pragma unroll
            for (int c = 0; c < CW; ++c) w[u][c] = ld_w(W + (size_t)min(n0 + c, a.N - 1) * a.ldw + (size_t)chc * CH);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + 64 * u;
            dec_fma_chunks<T, MR, CW>(acc, w[u], ch < nch, xw, K, ch < nch ? ch : ch0);
        }
    }
    // ---- reduce over the lanes: lane (row r = lane >> 4, i = lane & 15 < NV/4) ends up with output i + (NV/4) r
    float red[NV];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) red[c * MR + m] = acc[c][m];
    reduce_scatter64<NV>(red);
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) if ((lane & 15) == i) mine = red[i];
    if constexpr (KS) {                                     // the four K quarters meet in LDS (xs is dead: every wave passed the barrier)
        __syncthreads();
        float* ks = xs;                                     // [4 waves][NV]
        if ((lane & 15) < NV / 4) ks[wid * NV + (lane & 15) + (NV / 4) * (lane >> 4)] = mine;
        __syncthreads();
        if (wid != 0) return;
        if ((lane & 15) < NV / 4) {
            const int o_ = (lane & 15) + (NV / 4) * (lane >> 4);
            mine = (ks[o_] + ks[NV + o_]) + (ks[2 * NV + o_] + ks[3 * NV + o_]);
        }
    }
    if (!owner) return;
    const int c = oidx / MR, m = oidx % MR, n = n0 + c;
    float v = mine + bias_v;
    if (a.relu) v = fmaxf(v, 0.f);
    if constexpr (EPI == EPI_T) {
        reinterpret_cast<T*>(a.y)[(size_t)m * a.ldy + n] = ET<T>::from_f(v);
    } else if constexpr (EPI == EPI_F32) {
        reinterpret_cast<float*>(a.y)[(size_t)m * a.ldy + n] = v;
    } else if constexpr (EPI == EPI_RESID) {
        reinterpret_cast<float*>(a.y)[(size_t)m * a.ldy + n] = resid_v + v;
    } else {    // EPI_QKV: q to its buffer, k / v into the caches at position t
        const int d = a.N / 3, dh = a.dh, H = a.H;
        const int t = a.t_dev ? min(*a.t_dev, a.Mc - 1) : a.t;
        const int which = n / d, nn = n % d;
        if (which == 0) reinterpret_cast<T*>(a.y)[(size_t)m * a.ldy + nn] = ET<T>::from_f(v);
        else {
            T* cache = reinterpret_cast<T*>(which == 1 ? a.kcache : a.vcache);
            cache[(((size_t)m * H + nn / dh) * a.Mc + t) * dh + nn % dh] = ET<T>::from_f(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// attention of one new query per (sequence, head) against the cached keys 0..t, split over the keys:
// block (bh, split) -> part[bh][split] = (max, sum exp, sum exp * V) over its key range.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void dec_attn_kernel(const T* __restrict__ q, const T* __restrict__ kcache,
                                                       const T* __restrict__ vcache, const T* __restrict__ E,
                                                       const uint8_t* __restrict__ key_pad, int ld_pad, float* __restrict__ part,
                                                       int nsplit, int H, int M, int Mc, int t_host,
                                                       const int32_t* __restrict__ t_dev, float scale) {
    constexpr int CH = ET<T>::CH, CPR = DH / CH;            // 16-byte chunks per cache row
    constexpr int G = CPR <= 4 ? 4 : (CPR <= 8 ? 8 : 16);   // lanes per key (CPR = 6 / 12: the spare lanes idle)
    constexpr int KPW = 64 / G, KPI = 4 * KPW;              // keys per wave / per block and iteration
    constexpr int U = 4;                                    // iterations in flight
    __shared__ float qs[DH];
    __shared__ float ps[2048];
    __shared__ float red[8];
    __shared__ float osum[4][KPW][DH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bh = blockIdx.x, split = blockIdx.y, b = bh / H, head = bh % H;
    const int t = t_dev ? min(*t_dev, min(Mc, M) - 1) : t_host;
    const int n = t + 1;
    int per = (n + nsplit - 1) / nsplit;
    per = (per + KPI - 1) / KPI * KPI;
    const int j0 = split * per, j1 = min(n, j0 + per);
    float* pout = part + ((size_t)bh * nsplit + split) * ME_DEC_PART_REC(DH);
    if (j0 >= j1) {                                          // empty split
        if (tid < DH + 4) pout[tid] = tid == 0 ? -INFINITY : 0.f;
        return;
    }
    if (tid < DH) qs[tid] = ET<T>::to_f(q[(size_t)b * H * DH + head * DH + tid]) * scale;
    __syncthreads();
    const int kslot = lane / G, c = lane % G;
    const bool active = c < CPR;
    const int cc = active ? c : CPR - 1;
    float qv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) qv[i] = active ? qs[cc * CH + i] : 0.f;
    const T* kc = kcache + (size_t)bh * Mc * DH;
    const T* vc = vcache + (size_t)bh * Mc * DH;
    const T* er = E + (size_t)(M - 1 - t) * DH;              // relative row of key j: E[M-1-(t-j)] = er + j * DH
    const uint8_t* kp = key_pad ? key_pad + (size_t)b * ld_pad : nullptr;

    // ---- pass 1: scores of the range into LDS, running maximum.  The V rows of the first round (the only round up to
    //      KPI * U = 128 keys per split at dh = 64 bf16) are requested together with K and E: one memory latency less.
    float mx = -INFINITY;
    chunk16 v0[U];
    for (int jb = j0 + wid * KPW; jb < j1; jb += KPI * U) {
        chunk16 kk[U], ee[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jc = min(jb + u * KPI + kslot, j1 - 1);
            kk[u] = ld_kv(kc + (size_t)jc * DH + cc * CH);
            ee[u] = ld_kv(er + (size_t)jc * DH + cc * CH);
        }
        if (jb == j0 + wid * KPW) {
#pragma unroll
            for (int u = 0; u < U; ++u) v0[u] = ld_kv(vc + (size_t)min(jb + u * KPI + kslot, j1 - 1) * DH + cc * CH);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KPI + kslot;
            const T* ke = reinterpret_cast<const T*>(&kk[u]);
            const T* ev = reinterpret_cast<const T*>(&ee[u]);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) s = fmaf(qv[i], ET<T>::to_f(ke[i]) + ET<T>::to_f(ev[i]), s);
            s = group_sum<G>(s);
            if (j < j1) {
                if (kp && kp[j]) s = -INFINITY;
                if (c == 0) ps[j - j0] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float m_safe = mx == -INFINITY ? 0.f : mx;

    // ---- pass 2: exp, sum, P.V
    float o[CH], lsum = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = 0.f;
    for (int jb = j0 + wid * KPW; jb < j1; jb += KPI * U) {
        chunk16 vv[U];
        if (jb == j0 + wid * KPW) {
#pragma unroll
            for (int u = 0; u < U; ++u) vv[u] = v0[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jc = min(jb + u * KPI + kslot, j1 - 1);
                vv[u] = ld_kv(vc + (size_t)jc * DH + cc * CH);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KPI + kslot;
            const float p = j < j1 ? ET<T>::fexp(ps[j - j0] - m_safe) : 0.f;
            const T* ve = reinterpret_cast<const T*>(&vv[u]);
#pragma unroll
            for (int i = 0; i < CH; ++i) o[i] = fmaf(p, ET<T>::to_f(ve[i]), o[i]);
            if (c == 0) lsum += p;
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < CH; ++i) osum[wid][kslot][c * CH + i] = o[i];
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wid] = lsum;
    __syncthreads();
    if (tid < DH) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < KPW; ++k) s += osum[w][k][tid];
        pout[4 + tid] = s;
    }
    if (tid == 0) { pout[0] = mx; pout[1] = red[4] + red[5] + red[6] + red[7]; }
}


// ---------------------------------------------------------------------------------------------------------
// Fused {LayerNorm -> q | k | v projection of ONE head -> cache append -> key-split attention}: grid (row x head, nsplit + 1).
// The separate kernels (dec_gemv<PRO_LN, EPI_QKV> + dec_attn) are two launches with an all-to-all seam that does not
// exist per head: a head's attention needs only that head's q, and the keys 0..t-1 are already in the cache.  Roles:
//   y <  nsplit - 1 : cached keys [j0, j1) of [0, t): LayerNorm row + the head's 64 q columns (64 KB of weights, L2
//                     resident), then the usual (max, sum, P.V) partial;
//   y == nsplit - 1 : the new key, part A: q AND k_t columns (both weight sets requested together: one round trip),
//                     k_t appended to the cache, s_t = q.(k_t + E[M-1]) -> (max, sum) = (s_t, 1) of partial nsplit - 1;
//   y == nsplit     : the new key, part B: v_t columns, appended to the cache and written as the o part of that partial.
// Every block has exactly ONE dependent weight round trip; q never goes through memory; one launch and one boundary less per
// layer.  (A first version let one "owner" block compute q, k and v one after the other: three round trips, slower than the
// two launches it replaced -- 0.160 vs 0.150 ms per token.)  Needs nsplit >= 2.
// ---------------------------------------------------------------------------------------------------------
// EMBED = true: the FIRST layer's variant -- the block's input row is not a LayerNorm of the previous layer's sum but the
// embedding of the fed token (token row * sqrt(d - dc) | condition projection, + sinusoid row of position t, f32), so layer 0
// needs one launch like every other layer (round 4: 26 launches per token instead of 27).
template <typename T, int DH, bool EMBED = false>
__global__ __launch_bounds__(256) void dec_ln_qkv_attn_kernel(const DecArgs a, const T* __restrict__ E,
                                                              const uint8_t* __restrict__ key_pad, int ld_pad,
                                                              float* __restrict__ part, int nsplit, int M, float scale) {
    constexpr int CH = ET<T>::CH, CPR = DH / CH;            // 16-byte chunks per cache row
    constexpr int G = CPR <= 4 ? 4 : (CPR <= 8 ? 8 : 16);   // lanes per key
    constexpr int KPW = 64 / G, KPI = 4 * KPW;              // keys per wave / per block and iteration
    constexpr int U = 4;                                    // iterations in flight
    constexpr int CWQ = DH / 4;                             // projection columns per wave (one head = 4 waves x CWQ)
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [K]: the LayerNorm row, T-rounded values
    __shared__ float qs[DH], kn[DH];                                    // scaled q; the new key row (T-rounded values)
    __shared__ float ps[2048];
    __shared__ float red[12];
    __shared__ float osum[4][KPW][DH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int H = a.H, K = a.K, Mc = a.Mc;
    const int mh = blockIdx.x, role = blockIdx.y, m = mh / H, head = mh % H;
    const int t = a.t_dev ? min(*a.t_dev, min(Mc, M) - 1) : a.t;
    const bool part_a = role == nsplit - 1, part_b = role == nsplit;
    int per = (t + nsplit - 2) / (nsplit - 1);
    per = (per + KPI - 1) / KPI * KPI;
    const int j0 = role * per, j1 = min(t, j0 + per);                   // cached keys of a split block
    float* pout = part + ((size_t)mh * nsplit + min(role, nsplit - 1)) * ME_DEC_PART_REC(DH);
    if (!part_a && !part_b && j0 >= j1) {                               // empty split (short contexts)
        if (tid < DH + 4) pout[tid] = tid == 0 ? -INFINITY : 0.f;
        return;
    }
    // ---- input row first (round 5: loads return in order per wave -- issued behind the 16-32 weight loads of the wave, the four
    //      loads of the LayerNorm row waited for the weights' round trip: 1.8 of the kernel's 7.6 us), then the weights
    constexpr int NVR = 4;
    f32x4_t v[NVR];
    if constexpr (!EMBED) {
        const float* row = a.s_in + (size_t)m * K;
#pragma unroll
        for (int i = 0; i < NVR; ++i) {
            const int k = lane * 4 + 256 * i;
            v[i] = k < K ? *reinterpret_cast<const f32x4_t*>(row + k) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("" ::: "memory");
    }
    // ---- weights: in flight during the LayerNorm (q | q + k | v columns of the head)
    const int nch = K / CH, d = H * DH;
    const int chc0 = lane < nch ? lane : 0;
    const T* W0 = reinterpret_cast<const T*>(a.W) + (size_t)((part_b ? 2 * d : 0) + head * DH + wid * CWQ) * a.ldw;
    const T* W1 = W0 + (size_t)d * a.ldw;                               // part A: the k columns
    chunk16 w0[CWQ], w1[CWQ];
#pragma unroll
    for (int c = 0; c < CWQ; ++c) w0[c] = ld_w(W0 + (size_t)c * a.ldw + (size_t)chc0 * CH);
    if (part_a) {
#pragma unroll
        for (int c = 0; c < CWQ; ++c) w1[c] = ld_w(W1 + (size_t)c * a.ldw + (size_t)chc0 * CH);
    }
    // (requesting a split's K / E / V rows here, behind the weights, was measured twice -- round 3 and round 5 with the input row
    // first: 0.167 vs 0.153 ms per token on the same box -- the 12 extra loads per wave lengthen the issue phase by more than
    // the scores pass gains)
    const int kslot = lane / G, c = lane % G;
    const bool active = c < CPR;
    const int cc = active ? c : CPR - 1;
    const T* kc = reinterpret_cast<const T*>(a.kcache) + (size_t)mh * Mc * DH;
    const T* vc = reinterpret_cast<const T*>(a.vcache) + (size_t)mh * Mc * DH;
    const T* er = E + (size_t)(M - 1 - t) * DH;              // relative row of key j: E[M-1-(t-j)] = er + j * DH
    chunk16 v0[U];
    if constexpr (EMBED) {
        // ---- embedding row of sequence m (music_multi.py:89-101 for one position), same arithmetic as PRO_EMBED of dec_gemv_kernel
        const int dc = a.dc, de = K - dc;
        const float sq = sqrtf((float)de);
        for (int k = tid * 4; k < K; k += 1024) {
            f32x4_t v, r;
            const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(a.pe + (size_t)t * K + k);
            if (k < de) {
                const f32x4_t e4 = *reinterpret_cast<const f32x4_t*>(a.emb + (size_t)a.tokens[m] * de + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e4[e] * sq + p4[e];
            } else {
                const float c0 = a.cond[m * 2], c1 = a.cond[m * 2 + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a.cw[(k - de + e) * 2] * c0 + a.cw[(k - de + e) * 2 + 1] * c1 + a.cb[k - de + e] + p4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = round_to<T>(v[e]);
            if (a.x_out && head == 0 && part_a) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = v;
            *reinterpret_cast<f32x4_t*>(&xs[k]) = r;
        }
    } else {
        // ---- LayerNorm of row m (every wave computes the statistics: no cross-wave reduction, one barrier)
        constexpr int NV = NVR;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        const float mean = wave_sum(sum) / K;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane * 4 + 256 * i < K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d_ = v[i][e] - mean; vs += d_ * d_; }
            }
        }
        const float rstd = rsqrtf(wave_sum(vs) / K + a.eps);
        if (wid == 0) {                                                 // wave 0 writes the row (the others hold the same values)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = lane * 4 + 256 * i;
                if (k < K) {
                    const f32x4_t g = *reinterpret_cast<const f32x4_t*>(a.gamma + k);
                    const f32x4_t be = *reinterpret_cast<const f32x4_t*>(a.beta + k);
                    f32x4_t o, r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (v[i][e] - mean) * rstd * g[e] + be[e]; r[e] = round_to<T>(o[e]); }
                    if (a.x_out && head == 0 && part_a) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = o;
                    *reinterpret_cast<f32x4_t*>(&xs[k]) = r;
                }
            }
        }
    }
    __syncthreads();
    // ---- projection of this wave's CWQ columns; result (T-rounded) times `mul` to LDS / memory
    auto project = [&](const T* Wrow, chunk16 (&first)[CWQ], const float* bias, float* dst_lds, T* dst_cache, float* dst_f32, float mul)
                       __attribute__((always_inline)) {
        float acc[CWQ][1];
#pragma unroll
        for (int c = 0; c < CWQ; ++c) acc[c][0] = 0.f;
        dec_fma_chunks<T, 1, CWQ>(acc, first, lane < nch, xs, K, chc0);
        const int npass = (nch + 63) / 64;
        for (int ps_ = 1; ps_ < npass; ++ps_) {                          // further chunk positions (K > 64 CH)
            const int ch = lane + 64 * ps_, chc = ch < nch ? ch : 0;
            chunk16 w[CWQ];
#pragma unroll
            for (int c = 0; c < CWQ; ++c) w[c] = ld_w(Wrow + (size_t)c * a.ldw + (size_t)chc * CH);
            dec_fma_chunks<T, 1, CWQ>(acc, w, ch < nch, xs, K, chc);
        }
        float r[CWQ];
#pragma unroll
        for (int c = 0; c < CWQ; ++c) r[c] = acc[c][0];
        reduce_scatter64<CWQ>(r);
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < CWQ / 4; ++i) if ((lane & 15) == i) mine = r[i];
        if ((lane & 15) < CWQ / 4) {
            const int c = (lane & 15) + (CWQ / 4) * (lane >> 4), j = wid * CWQ + c;
            const float v = round_to<T>(mine + bias[j]);
            if (dst_lds) dst_lds[j] = v * mul;
            if (dst_cache) dst_cache[j] = ET<T>::from_f(v);
            if (dst_f32) dst_f32[j] = v;
        }
    };
    if (part_b) {                                                       // v_t: cache row + the o part of the new key's partial
        T* vcr = reinterpret_cast<T*>(a.vcache) + ((size_t)mh * Mc + t) * DH;
        project(W0, w0, a.bias + 2 * d + head * DH, nullptr, vcr, pout + 4, 1.f);
        return;
    }
    project(W0, w0, a.bias + head * DH, qs, nullptr, nullptr, scale);
    const uint8_t* kp = key_pad ? key_pad + (size_t)m * ld_pad : nullptr;
    if (part_a) {                                                       // k_t: cache row; (max, sum) = (s_t, 1)
        T* kcr = reinterpret_cast<T*>(a.kcache) + ((size_t)mh * Mc + t) * DH;
        project(W1, w1, a.bias + d + head * DH, kn, kcr, nullptr, 1.f);
        __syncthreads();
        if (wid == 0) {
            float sp = 0.f;
            for (int i = lane; i < DH; i += 64) sp = fmaf(qs[i], kn[i] + ET<T>::to_f(E[(size_t)(M - 1) * DH + i]), sp);
            sp = wave_sum(sp);
            if (kp && kp[t]) sp = -INFINITY;
            if (lane == 0) { pout[0] = sp; pout[1] = 1.f; }
        }
        return;
    }
    __syncthreads();

    // ---- attention over the cached keys [j0, j1) (same arithmetic as dec_attn_kernel)
    float qv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) qv[i] = active ? qs[cc * CH + i] : 0.f;
    float mx = -INFINITY;
    for (int jb = j0 + wid * KPW; jb < j1; jb += KPI * U) {
        chunk16 kk[U], ee[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jc = min(jb + u * KPI + kslot, j1 - 1);
            kk[u] = ld_kv(kc + (size_t)jc * DH + cc * CH);
            ee[u] = ld_kv(er + (size_t)jc * DH + cc * CH);
        }
        if (jb == j0 + wid * KPW) {
#pragma unroll
            for (int u = 0; u < U; ++u) v0[u] = ld_kv(vc + (size_t)min(jb + u * KPI + kslot, j1 - 1) * DH + cc * CH);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KPI + kslot;
            const T* ke = reinterpret_cast<const T*>(&kk[u]);
            const T* ev = reinterpret_cast<const T*>(&ee[u]);
            float s_ = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) s_ = fmaf(qv[i], ET<T>::to_f(ke[i]) + ET<T>::to_f(ev[i]), s_);
            s_ = group_sum<G>(s_);
            if (j < j1) {
                if (kp && kp[j]) s_ = -INFINITY;
                if (c == 0) ps[j - j0] = s_;
                mx = fmaxf(mx, s_);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float m_safe = mx == -INFINITY ? 0.f : mx;
    float o[CH], lsum = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = 0.f;
    for (int jb = j0 + wid * KPW; jb < j1; jb += KPI * U) {
        chunk16 vv[U];
        if (jb == j0 + wid * KPW) {
#pragma unroll
            for (int u = 0; u < U; ++u) vv[u] = v0[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jc = min(jb + u * KPI + kslot, j1 - 1);
                vv[u] = ld_kv(vc + (size_t)jc * DH + cc * CH);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u * KPI + kslot;
            const float p = j < j1 ? ET<T>::fexp(ps[j - j0] - m_safe) : 0.f;
            const T* ve = reinterpret_cast<const T*>(&vv[u]);
#pragma unroll
            for (int i = 0; i < CH; ++i) o[i] = fmaf(p, ET<T>::to_f(ve[i]), o[i]);
            if (c == 0) lsum += p;
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < CH; ++i) osum[wid][kslot][c * CH + i] = o[i];
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wid] = lsum;
    __syncthreads();
    if (tid < DH) {
        float s_ = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < KPW; ++k) s_ += osum[w][k][tid];
        pout[4 + tid] = s_;
    }
    if (tid == 0) { pout[0] = mx; pout[1] = red[4] + red[5] + red[6] + red[7]; }
}

template <typename T, int DH, bool EMBED = false>
int ln_qkv_attn_launch(const DecArgs& a, const void* E, const uint8_t* key_pad, int ld_pad, float* part, int nsplit, int M,
                       hipStream_t st) {
    const float scale = 1.f / sqrtf((float)DH);
    dec_ln_qkv_attn_kernel<T, DH, EMBED><<<dim3(a.Mr * a.H, nsplit + 1), 256, (size_t)a.K * sizeof(float), st>>>(
        a, (const T*)E, key_pad, ld_pad, part, nsplit, M, scale);
    return me_launch_status();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int PRO, int EPI, int MR, int CW, bool KS>
int gemv_launch3(const DecArgs& a, size_t lds, hipStream_t st) {
    const unsigned grid = (unsigned)(KS ? (a.N + CW - 1) / CW : (a.N + 4 * CW - 1) / (4 * CW));
    if (lds > 48 * 1024) {                                  // FFN_suf rows of the published 145 M model (8 x 3072 f32)
        static bool done[16] = {false};                     // the attribute is per device; idempotent: a benign race at worst
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        if (dev < 0 || dev >= 16 || !done[dev]) {
            (void)hipFuncSetAttribute((const void*)dec_gemv_kernel<T, PRO, EPI, MR, CW, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (dev >= 0 && dev < 16) done[dev] = true;
        }
    }
    dec_gemv_kernel<T, PRO, EPI, MR, CW, KS><<<grid, 256, lds, st>>>(a);
    return me_launch_status();
}

template <typename T, int PRO, int EPI, int MR>
int gemv_launch2(const DecArgs& a, size_t lds, hipStream_t st) {
    // Geometry: a cold weight stream arrives fastest when (nearly) every CU pulls a few KB, so the number of columns per
    // wave (4, 2 or 1) is the largest that still gives >= 192 blocks; rows of >= 256 chunks (FFN_suf) are additionally
    // split over the four waves of a block.  MIDIEMO_DEC_CW / MIDIEMO_DEC_KS override (experiments).
    static const int cw_env = getenv("MIDIEMO_DEC_CW") ? atoi(getenv("MIDIEMO_DEC_CW")) : 0;
    static const int ks_env = getenv("MIDIEMO_DEC_KS") ? atoi(getenv("MIDIEMO_DEC_KS")) : -1;
    constexpr int CH = ET<T>::CH;
    bool ks = PRO == PRO_PLAIN && a.K / CH >= 256;
    if (ks_env >= 0) ks = PRO == PRO_PLAIN && ks_env != 0;
    const int per_cw1 = ks ? a.N : (a.N + 3) / 4;           // blocks with one column per wave (per block when split over K)
    int cw = per_cw1 >= 4 * 192 ? 4 : (per_cw1 >= 2 * 192 ? 2 : 1);
    if (cw_env == 1 || cw_env == 2 || cw_env == 4) cw = cw_env;
    if constexpr (PRO == PRO_PLAIN) {
        if (ks) {
            if (cw == 4) return gemv_launch3<T, PRO, EPI, MR, 4, true>(a, lds, st);
            if (cw == 2) return gemv_launch3<T, PRO, EPI, MR, 2, true>(a, lds, st);
            return gemv_launch3<T, PRO, EPI, MR, 1, true>(a, lds, st);
        }
    }
    if (cw == 4) return gemv_launch3<T, PRO, EPI, MR, 4, false>(a, lds, st);
    if (cw == 2) return gemv_launch3<T, PRO, EPI, MR, 2, false>(a, lds, st);
    return gemv_launch3<T, PRO, EPI, MR, 1, false>(a, lds, st);
}

template <typename T, int PRO, int EPI>
int gemv_launch(const DecArgs& a, hipStream_t st) {
    constexpr int CH = ET<T>::CH;
    if (a.Mr < 1 || a.Mr > 8 || a.N <= 0 || a.K <= 0 || a.K % CH || a.K % 4 || a.ldw % CH) return ME_ERR_BAD_SHAPE;
    if (PRO == PRO_LN && a.K > 1024) return ME_ERR_BAD_SHAPE;               // the LayerNorm prologue keeps a row in registers
    if (PRO == PRO_EMBED && (a.dc < 0 || a.dc >= a.K || a.dc % 4 || (a.K - a.dc) % 4)) return ME_ERR_BAD_SHAPE;
    if (PRO == PRO_ATTN && (a.nsplit > DEC_NSMAX || a.dh % 4)) return ME_ERR_BAD_SHAPE;
    if ((PRO == PRO_HILO || PRO == PRO_PLAIN) && (a.ldx % CH || !aligned16(a.x_hi) || (a.x_lo && !aligned16(a.x_lo)))) return ME_ERR_ALIGNMENT;
    if (!a.W || !aligned16(a.W)) return a.W ? ME_ERR_ALIGNMENT : ME_ERR_NULL;
    const int mr = a.Mr <= 4 ? 4 : 8;
    size_t lds = (size_t)mr * a.K * sizeof(float);
    if (PRO == PRO_ATTN) lds += (size_t)a.Mr * a.H * DEC_NSMAX * sizeof(float);
    if (lds > 96 * 1024) return ME_ERR_BAD_SHAPE;
    if (mr == 4) return gemv_launch2<T, PRO, EPI, 4>(a, lds, st);
    return gemv_launch2<T, PRO, EPI, 8>(a, lds, st);
}

template <typename T, int DH>
int attn_launch(const void* q, const void* kc, const void* vc, const void* E, const uint8_t* key_pad, int ld_pad, float* part,
                int nsplit, int Mr, int H, int M, int Mc, int t, const int32_t* t_dev, hipStream_t st) {
    const float scale = 1.f / sqrtf((float)DH);
    dec_attn_kernel<T, DH><<<dim3(Mr * H, nsplit), 256, 0, st>>>((const T*)q, (const T*)kc, (const T*)vc, (const T*)E, key_pad, ld_pad,
                                                              part, nsplit, H, M, Mc, t, t_dev, scale);
    return me_launch_status();
}

}  // namespace

#define ME_DEC_T(CALL)                                          \
    if (dtype == ME_F32) { typedef float T; return CALL; }      \
    if (dtype == ME_BF16) { typedef bf16_t T; return CALL; }    \
    if (dtype == ME_F16) { typedef f16_t T; return CALL; }      \
    return ME_ERR_BAD_DTYPE;

extern "C" {

int me_dec_qkv(const float* s_in, const float* gamma, const float* beta, float eps, const void* x_hi, const void* x_lo,
               const void* Wqkv, const float* bqkv, float* x_out, void* q_out, void* kcache, void* vcache, int Mr, int d,
               int H, int dh, int Mc, int t, const int32_t* t_dev, int dtype, void* stream) {
    me_clear_error();
    if (!Wqkv || !q_out || !kcache || !vcache) return ME_ERR_NULL;
    if (!s_in && !x_hi) return ME_ERR_NULL;
    if (s_in && (!gamma || !beta)) return ME_ERR_NULL;
    if (H <= 0 || dh <= 0 || H * dh != d || Mc <= 0) return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc)) return ME_ERR_BAD_SHAPE;
    DecArgs a = {};
    a.s_in = s_in; a.gamma = gamma; a.beta = beta; a.eps = eps; a.x_hi = x_hi; a.x_lo = x_lo; a.ldx = d; a.x_out = x_out;
    a.W = Wqkv; a.ldw = d; a.bias = bqkv; a.Mr = Mr; a.N = 3 * d; a.K = d; a.y = q_out; a.ldy = d; a.kcache = kcache;
    a.vcache = vcache; a.Mc = Mc; a.t = t; a.t_dev = t_dev; a.H = H; a.dh = dh;
    hipStream_t st = (hipStream_t)stream;
    if (s_in) { ME_DEC_T((gemv_launch<T, PRO_LN, EPI_QKV>(a, st))) }
    ME_DEC_T((gemv_launch<T, PRO_HILO, EPI_QKV>(a, st)))
}

int me_dec_embed_qkv(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb,
                     const float* pe, int d_cond, const void* Wqkv, const float* bqkv, float* x_out, void* q_out, void* kcache,
                     void* vcache, int Mr, int d, int H, int dh, int Mc, int t, const int32_t* t_dev, int dtype, void* stream) {
    me_clear_error();
    if (!tokens || !emb || !pe || !Wqkv || !q_out || !kcache || !vcache) return ME_ERR_NULL;
    if (d_cond > 0 && (!cond || !cw || !cb)) return ME_ERR_NULL;
    if (H <= 0 || dh <= 0 || H * dh != d || Mc <= 0) return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc)) return ME_ERR_BAD_SHAPE;
    DecArgs a = {};
    a.tokens = tokens; a.cond = cond; a.emb = emb; a.cw = cw; a.cb = cb; a.pe = pe; a.dc = d_cond > 0 ? d_cond : 0; a.x_out = x_out;
    a.W = Wqkv; a.ldw = d; a.bias = bqkv; a.Mr = Mr; a.N = 3 * d; a.K = d; a.y = q_out; a.ldy = d; a.kcache = kcache;
    a.vcache = vcache; a.Mc = Mc; a.t = t; a.t_dev = t_dev; a.H = H; a.dh = dh;
    hipStream_t st = (hipStream_t)stream;
    ME_DEC_T((gemv_launch<T, PRO_EMBED, EPI_QKV>(a, st)))
}

int me_dec_attn(const void* q, const void* kcache, const void* vcache, const void* E, const uint8_t* key_pad, int ld_pad,
                float* part, int nsplit, int Mr, int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype,
                void* stream) {
    me_clear_error();
    if (!q || !kcache || !vcache || !E || !part) return ME_ERR_NULL;
    if (Mr <= 0 || H <= 0 || nsplit <= 0 || nsplit > DEC_NSMAX || Mc <= 0 || M <= 0) return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc || t >= M)) return ME_ERR_BAD_SHAPE;
    if ((Mc + nsplit - 1) / nsplit + 64 > 2048 + 64) return ME_ERR_BAD_SHAPE;       // the score buffer holds 2048 keys per split
    if (!aligned16(q) || !aligned16(kcache) || !aligned16(vcache) || !aligned16(E)) return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
#define ME_DEC_ATTN_CASE(DHV)                                                                                             \
    if (dh == DHV) {                                                                                                     \
        constexpr int DH = DHV;                                                                                          \
        ME_DEC_T((attn_launch<T, DH>(q, kcache, vcache, E, key_pad, ld_pad, part, nsplit, Mr, H, M, Mc, t, t_dev, st)))    \
    }
    ME_DEC_ATTN_CASE(64)
    ME_DEC_ATTN_CASE(48)
    ME_DEC_ATTN_CASE(32)
#undef ME_DEC_ATTN_CASE
    return ME_ERR_BAD_SHAPE;
}

int me_dec_ln_qkv_attn(const float* s_in, const float* gamma, const float* beta, float eps, const void* Wqkv, const float* bqkv,
                       float* x_out, void* kcache, void* vcache, const void* E, const uint8_t* key_pad, int ld_pad, float* part,
                       int nsplit, int Mr, int d, int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype,
                       void* stream) {
    me_clear_error();
    if (!s_in || !gamma || !beta || !Wqkv || !bqkv || !kcache || !vcache || !E || !part) return ME_ERR_NULL;
    if (Mr <= 0 || H <= 0 || dh <= 0 || H * dh != d || d > 1024 || d % 8 || nsplit < 2 || nsplit > DEC_NSMAX || Mc <= 0 || M <= 0)
        return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc || t >= M)) return ME_ERR_BAD_SHAPE;
    if ((Mc + nsplit - 2) / (nsplit - 1) + 64 > 2048 + 64) return ME_ERR_BAD_SHAPE;   // score buffer: 2048 keys per split
    if (!aligned16(Wqkv) || !aligned16(kcache) || !aligned16(vcache) || !aligned16(E) || !aligned16(s_in)) return ME_ERR_ALIGNMENT;
    DecArgs a = {};
    a.s_in = s_in; a.gamma = gamma; a.beta = beta; a.eps = eps; a.x_out = x_out; a.W = Wqkv; a.ldw = d; a.bias = bqkv; a.Mr = Mr;
    a.N = 3 * d; a.K = d; a.kcache = kcache; a.vcache = vcache; a.Mc = Mc; a.t = t; a.t_dev = t_dev; a.H = H; a.dh = dh;
    hipStream_t st = (hipStream_t)stream;
#define ME_DEC_FUSED_CASE(DHV)                                                                                    \
    if (dh == DHV) {                                                                                            \
        constexpr int DH = DHV;                                                                                 \
        ME_DEC_T((ln_qkv_attn_launch<T, DH>(a, E, key_pad, ld_pad, part, nsplit, M, st)))                         \
    }
    ME_DEC_FUSED_CASE(64)
    ME_DEC_FUSED_CASE(48)
    ME_DEC_FUSED_CASE(32)
#undef ME_DEC_FUSED_CASE
    return ME_ERR_BAD_SHAPE;
}

int me_dec_embed_qkv_attn(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb,
                          const float* pe, int d_cond, const void* Wqkv, const float* bqkv, float* x_out, void* kcache,
                          void* vcache, const void* E, const uint8_t* key_pad, int ld_pad, float* part, int nsplit, int Mr, int d,
                          int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype, void* stream) {
    me_clear_error();
    if (!tokens || !emb || !pe || !Wqkv || !bqkv || !kcache || !vcache || !E || !part) return ME_ERR_NULL;
    if (d_cond > 0 && (!cond || !cw || !cb)) return ME_ERR_NULL;
    if (Mr <= 0 || H <= 0 || dh <= 0 || H * dh != d || d > 1024 || d % 8 || nsplit < 2 || nsplit > DEC_NSMAX || Mc <= 0 || M <= 0 ||
        d_cond >= d || (d_cond > 0 && d_cond % 4))
        return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc || t >= M)) return ME_ERR_BAD_SHAPE;
    if ((Mc + nsplit - 2) / (nsplit - 1) + 64 > 2048 + 64) return ME_ERR_BAD_SHAPE;   // score buffer: 2048 keys per split
    if (!aligned16(Wqkv) || !aligned16(kcache) || !aligned16(vcache) || !aligned16(E) || !aligned16(emb) || !aligned16(pe)) return ME_ERR_ALIGNMENT;
    DecArgs a = {};
    a.tokens = tokens; a.cond = cond; a.emb = emb; a.cw = cw; a.cb = cb; a.pe = pe; a.dc = d_cond > 0 ? d_cond : 0; a.x_out = x_out;
    a.W = Wqkv; a.ldw = d; a.bias = bqkv; a.Mr = Mr; a.N = 3 * d; a.K = d; a.kcache = kcache; a.vcache = vcache; a.Mc = Mc; a.t = t;
    a.t_dev = t_dev; a.H = H; a.dh = dh;
    hipStream_t st = (hipStream_t)stream;
#define ME_DEC_FUSED_CASE(DHV)                                                                                    \
    if (dh == DHV) {                                                                                            \
        constexpr int DH = DHV;                                                                                 \
        ME_DEC_T((ln_qkv_attn_launch<T, DH, true>(a, E, key_pad, ld_pad, part, nsplit, M, st)))                   \
    }
    ME_DEC_FUSED_CASE(64)
    ME_DEC_FUSED_CASE(48)
    ME_DEC_FUSED_CASE(32)
#undef ME_DEC_FUSED_CASE
    return ME_ERR_BAD_SHAPE;
}

int me_dec_proj_resid(const float* part, int nsplit, int H, int dh, const void* x_T, int ldx, const void* W, int ldw,
                      const float* bias, const float* resid, float* out, int Mr, int N, int K, int dtype, void* stream) {
    me_clear_error();
    if (!W || !resid || !out || (!part && !x_T)) return ME_ERR_NULL;
    if (part && (nsplit <= 0 || H <= 0 || dh <= 0 || H * dh != K)) return ME_ERR_BAD_SHAPE;
    DecArgs a = {};
    a.part = part; a.nsplit = nsplit; a.H = H; a.dh = dh; a.x_hi = x_T; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias;
    a.Mr = Mr; a.N = N; a.K = K; a.y = out; a.ldy = N; a.resid = resid;
    hipStream_t st = (hipStream_t)stream;
    if (part) { ME_DEC_T((gemv_launch<T, PRO_ATTN, EPI_RESID>(a, st))) }
    ME_DEC_T((gemv_launch<T, PRO_PLAIN, EPI_RESID>(a, st)))
}

int me_dec_ln_proj(const float* s_in, const float* gamma, const float* beta, float eps, const void* W, int ldw,
                   const float* bias, float* x_out, void* y, int ldy, int Mr, int N, int K, int flags, int dtype,
                   void* stream) {
    me_clear_error();
    if (!s_in || !gamma || !beta || !W || !y) return ME_ERR_NULL;
    DecArgs a = {};
    a.s_in = s_in; a.gamma = gamma; a.beta = beta; a.eps = eps; a.x_out = x_out; a.W = W; a.ldw = ldw; a.bias = bias;
    a.Mr = Mr; a.N = N; a.K = K; a.y = y; a.ldy = ldy; a.relu = (flags & ME_EPI_RELU) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (flags & ME_EPI_OUT_F32) { ME_DEC_T((gemv_launch<T, PRO_LN, EPI_F32>(a, st))) }
    ME_DEC_T((gemv_launch<T, PRO_LN, EPI_T>(a, st)))
}

}  // extern "C"
