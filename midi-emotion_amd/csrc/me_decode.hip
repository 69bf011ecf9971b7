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
#include "me_common.h"
#include <type_traits>

namespace {

enum { PRO_LN = 0, PRO_HILO = 1, PRO_ATTN = 2, PRO_PLAIN = 3 };
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
    const float* part;      // PRO_ATTN: f32 [Mr*H][nsplit][dh + 2] = (max, sum, o[dh])
    int nsplit, H, dh;
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

template <typename T> ME_DEV float round_to(float x) { return ET<T>::to_f(ET<T>::from_f(x)); }

// sum over the G lanes (G = 4, 8, 16) of an aligned lane group; every lane receives the total
template <int G> ME_DEV float group_sum(float v) {
    v += dpp_move<0xB1>(v);                     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);                     // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_move<0x141>(v);        // row_half_mirror
    if (G >= 16) v += dpp_move<0x140>(v);       // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// y = epilogue( T(prologue(...)) . W^T + bias ).  Block = 4 waves, wave = CW output columns, lanes walk the contraction
// dimension in 16-byte chunks (K = 512 bf16: one chunk per lane and column), MR rows share every weight chunk.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int PRO, int EPI, int MR, int CW>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const DecArgs a) {
    constexpr int CH = ET<T>::CH;
    extern __shared__ __attribute__((aligned(16))) float xs[];          // [MR][K]: the projection's input rows (T-rounded values)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = a.K, Mr = a.Mr;

    // ---- prologue: input rows into LDS
    if constexpr (PRO == PRO_LN) {
        for (int m = wid; m < MR; m += 4) {
            if (m >= Mr) { for (int k = lane; k < K; k += 64) xs[m * K + k] = 0.f; continue; }
            const float* s = a.s_in + (size_t)m * K;
            float sum = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(s + k);
                *reinterpret_cast<f32x4_t*>(&xs[m * K + k]) = v;
                sum += v[0] + v[1] + v[2] + v[3];
            }
            const float mean = wave_sum(sum) / K;
            float vs = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(&xs[m * K + k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d_ = v[i] - mean; vs += d_ * d_; }
            }
            const float rstd = rsqrtf(wave_sum(vs) / K + a.eps);
            for (int k = lane * 4; k < K; k += 256) {
                f32x4_t v = *reinterpret_cast<const f32x4_t*>(&xs[m * K + k]);
                const f32x4_t g = *reinterpret_cast<const f32x4_t*>(a.gamma + k);
                const f32x4_t be = *reinterpret_cast<const f32x4_t*>(a.beta + k);
                f32x4_t o, r;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o[i] = (v[i] - mean) * rstd * g[i] + be[i]; r[i] = round_to<T>(o[i]); }
                if (a.x_out && blockIdx.x == 0) *reinterpret_cast<f32x4_t*>(a.x_out + (size_t)m * K + k) = o;
                *reinterpret_cast<f32x4_t*>(&xs[m * K + k]) = r;
            }
        }
    } else if constexpr (PRO == PRO_HILO || PRO == PRO_PLAIN) {
        const T* hi = reinterpret_cast<const T*>(a.x_hi);
        const T* lo = reinterpret_cast<const T*>(a.x_lo);
        for (int idx = tid; idx < MR * K; idx += 256) {
            const int m = idx / K, k = idx % K;
            float v = 0.f;
            if (m < Mr) {
                v = ET<T>::to_f(hi[(size_t)m * a.ldx + k]);
                if (PRO == PRO_HILO && lo) v += ET<T>::to_f(lo[(size_t)m * a.ldx + k]);
                if (a.x_out && blockIdx.x == 0) a.x_out[(size_t)m * K + k] = v;
            }
            xs[idx] = round_to<T>(v);
        }
    } else {    // PRO_ATTN: combine the key-split partials of every (row, head)
        const int dh = a.dh, ns = a.nsplit, rec = dh + 2;
        for (int idx = tid; idx < MR * K; idx += 256) {
            const int m = idx / K, k = idx % K;
            float v = 0.f;
            if (m < Mr) {
                const int h = k / dh, dd = k % dh;
                const float* p = a.part + (size_t)(m * a.H + h) * ns * rec;
                float mx = -INFINITY;
                for (int s = 0; s < ns; ++s) mx = fmaxf(mx, p[s * rec]);
                const float msafe = mx == -INFINITY ? 0.f : mx;
                float l = 0.f, o = 0.f;
                for (int s = 0; s < ns; ++s) {
                    const float w = ET<T>::fexp(p[s * rec] - msafe);          // exp(-inf) = 0 for empty / fully masked splits
                    l += w * p[s * rec + 1];
                    o += w * p[s * rec + 2 + dd];
                }
                v = o / l;                                                   // every key masked: 0 / 0 = NaN like the reference's softmax
                if (a.x_out && blockIdx.x == 0) a.x_out[(size_t)m * K + k] = v;
            }
            xs[idx] = round_to<T>(v);
        }
    }
    __syncthreads();

    // ---- projection: CW columns per wave
    const int n0 = (blockIdx.x * 4 + wid) * CW;
    if (n0 >= a.N) return;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int nch = K / CH;
    float acc[CW][MR];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
    constexpr int U = 2;                                    // chunk positions per lane in flight (x CW columns)
    for (int ch0 = lane; ch0 < nch; ch0 += 64 * U) {
        chunk16 w[U][CW];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + 64 * u;
            ok[u] = ch < nch;
            const int chc = ok[u] ? ch : ch0;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int n = min(n0 + c, a.N - 1);         // clamped: results of columns >= N are never stored
                w[u][c] = ld_chunk(W + (size_t)n * a.ldw + (size_t)chc * CH);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int chc = ok[u] ? ch0 + 64 * u : ch0;
            float wf[CW][CH];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const T* we = reinterpret_cast<const T*>(&w[u][c]);
#pragma unroll
                for (int i = 0; i < CH; ++i) wf[c][i] = ok[u] ? ET<T>::to_f(we[i]) : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float xv[CH];
#pragma unroll
                for (int q4 = 0; q4 < CH / 4; ++q4) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(&xs[m * K + chc * CH + 4 * q4]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xv[4 * q4 + i] = v[i];
                }
#pragma unroll
                for (int c = 0; c < CW; ++c)
#pragma unroll
                    for (int i = 0; i < CH; ++i) acc[c][m] = fmaf(wf[c][i], xv[i], acc[c][m]);
            }
        }
    }
    // ---- reduce over the lanes; lane c * MR + m keeps output (column c, row m)
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float tot = wave_sum(acc[c][m]);
            if (lane == c * MR + m) mine = tot;
        }
    if (lane >= CW * MR) return;
    const int c = lane / MR, m = lane % MR, n = n0 + c;
    if (n >= a.N || m >= Mr) return;
    float v = mine + (a.bias ? a.bias[n] : 0.f);
    if (a.relu) v = fmaxf(v, 0.f);
    if constexpr (EPI == EPI_T) {
        reinterpret_cast<T*>(a.y)[(size_t)m * a.ldy + n] = ET<T>::from_f(v);
    } else if constexpr (EPI == EPI_F32) {
        reinterpret_cast<float*>(a.y)[(size_t)m * a.ldy + n] = v;
    } else if constexpr (EPI == EPI_RESID) {
        reinterpret_cast<float*>(a.y)[(size_t)m * a.ldy + n] = a.resid[(size_t)m * a.N + n] + v;
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
    float* pout = part + ((size_t)bh * nsplit + split) * (DH + 2);
    if (j0 >= j1) {                                          // empty split
        if (tid < DH + 2) pout[tid] = tid == 0 ? -INFINITY : 0.f;
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

    // ---- pass 1: scores of the range into LDS, running maximum
    float mx = -INFINITY;
    for (int jb = j0 + wid * KPW; jb < j1; jb += KPI * U) {
        chunk16 kk[U], ee[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jc = min(jb + u * KPI + kslot, j1 - 1);
            kk[u] = ld_chunk(kc + (size_t)jc * DH + cc * CH);
            ee[u] = ld_chunk(er + (size_t)jc * DH + cc * CH);
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
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jc = min(jb + u * KPI + kslot, j1 - 1);
            vv[u] = ld_chunk(vc + (size_t)jc * DH + cc * CH);
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
        pout[2 + tid] = s;
    }
    if (tid == 0) { pout[0] = mx; pout[1] = red[4] + red[5] + red[6] + red[7]; }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int PRO, int EPI>
int gemv_launch(const DecArgs& a, hipStream_t st) {
    constexpr int CH = ET<T>::CH, CW = 4;
    if (a.Mr < 1 || a.Mr > 8 || a.N <= 0 || a.K <= 0 || a.K % CH || a.K % 4 || a.ldw % CH) return ME_ERR_BAD_SHAPE;
    if (!a.W || !aligned16(a.W)) return a.W ? ME_ERR_ALIGNMENT : ME_ERR_NULL;
    const int mr = a.Mr <= 4 ? 4 : 8;
    const size_t lds = (size_t)mr * a.K * sizeof(float);
    if (lds > 96 * 1024) return ME_ERR_BAD_SHAPE;
    const unsigned grid = (unsigned)((a.N + 4 * CW - 1) / (4 * CW));
    if (mr == 4) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dec_gemv_kernel<T, PRO, EPI, 4, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dec_gemv_kernel<T, PRO, EPI, 4, CW><<<grid, 256, lds, st>>>(a);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dec_gemv_kernel<T, PRO, EPI, 8, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dec_gemv_kernel<T, PRO, EPI, 8, CW><<<grid, 256, lds, st>>>(a);
    }
    return me_launch_status();
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

int me_dec_attn(const void* q, const void* kcache, const void* vcache, const void* E, const uint8_t* key_pad, int ld_pad,
                float* part, int nsplit, int Mr, int H, int dh, int M, int Mc, int t, const int32_t* t_dev, int dtype,
                void* stream) {
    me_clear_error();
    if (!q || !kcache || !vcache || !E || !part) return ME_ERR_NULL;
    if (Mr <= 0 || H <= 0 || nsplit <= 0 || nsplit > 64 || Mc <= 0 || M <= 0) return ME_ERR_BAD_SHAPE;
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
