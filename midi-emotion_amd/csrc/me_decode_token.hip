// One decode token as ONE persistent launch (gfx950): the model call of generate.py:116-119 for one new position per
// sequence -- embedding, n_layer x {LayerNorm -> q|k|v of a head -> cache append -> key-split attention, combine, Wo + residual,
// LayerNorm -> FFN_pre + ReLU, FFN_suf + residual}, LayerNorm -> vocabulary head (music_multi.py:89-135, 196-237) -- in a
// single kernel of one block per CU.
//
// Why: the per-stage launch chain of me_decode.hip (26 launches per token at 6 layers) spends 1.7 us per kernel boundary and
// starts every kernel with a cold weight round trip (~2.5 us) that nothing can overlap, because a kernel does not exist before
// its predecessor has ended.  A software grid barrier is no cheaper (3.6-10.7 us at 256 blocks, tools/ubench_grid_barrier.hip)
// -- but the seams do not need a barrier, they need the DATA: here every value that crosses a seam is an 8-byte record
// {payload, tag}, stored and polled with agent-scope atomic 64-bit accesses (single-copy atomic: the record validates
// itself; no counter, no fence, no L2 write-back), tag = (launch epoch, stage).  Measured: 2.2 us per all-to-all exchange of
// 2048 values between 256 blocks (tools/ubench_ll_exchange.hip), and since the consumer block already exists it requests
// its weight rows / cache rows BEFORE it polls -- the round trip that was serial is now under the exchange.
//
// Stage order per layer (tags 6 l + 0..5) and who needs what:
//   S0  every block: rows x = LayerNorm2(s2 of layer l-1) (or the embedding), kept in LDS rounded (projection operand) and
//       unrounded (residual of S2); q | k | v projection, 8 columns per block.  (A first version computed q inside the attention
//       stage like dec_ln_qkv_attn_kernel does: every block of a (sequence, head) group fetched the head's 64 KB of q rows --
//       16 MB per layer through L2, 8-12 us; profiles/r06_decode_token.txt.)                   exchange: qkv (a head's slice per reader)
//   S1  role blocks (sequence m, head h): splits 0..ns-2 = cached keys [j0, j1) of [0, t): scores, (max, sum, P.V) -> partial
//       records; split ns-1 = the new key: k_t, v_t appended to the cache, (s_t, 1, v_t).      exchange: part (group-local)
//   S1b the ns blocks of a (m, h) group poll the group's partials, each combines dh / ns of the head's outputs. exchange: att
//   S2  Wo + bias + residual(x)                                                                exchange: s1
//   S3  LayerNorm1 -> FFN_pre + bias + ReLU                                                    exchange: hid
//   S4  FFN_suf + bias + residual(LayerNorm1 output)                                           exchange: s2
// then LayerNorm2 of the last layer -> head -> logits (plain stores; the pick / sampling kernel is the next launch).
// Request order inside a layer: the block's q | k | v rows before the first poll; the split's K | E | V rows right behind it;
// its Wo / FFN rows after it has published q | k | v (loads return in order per wave: a poll issued behind cold rows waits for them).
// A buffer is rewritten one layer later; its writer has by then passed at least one all-to-all poll that every block only
// answers after its own read of the previous generation, so single buffers suffice.  Across launches the kernel boundary
// orders everything; the epoch (advanced by the last block to finish) keeps stale records of earlier tokens invalid.
//
// Arithmetic, rounding points and summation orders are those of me_decode.hip (shared helpers: me_decode_common.h): the
// per-column contraction order (lane -> chunks lane, lane + 64, ...; the four K quarters of FFN_suf summed pairwise), the
// LayerNorm reductions, the split softmax and the combine are identical, so the two paths agree bit for bit (tested).
#include "me_decode_common.h"

namespace {

#ifdef ME_TOK_PROF      // development build only (tools/prof_dec_token.py): 100 MHz stamps of three blocks behind the records
#define TOK_PROF_OFF (ME_DEC_TOKEN_ROWS * 8 * (1024 + 64 + 2))      /* behind the largest partial area (d <= 1024) */
#define TOK_STAMP(i) do { if (prof && threadIdx.x == 0) prof[(i)] = wall_clock64(); } while (0)
#else
#define TOK_STAMP(i) do { } while (0)
#endif
constexpr int TOK_SPIN_MAX = 1 << 18;       // poll rounds (~1 us each) before a block gives up and raises the error word

struct TokArgs {
    const int64_t* tokens; const float* cond; const float* emb; const float* cw; const float* cb; const float* pe; int dc;
    const me_dec_layer* L; int n_layer;     // device table
    const void* Wf; int ldwf; const float* bf; int V; float* logits; int ld_logits;
    int Mr, d, di, H, nsplit, M, Mc, t; const int32_t* t_dev; float eps, scale;
    unsigned* ctl;                          // [0] epoch, [1] blocks done, [2] error
    unsigned long long *x_s2, *x_s1, *x_att, *x_qkv, *x_hid, *x_part;
    int ks2;                                // FFN_suf contraction split over the four waves (K / CH >= 256, as gemv_launch2 decides)
};

ME_DEV unsigned long long ld_rec(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ME_DEV void st_rec(unsigned long long* p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ME_DEV unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
ME_DEV float u2f(unsigned x) { return __builtin_bit_cast(float, x); }
template <typename T> ME_DEV unsigned pack2(float a, float b) {          // two values of a 16-bit T in one payload word
    const T x = ET<T>::from_f(a), y = ET<T>::from_f(b);
    return (unsigned)__builtin_bit_cast(uint16_t, x) | ((unsigned)__builtin_bit_cast(uint16_t, y) << 16);
}

// Poll records 0 .. n-1 (record i at addr(i)) until each carries `tag`; sink(i, payload) once per record.  R records per thread
// and round in flight; a thread only re-requests what is still pending.
template <int R, typename A, typename F>
ME_DEV void poll_records(int tid, int n, unsigned tag, unsigned* err, A&& addr, F&& sink) {
    for (int base = 0; base < n; base += 256 * R) {
        unsigned long long w[R];
        unsigned pending = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) if (base + tid + 256 * r < n) pending |= 1u << r;
        int spins = 0;
        while (pending) {
#pragma unroll
            for (int r = 0; r < R; ++r) if (pending >> r & 1) w[r] = ld_rec(addr(base + tid + 256 * r));
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((pending >> r & 1) && (unsigned)(w[r] >> 32) == tag) { pending &= ~(1u << r); sink(base + tid + 256 * r, (unsigned)w[r]); }
            if (pending && (++spins & 1023) == 0) {        // every 1024 rounds: has another block given up? have we run into the bound?
                if (spins > TOK_SPIN_MAX) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > TOK_SPIN_MAX || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) pending = 0;
            }
        }
    }
}

// LayerNorm of one row held in LDS (f32 [K], K <= 1024) by one wave: the reductions and the per-lane element order of
// dec_gemv_kernel<PRO_LN> / dec_ln_qkv_attn_kernel.  g / be: this lane's gamma / beta quads (requested before the poll).
template <typename T>
ME_DEV void ln_row(const float* src, int K, const f32x4_t (&g)[4], const f32x4_t (&be)[4], float eps, bool valid, float* xo_row, float* xs_row, int lane) {
    f32x4_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + 256 * i;
        v[i] = (valid && k < K) ? *reinterpret_cast<const f32x4_t*>(src + k) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum(sum) / K;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane * 4 + 256 * i < K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d_ = v[i][e] - mean; vs += d_ * d_; }
        }
    }
    const float rstd = rsqrtf(wave_sum(vs) / K + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + 256 * i;
        if (k < K) {
            f32x4_t o = {0.f, 0.f, 0.f, 0.f}, r = o;
            if (valid) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (v[i][e] - mean) * rstd * g[i][e] + be[i][e]; r[e] = round_to<T>(o[e]); }
            }
            *reinterpret_cast<f32x4_t*>(xo_row + k) = o;
            *reinterpret_cast<f32x4_t*>(xs_row + k) = r;
        }
    }
}

// The projection of dec_gemv_kernel on the block's columns of virtual block vb: weights of the first chunk positions come
// in `wp` (requested by gemv_prefetch before the poll).  Returns the owner lanes' value; (oc, om) = its column offset / row.
template <typename T, int CW, bool KS>
ME_DEV void gemv_prefetch(int tid, chunk16 (&wp)[2][CW], const T* Wall, int ldw, int N, int K, int vb) {
    constexpr int CH = ET<T>::CH, PF = 2;
    const int lane = tid & 63, wid = tid >> 6;
    const int nch_all = K / CH, kq = KS ? (nch_all + 3) / 4 : 0, ch_lo = KS ? wid * kq : 0;
    const int nch = KS ? max(0, min(kq, nch_all - ch_lo)) : nch_all;
    const T* W = Wall + (size_t)ch_lo * CH;
    const int n0 = KS ? vb * CW : (vb * 4 + wid) * CW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int ch = lane + 64 * u, chc = ch < nch ? ch : 0;
        if (64 * u < nch) {
#pragma unroll
            for (int c = 0; c < CW; ++c) wp[u][c] = ld_w(W + (size_t)min(n0 + c, N - 1) * ldw + (size_t)chc * CH);
        }
    }
}
template <typename T, int MR, int CW, bool KS>
ME_DEV float gemv_run(int tid, chunk16 (&wp)[2][CW], const T* Wall, int ldw, int N, int K, int vb, float* xs, float* ks = nullptr) {
    constexpr int CH = ET<T>::CH, PF = 2, NV = CW * MR;
    const int lane = tid & 63, wid = tid >> 6;
    const int nch_all = K / CH, kq = KS ? (nch_all + 3) / 4 : 0, ch_lo = KS ? wid * kq : 0;
    const int nch = KS ? max(0, min(kq, nch_all - ch_lo)) : nch_all;
    const T* W = Wall + (size_t)ch_lo * CH;
    const float* xw = xs + ch_lo * CH;
    const int n0 = KS ? vb * CW : (vb * 4 + wid) * CW;
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
    constexpr int U = 2;
    for (int ch0 = lane + 64 * PF; ch0 < nch; ch0 += 64 * U) {
        chunk16 w[U][CW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + 64 * u, chc = ch < nch ? ch : ch0;
#pragma unroll
            for (int c = 0; c < CW; ++c) w[u][c] = ld_w(W + (size_t)min(n0 + c, N - 1) * ldw + (size_t)chc * CH);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ch = ch0 + 64 * u;
            dec_fma_chunks<T, MR, CW>(acc, w[u], ch < nch, xw, K, ch < nch ? ch : ch0);
        }
    }
    float red[NV];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int m = 0; m < MR; ++m) red[c * MR + m] = acc[c][m];
    reduce_scatter64<NV>(red);
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) if ((lane & 15) == i) mine = red[i];
    if constexpr (KS) {                                     // the four K quarters meet in LDS: [4 waves][NV] floats at ks (NOT in xs as in
        __syncthreads();                                    // dec_gemv_kernel: a block may walk several column groups over the same rows)
        if ((lane & 15) < NV / 4) ks[wid * NV + (lane & 15) + (NV / 4) * (lane >> 4)] = mine;
        __syncthreads();
        if (wid == 0 && (lane & 15) < NV / 4) {
            const int o_ = (lane & 15) + (NV / 4) * (lane >> 4);
            mine = (ks[o_] + ks[NV + o_]) + (ks[2 * NV + o_] + ks[3 * NV + o_]);
        }
    }
    return mine;
}
// owner lanes of gemv_run's result: output index oidx = (lane & 15) + (NV / 4)(lane >> 4) -> column n0 + oidx / MR, row oidx % MR
template <int MR, int CW, bool KS>
ME_DEV bool gemv_owner(int tid, int N, int Mr, int vb, int& n, int& m) {
    constexpr int NV = CW * MR;
    const int lane = tid & 63, wid = tid >> 6;
    const int n0 = KS ? vb * CW : (vb * 4 + wid) * CW;
    const int oidx = (lane & 15) + (NV / 4) * (lane >> 4);
    n = n0 + oidx / MR;
    m = oidx % MR;
    return (lane & 15) < NV / 4 && n < N && m < Mr && (!KS || wid == 0);
}

template <typename T, int DH, int MR>
__global__ __launch_bounds__(256, 1) void dec_token_kernel(const TokArgs a) {
    constexpr int CH = ET<T>::CH, CPR = DH / CH;
    constexpr int GK = CPR <= 4 ? 4 : (CPR <= 8 ? 8 : 16);     // lanes per key
    constexpr int KPW = 64 / GK, KPI = 4 * KPW;                 // keys per wave / per block and iteration
    constexpr int U = 4, NPF = 6;                               // iterations in flight in the loops / requested at the layer top
    constexpr int VPR = sizeof(T) == 2 ? 2 : 1;                 // values per qkv / att / hid record
    constexpr int PREC = DH + 2;                                // records per attention partial: max, sum, o[DH]
    constexpr int QREC = DH / VPR;                              // records of one head's slice of q (k, v)
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ float qs[DH], kn[DH], vn[DH];
    __shared__ float ps[2048];
    __shared__ float red[12];
    __shared__ float osum[4][KPW][DH];
    __shared__ float pg[DEC_NSMAX][PREC];
    __shared__ float wn[DEC_NSMAX];
    __shared__ float ost[MR * 16];
    __shared__ unsigned long long ltab[ME_DEC_MAX_LAYERS * 15];            // the layer table (15 pointers per layer): one copy per block
    const int G = gridDim.x, bid = blockIdx.x;
    for (int i = threadIdx.x; i < a.n_layer * 15; i += 256) ltab[i] = reinterpret_cast<const unsigned long long*>(a.L)[i];
    __syncthreads();
    auto tab = [&](int layer, int field) __attribute__((always_inline)) -> const void* {      // uniform pointer out of LDS
        const unsigned long long v = ltab[layer * 15 + field];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    };
    const int Mr = a.Mr, d = a.d, di = a.di, H = a.H, ns = a.nsplit, M = a.M, Mc = a.Mc;
    const int KX = max(d, di);
    float* xs = dyn;                        // [MR][K of the running projection]: its input rows, T-rounded values
    float* xo = xs + MR * KX;               // [MR][d]: unrounded LayerNorm / embedding rows (residual of the next projection)
    float* sin_ = xo + MR * d;              // [MR][d]: polled pre-norm sums
    const unsigned epoch = a.ctl[0] + 1u;   // written by the previous launch on this stream
    unsigned* err = a.ctl + 2;
#ifdef ME_TOK_PROF
    unsigned long long* prof0 = (bid == 0 || bid == 100 || bid == 255) ? a.x_part + (size_t)TOK_PROF_OFF + (bid == 0 ? 0 : bid == 100 ? 512 : 1024) : nullptr;
#endif
    const int t = a.t_dev ? min(*a.t_dev, min(Mc, M) - 1) : a.t;

    // roles of the attention stage
    const int ngroups = Mr * H, rc = ngroups * (ns - 1);
    const bool role_c = bid < rc, role_n = bid >= rc && bid < rc + ngroups;
    const int mh = role_c ? bid / (ns - 1) : (role_n ? bid - rc : 0);
    const int si = role_c ? bid % (ns - 1) : ns - 1;
    const int m_own = mh / H, head = mh % H;
    int per = (t + ns - 2) / (ns - 1);
    per = (per + KPI - 1) / KPI * KPI;
    const int j0 = si * per, j1 = min(t, j0 + per);
    const bool has_keys = role_c && j0 < j1;
    const int nit = has_keys ? (j1 - j0 + KPI - 1) / KPI : 0;
    // virtual blocks of the projections (8 columns each, FFN_suf with the contraction split over the waves: 2)
    const int nvb_q = (3 * d + 7) / 8, nvb_o = (d + 3) / 4, nvb_1 = (di + 7) / 8, nvb_2 = a.ks2 ? (d + 1) / 2 : (d + 7) / 8;

    for (int layer = 0; layer <= a.n_layer; ++layer) {
        // the thread index is re-derived per layer behind an opaque move: everything computed from it (record addresses of every
        // poll, LDS addresses, owner indices -- some 300 registers of 64-bit pointers) is loop invariant, and hipcc otherwise
        // hoists all of it out of the layer loop and spills (710 spilled registers in the first build)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wid = tid >> 6;
        const int kslot = lane / GK, cl = lane % GK;
        const bool active = cl < CPR;
        const int cc = active ? cl : CPR - 1;
        const bool is_head = layer == a.n_layer;
#ifdef ME_TOK_PROF
        unsigned long long* prof = prof0 ? prof0 + 16 * layer : nullptr;
#endif
        TOK_STAMP(0);
        me_dec_layer Lc;                                                    // fields in the order of the struct
        {
            const int li = is_head ? layer - 1 : layer;
            Lc.Wqkv = tab(li, 0); Lc.bqkv = (const float*)tab(li, 1); Lc.Wo = tab(li, 2); Lc.bo = (const float*)tab(li, 3);
            Lc.W1 = tab(li, 4); Lc.b1 = (const float*)tab(li, 5); Lc.W2 = tab(li, 6); Lc.b2 = (const float*)tab(li, 7);
            Lc.ln1_g = (const float*)tab(li, 8); Lc.ln1_b = (const float*)tab(li, 9);
            Lc.E = tab(li, 12); Lc.kcache = const_cast<void*>(tab(li, 13)); Lc.vcache = const_cast<void*>(tab(li, 14));
        }
        const float* ln2_g_prev = (const float*)tab(layer > 0 ? layer - 1 : 0, 10);
        const float* ln2_b_prev = (const float*)tab(layer > 0 ? layer - 1 : 0, 11);
        const unsigned tag0 = epoch * 256u + 6u * (unsigned)layer;

        f32x4_t lg[4], lb[4];
        if (layer > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = min(lane * 4 + 256 * i, d - 4);
                lg[i] = *reinterpret_cast<const f32x4_t*>(ln2_g_prev + k);
                lb[i] = *reinterpret_cast<const f32x4_t*>(ln2_b_prev + k);
            }
        }
        if (is_head) {
            // ---- LayerNorm2 of the last layer -> vocabulary head (me_dec_ln_proj with ME_EPI_OUT_F32)
            const T* Wf = reinterpret_cast<const T*>(a.Wf);
            const int nvb = (a.V + 3) / 4;
            chunk16 wp[2][1];
            if (bid < nvb) gemv_prefetch<T, 1, false>(tid, wp, Wf, a.ldwf, a.V, d, bid);
            int n_, m_;
            bool own = bid < nvb && gemv_owner<MR, 1, false>(tid, a.V, Mr, bid, n_, m_);
            float bias_v = own && a.bf ? a.bf[n_] : 0.f;
            asm volatile("" ::: "memory");
            if (bid < nvb) {
                poll_records<8>(tid, Mr * d, tag0 - 1u, err, [&](int i) { return a.x_s2 + i; }, [&](int i, unsigned w) { sin_[i] = u2f(w); });
                __syncthreads();
                for (int m = wid; m < MR; m += 4) ln_row<T>(sin_ + m * d, d, lg, lb, a.eps, m < Mr, xo + m * d, xs + m * d, lane);
                __syncthreads();
                for (int vb = bid; vb < nvb; vb += G) {
                    if (vb != bid) {
                        gemv_prefetch<T, 1, false>(tid, wp, Wf, a.ldwf, a.V, d, vb);
                        own = gemv_owner<MR, 1, false>(tid, a.V, Mr, vb, n_, m_);
                        bias_v = own && a.bf ? a.bf[n_] : 0.f;
                    }
                    const float mine = gemv_run<T, MR, 1, false>(tid, wp, Wf, a.ldwf, a.V, d, vb, xs);
                    if (own) a.logits[(size_t)m_ * a.ld_logits + n_] = mine + bias_v;
                }
            }
            break;
        }

        // =================================================================== layer top: every request that depends on nothing of
        // this token -- the block's weight rows of ALL four projections (a few chunks per lane), their biases, the first NPF
        // iterations of the split's K | E | V rows.  They land while the block waits for s2; later polls then have the load
        // queue to themselves (loads return in order per wave: a poll issued behind cold weight rows waits for them)
        const T* Wqkv = reinterpret_cast<const T*>(Lc.Wqkv);
        const T* Wo = reinterpret_cast<const T*>(Lc.Wo);
        const T* W1 = reinterpret_cast<const T*>(Lc.W1);
        const T* W2 = reinterpret_cast<const T*>(Lc.W2);
        chunk16 wpq[2][2], wpo[2][1], wp1[2][2], wp2[2][2];
        int nq_, mq_, no_, mo_, n1_, m1_, n2_, m2_;
        bool ownq = false, owno = false, own1 = false, own2 = false;
        float biasq = 0.f, biaso = 0.f, bias1 = 0.f, bias2 = 0.f;
        if (bid < nvb_q) {
            gemv_prefetch<T, 2, false>(tid, wpq, Wqkv, d, 3 * d, d, bid);
            ownq = gemv_owner<MR, 2, false>(tid, 3 * d, Mr, bid, nq_, mq_);
            if (ownq) biasq = Lc.bqkv[nq_];
        }
        const T* kc = reinterpret_cast<const T*>(Lc.kcache) + (size_t)mh * Mc * DH;
        const T* vc = reinterpret_cast<const T*>(Lc.vcache) + (size_t)mh * Mc * DH;
        const T* er = reinterpret_cast<const T*>(Lc.E) + (size_t)(M - 1 - t) * DH;      // relative row of key j: E[M-1-(t-j)] = er + j * DH
        chunk16 pre[3 * NPF];
        float e_new = 0.f;
        auto issue_kv = [&]() __attribute__((always_inline)) {
            if (has_keys) {
#pragma unroll
                for (int it = 0; it < NPF; ++it) {
                    if (it < nit) {
                        const unsigned off = (unsigned)min(j0 + wid * KPW + kslot + KPI * it, j1 - 1) * DH + cc * CH;
                        pre[it] = ld_kv(kc + off);
                        pre[NPF + it] = ld_kv(er + off);
                        pre[2 * NPF + it] = ld_kv(vc + off);
                    }
                }
            }
            if (role_n && tid < DH) e_new = ET<T>::to_f(reinterpret_cast<const T*>(Lc.E)[(size_t)(M - 1) * DH + tid]);
        };
        auto issue_rest = [&]() __attribute__((always_inline)) {
            if (bid < nvb_o) {
                gemv_prefetch<T, 1, false>(tid, wpo, Wo, d, d, d, bid);
                owno = gemv_owner<MR, 1, false>(tid, d, Mr, bid, no_, mo_);
                if (owno) biaso = Lc.bo[no_];
            }
            if (bid < nvb_1) {
                gemv_prefetch<T, 2, false>(tid, wp1, W1, d, di, d, bid);
                own1 = gemv_owner<MR, 2, false>(tid, di, Mr, bid, n1_, m1_);
                if (own1) bias1 = Lc.b1[n1_];
            }
            if (bid < nvb_2) {
                if (a.ks2) { gemv_prefetch<T, 2, true>(tid, wp2, W2, di, d, di, bid); own2 = gemv_owner<MR, 2, true>(tid, d, Mr, bid, n2_, m2_); }
                else { gemv_prefetch<T, 2, false>(tid, wp2, W2, di, d, di, bid); own2 = gemv_owner<MR, 2, false>(tid, d, Mr, bid, n2_, m2_); }
                if (own2) bias2 = Lc.b2[n2_];
            }
        };
        asm volatile("" ::: "memory");
        TOK_STAMP(1);

        // =================================================================== S0 rows: LayerNorm2 of layer - 1 | embedding; q | k | v
        if (layer == 0) {
            issue_kv();
            // embedding rows (music_multi.py:89-101 for one position; the arithmetic of dec_gemv_kernel<PRO_EMBED>)
            const int dc = a.dc, de = d - dc;
            const float sq = sqrtf((float)de);
            for (int i = tid; i < MR * (d / 4); i += 256) {
                const int m = i / (d / 4), k = (i % (d / 4)) * 4;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f}, r = v;
                if (m < Mr) {
                    const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(a.pe + (size_t)t * d + k);
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
                }
                *reinterpret_cast<f32x4_t*>(&xo[m * d + k]) = v;
                *reinterpret_cast<f32x4_t*>(&xs[m * d + k]) = r;
            }
        } else {
            poll_records<8>(tid, Mr * d, tag0 - 1u, err, [&](int i) { return a.x_s2 + i; }, [&](int i, unsigned w) { sin_[i] = u2f(w); });
            issue_kv();                                                     // lands under LayerNorm + q | k | v + the q exchange
            __syncthreads();
            TOK_STAMP(2);
            for (int m = wid; m < MR; m += 4) ln_row<T>(sin_ + m * d, d, lg, lb, a.eps, m < Mr, xo + m * d, xs + m * d, lane);
        }
        // LayerNorm1's scale / shift: requested now, used after S2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = min(lane * 4 + 256 * i, d - 4);
            lg[i] = *reinterpret_cast<const f32x4_t*>(Lc.ln1_g + k);
            lb[i] = *reinterpret_cast<const f32x4_t*>(Lc.ln1_b + k);
        }
        __syncthreads();
        TOK_STAMP(3);
        for (int vb = bid; vb < nvb_q; vb += G) {
            if (vb != bid) {
                gemv_prefetch<T, 2, false>(tid, wpq, Wqkv, d, 3 * d, d, vb);
                ownq = gemv_owner<MR, 2, false>(tid, 3 * d, Mr, vb, nq_, mq_);
                biasq = ownq ? Lc.bqkv[nq_] : 0.f;
            }
            const float mine = gemv_run<T, MR, 2, false>(tid, wpq, Wqkv, d, 3 * d, d, vb, xs);
            const float v = round_to<T>(mine + biasq);
            if constexpr (VPR == 2) {                       // the two columns of a wave's pair sit in different lanes: pair them through LDS
                if (ownq) ost[mq_ * 8 + (nq_ - vb * 8)] = v;
                __syncthreads();
                if (tid < MR * 4) {
                    const int m = tid / 4, p = tid % 4, n = vb * 8 + 2 * p;
                    if (m < Mr && n < 3 * d) st_rec(a.x_qkv + ((size_t)m * 3 * d + n) / 2, pack2<T>(ost[m * 8 + 2 * p], ost[m * 8 + 2 * p + 1]), tag0);
                }
                __syncthreads();
            } else {
                if (ownq) st_rec(a.x_qkv + (size_t)mq_ * 3 * d + nq_, f2u(v), tag0);
            }
        }
        issue_rest();                                                       // Wo / FFN rows of the block: land under the attention stage
        asm volatile("" ::: "memory");
        TOK_STAMP(4);

        // =================================================================== S1 attention partial of the block's role
        if (role_c || role_n) {
            unsigned long long* prec = a.x_part + ((size_t)mh * ns + si) * PREC;
            // the head's slice of q (new-key member: of q, k and v) of sequence m_own
            const int nsl = role_n ? 3 : 1;
            poll_records<1>(tid, nsl * QREC, tag0, err,
                            [&](int i) { return a.x_qkv + ((size_t)m_own * 3 * d + (size_t)(i / QREC) * d + head * DH) / VPR + i % QREC; },
                            [&](int i, unsigned w) {
                                float* dst = i / QREC == 0 ? qs : (i / QREC == 1 ? kn : vn);
                                const float mul = i / QREC == 0 ? a.scale : 1.f;
                                const int j = (i % QREC) * VPR;
                                if constexpr (VPR == 2) { dst[j] = lo16_f<T>(w) * mul; dst[j + 1] = hi16_f<T>(w) * mul; }
                                else dst[j] = u2f(w) * mul;
                            });
            __syncthreads();
            TOK_STAMP(5);
            if (role_n) {                                                   // the new key: cache append; partial (s_t, 1, v_t)
                if (tid < DH) {
                    reinterpret_cast<T*>(Lc.kcache)[((size_t)mh * Mc + t) * DH + tid] = ET<T>::from_f(kn[tid]);
                    reinterpret_cast<T*>(Lc.vcache)[((size_t)mh * Mc + t) * DH + tid] = ET<T>::from_f(vn[tid]);
                    st_rec(prec + 2 + tid, f2u(vn[tid]), tag0 + 1u);
                }
                if (wid == 0) {
                    float sp = 0.f;
                    for (int i = lane; i < DH; i += 64) sp = fmaf(qs[i], kn[i] + e_new, sp);       // DH <= 64: lane i holds E[M-1][i]
                    sp = wave_sum(sp);
                    if (lane == 0) { st_rec(prec, f2u(sp), tag0 + 1u); st_rec(prec + 1, f2u(1.f), tag0 + 1u); }
                }
            } else if (!has_keys) {                                         // empty split (short contexts)
                if (tid < PREC) st_rec(prec + tid, f2u(tid == 0 ? -INFINITY : 0.f), tag0 + 1u);
            } else {
                float qv[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) qv[i] = active ? qs[cc * CH + i] : 0.f;
                float mx = -INFINITY;
                auto score = [&](const chunk16& kk, const chunk16& ee, int j) __attribute__((always_inline)) {
                    const T* ke = reinterpret_cast<const T*>(&kk);
                    const T* ev = reinterpret_cast<const T*>(&ee);
                    float s_ = 0.f;
#pragma unroll
                    for (int i = 0; i < CH; ++i) s_ = fmaf(qv[i], ET<T>::to_f(ke[i]) + ET<T>::to_f(ev[i]), s_);
                    s_ = group_sum<GK>(s_);
                    if (j < j1) {
                        if (cl == 0) ps[j - j0] = s_;
                        mx = fmaxf(mx, s_);
                    }
                };
                const int jl = j0 + wid * KPW + kslot;                      // this lane's first key; then every KPI-th
#pragma unroll
                for (int it = 0; it < NPF; ++it) if (it < nit) score(pre[it], pre[NPF + it], jl + KPI * it);
                for (int it0 = NPF; it0 < nit; it0 += U) {
                    chunk16 kk[U], ee[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int jc = min(jl + KPI * (it0 + u), j1 - 1);
                        kk[u] = ld_kv(kc + (size_t)jc * DH + cc * CH);
                        ee[u] = ld_kv(er + (size_t)jc * DH + cc * CH);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) score(kk[u], ee[u], jl + KPI * (it0 + u));
                }
                mx = wave_max(mx);
                if (lane == 0) red[wid] = mx;
                __syncthreads();
                mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                const float m_safe = mx == -INFINITY ? 0.f : mx;
                float o[CH], lsum = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) o[i] = 0.f;
                auto pvacc = [&](const chunk16& vv, int j) __attribute__((always_inline)) {
                    const float p = j < j1 ? ET<T>::fexp(ps[j - j0] - m_safe) : 0.f;
                    const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
                    for (int i = 0; i < CH; ++i) o[i] = fmaf(p, ET<T>::to_f(ve[i]), o[i]);
                    if (cl == 0) lsum += p;
                };
#pragma unroll
                for (int it = 0; it < NPF; ++it) if (it < nit) pvacc(pre[2 * NPF + it], jl + KPI * it);
                for (int it0 = NPF; it0 < nit; it0 += U) {
                    chunk16 vv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) vv[u] = ld_kv(vc + (size_t)min(jl + KPI * (it0 + u), j1 - 1) * DH + cc * CH);
#pragma unroll
                    for (int u = 0; u < U; ++u) pvacc(vv[u], jl + KPI * (it0 + u));
                }
                if (active) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) osum[wid][kslot][cl * CH + i] = o[i];
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
                    st_rec(prec + 2 + tid, f2u(s_), tag0 + 1u);
                }
                if (tid == 0) { st_rec(prec, f2u(mx), tag0 + 1u); st_rec(prec + 1, f2u(red[4] + red[5] + red[6] + red[7]), tag0 + 1u); }
            }

            // =============================================================== S1b combine: dh / ns outputs of the head per member
            TOK_STAMP(6);
            poll_records<4>(tid, ns * PREC, tag0 + 1u, err, [&](int i) { return a.x_part + (size_t)mh * ns * PREC + i; },
                            [&](int i, unsigned w) { pg[i / PREC][i % PREC] = u2f(w); });
            __syncthreads();
            TOK_STAMP(7);
            if (tid == 0) {                                                // the combine prologue of dec_gemv_kernel<PRO_ATTN>
                float mxs = -INFINITY;
                for (int s_ = 0; s_ < ns; ++s_) mxs = fmaxf(mxs, pg[s_][0]);
                const float msafe = mxs == -INFINITY ? 0.f : mxs;
                float l = 0.f, w[DEC_NSMAX];
#pragma unroll
                for (int s_ = 0; s_ < DEC_NSMAX; ++s_) {
                    w[s_] = s_ < ns ? ET<T>::fexp(pg[min(s_, ns - 1)][0] - msafe) : 0.f;
                    l = mul_add_unfused(w[s_], pg[min(s_, ns - 1)][1], l);
                }
                const float inv = 1.f / l;
#pragma unroll
                for (int s_ = 0; s_ < DEC_NSMAX; ++s_) wn[s_] = w[s_] * inv;
            }
            __syncthreads();
            const int nper = DH / ns;                                       // outputs per member
            if (tid < nper / VPR) {
                float val[VPR];
#pragma unroll
                for (int e = 0; e < VPR; ++e) {
                    const int k = si * nper + tid * VPR + e;
                    float v = 0.f;
#pragma unroll
                    for (int s_ = 0; s_ < DEC_NSMAX; ++s_) v = s_ < ns ? fmaf(wn[s_], pg[min(s_, ns - 1)][2 + k], v) : v;
                    val[e] = round_to<T>(v);
                }
                const int k0 = m_own * d + head * DH + si * nper + tid * VPR;
                if constexpr (VPR == 2) st_rec(a.x_att + k0 / 2, pack2<T>(val[0], val[1]), tag0 + 2u);
                else st_rec(a.x_att + k0, f2u(val[0]), tag0 + 2u);
            }
        }

        // =================================================================== S2 Wo + bias + residual(x)                -> s1
        TOK_STAMP(8);
        if (bid < nvb_o) {
            for (int i = Mr * d + tid; i < MR * d; i += 256) xs[i] = 0.f;                 // rows beyond the batch
            if constexpr (VPR == 2)
                poll_records<8>(tid, Mr * d / 2, tag0 + 2u, err, [&](int i) { return a.x_att + i; },
                                [&](int i, unsigned w) { xs[2 * i] = lo16_f<T>(w); xs[2 * i + 1] = hi16_f<T>(w); });
            else
                poll_records<8>(tid, Mr * d, tag0 + 2u, err, [&](int i) { return a.x_att + i; }, [&](int i, unsigned w) { xs[i] = u2f(w); });
            __syncthreads();
            TOK_STAMP(9);
            for (int vb = bid; vb < nvb_o; vb += G) {
                if (vb != bid) {
                    gemv_prefetch<T, 1, false>(tid, wpo, Wo, d, d, d, vb);
                    owno = gemv_owner<MR, 1, false>(tid, d, Mr, vb, no_, mo_);
                    biaso = owno ? Lc.bo[no_] : 0.f;
                }
                const float mine = gemv_run<T, MR, 1, false>(tid, wpo, Wo, d, d, d, vb, xs);
                if (owno) st_rec(a.x_s1 + mo_ * d + no_, f2u(xo[mo_ * d + no_] + (mine + biaso)), tag0 + 3u);
            }
        }

        // =================================================================== S3 LayerNorm1 -> FFN_pre + bias + ReLU    -> hid
        TOK_STAMP(10);
        __syncthreads();                                                    // S2 has read xo (the residual): LayerNorm1 overwrites it
        poll_records<8>(tid, Mr * d, tag0 + 3u, err, [&](int i) { return a.x_s1 + i; }, [&](int i, unsigned w) { sin_[i] = u2f(w); });
        __syncthreads();
        TOK_STAMP(11);
        for (int m = wid; m < MR; m += 4) ln_row<T>(sin_ + m * d, d, lg, lb, a.eps, m < Mr, xo + m * d, xs + m * d, lane);
        __syncthreads();
        TOK_STAMP(12);
        for (int vb = bid; vb < nvb_1; vb += G) {
            if (vb != bid) {
                gemv_prefetch<T, 2, false>(tid, wp1, W1, d, di, d, vb);
                own1 = gemv_owner<MR, 2, false>(tid, di, Mr, vb, n1_, m1_);
                bias1 = own1 ? Lc.b1[n1_] : 0.f;
            }
            const float mine = gemv_run<T, MR, 2, false>(tid, wp1, W1, d, di, d, vb, xs);
            const float v = round_to<T>(fmaxf(mine + bias1, 0.f));
            if constexpr (VPR == 2) {
                if (own1) ost[m1_ * 8 + (n1_ - vb * 8)] = v;
                __syncthreads();
                if (tid < MR * 4) {
                    const int m = tid / 4, p = tid % 4, n = vb * 8 + 2 * p;
                    if (m < Mr && n < di) st_rec(a.x_hid + ((size_t)m * di + n) / 2, pack2<T>(ost[m * 8 + 2 * p], n + 1 < di ? ost[m * 8 + 2 * p + 1] : 0.f), tag0 + 4u);
                }
                __syncthreads();
            } else {
                if (own1) st_rec(a.x_hid + (size_t)m1_ * di + n1_, f2u(v), tag0 + 4u);
            }
        }

        // =================================================================== S4 FFN_suf + bias + residual(LayerNorm1 row) -> s2
        TOK_STAMP(13);
        if (bid < nvb_2) {
            float res_v = own2 ? xo[m2_ * d + n2_] : 0.f;
            for (int i = Mr * di + tid; i < MR * di; i += 256) xs[i] = 0.f;               // rows beyond the batch
            if constexpr (VPR == 2)
                poll_records<16>(tid, Mr * di / 2, tag0 + 4u, err, [&](int i) { return a.x_hid + i; },
                                 [&](int i, unsigned w) { xs[2 * i] = lo16_f<T>(w); xs[2 * i + 1] = hi16_f<T>(w); });
            else
                poll_records<16>(tid, Mr * di, tag0 + 4u, err, [&](int i) { return a.x_hid + i; }, [&](int i, unsigned w) { xs[i] = u2f(w); });
            __syncthreads();
            TOK_STAMP(14);
            auto stage4 = [&](auto ks_tag) __attribute__((always_inline)) {
                constexpr bool KS = decltype(ks_tag)::value;
                for (int vb = bid; vb < nvb_2; vb += G) {
                    if (vb != bid) {
                        __syncthreads();                                    // KS: the previous pass's quarters (in ost) have been summed
                        gemv_prefetch<T, 2, KS>(tid, wp2, W2, di, d, di, vb);
                        own2 = gemv_owner<MR, 2, KS>(tid, d, Mr, vb, n2_, m2_);
                        bias2 = own2 ? Lc.b2[n2_] : 0.f;
                        res_v = own2 ? xo[m2_ * d + n2_] : 0.f;
                    }
                    const float mine = gemv_run<T, MR, 2, KS>(tid, wp2, W2, di, d, di, vb, xs, ost);
                    if (own2) st_rec(a.x_s2 + m2_ * d + n2_, f2u(res_v + (mine + bias2)), tag0 + 5u);
                }
            };
            if (a.ks2) stage4(std::true_type{}); else stage4(std::false_type{});
        }
        __syncthreads();
        TOK_STAMP(15);
    }

    // ---- the last block to finish advances the epoch
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done + 1 == (unsigned)G) {
            __hip_atomic_store(a.ctl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int TOK_MR = ME_DEC_TOKEN_ROWS;

inline size_t tok_lds(int d, int di) { return ((size_t)TOK_MR * (d > di ? d : di) + 2 * (size_t)TOK_MR * d) * sizeof(float); }

template <typename T, int DH>
int tok_blocks(int d, int di) {
    // per device and instantiation: the attribute and the occupancy query are host-side runtime calls (not per token)
    static int cached[16] = {0};
    static size_t cached_lds[16] = {0};
    int dev = 0, ncu = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const size_t lds = tok_lds(d, di);
    if (dev >= 0 && dev < 16 && cached[dev] > 0 && cached_lds[dev] == lds) return cached[dev];
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipFuncSetAttribute((const void*)dec_token_kernel<T, DH, TOK_MR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dec_token_kernel<T, DH, TOK_MR>, 256, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const int r = occ > 0 ? ncu : 0;                    // one block per CU: every block is resident at once
    if (dev >= 0 && dev < 16) { cached[dev] = r; cached_lds[dev] = lds; }
    return r;
}

template <typename T, int DH>
int tok_launch(const TokArgs& a, int blocks, hipStream_t st) {
    const size_t lds = tok_lds(a.d, a.di);
    if (lds > 128 * 1024) return ME_ERR_BAD_SHAPE;                 // + ~20 KB of static LDS
    const int maxb = tok_blocks<T, DH>(a.d, a.di);
    if (maxb <= 0) return ME_ERR_LAUNCH;
    if (blocks <= 0 || blocks > maxb) blocks = maxb;
    if (a.Mr * a.H * a.nsplit > blocks) return ME_ERR_BAD_SHAPE;
    dec_token_kernel<T, DH, TOK_MR><<<blocks, 256, lds, st>>>(a);
    return me_launch_status();
}

}  // namespace

#define ME_TOK_T(CALL)                                          \
    if (dtype == ME_F32) { typedef float T; return CALL; }      \
    if (dtype == ME_BF16) { typedef bf16_t T; return CALL; }    \
    if (dtype == ME_F16) { typedef f16_t T; return CALL; }      \
    return ME_ERR_BAD_DTYPE;

extern "C" {

int me_dec_token_blocks(int dh, int d, int d_inner, int dtype) {
    me_clear_error();
    if (d <= 0 || d_inner <= 0 || d > 1024 || d % 8) return 0;
#define ME_TOK_BLOCKS_CASE(DHV) if (dh == DHV) { ME_TOK_T((tok_blocks<T, DHV>(d, d_inner))) }
    ME_TOK_BLOCKS_CASE(64)
    ME_TOK_BLOCKS_CASE(48)
    ME_TOK_BLOCKS_CASE(32)
#undef ME_TOK_BLOCKS_CASE
    return 0;
}

int me_dec_token(const int64_t* tokens, const float* cond, const float* emb, const float* cw, const float* cb, const float* pe,
                 int d_cond, const me_dec_layer* layers, int n_layer, const void* Wf, int ldwf, const float* bf, int V,
                 float* logits, int ld_logits, void* ws, size_t ws_bytes, int nsplit, int Mr, int d, int d_inner, int H, int dh,
                 int M, int Mc, int t, const int32_t* t_dev, float eps, int blocks, int dtype, void* stream) {
    me_clear_error();
    if (!tokens || !emb || !pe || !layers || !Wf || !logits || !ws) return ME_ERR_NULL;
    if (d_cond > 0 && (!cond || !cw || !cb)) return ME_ERR_NULL;
    if (n_layer <= 0 || n_layer > ME_DEC_MAX_LAYERS || (reinterpret_cast<uintptr_t>(layers) & 7) || Mr <= 0 || Mr > TOK_MR || H <= 0 || dh <= 0 || H * dh != d || d > 1024 || d % 8 ||
        d_inner <= 0 || d_inner % 8 || V <= 0 || Mc <= 0 || M <= 0 || d_cond >= d || (d_cond > 0 && d_cond % 4) || ldwf % 8 || ld_logits < V)
        return ME_ERR_BAD_SHAPE;
    if (!(nsplit == 2 || nsplit == 4 || nsplit == 8) || dh % nsplit || (dh / nsplit) % 2) return ME_ERR_BAD_SHAPE;
    if (!t_dev && (t < 0 || t >= Mc || t >= M)) return ME_ERR_BAD_SHAPE;
    if ((Mc + nsplit - 2) / (nsplit - 1) + 64 > 2048 + 64) return ME_ERR_BAD_SHAPE;       // score buffer: 2048 keys per split
    if (ws_bytes < me_dec_token_ws_bytes(d, d_inner) || (reinterpret_cast<uintptr_t>(ws) & 255)) return ME_ERR_WORKSPACE;
    if (!aligned16(emb) || !aligned16(pe) || !aligned16(Wf)) return ME_ERR_ALIGNMENT;
    TokArgs a = {};
    a.tokens = tokens; a.cond = cond; a.emb = emb; a.cw = cw; a.cb = cb; a.pe = pe; a.dc = d_cond > 0 ? d_cond : 0;
    a.L = layers;
    a.n_layer = n_layer; a.Wf = Wf; a.ldwf = ldwf; a.bf = bf; a.V = V; a.logits = logits; a.ld_logits = ld_logits;
    a.Mr = Mr; a.d = d; a.di = d_inner; a.H = H; a.nsplit = nsplit; a.M = M; a.Mc = Mc; a.t = t; a.t_dev = t_dev; a.eps = eps;
    a.scale = 1.f / sqrtf((float)dh);
    a.ctl = reinterpret_cast<unsigned*>(ws);
    unsigned long long* r = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + 256);
    a.x_s2 = r; r += (size_t)TOK_MR * d;
    a.x_s1 = r; r += (size_t)TOK_MR * d;
    a.x_att = r; r += (size_t)TOK_MR * d;
    a.x_qkv = r; r += (size_t)TOK_MR * 3 * d;
    a.x_hid = r; r += (size_t)TOK_MR * d_inner;
    a.x_part = r;
    const int ch = dtype == ME_F32 ? 4 : 8;
    a.ks2 = d_inner / ch >= 256;
    hipStream_t st = (hipStream_t)stream;
#define ME_TOK_CASE(DHV) if (dh == DHV) { ME_TOK_T((tok_launch<T, DHV>(a, blocks, st))) }
    ME_TOK_CASE(64)
    ME_TOK_CASE(48)
    ME_TOK_CASE(32)
#undef ME_TOK_CASE
    return ME_ERR_BAD_SHAPE;
}

}  // extern "C"
