// Relative global attention (Music-Transformer RGA) for gfx950.
//
//   logits[q,key] = ( Q[q].K[key] + Q[q].E[M-1-(q-key)] ) / sqrt(dh),  key <= q, key not pad
//
// Forward is flash style: the L x L scores, the relative term and the float masks of the
// reference (music_multi.py:215-231) are never materialised.  32 x 32 (q x key) tiles, one
// wavefront per 32 query rows, MFMA macro-atoms from me_common.h.
//
// Relative term.  For a tile (q0, k0) the 63 rows E[e_lo .. e_lo+62], e_lo = M-32-q0+k0 (a
// multiple of 32) are needed: two aligned 32-row blocks "lo", "hi".  G = Q.E_blk^T is an
// ordinary MFMA product; the Toeplitz skew  Srel[a][b] = G[a][31-a+b]  is one trip through a
// wave-private LDS ring (written in accumulator layout, read back with a per-lane shifted
// address -- conflict free both ways).  hi of step t == lo of step t+1, so a query-owned wave
// computes one new block per step.
//
// Layout trick.  Query-owned kernels compute the TRANSPOSED tile S^T[key][q] = mfma(K, Q): in
// the accumulator layout a lane owns one query column and 16 key rows, so softmax statistics
// are lane-local (+1 half-wave exchange), P^T packs straight into the B operand of
// O^T[d][q] += V^T[d][key] P^T[key][q] (the V^T fragment is read with the accumulator's own
// k-map), and the per-row rescale of O is a per-lane scalar.
//
// Contraction-over-rows operands (V^T, K^T, Q^T, dO^T fragments) are read straight from NATURAL
// LDS tiles with the hardware transpose read (frag_load_tr, me_common.h): no transposed copies in
// memory, no LDS scatter.
//
// Backward.  The query-owned kernel recomputes P, forms dS, accumulates dQ (key part and
// relative part) and MATERIALISES, for the current layer only, P^T, dS^T ([bh][key][q]) and
// the skewed dG^T ([bh][c][q], c = E-row - (M - Lp)) in the compute type.  dK/dV and dE are
// then plain streaming tile products (contraction over q) that read those once:
//     dV = P^T dO,  dK = dS^T Q            (rga_bwd_kv_kernel, key-owned, HBM-bound)
//     dE[e] += sum_bh dG^T[bh][e] Q^T      (rga_bwd_e_kernel, E-row-owned, HBM-bound)
// Only tiles on/below the diagonal are ever written or read; the workspaces must be
// zero-initialised once (rows q >= L and the unreachable corner stay zero).
#include "me_common.h"
#include <type_traits>

#ifndef ME_ABL
#define ME_ABL 0
#endif

namespace {

constexpr int LDG = 36;   // G ring row (floats): 32 + 4 -> conflict-free b128 writes, b32 skew reads
constexpr int LDG2 = 68;  // forward G ring row: 64-column ring + 4

template <typename T, int DH> struct ACfg {
    static constexpr int CH = ET<T>::CH;
    static constexpr int KA = DH / 16;       // contraction atoms over the head dim
    static constexpr int DB = (DH + 31) / 32;   // 32-wide blocks of the head dim; DH = 48: the upper half of block 1 is padding
                                                // (guarded global loads / stores; MFMA garbage there only reaches discarded outputs)
    static constexpr int LDN = DH + CH;      // natural [row][DH] tile row (elements), read with 16-byte fragment loads
    // tile only read through transpose reads: a row stride of 192 B (mod 256) puts the 4 x 2 row segments of a
    // 32-lane half on disjoint banks
    static constexpr int LDV = sizeof(T) == 2 ? (DH > 32 ? 96 : 32) : DH + 4;
    // packed relative table (me_rga_pack_rel): per 32-row block KA fragment images of E rows, then 2 DB images of E^T
    static constexpr int PK_B = KA * 512;                 // element offset of the E^T images inside a block
    static constexpr int PK = (KA + 2 * DB) * 512;        // elements per packed block
};

// ---- generic ROWS x COLS chunk tiles (16-byte chunks, lanes walk a row) ----------------
template <typename T, int ROWS, int COLS> struct TileT {
    static constexpr int CH = ET<T>::CH;
    static constexpr int CPR = COLS / CH;
    static constexpr int NCH = ROWS * CPR;
    static constexpr int NPT = (NCH + 255) / 256;
};
template <typename T, int ROWS, int COLS>
ME_DEV void tile_gload(chunk16* r, const T* origin, size_t ld, int rows_valid, int tid) {
    using TT = TileT<T, ROWS, COLS>;
#pragma unroll
    for (int i = 0; i < TT::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < TT::NCH) {
            const int row = c / TT::CPR, cc = (c % TT::CPR) * TT::CH;
            r[i] = row < rows_valid ? ld_chunk(origin + (size_t)row * ld + cc) : zero_chunk();
        }
    }
}
// all ROWS rows in bounds: no predicate, so the loads stay in the caller's basic block (exact vmcnt bookkeeping)
template <typename T, int ROWS, int COLS>
ME_DEV void tile_gload_full(chunk16* r, const T* origin, size_t ld, int tid) {
    using TT = TileT<T, ROWS, COLS>;
#pragma unroll
    for (int i = 0; i < TT::NPT; ++i) {
        const int c = (TT::NCH % 256 == 0) ? tid + i * 256 : min(tid + i * 256, TT::NCH - 1);   // spare threads re-load the last chunk
        r[i] = ld_chunk(origin + (size_t)(c / TT::CPR) * ld + (c % TT::CPR) * TT::CH);
    }
}
template <typename T, int ROWS, int COLS, int LDS_LD>
ME_DEV void tile_sstore(const chunk16* r, T* S, int tid) {
    using TT = TileT<T, ROWS, COLS>;
#pragma unroll
    for (int i = 0; i < TT::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < TT::NCH) {
            T* dst = &S[(c / TT::CPR) * LDS_LD + (c % TT::CPR) * TT::CH];
            if constexpr ((LDS_LD * sizeof(T)) % 16 == 0) {
                st_chunk(dst, r[i]);
            } else {                              // 8-byte aligned rows (bf16, LDT = 36): two halves
                const uint64_t* h = reinterpret_cast<const uint64_t*>(&r[i]);
                reinterpret_cast<uint64_t*>(dst)[0] = h[0];
                reinterpret_cast<uint64_t*>(dst)[1] = h[1];
            }
        }
    }
}

// fragments of one row (8 contiguous elements per atom) straight from global memory
template <typename T, int DH>
ME_DEV void row_frags(Frag<T>* f, const T* rowptr, bool valid, int h) {
#pragma unroll
    for (int kk = 0; kk < DH / 16; ++kk) {
        if (valid) frag_load(f[kk], rowptr + kk * 16 + h * 8);
        else frag_zero(f[kk]);
    }
}

// P^T / dS^T workspace of one (b, head): element offset of the 32-query row segment (key, query tile qt).
// Layout 2 (default): [key tile][query tile][32 key][32 q], every tile 32 x 32 contiguous; 0: plain [key][q] rows;
// 1: query-tile-major [Lp/32 qt][Lp key][32 q].  Measured in the train step (q / kv kernels, us): 0: 319 / 188, 1: 303 / 178, 2: 298 / 173.
#ifndef ME_WS_LAYOUT
#define ME_WS_LAYOUT 2
#endif
ME_DEV size_t ws_row(int key, int qt, int Lp) {
    if (ME_WS_LAYOUT == 0) return (size_t)key * Lp + qt * 32;
    if (ME_WS_LAYOUT == 1) return ((size_t)qt * Lp + key) * 32;
    return (((size_t)(key >> 5) * (Lp >> 5) + qt) * 32 + (key & 31)) * 32;
}

// v_exp_f32 without the denormal-range fix-up of exp2f (arguments here are <= 0: tiny results may flush to 0)
ME_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename T> ME_DEV void st4(T* p, float a, float b, float c, float d);
template <> ME_DEV void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    bf16x4_t v; v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
    *reinterpret_cast<bf16x4_t*>(p) = v;
}
template <> ME_DEV void st4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4_t*>(p) = (f32x4_t){a, b, c, d};
}

// =====================================================================================
// forward
// =====================================================================================
// Per key tile and wave: 4 (K.Q) + 4 (new E block . Q) + 4 (V^T.P^T) macro-atoms.  K / V^T tiles
// are double buffered in LDS (one barrier per step; the next tile's global loads are in flight
// during the whole step), the E fragments of the NEXT step's new block are fetched into
// registers right after the current block's MFMAs were issued, the pad flags travel with the
// tile, and tiles that need no masking skip all per-element predicates.  exp2-domain softmax.
// CAUSAL = false: the bidirectional variant of MusicRegression (models/music_regression.py:79, mask = None): every key
// is attended; the relative term exists only for key <= q (the reference's _qe_masking + _skewing leave exact zeros
// above the diagonal), nothing is masked but keys >= L and padded keys.
template <typename T, int DH, bool CAUSAL = true>
__global__ __launch_bounds__(256, 3) void rga_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ Epk, const uint8_t* __restrict__ key_pad,
                                                      T* __restrict__ out, float* __restrict__ lse, int B, int L, int H, int M,
                                                      float scale) {
    using C = ACfg<T, DH>;
    __shared__ __attribute__((aligned(16))) T Ks[2][32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Vs[2][32 * C::LDN];      // natural V tile, transpose-read (144-byte rows: 2-way conflicts, but 3 blocks/CU)
    __shared__ __attribute__((aligned(16))) float Gs[4][32 * LDG2];        // per wave: [q][64-column ring]
    __shared__ uint32_t Ps[2][32];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int nqb = (L + 127) / 128;
    const int bh = blockIdx.x % (B * H), qb = nqb - 1 - (int)(blockIdx.x / (B * H));   // heavy q-blocks first
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const T* qb_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* kb_ = qb_ + dm;
    const T* vb_ = qb_ + 2 * dm;
    const int q0 = qb * 128 + wid * 32;
    const int q = q0 + a;
    const bool wave_on = q0 < L;
    const int nkt = CAUSAL ? min((L + 31) / 32, qb * 4 + 4) : (L + 31) / 32;
    const int my_last_kt = qb * 4 + wid;            // diagonal tile of this wave
    const float c2 = scale * 1.4426950408889634f;   // logits are kept in log2 units

    Frag<T> qf[C::KA];
    row_frags<T, DH>(qf, qb_ + (size_t)q * ldq, wave_on && q < L, h);

    f32x16_t o[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) acc_zero(o[i]);
    float m_run = -INFINITY, l_run = 0.f;

    chunk16 rk[TileT<T, 32, DH>::NPT], rv[TileT<T, 32, DH>::NPT];
    uint32_t rp = 0, rpm = 0;
    // pad flags: always one byte load per thread (a valid dummy row when there is no mask), masked when stored:
    // a load under a branch, or an early use, would make the later vmcnt waits conservative
    const uint8_t* kp_ = key_pad ? key_pad + (size_t)b * L : reinterpret_cast<const uint8_t*>(qkv);
    const uint32_t kp_on = key_pad ? 0xffu : 0u;
    auto gload = [&](int kt) __attribute__((always_inline)) {
        tile_gload<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
        tile_gload<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
        rp = kp_[min(kt * 32 + (tid & 31), L - 1)];
        rpm = kt * 32 + (tid & 31) < L ? kp_on : 0u;
    };
    auto gload_full = [&](int kt) __attribute__((always_inline)) {       // tile kt entirely below L
        tile_gload_full<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, tid);
        tile_gload_full<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, tid);
        rp = kp_[kt * 32 + (tid & 31)];
        rpm = kp_on;
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        tile_sstore<T, 32, DH, C::LDN>(rk, Ks[buf], tid);
        tile_sstore<T, 32, DH, C::LDN>(rv, Vs[buf], tid);
        if (tid < 32) Ps[buf][tid] = rp & rpm;
    };
    auto g_block = [&](const Frag<T>* ef, int eb) {     // G^T[m][q] = E[eb*32+m] . Q[q] -> ring slot eb&1
        f32x16_t g; acc_zero(g);
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) mma32(g, ef[kk], qf[kk]);
        float* gs = &Gs[wid][a * LDG2 + (eb & 1) * 32];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4_t*>(gs + 8 * gq + 4 * h) = (f32x4_t){g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
    };

    // E rows of block eb as A-operand fragments: one contiguous 1 KB image per contraction atom (me_rga_pack_rel);
    // fragment-shaped loads from the natural [M][dh] table touch 32 cache lines per instruction
    auto e_frags = [&](Frag<T>* f, int eb) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) frag_load(f[kk], Epk + (size_t)eb * C::PK + (kk * 64 + lane) * 8);
    };
    gload(0);
    const int eb0 = (M - 32 - q0) >> 5;
    Frag<T> ef[C::KA];
    if (wave_on) {
        e_frags(ef, eb0);
        g_block(ef, eb0);
        if (my_last_kt > 0) e_frags(ef, eb0 + 1);
    }
    sstore(0);
    if (nkt > 1) gload(1);
    __syncthreads();

    // One key tile.  MAIN = every wave of the block is strictly above its diagonal tile and tiles kt + 1, kt + 2
    // lie entirely below L: no branch encloses a global load, so the s_waitcnt bookkeeping stays exact (a
    // conservative vmcnt(0) at the top of the step exposes the K / V prefetch issued just before the barrier).
    auto step = [&](int kt, auto main_tag) __attribute__((always_inline)) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const int buf = kt & 1;
        if (MAIN || (wave_on && (!CAUSAL || kt <= my_last_kt))) {
            const int k0 = kt * 32;
            const bool diag = !MAIN && kt == my_last_kt;
            const bool upper = !CAUSAL && !MAIN && kt > my_last_kt;      // bidirectional only: tile above the diagonal, no relative term
            const int eb_lo = eb0 + kt;
            if constexpr (MAIN) {
                g_block(ef, eb_lo + 1);
                e_frags(ef, min(eb_lo + 2, (M >> 5) - 1));          // clamped: unused past the diagonal
            } else if (!diag && !upper) {
                g_block(ef, eb_lo + 1);
                if (kt + 1 < my_last_kt) e_frags(ef, eb_lo + 2);
            }
            f32x16_t s; acc_zero(s);
#pragma unroll
            for (int kk = 0; kk < C::KA; ++kk) {
                Frag<T> kf; frag_load(kf, &Ks[buf][a * C::LDN + kk * 16 + h * 8]);
                mma32(s, kf, qf[kk]);                       // S^T[key][q]
            }
            uint32_t pbits = 0;
            if (key_pad) pbits = __builtin_amdgcn_readfirstlane((uint32_t)__ballot(lane < 32 && Ps[buf][a] != 0));
            // band element m (0..62) of this tile sits at ring column ((eb_lo & 1) * 32 + m) & 63
            const float* grow = &Gs[wid][a * LDG2];
            const int t0 = (eb_lo & 1) * 32 + 31 - a + 4 * h;
            float mt = -INFINITY;
            if (!diag && !upper && pbits == 0u && k0 + 32 <= L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = (s[r] + grow[(t0 + (r & 3) + 8 * (r >> 2)) & 63]) * c2;
                    mt = fmaxf(mt, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bk = (r & 3) + 8 * (r >> 2) + 4 * h, key = k0 + bk;
                    const bool masked = (CAUSAL && key > q) || key >= L || ((pbits >> bk) & 1u);
                    float v = -INFINITY;
                    if (!masked) {
                        float g = 0.f;
                        if (CAUSAL || (!upper && key <= q)) g = grow[(t0 + (r & 3) + 8 * (r >> 2)) & 63];
                        v = (s[r] + g) * c2;
                    }
                    s[r] = v;
                    mt = fmaxf(mt, v);
                }
            }
            mt = half_max(mt);
            const float m_new = fmaxf(m_run, mt);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_safe);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r] - m_safe); rs += s[r]; }
            l_run = l_run * alpha + rs;
            if (__any(m_new != m_run)) {                 // running maxima settle quickly: most steps skip the rescale
#pragma unroll
                for (int i = 0; i < C::DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 2; ++t) {                   // O^T[d][q] += V^T[d][key] . P^T[key][q]
                Frag<T> pf; frag_from_acc(pf, s, t);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) {
                    Frag<T> vf;                              // V^T[d][key] for the accumulator's key map
                    frag_load_tr(vf, Vs[buf], C::LDN, 16 * t + 4 * h, 16 * t + 8 + 4 * h, i * 32, lane);
                    mma32(o[i], vf, pf);
                }
            }
        }
        if constexpr (MAIN) {
            sstore(buf ^ 1);                      // buf^1 was last read in step kt-1 (barrier since)
            gload_full(kt + 2);
        } else if (kt + 1 < nkt) {
            sstore(buf ^ 1);
            if (kt + 2 < nkt) gload(kt + 2);
        }
        block_sync_lds();               // LDS hand-over only: prefetch loads stay in flight
    };
    const int nmain = (qb * 128 + 96 < L) ? max(0, min(qb * 4, (L >> 5) - 2)) : 0;
    int kt = 0;
    for (; kt < nmain; ++kt) step(kt, std::true_type{});
    for (; kt < nkt; ++kt) step(kt, std::false_type{});
    if (!wave_on || q >= L) return;
    const float l_tot = half_sum(l_run);
    const float inv = 1.f / l_tot;
    if (h == 0) lse[((size_t)b * H + head) * L + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    T* op = out + ((size_t)b * L + q) * dm + head * DH;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            if (i * 32 + 8 * gq + 4 * h < DH)
                st4<T>(op + i * 32 + 8 * gq + 4 * h, o[i][4 * gq] * inv, o[i][4 * gq + 1] * inv, o[i][4 * gq + 2] * inv,
                       o[i][4 * gq + 3] * inv);
}

// =====================================================================================
// backward 1/3 (query-owned): delta, dQ, and the materialised P^T, dS^T, dG^T
// =====================================================================================
// CAUSAL = false: backward of the bidirectional forward (MusicRegression): every key tile is visited and written; tiles
// above the diagonal have no relative term (no G, no dG, no E^T product).
template <typename T, int DH, bool CAUSAL = true>
__global__ __launch_bounds__(256, 2) void rga_bwd_q_kernel(
    const T* __restrict__ qkv, const T* __restrict__ Epk,
    const uint8_t* __restrict__ key_pad, const T* __restrict__ out, const float* __restrict__ lse,
    const T* __restrict__ dout, T* __restrict__ dqkv, float* __restrict__ delta_ws, T* __restrict__ PT,
    T* __restrict__ dST, int B, int L, int Lp, int H, int M, float scale) {
    using C = ACfg<T, DH>;
    constexpr int LDR = 72;                         // dG ring row (elements of T): 64-column ring + 8
    __shared__ __attribute__((aligned(16))) T Ks[2][32 * C::LDN];      // natural K tile: 16-byte fragment reads (S) and transpose reads (dQ)
    __shared__ __attribute__((aligned(16))) T Vs[2][32 * C::LDN];
    __shared__ __attribute__((aligned(16))) float Gs[4][32 * LDG2];     // per wave: [q][64-column ring]
    __shared__ __attribute__((aligned(16))) T Ds[4][32 * LDR];          // per wave: [q][64-column ring] of dG
    __shared__ uint32_t Ps[2][32];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int nqb = (L + 127) / 128;
    const int bh = blockIdx.x % (B * H), qb = nqb - 1 - (int)(blockIdx.x / (B * H));
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const T* qb_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* kb_ = qb_ + dm;
    const T* vb_ = qb_ + 2 * dm;
    const int q0 = qb * 128 + wid * 32;
    const int q = q0 + a;
    const bool wave_on = q0 < L;
    const bool row_on = wave_on && q < L;
    const int nkt = CAUSAL ? min((L + 31) / 32, qb * 4 + 4) : (L + 31) / 32;
    const int my_last_kt = qb * 4 + wid;
    const float c2 = scale * 1.4426950408889634f;

    Frag<T> qf[C::KA], dof[C::KA];
    row_frags<T, DH>(qf, qb_ + (size_t)q * ldq, row_on, h);
    const size_t orow = ((size_t)b * L + q) * dm + head * DH;
    row_frags<T, DH>(dof, dout + orow, row_on, h);
    float delta = 0.f, lse2 = 0.f;
    if (row_on) {
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                delta += ET<T>::to_f(dout[orow + kk * 16 + h * 8 + e]) * ET<T>::to_f(out[orow + kk * 16 + h * 8 + e]);
        lse2 = lse[((size_t)b * H + head) * L + q] * 1.4426950408889634f;
    }
    delta = half_sum(delta);
    if (row_on && h == 0) delta_ws[((size_t)b * H + head) * L + q] = delta;

    f32x16_t dq[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) acc_zero(dq[i]);
    // dG ring starts zeroed: the first lo block only receives its upper-right triangle
    for (int i = lane; i < 32 * LDR; i += 64) Ds[wid][i] = ET<T>::from_f(0.f);

    chunk16 rk[TileT<T, 32, DH>::NPT], rv[TileT<T, 32, DH>::NPT];
    uint32_t rp = 0, rpm = 0;
    // pad flags: always one byte load per thread (a valid dummy row when there is no mask) -- a load under a
    // branch would make every later vmcnt wait conservative
    const uint8_t* kp_ = key_pad ? key_pad + (size_t)b * L : reinterpret_cast<const uint8_t*>(lse);
    const uint32_t kp_on = key_pad ? 0xffu : 0u;
    auto gload = [&](int kt) __attribute__((always_inline)) {
        tile_gload<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
        tile_gload<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
        rp = kp_[min(kt * 32 + (tid & 31), L - 1)];
        rpm = kt * 32 + (tid & 31) < L ? kp_on : 0u;
    };
    auto gload_full = [&](int kt) __attribute__((always_inline)) {       // tile kt entirely below L
        tile_gload_full<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, tid);
        tile_gload_full<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, tid);
        rp = kp_[kt * 32 + (tid & 31)];            // masked when it is stored: no early use, no early wait
        rpm = kp_on;
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        tile_sstore<T, 32, DH, C::LDN>(rk, Ks[buf], tid);
        tile_sstore<T, 32, DH, C::LDN>(rv, Vs[buf], tid);
        if (tid < 32) Ps[buf][tid] = rp & rpm;
    };
    auto g_block = [&](const Frag<T>* ef, int eb) {
        f32x16_t g; acc_zero(g);
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) mma32(g, ef[kk], qf[kk]);
        float* gs = &Gs[wid][a * LDG2 + (eb & 1) * 32];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4_t*>(gs + 8 * gq + 4 * h) = (f32x4_t){g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
    };
    // packed relative table (me_rga_pack_rel): every fragment is one contiguous 1 KB image
    auto e_frags = [&](Frag<T>* f, int eb) __attribute__((always_inline)) {      // E rows of block eb: A operand of G^T = E . Q^T
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) frag_load(f[kk], Epk + (size_t)eb * C::PK + (kk * 64 + lane) * 8);
    };
    // E^T of block eb: A operand of dQ^T[d][q] += E^T[d][e] dG^T[e][q], contraction map e = 16 t + 8 h + j on both operands
    auto et_frags = [&](Frag<T> (*f)[2], int eb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::DB; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) frag_load(f[i][t], Epk + (size_t)eb * C::PK + C::PK_B + ((i * 2 + t) * 64 + lane) * 8);
    };
    gload(0);
    const int eb0 = (M - 32 - q0) >> 5;
    Frag<T> ef[C::KA];
    Frag<T> etf[C::DB][2];
    if (wave_on) {
        e_frags(ef, eb0);
        g_block(ef, eb0);
        if (my_last_kt > 0) e_frags(ef, eb0 + 1);
    }
    const size_t ws_bh = (size_t)bh * Lp * Lp;
    sstore(0);
    if (nkt > 1) gload(1);
    __syncthreads();
    // One key tile.  MAIN = every wave of the block is strictly above its diagonal tile and tiles kt + 1,
    // kt + 2 lie entirely below L: no wave-, tile- or bounds-dependent branch encloses a global load or
    // store, so the compiler's s_waitcnt bookkeeping stays exact (vmcnt is in order: one conservative
    // vmcnt(0) exposes the K / V prefetch latency and the tile-store acknowledgements in every step).
    auto step = [&](int kt, auto main_tag) __attribute__((always_inline)) {
        constexpr bool MAIN = decltype(main_tag)::value;
        const int buf = kt & 1;
        if (MAIN || (wave_on && (!CAUSAL || kt <= my_last_kt))) {
            const int k0 = kt * 32;
            const bool diag = !MAIN && kt == my_last_kt;
            const bool upper = !CAUSAL && !MAIN && kt > my_last_kt;      // bidirectional only: above the diagonal, no relative term
            const int eb_lo = eb0 + kt;
            if constexpr (MAIN) {
                g_block(ef, eb_lo + 1);               // the next block's E rows are fetched after the softmax (register budget)
            } else if (!diag && !upper) {
                g_block(ef, eb_lo + 1);
                if (kt + 1 < my_last_kt) e_frags(ef, eb_lo + 2);
            }
            if (ME_ABL != 3 && !(MAIN && ME_ABL == 8) && !upper) et_frags(etf, eb_lo);      // E^T block of this step's lo block: in flight during S / dP / softmax
            f32x16_t s, dp; acc_zero(s); acc_zero(dp);
#pragma unroll
            for (int kk = 0; kk < C::KA; ++kk) {
                Frag<T> kf, vf;
                frag_load(kf, &Ks[buf][a * C::LDN + kk * 16 + h * 8]);
                frag_load(vf, &Vs[buf][a * C::LDN + kk * 16 + h * 8]);
                mma32(s, kf, qf[kk]);          // S^T[key][q]
                mma32(dp, vf, dof[kk]);        // dP^T[key][q] = V[key] . dO[q]
            }
            uint32_t pbits = 0;
            if (key_pad) pbits = __builtin_amdgcn_readfirstlane((uint32_t)__ballot(lane < 32 && Ps[buf][a] != 0));
            // band element m (0..62) of this tile sits at ring column ((eb_lo & 1) * 32 + m) & 63 (G and dG rings)
            const float* grow = &Gs[wid][a * LDG2];
            T* drow = &Ds[wid][a * LDR];
            const int t0 = (eb_lo & 1) * 32 + 31 - a + 4 * h;
            const bool plain = !diag && !upper && pbits == 0u && k0 + 32 <= L && q0 + 32 <= L;
            // the 16 ring reads are issued as one batch and every element is computed branch-free: per-element
            // exec-mask branches serialise the LDS latency (one read -> wait -> exp per basic block)
            const float nds = -delta * scale;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {                 // two batches of 8 ring reads (register budget)
                float gv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] = grow[(t0 + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2)) & 63];
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = r0 + j;
                        const float p = fast_exp2(fmaf(s[r] + gv[j], c2, -lse2));
                        s[r] = p * fmaf(dp[r], scale, nds);
                        dp[r] = p;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = r0 + j;
                        const int bk = (r & 3) + 8 * (r >> 2) + 4 * h, key = k0 + bk;
                        const bool masked = !row_on || (CAUSAL && key > q) || key >= L || ((pbits >> bk) & 1u);
                        const float gq = (CAUSAL || (!upper && key <= q)) ? gv[j] : 0.f;                   // no relative term above the diagonal
                        const float p = fast_exp2(masked ? -INFINITY : fmaf(s[r] + gq, c2, -lse2));       // exp2(-inf) = 0
                        s[r] = masked ? 0.f : p * fmaf(dp[r], scale, nds);
                        dp[r] = p;
                    }
                }
            }
            if (ME_ABL != 4 && !upper) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // dG collects dS only where the relative term exists (bidirectional diagonal tile: key <= q)
                    const bool rel = CAUSAL || k0 + (r & 3) + 8 * (r >> 2) + 4 * h <= q;
                    drow[(t0 + (r & 3) + 8 * (r >> 2)) & 63] = ET<T>::from_f(rel ? s[r] : 0.f);
                }
            }
            // ---- materialise P^T, dS^T tiles [key][q]: transpose through the (now dead) lo slot of the G
            //      ring so that the tiles leave as 16-byte row-contiguous stores.  Rows key >= L and
            //      columns q >= L carry exact zeros (masked), consistent with the zero-initialised workspace.
            // staging = the 32 dead lo columns of every ring row (row stride LDG2 floats)
            T* stg = reinterpret_cast<T*>(&Gs[wid][(eb_lo & 1) * 32]);
            constexpr int LDX = LDG2 * (int)(sizeof(float) / sizeof(T));   // staging row stride in elements of T
            constexpr int CPRX = 32 / C::CH;                        // chunks per 32-wide row
            auto flush_tile = [&](T* gdst) {                        // stg[32][LDX] -> one contiguous [32 key][32 q] tile
#pragma unroll
                for (int it = 0; it < 32 * CPRX / 64; ++it) {
                    const int c = it * 64 + lane, row = c / CPRX, cc = (c % CPRX) * C::CH;
                    st_chunk(gdst + (size_t)row * (ME_WS_LAYOUT == 0 ? Lp : 32) + cc, ld_chunk(&stg[row * LDX + cc]));
                }
            };
            const size_t tile_off = ws_bh + ws_row(k0, q0 >> 5, Lp);
            if constexpr (MAIN && ME_ABL != 8)      // issued BEFORE the tile stores: in-order vmcnt then never makes the next step wait for them
                e_frags(ef, min(eb_lo + 2, (M >> 5) - 1));          // clamped: unused past the diagonal
            if constexpr (sizeof(T) == 2) {
                // 16-bit tier: the accumulator rows go to LDS in their NATURAL [q][key] order (4 x ds_write_b64: the lane's
                // 4 consecutive keys of each register quad) and come back transposed through ds_read_b64_tr_b16 -- 8 LDS
                // instructions per tile instead of 16 two-byte scatters + 2 reads, and two 16-byte stores that each write 16
                // complete 64-byte tile rows.  Chunk slot XOR (row >> 3) & 1: the 16-lane write groups hit 16 distinct bank pairs.
                char* stgb = reinterpret_cast<char*>(stg);
                constexpr int LDB = LDG2 * 4;                                         // staging row stride (bytes)
                const int gidx = lane >> 4, l16 = lane & 15;
                auto put_tile = [&](const f32x16_t& v) __attribute__((always_inline)) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        st4<T>(reinterpret_cast<T*>(stgb + a * LDB + (((2 * g + h) ^ ((a >> 3) & 1)) << 3)), v[4 * g], v[4 * g + 1], v[4 * g + 2],
                               v[4 * g + 3]);
                };
                // lane (group gidx, l16) supplies the chunk (row 8 gidx + 4 s + l16 / 4, keys 16 kh + 4 (l16 % 4) ..) and receives
                // key 16 kh + l16, queries 8 gidx + 4 s .. + 3: with s = 0, 1 that is 16 contiguous bytes of tile row `key`
                auto flush_tr = [&](T* gdst) __attribute__((always_inline)) {
                    typedef short v4s __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        v4s x[2];
#pragma unroll
                        for (int s_ = 0; s_ < 2; ++s_) {
                            const char* src = stgb + (8 * gidx + 4 * s_ + (l16 >> 2)) * LDB + (((kh * 4 + (l16 & 3)) ^ (gidx & 1)) << 3);
                            x[s_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)src);
                        }
                        chunk16 c;
                        reinterpret_cast<v4s*>(&c)[0] = x[0];
                        reinterpret_cast<v4s*>(&c)[1] = x[1];
                        st_chunk(gdst + (size_t)(kh * 16 + l16) * (ME_WS_LAYOUT == 0 ? Lp : 32) + 8 * gidx, c);
                    }
                };
                if (ME_ABL != 1) {
                    put_tile(dp);
                    flush_tr(PT + tile_off);
                    put_tile(s);
                    flush_tr(dST + tile_off);
                }
            } else if (ME_ABL != 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * LDX + a] = ET<T>::from_f(dp[r]);
                flush_tile(PT + tile_off);
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * LDX + a] = ET<T>::from_f(s[r]);
                flush_tile(dST + tile_off);
            }
            // ---- dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> dsf; frag_from_acc(dsf, s, t);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) {
                    Frag<T> kf;                              // K^T[d][key] for the accumulator's key map
                    frag_load_tr(kf, Ks[buf], C::LDN, 16 * t + 4 * h, 16 * t + 8 + 4 * h, i * 32, lane);
                    mma32(dq[i], kf, dsf);
                }
            }
            // ---- the lo block of dG is complete now: relative part of dQ and flush of dG^T
            if (!upper)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> dgf;
                const T* dlo = drow + (eb_lo & 1) * 32;
                frag_load(dgf, dlo + 16 * t + 8 * h);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) { if (ME_ABL != 3) mma32(dq[i], etf[i][t], dgf); }
            }
        }
        if constexpr (MAIN) {
            sstore(buf ^ 1);                      // buf^1 was last read in step kt-1 (barrier since)
            gload_full(kt + 2);
        } else if (kt + 1 < nkt) {
            sstore(buf ^ 1);
            if (kt + 2 < nkt) gload(kt + 2);
        }
        if (!(MAIN && ME_ABL == 7)) block_sync_lds();               // LDS hand-over only: prefetch loads / tile stores stay in flight
    };
    // MAIN steps: all four waves on and off-diagonal (kt < 4 qb), tiles kt + 1, kt + 2 whole (kt + 3 <= L / 32)
    const int nmain = (qb * 128 + 96 < L) ? max(0, min(qb * 4, (L >> 5) - 2)) : 0;
    int kt = 0;
    for (; kt < nmain; ++kt) step(kt, std::true_type{});
    for (; kt < nkt; ++kt) step(kt, std::false_type{});
    if (!row_on) return;
    T* dqp = dqkv + ((size_t)b * L + q) * ldq + head * DH;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            if (i * 32 + 8 * gq + 4 * h < DH)
                st4<T>(dqp + i * 32 + 8 * gq + 4 * h, dq[i][4 * gq], dq[i][4 * gq + 1], dq[i][4 * gq + 2], dq[i][4 * gq + 3]);
}

// =====================================================================================
// backward 2/3 (key-owned, streaming):  dV[key] = sum_q P^T[key][q] dO[q],  dK[key] = sum_q dS^T[key][q] Q[q]
// =====================================================================================
// Block = 128 keys (4 waves x 32) x DH.  Every step stages a 32-query slab: P^T, dS^T tiles
// [128 key][32 q] and dO^T, Q^T tiles [DH][32 q]; all four are contraction-contiguous, so the
// fragments are plain 16-byte LDS reads.  The materialised tensors are read exactly once:
// the kernel is HBM-bound (2 x Lp^2/2 elements per (b, head)).
template <typename T, int DH, int NWK, bool CAUSAL = true>
__global__ __launch_bounds__(NWK * 64) void rga_bwd_kv_kernel(const T* __restrict__ PT, const T* __restrict__ dST,
                                                              const T* __restrict__ qkv, const T* __restrict__ dout,
                                                              T* __restrict__ dqkv, int B, int L, int Lp, int H) {
    // NWK waves x 32 keys per block.  8 waves (256 keys): the Q / dO slabs every block streams are fetched half as often
    constexpr int CH = ET<T>::CH, LDP = 32 + CH, DB = ACfg<T, DH>::DB, LDV = ACfg<T, DH>::LDV;
    constexpr int KB = NWK * 32, NTHR = NWK * 64;
    __shared__ __attribute__((aligned(16))) T Pt[2][KB * LDP];
    __shared__ __attribute__((aligned(16))) T St[2][KB * LDP];
    __shared__ __attribute__((aligned(16))) T Os[2][32 * LDV];         // natural dO / Q slabs [32 q][DH], transpose-read
    __shared__ __attribute__((aligned(16))) T Qs[2][32 * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int BH = B * H;
    const int bh = blockIdx.x % BH, kb = blockIdx.x / BH;          // low key blocks (long loops) first
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const int k0 = kb * KB + wid * 32;
    const bool wave_on = k0 < L;
    const int nqt = (L + 31) / 32;
    const int qs0 = CAUSAL ? kb * NWK : 0;                          // bidirectional: every query tile contributes
    const int rows_valid = min(KB, Lp - kb * KB);
    const T* pt_ = PT + (size_t)bh * Lp * Lp;
    const T* st_ = dST + (size_t)bh * Lp * Lp;
    const T* q_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* o_ = dout + (size_t)b * L * dm + head * DH;

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) { acc_zero(dk[i]); acc_zero(dv[i]); }

    constexpr int CPRP = 32 / CH, NPTP = KB * CPRP / NTHR;              // P^T / dS^T tile [KB][32]: chunks per thread
    constexpr int CPRQ = DH / CH, NCHQ = 32 * CPRQ, NPTQ = (NCHQ + NTHR - 1) / NTHR;   // Q / dO slab [32][DH]
    chunk16 rp[NPTP], rs[NPTP], ro[NPTQ], rq[NPTQ];
    auto gload = [&](int qs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * NTHR, row = c / CPRP, cc = (c % CPRP) * CH;
            const size_t o = ws_row(kb * KB + min(row, rows_valid - 1), qs, Lp) + cc;
            rp[i] = row < rows_valid ? ld_chunk(pt_ + o) : zero_chunk();
            rs[i] = row < rows_valid ? ld_chunk(st_ + o) : zero_chunk();
        }
        const int qv = L - qs * 32;
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * NTHR, row = c / CPRQ, cc = (c % CPRQ) * CH;
            const bool ok = c < NCHQ && row < qv;
            ro[i] = ok ? ld_chunk(o_ + ((size_t)qs * 32 + row) * dm + cc) : zero_chunk();
            rq[i] = ok ? ld_chunk(q_ + ((size_t)qs * 32 + row) * ldq + cc) : zero_chunk();
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * NTHR, row = c / CPRP, cc = (c % CPRP) * CH;
            st_chunk(&Pt[buf][row * LDP + cc], rp[i]);
            st_chunk(&St[buf][row * LDP + cc], rs[i]);
        }
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * NTHR, row = c / CPRQ, cc = (c % CPRQ) * CH;
            if (c < NCHQ) { st_chunk(&Os[buf][row * LDV + cc], ro[i]); st_chunk(&Qs[buf][row * LDV + cc], rq[i]); }
        }
    };
    gload(qs0);
    sstore(0);
    if (qs0 + 1 < nqt) gload(qs0 + 1);
    __syncthreads();
    for (int qs = qs0; qs < nqt; ++qs) {
        const int buf = (qs - qs0) & 1;
        if (wave_on && (!CAUSAL || qs * 32 + 31 >= k0)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> pf, sf;
                frag_load(pf, &Pt[buf][(wid * 32 + a) * LDP + 16 * t + 8 * h]);
                frag_load(sf, &St[buf][(wid * 32 + a) * LDP + 16 * t + 8 * h]);
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    Frag<T> of, qf;
                    frag_load_tr(of, Os[buf], LDV, 16 * t + 8 * h, 16 * t + 8 * h + 4, i * 32, lane);     // dO^T[d][q]
                    frag_load_tr(qf, Qs[buf], LDV, 16 * t + 8 * h, 16 * t + 8 * h + 4, i * 32, lane);     // Q^T[d][q]
                    mma32(dv[i], pf, of);
                    mma32(dk[i], sf, qf);
                }
            }
        }
        if (qs + 1 < nqt) {
            sstore(buf ^ 1);
            if (qs + 2 < nqt) gload(qs + 2);
        }
        block_sync_lds();               // LDS hand-over only: prefetch loads / tile stores stay in flight
    }
    if (!wave_on) return;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + c_row(r, lane);
            if (key < L && i * 32 + a < DH) {
                T* base = dqkv + ((size_t)b * L + key) * ldq + head * DH + i * 32 + a;
                base[dm] = ET<T>::from_f(dk[i][r]);
                base[2 * dm] = ET<T>::from_f(dv[i][r]);
            }
        }
}

// =====================================================================================
// backward 3/3 (E-row-owned, streaming):  dE[e][d] += sum_{bh, q} dG^T[bh][c(e)][q] Q^T[bh][d][q]
// =====================================================================================
// dG is dS re-indexed: dG^T[c][q] = dS^T[key][q] with key = c - (Lp - 1) + q (c = e - (M - Lp)).  The kernel
// therefore reads the dS^T workspace the query kernel writes for dK anyway and applies the shear while
// staging: a [128 c][32 q] operand tile is the band of 159 dS^T rows key = kb + kk (kb = c0 - Lp + 1 + 32 qs)
// with element (kk, qq) landing at tile row kk - qq.  The band arrives as natural 16-byte row chunks and is
// scattered element-wise into LDS (the kernel is HBM bound with 4 MFMAs per step, the extra LDS stores are
// free) -- this removed the separate dG^T workspace: 268 MB written and read per layer.
// Block = 128 rows c (4 waves x 32) x DH, one split of the (b, head) range; register accumulation over
// (bh, q) and one atomic flush per block.  Column block cb only receives queries
// q >= 32*(Lp/32 - 1 - cb); earlier slabs are skipped.
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_bwd_e_kernel(const T* __restrict__ dST, const T* __restrict__ qkv,
                                                        float* __restrict__ dE, int B, int L, int Lp, int H, int M) {
    constexpr int CH = ET<T>::CH, LDP = 32 + CH, DB = ACfg<T, DH>::DB, LDV = ACfg<T, DH>::LDV;
    __shared__ __attribute__((aligned(16))) T Gt[2][(128 + 16) * LDP];  // + 16 dump rows: out-of-tile band elements are stored there, not branched around
    __shared__ __attribute__((aligned(16))) T Qs[2][32 * LDV];         // natural Q slab [32 q][DH], transpose-read

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int BH = B * H;
    const int nqt = (L + 31) / 32;
    const int ncb = Lp / 32;
    const int ngx = (Lp + 127) / 128;                   // groups of 4 column blocks (128 rows of dG^T)
    // 1-D grid; group x needs nq(x) query slabs per (b, head) -> it gets blocks in proportion to nq(x)
    // (the groups near column 0 only see the last few query tiles, the last group sees all of them)
    int gx = 0, slot = 0, nslots = 1;
    {
        int total = 0;
        for (int x = 0; x < ngx; ++x) total += nqt - max(0, ncb - 1 - min(ncb - 1, x * 4 + 3));
        int first = 0;
        for (int x = 0; x < ngx; ++x) {
            const int nqx = nqt - max(0, ncb - 1 - min(ncb - 1, x * 4 + 3));
            const int cnt = 1 + (int)(((long)((int)gridDim.x - ngx) * nqx) / total);     // sum <= gridDim.x
            if ((int)blockIdx.x >= first && (int)blockIdx.x < first + cnt) { gx = x; slot = blockIdx.x - first; nslots = cnt; }
            first += cnt;
        }
        if ((int)blockIdx.x >= first) return;             // rounding leftovers
    }
    const int c0 = gx * 128;
    const int cbw = gx * 4 + wid;
    const bool wave_on = cbw * 32 < Lp;
    const int my_qmin = max(0, ncb - 1 - cbw);
    const int qs0 = max(0, ncb - 1 - min(ncb - 1, gx * 4 + 3));       // earliest slab any wave needs
    const int nq = nqt - qs0;
    const int per = (BH + nslots - 1) / nslots;
    const int bh_lo = slot * per, bh_hi = min(BH, bh_lo + per);
    const int nsteps = (bh_hi - bh_lo) * nq;
    if (nsteps <= 0 || nq <= 0) return;

    f32x16_t acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) acc_zero(acc[i]);

    using BandT = TileT<T, 160, 32>;                    // 159 band rows (+1 pad) x 32 queries, CPR chunks per row
    chunk16 rg[BandT::NPT], rq[TileT<T, 32, DH>::NPT];
    bool rgv[BandT::NPT];                               // band row inside the workspace?  applied when the chunk is sheared into LDS
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    // No branch encloses a global load or an LDS store in the steady state: rows are clamped, invalid band rows
    // are zeroed by a select at shear time and out-of-tile elements go to dump rows (exact s_waitcnt bookkeeping,
    // no exec-mask branch per element).
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int bh = bh_lo + s / nq, qs = qs0 + s % nq;
        const int kb = c0 - Lp + 1 + qs * 32;            // key of tile element (row 0, column 0)
        const T* src = dST + (size_t)bh * Lp * Lp;
#pragma unroll
        for (int i = 0; i < BandT::NPT; ++i) {
            const int c = min(tid + i * 256, BandT::NCH - 1), kk = c / BandT::CPR, cc = (c % BandT::CPR) * CH;
            const int key = kb + kk;
            rgv[i] = kk < 159 && key >= 0 && key < Lp;
            rg[i] = ld_chunk(src + ws_row(min(max(key, 0), Lp - 1), qs, Lp) + cc);
        }
        const T* qsrc = qkv + ((size_t)(bh / H) * L + qs * 32) * ldq + (bh % H) * DH;
        using QT = TileT<T, 32, DH>;
#pragma unroll
        for (int i = 0; i < QT::NPT; ++i) {              // rows past L repeat the last row: their dS^T columns are exact zeros
            const int c = min(tid + i * 256, QT::NCH - 1);
            rq[i] = ld_chunk(qsrc + (size_t)min(c / QT::CPR, L - 1 - qs * 32) * ldq + (c % QT::CPR) * CH);
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        // shear: band element (kk, qq) -> tile row kk - qq; every tile element is written exactly once per step
#pragma unroll
        for (int i = 0; i < BandT::NPT; ++i) {
            const int c = tid + i * 256, kk = c / BandT::CPR, cc = (c % BandT::CPR) * CH;
            const T* v = reinterpret_cast<const T*>(&rg[i]);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int cl = kk - cc - e;
                const int row = (cl >= 0 && cl < 128) ? cl : 128 + (kk & 15);     // threads past the band (kk >= 160) land in the dump rows too
                Gt[buf][row * LDP + cc + e] = rgv[i] ? v[e] : ET<T>::from_f(0.f);
            }
        }
        tile_sstore<T, 32, DH, LDV>(rq, Qs[buf], tid);
    };
    gload(0);
    sstore(0);
    gload(min(1, nsteps - 1));
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        const int qs = qs0 + s % nq;
        if (wave_on && qs >= my_qmin) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> gf;
                frag_load(gf, &Gt[buf][(wid * 32 + a) * LDP + 16 * t + 8 * h]);
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    Frag<T> qf;
                    frag_load_tr(qf, Qs[buf], LDV, 16 * t + 8 * h, 16 * t + 8 * h + 4, i * 32, lane);     // Q^T[d][q]
                    mma32(acc[i], gf, qf);
                }
            }
        }
        sstore(buf ^ 1);                            // past the last step: a harmless re-store of the last slab
        gload(min(s + 2, nsteps - 1));
        block_sync_lds();               // LDS hand-over only: prefetch loads / tile stores stay in flight
    }
    if (!wave_on) return;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = cbw * 32 + c_row(r, lane) + (M - Lp);
            const float v = acc[i][r];
            if (e >= 0 && e < M && v != 0.f && i * 32 + a < DH) atomicAdd(&dE[(size_t)e * DH + i * 32 + a], v);
        }
}

// =====================================================================================
// relative table E [M][DH] -> packed fragment images (what rga_fwd / rga_bwd_q load with one 16-byte chunk per lane)
// =====================================================================================
// block eb (32 rows of E), PK elements:   image kk < KA      : lane (a, h) holds E[32 eb + a][16 kk + 8 h + 0..7]
//                                         image KA + 2 i + t : lane (a, h) holds E[32 eb + 16 t + 8 h + 0..7][32 i + a]
// (zero where 32 i + a >= DH).  The multi-tensor weight refresh (me_cast_transpose_multi, mode 1) writes the same layout.
template <typename T, int DH>
__global__ __launch_bounds__(256) void rel_pack_kernel(const T* __restrict__ E, T* __restrict__ Epk) {
    using C = ACfg<T, DH>;
    const int eb = blockIdx.x;
    for (int idx = threadIdx.x; idx < C::PK; idx += 256) {
        const int img = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7, a = lane & 31, h = lane >> 5;
        T v = ET<T>::from_f(0.f);
        if (img < C::KA) v = E[(size_t)(eb * 32 + a) * DH + img * 16 + h * 8 + j];
        else {
            const int i = (img - C::KA) >> 1, t = (img - C::KA) & 1;
            if (i * 32 + a < DH) v = E[(size_t)(eb * 32 + 16 * t + 8 * h + j) * DH + i * 32 + a];
        }
        Epk[(size_t)eb * C::PK + idx] = v;
    }
}
template <typename T, int DH>
int pack_launch(const void* E, void* Epk, int M, hipStream_t st) {
    rel_pack_kernel<T, DH><<<M / 32, 256, 0, st>>>((const T*)E, (T*)Epk);
    return me_launch_status();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int DH>
int fwd_launch(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, int B, int L, int H, int M,
               int causal, hipStream_t st) {
    const int nqb = (L + 127) / 128;
    const float scale = 1.f / sqrtf((float)DH);
    if (causal)
        rga_fwd_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (T*)out, lse, B, L, H, M, scale);
    else
        rga_fwd_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (T*)out, lse, B, L, H, M, scale);
    return me_launch_status();
}

template <typename T, int DH>
int bwd_launch(const void* qkv, const void* Epk, const uint8_t* key_pad, const void* out, const float* lse,
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* PT, void* dST, int B, int L,
               int Lp, int H, int M, int causal, hipStream_t st) {
    const int nqb = (L + 127) / 128;
    const float scale = 1.f / sqrtf((float)DH);
    if (causal)
        rga_bwd_q_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (const T*)out, lse,
                                                                  (const T*)dout, (T*)dqkv, delta_ws, (T*)PT, (T*)dST, B, L, Lp,
                                                                  H, M, scale);
    else
        rga_bwd_q_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (const T*)out, lse,
                                                                   (const T*)dout, (T*)dqkv, delta_ws, (T*)PT, (T*)dST, B, L, Lp,
                                                                   H, M, scale);
    int rc = me_launch_status();
    if (rc) return rc;
    // 256-key blocks (MIDIEMO_KV8=1, 16-bit tier) fetch the Q / dO slabs half as often, but measured over several runs on
    // one box they are not faster (128-key: 168 / 176 / 173 us, 256-key: 187 / 199 / 172 us at C2): off by default
    static const int kv8 = getenv("MIDIEMO_KV8") ? atoi(getenv("MIDIEMO_KV8")) : 0;
    bool big = false;
    if constexpr (sizeof(T) == 2) {
        if (kv8 && causal) {
            big = true;
            rga_bwd_kv_kernel<T, DH, 8, true><<<B * H * ((L + 255) / 256), 512, 0, st>>>((const T*)PT, (const T*)dST, (const T*)qkv,
                                                                                        (const T*)dout, (T*)dqkv, B, L, Lp, H);
        }
    }
    if (!big) {
        if (causal)
            rga_bwd_kv_kernel<T, DH, 4, true><<<B * H * nqb, 256, 0, st>>>((const T*)PT, (const T*)dST, (const T*)qkv, (const T*)dout,
                                                                          (T*)dqkv, B, L, Lp, H);
        else
            rga_bwd_kv_kernel<T, DH, 4, false><<<B * H * nqb, 256, 0, st>>>((const T*)PT, (const T*)dST, (const T*)qkv, (const T*)dout,
                                                                           (T*)dqkv, B, L, Lp, H);
    }
    rc = me_launch_status();
    if (rc) return rc;
    const int ngx = (Lp + 127) / 128;
    // grid sweep at C2 (us per launch): 512: 131, 768: 118, 1024: 127, 1160: 108, 1536: 106, 2048: 113, 4096: 111 --
    // ~6 blocks per CU (4 resident): short blocks fill the tail left by the proportional group split
    int eblocks = 1536;
    if (eblocks > ngx * B * H) eblocks = ngx * B * H;
    if (eblocks < ngx) eblocks = ngx;
    rga_bwd_e_kernel<T, DH><<<eblocks, 256, 0, st>>>((const T*)dST, (const T*)qkv, dE, B, L, Lp, H, M);
    return me_launch_status();
}

}  // namespace

#define ME_ATTN_DISPATCH(CALL)                                                   \
    if (dtype == ME_F32) {                                                       \
        if (dh == 64) { typedef float T; constexpr int DH = 64; return CALL; }   \
        if (dh == 48) { typedef float T; constexpr int DH = 48; return CALL; }   \
        if (dh == 32) { typedef float T; constexpr int DH = 32; return CALL; }   \
    } else if (dtype == ME_BF16) {                                               \
        if (dh == 64) { typedef bf16_t T; constexpr int DH = 64; return CALL; }  \
        if (dh == 48) { typedef bf16_t T; constexpr int DH = 48; return CALL; }  \
        if (dh == 32) { typedef bf16_t T; constexpr int DH = 32; return CALL; }  \
    } else return ME_ERR_BAD_DTYPE;                                              \
    return ME_ERR_BAD_SHAPE;

extern "C" {

int me_rga_pack_rel(const void* E, void* Epk, int M, int dh, int dtype, void* stream) {
    me_clear_error();
    if (!E || !Epk) return ME_ERR_NULL;
    if (M <= 0 || (M & 31)) return ME_ERR_BAD_SHAPE;
    if (!aligned16(Epk)) return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((pack_launch<T, DH>(E, Epk, M, st)))
}

int me_rga_fwd(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, int B, int L, int H, int dh,
               int M, int causal, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31)) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out)) return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((fwd_launch<T, DH>(qkv, Epk, key_pad, out, lse, B, L, H, M, causal, st)))
}

int me_rga_bwd(const void* qkv, const void* Epk, const uint8_t* key_pad, const void* out, const float* lse,
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* PT, void* dST, int B, int L,
               int Lp, int H, int dh, int M, int causal, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse || !dout || !dqkv || !dE || !delta_ws || !PT || !dST) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31) || (Lp & 31) || Lp < L || Lp > M) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv) ||
        !aligned16(PT) || !aligned16(dST))
        return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((bwd_launch<T, DH>(qkv, Epk, key_pad, out, lse, dout, dqkv, dE, delta_ws, PT, dST, B, L, Lp, H, M,
                                        causal, st)))
}

}  // extern "C"
