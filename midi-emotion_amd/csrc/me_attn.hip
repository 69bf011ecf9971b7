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
// Backward.  The query-owned kernel recomputes P, forms dS, accumulates dQ (key part and relative part) and
// MATERIALISES, for the current layer only, two tensors in the compute type as contiguous 32 x 32 tiles:
//     P^T  [bh][key tile][query tile >= key tile]   (packed lower triangle; full square for the bidirectional variant)
//     dG^T [bh][query tile qt][step t <= qt]        (the skewed dS of E block eb0(qt) + t: rows = E row, columns = query)
// Every tile of both is written by exactly one wave in every call: no zero-initialisation contract.
//     dV = P^T dO,  dS^T = P^T o (V dO^T - delta) / sqrt(dh),  dK = dS^T Q   (rga_bwd_kv_kernel, key-owned: streams P^T once,
//                                                                          recomputes dS from it -- no exp, no skew)
//     dE[e] += sum_{bh, q} dG^T[e][q] Q[q]                                 (rga_bwd_e_kernel, E-row-owned: plain tile stream)
// Round 1 also materialised dS^T (the key-owned kernel read it, the E kernel re-read it as sheared bands): 2.33 GB of HBM
// traffic per layer at the headline shape against 1.2 GB now.
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

// Workspace tiles of one (b, head); every tile is a contiguous 32 x 32 block of T (1024 elements).
//   P^T : causal: packed lower triangle, tile (kt, qt >= kt); bidirectional: full square (kt, qt)
//   dG^T: tile (qt, t <= qt)
ME_DEV size_t pt_tile(int kt, int qt, int nq, bool causal) {
    return causal ? (size_t)kt * nq - (size_t)kt * (kt - 1) / 2 + (qt - kt) : (size_t)kt * nq + qt;
}
ME_DEV size_t pt_tiles(int nq, bool causal) { return causal ? (size_t)nq * (nq + 1) / 2 : (size_t)nq * nq; }
ME_DEV size_t dg_tile(int qt, int t) { return (size_t)qt * (qt + 1) / 2 + t; }

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
        // halo: columns 64..66 of a ring row mirror columns 0..2, so that the 4-element groups of the skewed read never have
        // to wrap inside a group (one base address per group instead of an add / and / shift per element).  Branch-free:
        // lanes that do not own columns 0..3 of slot 0 rewrite their own first quad in place.
        float* hs = &Gs[wid][a * LDG2] + (((eb & 1) | h) ? (eb & 1) * 32 + 4 * h : 64);
        *reinterpret_cast<f32x4_t*>(hs) = (f32x4_t){g[0], g[1], g[2], g[3]};
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
            // band element m (0..62) of this tile sits at ring column ((eb_lo & 1) * 32 + m) & 63; the lane's 16 elements
            // are four groups of 4 consecutive columns (a group may run into the halo columns 64..66, never wraps)
            const float* grow = &Gs[wid][a * LDG2];
            const int t0 = (eb_lo & 1) * 32 + 31 - a + 4 * h;
            float gv[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float* gp = grow + ((t0 + 8 * gq) & 63);
#pragma unroll
                for (int i = 0; i < 4; ++i) gv[4 * gq + i] = gp[i];
            }
            // the running maximum is kept in RAW logit units (before the scale / log2e factor c2 > 0): the exponent is
            // one fma per element, exp2(s * c2 - m * c2)
            float mt = -INFINITY;
            if (!diag && !upper && pbits == 0u && k0 + 32 <= L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] += gv[r];
                    mt = fmaxf(mt, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bk = (r & 3) + 8 * (r >> 2) + 4 * h, key = k0 + bk;
                    const bool masked = (CAUSAL && key > q) || key >= L || ((pbits >> bk) & 1u);
                    const float g = (CAUSAL || (!upper && key <= q)) ? gv[r] : 0.f;
                    const float v = masked ? -INFINITY : s[r] + g;
                    s[r] = v;
                    mt = fmaxf(mt, v);
                }
            }
            mt = half_max(mt);
            const float m_new = fmaxf(m_run, mt);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = fast_exp2((m_run - m_safe) * c2);
            const float nm = -m_safe * c2;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], c2, nm)); rs += s[r]; }
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
    if (h == 0) lse[((size_t)b * H + head) * L + q] = (m_run * c2 + log2f(l_tot)) * 0.6931471805599453f;
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
    T* __restrict__ dGT, int B, int L, int Lp, int H, int M, float scale) {
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
        // halo: columns 64..66 of a ring row mirror columns 0..2, so that the 4-element groups of the skewed read never have
        // to wrap inside a group (one base address per group instead of an add / and / shift per element).  Branch-free:
        // lanes that do not own columns 0..3 of slot 0 rewrite their own first quad in place.
        float* hs = &Gs[wid][a * LDG2] + (((eb & 1) | h) ? (eb & 1) * 32 + 4 * h : 64);
        *reinterpret_cast<f32x4_t*>(hs) = (f32x4_t){g[0], g[1], g[2], g[3]};
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
    const int nq32 = Lp >> 5;
    T* const ptb = PT + (size_t)bh * pt_tiles(nq32, CAUSAL) * 1024;
    T* const dgb = dGT + (size_t)bh * pt_tiles(nq32, true) * 1024;
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
                for (int gq = 0; gq < 2; ++gq) {             // group bases: a group may run into the halo columns, never wraps
                    const float* gp = grow + ((t0 + 2 * r0 + 8 * gq) & 63);     // r0 = 0 / 8: register quads 0,1 / 2,3 = columns +0,+8 / +16,+24
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[4 * gq + i] = gp[i];
                }
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
            // ---- materialise the P^T tile [key][q]: transpose through the (now dead) lo slot of the G ring so that
            //      the tile leaves as 16-byte row-contiguous stores.  Rows key >= L and columns q >= L carry exact
            //      zeros (masked).
            // staging = the 32 dead lo columns of every ring row (row stride LDG2 floats)
            T* stg = reinterpret_cast<T*>(&Gs[wid][(eb_lo & 1) * 32]);
            constexpr int LDX = LDG2 * (int)(sizeof(float) / sizeof(T));   // staging row stride in elements of T
            constexpr int CPRX = 32 / C::CH;                        // chunks per 32-wide row
            auto flush_tile = [&](T* gdst) {                        // stg[32][LDX] -> one contiguous [32 key][32 q] tile
#pragma unroll
                for (int it = 0; it < 32 * CPRX / 64; ++it) {
                    const int c = it * 64 + lane, row = c / CPRX, cc = (c % CPRX) * C::CH;
                    st_chunk(gdst + (size_t)row * 32 + cc, ld_chunk(&stg[row * LDX + cc]));
                }
            };
            T* const pt_dst = ptb + pt_tile(kt, q0 >> 5, nq32, CAUSAL) * 1024;
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
                        st_chunk(gdst + (size_t)(kh * 16 + l16) * 32 + 8 * gidx, c);
                    }
                };
                if (ME_ABL != 1) {
                    put_tile(dp);
                    flush_tr(pt_dst);
                }
            } else if (ME_ABL != 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * LDX + a] = ET<T>::from_f(dp[r]);
                flush_tile(pt_dst);
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
            // ---- the lo block of dG is complete now: relative part of dQ, and the block leaves as the dG^T tile
            //      (qt, t = kt) [E row m][query]: the ring rows [q][m] are read back transposed (16-bit tier:
            //      ds_read_b64_tr_b16, two 16-byte stores per lane) -- the E kernel streams these tiles as they are.
            if (!upper) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    Frag<T> dgf;
                    const T* dlo = drow + (eb_lo & 1) * 32;
                    frag_load(dgf, dlo + 16 * t + 8 * h);
#pragma unroll
                    for (int i = 0; i < C::DB; ++i) { if (ME_ABL != 3) mma32(dq[i], etf[i][t], dgf); }
                }
                T* const dg_dst = dgb + dg_tile(q0 >> 5, kt) * 1024;
                const T* ring = &Ds[wid][(eb_lo & 1) * 32];                  // lo block: ring rows q, 32 columns m, row stride LDR
                if constexpr (ME_ABL == 9) {
                } else if constexpr (sizeof(T) == 2) {
                    typedef short v4s __attribute__((ext_vector_type(4)));
                    const int gidx = lane >> 4, l16 = lane & 15;
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        v4s x[2];
#pragma unroll
                        for (int s_ = 0; s_ < 2; ++s_) {
                            const T* src = ring + (8 * gidx + 4 * s_ + (l16 >> 2)) * LDR + kh * 16 + 4 * (l16 & 3);
                            x[s_] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)src);
                        }
                        chunk16 c;
                        reinterpret_cast<v4s*>(&c)[0] = x[0];
                        reinterpret_cast<v4s*>(&c)[1] = x[1];
                        st_chunk(dg_dst + (size_t)(kh * 16 + l16) * 32 + 8 * gidx, c);     // row m = 16 kh + l16, queries 8 gidx .. + 7
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        const int idx = it * 64 + lane, m = idx >> 5, qq = idx & 31;
                        dg_dst[idx] = ring[qq * LDR + m];
                    }
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
// backward 2/3 (key-owned):  dV[key] = sum_q P^T[key][q] dO[q],  dK[key] = sum_q dS^T[key][q] Q[q]
// =====================================================================================
// Block = 128 keys (4 waves x 32) x DH.  Every step stages a 32-query slab: the P^T tiles [128 key][32 q], the dO and Q
// slabs [32 q][DH] and -delta / sqrt(dh) of the 32 queries.  dS^T is NOT read from memory: with the wave's V rows resident
// in registers, dP^T = V dO^T is KA macro-atoms and dS = P o (dP - delta) / sqrt(dh) a multiply-add per element (no
// exponential, no relative-term skew: P already contains both) -- P^T is the only O(L^2) tensor this kernel streams.
// dP is accumulated as dP[q][key] (lane = key column), so that its registers carry the same (key, 8 queries) elements
// as the P^T fragment read with the accumulator's k-map; all contraction-over-q operands use that map.
template <typename T, int DH, bool CAUSAL = true>
__global__ __launch_bounds__(256) void rga_bwd_kv_kernel(const T* __restrict__ PT, const T* __restrict__ qkv,
                                                         const T* __restrict__ dout, const float* __restrict__ delta_ws,
                                                         T* __restrict__ dqkv, int B, int L, int Lp, int H, float scale) {
    using C = ACfg<T, DH>;
    constexpr int CH = ET<T>::CH, LDP = 32 + CH, DB = C::DB, LDV = C::LDV, KA = C::KA;
    constexpr int KB = 128;
    __shared__ __attribute__((aligned(16))) T Pt[2][KB * LDP];
    __shared__ __attribute__((aligned(16))) T Os[2][32 * LDV];         // natural dO / Q slabs [32 q][DH]
    __shared__ __attribute__((aligned(16))) T Qs[2][32 * LDV];
    __shared__ __attribute__((aligned(16))) float Dl[2][32];           // -delta[q] / sqrt(dh)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int BH = B * H;
    const int bh = blockIdx.x % BH, kb = blockIdx.x / BH;          // low key blocks (long loops) first
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const int k0 = kb * KB + wid * 32;
    const bool wave_on = k0 < L;
    const int nqt = (L + 31) / 32, nq32 = Lp >> 5;
    const int qs0 = CAUSAL ? kb * 4 : 0;                            // bidirectional: every query tile contributes
    const T* ptb = PT + (size_t)bh * pt_tiles(nq32, CAUSAL) * 1024;
    const T* q_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* o_ = dout + (size_t)b * L * dm + head * DH;
    const float* dl_ = delta_ws + (size_t)bh * L;

    Frag<T> vf[KA];                                                 // V rows of this wave's 32 keys (B operand of dP)
    row_frags<T, DH>(vf, q_ + 2 * dm + (size_t)(k0 + a) * ldq, wave_on && k0 + a < L, h);

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) { acc_zero(dk[i]); acc_zero(dv[i]); }

    constexpr int CPRP = 32 / CH, NPTP = KB * CPRP / 256;              // P^T tiles [128][32]: chunks per thread
    constexpr int CPRQ = DH / CH, NCHQ = 32 * CPRQ, NPTQ = (NCHQ + 255) / 256;   // Q / dO slab [32][DH]
    chunk16 rp[NPTP], ro[NPTQ], rq[NPTQ];
    float rd = 0.f;
    auto gload = [&](int qs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * 256, row = c / CPRP, cc = (c % CPRP) * CH;
            const int kt = min(kb * 4 + (row >> 5), nq32 - 1);
            // tiles above the diagonal do not exist in the packed triangle: clamped to the diagonal tile (never used)
            rp[i] = ld_chunk(ptb + pt_tile(kt, CAUSAL ? max(qs, kt) : qs, nq32, CAUSAL) * 1024 + (row & 31) * 32 + cc);
        }
        const int qv = L - qs * 32;
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * 256, row = c / CPRQ, cc = (c % CPRQ) * CH;
            const bool ok = c < NCHQ && row < qv;
            ro[i] = ok ? ld_chunk(o_ + ((size_t)qs * 32 + row) * dm + cc) : zero_chunk();
            rq[i] = ok ? ld_chunk(q_ + ((size_t)qs * 32 + row) * ldq + cc) : zero_chunk();
        }
        rd = dl_[min(qs * 32 + (tid & 31), L - 1)];
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * 256, row = c / CPRP, cc = (c % CPRP) * CH;
            st_chunk(&Pt[buf][row * LDP + cc], rp[i]);
        }
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * 256, row = c / CPRQ, cc = (c % CPRQ) * CH;
            if (c < NCHQ) { st_chunk(&Os[buf][row * LDV + cc], ro[i]); st_chunk(&Qs[buf][row * LDV + cc], rq[i]); }
        }
        if (tid < 32) Dl[buf][tid] = -rd * scale;
    };
    gload(qs0);
    sstore(0);
    if (qs0 + 1 < nqt) gload(qs0 + 1);
    __syncthreads();
    for (int qs = qs0; qs < nqt; ++qs) {
        const int buf = (qs - qs0) & 1;
        if (wave_on && (!CAUSAL || qs * 32 + 31 >= k0)) {
            f32x16_t dp; acc_zero(dp);                              // dP[q][key] = dO[q] . V[key]
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
                Frag<T> of;
                frag_load(of, &Os[buf][a * LDV + kk * 16 + h * 8]);
                mma32(dp, of, vf[kk]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // accumulator registers 8t .. 8t+7 of lane (key a, h) = queries 16t + 4h + {0..3} and 16t + 8 + 4h + {0..3}
                const int qa = 16 * t + 4 * h, qb_ = qa + 8;
                Frag<T> pf, sf;
                frag_load_4x2(pf, &Pt[buf][(wid * 32 + a) * LDP + qa], &Pt[buf][(wid * 32 + a) * LDP + qb_]);
                const f32x4_t da = *reinterpret_cast<const f32x4_t*>(&Dl[buf][qa]);
                const f32x4_t db_ = *reinterpret_cast<const f32x4_t*>(&Dl[buf][qb_]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float nd = e < 4 ? da[e] : db_[e - 4];
                    frag_set(sf, e, frag_get(pf, e) * fmaf(dp[8 * t + e], scale, nd));
                }
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    Frag<T> of, qf;
                    frag_load_tr(of, Os[buf], LDV, qa, qb_, i * 32, lane);     // dO^T[d][q], accumulator k-map
                    frag_load_tr(qf, Qs[buf], LDV, qa, qb_, i * 32, lane);     // Q^T[d][q]
                    mma32(dv[i], pf, of);
                    mma32(dk[i], sf, qf);
                }
            }
        }
        if (qs + 1 < nqt) {
            sstore(buf ^ 1);
            if (qs + 2 < nqt) gload(qs + 2);
        }
        block_sync_lds();               // LDS hand-over only: prefetch loads stay in flight
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
// backward 3/3 (E-row-owned, streaming):  dE[e][d] += sum_{bh, q} dG^T[bh][e][q] Q[bh][q][d]
// =====================================================================================
// The query kernel left dG^T as tiles (query tile qt, step t) = E block cb = Lp/32 - 1 - qt + t (relative to row
// M - Lp), rows = E row, columns = query: a [128 c][32 q] operand tile of column-block group gx and query slab qs is
// the four consecutive tiles t0 .. t0 + 3, t0 = 4 gx - (Lp/32 - 1) + qs (8 KB contiguous where they exist; tiles with
// t < 0 or t > qs do not exist and count as zero).  Plain 16-byte chunk traffic: the kernel is HBM bound.
// Block = 128 rows c (4 waves x 32) x DH, one split of the (b, head) range; register accumulation over
// (bh, q) and one atomic flush per block.  Column block cb only receives queries
// q >= 32*(Lp/32 - 1 - cb); earlier slabs are skipped.
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_bwd_e_kernel(const T* __restrict__ dGT, const T* __restrict__ qkv,
                                                        float* __restrict__ dE, int B, int L, int Lp, int H, int M) {
    constexpr int CH = ET<T>::CH, LDP = 32 + CH, DB = ACfg<T, DH>::DB, LDV = ACfg<T, DH>::LDV;
    __shared__ __attribute__((aligned(16))) T Gt[2][128 * LDP];
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

    using GT = TileT<T, 128, 32>;                       // four dG^T tiles [32 c][32 q]
    chunk16 rg[GT::NPT], rq[TileT<T, 32, DH>::NPT];
    bool rgv[GT::NPT];                                  // tile exists?  applied when the chunk is stored into LDS
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const size_t dg_bh = pt_tiles(ncb, true) * 1024;
    // No branch encloses a global load or an LDS store in the steady state: tile indices are clamped, non-existent
    // tiles are zeroed by a select when they are stored (exact s_waitcnt bookkeeping).
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int bh = bh_lo + s / nq, qs = qs0 + s % nq;
        const T* src = dGT + (size_t)bh * dg_bh;
#pragma unroll
        for (int i = 0; i < GT::NPT; ++i) {
            const int c = tid + i * 256, row = c / GT::CPR, cc = (c % GT::CPR) * CH;
            const int cb = gx * 4 + (row >> 5), t = cb - (ncb - 1) + qs;
            rgv[i] = cb < ncb && t >= 0 && t <= qs;
            rg[i] = ld_chunk(src + dg_tile(qs, min(max(t, 0), qs)) * 1024 + (row & 31) * 32 + cc);
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
#pragma unroll
        for (int i = 0; i < GT::NPT; ++i) {
            const int c = tid + i * 256, row = c / GT::CPR, cc = (c % GT::CPR) * CH;
            st_chunk(&Gt[buf][row * LDP + cc], rgv[i] ? rg[i] : zero_chunk());
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
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* PT, void* dGT, int B, int L,
               int Lp, int H, int M, int causal, hipStream_t st) {
    const int nqb = (L + 127) / 128;
    const float scale = 1.f / sqrtf((float)DH);
    if (causal)
        rga_bwd_q_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (const T*)out, lse,
                                                                  (const T*)dout, (T*)dqkv, delta_ws, (T*)PT, (T*)dGT, B, L, Lp,
                                                                  H, M, scale);
    else
        rga_bwd_q_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (const T*)out, lse,
                                                                   (const T*)dout, (T*)dqkv, delta_ws, (T*)PT, (T*)dGT, B, L, Lp,
                                                                   H, M, scale);
    int rc = me_launch_status();
    if (rc) return rc;
    if (causal)
        rga_bwd_kv_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)PT, (const T*)qkv, (const T*)dout, delta_ws, (T*)dqkv,
                                                                   B, L, Lp, H, scale);
    else
        rga_bwd_kv_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)PT, (const T*)qkv, (const T*)dout, delta_ws, (T*)dqkv,
                                                                    B, L, Lp, H, scale);
    rc = me_launch_status();
    if (rc) return rc;
    const int ngx = (Lp + 127) / 128;
    // grid sweep at C2 (us per launch): 512: 131, 768: 118, 1024: 127, 1160: 108, 1536: 106, 2048: 113, 4096: 111 --
    // ~6 blocks per CU (4 resident): short blocks fill the tail left by the proportional group split
    int eblocks = 1536;
    if (eblocks > ngx * B * H) eblocks = ngx * B * H;
    if (eblocks < ngx) eblocks = ngx;
    rga_bwd_e_kernel<T, DH><<<eblocks, 256, 0, st>>>((const T*)dGT, (const T*)qkv, dE, B, L, Lp, H, M);
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
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* PT, void* dGT, int B, int L,
               int Lp, int H, int dh, int M, int causal, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse || !dout || !dqkv || !dE || !delta_ws || !PT || !dGT) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31) || (Lp & 31) || Lp < L || Lp > M) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv) ||
        !aligned16(PT) || !aligned16(dGT))
        return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((bwd_launch<T, DH>(qkv, Epk, key_pad, out, lse, dout, dqkv, dE, delta_ws, PT, dGT, B, L, Lp, H, M,
                                        causal, st)))
}

}  // extern "C"
