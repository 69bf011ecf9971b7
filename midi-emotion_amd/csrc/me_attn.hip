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
// Backward.  Nothing of the softmax is recomputed: in training mode the forward leaves, per layer, its unnormalised
// probability tiles p = exp2((s - m_t) c2) (compute type, natural [q][key] 32 x 32 tiles: the packed lower triangle
// (key tile, query tile >= key tile), the full square for the bidirectional variant) and the running maxima m_t; the
// backward kernels rebuild P = p * exp2(m_t c2 - lse log2e).  For the layer being differentiated the query-owned kernel
// additionally writes dG^T [bh][query tile qt][step t <= qt] (the skewed dS of E block eb0(qt) + t: rows = E row,
// columns = query).  Every tile is written by exactly one wave before it is read: no zero-initialisation contract.
//     dQ  = dS (K + E_skewed)             (rga_bwd_q_kernel: P tile read back by the lanes that wrote it)
//     dV = P^T dO,  dK = dS^T Q           (rga_bwd_kv_kernel, key-owned: streams the P tiles once, dS from P and V dO^T)
//     dE[e] += sum_{bh, q} dG^T[e][q] Q[q]   (rga_bwd_e_kernel, E-row-owned: plain tile stream)
// with dS = P o (dP - delta) / sqrt(dh).  Round 1 recomputed S / G / exp in the query kernel and materialised P^T and dS^T
// there (2.33 GB of HBM traffic per layer at the headline shape); dS^T is never stored now.
#include "me_common.h"
#include <type_traits>


#include "me_attn_common.h"

namespace {
using namespace me_attn;

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
// STORE_P (training): every wave also leaves its UNNORMALISED probability tile p = exp2((s - m_t) c2) -- [32 q][32 key]
// in register-image order (p_col), straight from the registers that feed the P.V product -- and the running maximum m_t it was
// taken against (raw logit units).  The backward kernels rebuild P = p * exp2(m_t c2 - lse log2e) from them instead of
// recomputing Q.K^T, the relative term and the exponential (round 2: the query-owned backward kernel was bound by exactly
// that recomputation).  Rows q >= L of a tile hold garbage: the consumers' factor is 0 there.
template <typename T, int DH, bool CAUSAL = true, bool STORE_P = false>
__global__ __launch_bounds__(256, 3) void rga_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ Epk, const uint8_t* __restrict__ key_pad,
                                                      T* __restrict__ out, float* __restrict__ lse, T* __restrict__ PT,
                                                      float* __restrict__ MT, int B, int L, int Lp, int H, int M, float scale) {
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
            // the running maximum is kept in RAW logit units (before the scale / log2e factor c2 > 0): the exponent is
            // one fma per element, exp2(s * c2 - m * c2).  Two batches of 8 ring reads (register budget: the kernel sits at
            // the 168 registers of 3 waves per SIMD).
            float mt = -INFINITY;
            const bool plain = !diag && !upper && pbits == 0u && k0 + 32 <= L;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float gv[8];
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    const float* gp = grow + ((t0 + 2 * r0 + 8 * gq) & 63);      // register quads r0/4 + gq = columns + 8 (r0/4 + gq)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[4 * gq + i] = gp[i];
                }
                if (plain) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s[r0 + j] += gv[j];
                        mt = fmaxf(mt, s[r0 + j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = r0 + j;
                        const int bk = (r & 3) + 8 * (r >> 2) + 4 * h, key = k0 + bk;
                        const bool masked = (CAUSAL && key > q) || key >= L || ((pbits >> bk) & 1u);
                        const float g = (CAUSAL || (!upper && key <= q)) ? gv[j] : 0.f;
                        const float v = masked ? -INFINITY : s[r] + g;
                        s[r] = v;
                        mt = fmaxf(mt, v);
                    }
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
            T* tile = nullptr;
            if constexpr (STORE_P) {
                // wave-uniform tile origin in SGPRs (readfirstlane: the compiler cannot prove tid >> 6 uniform), one lane
                // offset register: the forward sits exactly at its 168-register budget (3 waves per SIMD)
                const int nq32 = Lp >> 5;
                const int qt_u = __builtin_amdgcn_readfirstlane(q0 >> 5);
                T* tile_u = PT + ((size_t)bh * pt_tiles(nq32, CAUSAL) + pt_tile(kt, qt_u, nq32, CAUSAL)) * 1024;
                tile = tile_u + (a * 32 + 16 * h);
                float* mt_u = MT + ((size_t)bh * nq32 + kt) * Lp + qt_u * 32;
                mt_u[a] = m_safe;                           // both half-waves write the same value: no exec-mask branch
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {                   // O^T[d][q] += V^T[d][key] . P^T[key][q]
                Frag<T> pf; frag_from_acc(pf, s, t);
                if constexpr (STORE_P) frag_store(tile + 8 * t, pf);     // registers 8 t .. 8 t + 7 of the lane's image
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
// backward 1/3 (query-owned): delta, dQ (key part + relative part), and the dG^T tiles
// =====================================================================================
// P is NOT recomputed: the forward pass left the unnormalised tile p (register image: a lane reads back exactly the
// 16 contiguous elements of its query row that it wrote) and the running maximum m_t, so P = p * exp2(m_t c2 - lse log2e) is one
// exponential per lane and step instead of K.Q^T, the E block product, the ring skew and 16 exponentials.  Per key tile
// and wave: dP^T = V dO^T (KA atoms), dS = P o (dP - delta) / sqrt(dh), dQ^T += K^T dS^T (2 DB atoms), the skewed dS
// goes through the dG ring (LDS) and gives the relative part dQ^T += E^T dG^T (2 DB atoms) and the dG^T tile for dE.
// CAUSAL = false: backward of the bidirectional forward (MusicRegression): every key tile is visited; tiles above the
// diagonal have no relative term (no dG, no E^T product).
template <typename T> struct PHalf { typedef typename V16<T>::x8 type; static constexpr int N = 8; };     // two 16-byte loads per lane and tile
template <> struct PHalf<float> { typedef f32x4_t type; static constexpr int N = 4; };

// (186 registers, two waves per SIMD; a cap of 168 for three spills 152 bytes per lane: 411 vs 388 us per backward)
template <typename T, int DH, bool CAUSAL = true>
__global__ __launch_bounds__(256, 2) void rga_bwd_q_kernel(
    const T* __restrict__ qkv, const T* __restrict__ Epk, const T* __restrict__ out, const float* __restrict__ lse,
    const T* __restrict__ dout, T* __restrict__ dqkv, float* __restrict__ delta_ws, const T* __restrict__ PT,
    const float* __restrict__ MT, T* __restrict__ dGT, int B, int L, int Lp, int H, int M, float scale) {
    using C = ACfg<T, DH>;
    using PH = typename PHalf<T>::type;
    constexpr int PE = PHalf<T>::N, PN = 16 / PE;        // elements per piece, pieces per lane and tile
    constexpr int LDR = 72;                         // dG ring row (elements of T): 64-column ring + 8
    __shared__ __attribute__((aligned(16))) T Ks[2][32 * C::LDV];      // natural K tile, only read transposed (dQ): the transpose-read row stride
    __shared__ __attribute__((aligned(16))) T Vs[2][32 * C::LDN];      // natural V tile: 16-byte fragment reads (dP)
    __shared__ __attribute__((aligned(16))) T Ds[4][32 * LDR];          // per wave: [q][64-column ring] of dG
    // E^T fragment images of the relative blocks in use (16-bit tier): at step kt wave w needs block ebB - w + kt, i.e. the
    // block's four waves use four consecutive blocks and only ONE is new per step.  It is fetched once per block and step
    // (one 16-byte chunk per thread, two steps ahead, like the key tiles) instead of once per wave: 16 KB -> 4 KB of L1
    // requests per step.  (round 3's per-phase s_memtime sums showed the waves stalling at the ISSUE of their loads: the CU's L1 miss
    // queue, not the latency of any single load, bounds this kernel.)  5 slots: 4 in use + the one being written.
    constexpr bool ERING = sizeof(T) == 2;
    constexpr int EIMG = 2 * C::DB * 512, ENCH = EIMG * (int)sizeof(T) / 16, ENPT = (ENCH + 255) / 256, ESLOTS = 5;
    __shared__ __attribute__((aligned(16))) T Es[ERING ? ESLOTS * EIMG : 8];

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

    Frag<T> dof[C::KA];
    const size_t orow = ((size_t)b * L + q) * dm + head * DH;
    // Prologue: EVERY load of the block's start-up (dO / O rows, lse, key tiles 0 and 1, probability tiles 0 and 1) is
    // requested before the first use -- one memory round trip under load instead of three dependent ones (per-phase
    // s_memtime sums of round 3: the prologue was 18 % of the waves' time).
    Frag<T> oof[C::KA];
    row_frags<T, DH>(dof, dout + orow, row_on, h);
    row_frags<T, DH>(oof, out + orow, row_on, h);
    float lse2 = row_on ? lse[((size_t)b * H + head) * L + q] * 1.4426950408889634f : 0.f;

    f32x16_t dq[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) acc_zero(dq[i]);
    // dG ring starts zeroed: the first lo block only receives its upper-right triangle
    for (int i = lane; i < 32 * LDR; i += 64) Ds[wid][i] = ET<T>::from_f(0.f);

    chunk16 rk[TileT<T, 32, DH>::NPT], rv[TileT<T, 32, DH>::NPT];
    auto gload = [&](int kt) __attribute__((always_inline)) {
        tile_gload<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
        tile_gload<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, L - kt * 32, tid);
    };
    auto gload_full = [&](int kt) __attribute__((always_inline)) {       // tile kt entirely below L
        tile_gload_full<T, 32, DH>(rk, kb_ + (size_t)kt * 32 * ldq, ldq, tid);
        tile_gload_full<T, 32, DH>(rv, vb_ + (size_t)kt * 32 * ldq, ldq, tid);
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        tile_sstore<T, 32, DH, C::LDV>(rk, Ks[buf], tid);
        tile_sstore<T, 32, DH, C::LDN>(rv, Vs[buf], tid);
    };
    // E^T of block eb (packed relative table): A operand of dQ^T[d][q] += E^T[d][e] dG^T[e][q]
    auto et_frags = [&](Frag<T> (*f)[2], int eb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::DB; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) frag_load(f[i][t], Epk + (size_t)eb * C::PK + C::PK_B + ((i * 2 + t) * 64 + lane) * 8);
    };
    // the wave's probability tile of step kt (what the forward wrote) and its running maximum: fetched a step ahead
    const int nq32 = Lp >> 5;
    const T* const ptb = PT + (size_t)bh * pt_tiles(nq32, CAUSAL) * 1024 + a * 32 + 16 * h;
    const float* const mtb = MT + (size_t)bh * nq32 * Lp + min(q, Lp - 1);
    T* const dgb = dGT + (size_t)bh * pt_tiles(nq32, true) * 1024;
    const int kt_hi = CAUSAL ? min(my_last_kt, nq32 - 1) : nq32 - 1;     // last tile this wave owns (clamp for the prefetch)
    const int qt = min(q0 >> 5, nq32 - 1);
    // two tiles in flight per wave: tile kt + 2 is requested as soon as the registers of tile kt are free (steps are
    // unrolled in pairs, U = kt & 1 selects the register set at compile time)
    constexpr int UNR = 2;               // (four tiles in flight, UNR = 4: 224 registers, +4 %)
    PH pp[UNR][PN];
    float mtn[UNR];
    auto load_p = [&](int kt, auto u_tag) __attribute__((always_inline)) {
        constexpr int U = decltype(u_tag)::value;
        const int ktc = min(kt, kt_hi);
        const T* tp = ptb + pt_tile(ktc, qt, nq32, CAUSAL) * 1024;
#pragma unroll
        for (int g = 0; g < PN; ++g) pp[U][g] = nt_load(reinterpret_cast<const PH*>(tp + PE * g));      // the lane's own 16 elements, contiguous
        mtn[U] = mtb[(size_t)ktc * Lp];
    };
    chunk16 rk0[TileT<T, 32, DH>::NPT], rv0[TileT<T, 32, DH>::NPT];          // key tile 0 (start-up only); rk / rv receive tile 1
    tile_gload<T, 32, DH>(rk0, kb_, ldq, L, tid);
    tile_gload<T, 32, DH>(rv0, vb_, ldq, L, tid);
    if (nkt > 1) gload(1);
    load_p(0, std::integral_constant<int, 0>{});
    load_p(1, std::integral_constant<int, 1>{});
    const int eb0 = (M - 32 - q0) >> 5;
    const int ebB = (M - 32 - qb * 128) >> 5, eb_max = (M >> 5) - 1;     // wave 0's block at step 0 (>= 3); last block of the table
    chunk16 re[ENPT], re0[4][ENPT];
    auto eload = [&](chunk16* r, int eb) __attribute__((always_inline)) {     // images of block eb (clamped: steps past the table never use them)
        const T* src = Epk + (size_t)min(max(eb, 0), eb_max) * C::PK + C::PK_B;
#pragma unroll
        for (int i = 0; i < ENPT; ++i) r[i] = ld_chunk(src + (size_t)min(tid + i * 256, ENCH - 1) * (16 / sizeof(T)));
    };
    auto estore = [&](const chunk16* r, int eb) __attribute__((always_inline)) {
        T* dst = Es + (eb % ESLOTS) * EIMG;
#pragma unroll
        for (int i = 0; i < ENPT; ++i) st_chunk(dst + (size_t)min(tid + i * 256, ENCH - 1) * (16 / sizeof(T)), r[i]);
    };
    if constexpr (ERING) {
#pragma unroll
        for (int j = 0; j < 4; ++j) eload(re0[j], ebB - 3 + j);           // blocks of step 0
        eload(re, ebB + 1);                                              // wave 0's block of step 1
    }
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < C::KA; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) delta += frag_get(dof[kk], e) * frag_get(oof[kk], e);
    delta = half_sum(delta);
    if (delta_ws && row_on && h == 0) delta_ws[((size_t)b * H + head) * L + q] = delta;
    tile_sstore<T, 32, DH, C::LDV>(rk0, Ks[0], tid);
    tile_sstore<T, 32, DH, C::LDN>(rv0, Vs[0], tid);
    if constexpr (ERING) {
#pragma unroll
        for (int j = 0; j < 4; ++j) estore(re0[j], ebB - 3 + j);
    }
    __syncthreads();
    // One key tile.  MAIN = every wave of the block is strictly above its diagonal tile and tiles kt + 1, kt + 2 lie
    // entirely below L: no wave-, tile- or bounds-dependent branch encloses a global load or store (exact s_waitcnt
    // bookkeeping: vmcnt is in order).
    auto step = [&](int kt, auto u_tag, auto main_tag) __attribute__((always_inline)) {
        constexpr bool MAIN = decltype(main_tag)::value;
        constexpr int U = decltype(u_tag)::value, buf = U;
        if (MAIN || (wave_on && (!CAUSAL || kt <= my_last_kt))) {
            const int k0 = kt * 32;
            const bool upper = !CAUSAL && !MAIN && kt > my_last_kt;      // bidirectional only: above the diagonal, no relative term
            const int eb_lo = eb0 + kt;
            // (Fetching the E^T images a step ahead as well costs 32 registers and was not faster, rounds 2 and 3.)
            Frag<T> etf[C::DB][2];
            if constexpr (!ERING) { if (!upper) et_frags(etf, eb_lo); }  // f32 tier: straight from global memory, in flight during dP / dS
            const float fac = row_on ? fast_exp2(fmaf(mtn[U], c2, -lse2)) : 0.f;
            f32x16_t s, dp; acc_zero(dp);
#pragma unroll
            for (int kk = 0; kk < C::KA; ++kk) {
                Frag<T> vf;
                frag_load(vf, &Vs[buf][a * C::LDN + kk * 16 + h * 8]);
                mma32(dp, vf, dof[kk]);        // dP^T[key][q] = V[key] . dO[q]
            }
            const float nds = -delta * scale;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (ET<T>::to_f(pp[U][r / PE][r % PE]) * fac) * fmaf(dp[r], scale, nds);
            // the tile after next into the registers just consumed.  The asm pins the order: without it the compiler
            // hoists the loads above the last use of the old tile and keeps the old tile alive in copies whose v_movs
            // then wait for the newest loads (the prefetch distance collapses to one step).
            asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),
                              "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            load_p(kt + UNR, u_tag);
            __builtin_amdgcn_sched_barrier(0);
            T* drow = &Ds[wid][a * LDR];
            const int t0 = U * 32 + 31 - a + 4 * h;            // band element m sits at ring column (U * 32 + m) & 63 (the lo block of a step is the hi block of the one before)
            if (!upper) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // dG collects dS only where the relative term exists (bidirectional diagonal tile: key <= q)
                    const bool rel = CAUSAL || k0 + (r & 3) + 8 * (r >> 2) + 4 * h <= q;
                    drow[(t0 + (r & 3) + 8 * (r >> 2)) & 63] = ET<T>::from_f(rel ? s[r] : 0.f);
                }
            }
            // ---- dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> dsf; frag_from_acc(dsf, s, t);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) {
                    Frag<T> kf;                              // K^T[d][key] for the accumulator's key map
                    frag_load_tr(kf, Ks[buf], C::LDV, 16 * t + 4 * h, 16 * t + 8 + 4 * h, i * 32, lane);
                    mma32(dq[i], kf, dsf);
                }
            }
            // ---- the lo block of dG is complete now: relative part of dQ, and the block leaves as the dG^T tile
            //      (qt, t = kt) [E row m][query]: the ring rows [q][m] are read back transposed (16-bit tier:
            //      ds_read_b64_tr_b16, two 16-byte stores per lane) -- the E kernel streams these tiles as they are.
            if (!upper) {
                if constexpr (ERING) {
                    const T* eimg = Es + (eb_lo % ESLOTS) * EIMG + lane * 8;
#pragma unroll
                    for (int i = 0; i < C::DB; ++i)
#pragma unroll
                        for (int t = 0; t < 2; ++t) frag_load(etf[i][t], eimg + (i * 2 + t) * 512);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    Frag<T> dgf;
                    const T* dlo = drow + U * 32;
                    frag_load(dgf, dlo + 16 * t + 8 * h);
#pragma unroll
                    for (int i = 0; i < C::DB; ++i) mma32(dq[i], etf[i][t], dgf);
                }
                T* const dg_dst = dgb + dg_tile(q0 >> 5, kt) * 1024;
                const T* ring = &Ds[wid][U * 32];                  // lo block: ring rows q, 32 columns m, row stride LDR
                if constexpr (sizeof(T) == 2) {
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
                        // row m = 16 kh + l16, queries 8 gidx .. + 7 -> fragment-image position (dg_pos)
                        nt_store(c.v, reinterpret_cast<u32x4_t*>(dg_dst + (gidx >> 1) * 512 + (kh * 16 + l16 + 32 * (gidx & 1)) * 8));
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        const int idx = it * 64 + lane, m = idx >> 5, qq = idx & 31;
                        dg_dst[dg_pos(m, qq)] = ring[qq * LDR + m];
                    }
                }
            }
        }
        if constexpr (MAIN) {
            sstore(buf ^ 1);                      // buf^1 was last read in step kt-1 (barrier since)
            if constexpr (ERING) { estore(re, ebB + kt + 1); eload(re, ebB + kt + 2); }     // slot of block ebB + kt - 4: last read in step kt - 1
            gload_full(kt + 2);
        } else if (kt + 1 < nkt) {
            sstore(buf ^ 1);
            if constexpr (ERING) { estore(re, ebB + kt + 1); eload(re, ebB + kt + 2); }
            if (kt + 2 < nkt) gload(kt + 2);
        }
        block_sync_lds();               // LDS hand-over only: prefetch loads / tile stores stay in flight
    };
    // MAIN steps: all four waves on and off-diagonal (kt < 4 qb), tiles kt + 1, kt + 2 whole (kt + 3 <= L / 32)
    const int nmain = (qb * 128 + 96 < L) ? max(0, min(qb * 4, (L >> 5) - 2)) : 0;
    int kt = 0;
    vm_drain();                         // loop entry state = nothing in flight: the header's waits are the back edge's exact counts
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    for (; kt + 1 < nmain; kt += 2) { step(kt, U0{}, std::true_type{}); step(kt + 1, U1{}, std::true_type{}); }
    for (; kt < nkt; kt += 2) {
        step(kt, U0{}, std::false_type{});
        if (kt + 1 < nkt) step(kt + 1, U1{}, std::false_type{});
    }
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
// backward 2/3 (key-owned):  dV[key] = sum_q P[q][key] dO[q],  dK[key] = sum_q dS[q][key] Q[q]
// =====================================================================================
// Block = 128 keys (4 waves x 32) x DH.  Every step stages a 32-query slab: the forward's probability tiles
// [32 q][32 key] of the four key tiles, the dO and Q slabs [32 q][DH], -delta / sqrt(dh) and, per key tile, the factor
// exp2(m_t c2 - lse log2e) that normalises the tile (0 for rows q >= L).  Neither S nor dS is read from memory: with the
// wave's V rows resident in registers dP = dO V^T is KA macro-atoms and dS = P o (dP - delta) / sqrt(dh) a few
// multiply-adds per element -- the probability tiles are the only O(L^2) tensor this kernel streams.
// dP is accumulated as dP[q][key] (lane = key column), so that its registers carry the same (key, 8 queries) elements
// as the P fragment read (transposed, ds_read_b64_tr_b16) with the accumulator's k-map; all contraction-over-q operands
// use that map.
// frag_load_tr for a probability tile in register-image order: the lane's operand column is KEY lane & 31, which sits at
// position p_col(key) of every row; a 4-key group stays a contiguous 4-element group (key group j -> position group
// 4 (j & 1) + (j >> 1)), so the 16-bit transpose read only needs the permuted group address.
template <typename T> ME_DEV void p_frag_tr(Frag<T>& f, const T* tile, int ld, int rA, int rB, int lane) {
    const int l16 = lane & 15, j = (l16 & 3) + 4 * ((lane >> 4) & 1);
    const T* p = tile + (l16 >> 2) * ld + 4 * (4 * (j & 1) + (j >> 1));
    f.v = __builtin_shufflevector(lds_tr4(p + rA * ld), lds_tr4(p + rB * ld), 0, 1, 2, 3, 4, 5, 6, 7);
}
ME_DEV void p_frag_tr(Frag<float>& f, const float* tile, int ld, int rA, int rB, int lane) {
    const float* p = tile + p_col(lane & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.lo[e] = p[(rA + e) * ld]; f.hi[e] = p[(rB + e) * ld]; }
}

template <typename T, int DH, bool CAUSAL = true>
__global__ __launch_bounds__(256) void rga_bwd_kv_kernel(const T* __restrict__ PT, const float* __restrict__ MT,
                                                         const T* __restrict__ qkv, const T* __restrict__ dout,
                                                         const float* __restrict__ lse, const float* __restrict__ delta_ws,
                                                         T* __restrict__ dqkv, int B, int L, int Lp, int H, float scale) {
    using C = ACfg<T, DH>;
    // probability tile rows: 64 B (bf16, no padding): the four consecutive rows of a transpose read then sit on the four
    // 16-bank quarters of the LDS (a 80-byte stride wraps row 3 onto row 0's banks), and the 16-byte tile stores are linear
    constexpr int CH = ET<T>::CH, LDP = sizeof(T) == 2 ? 32 : 32 + CH, DB = C::DB, LDV = C::LDV, KA = C::KA;
    constexpr int KB = 128;
    __shared__ __attribute__((aligned(16))) T Pt[2][KB * LDP];         // four tiles [32 q][32 key], rows = w * 32 + q
    __shared__ __attribute__((aligned(16))) T Os[2][32 * LDV];         // natural dO / Q slabs [32 q][DH]
    __shared__ __attribute__((aligned(16))) T Qs[2][32 * LDV];
    // second copy of the dO slab with the natural-read stride: the 192-byte rows of Os are conflict free for the transpose reads
    // (dV) but put rows r and r + 4 on the same banks for the 16-byte fragment reads of dP (4-way conflicts)
    __shared__ __attribute__((aligned(16))) T On[2][32 * C::LDN];
    __shared__ __attribute__((aligned(16))) float Dl[2][32];           // -delta[q] / sqrt(dh)
    __shared__ __attribute__((aligned(16))) float Fs[2][4 * 32];       // per key tile: exp2(m_t c2 - lse log2e)

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
    const float c2 = scale * 1.4426950408889634f;
    const T* ptb = PT + (size_t)bh * pt_tiles(nq32, CAUSAL) * 1024;
    const float* mtb = MT + (size_t)bh * nq32 * Lp;
    const T* q_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* o_ = dout + (size_t)b * L * dm + head * DH;
    const float* dl_ = delta_ws + (size_t)bh * L;
    const float* ls_ = lse + (size_t)bh * L;

    Frag<T> vf[KA];                                                 // V rows of this wave's 32 keys (B operand of dP)
    row_frags<T, DH>(vf, q_ + 2 * dm + (size_t)(k0 + a) * ldq, wave_on && k0 + a < L, h);

    f32x16_t dk[DB], dv[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) { acc_zero(dk[i]); acc_zero(dv[i]); }

    constexpr int CPRP = 32 / CH, NPTP = KB * CPRP / 256;              // probability tiles: chunks per thread
    constexpr int CPRQ = DH / CH, NCHQ = 32 * CPRQ, NPTQ = (NCHQ + 255) / 256;   // Q / dO slab [32][DH]
    chunk16 rp[NPTP], ro[NPTQ], rq[NPTQ];
    float rd = 0.f, rl = 0.f, rm = 0.f;
    auto gload = [&](int qs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * 256, row = c / CPRP, cc = (c % CPRP) * CH;
            const int kt = min(kb * 4 + (row >> 5), nq32 - 1);
            // tiles above the diagonal do not exist in the packed triangle: clamped to the diagonal tile (never used)
            rp[i].v = nt_load(reinterpret_cast<const u32x4_t*>(ptb + pt_tile(kt, CAUSAL ? max(qs, kt) : qs, nq32, CAUSAL) * 1024 + (row & 31) * 32 + cc));
        }
        const int qv = L - qs * 32;
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * 256, row = c / CPRQ, cc = (c % CPRQ) * CH;
            const bool ok = c < NCHQ && row < qv;
            ro[i] = ok ? ld_chunk(o_ + ((size_t)qs * 32 + row) * dm + cc) : zero_chunk();
            rq[i] = ok ? ld_chunk(q_ + ((size_t)qs * 32 + row) * ldq + cc) : zero_chunk();
        }
        const int qq = min(qs * 32 + (tid & 31), L - 1);
        rd = dl_[qq];
        rl = ls_[qq];
        rm = mtb[(size_t)min(kb * 4 + ((tid >> 5) & 3), nq32 - 1) * Lp + qs * 32 + (tid & 31)];      // tid < 128: (key tile tid / 32, query tid % 32)
    };
    auto sstore = [&](int buf, int qs) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPTP; ++i) {
            const int c = tid + i * 256, row = c / CPRP, cc = (c % CPRP) * CH;
            st_chunk(&Pt[buf][row * LDP + cc], rp[i]);
        }
#pragma unroll
        for (int i = 0; i < NPTQ; ++i) {
            const int c = tid + i * 256, row = c / CPRQ, cc = (c % CPRQ) * CH;
            if (c < NCHQ) { st_chunk(&Os[buf][row * LDV + cc], ro[i]); st_chunk(&Qs[buf][row * LDV + cc], rq[i]); st_chunk(&On[buf][row * C::LDN + cc], ro[i]); }
        }
        if (tid < 32) Dl[buf][tid] = -rd * scale;
        if (tid < 128) Fs[buf][tid] = qs * 32 + (tid & 31) < L ? fast_exp2(fmaf(rm, c2, -rl * 1.4426950408889634f)) : 0.f;
    };
    gload(qs0);
    sstore(0, qs0);
    if (qs0 + 1 < nqt) gload(qs0 + 1);
    __syncthreads();
    for (int qs = qs0; qs < nqt; ++qs) {
        const int buf = (qs - qs0) & 1;
        if (wave_on && (!CAUSAL || qs * 32 + 31 >= k0)) {
            f32x16_t dp; acc_zero(dp);                              // dP[q][key] = dO[q] . V[key]
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
                Frag<T> of;
                frag_load(of, &On[buf][a * C::LDN + kk * 16 + h * 8]);
                mma32(dp, of, vf[kk]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // accumulator registers 8t .. 8t+7 of lane (key a, h) = queries 16t + 4h + {0..3} and 16t + 8 + 4h + {0..3}
                const int qa = 16 * t + 4 * h, qb_ = qa + 8;
                Frag<T> pr, pf, sf;
                p_frag_tr(pr, &Pt[buf][wid * 32 * LDP], LDP, qa, qb_, lane);             // p[q][key a], q in the k-map above
                const f32x4_t da = *reinterpret_cast<const f32x4_t*>(&Dl[buf][qa]);
                const f32x4_t db_ = *reinterpret_cast<const f32x4_t*>(&Dl[buf][qb_]);
                const f32x4_t fa = *reinterpret_cast<const f32x4_t*>(&Fs[buf][wid * 32 + qa]);
                const f32x4_t fb = *reinterpret_cast<const f32x4_t*>(&Fs[buf][wid * 32 + qb_]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float nd = e < 4 ? da[e] : db_[e - 4];
                    const float pn = frag_get(pr, e) * (e < 4 ? fa[e] : fb[e - 4]);      // P = p * exp2(m_t c2 - lse log2e)
                    frag_set(pf, e, pn);
                    frag_set(sf, e, pn * fmaf(dp[8 * t + e], scale, nd));
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
            sstore(buf ^ 1, qs + 1);
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
    constexpr int DB = ACfg<T, DH>::DB, LDV = ACfg<T, DH>::LDV;
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

    // The wave's dG^T tile of a step is its own A operand, fragment for fragment: lane (a, h) needs row a, queries
    // 16 t + 8 h .. + 7 -- 16 contiguous bytes (bf16) of the tile image (dg_pos).  It comes straight from global memory into the
    // registers, RG - 1 steps ahead (a wave reads a whole 2 KB tile with two instructions); only the Q slab, which all
    // four waves read transposed, goes through LDS.  (Round 1 staged the four tiles through LDS as well: 32 KB per
    // block = 5 blocks per CU and a second, mostly idle round of blocks; now 12 KB.)
    // The (bh, qs) cursors advance by increments: no division in the loop (it was ~100 SALU instructions per step).
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const size_t dg_bh = pt_tiles(ncb, true) * 1024;
    const int tshift = cbw - (ncb - 1);                 // tile index of the wave at slab qs: t = qs + tshift
    constexpr int RG = 3;          // ring of dG^T fragment sets: steps s .. s + RG - 1 (RG - 1 loads in flight)
    Frag<T> gq[RG][2];
    chunk16 rq[TileT<T, 32, DH>::NPT];
    int l_bh = bh_lo, l_qs = qs0, l_left = nsteps;      // cursor of the Q slab loads
    auto qload = [&]() __attribute__((always_inline)) {
        const T* qsrc = qkv + ((size_t)(l_bh / H) * L + l_qs * 32) * ldq + (l_bh % H) * DH;
        using QT = TileT<T, 32, DH>;
#pragma unroll
        for (int i = 0; i < QT::NPT; ++i) {              // rows past L repeat the last row: their dG^T columns are exact zeros
            const int c = min(tid + i * 256, QT::NCH - 1);
            rq[i] = ld_chunk(qsrc + (size_t)min(c / QT::CPR, L - 1 - l_qs * 32) * ldq + (c % QT::CPR) * ET<T>::CH);
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {   // past the last step the cursor stays on it (harmless re-loads)
        if (l_left > 1) {
            --l_left;
            if (++l_qs == nqt) { l_qs = qs0; ++l_bh; }
        }
    };
    // The dG^T stream is the HBM stream (the Q slabs are shared by the 8 row groups through L2): it runs RG - 1 steps
    // ahead with its own cursor; the Q slab of step s + 2 is fetched while step s is multiplied.
    int g_bh = bh_lo, g_qs = qs0, g_left = nsteps;      // cursor of the dG^T loads
    auto gload = [&](Frag<T>* g) __attribute__((always_inline)) {
        const T* src = dGT + (size_t)g_bh * dg_bh + dg_tile(g_qs, min(max(g_qs + tshift, 0), g_qs)) * 1024 + lane * 8;
        frag_load_nt(g[0], src);
        frag_load_nt(g[1], src + 512);
        if (g_left > 1) {                               // past the last step the cursor stays on it (harmless re-loads)
            --g_left;
            if (++g_qs == nqt) { g_qs = qs0; ++g_bh; }
        }
    };
    // prologue: Q slab 0 -> LDS, Q slab 1 -> registers; dG^T of steps 0 .. RG - 2 -> ring
#pragma unroll
    for (int u = 0; u < RG - 1; ++u) gload(gq[u]);
    qload(); advance();
    tile_sstore<T, 32, DH, LDV>(rq, Qs[0], tid);
    qload(); advance();
    __syncthreads();
    vm_drain();
    int c_qs = qs0;                                     // slab of the step being multiplied
    auto step = [&](int buf, Frag<T>* gcur, Frag<T>* gfar) __attribute__((always_inline)) {
        gload(gfar);                                    // step s + RG - 1
        if (wave_on && c_qs >= my_qmin) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    Frag<T> qf;
                    frag_load_tr(qf, Qs[buf], LDV, 16 * t + 8 * h, 16 * t + 8 * h + 4, i * 32, lane);     // Q^T[d][q]
                    mma32(acc[i], gcur[t], qf);
                }
            }
        }
        tile_sstore<T, 32, DH, LDV>(rq, Qs[buf ^ 1], tid);      // Q slab of step s + 1 (past the last step: a harmless re-store)
        qload();                                                // Q slab of step s + 2
        advance();
        if (++c_qs == nqt) c_qs = qs0;
        block_sync_lds();               // LDS hand-over only: prefetch loads stay in flight
    };
    for (int s = 0; s < nsteps; s += RG) {              // unrolled over the ring so that the fragment indices are static
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            if (s + u >= nsteps) break;
            step((s + u) & 1, gq[u], gq[(u + RG - 1) % RG]);
        }
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
int fwd_launch(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, void* PT, float* MT, int B,
               int L, int H, int M, int causal, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && DH == 64) {
        if (causal) return me_attn64::fwd_launch<T>(qkv, Epk, key_pad, out, lse, PT, MT, B, L, H, M, st);
    }
    const int nqb = (L + 127) / 128, Lp = ((L + 31) / 32) * 32;
    const float scale = 1.f / sqrtf((float)DH);
    const dim3 grid(B * H * nqb);
#define ME_FWD(CA, SP) rga_fwd_kernel<T, DH, CA, SP><<<grid, 256, 0, st>>>((const T*)qkv, (const T*)Epk, key_pad, (T*)out, lse, (T*)PT, MT, B, L, Lp, H, M, scale)
    if (causal) { if (PT) ME_FWD(true, true); else ME_FWD(true, false); }
    else { if (PT) ME_FWD(false, true); else ME_FWD(false, false); }
#undef ME_FWD
    return me_launch_status();
}

template <typename T, int DH>
int bwd_launch(const void* qkv, const void* Epk, const void* out, const float* lse, const void* dout, void* dqkv, float* dE,
               float* delta_ws, const void* PT, const float* MT, void* dGT, int B, int L, int Lp, int H, int M, int causal,
               hipStream_t st, int phases = 7) {
    // phases (me_rga_bwd_phases): bit 0 = query-owned kernel (dQ, delta, dG^T), bit 1 = key-owned kernel (dK, dV; needs delta),
    // bit 2 = E-row-owned kernel (dE; needs dG^T).  The kernels of bits 1 and 2 are independent of each other: a caller with
    // two streams may run them side by side (ops.rga_bwd).
    // One launch of each kernel over the whole batch.  Splitting the batch so that a chunk's probability / dG^T tiles
    // stay in the 256 MB Infinity Cache between the three kernels was measured and is slower (B = 32 in chunks of
    // 16 / 8 / 4: 449 / 524 / 898 us against 424 us): these kernels are latency bound, not HBM bound, and smaller
    // launches leave CUs idle.
    const int nqb = (L + 127) / 128;
    const float scale = 1.f / sqrtf((float)DH);
    int rc = 0;
    float* const delta_q = delta_ws;
    if (phases & 1) {
    if (causal)
        rga_bwd_q_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, (const T*)out, lse, (const T*)dout,
                                                                  (T*)dqkv, delta_q, (const T*)PT, MT, (T*)dGT, B, L, Lp, H, M, scale);
    else
        rga_bwd_q_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)Epk, (const T*)out, lse, (const T*)dout,
                                                                   (T*)dqkv, delta_q, (const T*)PT, MT, (T*)dGT, B, L, Lp, H, M, scale);
    rc = me_launch_status();
    if (rc) return rc;
    }
    if (phases & 2) {
    if (causal)
        rga_bwd_kv_kernel<T, DH, true><<<B * H * nqb, 256, 0, st>>>((const T*)PT, MT, (const T*)qkv, (const T*)dout, lse, delta_ws,
                                                                   (T*)dqkv, B, L, Lp, H, scale);
    else
        rga_bwd_kv_kernel<T, DH, false><<<B * H * nqb, 256, 0, st>>>((const T*)PT, MT, (const T*)qkv, (const T*)dout, lse, delta_ws,
                                                                    (T*)dqkv, B, L, Lp, H, scale);
    rc = me_launch_status();
    if (rc) return rc;
    }
    if (!(phases & 4)) return rc;
    const int ngx = (Lp + 127) / 128;
    // grid sweep at C2 (us per launch, round 2 kernel): 768: 85.8, 1024: 90.8, 1280: 82.5, 1536: 85.4, 2048: 85.2, 4096: 85.9 -- flat:
    // the launch is bound by the per-step latency chain (Q slab -> LDS -> barrier -> transpose reads -> 4 MFMAs), not by the tail
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
    } else if (dtype == ME_F16) {                                                \
        if (dh == 64) { typedef f16_t T; constexpr int DH = 64; return CALL; }   \
        if (dh == 48) { typedef f16_t T; constexpr int DH = 48; return CALL; }   \
        if (dh == 32) { typedef f16_t T; constexpr int DH = 32; return CALL; }   \
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

int me_rga_fwd(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, void* PT, float* MT, int B,
               int L, int H, int dh, int M, int causal, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse || ((PT == nullptr) != (MT == nullptr))) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31)) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out) || !aligned16(PT)) return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((fwd_launch<T, DH>(qkv, Epk, key_pad, out, lse, PT, MT, B, L, H, M, causal, st)))
}

int me_rga_bwd(const void* qkv, const void* Epk, const void* out, const float* lse, const void* dout, void* dqkv, float* dE,
               float* delta_ws, const void* PT, const float* MT, void* dGT, int B, int L, int Lp, int H, int dh, int M,
               int causal, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse || !dout || !dqkv || !dE || !delta_ws || !PT || !MT || !dGT) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31) || Lp != ((L + 31) / 32) * 32 || Lp > M) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv) ||
        !aligned16(PT) || !aligned16(dGT))
        return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((bwd_launch<T, DH>(qkv, Epk, out, lse, dout, dqkv, dE, delta_ws, PT, MT, dGT, B, L, Lp, H, M, causal, st)))
}

int me_rga_bwd_phases(const void* qkv, const void* Epk, const void* out, const float* lse, const void* dout, void* dqkv, float* dE,
                      float* delta_ws, const void* PT, const float* MT, void* dGT, int B, int L, int Lp, int H, int dh, int M,
                      int causal, int phases, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !Epk || !out || !lse || !dout || !dqkv || !dE || !delta_ws || !PT || !MT || !dGT) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31) || Lp != ((L + 31) / 32) * 32 || Lp > M || phases < 1 || phases > 7) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(Epk) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv) ||
        !aligned16(PT) || !aligned16(dGT))
        return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((bwd_launch<T, DH>(qkv, Epk, out, lse, dout, dqkv, dE, delta_ws, PT, MT, dGT, B, L, Lp, H, M, causal, st, phases)))
}

}  // extern "C"

