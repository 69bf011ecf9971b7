// Relative global attention, hot-path instantiation: 16-bit storage (bf16 or f16), head dim 64, causal (BASELINE configs 2-4).
//
// Same algorithm and the same saved-tile formats as the generic kernels of me_attn.hip (which remain the f32 tier, the
// other head dims and the bidirectional variant); what changes is the step: a wave still owns 32 queries, but one step
// covers 64 keys (two 32 x 32 tiles A, B).  What that buys (round 3, from the ISA of the 32-key kernels: 282 VALU + 124
// SALU instructions around 12 MFMAs per step):
//   * one barrier, one running-maximum update / rescale test, one set of pointer updates per 64 keys;
//   * the E-block parity of a tile is fixed by its position in the step (tile A: block parity of eb0, tile B: the
//     other one), so every address of the G ring -- block writes, halo, skewed band reads -- is loop invariant:
//     no per-step ring address arithmetic at all;
//   * two independent MFMA chains per phase (S_A | S_B, G_mid | G_hi, 8 P.V atoms) instead of one dependent chain.
// LDS: K tiles [64][72], V tiles [64][96] (transpose-read stride), both double buffered, + four wave-private G rings
// = 78 KB per block -> two blocks (two waves per SIMD) per CU, 256 registers per wave.
#include "me_attn_common.h"

namespace me_attn64 {
using namespace me_attn;


constexpr int DH = 64, KA = 4, DB = 2;
constexpr int LDK = 72;                        // K tile row (elements): 144 B, conflict-free ds_read_b128 fragments
constexpr int LDV = 96;                        // V tile row: 192 B, conflict-free ds_read_b64_tr_b16
constexpr int LDR = 68;                        // G ring row (floats): 64-column ring + 4 halo columns
constexpr int K_BYTES = 64 * LDK * 2;          // 9216
constexpr int V_BYTES = 64 * LDV * 2;          // 12288
constexpr int G_BYTES = 32 * LDR * 4;          // 8704 per wave
constexpr int NW = 4;                         // waves per block: 128 queries, two blocks per CU (256-query blocks of 8 waves measured 2-10 % slower: profiles/r05_attn_qb256.txt)
constexpr int FWD_LDS = 2 * K_BYTES + 2 * V_BYTES + NW * G_BYTES;      // 77824

// =====================================================================================
// forward
// =====================================================================================
// Step s covers key tiles 2 s (A) and 2 s + 1 (B).  With X = eb0 + 2 s the lo block of tile A: the ring holds block X
// (slot X & 1) on entry; G(X + 1) goes to the other slot, tile A reads its band (blocks X, X + 1), G(X + 2) then overwrites
// block X and tile B reads (X + 1, X + 2).  LDS instructions of one wave execute in order, so the overwrite needs no wait.
// MAIN steps (all four waves strictly below their diagonal in both tiles, no padded key in the sequence, the tiles of
// step s + 2 entirely below L) contain no branch around a memory instruction and no per-element predicate; everything
// else (diagonal tiles, ragged ends, pad masks) runs the general per-tile path.
template <typename T, bool STORE_P>
__global__ __launch_bounds__(64 * NW, 2) void rga_fwd64_kernel(const T* __restrict__ qkv, const T* __restrict__ Epk,
                                                        const uint8_t* __restrict__ key_pad, T* __restrict__ out,
                                                        float* __restrict__ lse, T* __restrict__ PT, float* __restrict__ MT,
                                                        int B, int L, int Lp, int H, int M, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the rings first: their 8 + 4 loop-invariant lane addresses then need no base constant (ds_read2_b32 offsets reach 1 KB)
    float* const Gsm = reinterpret_cast<float*>(smem);                                   // [NW][32 * LDR]
    T* const Vsm = reinterpret_cast<T*>(smem + NW * G_BYTES);                            // [2][64 * LDV]
    T* const Ksm = reinterpret_cast<T*>(smem + NW * G_BYTES + 2 * V_BYTES);              // [2][64 * LDK]
    constexpr int QB = 32 * NW;                                                          // queries per block
    constexpr int NI = 8 / NW;                                                           // 16-byte chunks per thread, operand and step

    const int tid = threadIdx.x, lane = tid & 63, a = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int BH = B * H, nqb = (L + QB - 1) / QB;
    const int bh = blockIdx.x % BH, qb = nqb - 1 - (int)(blockIdx.x / BH);             // heavy q-blocks first
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const T* qb_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* kb_ = qb_ + dm;
    const T* vb_ = qb_ + 2 * dm;
    const int q0 = qb * QB + wid * 32;
    const int q = q0 + a;
    const bool wave_on = q0 < L;
    const int nkt = min((L + 31) / 32, qb * NW + NW);
    const int nst = (nkt + 1) >> 1;
    const int my_last_kt = qb * NW + wid;                 // diagonal tile of this wave
    const float c2 = scale * 1.4426950408889634f;         // logits are kept in log2 units
    const int nE = M >> 5;

    // ---- K / V tile stream: 64 rows x 128 B per operand and step = two 16-byte chunks per thread.  The pad flags of the
    // step's 64 keys travel with it: EVERY wave loads all 64 of them (one byte per lane), so the step's pad mask is a
    // wave-local ballot -- no scan of the sequence, no LDS flags, no block-wide reduction.
    chunk16 rk[NI], rv[NI];
    uint32_t rp = 0;
    const int lrow = tid >> 3, lcc = (tid & 7) * 8;
    const uint8_t* kp_ = key_pad ? key_pad + (size_t)b * L : reinterpret_cast<const uint8_t*>(qkv);     // valid dummy when there is no mask
    const uint32_t kp_on = key_pad ? 0xffu : 0u;
    auto gload = [&](int s) __attribute__((always_inline)) {            // rows past L are zeros
        const int k0 = s * 64;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = k0 + lrow + 32 * i;
            rk[i] = row < L ? ld_chunk(kb_ + (size_t)row * ldq + lcc) : zero_chunk();
            rv[i] = row < L ? ld_chunk(vb_ + (size_t)row * ldq + lcc) : zero_chunk();
        }
        rp = kp_[min(k0 + lane, L - 1)];                                 // keys >= L are masked by the bound test anyway
    };
    auto gload_full = [&](int s) __attribute__((always_inline)) {       // both tiles entirely below L: no predicate
        const T* kp2 = kb_ + (size_t)(s * 64 + lrow) * ldq + lcc;
        const T* vp2 = vb_ + (size_t)(s * 64 + lrow) * ldq + lcc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            rk[i] = ld_chunk(kp2 + (size_t)(32 * i) * ldq);
            rv[i] = ld_chunk(vp2 + (size_t)(32 * i) * ldq);
        }
        rp = kp_[s * 64 + lane];
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        T* kd = Ksm + buf * (64 * LDK) + lrow * LDK + lcc;
        T* vd = Vsm + buf * (64 * LDV) + lrow * LDV + lcc;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            st_chunk(kd + 32 * i * LDK, rk[i]);
            st_chunk(vd + 32 * i * LDV, rv[i]);
        }
    };
    auto pad_mask = [&]() __attribute__((always_inline)) -> unsigned long long { return __ballot((rp & kp_on) != 0u); };

    // ---- G ring (wave private): row = query a, block X at columns (X & 1) * 32 + m; columns 64..67 mirror 0..3
    float* const grow = Gsm + wid * (32 * LDR) + a * LDR;
    const int eb0 = (M - 32 - q0) >> 5;
    const int pX = eb0 & 1;                               // slot of the even-offset blocks X = eb0 + 2 s
    float* const wB = grow + pX * 32 + 4 * h;             // block X + 2 (and the prologue's block eb0): slot pX
    float* const wA = grow + (pX ^ 1) * 32 + 4 * h;       // block X + 1: the other slot
    float* const hB = (pX | h) ? wB : grow + 64;          // halo copy of columns 0..3 (lanes that do not own them rewrite their quad)
    float* const hA = ((pX ^ 1) | h) ? wA : grow + 64;
    const float* rA[4];                                   // band reads: register quad j of tile A / B
    const float* rB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cA = (pX * 32 + 31 - a + 4 * h + 8 * j) & 63;
        rA[j] = grow + cA;
        rB[j] = grow + (cA ^ 32);
    }
    // diagonal tile: register r holds key b(r) = (r & 3) + 8 (r >> 2) + 4 h of the tile; masked where b > a
    uint32_t causal16 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) causal16 |= ((r & 3) + 8 * (r >> 2) + 4 * h > a ? 1u : 0u) << r;
    auto ring_write = [&](const f32x16_t& g, float* w, float* hh) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4_t*>(w + 8 * gq) = (f32x4_t){g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
        *reinterpret_cast<f32x4_t*>(hh) = (f32x4_t){g[0], g[1], g[2], g[3]};
    };
    auto e_frags = [&](Frag<T>* f, int eb) __attribute__((always_inline)) {
        const T* src = Epk + (size_t)min(eb, nE - 1) * ACfg<T, DH>::PK + lane * 8;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) frag_load(f[kk], src + kk * 512);
    };
    Frag<T> qf[KA];
    auto g_mma = [&](f32x16_t& g, const Frag<T>* ef) __attribute__((always_inline)) {
        acc_zero(g);
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) mma32(g, ef[kk], qf[kk]);                 // G^T[m][q] = E[32 X + m] . Q[q]
    };
    auto s_mma = [&](f32x16_t& s, const T* Kt) __attribute__((always_inline)) {
        acc_zero(s);
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
            Frag<T> kf;
            frag_load(kf, Kt + a * LDK + kk * 16 + h * 8);
            mma32(s, kf, qf[kk]);                                                  // S^T[key][q]
        }
    };
    f32x16_t o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) acc_zero(o[i]);
    float m_run = -INFINITY, l_run = 0.f;
    // P^T tile -> B operand of O^T[d][q] += V^T[d][key] P^T[key][q]; training: the tile leaves as the lane's register image
    auto pv = [&](const f32x16_t& p, const T* Vt, T* ptile, bool store) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Frag<T> pf;
            frag_from_acc(pf, p, t);
            if constexpr (STORE_P) {
                // written once, read by the backward much later: streaming (non-temporal) stores
                if (store) __builtin_nontemporal_store(pf.v, reinterpret_cast<typename V16<T>::x8*>(ptile + 8 * t));
            }
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                Frag<T> vf;
                frag_load_tr(vf, Vt, LDV, 16 * t + 4 * h, 16 * t + 8 + 4 * h, i * 32, lane);
                mma32(o[i], vf, pf);
            }
        }
    };
    const int nq32 = Lp >> 5;
    const int qt = q0 >> 5;
    T* const pt_lane = STORE_P ? PT + (size_t)bh * pt_tiles(nq32, true) * 1024 + (a * 32 + 16 * h) : nullptr;
    float* const mt_lane = STORE_P ? MT + (size_t)bh * nq32 * Lp + min(qt * 32 + a, Lp - 1) : nullptr;

    // ---- prologue: every load of the block's first round trip is requested before anything waits
    Frag<T> efA[KA], efB[KA];
    gload(0);
    row_frags<T, DH>(qf, qb_ + (size_t)q * ldq, wave_on && q < L, h);
    if (wave_on) {
        Frag<T> ef0[KA];
        e_frags(ef0, eb0);
        e_frags(efA, eb0 + 1);
        e_frags(efB, eb0 + 2);
        f32x16_t g;
        g_mma(g, ef0);
        ring_write(g, wB, hB);
    }
    sstore(0);
    unsigned long long pb_cur = pad_mask();
    if (nst > 1) gload(1);
    __syncthreads();

    // ---- two tiles with one running-maximum update and no per-element predicate.  TAIL: the wave may be at its diagonal.
    // The keys a diagonal tile must not see (key > q) are exactly its band elements m >= 32, i.e. its hi block -- which the
    // diagonal tile does not need: a block of -inf in that ring slot masks them through the ordinary s += g (infA / infB);
    // a tile above the diagonal (lo and hi = -inf) turns into exact zeros and only its stores are skipped (storeB).
    auto body_pair = [&](int s, const T* Kt, const T* Vt, auto tail_tag, bool infA, bool infB, bool storeB) __attribute__((always_inline)) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        f32x16_t g1, g2, s0, s1;
        if (TAIL && infA) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = -INFINITY;
        } else g_mma(g1, efA);
        s_mma(s0, Kt);
        if (TAIL && infB) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g2[r] = -INFINITY;
        } else g_mma(g2, efB);
        s_mma(s1, Kt + 32 * LDK);
        const int X = eb0 + 2 * s;
        e_frags(efA, X + 3);
        e_frags(efB, X + 4);
        ring_write(g1, wA, hA);
        float gv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[4 * j + i] = rA[j][i];
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] += gv[r];
        ring_write(g2, wB, hB);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[4 * j + i] = rB[j][i];
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] += gv[r];
        float mt = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, fmaxf(s0[r], s1[r]));
        mt = half_max(mt);
        const float m_new = fmaxf(m_run, mt);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = fast_exp2((m_run - m_safe) * c2);
        const float nm = -m_safe * c2;
        {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t c2v = {c2, c2}, nmv = {nm, nm};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                        // v_pk_fma_f32: two exponents per instruction
                f32x2_t x0 = {s0[r], s0[r + 1]}, x1 = {s1[r], s1[r + 1]};
                x0 = __builtin_elementwise_fma(x0, c2v, nmv);
                x1 = __builtin_elementwise_fma(x1, c2v, nmv);
                s0[r] = fast_exp2(x0[0]); s0[r + 1] = fast_exp2(x0[1]);
                s1[r] = fast_exp2(x1[0]); s1[r + 1] = fast_exp2(x1[1]);
            }
        }
        float u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = s0[r] + s1[r];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int r = 0; r < w; ++r) u[r] += u[r + w];
        l_run = l_run * alpha + u[0];
        if (__any(m_new != m_run)) {                 // running maxima settle quickly: most steps skip the rescale
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        m_run = m_new;
        T* tileA = nullptr;
        T* tileB = nullptr;
        if constexpr (STORE_P) {
            const size_t ta = pt_tile(2 * s, qt, nq32, true);
            tileA = pt_lane + ta * 1024;
            tileB = tileA + (size_t)(nq32 - 2 * s - 1) * 1024;       // pt_tile(kt + 1, qt) - pt_tile(kt, qt) = nq - kt - 1
            float* mp = mt_lane + (size_t)(2 * s) * Lp;
            mp[0] = m_safe;                                          // both tiles were taken against the same maximum
            if (!TAIL || storeB) mp[Lp] = m_safe;
        }
        pv(s0, Vt, tileA, true);
        pv(s1, Vt + 32 * LDV, tileB, !TAIL || storeB);
    };

    // ---- MAIN step: every wave of the block runs body_pair; no branch around any memory instruction
    unsigned long long pb_nxt = 0;
    auto step_main = [&](int s) __attribute__((always_inline)) {
        const int buf = s & 1;
        // tiles of step s + 1 (registers) -> the other buffer (free since the barrier); tiles of step s + 2 -> registers
        sstore(buf ^ 1);
        pb_nxt = pad_mask();
        gload_full(s + 2);
        body_pair(s, Ksm + buf * (64 * LDK), Vsm + buf * (64 * LDV), std::false_type{}, false, false, true);
        block_sync_lds();
    };

    // ---- general tile: own running-maximum update; mask word mk (bit r = element r masked): diagonal, keys >= L, padded keys
    auto tile_gen = [&](int kt, const T* Kt, const T* Vt, uint32_t pb32, const Frag<T>* ef, float* w, float* hh,
                        const float* const* rr) __attribute__((always_inline)) {
        if (!(wave_on && kt <= my_last_kt && kt < nkt)) return;
        const int k0 = kt * 32;
        const bool diag = kt == my_last_kt;
        if (!diag) {
            f32x16_t g;
            g_mma(g, ef);
            ring_write(g, w, hh);
        }
        f32x16_t s;
        s_mma(s, Kt);
        uint32_t mk = diag ? causal16 : 0u;
        if (k0 + 32 > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mk |= (k0 + (r & 3) + 8 * (r >> 2) + 4 * h >= L ? 1u : 0u) << r;
        }
        if (pb32) {
            const uint32_t ph = pb32 >> (4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) mk |= ((ph >> ((r & 3) + 8 * (r >> 2))) & 1u) << r;
        }
        float mt = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * j + i;
                const float v = ((mk >> r) & 1u) ? -INFINITY : s[r] + rr[j][i];
                s[r] = v;
                mt = fmaxf(mt, v);
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
        if (__any(m_new != m_run)) {
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        m_run = m_new;
        T* tile = nullptr;
        if constexpr (STORE_P) {
            tile = pt_lane + pt_tile(kt, qt, nq32, true) * 1024;
            mt_lane[(size_t)kt * Lp] = m_safe;
        }
        pv(s, Vt, tile, true);
    };
    auto step_gen = [&](int s) __attribute__((always_inline)) {
        const int buf = s & 1;
        const T* Kt = Ksm + buf * (64 * LDK);
        const T* Vt = Vsm + buf * (64 * LDV);
        if (s + 1 < nst) {
            sstore(buf ^ 1);
            pb_nxt = pad_mask();
            if (s + 2 < nst) gload(s + 2);
        }
        const int dgn = my_last_kt - 2 * s;       // 0: tile A is the wave's diagonal tile, 1: tile B, < 0: nothing left for this wave
        if (wave_on && dgn >= 0 && pb_cur == 0ull && 64 * s + 32 <= L && (64 * s + 64 <= L || dgn == 0)) {
            // no padded key, both tiles inside the sequence (or tile B above the diagonal anyway): predicate-free path
            body_pair(s, Kt, Vt, std::true_type{}, dgn == 0, dgn <= 1, dgn >= 1);
        } else {
            tile_gen(2 * s, Kt, Vt, (uint32_t)pb_cur, efA, wA, hA, rA);
            tile_gen(2 * s + 1, Kt + 32 * LDK, Vt + 32 * LDV, (uint32_t)(pb_cur >> 32), efB, wB, hB, rB);
            if (wave_on && 2 * s + 2 <= my_last_kt) {
                const int X = eb0 + 2 * s;
                e_frags(efA, X + 3);
                e_frags(efB, X + 4);
            }
        }
        block_sync_lds();
    };

    // MAIN: s < 2 qb (both tiles strictly below every wave's diagonal), all four waves on, tiles of step s + 2 whole,
    // no padded key among the step's 64 (the loop ends at the first step that has one)
    const int nmain = (qb * QB + QB - 32 < L) ? max(0, min((NW / 2) * qb, (L >> 6) - 2)) : 0;
    int s = 0;
    vm_drain();
    for (; s < nmain && pb_cur == 0ull; ++s) { step_main(s); pb_cur = pb_nxt; }
    for (; s < nst; ++s) { step_gen(s); pb_cur = pb_nxt; }

    if (!wave_on) return;
    // ---- write-out: O^T (lane = query, registers = head-dim rows) is normalised, staged through the wave's ring area as
    // [32 q][64 d] and leaves as full 128-byte rows (per-lane 8-byte pieces at a 1 KB row stride touch 32 lines per store)
    const float l_tot = half_sum(l_run);
    const float inv = 1.f / l_tot;
    if (h == 0 && q < L) lse[((size_t)b * H + head) * L + q] = (m_run * c2 + log2f(l_tot)) * 0.6931471805599453f;
    T* const stg = reinterpret_cast<T*>(Gsm + wid * (32 * LDR));              // 8704 B per wave; rows of 72 elements
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            st4<T>(stg + a * 72 + i * 32 + 8 * gq + 4 * h, o[i][4 * gq] * inv, o[i][4 * gq + 1] * inv, o[i][4 * gq + 2] * inv,
                   o[i][4 * gq + 3] * inv);
    T* const ob = out + ((size_t)b * L + q0) * dm + head * DH;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const chunk16 v = ld_chunk(stg + row * 72 + ch * 8);
        if (q0 + row < L) st_chunk(ob + (size_t)row * dm + ch * 8, v);
    }

}


static void set_lds_limit(const void* fn, int bytes, bool* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 16) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); return; }
    if (done[dev]) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);     // the attribute is per device
    done[dev] = true;
}

template <typename T>
int fwd_launch(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, void* PT, float* MT, int B,
               int L, int H, int M, hipStream_t st) {
    const int nqb = (L + 32 * NW - 1) / (32 * NW), Lp = ((L + 31) / 32) * 32;
    const float scale = 1.f / sqrtf((float)DH);
    const dim3 grid(B * H * nqb);
    static bool done_t[16] = {false}, done_i[16] = {false};
    if (PT) {
        set_lds_limit((const void*)rga_fwd64_kernel<T, true>, FWD_LDS, done_t);
        rga_fwd64_kernel<T, true><<<grid, 64 * NW, FWD_LDS, st>>>((const T*)qkv, (const T*)Epk, key_pad, (T*)out, lse, (T*)PT, MT, B, L, Lp, H, M,
                                                                 scale);
    } else {
        set_lds_limit((const void*)rga_fwd64_kernel<T, false>, FWD_LDS, done_i);
        rga_fwd64_kernel<T, false><<<grid, 64 * NW, FWD_LDS, st>>>((const T*)qkv, (const T*)Epk, key_pad, (T*)out, lse, nullptr, nullptr, B, L,
                                                                  Lp, H, M, scale);
    }
    return me_launch_status();
}
template int fwd_launch<bf16_t>(const void*, const void*, const uint8_t*, void*, float*, void*, float*, int, int, int, int, hipStream_t);
template int fwd_launch<f16_t>(const void*, const void*, const uint8_t*, void*, float*, void*, float*, int, int, int, int, hipStream_t);

}  // namespace me_attn64
