// Relative global attention (Music-Transformer RGA) for gfx950, flash style:
// the L x L score matrix, the L x L relative term and the float masks of the
// reference (music_multi.py:215-231) are never materialised.
//
//   logits[q,key] = ( Q[q].K[key] + Q[q].E[M-1-(q-key)] ) / sqrt(dh),  key <= q, key not pad
//
// Tiling: 32 x 32 (q x key) tiles, one wavefront per 32 rows, MFMA macro-atoms
// from me_common.h.  For a tile (q0, k0) the relative term needs the 63 rows
// E[e_lo .. e_lo+62], e_lo = M-32-q0+k0 (a multiple of 32), i.e. two aligned
// 32-row blocks of E ("lo", "hi").  G = Q.E_blk^T is an ordinary MFMA product;
// the Toeplitz "skew"  Srel[a][b] = G[a][31-a+b]  is one trip through a
// wave-private LDS buffer (written in accumulator layout, read back with a
// per-lane shifted address -- conflict free both ways).  Consecutive key tiles
// share a block (hi of step t == lo of step t+1), so each step computes one new
// block only.
//
// Layout trick: the forward and dQ kernels compute the TRANSPOSED score tile
// S^T[key][q] = mfma(K, Q); in the accumulator layout each lane then owns one
// query column and 16 key rows, so (i) softmax statistics are lane-local (+1
// half-wave exchange), (ii) P^T packs straight into the B operand of
// O^T[d][q] += V^T[d][key] P^T[key][q] with no cross-lane traffic, and (iii)
// the per-row rescale of O is a per-lane scalar.  The dK/dV kernel uses the
// untransposed tile for the same reason with key as the lane-owned index.
#include "me_common.h"

#ifndef ME_ABL
#define ME_ABL 0
#endif

namespace {

constexpr int LDG = 36;   // G ring row (floats): 32 + 4 -> conflict-free b128 writes, b32 skew reads
constexpr int LDD = 68;   // dG skew buffer row (floats)
constexpr int LDT = 36;   // transposed tile row (elements): 32 + 4

template <typename T, int DH> struct ACfg {
    static constexpr int CH = ET<T>::CH;
    static constexpr int KA = DH / 16;       // contraction atoms over the head dim
    static constexpr int DB = DH / 32;       // 32-wide blocks of the head dim
    static constexpr int LDN = DH + CH;      // natural tile row (elements)
    static constexpr int NCHUNK = 32 * DH / CH;   // 16-byte chunks per 32 x DH tile
    static constexpr int NPT = (NCHUNK + 255) / 256;
};

// 32 x DH tile, "row-walking" thread map (lane -> row): used when the tile is also
// needed transposed; lanes then scatter consecutive LDS addresses.
template <typename T, int DH>
ME_DEV void tile_gload_rw(chunk16* r, const T* base, size_t ld, int row0, int nrows_valid_end, int tid) {
    using C = ACfg<T, DH>;
#pragma unroll
    for (int i = 0; i < C::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < C::NCHUNK) {
            const int row = c & 31, cc = (c >> 5) * C::CH;
            r[i] = (row0 + row < nrows_valid_end) ? ld_chunk(base + (size_t)(row0 + row) * ld + cc) : zero_chunk();
        }
    }
}
template <typename T, int DH>
ME_DEV void tile_sstore_nat_rw(const chunk16* r, T* S, int tid) {
    using C = ACfg<T, DH>;
#pragma unroll
    for (int i = 0; i < C::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < C::NCHUNK) st_chunk(&S[(c & 31) * C::LDN + (c >> 5) * C::CH], r[i]);
    }
}
template <typename T, int DH>
ME_DEV void tile_sstore_tr_rw(const chunk16* r, T* St, int tid) {
    using C = ACfg<T, DH>;
#pragma unroll
    for (int i = 0; i < C::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < C::NCHUNK) {
            const int row = c & 31, cc = (c >> 5) * C::CH;
            const T* e = reinterpret_cast<const T*>(&r[i]);
#pragma unroll
            for (int k = 0; k < C::CH; ++k) St[(cc + k) * LDT + row] = e[k];
        }
    }
}
// coalesced map (lane -> chunk within a row) for tiles only needed in natural layout
template <typename T, int DH>
ME_DEV void tile_gload_co(chunk16* r, const T* base, size_t ld, int row0, int nrows_valid_end, int tid) {
    using C = ACfg<T, DH>;
    constexpr int CPR = DH / C::CH;
#pragma unroll
    for (int i = 0; i < C::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < C::NCHUNK) {
            const int row = c / CPR, cc = (c % CPR) * C::CH;
            r[i] = (row0 + row < nrows_valid_end) ? ld_chunk(base + (size_t)(row0 + row) * ld + cc) : zero_chunk();
        }
    }
}
template <typename T, int DH>
ME_DEV void tile_sstore_nat_co(const chunk16* r, T* S, int tid) {
    using C = ACfg<T, DH>;
    constexpr int CPR = DH / C::CH;
#pragma unroll
    for (int i = 0; i < C::NPT; ++i) {
        const int c = tid + i * 256;
        if (c < C::NCHUNK) st_chunk(&S[(c / CPR) * C::LDN + (c % CPR) * C::CH], r[i]);
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

// 32-bit key-pad bitmask of a key tile (bit j = key k0+j is padding)
ME_DEV uint32_t pad_bits(const uint8_t* key_pad, int b, int L, int k0, int lane) {
    if (!key_pad) return 0u;
    const int key = k0 + (lane & 31);
    const bool f = (lane < 32) && key < L && key_pad[(size_t)b * L + key];
    return (uint32_t)__ballot(f);
}

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
// are double buffered in LDS (one barrier per step; the next tile's global loads are in
// flight during the whole step), the E fragments of the NEXT step's new block are fetched
// into registers right after the current block's MFMAs were issued, the pad flags travel
// with the tile, and tiles that need no masking (not diagonal, no pad, not the ragged tail)
// skip all per-element predicates.  Softmax runs in the exp2 domain.
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ E,
                                                      const uint8_t* __restrict__ key_pad, T* __restrict__ out,
                                                      float* __restrict__ lse, int B, int L, int H, int M, float scale) {
    using C = ACfg<T, DH>;
    __shared__ __attribute__((aligned(16))) T Ks[2][32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Vt[2][DH * LDT];
    __shared__ __attribute__((aligned(16))) float Gs[4][2][32 * LDG];
    __shared__ uint32_t Ps[2][32];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int nqb = (L + 127) / 128;
    const int bh = blockIdx.x / nqb, qb = nqb - 1 - (int)(blockIdx.x % nqb);
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const T* qb_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* kb_ = qb_ + dm;
    const T* vb_ = qb_ + 2 * dm;
    const int q0 = qb * 128 + wid * 32;
    const int q = q0 + a;
    const bool wave_on = q0 < L;
    const int nkt = min((L + 31) / 32, qb * 4 + 4);
    const int my_last_kt = qb * 4 + wid;            // diagonal tile of this wave
    const float c2 = scale * 1.4426950408889634f;   // logits are kept in log2 units

    Frag<T> qf[C::KA];
    row_frags<T, DH>(qf, qb_ + (size_t)q * ldq, wave_on && q < L, h);

    f32x16_t o[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) acc_zero(o[i]);
    float m_run = -INFINITY, l_run = 0.f;

    chunk16 rk[C::NPT], rv[C::NPT];
    uint32_t rp = 0;
    auto gload = [&](int kt) {
        tile_gload_co<T, DH>(rk, kb_, ldq, kt * 32, L, tid);
        tile_gload_rw<T, DH>(rv, vb_, ldq, kt * 32, L, tid);
        if (key_pad && tid < 32) { const int key = kt * 32 + tid; rp = key < L ? key_pad[(size_t)b * L + key] : 0; }
    };
    auto sstore = [&](int buf) {
        tile_sstore_nat_co<T, DH>(rk, Ks[buf], tid);
        tile_sstore_tr_rw<T, DH>(rv, Vt[buf], tid);
        if (key_pad && tid < 32) Ps[buf][tid] = rp;
    };
    auto g_block = [&](const Frag<T>* ef, int eb) {     // G^T[m][q] = E[eb*32+m] . Q[q] -> ring slot eb&1
        f32x16_t g; acc_zero(g);
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) mma32(g, ef[kk], qf[kk]);
        float* gs = &Gs[wid][eb & 1][a * LDG];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4_t*>(gs + 8 * gq + 4 * h) = (f32x4_t){g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
    };

    gload(0);
    // E blocks of step 0: lo now, hi prefetched (not needed if step 0 is already the diagonal)
    const int eb0 = (M - 32 - q0) >> 5;
    Frag<T> ef[C::KA];
    if (wave_on) {
        row_frags<T, DH>(ef, E + (size_t)(eb0 * 32 + a) * DH, true, h);
        g_block(ef, eb0);
        if (my_last_kt > 0) row_frags<T, DH>(ef, E + (size_t)((eb0 + 1) * 32 + a) * DH, true, h);
    }
    sstore(0);
    if (nkt > 1) gload(1);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (wave_on && kt <= my_last_kt) {
            const int k0 = kt * 32;
            const bool diag = kt == my_last_kt;
            const int eb_lo = eb0 + kt;
#if ME_ABL != 1
            if (!diag) {
                g_block(ef, eb_lo + 1);
                if (kt + 1 < my_last_kt) row_frags<T, DH>(ef, E + (size_t)((eb_lo + 2) * 32 + a) * DH, true, h);
            }
#endif
            // ---- S^T[key][q] = K[key] . Q[q]
            f32x16_t s; acc_zero(s);
#pragma unroll
            for (int kk = 0; kk < C::KA; ++kk) {
                Frag<T> kf; frag_load(kf, &Ks[buf][a * C::LDN + kk * 16 + h * 8]);
                mma32(s, kf, qf[kk]);
            }
            // ---- + Srel (skewed read), log2-scale, mask, online softmax (lane owns query q)
            uint32_t pbits = 0;
            if (key_pad) pbits = __builtin_amdgcn_readfirstlane((uint32_t)__ballot(lane < 32 && Ps[buf][a] != 0));
            const float* glo = &Gs[wid][eb_lo & 1][a * LDG];
            const float* ghi = &Gs[wid][(eb_lo + 1) & 1][a * LDG];
            const int mbase = 31 - a + 4 * h;
            float mt = -INFINITY;
            if (!diag && pbits == 0u && k0 + 32 <= L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
#if ME_ABL == 1 || ME_ABL == 6
                    const float g = 0.f;
#else
                    const float g = m < 32 ? glo[m] : ghi[m - 32];
#endif
                    s[r] = (s[r] + g) * c2;
                    mt = fmaxf(mt, s[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bk = (r & 3) + 8 * (r >> 2) + 4 * h, key = k0 + bk;
                    const int m = 31 - a + bk;
                    const bool masked = key > q || key >= L || ((pbits >> bk) & 1u);
                    float v = -INFINITY;
                    if (!masked) v = (s[r] + (m < 32 ? glo[m] : ghi[m - 32])) * c2;
                    s[r] = v;
                    mt = fmaxf(mt, v);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = exp2f(m_run - m_safe);
            float rs = 0.f;
#pragma unroll
#if ME_ABL == 2
            for (int r = 0; r < 16; ++r) { s[r] = (s[r] - m_safe); rs += s[r]; }
#else
            for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - m_safe); rs += s[r]; }
#endif
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < C::DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]
#if ME_ABL == 3
            o[0][0] += s[0] + s[15];
#else
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> pf; frag_from_acc(pf, s, t);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) {
                    Frag<T> vf;
                    const T* vp = &Vt[buf][(i * 32 + a) * LDT + 16 * t + 4 * h];
                    frag_load_4x2(vf, vp, vp + 8);
                    mma32(o[i], vf, pf);
                }
            }
#endif
        }
#if ME_ABL != 4
        if (kt + 1 < nkt) {
            sstore(buf ^ 1);                      // buf^1 was last read in step kt-1 (barrier since)
            if (kt + 2 < nkt) gload(kt + 2);
        }
#endif
#if ME_ABL != 5
        __syncthreads();
#endif
    }
    if (!wave_on || q >= L) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (h == 0) lse[((size_t)b * H + head) * L + q] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    T* op = out + ((size_t)b * L + q) * dm + head * DH;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            st4<T>(op + i * 32 + 8 * gq + 4 * h, o[i][4 * gq] * inv, o[i][4 * gq + 1] * inv, o[i][4 * gq + 2] * inv,
                   o[i][4 * gq + 3] * inv);
}

// =====================================================================================
// backward 1/3: query-owned.  delta, dS (-> workspace, scaled), dQ
// =====================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ E,
                                                         const T* __restrict__ ET_, const uint8_t* __restrict__ key_pad,
                                                         const T* __restrict__ out, const float* __restrict__ lse,
                                                         const T* __restrict__ dout, T* __restrict__ dqkv,
                                                         float* __restrict__ delta_ws, T* __restrict__ ds_ws, int B, int L,
                                                         int H, int M, float scale) {
    using C = ACfg<T, DH>;
    __shared__ __attribute__((aligned(16))) T Ks[32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Vs[32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Kt[DH * LDT];
    __shared__ __attribute__((aligned(16))) float Gs[4][2][32 * LDG];
    __shared__ __attribute__((aligned(16))) float Ds[4][32 * LDD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int nqb = (L + 127) / 128;
    const int bh = blockIdx.x / nqb, qb = nqb - 1 - (int)(blockIdx.x % nqb);
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
    const int nkt = min((L + 31) / 32, qb * 4 + 4);
    const int my_last_kt = qb * 4 + wid;

    Frag<T> qf[C::KA], dof[C::KA];
    row_frags<T, DH>(qf, qb_ + (size_t)q * ldq, row_on, h);
    const size_t orow = ((size_t)b * L + q) * dm + head * DH;
    row_frags<T, DH>(dof, dout + orow, row_on, h);
    float delta = 0.f, lse_q = 0.f;
    if (row_on) {
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                delta += ET<T>::to_f(dout[orow + kk * 16 + h * 8 + e]) * ET<T>::to_f(out[orow + kk * 16 + h * 8 + e]);
        lse_q = lse[((size_t)b * H + head) * L + q];
    }
    delta += __shfl_xor(delta, 32, 64);
    if (row_on && h == 0) delta_ws[((size_t)b * H + head) * L + q] = delta;

    f32x16_t dq[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) acc_zero(dq[i]);

    chunk16 rk[C::NPT], rv[C::NPT];
    tile_gload_rw<T, DH>(rk, kb_, ldq, 0, L, tid);
    tile_gload_co<T, DH>(rv, vb_, ldq, 0, L, tid);

    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_sstore_nat_rw<T, DH>(rk, Ks, tid);
        tile_sstore_tr_rw<T, DH>(rk, Kt, tid);
        tile_sstore_nat_co<T, DH>(rv, Vs, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_gload_rw<T, DH>(rk, kb_, ldq, (kt + 1) * 32, L, tid);
            tile_gload_co<T, DH>(rv, vb_, ldq, (kt + 1) * 32, L, tid);
        }
        if (!wave_on || kt > my_last_kt) continue;

        const int k0 = kt * 32;
        const bool diag = kt == my_last_kt;
        const int eb_lo = (M - 32 - q0 + k0) >> 5;
        for (int w = (kt == 0 ? 0 : 1); w < (diag ? 1 : 2); ++w) {
            const int eb = eb_lo + w;
            f32x16_t g; acc_zero(g);
            const T* erow = E + (size_t)(eb * 32 + a) * DH;
#pragma unroll
            for (int kk = 0; kk < C::KA; ++kk) {
                Frag<T> ef; frag_load(ef, erow + kk * 16 + h * 8);
                mma32(g, ef, qf[kk]);
            }
            float* gs = &Gs[wid][eb & 1][a * LDG];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<f32x4_t*>(gs + 8 * gq + 4 * h) = (f32x4_t){g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
        }
        f32x16_t s, dp; acc_zero(s); acc_zero(dp);
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) {
            Frag<T> kf, vf;
            frag_load(kf, &Ks[a * C::LDN + kk * 16 + h * 8]);
            frag_load(vf, &Vs[a * C::LDN + kk * 16 + h * 8]);
            mma32(s, kf, qf[kk]);          // S^T[key][q]
            mma32(dp, vf, dof[kk]);        // dP^T[key][q] = V[key] . dO[q]
        }
        const uint32_t pbits = pad_bits(key_pad, b, L, k0, lane);
        float* dsk = &Ds[wid][a * LDD];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bk = c_row(r, lane), key = k0 + bk;
            const int m = 31 - a + bk;
            const bool masked = key > q || key >= L || !row_on || ((pbits >> bk) & 1u);
            float ds = 0.f;
            if (!masked) {
                const float v = (s[r] + Gs[wid][(eb_lo + (m >> 5)) & 1][a * LDG + (m & 31)]) * scale;
                const float p = ET<T>::fexp(v - lse_q);
                ds = p * (dp[r] - delta) * scale;
            }
            s[r] = ds;
            dsk[m] = ds;                    // skewed position (q, m) for the relative part
        }
        // ---- scaled dS -> workspace [bh, q, key] (consumed by the dE kernel)
        if (row_on) {
            T* dsp = ds_ws + (((size_t)b * H + head) * L + q) * L + k0;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int bk = 8 * gq + 4 * h;
                if ((L & 3) == 0 && k0 + bk + 3 < L) st4<T>(dsp + bk, s[4 * gq], s[4 * gq + 1], s[4 * gq + 2], s[4 * gq + 3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k0 + bk + e < L) dsp[bk + e] = ET<T>::from_f(s[4 * gq + e]);
                }
            }
        }
        // ---- dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Frag<T> dsf; frag_from_acc(dsf, s, t);
#pragma unroll
            for (int i = 0; i < C::DB; ++i) {
                Frag<T> kf;
                const T* kp = &Kt[(i * 32 + a) * LDT + 16 * t + 4 * h];
                frag_load_4x2(kf, kp, kp + 8);
                mma32(dq[i], kf, dsf);
            }
        }
        // ---- dQ^T[d][q] += E^T[d][e_lo+m] . dG^T[m][q],  dG[q][m] = dS[q][m-31+a] (un-skew through LDS)
        for (int mb = 0; mb < (diag ? 1 : 2); ++mb) {
            f32x16_t dg;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int m0 = mb * 32 + 8 * gq + 4 * h;
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(dsk + m0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bk = m0 + e - 31 + a;
                    dg[4 * gq + e] = (bk >= 0 && bk < 32) ? v[e] : 0.f;
                }
            }
            const int ecol = (eb_lo + mb) * 32;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag<T> dgf; frag_from_acc(dgf, dg, t);
#pragma unroll
                for (int i = 0; i < C::DB; ++i) {
                    Frag<T> ef;
                    const T* ep = ET_ + (size_t)(i * 32 + a) * M + ecol + 16 * t + 4 * h;
                    frag_load_4x2(ef, ep, ep + 8);
                    mma32(dq[i], ef, dgf);
                }
            }
        }
    }
    if (!row_on) return;
    T* dqp = dqkv + ((size_t)b * L + q) * ldq + head * DH;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            st4<T>(dqp + i * 32 + 8 * gq + 4 * h, dq[i][4 * gq], dq[i][4 * gq + 1], dq[i][4 * gq + 2], dq[i][4 * gq + 3]);
}

// =====================================================================================
// backward 2/3: key-owned.  dK, dV  (untransposed tile: lane owns a key column)
// =====================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ E,
                                                          const uint8_t* __restrict__ key_pad, const float* __restrict__ lse,
                                                          const float* __restrict__ delta_ws, const T* __restrict__ dout,
                                                          T* __restrict__ dqkv, int B, int L, int H, int M, float scale) {
    using C = ACfg<T, DH>;
    __shared__ __attribute__((aligned(16))) T Qs[32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Os[32 * C::LDN];
    __shared__ __attribute__((aligned(16))) T Qt[DH * LDT];
    __shared__ __attribute__((aligned(16))) T Ot[DH * LDT];
    __shared__ float lse_s[32], del_s[32];
    __shared__ __attribute__((aligned(16))) float Gs[4][2][32 * LDG];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, a = lane & 31, h = lane >> 5;
    const int nkb = (L + 127) / 128;
    const int bh = blockIdx.x / nkb, kb = (int)(blockIdx.x % nkb);     // low key blocks are the long ones, they come first
    const int b = bh / H, head = bh % H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const T* qb_ = qkv + (size_t)b * L * ldq + head * DH;
    const T* kb_ = qb_ + dm;
    const T* vb_ = qb_ + 2 * dm;
    const T* dob_ = dout + (size_t)b * L * dm + head * DH;
    const int k0 = kb * 128 + wid * 32;
    const int key = k0 + a;
    const bool wave_on = k0 < L;
    const bool key_on = wave_on && key < L && !(key_pad && key_pad[(size_t)b * L + key]);
    const int nqt = (L + 31) / 32;
    const int my_first_qt = kb * 4 + wid;

    Frag<T> kf[C::KA], vf[C::KA];
    row_frags<T, DH>(kf, kb_ + (size_t)key * ldq, wave_on && key < L, h);
    row_frags<T, DH>(vf, vb_ + (size_t)key * ldq, wave_on && key < L, h);
    f32x16_t dk[C::DB], dv[C::DB];
#pragma unroll
    for (int i = 0; i < C::DB; ++i) { acc_zero(dk[i]); acc_zero(dv[i]); }

    chunk16 rq[C::NPT], ro[C::NPT];
    float r_lse = 0.f, r_del = 0.f;
    auto gload = [&](int qt) {
        tile_gload_rw<T, DH>(rq, qb_, ldq, qt * 32, L, tid);
        tile_gload_rw<T, DH>(ro, dob_, (size_t)dm, qt * 32, L, tid);
        if (tid < 32) {
            const int qq = qt * 32 + tid;
            r_lse = qq < L ? lse[((size_t)b * H + head) * L + qq] : 0.f;
            r_del = qq < L ? delta_ws[((size_t)b * H + head) * L + qq] : 0.f;
        }
    };
    const int qt_begin = kb * 4;
    gload(qt_begin);
    for (int qt = qt_begin; qt < nqt; ++qt) {
        __syncthreads();
        tile_sstore_nat_rw<T, DH>(rq, Qs, tid);
        tile_sstore_tr_rw<T, DH>(rq, Qt, tid);
        tile_sstore_nat_rw<T, DH>(ro, Os, tid);
        tile_sstore_tr_rw<T, DH>(ro, Ot, tid);
        if (tid < 32) { lse_s[tid] = r_lse; del_s[tid] = r_del; }
        __syncthreads();
        if (qt + 1 < nqt) gload(qt + 1);
        if (!wave_on || qt < my_first_qt) continue;

        const int q0 = qt * 32;
        const int eb_lo = (M - 32 - q0 + k0) >> 5;
        // ---- S[q][key], dP[q][key], and both band blocks G[q][m] = Q[q] . E[e_lo+m] (the query tile
        //      changes every step here, so unlike the query-owned kernels nothing can be reused)
        const bool diag = qt == my_first_qt;
        f32x16_t s, dp, g0, g1; acc_zero(s); acc_zero(dp); acc_zero(g0); acc_zero(g1);
        const T* erow = E + (size_t)(eb_lo * 32 + a) * DH;
#pragma unroll
        for (int kk = 0; kk < C::KA; ++kk) {
            Frag<T> qf, of, ef;
            frag_load(qf, &Qs[a * C::LDN + kk * 16 + h * 8]);
            frag_load(of, &Os[a * C::LDN + kk * 16 + h * 8]);
            frag_load(ef, erow + kk * 16 + h * 8);
            mma32(s, qf, kf[kk]);
            mma32(dp, of, vf[kk]);
            mma32(g0, qf, ef);
            if (!diag) {          // hi block rows are >= M on the diagonal tile (and never needed there)
                frag_load(ef, erow + (size_t)32 * DH + kk * 16 + h * 8);
                mma32(g1, qf, ef);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Gs[wid][0][c_row(r, lane) * LDG + a] = g0[r];
            Gs[wid][1][c_row(r, lane) * LDG + a] = g1[r];
        }
        // ---- p, ds   (rows = queries, lane column = key)
        f32x16_t p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int aq = c_row(r, lane), qq = q0 + aq;
            const int m = 31 - aq + a;
            const bool masked = !key_on || key > qq || qq >= L;
            float pv = 0.f, ds = 0.f;
            if (!masked) {
                const float v = (s[r] + Gs[wid][m >> 5][aq * LDG + (m & 31)]) * scale;
                pv = ET<T>::fexp(v - lse_s[aq]);
                ds = pv * (dp[r] - del_s[aq]) * scale;
            }
            p[r] = pv;
            s[r] = ds;
        }
        // ---- dV[key][d] += P[q][key] dO[q][d] ;  dK[key][d] += dS[q][key] Q[q][d]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Frag<T> pf, dsf;
            frag_from_acc(pf, p, t);
            frag_from_acc(dsf, s, t);
#pragma unroll
            for (int i = 0; i < C::DB; ++i) {
                Frag<T> otf, qtf;
                const T* op = &Ot[(i * 32 + a) * LDT + 16 * t + 4 * h];
                const T* qp = &Qt[(i * 32 + a) * LDT + 16 * t + 4 * h];
                frag_load_4x2(otf, op, op + 8);
                frag_load_4x2(qtf, qp, qp + 8);
                mma32(dv[i], pf, otf);
                mma32(dk[i], dsf, qtf);
            }
        }
    }
    if (!wave_on) return;
#pragma unroll
    for (int i = 0; i < C::DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk_ = k0 + c_row(r, lane);
            if (kk_ < L) {
                T* base = dqkv + ((size_t)b * L + kk_) * ldq + head * DH + i * 32 + a;
                base[dm] = ET<T>::from_f(dk[i][r]);
                base[2 * dm] = ET<T>::from_f(dv[i][r]);
            }
        }
}

// =====================================================================================
// backward 3/3: dE[e_lo+m][d] += sum_{bh, q} dS[q][m-31+a] Q[q][d]  organised by tile diagonal
// (all tile pairs with q0-k0 = 32*delta hit the same 64 rows of E) -> register accumulation
// over (bh, q-tile), one atomic flush per wave.
// =====================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_bwd_de_kernel(const T* __restrict__ qkv, const T* __restrict__ ds_ws,
                                                         float* __restrict__ dE, int B, int L, int H, int M) {
    using C = ACfg<T, DH>;
    constexpr int LDS_ = 32 + C::CH;
    __shared__ __attribute__((aligned(16))) T dSs[4][32 * LDS_];
    __shared__ __attribute__((aligned(16))) T Qs[4][32 * C::LDN];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, a = lane & 31, h = lane >> 5;
    const int nt = (L + 31) / 32;
    const int delta = blockIdx.x;                       // tile diagonal: kt = qt - delta
    const int BH = B * H;
    const int dm = H * DH;
    const size_t ldq = (size_t)3 * dm;
    const long items = (long)(nt - delta) * BH;
    const int nwaves = gridDim.y * 4;
    const bool vec_ok = (L % C::CH) == 0;

    f32x16_t acc[2][C::DB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < C::DB; ++i) acc_zero(acc[mb][i]);

    T* dss = dSs[wid];
    T* qs = Qs[wid];
    for (long it = (long)blockIdx.y * 4 + wid; it < items; it += nwaves) {
        const int bh = (int)(it % BH), qt = delta + (int)(it / BH);
        const int q0 = qt * 32, k0 = (qt - delta) * 32;
        const int b = bh / H, head = bh % H;
        const T* dsg = ds_ws + ((size_t)bh * L + q0) * L + k0;
        const T* qg = qkv + ((size_t)b * L + q0) * ldq + head * DH;
        // wave-private staging (LDS ops of one wave are ordered; no block barrier)
        if (vec_ok) {
            for (int c = lane; c < 32 * 32 / C::CH; c += 64) {
                const int row = c / (32 / C::CH), cc = (c % (32 / C::CH)) * C::CH;
                chunk16 v = (q0 + row < L && k0 + cc < L) ? ld_chunk(dsg + (size_t)row * L + cc) : zero_chunk();
                st_chunk(&dss[row * LDS_ + cc], v);
            }
        } else {
            for (int c = lane; c < 32 * 32; c += 64) {
                const int row = c >> 5, cc = c & 31;
                dss[row * LDS_ + cc] = (q0 + row < L && k0 + cc < L) ? dsg[(size_t)row * L + cc] : ET<T>::from_f(0.f);
            }
        }
        for (int c = lane; c < C::NCHUNK; c += 64) {
            const int row = c / (DH / C::CH), cc = (c % (DH / C::CH)) * C::CH;
            chunk16 v = (q0 + row < L) ? ld_chunk(qg + (size_t)row * ldq + cc) : zero_chunk();
            st_chunk(&qs[row * C::LDN + cc], v);
        }
        __builtin_amdgcn_s_waitcnt(0);   // staged tile complete before the gather reads (same wave)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Frag<T> qtf[C::DB];
#pragma unroll
            for (int i = 0; i < C::DB; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    frag_set(qtf[i], e, ET<T>::to_f(qs[(16 * t + 8 * h + e) * C::LDN + i * 32 + a]));
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                Frag<T> dgf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int aq = 16 * t + 8 * h + e;
                    const int bk = mb * 32 + a - 31 + aq;
                    const int bkc = bk < 0 ? 0 : (bk > 31 ? 31 : bk);
                    const float v = ET<T>::to_f(dss[aq * LDS_ + bkc]);
                    frag_set(dgf, e, (bk >= 0 && bk < 32) ? v : 0.f);
                }
#pragma unroll
                for (int i = 0; i < C::DB; ++i) mma32(acc[mb][i], dgf, qtf[i]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const int e_lo = M - 32 - 32 * delta;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < C::DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = e_lo + mb * 32 + c_row(r, lane);
                const float v = acc[mb][i][r];
                if (er >= 0 && er < M && v != 0.f) atomicAdd(&dE[(size_t)er * DH + i * 32 + a], v);
            }
}

// =====================================================================================
// cached decode step: one query (position t) per (b, head) against t+1 cached keys
// =====================================================================================
template <typename T, int DH>
__global__ __launch_bounds__(256) void rga_decode_kernel(const T* __restrict__ qkv_new, T* __restrict__ kcache,
                                                         T* __restrict__ vcache, const T* __restrict__ E,
                                                         const uint8_t* __restrict__ key_pad, int ld_pad,
                                                         T* __restrict__ out, int H, int M, int Mc, int t, float scale) {
    __shared__ float qs[DH];
    __shared__ float ps[2048 + 8];
    __shared__ float red[256];
    __shared__ float osum[256 / DH > 0 ? 256 : 256];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bh = blockIdx.x, b = bh / H, head = bh % H;
    const int dm = H * DH;
    const T* qn = qkv_new + (size_t)b * 3 * dm + head * DH;
    T* kc = kcache + (size_t)bh * Mc * DH;
    T* vc = vcache + (size_t)bh * Mc * DH;
    if (tid < DH) {
        qs[tid] = ET<T>::to_f(qn[tid]);
        kc[(size_t)t * DH + tid] = qn[dm + tid];
        vc[(size_t)t * DH + tid] = qn[2 * dm + tid];
    }
    __syncthreads();
    // scores
    float mx = -INFINITY;
    for (int j = tid; j <= t; j += 256) {
        const T* kr = (j == t) ? (qn + dm) : (kc + (size_t)j * DH);
        const T* er = E + (size_t)(M - 1 - (t - j)) * DH;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < DH; ++d) s += qs[d] * (ET<T>::to_f(kr[d]) + ET<T>::to_f(er[d]));
        s *= scale;
        if (key_pad && key_pad[(size_t)b * ld_pad + j]) s = -INFINITY;
        ps[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float m_safe = mx == -INFINITY ? 0.f : mx;
    float sm = 0.f;
    for (int j = tid; j <= t; j += 256) { const float p = expf(ps[j] - m_safe); ps[j] = p; sm += p; }
    sm = wave_sum(sm);
    __syncthreads();
    if (lane == 0) red[wid] = sm;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    // out[d] = sum_j p_j v[j][d] : thread (d, group) strides j
    constexpr int NG = 256 / DH;
    const int d = tid % DH, grp = tid / DH;
    float acc = 0.f;
    for (int j = grp; j <= t; j += NG) {
        const T* vr = (j == t) ? (qn + 2 * dm) : (vc + (size_t)j * DH);
        acc += ps[j] * ET<T>::to_f(vr[d]);
    }
    osum[tid] = acc;
    __syncthreads();
    if (tid < DH) {
        float s = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) s += osum[g2 * DH + tid];
        out[(size_t)b * dm + head * DH + tid] = ET<T>::from_f(s * inv);
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int DH>
int fwd_launch(const void* qkv, const void* E, const uint8_t* key_pad, void* out, float* lse, int B, int L, int H, int M,
               hipStream_t st) {
    const int nqb = (L + 127) / 128;
    const float scale = 1.f / sqrtf((float)DH);
    rga_fwd_kernel<T, DH><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)E, key_pad, (T*)out, lse, B, L, H, M, scale);
    return me_launch_status();
}

template <typename T, int DH>
int bwd_launch(const void* qkv, const void* E, const void* ET_, const uint8_t* key_pad, const void* out, const float* lse,
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* ds_ws, int B, int L, int H, int M,
               hipStream_t st) {
    const int nqb = (L + 127) / 128;
    const int nt = (L + 31) / 32;
    const float scale = 1.f / sqrtf((float)DH);
    rga_bwd_dq_kernel<T, DH><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)E, (const T*)ET_, key_pad, (const T*)out, lse,
                                                         (const T*)dout, (T*)dqkv, delta_ws, (T*)ds_ws, B, L, H, M, scale);
    int rc = me_launch_status();
    if (rc) return rc;
    rga_bwd_dkv_kernel<T, DH><<<B * H * nqb, 256, 0, st>>>((const T*)qkv, (const T*)E, key_pad, lse, delta_ws, (const T*)dout,
                                                          (T*)dqkv, B, L, H, M, scale);
    rc = me_launch_status();
    if (rc) return rc;
    int splits = (B * H + 3) / 4;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
    rga_bwd_de_kernel<T, DH><<<dim3(nt, splits), 256, 0, st>>>((const T*)qkv, (const T*)ds_ws, dE, B, L, H, M);
    return me_launch_status();
}

template <typename T, int DH>
int dec_launch(const void* qkv_new, void* kc, void* vc, const void* E, const uint8_t* key_pad, int ld_pad, void* out, int B,
               int H, int M, int Mc, int t, hipStream_t st) {
    const float scale = 1.f / sqrtf((float)DH);
    rga_decode_kernel<T, DH><<<B * H, 256, 0, st>>>((const T*)qkv_new, (T*)kc, (T*)vc, (const T*)E, key_pad, ld_pad, (T*)out, H, M,
                                                   Mc, t, scale);
    return me_launch_status();
}

}  // namespace

#define ME_ATTN_DISPATCH(CALL)                                                   \
    if (dtype == ME_F32) {                                                       \
        if (dh == 64) { typedef float T; constexpr int DH = 64; return CALL; }   \
        if (dh == 32) { typedef float T; constexpr int DH = 32; return CALL; }   \
    } else if (dtype == ME_BF16) {                                               \
        if (dh == 64) { typedef bf16_t T; constexpr int DH = 64; return CALL; }  \
        if (dh == 32) { typedef bf16_t T; constexpr int DH = 32; return CALL; }  \
    } else return ME_ERR_BAD_DTYPE;                                              \
    return ME_ERR_BAD_SHAPE;

extern "C" {

int me_rga_fwd(const void* qkv, const void* E, const uint8_t* key_pad, void* out, float* lse, int B, int L, int H, int dh,
               int M, int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !E || !out || !lse) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31)) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(E) || !aligned16(out)) return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((fwd_launch<T, DH>(qkv, E, key_pad, out, lse, B, L, H, M, st)))
}

int me_rga_bwd(const void* qkv, const void* E, const void* ET_, const uint8_t* key_pad, const void* out, const float* lse,
               const void* dout, void* dqkv, float* dE, float* delta_ws, void* ds_ws, int B, int L, int H, int dh, int M,
               int dtype, void* stream) {
    me_clear_error();
    if (!qkv || !E || !ET_ || !out || !lse || !dout || !dqkv || !dE || !delta_ws || !ds_ws) return ME_ERR_NULL;
    if (B <= 0 || L <= 0 || H <= 0 || L > M || (M & 31)) return ME_ERR_BAD_SHAPE;
    if (!aligned16(qkv) || !aligned16(E) || !aligned16(ET_) || !aligned16(out) || !aligned16(dout) || !aligned16(dqkv) ||
        !aligned16(ds_ws))
        return ME_ERR_ALIGNMENT;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((bwd_launch<T, DH>(qkv, E, ET_, key_pad, out, lse, dout, dqkv, dE, delta_ws, ds_ws, B, L, H, M, st)))
}

int me_rga_decode_step(const void* qkv_new, void* kcache, void* vcache, const void* E, const uint8_t* key_pad, int ld_pad,
                       void* out, int B, int H, int dh, int M, int Mc, int t, int dtype, void* stream) {
    me_clear_error();
    if (!qkv_new || !kcache || !vcache || !E || !out) return ME_ERR_NULL;
    if (B <= 0 || H <= 0 || t < 0 || t >= Mc || t >= M || t >= 2048) return ME_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    ME_ATTN_DISPATCH((dec_launch<T, DH>(qkv_new, kcache, vcache, E, key_pad, ld_pad, out, B, H, M, Mc, t, st)))
}

}  // extern "C"
