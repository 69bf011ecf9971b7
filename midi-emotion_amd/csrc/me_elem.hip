// HBM-bound kernels of the midiemo hot path (gfx950): embedding prologue,
// residual+dropout+LayerNorm, cross-entropy head, global-norm clip + AdamW,
// key-pad mask, greedy pick.  One wavefront (64 lanes) owns one row; every lane
// moves 16-byte chunks; row statistics are wave shuffles, no LDS.
#include "me_common.h"
#include <type_traits>

namespace {

constexpr int MAXC = 4;   // chunks per lane per row  =>  d <= 64*4*CH  (2048 bf16 / 1024 f32)

template <typename T> ME_DEV void chunk_to_f(const chunk16& c, float* f) {
    const T* e = reinterpret_cast<const T*>(&c);
#pragma unroll
    for (int i = 0; i < ET<T>::CH; ++i) f[i] = ET<T>::to_f(e[i]);
}
template <typename T> ME_DEV chunk16 f_to_chunk(const float* f) {
    chunk16 c;
    T* e = reinterpret_cast<T*>(&c);
#pragma unroll
    for (int i = 0; i < ET<T>::CH; ++i) e[i] = ET<T>::from_f(f[i]);
    return c;
}

// dropout multipliers (0 or 1/(1-p)) for CH consecutive elements starting at even index idx0
template <int CH> ME_DEV void drop_mult(float* mult, uint64_t seed, uint32_t site, uint64_t idx0, uint32_t thr16, float inv_keep) {
#pragma unroll
    for (int i = 0; i < CH; i += 2) {
        const uint32_t r = me_rng_pair(seed, site, (idx0 + i) >> 1);
        mult[i] = ((r & 0xFFFFu) >= thr16) ? inv_keep : 0.f;
        mult[i + 1] = ((r >> 16) >= thr16) ? inv_keep : 0.f;
    }
}
inline uint32_t thr_of(float p) {
    if (p <= 0.f) return 0u;
    float t = p * 65536.f + 0.5f;
    return t > 65535.f ? 65535u : (uint32_t)t;
}

// ------------------------------------------------------------------ embedding prologue
// ---- 8-bit low half of a 16-bit residual value (round 6: the residual stream's low halves are 0.22 ms of the step as two 16-bit
// arrays, profiles / DESIGN): q = round((v - hi) / (ulp(hi) / 256)) in [-127, 127], one byte per element.  hi has MB stored mantissa bits
// (bf16 7, f16 10), so hi + q * ulp / 256 carries MB + 8 of them: 15 for bf16 (the 16-bit low half gave ~16), 18 for f16.  Values whose
// exponent is too small for the step to be a normal float (|hi| < 2^-111) keep no low half.
template <typename T> struct LoMant { static constexpr uint32_t MB = 7; };
template <> struct LoMant<f16_t> { static constexpr uint32_t MB = 10; };
template <typename T> ME_DEV uint32_t lo8_enc(float v, float hf) {
    constexpr uint32_t SH = LoMant<T>::MB + 8;
    const uint32_t e = (__builtin_bit_cast(uint32_t, hf) >> 23) & 0xffu;
    if (e <= SH) return 0u;
    const float inv = __builtin_bit_cast(float, (254u + SH - e) << 23);
    const float q = fminf(fmaxf(rintf((v - hf) * inv), -127.f), 127.f);
    return (uint32_t)(int)q & 0xffu;
}
template <typename T> ME_DEV float lo8_dec(float hf, uint32_t byte) {
    constexpr uint32_t SH = LoMant<T>::MB + 8;
    const uint32_t e = (__builtin_bit_cast(uint32_t, hf) >> 23) & 0xffu;
    if (e <= SH) return 0.f;
    const float step = __builtin_bit_cast(float, (e - SH) << 23);
    return (float)(int)(int8_t)byte * step;
}
// CH = 8 values of a lane <-> 8 bytes
template <typename T> ME_DEV uint2 lo8_pack(const float (&v)[8], const float (&hf)[8]) {
    uint2 r = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.x |= lo8_enc<T>(v[i], hf[i]) << (8 * i); r.y |= lo8_enc<T>(v[4 + i], hf[4 + i]) << (8 * i); }
    return r;
}
template <typename T> ME_DEV void lo8_add(float (&xv)[8], uint2 q) {          // xv (the high halves as floats) += low halves
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float l0 = lo8_dec<T>(xv[i], (q.x >> (8 * i)) & 0xffu), l1 = lo8_dec<T>(xv[4 + i], (q.y >> (8 * i)) & 0xffu);
        xv[i] += l0;
        xv[4 + i] += l1;
    }
}

template <typename T, bool LO8 = false>
__global__ __launch_bounds__(256) void embed_fwd_kernel(T* __restrict__ out, T* __restrict__ out_lo, const int64_t* __restrict__ tokens,
                                                        const float* __restrict__ cond, const float* __restrict__ emb,
                                                        const float* __restrict__ cw0, const float* __restrict__ cb0,
                                                        const float* __restrict__ cw1, const float* __restrict__ cb1,
                                                        const float* __restrict__ pe, const int32_t* __restrict__ pos_dev,
                                                        int mode, int B, int Ltok, int d,
                                                        int dc, uint32_t thr16, float inv_keep, uint64_t seed) {
    constexpr int CH = ET<T>::CH;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int shift = mode == ME_COND_TOKEN ? 2 : 0;
    const int pos0 = pos_dev ? *pos_dev : 0;          // decode: first position comes from device memory (graph replay)
    const int Lm = Ltok + shift;
    const int64_t row = (int64_t)blockIdx.x * 4 + wid;
    if (row >= (int64_t)B * Lm) return;
    const int b = (int)(row / Lm), l = (int)(row % Lm);
    const int de = d - dc;
    const float sq = sqrtf((float)de);
    const float c0 = cond ? cond[b * 2] : 0.f, c1 = cond ? cond[b * 2 + 1] : 0.f;
    int64_t tok = 0;
    if (l >= shift) tok = tokens[(int64_t)b * Ltok + (l - shift)];
    const bool vec_ok = (de % 4) == 0 && (d % 4) == 0 && (dc % 4) == 0;
    for (int col = lane * CH; col < d; col += 64 * CH) {
        float v[CH];
        const float* pep = pe + (int64_t)(l + pos0) * d + col;
        if (vec_ok && l >= shift && col + CH <= de) {
            // chunk inside the token embedding: 16-byte loads of the table row and of the positional encoding
            const float* ep = emb + tok * de + col;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const f32x4_t e4 = *reinterpret_cast<const f32x4_t*>(ep + 4 * q);
                const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(pep + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * q + i] = e4[i] * sq + p4[i];
            }
        } else if (vec_ok && l >= shift && col >= de) {
            // chunk inside the concatenated condition projection: cw0 is [dc][2] row-major
            const int jc = col - de;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const f32x4_t wa = *reinterpret_cast<const f32x4_t*>(cw0 + (jc + 4 * q) * 2);
                const f32x4_t wb = *reinterpret_cast<const f32x4_t*>(cw0 + (jc + 4 * q) * 2 + 4);
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(cb0 + jc + 4 * q);
                const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(pep + 4 * q);
                v[4 * q + 0] = wa[0] * c0 + wa[1] * c1 + b4[0] + p4[0];
                v[4 * q + 1] = wa[2] * c0 + wa[3] * c1 + b4[1] + p4[1];
                v[4 * q + 2] = wb[0] * c0 + wb[1] * c1 + b4[2] + p4[2];
                v[4 * q + 3] = wb[2] * c0 + wb[3] * c1 + b4[3] + p4[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int j = col + i;
                float x;
                if (l < shift) {
                    x = l == 0 ? cw0[j] * c0 + cb0[j] : cw1[j] * c1 + cb1[j];
                } else if (j < de) {
                    x = emb[tok * de + j] * sq;
                } else {
                    const int jc = j - de;
                    x = cw0[jc * 2] * c0 + cw0[jc * 2 + 1] * c1 + cb0[jc];
                }
                v[i] = x + pe[(int64_t)(l + pos0) * d + j];
            }
        }
        if (thr16) {
            float mult[CH];
            drop_mult<CH>(mult, seed, 0u, (uint64_t)row * d + col, thr16, inv_keep);
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] *= mult[i];
        }
        const chunk16 hi = f_to_chunk<T>(v);
        st_chunk(out + row * d + col, hi);
        if (out_lo) {                                   // low-order part of the residual stream: v - float(T(v))
            float hf[CH];
            chunk_to_f<T>(hi, hf);
            if constexpr (LO8 && CH == 8) {
                *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out_lo) + row * d + col) = lo8_pack<T>(v, hf);
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) hf[i] = v[i] - hf[i];
                st_chunk(out_lo + row * d + col, f_to_chunk<T>(hf));
            }
        }
    }
}

// grid (nslab, B); block loops over the rows of its slab, per-lane column partials for
// the condition projection are reduced across the 4 waves in LDS -> one atomic per column
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ tokens,
                                                        const float* __restrict__ cond, float* __restrict__ g_emb,
                                                        float* __restrict__ g_cw0, float* __restrict__ g_cb0,
                                                        float* __restrict__ g_cw1, float* __restrict__ g_cb1, int mode,
                                                        int B, int Ltok, int d, int dc, int pad_token, uint32_t thr16,
                                                        float inv_keep, uint64_t seed, int skip_emb) {
    constexpr int CH = ET<T>::CH;
    __shared__ float red[4][MAXC * 64 * 8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int shift = mode == ME_COND_TOKEN ? 2 : 0;
    const int Lm = Ltok + shift;
    const int b = blockIdx.y;
    const int de = d - dc;
    const float sq = sqrtf((float)de);
    const int rows_per = (Lm + gridDim.x - 1) / gridDim.x;
    const int l_begin = blockIdx.x * rows_per, l_end = min(Lm, l_begin + rows_per);
    float part[MAXC][CH];   // column partial sums of the concat-condition part
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int i = 0; i < CH; ++i) part[c][i] = 0.f;

    for (int l = l_begin + wid; l < l_end; l += 4) {
        const int64_t row = (int64_t)b * Lm + l;
        int64_t tok = 0;
        if (l >= shift) tok = tokens[(int64_t)b * Ltok + (l - shift)];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + c * 64) * CH;
            if (col >= d) break;
            if (skip_emb && l >= shift && col + CH <= de) continue;      // embed_bwd_table_kernel owns these columns
            float g[CH];
            chunk_to_f<T>(ld_chunk(dout + row * d + col), g);
            if (thr16) {
                float mult[CH];
                drop_mult<CH>(mult, seed, 0u, (uint64_t)row * d + col, thr16, inv_keep);
#pragma unroll
                for (int i = 0; i < CH; ++i) g[i] *= mult[i];
            }
            if (l < shift) {
                // continuous_token prefix rows: d W_l[j] += g*cond[b][l], d b_l[j] += g
                const float cv = cond[b * 2 + l];
                float* gw = l == 0 ? g_cw0 : g_cw1;
                float* gb = l == 0 ? g_cb0 : g_cb1;
#pragma unroll
                for (int i = 0; i < CH; ++i) { atomicAdd(&gw[col + i], g[i] * cv); atomicAdd(&gb[col + i], g[i]); }
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int j = col + i;
                    if (j < de) { if (!skip_emb && tok != pad_token) atomicAdd(&g_emb[tok * de + j], g[i] * sq); }
                    else part[c][i] += g[i];
                }
            }
        }
    }
    if (mode != ME_COND_CONCAT || dc <= 0) return;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int i = 0; i < CH; ++i) red[wid][(c * 64 + lane) * CH + i] = part[c][i];
    __syncthreads();
    const float c0 = cond[b * 2], c1 = cond[b * 2 + 1];
    for (int j = threadIdx.x; j < d; j += 256) {
        if (j < de) continue;
        const float s = red[0][j] + red[1][j] + red[2][j] + red[3][j];
        const int jc = j - de;
        atomicAdd(&g_cw0[jc * 2], s * c0);
        atomicAdd(&g_cw0[jc * 2 + 1], s * c1);
        atomicAdd(&g_cb0[jc], s);
    }
}

// Condition-projection gradient of the concat mode when the table kernel owns the embedding columns:
//   d Wc[j][0..1] += sum_rows g[row][de + j] * cond[b][0..1],   d bc[j] += sum_rows g[row][de + j].
// Only the dc trailing columns are read: a wave covers 64 * CH / dc rows at once (lane = row-in-group x chunk), the
// three partial sums stay in registers over the block's rows, waves are combined through LDS slices (plain stores) and
// a block issues one atomic per output.  <= 128 blocks of 8 waves: with 512 four-wave blocks the same-address atomics
// were 40 of the old kernel's 44 us.
template <typename T>
__global__ __launch_bounds__(512) void embed_bwd_cond_kernel(const T* __restrict__ dout, const float* __restrict__ cond,
                                                               float* __restrict__ g_cw0, float* __restrict__ g_cb0, int B,
                                                               int Lm, int d, int dc, uint32_t thr16, float inv_keep,
                                                               uint64_t seed) {
    constexpr int CH = ET<T>::CH;
    __shared__ float red[8][3][64 * CH];                 // 48 KB (16-bit tier)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cpr = dc / CH;                        // chunks per row of the condition part (<= 64)
    const int rpw = 64 / cpr;                       // rows per wave pass
    const int rl = lane / cpr, ch = lane % cpr;
    const int de = d - dc;
    const int64_t rows = (int64_t)B * Lm;
    float pw0[CH], pw1[CH], pb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { pw0[i] = 0.f; pw1[i] = 0.f; pb[i] = 0.f; }
    const bool lane_on = rl < rpw;
    // U row groups of the wave in flight at once (unconditional clamped loads: one group at a time left a single 16-byte load
    // per lane outstanding -- 20 us for 8 MB of partial rows, round 4)
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * 8 * rpw;
    const int col = de + ch * CH;
    for (int64_t r0 = ((int64_t)blockIdx.x * 8 + wid) * rpw; r0 < rows; r0 += stride * U) {
        chunk16 c[U];
        float c0[U], c1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = min(r0 + u * stride + (lane_on ? rl : 0), rows - 1);
            c[u] = ld_chunk(dout + row * d + col);
            const int b = (int)(row / Lm);
            c0[u] = cond[b * 2]; c1[u] = cond[b * 2 + 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r0 + u * stride + rl;
            if (!lane_on || row >= rows) continue;
            float g[CH];
            chunk_to_f<T>(c[u], g);
            if (thr16) {
                float mult[CH];
                drop_mult<CH>(mult, seed, 0u, (uint64_t)row * d + col, thr16, inv_keep);
#pragma unroll
                for (int i = 0; i < CH; ++i) g[i] *= mult[i];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) { pw0[i] += g[i] * c0[u]; pw1[i] += g[i] * c1[u]; pb[i] += g[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        red[wid][0][i * 64 + lane] = pw0[i];
        red[wid][1][i * 64 + lane] = pw1[i];
        red[wid][2][i * 64 + lane] = pb[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 3 * dc; j += 512) {
        const int which = j / dc, jc = j % dc, c = jc / CH, i = jc % CH;
        float acc = 0.f;
        for (int w = 0; w < 8; ++w)
            for (int r = 0; r < rpw; ++r) acc += red[w][which][i * 64 + r * cpr + c];
        if (which == 0) atomicAdd(&g_cw0[jc * 2], acc);
        else if (which == 1) atomicAdd(&g_cw0[jc * 2 + 1], acc);
        else atomicAdd(&g_cb0[jc], acc);
    }
}

// Embedding-table gradient with LDS-privatised accumulation: a block owns an 8-column slice of the
// table (V x 8 floats in LDS) and a slice of the tokens; gradients are scattered with LDS atomics
// and flushed once, coalesced, with one global atomic per table element.  Replaces T x (d - dc)
// contended global atomics (12.6 M per step at the headline config).
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_table_kernel(const T* __restrict__ dout, const int64_t* __restrict__ tokens,
                                                              float* __restrict__ g_emb, int B, int Ltok, int shift, int d,
                                                              int de, int V, int pad_token, uint32_t thr16, float inv_keep,
                                                              uint64_t seed) {
    extern __shared__ float tab[];                  // [V][8]
    const int c0 = blockIdx.x * 8;
    const int Lm = Ltok + shift;
    const int64_t rows = (int64_t)B * Ltok;
    const float sq = sqrtf((float)de);
    for (int i = threadIdx.x; i < V * 8; i += 256) tab[i] = 0.f;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.y * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.y * 256) {
        const int b = (int)(r / Ltok), lt = (int)(r % Ltok);
        const int64_t tok = tokens[r];
        if (tok == pad_token) continue;
        const int64_t row = (int64_t)b * Lm + lt + shift;
        const T* src = dout + row * d + c0;
        float g[8];
        if constexpr (sizeof(T) == 2) {
            chunk_to_f<T>(ld_chunk(src), g);
        } else {
            chunk_to_f<T>(ld_chunk(src), g);
            chunk_to_f<T>(ld_chunk(src + 4), g + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c0 + e < de) {
                float v = g[e] * sq;
                if (thr16) v = me_keep(seed, 0u, (uint64_t)row * d + c0 + e, thr16) ? v * inv_keep : 0.f;
                atomicAdd(&tab[tok * 8 + e], v);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V * 8; i += 256) {
        const int v = i >> 3, e = i & 7;
        const float x = tab[i];
        if (c0 + e < de && x != 0.f) atomicAdd(&g_emb[(size_t)v * de + c0 + e], x);
    }
}

// Embedding-table gradient, one block per table row (round 3; the LDS-atomic version above read 199 MB for 33.6 MB of input
// -- every block one 16-byte column slice of every row -- and issued 12.6 M LDS atomic lane-operations: 95 us).
// Block v scans the token ids (256 KB at the headline shape, L2 resident; 16 loads per thread in flight, no barrier until the
// whole list has been seen), collects the positions holding token v in an LDS list, then reads exactly those rows WHOLE (coalesced, four rows per wave in
// flight) and sums them in registers; one plain += of the table row at the end: no sort pass, no workspace, no atomics on
// the table.  The order inside a round follows the arrival order of the LDS counter: sums are reproducible up to f32
// rounding order, like the accumulation order of the old kernel.
constexpr int EMB_HEAVY = 192, EMB_MAXH = 127, EMB_SLICE = 128;   // tokens with more occurrences go to the helper launch; capacity of its
                                                                   // (token, count) list (1 KB workspace); rows per helper work item
template <typename T, int TPB>
__global__ __launch_bounds__(256) void embed_bwd_gather_kernel(const T* __restrict__ dout, const int64_t* __restrict__ tokens,
                                                               float* __restrict__ g_emb, int* __restrict__ hv, int rows, int Ltok, int shift, int d, int de,
                                                               int vocab, int pad_token, uint32_t thr16, float inv_keep, uint64_t seed) {
    // TPB consecutive table rows per block: ONE scan of the token ids serves all of them (the scan -- every block reads the
    // whole 256 KB id list through L2 -- is what bounds this kernel: 1007 blocks x 1 row 34 us, 504 x 2 rows see the launcher)
    constexpr int CH = ET<T>::CH, CAP = 2048, MAXC = 2;                   // de <= 128 CH (bf16: 1024 columns)
    __shared__ int list[TPB][CAP];
    __shared__ int cnt[TPB];
    __shared__ int pub_idx;
    __shared__ float red[4][64 * CH];
    const int v0 = blockIdx.x * TPB, tid = threadIdx.x, lane = tid & 63, slot = tid >> 6;
    const int Lm = Ltok + shift, nch = de / CH;
    const float sq = sqrtf((float)de);
    float acc[MAXC][CH];
    // positions [r_lo, r_hi) holding one of the tokens [vlo, vlo + nv) -> their lists (entries past CAP are counted, not stored)
    auto scan = [&](int r_lo, int r_hi, int vlo, int nv) __attribute__((always_inline)) {
        for (int r0 = r_lo; r0 < r_hi; r0 += 256 * 16) {                  // 16 token loads per thread in flight, no barrier in between
            int64_t tk[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int r = r0 + tid + 256 * u; tk[u] = r < r_hi ? tokens[r] : -1; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t dv = tk[u] - vlo;
                if (dv >= 0 && dv < nv) {
                    const int li = (int)(tk[u] - v0);
                    const int pos = atomicAdd(&cnt[li], 1);
                    if (pos < CAP) list[li][pos] = r0 + tid + 256 * u;
                }
            }
        }
    };
    auto gather = [&](const int* lst, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb) {
            const int c = cb * 64 + lane;
            const bool on = c < nch;
            if (cb * 64 >= nch) break;
            for (int i0 = slot; i0 < n; i0 += 16) {                       // four rows per wave in flight, four waves
                chunk16 x[4];
                int64_t rowi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = lst[min(i0 + 4 * u, n - 1)];
                    rowi[u] = (int64_t)(r / Ltok) * Lm + r % Ltok + shift;
                    x[u] = on ? ld_chunk(dout + rowi[u] * d + c * CH) : zero_chunk();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + 4 * u < n && on) {
                        float g[CH];
                        chunk_to_f<T>(x[u], g);
                        if (thr16) {
                            float mult[CH];
                            drop_mult<CH>(mult, seed, 0u, (uint64_t)rowi[u] * d + c * CH, thr16, inv_keep);
#pragma unroll
                            for (int e = 0; e < CH; ++e) g[e] *= mult[e];
                        }
#pragma unroll
                        for (int e = 0; e < CH; ++e) acc[cb][e] += g[e];
                    }
                }
            }
        }
    };
    if (tid < TPB) cnt[tid] = 0;
    __syncthreads();
    scan(0, rows, v0, TPB);                                               // the common case: the whole token list in one go
    __syncthreads();
    int total[TPB];
#pragma unroll
    for (int j = 0; j < TPB; ++j) total[j] = cnt[j];
#pragma unroll
    for (int j = 0; j < TPB; ++j) {
        const int v = v0 + j;
        if (v >= vocab || v == pad_token || total[j] == 0) continue;      // block uniform
        if (hv && total[j] > EMB_HEAVY) {
            // a frequent token (real MIDI streams: time shifts, common notes -- thousands of occurrences): one CU cannot pull
            // its rows fast enough (5.8 k rows: 680 us), so it is published and embed_bwd_heavy_kernel spreads it over 64 blocks
            __syncthreads();
            if (tid == 0) {
                const int idx = atomicAdd(&hv[0], 1);
                if (idx < EMB_MAXH) { hv[2 + 2 * idx] = v; hv[3 + 2 * idx] = total[j]; }
                pub_idx = idx;
            }
            __syncthreads();
            if (pub_idx < EMB_MAXH) continue;                              // list full: handled here after all
        }
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[cb][e] = 0.f;
        if (total[j] <= CAP) gather(list[j], total[j]);
        else {                                                            // a very frequent token: rounds of CAP positions
            for (int r0 = 0; r0 < rows; r0 += CAP) {
                __syncthreads();
                if (tid == 0) cnt[j] = 0;
                __syncthreads();
                scan(r0, min(rows, r0 + CAP), v, 1);
                __syncthreads();
                const int n = cnt[j];
                if (n) gather(list[j], n);
            }
        }
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb) {
            if (cb * 64 >= nch) break;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < CH; ++e) red[slot][lane * CH + e] = acc[cb][e];
            __syncthreads();
            for (int jj = tid; jj < 64 * CH; jj += 256) {
                const int col = cb * 64 * CH + jj;
                if (col < de) g_emb[(size_t)v * de + col] += (red[0][jj] + red[1][jj] + red[2][jj] + red[3][jj]) * sq;
            }
        }
    }
}

// Frequent tokens published by embed_bwd_gather_kernel (token, count): a token with n occurrences becomes
// g = min(64, ceil(n / 128)) work items, item s = the s-th of g equal slices of the token-id list; the blocks take the
// items round robin, scan their slice, gather the matching rows and add the partial sum with atomics (g per table element,
// for a handful of rows).  Always launched after the gather kernel when the caller gave a workspace (a few us when the
// list is empty); the last block to finish leaves the workspace zeroed for the next call.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_heavy_kernel(const T* __restrict__ dout, const int64_t* __restrict__ tokens,
                                                              float* __restrict__ g_emb, int* __restrict__ hv, int rows, int Ltok, int shift,
                                                              int d, int de, uint32_t thr16, float inv_keep, uint64_t seed) {
    constexpr int CH = ET<T>::CH, CAP = 2048, MAXC = 2;
    __shared__ int list[CAP];
    __shared__ int cnt;
    __shared__ int first[EMB_MAXH + 1];                                  // first work item of every listed token
    __shared__ float red[4][64 * CH];
    const int tid = threadIdx.x, lane = tid & 63, slot = tid >> 6;
    const int nh = min(hv[0], EMB_MAXH);
    if (nh <= 0) return;                                                  // every block sees the same value: nothing to reset either
    const int Lm = Ltok + shift, nch = de / CH;
    const float sq = sqrtf((float)de);
    if (tid == 0) {
        int o = 0;
        for (int i = 0; i < nh; ++i) { first[i] = o; o += min(64, (hv[3 + 2 * i] + EMB_SLICE - 1) / EMB_SLICE); }
        first[nh] = o;
    }
    __syncthreads();
    const int nitems = first[nh];
    int ti = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        while (item >= first[ti + 1]) ++ti;                               // items ascend: the cursor only moves forward
        const int v = hv[2 + 2 * ti], g = first[ti + 1] - first[ti], sl = item - first[ti];
        const int r_lo = (int)((int64_t)rows * sl / g), r_hi = (int)((int64_t)rows * (sl + 1) / g);
        float acc[MAXC][CH];
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[cb][e] = 0.f;
        for (int r0 = r_lo; r0 < r_hi; r0 += 256 * 16) {
            __syncthreads();
            if (tid == 0) cnt = 0;
            __syncthreads();
            {                                                             // 16 token loads per thread in flight
                int64_t tk[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int r = r0 + tid + 256 * u; tk[u] = r < r_hi ? tokens[r] : -1; }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (tk[u] == v) { const int pos = atomicAdd(&cnt, 1); if (pos < CAP) list[pos] = r0 + tid + 256 * u; }
            }
            __syncthreads();
            const int n = cnt;                                            // may exceed CAP (more than half of the positions): re-scanned in halves below
            auto gather = [&](int n_) __attribute__((always_inline)) {
#pragma unroll
                for (int cb = 0; cb < MAXC; ++cb) {
                    const int c = cb * 64 + lane;
                    const bool on = c < nch;
                    if (cb * 64 >= nch) break;
                    for (int i0 = slot; i0 < n_; i0 += 16) {
                        chunk16 x[4];
                        int64_t rowi[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int r = list[min(i0 + 4 * u, n_ - 1)];
                            rowi[u] = (int64_t)(r / Ltok) * Lm + r % Ltok + shift;
                            x[u] = on ? ld_chunk(dout + rowi[u] * d + c * CH) : zero_chunk();
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (i0 + 4 * u < n_ && on) {
                                float gq[CH];
                                chunk_to_f<T>(x[u], gq);
                                if (thr16) {
                                    float mult[CH];
                                    drop_mult<CH>(mult, seed, 0u, (uint64_t)rowi[u] * d + c * CH, thr16, inv_keep);
#pragma unroll
                                    for (int e = 0; e < CH; ++e) gq[e] *= mult[e];
                                }
#pragma unroll
                                for (int e = 0; e < CH; ++e) acc[cb][e] += gq[e];
                            }
                        }
                    }
                }
            };
            if (n <= CAP) { if (n) gather(n); }
            else {
                for (int h0 = r0; h0 < min(r_hi, r0 + 256 * 16); h0 += CAP) {      // CAP positions at a time: the list cannot overflow
                    __syncthreads();
                    if (tid == 0) cnt = 0;
                    __syncthreads();
                    for (int r = h0 + tid; r < min(r_hi, h0 + CAP); r += 256)
                        if (tokens[r] == v) list[atomicAdd(&cnt, 1)] = r;
                    __syncthreads();
                    const int n2 = cnt;
                    if (n2) gather(n2);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb) {
            if (cb * 64 >= nch) break;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < CH; ++e) red[slot][lane * CH + e] = acc[cb][e];
            __syncthreads();
            for (int jj = tid; jj < 64 * CH; jj += 256) {
                const int col = cb * 64 * CH + jj;
                const float t = (red[0][jj] + red[1][jj] + red[2][jj] + red[3][jj]) * sq;
                if (col < de && t != 0.f) atomicAdd(&g_emb[(size_t)v * de + col], t);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && atomicAdd(&hv[1], 1) == (int)gridDim.x - 1) {
        for (int i = 0; i < 2 + 2 * EMB_MAXH; ++i) hv[i] = 0;
    }
}

__global__ void key_pad_kernel(uint8_t* __restrict__ kp, const int64_t* __restrict__ tokens, int B, int Ltok, int shift,
                               int pad_token) {
    const int Lm = Ltok + shift;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Lm) return;
    const int b = (int)(i / Lm), l = (int)(i % Lm);
    kp[i] = (l >= shift && tokens[(int64_t)b * Ltok + (l - shift)] == pad_token) ? 1 : 0;
}

// ------------------------------------------------------------------ residual + dropout + LayerNorm
template <typename T, bool LO8 = false>
__global__ __launch_bounds__(256) void resid_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ x_lo, const T* __restrict__ a,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           T* __restrict__ y, T* __restrict__ y_lo, T* __restrict__ s_out, float* __restrict__ stats,
                                                           int rows, int d, float eps, uint32_t thr16, float inv_keep,
                                                           uint64_t seed, uint32_t site) {
    constexpr int CH = ET<T>::CH;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
        float s[MAXC][CH];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + c * 64) * CH;
            if (col < d) {
                float xv[CH], av[CH];
                chunk_to_f<T>(ld_chunk(x + row * d + col), xv);
                chunk_to_f<T>(ld_chunk(a + row * d + col), av);
                if (x_lo) {                             // residual stream = hi + lo (the reference keeps it in fp32 under autocast)
                    if constexpr (LO8 && CH == 8) {
                        lo8_add<T>(xv, *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(x_lo) + row * d + col));
                    } else {
                        float lv[CH];
                        chunk_to_f<T>(ld_chunk(x_lo + row * d + col), lv);
#pragma unroll
                        for (int i = 0; i < CH; ++i) xv[i] += lv[i];
                    }
                }
                if (thr16) {
                    float mult[CH];
                    drop_mult<CH>(mult, seed, site, (uint64_t)row * d + col, thr16, inv_keep);
#pragma unroll
                    for (int i = 0; i < CH; ++i) av[i] *= mult[i];
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) { s[c][i] = xv[i] + av[i]; sum += s[c][i]; }
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) s[c][i] = 0.f;
            }
        }
        const float mean = wave_sum(sum) / d;
        float vs = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + c * 64) * CH;
            if (col < d) {
#pragma unroll
                for (int i = 0; i < CH; ++i) { const float t = s[c][i] - mean; vs += t * t; }
            }
        }
        const float rstd = rsqrtf(wave_sum(vs) / d + eps);
        if (stats && lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + c * 64) * CH;
            if (col < d) {
                float o[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) o[i] = (s[c][i] - mean) * rstd * gamma[col + i] + beta[col + i];
                const chunk16 hi = f_to_chunk<T>(o);
                st_chunk(y + row * d + col, hi);
                if (y_lo) {
                    float hf[CH];
                    chunk_to_f<T>(hi, hf);
                    if constexpr (LO8 && CH == 8) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(y_lo) + row * d + col) = lo8_pack<T>(o, hf);
                    } else {
#pragma unroll
                        for (int i = 0; i < CH; ++i) hf[i] = o[i] - hf[i];
                        st_chunk(y_lo + row * d + col, f_to_chunk<T>(hf));
                    }
                }
                if (s_out) st_chunk(s_out + row * d + col, f_to_chunk<T>(s[c]));
            }
        }
    }
}

// d == 64 CH (one 16-byte chunk per lane and row: 512 bf16 / 256 f32): R rows per wave in flight, the next R rows are
// requested before the current ones are reduced (a wave that loads, reduces and stores one row at a time leaves the
// memory pipe idle during its two dependent reductions).  Loads are unconditional (clamped to the last row).
template <typename T, int R, bool LO8 = false>
__global__ __launch_bounds__(256) void resid_ln_fwd1_kernel(const T* __restrict__ x, const T* __restrict__ x_lo, const T* __restrict__ a,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, T* __restrict__ y_lo, T* __restrict__ s_out, float* __restrict__ stats,
                                                            int rows, float eps, uint32_t thr16, float inv_keep,
                                                            uint64_t seed, uint32_t site) {
    constexpr int CH = ET<T>::CH;
    constexpr int d = 64 * CH;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane * CH;
    float gam[CH], bet[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { gam[i] = gamma[col + i]; bet[i] = beta[col + i]; }
    const int64_t stride = (int64_t)gridDim.x * 4 * R;
    int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * R;
    chunk16 nx[R], na[R], nl[R];
    uint2 nq[R];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = min(r0 + u, (int64_t)rows - 1);
            nx[u] = ld_chunk(x + r * d + col);
            na[u] = ld_chunk(a + r * d + col);
            if (x_lo) {
                if constexpr (LO8 && CH == 8) nq[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(x_lo) + r * d + col);
                else nl[u] = ld_chunk(x_lo + r * d + col);
            }
        }
    };
    fetch(row0);
    for (; row0 < rows; row0 += stride) {
        float s[R][CH];
        float sum[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float xv[CH], av[CH];
            chunk_to_f<T>(nx[u], xv);
            chunk_to_f<T>(na[u], av);
            if (x_lo) {
                if constexpr (LO8 && CH == 8) {
                    lo8_add<T>(xv, nq[u]);
                } else {
                    float lv[CH];
                    chunk_to_f<T>(nl[u], lv);
#pragma unroll
                    for (int i = 0; i < CH; ++i) xv[i] += lv[i];
                }
            }
            if (thr16) {
                float mult[CH];
                drop_mult<CH>(mult, seed, site, (uint64_t)min(row0 + u, (int64_t)rows - 1) * d + col, thr16, inv_keep);
#pragma unroll
                for (int i = 0; i < CH; ++i) av[i] *= mult[i];
            }
            sum[u] = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) { s[u][i] = xv[i] + av[i]; sum[u] += s[u][i]; }
        }
        fetch(row0 + stride);
        float mean[R], rstd[R];
#pragma unroll
        for (int u = 0; u < R; ++u) mean[u] = wave_sum(sum[u]) / d;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) { const float t = s[u][i] - mean[u]; vs += t * t; }
            sum[u] = vs;
        }
#pragma unroll
        for (int u = 0; u < R; ++u) rstd[u] = rsqrtf(wave_sum(sum[u]) / d + eps);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t row = row0 + u;
            if (row < rows) {
                if (stats && lane == 0) { stats[row * 2] = mean[u]; stats[row * 2 + 1] = rstd[u]; }
                float o[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) o[i] = (s[u][i] - mean[u]) * rstd[u] * gam[i] + bet[i];
                const chunk16 hi = f_to_chunk<T>(o);
                st_chunk(y + row * d + col, hi);
                if (y_lo) {
                    float hf[CH];
                    chunk_to_f<T>(hi, hf);
                    if constexpr (LO8 && CH == 8) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(y_lo) + row * d + col) = lo8_pack<T>(o, hf);
                    } else {
#pragma unroll
                        for (int i = 0; i < CH; ++i) hf[i] = o[i] - hf[i];
                        st_chunk(y_lo + row * d + col, f_to_chunk<T>(hf));
                    }
                }
                if (s_out) st_chunk(s_out + row * d + col, f_to_chunk<T>(s[u]));
            }
        }
    }
}

// NW waves per block: every wave keeps per-lane column partials of dgamma / dbeta over its rows, the block combines
// them with LDS atomics and issues ONE global atomic per column -- the flush (blocks x 2 d global atomics) is a
// visible part of the kernel, so blocks are fat (16 waves) rather than many.
template <typename T, int NC, int NW, int R = 1>
__global__ __launch_bounds__(NW * 64) void resid_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           T* __restrict__ dx, T* __restrict__ da, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int rows, int d, uint32_t thr16,
                                                           float inv_keep, uint64_t seed, uint32_t site) {
    constexpr int CH = ET<T>::CH;
    // per-wave partial sums of d(gamma) / d(beta): one slice per wave, plain stores, summed after a barrier.
    // (ds_add_f32 costs ~85 cycles per wave instruction: combining through LDS atomics held every block for
    // 1.3 us x waves -- 21 of the kernel's 41 us at 16 waves.)
    __shared__ float red[NW][2][NC * 64 * CH];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float pg[NC][CH], pb[NC][CH];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < CH; ++i) { pg[c][i] = 0.f; pb[c][i] = 0.f; }

    // gamma is row independent; the next row's chunks and statistics are fetched before the current row's
    // reductions so that every wave always has loads in flight (one row at a time ran at 2.7 TB/s)
    float gam[NC][CH];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = (lane + c * 64) * CH;
#pragma unroll
        for (int i = 0; i < CH; ++i) gam[c][i] = col < d ? gamma[col + i] : 0.f;
    }
    const int64_t stride = (int64_t)gridDim.x * NW * R;
    int64_t row0 = ((int64_t)blockIdx.x * NW + wid) * R;
    chunk16 ndy[R][NC], ns[R][NC];
    float nmean[R], nrstd[R];
    // unconditional (clamped to the last row): a load under a branch makes the compiler wait vmcnt(0) at the join,
    // i.e. also for the stores of the row before
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
            nmean[u] = stats[r * 2];
            nrstd[u] = stats[r * 2 + 1];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = min((lane + c * 64) * CH, d - CH);
                ndy[u][c] = ld_chunk(dy + r * d + col);
                ns[u][c] = ld_chunk(s + r * d + col);
            }
        }
    };
    fetch(row0);
    for (; row0 < rows; row0 += stride) {
        float mean[R], rstd[R];
        float g[R][NC][CH], xh[R][NC][CH];
        float s1[R], s2[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            mean[u] = nmean[u]; rstd[u] = nrstd[u];
            s1[u] = 0.f; s2[u] = 0.f;
            const bool on = row0 + u < rows;           // a clamped duplicate of the last row must not count twice
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = (lane + c * 64) * CH;
                if (col < d) {
                    float dyv[CH], sv[CH];
                    chunk_to_f<T>(ndy[u][c], dyv);
                    chunk_to_f<T>(ns[u][c], sv);
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        xh[u][c][i] = (sv[i] - mean[u]) * rstd[u];
                        g[u][c][i] = dyv[i] * gam[c][i];
                        s1[u] += g[u][c][i];
                        s2[u] += g[u][c][i] * xh[u][c][i];
                        if (R == 1 || on) {
                            pg[c][i] += dyv[i] * xh[u][c][i];
                            pb[c][i] += dyv[i];
                        }
                    }
                }
            }
        }
        fetch(row0 + stride);
        float c1[R], c2[R];
#pragma unroll
        for (int u = 0; u < R; ++u) c1[u] = wave_sum(s1[u]) / d;
#pragma unroll
        for (int u = 0; u < R; ++u) c2[u] = wave_sum(s2[u]) / d;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int64_t row = row0 + u;
            if (R > 1 && row >= rows) break;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = (lane + c * 64) * CH;
                if (col < d) {
                    float o[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) o[i] = rstd[u] * (g[u][c][i] - c1[u] - xh[u][c][i] * c2[u]);
                    st_chunk(dx + row * d + col, f_to_chunk<T>(o));
                    if (thr16) {
                        float mult[CH];
                        drop_mult<CH>(mult, seed, site, (uint64_t)row * d + col, thr16, inv_keep);
#pragma unroll
                        for (int i = 0; i < CH; ++i) o[i] *= mult[i];
                    }
                    st_chunk(da + row * d + col, f_to_chunk<T>(o));
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            // element-major layout [i][c][lane]: the 64 lanes of one store hit 64 consecutive words
            red[wid][0][(i * NC + c) * 64 + lane] = pg[c][i];
            red[wid][1][(i * NC + c) * 64 + lane] = pb[c][i];
        }
    __syncthreads();
    for (int j0 = threadIdx.x; j0 < 2 * d; j0 += NW * 64) {
        const int j = (j0 + (int)blockIdx.x * 64) % (2 * d);      // blocks start their flush at different columns (32.8 -> 32.0 us at C2)
        const int which = j >= d, col = which ? j - d : j;
        const int q = col / CH, idx = ((col % CH) * NC + q / 64) * 64 + (q & 63);      // column = ((c * 64 + lane) * CH + i)
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += red[w][which][idx];
        atomicAdd(which ? &dbeta[col] : &dgamma[col], acc);
    }
}

// ------------------------------------------------------------------ cross-entropy head
// CE_Q = 16-byte pieces per lane of the register-resident row (V <= 256 CE_Q): 4 for the 1007 / 1017 vocabularies
// CE_NW waves per block, at most one block per CU: the loss / count partials end in ONE pair of same-address global
// atomics per block, and same-address atomics serialise at ~15 ns each -- with 2048 four-wave blocks they were 55 of the
// kernel's 64 us.
constexpr int CE_NW = 16;
// LT = type of the logits: float (f32 tier, MusicRegression) or bf16 (bf16 tier: the head GEMM writes T logits, which is
// what the reference's autocast F.linear produces as well; the loss arithmetic is f32 either way)
template <typename LT, int CE_Q>
__global__ __launch_bounds__(CE_NW * 64) void ce_fwd_kernel(const LT* __restrict__ logits, int ld,
                                                     const int64_t* __restrict__ target, float* __restrict__ row_lse,
                                                     float* __restrict__ loss_sum, float* __restrict__ n_valid, int rows,
                                                     int V, int ignore_index) {
    __shared__ float red[2][CE_NW];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float bl = 0.f, bn = 0.f;
    const int64_t stride = (int64_t)gridDim.x * CE_NW;
    int64_t row = (int64_t)blockIdx.x * CE_NW + wid;
    constexpr int EP = 16 / sizeof(LT);                                   // logits per 16-byte piece
    if (V <= 64 * EP * CE_Q && (ld % EP) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        // One pass: the row (<= 2048 logits) sits in registers, 16-byte loads.  The next row (and its target) is
        // fetched BEFORE this row's lse is stored and the target logit comes out of the registers: vmcnt is in
        // order, so a load issued after a store cannot be consumed before that store has been acknowledged.
        chunk16 nv[CE_Q];
        int64_t nt = 0;
        auto fetch = [&](int64_t r) {
            r = r < rows ? r : rows - 1;                  // clamped, not skipped: loads under a branch serialise the loop
            const LT* lg = logits + r * ld;
            nt = target[r];
#pragma unroll
            for (int q = 0; q < CE_Q; ++q) {
                const int j = (q * 64 + lane) * EP;
                // rows are padded to a multiple of 16 bytes (ld % EP == 0), so the 16-byte load is always in bounds for
                // j < ld (clamped otherwise); entries at or beyond V are masked when the piece is unpacked
                nv[q] = ld_chunk(lg + (j < ld ? j : 0));
            }
        };
        fetch(row);
        for (; row < rows; row += stride) {
            float v[CE_Q][EP];
            const int64_t t = nt;
#pragma unroll
            for (int q = 0; q < CE_Q; ++q) {
                const int j = (q * 64 + lane) * EP;
#pragma unroll
                for (int i = 0; i < EP; ++i) {
                    float x;
                    if constexpr (sizeof(LT) == 4) x = reinterpret_cast<const float*>(&nv[q])[i];
                    else x = (float)reinterpret_cast<const LT*>(&nv[q])[i];
                    v[q][i] = j + i < V ? x : -INFINITY;
                }
            }
            fetch(row + stride);
            float mx = -INFINITY, se = 0.f, tv = 0.f;
#pragma unroll
            for (int q = 0; q < CE_Q; ++q) {
                const int j = (q * 64 + lane) * EP;
#pragma unroll
                for (int i = 0; i < EP; ++i) {
                    mx = fmaxf(mx, v[q][i]);
                    tv = (t == j + i) ? v[q][i] : tv;
                }
            }
            mx = wave_max(mx);
#pragma unroll
            for (int q = 0; q < CE_Q; ++q)
#pragma unroll
                for (int i = 0; i < EP; ++i) se += __builtin_amdgcn_exp2f((v[q][i] - mx) * 1.4426950408889634f);   // exp2(-inf) = 0 for the padding
            se = wave_sum(se);
            tv = wave_sum(tv);                                       // the one lane that holds logit[t]
            const float lse = mx + logf(se);
            if (lane == 0) {
                if (row_lse) row_lse[row] = lse;
                if (t != ignore_index) { bl += lse - tv; bn += 1.f; }
            }
        }
    } else {
        for (; row < rows; row += stride) {
            const LT* lg = logits + row * ld;
            float mx = -INFINITY, se = 0.f;
            for (int j = lane; j < V; j += 64) mx = fmaxf(mx, (float)lg[j]);
            mx = wave_max(mx);
            for (int j = lane; j < V; j += 64) se += expf((float)lg[j] - mx);
            se = wave_sum(se);
            const float lse = mx + logf(se);
            if (lane == 0) {
                if (row_lse) row_lse[row] = lse;
                const int64_t t = target[row];
                if (t != ignore_index) { bl += lse - (float)lg[t]; bn += 1.f; }
            }
        }
    }
    if (lane == 0) { red[0][wid] = bl; red[1][wid] = bn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sl = 0.f, sn = 0.f;
#pragma unroll
        for (int w = 0; w < CE_NW; ++w) { sl += red[0][w]; sn += red[1][w]; }
        atomicAdd(loss_sum, sl);
        atomicAdd(n_valid, sn);
    }
}

// dbias (optional): the head's bias gradient = column sums of the f32 dlogits BEFORE they are rounded to T -- summing
// the bf16-rounded dlogits in the weight-gradient GEMM gave that tensor a 2e-3 relative error against 4e-4 for the
// reference's own autocast; here every lane keeps f32 partial sums of its columns over the rows of its wave (16-byte
// path: <= 4 chunks of 8 columns per lane), waves are combined through LDS and a block issues one atomic per column.
// HAS_DBIAS is a template flag: the 33.8 KB of LDS and the 32 partial-sum registers exist only in the instantiation that
// emits the bias gradient (the f32 tier and the autograd path run without: ADVICE r3).
template <typename T, typename LT, bool HAS_DBIAS>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const LT* __restrict__ logits, int ld,
                                                     const int64_t* __restrict__ target, const float* __restrict__ row_lse,
                                                     T* __restrict__ dlogits, int ld_d, const float* __restrict__ n_valid,
                                                     float extra_scale, const float* __restrict__ loss_scale, int rows, int V,
                                                     int ignore_index, float* __restrict__ dbias) {
    __shared__ float red[HAS_DBIAS ? 4 : 1][HAS_DBIAS ? 64 * 33 : 1];     // dbias: [wave][lane][32 partial sums], +1 padding
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // n_valid == 0 -> inf/nan like torch's 0/0 mean; loss_scale: the dynamic loss scale of the f16 tier (me_scaler_step), device resident
    const float scale = extra_scale * (loss_scale ? *loss_scale : 1.f) / fmaxf(*n_valid, 0.f);
    float bs[HAS_DBIAS ? 4 : 1][8];
#pragma unroll
    for (int c = 0; c < (HAS_DBIAS ? 4 : 1); ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[c][e] = 0.f;
    bool fast = false, wide = false;
    if constexpr (sizeof(LT) == 2 && sizeof(T) == 2) {
        // both 16-bit: 8 columns per lane and 16-byte access (2-byte accesses made this kernel issue bound: 37 us
        // for 133 MB); columns in [V, ld) of the logits are never used, columns in [V, ld_d) are written as 0
        const bool ok16 = (ld & 7) == 0 && (ld_d & 7) == 0 && ld >= ld_d &&
                          ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
        fast = ok16 && ld_d <= 2048;
        wide = ok16 && !fast && !HAS_DBIAS;                // vocabularies above 2048 columns: unbounded 16-byte column loop
    }
    if (wide) {
        for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
            const int64_t t = target[row];
            const float lse = row_lse[row];
            const bool valid = t != ignore_index;
            for (int j = lane * 8; j < ld_d; j += 512) {
                const chunk16 in = ld_chunk(logits + row * ld + j);
                chunk16 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g = 0.f;
                    if (valid && j + e < V)
                        g = (__builtin_amdgcn_exp2f(((float)reinterpret_cast<const LT*>(&in)[e] - lse) * 1.4426950408889634f) -
                             (j + e == t ? 1.f : 0.f)) * scale;
                    reinterpret_cast<T*>(&o)[e] = ET<T>::from_f(g);
                }
                st_chunk(dlogits + row * ld_d + j, o);
            }
        }
    } else if (fast) {
        // RU rows of a wave in flight at once (2 x RU 16-byte loads per lane at ld_d = 1024): the bias-gradient variant runs on
        // 256 blocks (one atomic per column and block) and would otherwise leave the memory pipe three quarters empty (65 us)
        constexpr int RU = 4;
        const int64_t stride = (int64_t)gridDim.x * 4;
        for (int64_t row0 = (int64_t)blockIdx.x * 4 + wid; row0 < rows; row0 += stride * RU) {
            chunk16 ch[RU][4];
            int64_t tg[RU];
            float ls[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t rc = min(row0 + u * stride, (int64_t)rows - 1);
                tg[u] = target[rc];
                ls[u] = row_lse[rc];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = lane * 8 + 512 * c;
                    if (j < ld_d) ch[u][c] = ld_chunk(logits + rc * ld + j);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t row = row0 + u * stride;
                if (row >= rows) break;
                const bool valid = tg[u] != ignore_index;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = lane * 8 + 512 * c;
                    if (j >= ld_d) break;
                    chunk16 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float g = 0.f;
                        if (valid && j + e < V)
                            g = (__builtin_amdgcn_exp2f(((float)reinterpret_cast<const LT*>(&ch[u][c])[e] - ls[u]) * 1.4426950408889634f) -
                                 (j + e == tg[u] ? 1.f : 0.f)) * scale;
                        if constexpr (HAS_DBIAS) bs[c][e] += g;
                        reinterpret_cast<T*>(&o)[e] = ET<T>::from_f(g);
                    }
                    st_chunk(dlogits + row * ld_d + j, o);
                }
            }
        }
    } else {
        for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
            const LT* lg = logits + row * ld;
            const int64_t t = target[row];
            const float lse = row_lse[row];
            const bool valid = t != ignore_index;
            for (int j = lane; j < ld_d; j += 64) {
                float g = 0.f;
                if (valid && j < V) g = (expf((float)lg[j] - lse) - (j == t ? 1.f : 0.f)) * scale;
                dlogits[row * ld_d + j] = ET<T>::from_f(g);
            }
        }
    }
    if constexpr (HAS_DBIAS) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wid][lane * 33 + c * 8 + e] = bs[c][e];
        __syncthreads();
        for (int col = threadIdx.x; col < V; col += 256) {     // column = 512 c + 8 lane' + e
            const int c = col >> 9, l2 = (col & 511) >> 3, e = col & 7;
            const int idx = l2 * 33 + c * 8 + e;
            atomicAdd(&dbias[col], red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx]);
        }
    }
}

// ------------------------------------------------------------------ clip + AdamW
// <= 256 blocks of 16 waves.  The block sums are combined in a FIXED order: every block leaves its sum in the caller's
// workspace, takes a ticket, and the block that draws the last ticket adds the slots 0 .. grid-1 in index order (one wave,
// the same instruction sequence whatever the arrival order) -- so that the clip coefficient, and with it every parameter
// after the update, is bit-identical on all data-parallel ranks (they hold bit-identical reduced gradients).  Round 3 added
// the block sums with one same-address atomic per block: ranks then differed in the last bit of the coefficient, a few
// thousand parameters per step moved to the other f32 neighbour and every few steps one of them crossed a bf16 rounding
// boundary on one rank only (round-4 DDP diagnosis, profiles/r04_ddp_diagnosis.txt).  part == nullptr: the atomic variant.
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out,
                                                     float* __restrict__ part, unsigned int* __restrict__ ticket) {
    __shared__ float red[16];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 1024) {
        const f32x4_t v = g4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        if (part == nullptr) {
            atomicAdd(out, t);
            last = false;
        } else {
            __hip_atomic_store(&part[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // release: the slot is visible device-wide before the ticket is; acquire on the same atomic for the last block
            const unsigned int tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = tk == gridDim.x - 1;
        }
    }
    __syncthreads();
    if (!last || wid != 0) return;
    float t = 0.f;
    for (int i = lane; i < (int)gridDim.x; i += 64)          // grid <= 256: at most four slots per lane, index order
        t += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = wave_sum(t);
    if (lane == 0) {
        *out += t;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // left zeroed for the next call
    }
}

// GradScaler.update() on the device (torch/cuda/amp/grad_scaler.py semantics; train.py:101,317-324): one thread.
// state: see ME_SCALER_* in midiemo.h.  sumsq = squared norm of the SCALED gradients (me_sumsq).
__global__ void scaler_step_kernel(float* __restrict__ state, const float* __restrict__ sumsq, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float scale = state[ME_SCALER_SCALE];
    const float ss = *sumsq;
    const bool bad = !(ss == ss) || ss == INFINITY || ss == -INFINITY;
    state[ME_SCALER_INV] = 1.f / scale;
    state[ME_SCALER_FOUND_INF] = bad ? 1.f : 0.f;
    if (bad) {
        state[ME_SCALER_SCALE] = scale * backoff;
        state[ME_SCALER_TRACKER] = 0.f;
        state[ME_SCALER_SKIPPED] += 1.f;
    } else {
        state[ME_SCALER_STEP] += 1.f;
        const float tr = state[ME_SCALER_TRACKER] + 1.f;
        if (tr >= (float)interval) {
            const float grown = scale * growth;
            state[ME_SCALER_SCALE] = (grown == INFINITY) ? scale : grown;       // torch: the scale never grows to inf
            state[ME_SCALER_TRACKER] = 0.f;
        } else state[ME_SCALER_TRACKER] = tr;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ sumsq,
                                                    float clip, float grad_scale, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, int zero_grad, const float* __restrict__ scaler) {
    // scaler (f16 tier, me_scaler_step ran just before): the gradients carry the loss scale -> unscale with ME_SCALER_INV; a
    // non-finite gradient norm (ME_SCALER_FOUND_INF) skips the update like GradScaler.step (train.py:322), only the
    // gradients are cleared; the bias corrections come from the device-side count of the steps actually taken
    if (scaler) {
        grad_scale *= scaler[ME_SCALER_INV];
        const double t = (double)scaler[ME_SCALER_STEP];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2 = (float)(1.0 - pow((double)b2, t));
        if (scaler[ME_SCALER_FOUND_INF] != 0.f) {
            if (zero_grad) {
                for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] = 0.f;
            }
            return;
        }
    }
    float coef = grad_scale;
    if (clip > 0.f && sumsq) {
        const float total = sqrtf(*sumsq) * fabsf(grad_scale);
        coef *= fminf(1.f, clip / (total + 1e-6f));
    }
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp = pp * decay - step_size * (mm / denom);
    };
    const int64_t n4 = n >> 2;
    f32x4_t* p4 = reinterpret_cast<f32x4_t*>(p);
    f32x4_t* g4 = reinterpret_cast<f32x4_t*>(g);
    f32x4_t* m4 = reinterpret_cast<f32x4_t*>(m);
    f32x4_t* v4 = reinterpret_cast<f32x4_t*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4_t pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe_ = pp[e], me_ = mm[e], ve_ = vv[e];
            upd(pe_, gg[e], me_, ve_);
            pp[e] = pe_; mm[e] = me_; vv[e] = ve_;
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad) g4[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        upd(p[i], g[i], m[i], v[i]);
        if (zero_grad) g[i] = 0.f;
    }
}

// ------------------------------------------------------------------ greedy pick (generate top_k=1)
__global__ __launch_bounds__(256) void greedy_pick_kernel(const float* __restrict__ logits, int ld, int V,
                                                          const int32_t* __restrict__ special, int n_special,
                                                          int64_t* __restrict__ out_ids) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* lg = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int j = threadIdx.x; j < V; j += 256) {
        float v = lg[j];
        if (v != v) v = 0.f;                                  // generate.py:123 (NaN -> 0)
        for (int s = 0; s < n_special; ++s) if (special[s] == j) v = -INFINITY;   // generate.py:131-136
        if (v > best || (v == best && j < besti)) { best = v; besti = j; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float ov = bv[threadIdx.x + o]; const int oi = bi[threadIdx.x + o];
            if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out_ids[blockIdx.x] = bi[0] == 0x7fffffff ? 0 : bi[0];
}

// The same pick for all B rows in ONE block (wave per row), followed by the decode bookkeeping of
// decode_commit_kernel: history[b][*pos] = id ; *pos += 1 -- one launch instead of two at the end of a greedy step.
__global__ __launch_bounds__(256) void greedy_pick_commit_kernel(const float* __restrict__ logits, int ld, int V,
                                                                 const int32_t* __restrict__ special, int n_special,
                                                                 int64_t* __restrict__ out_ids, int64_t* __restrict__ history,
                                                                 int ld_hist, int32_t* __restrict__ pos, int B) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p = *pos;
    for (int b = wid; b < B; b += 4) {
        const float* lg = logits + (int64_t)b * ld;
        float best = -INFINITY;
        int besti = 0x7fffffff;
        for (int j0 = lane; j0 < V; j0 += 64 * 16) {             // 16 independent loads in flight (V <= 1024: one round)
            float vals[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) vals[u] = lg[min(j0 + 64 * u, V - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + 64 * u;
                float v = vals[u];
                if (v != v) v = 0.f;
                for (int s = 0; s < n_special; ++s) if (special[s] == j) v = -INFINITY;
                if (j < V && (v > best || (v == best && j < besti))) { best = v; besti = j; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(besti, o);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if (lane == 0) {
            const int64_t id = besti == 0x7fffffff ? 0 : besti;
            out_ids[b] = id;
            if (p < ld_hist) history[(size_t)b * ld_hist + p] = id;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *pos = p + 1;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int row_grid(int64_t rows, int cap) { int64_t g = (rows + 3) / 4; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace

#define ME_DISPATCH(dtype, CALL)                         \
    if ((dtype) == ME_F32) { typedef float T; CALL; }    \
    else if ((dtype) == ME_BF16) { typedef bf16_t T; CALL; } \
    else if ((dtype) == ME_F16) { typedef f16_t T; CALL; } \
    else return ME_ERR_BAD_DTYPE;


// decode bookkeeping on the device (so that a whole greedy step can be replayed as one HIP graph):
// history[b][*pos] = tok[b];  *pos += 1
__global__ void decode_commit_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ history, int ld_hist,
                                     int32_t* __restrict__ pos, int B) {
    const int p = *pos;
    if ((int)threadIdx.x < B && p < ld_hist) history[(size_t)threadIdx.x * ld_hist + p] = tok[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) *pos = p + 1;
}


// ------------------------------------------------------------------ sampling tail of generate()
// One block per sequence, V <= 1024.  Mirrors generate.py:122-189 for one step: NaN -> 0, special ids -> -inf,
// log_softmax, divide by the row's temperature, keep the top_k largest (all if top_k <= 0), nucleus cut (drop
// sorted entries whose cumulative probability exceeds top_p, never the first), renormalise, draw by inverse CDF
// from the caller's uniform u[b] in [0,1), report n_choices = #(probability > 0).  Descending order comes from
// a bitonic sort of (value, index) pairs in LDS (ties: lower index first).  dbg_p / dbg_i (optional, [B][1024])
// receive the final sorted probabilities and their vocabulary ids so tests can compare with the torch path.
// Step mode (me_sample_step, st != NULL): the per-row temperature of generate.py:138-163 is computed here from the token
// that was just fed (note temperature right after a TIMESHIFT, else the rest temperature, raised by
// max(0, log((repeats + 1) / 4) * penalty) times itself -- the same f32 operations, unfused, as the torch expression of
// generate.py's sampling_temperature()), the uniform comes from row (*pos - pos0) of a table drawn in advance, and the
// repeat counter is updated (generate.py:186-189) -- so a whole sampled decode step can be replayed as one HIP graph.
struct SampleStep {
    const int64_t* prev_tok;        // [B] token fed at this step
    const uint8_t* is_timeshift;    // [V]
    float* repeat_counts;           // [B] in / out
    const int32_t* pos;             // device position counter
    int pos0, u_ld;
    float temp_note, temp_rest, penalty;
};

// NP = sort width (power of two >= V: 1024 for the reference's vocabularies, up to 4096), EPT = NP / 256 sorted entries per thread
template <int NP>
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, int ld, int V,
                                                     const int32_t* __restrict__ special, int n_special,
                                                     const float* __restrict__ temp, int top_k, float top_p,
                                                     const float* __restrict__ u, int64_t* __restrict__ out_ids,
                                                     int32_t* __restrict__ n_choices, float* __restrict__ dbg_p,
                                                     int32_t* __restrict__ dbg_i, SampleStep st) {
    constexpr int EPT = NP / 256;
    // key / idx double as the exchange buffer of the sort's three cross-wave stages (one 64-bit word per entry)
    __shared__ unsigned long long xbuf[NP];
    float* const key = reinterpret_cast<float*>(xbuf);
    int* const idx = reinterpret_cast<int*>(xbuf) + NP;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    const float* lg = logits + (size_t)b * ld;
    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wid] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    // inclusive prefix over the block's 256 threads: shuffle scan inside the wave, the three wave totals in front added in order
    auto block_scan = [&](float v) -> float {
        float x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        __syncthreads();
        if (lane == 63) red[wid] = x;
        __syncthreads();
        float base = 0.f;
        for (int w = 0; w < wid; ++w) base += red[w];
        return x + base;
    };
    auto block_max = [&](float v) -> float {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) red[wid] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    // ---- load, NaN -> 0, specials -> -inf
    for (int j = tid; j < NP; j += 256) {
        float x = -INFINITY;
        if (j < V) { x = lg[j]; if (x != x) x = 0.f; }
        key[j] = x;
        idx[j] = j;
    }
    __syncthreads();
    for (int j = tid; j < n_special; j += 256) { const int sidx = special[j]; if (sidx >= 0 && sidx < V) key[sidx] = -INFINITY; }
    __syncthreads();
    // ---- log_softmax, / temperature
    float mx = -INFINITY;
    for (int j = tid; j < NP; j += 256) mx = fmaxf(mx, key[j]);
    mx = block_max(mx);
    float se = 0.f;
    for (int j = tid; j < NP; j += 256) se += expf(key[j] - mx);
    se = block_sum(se);
    float row_temp, row_u;
    if (st.prev_tok) {
        const int64_t pt = st.prev_tok[b];
        row_temp = (pt >= 0 && pt < V && st.is_timeshift[pt]) ? st.temp_note : st.temp_rest;
        if (st.penalty > 0.f) {
            const float mult = fmaxf(__fmul_rn(logf(__fdiv_rn(__fadd_rn(st.repeat_counts[b], 1.f), 4.f)), st.penalty), 0.f);
            row_temp = __fadd_rn(row_temp, __fmul_rn(mult, row_temp));
        }
        row_u = u[(size_t)(st.pos[0] - st.pos0) * st.u_ld + b];
    } else {
        row_temp = temp[b];
        row_u = u[b];
    }
    const float lse = mx + logf(se), inv_t = 1.f / row_temp;
    for (int j = tid; j < NP; j += 256) key[j] = (key[j] - lse) * inv_t;
    __syncthreads();
    // ---- bitonic sort, descending by value, ascending index among equals -- in REGISTERS (round 5: the LDS version took a
    // barrier per compare stage, 55 at NP = 1024, ~25 of the kernel's 33 us).  Thread tid owns entries tid EPT .. + EPT - 1; an entry
    // is ONE 64-bit word (order-preserving image of the float << 32 | ~index: "greater word" = "earlier in the order", ties
    // included), a stage with partner distance j < EPT is a compare-exchange inside the thread, j < 64 EPT a lane shuffle
    // (partner lane ^ (j / EPT)), and only the last three stage groups (j >= 64 EPT: partner in another wave) go through LDS.
    unsigned long long sv[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;
        uint32_t bts = __float_as_uint(key[i] + 0.f);                       // + 0: one zero
        bts ^= (bts >> 31) ? 0xffffffffu : 0x80000000u;
        sv[q] = ((unsigned long long)bts << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
    }
    __syncthreads();                                                        // key / idx are free: the exchange buffer from here on
#pragma unroll
    for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < EPT) {
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    if (q & j) continue;
                    const bool desc = ((tid * EPT + q) & k) == 0;
                    const unsigned long long a = sv[q], c = sv[q | j];
                    const unsigned long long hi = a > c ? a : c, lo = a > c ? c : a;
                    sv[q] = desc ? hi : lo;
                    sv[q | j] = desc ? lo : hi;
                }
            } else {
                unsigned long long ot[EPT];
                if (j < 64 * EPT) {
#pragma unroll
                    for (int q = 0; q < EPT; ++q) {
                        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)sv[q], j / EPT);
                        const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(sv[q] >> 32), j / EPT);
                        ot[q] = ((unsigned long long)ohi << 32) | olo;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < EPT; ++q) xbuf[tid * EPT + q] = sv[q];
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < EPT; ++q) ot[q] = xbuf[(tid ^ (j / EPT)) * EPT + q];
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = tid * EPT + q;
                    const bool keep_max = ((i & j) == 0) == ((i & k) == 0);   // the lower position of a descending pair keeps the greater word
                    const bool mine_gt = sv[q] > ot[q];
                    sv[q] = (keep_max == mine_gt) ? sv[q] : ot[q];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        uint32_t bts = (uint32_t)(sv[q] >> 32);
        bts ^= (bts >> 31) ? 0x80000000u : 0xffffffffu;
        key[tid * EPT + q] = __uint_as_float(bts);
        idx[tid * EPT + q] = (int)(0xffffffffu - (uint32_t)sv[q]);
    }
    __syncthreads();
    // ---- top-k, softmax over the kept head, nucleus cut
    const int k_eff = (top_k <= 0 || top_k > V) ? V : top_k;
    const float y0 = key[0];
    float e[EPT];
    float loc = 0.f;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;
        e[q] = i < k_eff ? expf(key[i] - y0) : 0.f;
        loc += e[q];
    }
    const float tot = block_sum(loc);
    // inclusive prefix of the chunk sums (each thread owns EPT consecutive sorted entries)
    const float incl1 = block_scan(loc);
    float run = (incl1 - loc) / tot;
    float loc2 = 0.f;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;
        run += e[q] / tot;                                                   // cumulative probability incl. entry i
        const bool cut = top_p > 0.f && top_p < 1.f && i > 0 && run > top_p;
        if (cut || i >= k_eff) e[q] = 0.f;
        loc2 += e[q];
    }
    const float tot2 = block_sum(loc2);
    // ---- renormalised CDF and inverse-CDF draw
    const float incl2 = block_scan(loc2);
    const float target = row_u * tot2;
    float c0 = incl2 - loc2;
    int cnt = 0, pick = -1;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;
        const float pq = e[q] / tot2;
        if (pq > 0.f) ++cnt;
        const float c1 = c0 + e[q];
        if (pick < 0 && e[q] > 0.f && target < c1 && target >= c0) pick = i;
        c0 = c1;
        if (dbg_p) { dbg_p[(size_t)b * NP + i] = pq; dbg_i[(size_t)b * NP + i] = idx[i]; }
    }
    // number of choices and the (unique) picked position; fall back to the last positive entry for u ~ 1
    __shared__ int s_pick, s_cnt, s_last;
    if (tid == 0) { s_pick = 1 << 30; s_cnt = 0; s_last = 0; }
    __syncthreads();
    // one LDS atomic of each kind per WAVE (a thousand atomicMax on one word cost ~8 us with every entry surviving)
    int w_pick = pick >= 0 ? pick : (1 << 30), w_cnt = cnt, w_last = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) if (e[q] > 0.f) w_last = tid * EPT + q;      // ascending in q: the thread's last positive entry
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        w_pick = min(w_pick, __shfl_xor(w_pick, m));
        w_cnt += __shfl_xor(w_cnt, m);
        w_last = max(w_last, __shfl_xor(w_last, m));
    }
    if (lane == 0) { atomicMin(&s_pick, w_pick); atomicAdd(&s_cnt, w_cnt); atomicMax(&s_last, w_last); }
    __syncthreads();
    if (tid == 0) {
        const int pos = s_pick < (1 << 30) ? s_pick : s_last;
        out_ids[b] = idx[pos];
        if (n_choices) n_choices[b] = s_cnt;
        if (st.prev_tok) {                                  // generate.py:186-189
            const float rc = st.repeat_counts[b];
            st.repeat_counts[b] = s_cnt <= 2 ? rc + 1.f : floorf(rc * 0.5f);
        }
    }
}

extern "C" {

int me_embed_fwd(void* out, void* out_lo, int dtype, const int64_t* tokens, const float* cond, const float* emb, const float* cw0,
                 const float* cb0, const float* cw1, const float* cb1, const float* pe, const int32_t* pos_dev, int mode, int B, int Ltok,
                 int d, int dc, float p, uint64_t seed, void* stream) {
    me_clear_error();
    const bool lo8 = (dtype & ME_LO8) != 0;
    dtype &= ~ME_LO8;
    if (lo8 && dtype == ME_F32) return ME_ERR_BAD_DTYPE;
    if (!out || !tokens || !emb || !pe) return ME_ERR_NULL;
    if (mode == ME_COND_CONCAT && (!cond || !cw0 || !cb0 || dc <= 0 || dc >= d)) return ME_ERR_NULL;
    if (mode == ME_COND_TOKEN && (!cond || !cw0 || !cb0 || !cw1 || !cb1)) return ME_ERR_NULL;
    if (mode != ME_COND_CONCAT) dc = 0;
    if (B <= 0 || Ltok < 0 || d <= 0 || d % 8) return ME_ERR_BAD_SHAPE;
    if (!aligned16(out) || (out_lo && !aligned16(out_lo))) return ME_ERR_ALIGNMENT;
    const int Lm = Ltok + (mode == ME_COND_TOKEN ? 2 : 0);
    const int64_t rows = (int64_t)B * Lm;
    if (rows == 0) return ME_OK;
    const uint32_t thr = thr_of(p);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    hipStream_t st = (hipStream_t)stream;
    if (lo8) {
        ME_DISPATCH(dtype, (embed_fwd_kernel<T, true><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(
                               (T*)out, (T*)out_lo, tokens, cond, emb, cw0, cb0, cw1, cb1, pe, pos_dev, mode, B, Ltok, d, dc, thr, inv_keep, seed)));
        return me_launch_status();
    }
    ME_DISPATCH(dtype, (embed_fwd_kernel<T><<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(
                           (T*)out, (T*)out_lo, tokens, cond, emb, cw0, cb0, cw1, cb1, pe, pos_dev, mode, B, Ltok, d, dc, thr, inv_keep, seed)));
    return me_launch_status();
}

int me_embed_bwd(const void* dout, int dtype, const int64_t* tokens, const float* cond, float* g_emb, float* g_cw0,
                 float* g_cb0, float* g_cw1, float* g_cb1, int mode, int B, int Ltok, int d, int dc, int vocab, int pad_token,
                 float p, uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
    me_clear_error();
    if (!dout || !tokens || !g_emb) return ME_ERR_NULL;
    if (ws && (ws_bytes < 1024 || (reinterpret_cast<uintptr_t>(ws) & 15))) return ME_ERR_WORKSPACE;
    if (mode == ME_COND_CONCAT && (!cond || !g_cw0 || !g_cb0)) return ME_ERR_NULL;
    if (mode == ME_COND_TOKEN && (!cond || !g_cw0 || !g_cb0 || !g_cw1 || !g_cb1)) return ME_ERR_NULL;
    if (mode != ME_COND_CONCAT) dc = 0;
    if (d % 8 || d > 64 * MAXC * 4) return ME_ERR_BAD_SHAPE;
    const int Lm = Ltok + (mode == ME_COND_TOKEN ? 2 : 0);
    if (B <= 0 || Lm <= 0) return ME_OK;
    const uint32_t thr = thr_of(p);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    int nslab = (Lm + 63) / 64;
    if (nslab > 32) nslab = 32;
    dim3 grid(nslab, B);
    hipStream_t st = (hipStream_t)stream;
    const int de = d - dc;
    const bool table = vocab > 0 && (size_t)vocab * 32 <= 65536 && Ltok > 0 && (de & 7) == 0;
    const int64_t rows64 = (int64_t)B * Ltok;
    // one gather block per table row (16-bit: de <= 1024 columns, f32: 512)
    const bool grouped = vocab > 0 && Ltok > 0 && (de & 7) == 0 && de / (dtype == ME_F32 ? 4 : 8) <= 128 && rows64 < (1ll << 30);
    if (grouped) {
        const int shift = mode == ME_COND_TOKEN ? 2 : 0;
#ifndef EMB_TPB
#define EMB_TPB 2
#endif
        int* hv = reinterpret_cast<int*>(ws);
        ME_DISPATCH(dtype, (embed_bwd_gather_kernel<T, EMB_TPB><<<(vocab + EMB_TPB - 1) / EMB_TPB, 256, 0, st>>>(
                               (const T*)dout, tokens, g_emb, hv, (int)rows64, Ltok, shift, d, de, vocab, pad_token, thr, inv_keep, seed)));
        if (hv) {
            ME_DISPATCH(dtype, (embed_bwd_heavy_kernel<T><<<256, 256, 0, st>>>((const T*)dout, tokens, g_emb, hv, (int)rows64, Ltok, shift, d,
                                                                               de, thr, inv_keep, seed)));
        }
    }
    if (table || grouped) {
        if (!grouped) {
        int ysplit = (int)(((int64_t)B * Ltok + 256 * 16 - 1) / (256 * 16));
        if (ysplit < 1) ysplit = 1;
        if (ysplit > 16) ysplit = 16;
        dim3 g2(de / 8, ysplit);
        const int shift = mode == ME_COND_TOKEN ? 2 : 0;
        ME_DISPATCH(dtype, (embed_bwd_table_kernel<T><<<g2, 256, (size_t)vocab * 32, st>>>((const T*)dout, tokens, g_emb, B, Ltok,
                                                                                           shift, d, de, vocab, pad_token, thr,
                                                                                           inv_keep, seed)));
        }
        int rc = me_launch_status();
        if (rc) return rc;
        if (mode == ME_COND_NONE) return ME_OK;        // nothing but the table to differentiate
        const int chx = dtype == ME_F32 ? 4 : 8;
        if (mode == ME_COND_CONCAT && dc > 0 && dc % chx == 0 && dc / chx <= 64 && 64 % (dc / chx) == 0) {
            const int rpw = 64 / (dc / chx);
            int64_t nb = ((int64_t)B * Lm + 8 * rpw * 8 - 1) / (8 * rpw * 8);         // >= 8 passes per wave
            if (nb > 128) nb = 128;
            if (nb < 1) nb = 1;
            ME_DISPATCH(dtype, (embed_bwd_cond_kernel<T><<<(unsigned)nb, 512, 0, st>>>((const T*)dout, cond, g_cw0, g_cb0, B, Lm, d,
                                                                                       dc, thr, inv_keep, seed)));
            return me_launch_status();
        }
    }
    ME_DISPATCH(dtype, (embed_bwd_kernel<T><<<grid, 256, 0, st>>>((const T*)dout, tokens, cond, g_emb, g_cw0, g_cb0, g_cw1,
                                                                  g_cb1, mode, B, Ltok, d, dc, pad_token, thr, inv_keep, seed,
                                                                  (table || grouped) ? 1 : 0)));
    return me_launch_status();
}

int me_key_pad_mask(uint8_t* key_pad, const int64_t* tokens, int B, int Ltok, int shift, int pad_token, void* stream) {
    me_clear_error();
    if (!key_pad || !tokens) return ME_ERR_NULL;
    const int64_t n = (int64_t)B * (Ltok + shift);
    if (n <= 0) return ME_OK;
    key_pad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(key_pad, tokens, B, Ltok, shift, pad_token);
    return me_launch_status();
}

int me_resid_ln_fwd(const void* x, const void* x_lo, const void* a, const float* gamma, const float* beta, void* y,
                    void* y_lo, void* s_out, float* stats, int rows, int d, float eps, float p, uint64_t seed,
                    uint32_t site, int dtype, void* stream) {
    me_clear_error();
    const bool lo8 = (dtype & ME_LO8) != 0;
    dtype &= ~ME_LO8;
    if (lo8 && dtype == ME_F32) return ME_ERR_BAD_DTYPE;
    if (!x || !a || !gamma || !beta || !y) return ME_ERR_NULL;
    if (rows <= 0) return ME_OK;
    const int ch = dtype == ME_F32 ? 4 : 8;
    if (d <= 0 || d % ch || d > 64 * MAXC * ch) return ME_ERR_BAD_SHAPE;
    if (!aligned16(x) || !aligned16(a) || !aligned16(y) || (s_out && !aligned16(s_out)) || (x_lo && !aligned16(x_lo)) ||
        (y_lo && !aligned16(y_lo)))
        return ME_ERR_ALIGNMENT;
    const uint32_t thr = thr_of(p);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    hipStream_t st = (hipStream_t)stream;
    if (d == 64 * ch) {
        // one chunk per lane and row (d = 512 bf16): two rows per wave in flight, next pair prefetched -- 42.8 -> 37.9 us at
        // T = 32768 (5.3 TB/s of its 201 MB; rows per wave 1 / 2 / 4: 38.8 / 37.9 / 40.4, grid 512 .. 8192 flat within 1.5 us;
        // non-temporal loads / stores: 39.9)
        if (lo8) {
            ME_DISPATCH(dtype, (resid_ln_fwd1_kernel<T, 2, true><<<row_grid((rows + 1) / 2, 4096), 256, 0, st>>>(
                                   (const T*)x, (const T*)x_lo, (const T*)a, gamma, beta, (T*)y, (T*)y_lo, (T*)s_out, stats, rows, eps, thr,
                                   inv_keep, seed, site)));
            return me_launch_status();
        }
        ME_DISPATCH(dtype, (resid_ln_fwd1_kernel<T, 2><<<row_grid((rows + 1) / 2, 4096), 256, 0, st>>>(
                               (const T*)x, (const T*)x_lo, (const T*)a, gamma, beta, (T*)y, (T*)y_lo, (T*)s_out, stats, rows, eps, thr,
                               inv_keep, seed, site)));
        return me_launch_status();
    }
    if (lo8) {
        ME_DISPATCH(dtype, (resid_ln_fwd_kernel<T, true><<<row_grid(rows, 8192), 256, 0, st>>>(
                               (const T*)x, (const T*)x_lo, (const T*)a, gamma, beta, (T*)y, (T*)y_lo, (T*)s_out, stats, rows, d, eps, thr,
                               inv_keep, seed, site)));
        return me_launch_status();
    }
    ME_DISPATCH(dtype, (resid_ln_fwd_kernel<T><<<row_grid(rows, 8192), 256, 0, st>>>(
                           (const T*)x, (const T*)x_lo, (const T*)a, gamma, beta, (T*)y, (T*)y_lo, (T*)s_out, stats, rows, d, eps, thr,
                           inv_keep, seed, site)));
    return me_launch_status();
}

int me_resid_ln_bwd(const void* dy, const void* s, const float* stats, const float* gamma, void* dx, void* da,
                    float* dgamma, float* dbeta, int rows, int d, float p, uint64_t seed, uint32_t site, int dtype,
                    void* stream) {
    me_clear_error();
    if (!dy || !s || !stats || !gamma || !dx || !da || !dgamma || !dbeta) return ME_ERR_NULL;
    if (rows <= 0) return ME_OK;
    const int ch = dtype == ME_F32 ? 4 : 8;
    if (d <= 0 || d % ch || d > 64 * MAXC * ch) return ME_ERR_BAD_SHAPE;
    if (!aligned16(dy) || !aligned16(s) || !aligned16(dx) || !aligned16(da)) return ME_ERR_ALIGNMENT;
    const uint32_t thr = thr_of(p);
    const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    hipStream_t st = (hipStream_t)stream;
    // chunks per lane per row: 1 for d <= 64 CH (512 bf16), 2, or 4 -- fewer chunks = less shared memory / registers
    ME_DISPATCH(dtype, ({
        const int nc = (d + 64 * ET<T>::CH - 1) / (64 * ET<T>::CH);
        // 16 waves per block for one chunk per lane; wider rows get 8-wave blocks (twice the blocks): a 1024-thread block
        // caps the kernel at 128 VGPRs and the 2- and 4-chunk variants spilled to scratch under it
        auto launch = [&](auto nc_tag, auto nw_tag, auto r_tag) {
            constexpr int NC = decltype(nc_tag)::value, NW = decltype(nw_tag)::value, R = decltype(r_tag)::value;
            int64_t g = (rows + NW * R - 1) / (NW * R);
            const int cap = 256 * 16 / NW;            // 256 x 16 waves: 30.5 us at C2 (512 blocks: 38.5; 8-wave blocks: 32.0)
            const int grid = (int)(g < 1 ? 1 : (g > cap ? cap : g));
            resid_ln_bwd_kernel<T, NC, NW, R><<<grid, NW * 64, 0, st>>>(
                (const T*)dy, (const T*)s, stats, gamma, (T*)dx, (T*)da, dgamma, dbeta, rows, d, thr, inv_keep, seed, site);
        };
        using I1 = std::integral_constant<int, 1>;
        // two rows per wave and iteration (independent reductions overlap): 34.3 -> 33.0 us at C2
        // (four rows per iteration spill under the 128-register cap of a 16-wave block: 77 us)
        if (nc <= 1) launch(I1{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
        else if (nc == 2) launch(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{}, I1{});
        else launch(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{}, I1{});
    }));
    return me_launch_status();
}

int me_ce_fwd(const void* logits, int ld, const int64_t* target, float* row_lse, float* loss_sum, float* n_valid,
              int rows, int V, int ignore_index, int logits_dtype, void* stream) {
    me_clear_error();
    if (!logits || !target || !loss_sum || !n_valid) return ME_ERR_NULL;
    if (logits_dtype != ME_F32 && logits_dtype != ME_BF16 && logits_dtype != ME_F16) return ME_ERR_BAD_DTYPE;
    if (rows <= 0) return ME_OK;
    if (V <= 0 || ld < V) return ME_ERR_BAD_SHAPE;
    const int64_t gb = ((int64_t)rows + CE_NW - 1) / CE_NW;
    const int grid = (int)(gb > 256 ? 256 : gb);
    hipStream_t st = (hipStream_t)stream;
    if (logits_dtype == ME_F32) {
        const float* lg = (const float*)logits;
        if (V <= 1024) ce_fwd_kernel<float, 4><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
        else ce_fwd_kernel<float, 8><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
    } else if (logits_dtype == ME_BF16) {
        const bf16_t* lg = (const bf16_t*)logits;
        if (V <= 1024) ce_fwd_kernel<bf16_t, 2><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
        else ce_fwd_kernel<bf16_t, 4><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
    } else {
        const f16_t* lg = (const f16_t*)logits;
        if (V <= 1024) ce_fwd_kernel<f16_t, 2><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
        else ce_fwd_kernel<f16_t, 4><<<grid, CE_NW * 64, 0, st>>>(lg, ld, target, row_lse, loss_sum, n_valid, rows, V, ignore_index);
    }
    return me_launch_status();
}

int me_ce_bwd(const void* logits, int ld, const int64_t* target, const float* row_lse, void* dlogits, int ld_d,
              const float* n_valid, float extra_scale, const float* loss_scale_dev, float* dbias, int rows, int V, int ignore_index,
              int logits_dtype, int dtype, void* stream) {
    me_clear_error();
    if (!logits || !target || !row_lse || !dlogits || !n_valid) return ME_ERR_NULL;
    if (logits_dtype != ME_F32 && logits_dtype != ME_BF16 && logits_dtype != ME_F16) return ME_ERR_BAD_DTYPE;
    if (logits_dtype != ME_F32 && logits_dtype != dtype) return ME_ERR_BAD_DTYPE;      // 16-bit logits: the tier's own type
    if (rows <= 0) return ME_OK;
    if (V <= 0 || ld < V || ld_d < V) return ME_ERR_BAD_SHAPE;
    if (dbias) {
        // fused bias gradient: 16-bit logits and dlogits, 16-byte rows, at most 2048 columns (four chunks per lane)
        const bool ok = logits_dtype != ME_F32 && (ld & 7) == 0 && (ld_d & 7) == 0 && ld >= ld_d && ld_d <= 2048 &&
                        ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
        if (!ok) return ME_ERR_BAD_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int cap = dbias ? 512 : 2048;                     // dbias: one atomic per column and block
    if (logits_dtype == ME_F32) {
        ME_DISPATCH(dtype, (ce_bwd_kernel<T, float, false><<<row_grid(rows, cap), 256, 0, st>>>((const float*)logits, ld, target, row_lse,
                                                                                              (T*)dlogits, ld_d, n_valid, extra_scale, loss_scale_dev, rows, V, ignore_index, nullptr)));
    } else if (dbias) {
        ME_DISPATCH(dtype, (ce_bwd_kernel<T, T, true><<<row_grid(rows, cap), 256, 0, st>>>((const T*)logits, ld, target, row_lse,
                                                                                         (T*)dlogits, ld_d, n_valid, extra_scale, loss_scale_dev, rows, V, ignore_index, dbias)));
    } else {
        ME_DISPATCH(dtype, (ce_bwd_kernel<T, T, false><<<row_grid(rows, cap), 256, 0, st>>>((const T*)logits, ld, target, row_lse,
                                                                                          (T*)dlogits, ld_d, n_valid, extra_scale, loss_scale_dev, rows, V, ignore_index, nullptr)));
    }
    return me_launch_status();
}

int me_sumsq(const float* g, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
    me_clear_error();
    if (!g || !out) return ME_ERR_NULL;
    if (n <= 0) return ME_OK;
    if (!aligned16(g)) return ME_ERR_ALIGNMENT;
    if (ws && (ws_bytes < ME_SUMSQ_WS_BYTES || (reinterpret_cast<uintptr_t>(ws) & 3))) return ME_ERR_WORKSPACE;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    // workspace layout: [0] ticket counter (zero between calls), [1 .. 256] block sums
    unsigned int* ticket = reinterpret_cast<unsigned int*>(ws);
    float* part = ws ? reinterpret_cast<float*>(ws) + 1 : nullptr;
    sumsq_kernel<<<(unsigned)blocks, 1024, 0, (hipStream_t)stream>>>(g, n, out, part, ticket);
    return me_launch_status();
}

int me_scaler_step(float* state, const float* sumsq, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    me_clear_error();
    if (!state || !sumsq) return ME_ERR_NULL;
    if (!(growth_factor > 1.f) || !(backoff_factor > 0.f && backoff_factor < 1.f) || growth_interval <= 0) return ME_ERR_BAD_SHAPE;
    scaler_step_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, sumsq, growth_factor, backoff_factor, growth_interval);
    return me_launch_status();
}

int me_adamw_step(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float clip, float grad_scale,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                  int zero_grad, const float* scaler_state, void* stream) {
    me_clear_error();
    if (!p || !g || !m || !v) return ME_ERR_NULL;
    if (n <= 0) return ME_OK;
    if (clip > 0.f && !sumsq) return ME_ERR_NULL;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return ME_ERR_ALIGNMENT;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (scaler_state && !sumsq) return ME_ERR_NULL;
    adamw_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, sumsq, clip, grad_scale, lr, beta1, beta2, eps,
                                                                   weight_decay, bias_corr1, bias_corr2, zero_grad, scaler_state);
    return me_launch_status();
}

int me_sample_topk_topp(const float* logits, int ld, int V, const int32_t* special, int n_special, const float* temp,
                        int top_k, float top_p, const float* u, int64_t* out_ids, int32_t* n_choices, float* dbg_p,
                        int32_t* dbg_i, int B, void* stream) {
    me_clear_error();
    if (!logits || !temp || !u || !out_ids) return ME_ERR_NULL;
    if (B <= 0 || V <= 0 || V > 4096 || ld < V || (dbg_p && !dbg_i)) return ME_ERR_BAD_SHAPE;
    SampleStep none = {};
#define ME_SAMPLE(NPV) sample_kernel<NPV><<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, V, special, special ? n_special : 0, temp, top_k, top_p, u, \
                                                      out_ids, n_choices, dbg_p, dbg_i, none)
    if (V <= 1024) ME_SAMPLE(1024); else if (V <= 2048) ME_SAMPLE(2048); else ME_SAMPLE(4096);
#undef ME_SAMPLE
    return me_launch_status();
}

int me_sample_step(const float* logits, int ld, int V, const int32_t* special, int n_special, const int64_t* prev_tok,
                   const uint8_t* is_timeshift, float* repeat_counts, float temp_note, float temp_rest, float penalty_coeff,
                   int top_k, float top_p, const float* u_table, int u_ld, const int32_t* pos, int pos0, int64_t* out_ids,
                   int32_t* n_choices, int B, void* stream) {
    me_clear_error();
    if (!logits || !prev_tok || !is_timeshift || !repeat_counts || !u_table || !pos || !out_ids) return ME_ERR_NULL;
    if (B <= 0 || V <= 0 || V > 4096 || ld < V || u_ld < B) return ME_ERR_BAD_SHAPE;
    SampleStep st = {prev_tok, is_timeshift, repeat_counts, pos, pos0, u_ld, temp_note, temp_rest, penalty_coeff};
#define ME_SAMPLE(NPV) sample_kernel<NPV><<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, V, special, special ? n_special : 0, nullptr, top_k, top_p, \
                                                      u_table, out_ids, n_choices, nullptr, nullptr, st)
    if (V <= 1024) ME_SAMPLE(1024); else if (V <= 2048) ME_SAMPLE(2048); else ME_SAMPLE(4096);
#undef ME_SAMPLE
    return me_launch_status();
}

int me_decode_commit(const int64_t* tok, int64_t* history, int ld_hist, int32_t* pos, int B, void* stream) {
    me_clear_error();
    if (!tok || !history || !pos) return ME_ERR_NULL;
    if (B <= 0 || B > 1024 || ld_hist <= 0) return ME_ERR_BAD_SHAPE;
    decode_commit_kernel<<<1, ((B + 63) / 64) * 64, 0, (hipStream_t)stream>>>(tok, history, ld_hist, pos, B);
    return me_launch_status();
}

int me_greedy_pick_commit(const float* logits, int ld, int V, const int32_t* special, int n_special, int64_t* out_ids,
                          int64_t* history, int ld_hist, int32_t* pos, int B, void* stream) {
    me_clear_error();
    if (!logits || !out_ids || !history || !pos || (n_special > 0 && !special)) return ME_ERR_NULL;
    if (B <= 0) return ME_OK;
    greedy_pick_commit_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, ld, V, special, n_special, out_ids, history, ld_hist, pos, B);
    return me_launch_status();
}

int me_greedy_pick(const float* logits, int ld, int V, const int32_t* special, int n_special, int64_t* out_ids, int B,
                   void* stream) {
    me_clear_error();
    if (!logits || !out_ids || (n_special > 0 && !special)) return ME_ERR_NULL;
    if (B <= 0) return ME_OK;
    greedy_pick_kernel<<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, V, special, n_special, out_ids);
    return me_launch_status();
}

}  // extern "C"
