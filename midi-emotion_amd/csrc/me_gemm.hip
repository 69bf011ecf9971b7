// GEMM kernels of the midiemo hot path (gfx950).
//   me_gemm_nt      C[M,N]  = A[M,K] . B[N,K]^T (+bias, relu, +add, relu-gate)
//   me_gemm_tn_acc  dW[N,K] += A[T,N]^T . B[T,K]   (split over T; partial tiles through a CALLER-owned workspace)
//   me_cast_transpose, me_cast_transpose_multi
// bf16 shapes of the train step run the persistent 256x256 kernels (gemm_nt256_kernel, gemm_tn256_kernel: 8 waves,
// 64-deep slabs, register-staged operand feed, one raw barrier per slab); everything else (f32 tier, ragged shapes)
// the generic 128x128 kernels (4 waves of 64x64, register-staged double buffer, padded LDS rows).
// The library never allocates, frees or synchronises: workspaces come from the caller (me_workspace_bytes).
#include <stdlib.h>

#include "me_common.h"
#include <type_traits>


namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;


ME_DEV void acc_zero_quad(f32x16_t& a, int g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[4 * g + e] = 0.f;
}

// Epilogue of one TRANSPOSED 32x32 accumulator block (mfma(B, A)): lane = row of C, register
// quad = 4 consecutive columns -> 8/16-byte row-contiguous stores and add/gate loads.
template <typename T, bool OUT_F32>
ME_DEV void epi_block_swapped(const f32x16_t& acc, int row_base, int col_base, int lane, void* Cv, int ldc,
                              const float* bias, const T* add, int ldadd, const T* gate, int ldgate, int M, int N,
                              bool relu) {
    const int row = row_base + (lane & 31);
    if (row >= M) return;
    const bool vec_c = (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cv) & 15) == 0;
    const bool vec_add = add && (ldadd & 3) == 0 && (reinterpret_cast<uintptr_t>(add) & 15) == 0;
    const bool vec_gate = gate && (ldgate & 3) == 0 && (reinterpret_cast<uintptr_t>(gate) & 15) == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = col_base + 8 * g + 4 * (lane >> 5);
        if (col >= N) continue;
        float v[4];
        const bool full = col + 3 < N;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = acc[4 * g + e] + ((bias && col + e < N) ? bias[col + e] : 0.f);
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        if (add) {
            const T* ap = add + (size_t)row * ldadd + col;
            if (full && vec_add) {
                T t4[4];
                if constexpr (sizeof(T) == 2) *reinterpret_cast<uint64_t*>(t4) = *reinterpret_cast<const uint64_t*>(ap);
                else *reinterpret_cast<f32x4_t*>(t4) = *reinterpret_cast<const f32x4_t*>(ap);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += ET<T>::to_f(t4[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < N) v[e] += ET<T>::to_f(ap[e]);
            }
        }
        if (gate) {
            const T* gp = gate + (size_t)row * ldgate + col;
            if (full && vec_gate) {
                T t4[4];
                if constexpr (sizeof(T) == 2) *reinterpret_cast<uint64_t*>(t4) = *reinterpret_cast<const uint64_t*>(gp);
                else *reinterpret_cast<f32x4_t*>(t4) = *reinterpret_cast<const f32x4_t*>(gp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ET<T>::to_f(t4[e]) > 0.f ? v[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < N) v[e] = ET<T>::to_f(gp[e]) > 0.f ? v[e] : 0.f;
            }
        }
        if (OUT_F32) {
            float* cp = reinterpret_cast<float*>(Cv) + (size_t)row * ldc + col;
            if (full && vec_c) *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < N) cp[e] = v[e];
            }
        } else {
            T* cp = reinterpret_cast<T*>(Cv) + (size_t)row * ldc + col;
            if (full && vec_c) st4<T>(cp, v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < N) cp[e] = ET<T>::from_f(v[e]);
            }
        }
    }
}

// ReLU sign mask (ME_WS_RELU_MASK): 1 bit per element, stored as the write-out sees the elements.  A REGION = 128 rows x 64
// columns = 1 KB = [64 lanes][16 bytes]; bit e of byte b of lane l is element (row 32 (b / 4) + 8 (b % 4) + l / 8, column
// 8 (l % 8) + e) of the region: the chunk lane l stores with instruction (i, it) = (b / 4, b % 4) of the region's write-out.
// Regions are ordered [column group][row group] with the row count rounded up to 256: one 16-byte load / store per lane and
// wave tile, nothing to gather.
ME_DEV char* relu_mask_region(const void* mask, int M, int row0, int col0) {
    const size_t rb_total = (size_t)((M + 255) >> 8) * 32;
    return reinterpret_cast<char*>(const_cast<void*>(mask)) + ((size_t)(col0 >> 6) * rb_total + (size_t)(row0 >> 3)) * 64;
}

// ---- tile write-out of the 256 x 256 kernels (shared by the main-loop variants): stage 32 rows x 128 B at a time through
// the wave's private 4 KB so that every store instruction writes 8 full 128-byte row segments (per-lane 8-byte pieces
// across 32 rows are L2-transaction bound).  bf16: a pass = 32 rows x 64 columns; f32: 32 rows x 32 columns.
// Ends with the accumulators zeroed for the next tile.
template <typename T, bool OUT_F32, int WR, int WC, int EPI>
ME_DEV void nt256_write_tile(f32x16_t (&acc)[256 / WR / 32][256 / WC / 32], char* smem, int m0, int n0, int wid, int lane,
                             void* __restrict__ Cv, int ldc, const float* __restrict__ bias, const T* __restrict__ add,
                             int ldadd, const T* __restrict__ gate, int ldgate, int M, int N, bool relu, bool vec_c,
                             chunk16* mreg = nullptr) {
    typedef typename V16<T>::x4 Tx4;
    constexpr int NW = WR * WC;
    constexpr int TM = 256 / WR, TN = 256 / WC;
    constexpr int AI = TM / 32, BJ = TN / 32;
    static_assert(EPI < 4 || (TM == 128 && !OUT_F32), "the sign-mask region of a wave is 128 rows x 64 columns = 1 KB");
    const int wr = wid / WC, wc = wid % WC;
    const int h = lane >> 5;
    char* stg = smem + 2 * 65536 + wid * (32768 / NW);
    constexpr int NJ = OUT_F32 ? 1 : 2;                                // accumulator blocks per pass
    constexpr int JP = BJ / NJ;                                        // passes per 32-row group
    constexpr int NP = AI * JP;                                        // passes
    const int lr = lane & 31;
    // one quad (4 consecutive columns of the lane's row) -> staging; 16-byte slot index XOR (row & 7) spreads
    // the 32 rows of a store over the banks
    auto stage = [&](int jj, int g, const float* v) __attribute__((always_inline)) {
        if constexpr (OUT_F32) {
            const int slot = (2 * g + h) ^ (lr & 7);
            *reinterpret_cast<f32x4_t*>(stg + lr * 128 + (slot << 4)) = (f32x4_t){v[0], v[1], v[2], v[3]};
        } else {
            const int slot = (jj * 4 + g) ^ (lr & 7);
            st4<T>(reinterpret_cast<T*>(stg + lr * 128 + (slot << 4) + h * 8), v[0], v[1], v[2], v[3]);
        }
    };
    // read back: lane -> (row it*8 + lane/8, 16-byte chunk lane%8): full-row coalesced stores
    auto write_out = [&](int ps) __attribute__((always_inline)) {
        const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = it * 8 + (lane >> 3), ch = lane & 7;
            const chunk16 v = ld_chunk(stg + rr * 128 + ((ch ^ (rr & 7)) << 4));
            const int orow = m0 + wr * TM + i * 32 + rr;
            constexpr int EPC = OUT_F32 ? 4 : 8;                       // elements per chunk
            const int col = n0 + wc * TN + j0 * 32 + ch * EPC;
            if constexpr (EPI == 4) {
                // ReLU sign mask (layout: relu_mask_region): the 8 elements of this chunk are bit e of byte b = 4 i + it of the
                // lane's 16 mask bytes.  Pure per-lane arithmetic on the rounded 16-bit halves (bf16 and f16 alike: sign-magnitude): x > 0 <=> the half, as a signed
                // 16-bit integer, is > 0 (the ReLU left no NaN) -> packed clamp to {0, 1}, the two bits of a dword side by side
                typedef short i16x2_t __attribute__((ext_vector_type(2)));
                uint32_t t = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t dw = v.v[w];                          // (a bit_cast of the vector ELEMENT reads element 0: hipcc 7.2)
                    i16x2_t x = __builtin_bit_cast(i16x2_t, dw);
                    x = __builtin_elementwise_min(__builtin_elementwise_max(x, (i16x2_t){0, 0}), (i16x2_t){1, 1});
                    t |= __builtin_bit_cast(uint32_t, x) << (2 * w);     // even elements at bits 0 2 4 6, odd ones at 16 18 20 22
                }
                const uint32_t byte = (t | (t >> 15)) & 0xffu;
                const int b = i * 4 + it, grp = ps % JP;
                mreg[grp].v[b >> 2] |= byte << (8 * (b & 3));
            }
            if (orow < M && col < N) {
                if (col + EPC <= N && vec_c) {
                    if constexpr (OUT_F32) st_chunk(reinterpret_cast<float*>(Cv) + (size_t)orow * ldc + col, v);
                    else st_chunk(reinterpret_cast<T*>(Cv) + (size_t)orow * ldc + col, v);
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e)
                        if (col + e < N) {
                            if constexpr (OUT_F32) (reinterpret_cast<float*>(Cv))[(size_t)orow * ldc + col + e] = reinterpret_cast<const float*>(&v)[e];
                            else (reinterpret_cast<T*>(Cv))[(size_t)orow * ldc + col + e] = reinterpret_cast<const T*>(&v)[e];
                        }
                }
            }
        }
    };
    // vmcnt is in-order: a global load issued between two passes waits for the previous pass's STORES to retire
    // (measured: 20 us of an 80 us launch).  Hence two code paths (registers = max, not sum):
    if constexpr (EPI == 0 || EPI == 4) {
        // (a) bias (+ReLU; EPI 4: + the ReLU's sign mask, written by write_out): the wave's bias values are fetched once, before any store
        f32x4_t bv[BJ][4];
        const bool vec_bias = (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wc * TN + j * 32 + 8 * g + 4 * h;
                bv[j][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (bias) {
                    if (vec_bias && col + 3 < N) bv[j][g] = *reinterpret_cast<const f32x4_t*>(bias + col);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (col + e < N) bv[j][g][e] = bias[col + e];
                    }
                }
            }
        if constexpr (EPI == 4) {
#pragma unroll
            for (int g = 0; g < BJ / 2; ++g) mreg[g] = zero_chunk();
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j0 + jj][4 * g + e] + bv[j0 + jj][g][e];
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    stage(jj, g, v);
                }
            write_out(ps);
        }
        if constexpr (EPI == 4) {
#pragma unroll
            for (int g = 0; g < BJ / 2; ++g) {
                char* mp = relu_mask_region(gate, M, m0 + wr * TM, n0 + wc * TN + g * 64);
                if (m0 + wr * TM < M && n0 + wc * TN + g * 64 < N) st_chunk(mp + lane * 16, mreg[g]);      // one 1 KB store per wave
            }
        }
    } else if constexpr (EPI == 1) {
        // (g) ReLU gate alone (the FFN_suf dgrad, N = d_inner): a select commutes with the rounding, so the gate is
        // applied to the ROUNDED tile in its row-contiguous staged form -- the gate operand is then read exactly like the
        // output is written (16 bytes per lane, eight lanes per 128-byte row segment) instead of as 8-byte pieces of 32
        // different rows per instruction (same box, interleaved, N2048.K512: 130.0 -> 96.0 us; the product without a gate 76-78 us;
        // the remaining 20 us are the 134 MB of gate rows at the HBM rate).
        // Chunks of pass ps + 1 are requested before the stores of pass ps.
        chunk16 gq[2][4];
        auto fetch_gate = [&](int ps, chunk16 (&q)[4]) __attribute__((always_inline)) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = it * 8 + (lane >> 3), ch = lane & 7;
                const int orow = min(m0 + wr * TM + i * 32 + rr, M - 1);
                const int col = min(n0 + wc * TN + j0 * 32 + ch * 8, N - 8);
                q[it] = ld_chunk(gate + (size_t)orow * ldgate + col);
            }
        };
        fetch_gate(0, gq[0]);
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j0 + jj][4 * g + e];
                    stage(jj, g, v);
                }
            if (ps + 1 < NP) fetch_gate(ps + 1, gq[(ps + 1) & 1]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = it * 8 + (lane >> 3), ch = lane & 7;
                chunk16 v = ld_chunk(stg + rr * 128 + ((ch ^ (rr & 7)) << 4));
                const T* gp = reinterpret_cast<const T*>(&gq[ps & 1][it]);
                T* vp = reinterpret_cast<T*>(&v);
#pragma unroll
                for (int e = 0; e < 8; ++e) vp[e] = (float)gp[e] > 0.f ? vp[e] : (T)0.f;
                const int orow = m0 + wr * TM + i * 32 + rr;
                const int col = n0 + wc * TN + j0 * 32 + ch * 8;
                if (orow < M && col < N) st_chunk(reinterpret_cast<T*>(Cv) + (size_t)orow * ldc + col, v);
            }
        }
    } else if constexpr (EPI == 5) {
        // (m) ReLU gate from the sign mask the forward's EPI 4 launch left: the wave's whole region (1 KB per 128 rows x 64
        // columns instead of 16 KB of activations, the lane's own 16 bytes) arrived in mreg with ONE load issued a tile ahead
        // (gemm_nt256_kernel); bit e of byte 4 i + it keeps element e of the lane's chunk.  No memory instruction besides
        // the stores.
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int i = ps / JP, j0 = (ps % JP) * NJ, grp = ps % JP;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j0 + jj][4 * g + e];
                    stage(jj, g, v);
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = it * 8 + (lane >> 3), ch = lane & 7;
                chunk16 v = ld_chunk(stg + rr * 128 + ((ch ^ (rr & 7)) << 4));
                const int b = i * 4 + it;
                const uint32_t bits = mreg[grp].v[b >> 2];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8 * (b & 3) + 2 * w, 1);          // 0 / 0xffffffff
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8 * (b & 3) + 2 * w + 1, 1);
                    v.v[w] &= (hi & 0xffff0000u) | (lo & 0x0000ffffu);
                }
                const int orow = m0 + wr * TM + i * 32 + rr;
                const int col = n0 + wc * TN + j0 * 32 + ch * 8;
                if (orow < M && col < N) st_chunk(reinterpret_cast<T*>(Cv) + (size_t)orow * ldc + col, v);
            }
        }
    } else if constexpr (EPI == 2) {
        // (r) residual add alone (the FFN_pre and qkv dgrads): the operand is read row-contiguously like the output is
        // written (16 bytes per lane) and brought into the accumulators' layout through the wave's staging buffer --
        // chunks in, 8-byte quads out, the mapping `stage` / `write_out` use in the other direction -- so the sum is still
        // formed in f32 before the one rounding (bit-identical to the element-wise path b), but an instruction touches 8
        // full 128-byte row segments instead of 16 bytes in each of 32 rows (same box, interleaved: +9.5 us -> see profiles).
        chunk16 aq[2][4];
        auto fetch_add = [&](int ps, chunk16 (&q)[4]) __attribute__((always_inline)) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = it * 8 + (lane >> 3), ch = lane & 7;
                const int orow = min(m0 + wr * TM + i * 32 + rr, M - 1);
                const int col = min(n0 + wc * TN + j0 * 32 + ch * 8, N - 8);
                q[it] = ld_chunk(add + (size_t)orow * ldadd + col);
            }
        };
        fetch_add(0, aq[0]);
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = it * 8 + (lane >> 3), ch = lane & 7;
                st_chunk(stg + rr * 128 + ((ch ^ (rr & 7)) << 4), aq[ps & 1][it]);
            }
            Tx4 av[NJ][4];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    av[jj][g] = *reinterpret_cast<const Tx4*>(stg + lr * 128 + (((jj * 4 + g) ^ (lr & 7)) << 4) + h * 8);
            if (ps + 1 < NP) fetch_add(ps + 1, aq[(ps + 1) & 1]);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wc * TN + (j0 + jj) * 32 + 8 * g + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j0 + jj][4 * g + e] + ((bias && col + e < N) ? bias[col + e] : 0.f);
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                        v[e] += (float)av[jj][g][e];
                    }
                    stage(jj, g, v);
                }
            write_out(ps);
        }
    } else {
        // (b) residual add / ReLU gate (the backward GEMMs): their operands for the NEXT pass are fetched (8 bytes
        // per quad) before this pass's stores are issued
        const bool vec_add = add && (ldadd & 3) == 0 && (reinterpret_cast<uintptr_t>(add) & 7) == 0;
        const bool vec_gate = gate && (ldgate & 3) == 0 && (reinterpret_cast<uintptr_t>(gate) & 7) == 0;
        auto load4 = [&](const T* base, int ld, bool vec, int row, int col) __attribute__((always_inline)) -> Tx4 {
            Tx4 r = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
            if (base && row < M) {
                if (vec && col + 3 < N) r = *reinterpret_cast<const Tx4*>(base + (size_t)row * ld + col);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) r[e] = base[(size_t)row * ld + col + e];
                }
            }
            return r;
        };
        Tx4 av[NJ][4], gv[NJ][4];
        auto fetch_ag = [&](int ps) __attribute__((always_inline)) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
            const int row = m0 + wr * TM + i * 32 + lr;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wc * TN + (j0 + jj) * 32 + 8 * g + 4 * h;
                    if (add) av[jj][g] = load4(add, ldadd, vec_add, row, col);
                    if (gate) gv[jj][g] = load4(gate, ldgate, vec_gate, row, col);
                }
        };
        fetch_ag(0);
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int i = ps / JP, j0 = (ps % JP) * NJ;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wc * TN + (j0 + jj) * 32 + 8 * g + 4 * h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j0 + jj][4 * g + e] + ((bias && col + e < N) ? bias[col + e] : 0.f);
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                        if (add) v[e] += (float)av[jj][g][e];
                        if (gate) v[e] = (float)gv[jj][g][e] > 0.f ? v[e] : 0.f;
                    }
                    stage(jj, g, v);
                }
            if (ps + 1 < NP) fetch_ag(ps + 1);
            write_out(ps);
        }
    }
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j) acc_zero(acc[i][j]);
}

// ---------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 macro-atoms), bf16, K % 64 == 0.
// Each slab [256 rows][64 k] of A and of B lives in LDS as a LINEAR image (128 B per row); a "piece" is
// what one wave instruction moves: 8 rows x 128 B, one 16-byte chunk per lane.  Bank conflicts of the
// ds_read_b128 fragment reads are avoided by permuting the SOURCE chunk of every lane (slot c of row r
// holds chunk c ^ swz(r)) and applying the same XOR on the read.
//
// Schedule (what the measurements on MI355X say, tools/ubench_ingest.hip + tools/ubench_mfma.hip):
//   * with random operands the matrix pipe is power limited to ~1750 TF/s (1500 with the 6 ds_read_b128 per
//     8 MFMAs this tile shape needs); one CU can ingest 65-70 GB/s of this operand mix; both at once in a plain
//     8-wave loop with ONE barrier per slab reach 53 us for the K=2048 shape (vendor GEMM: 51 us);
//   * every additional s_barrier per slab costs ~150 cycles x 8 waves (a ping-pong schedule with a barrier per
//     k-phase lost 25 %), and direct-to-LDS pieces (global_load_lds) cost the issuing wave 60-180 cycles each
//     and must land within one slab period;
//   so: operands are fetched into REGISTERS a whole slab ahead (8 x global_load_dwordx4 per wave always in
//   flight), stored into the other LDS buffer while the current slab is multiplied (ds_write_b128, waits counted:
//   the loop body is straight-line code so the compiler's vmcnt bookkeeping stays exact), one raw s_barrier per
//   slab (no vmcnt(0)), the two waves of a SIMD interleave MFMA bursts and LDS traffic by themselves.
// The epilogue has its own 32 KB staging area (4 KB per wave), so the slab buffers keep streaming.
// Rows beyond M / N are clamped on the load side (their results are never stored).
// ---------------------------------------------------------------------------------------------
constexpr int NT256_LDS = 2 * 65536 + 8 * 4096;

ME_DEV void slot_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// WR x WC = wave grid over the 256 x 256 tile: 2 x 4 (8 waves, 128 x 64 per wave, two waves per SIMD) or
// 2 x 2 (4 waves, 128 x 128 per wave, one wave per SIMD with the whole 512-entry register file: 8 instead of
// 12 fragment reads per 16 MFMAs).
// EPI selects the ONE write-out path an instantiation contains (the host picks it from the operands and their alignment):
//   0 = plain / bias / ReLU (a);  1 = ReLU gate applied to the staged rows (g);  2 = residual add through the staging buffer (r);
//   3 = the general element-wise path (b: any combination, any alignment).
// One path per instantiation because the kernel sits at the 256-register limit: with all four in one body hipcc spilled an
// operand piece per slab inside the main loop (round 4).
template <typename T, bool OUT_F32, int WR, int WC, int EPI>
__global__ __launch_bounds__(WR * WC * 64) void gemm_nt256_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc,
    const float* __restrict__ bias, const T* __restrict__ add, int ldadd, const T* __restrict__ gate,
    int ldgate, int M, int N, int K, int flags) {
    constexpr int NW = WR * WC;
    constexpr int TM = 256 / WR, TN = 256 / WC;                           // wave tile
    constexpr int AI = TM / 32, BJ = TN / 32;                             // macro-atoms per wave tile
    constexpr int PPH = 32 / NW;                                          // pieces per operand, wave and slab
    constexpr int PP = 2 * PPH;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][A 32 KB | B 32 KB] | staging
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WC, wc = wid % WC;
    const int ntn = (N + 255) / 256, ntiles = ntn * ((M + 255) / 256);
    const int nk = K / 64;
    // persistent over tiles: block b handles tiles b, b+grid, ...
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * nk;
    if (nsteps <= 0) return;
    auto tile_origin = [&](int it, int& m0, int& n0) __attribute__((always_inline)) {
        int t = it * gridDim.x + blockIdx.x;
        if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // contiguous tile range per XCD
        m0 = (t / ntn) * 256;
        n0 = (t % ntn) * 256;
    };

    // ---- operand stream.  A piece = 8 rows x 128 B (one 16-byte chunk per lane).  It is fetched into
    // registers a whole slab ahead (R[i] always has a load in flight) and stored to the LINEAR slab image
    // in LDS once the buffer is free: per wave and slab PP x (global_load_dwordx4 + ds_write_b128), a few
    // issue cycles each -- a direct-to-LDS piece (global_load_lds) costs its wave 60-180 issue cycles and
    // must land within the same slab period (measured: 78-82 us vs 64 us without any feed on the K=2048 shape).
    // Loads and stores are UNCONDITIONAL (past the end the stream stays on the last slab and the store goes to
    // the free buffer): the loop body is straight-line code, so the compiler's vmcnt bookkeeping stays exact
    // (it waits for the oldest of the loads in flight, not for all of them).
    const int lrow = lane >> 3;
    int d_step = 0, d_k = 0, d_tile = 0, d_m0, d_n0;                      // slab being fetched
    tile_origin(0, d_m0, d_n0);
    chunk16 R[PP];
    // piece i of the wave (i < PPH: A rows, else B rows): 8-row block blk = PPH wid + ii, rows r0 + 8 ii,
    // chunk ch0 ^ ii (swz(r) = (lrow ^ blk) & 7 = swz(r0) ^ ii); 32-bit byte offsets against the scalar bases
    const int r0 = wid * PPH * 8 + lrow;
    const int ch0 = (lane & 7) ^ ((lrow ^ (wid * PPH)) & 7);
    const uint32_t lda2 = (uint32_t)lda * 2u, ldb2 = (uint32_t)ldb * 2u;
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);
    auto ld_piece = [&](int i) __attribute__((always_inline)) {
        const int ii = i % PPH;
        const uint32_t col = (uint32_t)((ch0 ^ ii) * 16 + d_k * 2);
        if (i < PPH) R[i] = ld_chunk(Ab + ((uint32_t)min(d_m0 + r0 + 8 * ii, M - 1) * lda2 + col));
        else R[i] = ld_chunk(Bb + ((uint32_t)min(d_n0 + r0 + 8 * ii, N - 1) * ldb2 + col));
    };
    char* const st_base = smem + wid * PPH * 1024 + lane * 16;
    auto st_piece = [&](int i, int slab) __attribute__((always_inline)) {
        st_chunk(st_base + (slab & 1) * 65536 + ((i / PPH) * 32768 + (i % PPH) * 1024), R[i]);
    };
    auto ld_advance = [&]() __attribute__((always_inline)) {
        if (d_step + 1 >= nsteps) return;                                  // stay on the last slab
        ++d_step;
        d_k += 64;
        if (d_k == K) { d_k = 0; ++d_tile; tile_origin(d_tile, d_m0, d_n0); }
    };

    f32x16_t acc[AI][BJ];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j) acc_zero(acc[i][j]);

    const int frow = lane & 31, h = lane >> 5;
    const bool relu = flags & ME_EPI_RELU;
    const bool vec_c = (ldc % (OUT_F32 ? 4 : 8)) == 0 && (reinterpret_cast<uintptr_t>(Cv) & 15) == 0;
    static_assert(!(OUT_F32 && (EPI == 1 || EPI == 2)), "the row paths stage bf16 rows");

    // swz(r) = (r ^ (r >> 3)) & 7 makes every 16-lane group of a ds_read_b128 hit 16 distinct
    // 16-byte slots of the 256-byte bank row (conflict free).
    // Address of k-phase kk = (row base | chunk slot) ^ (kk << 5): (2 kk + h) ^ swz = (2 kk) ^ (h ^ swz).
    // Rows r and r + 32 differ in swz by 4 ((r >> 3) & 7 advances by 4), i.e. their slot offsets by XOR 64.
    Frag<T> fa[AI], fb[BJ];
    uint32_t pa0, pb0;
    { const int r = wr * TM + frow; pa0 = r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    { const int r = wc * TN + frow; pb0 = 32768 + r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    auto lfrag = [&](uint32_t bufoff, int kk) __attribute__((always_inline)) {
        const char* ae = smem + ((pa0 + bufoff) ^ (uint32_t)(kk << 5));
        const char* ao = smem + ((pa0 + bufoff) ^ (uint32_t)((kk << 5) ^ 64));
        const char* be = smem + ((pb0 + bufoff) ^ (uint32_t)(kk << 5));
        const char* bo = smem + ((pb0 + bufoff) ^ (uint32_t)((kk << 5) ^ 64));
#pragma unroll
        for (int i = 0; i < AI; ++i) frag_load(fa[i], reinterpret_cast<const T*>(((i & 1) ? ao : ae) + i * 4096));
#pragma unroll
        for (int j = 0; j < BJ; ++j) frag_load(fb[j], reinterpret_cast<const T*>(((j & 1) ? bo : be) + j * 4096));
    };

    // sign-mask write-outs (EPI 4 / 5): the wave's 1 KB mask region(s) travel in registers; EPI 5 requests the region of the NEXT
    // write-out a whole tile ahead (one 16-byte load per lane in the queue of the operand stream: no wait of its own)
    chunk16 mreg[BJ / 2 > 0 ? BJ / 2 : 1];
    auto mask_fetch = [&](int tile_it) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(tile_it, m0, n0);
#pragma unroll
        for (int g = 0; g < BJ / 2; ++g)
            mreg[g] = ld_chunk(relu_mask_region(gate, M, m0 + wr * TM, min(n0 + wc * TN + g * 64, N - 64)) + lane * 16);
    };
    auto epilogue = [&](int tile_it) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(tile_it, m0, n0);
        nt256_write_tile<T, OUT_F32, WR, WC, EPI>(acc, smem, m0, n0, wid, lane, Cv, ldc, bias, add, ldadd, gate, ldgate, M, N, relu, vec_c,
                                               mreg);
        if constexpr (EPI == 5) mask_fetch(min(tile_it + 1, my_tiles - 1));
    };
    if constexpr (EPI == 5) mask_fetch(0);

    // ---- prologue: slab 0 into LDS, slab 1 into the registers
#pragma unroll
    for (int i = 0; i < PP; ++i) ld_piece(i);
    ld_advance();
#pragma unroll
    for (int i = 0; i < PP; ++i) st_piece(i, 0);
#pragma unroll
    for (int i = 0; i < PP; ++i) ld_piece(i);
    ld_advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot_barrier();

    for (int step = 0; step < nsteps; ++step) {
        const uint32_t buf = (uint32_t)(step & 1) * 65536u;
        // slab step+1 (in the registers since the previous iteration) -> the other buffer; refetch slab step+2.
        // PP / 4 pieces per k-phase, scheduled by the compiler between the MFMAs.
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            lfrag(buf, kk);
#pragma unroll
            for (int q = 0; q < PP / 4; ++q) st_piece((PP / 4) * kk + q, step + 1);
#pragma unroll
            for (int q = 0; q < PP / 4; ++q) ld_piece((PP / 4) * kk + q);
#pragma unroll
            for (int g = 0; g < AI; ++g)
#pragma unroll
                for (int j = 0; j < BJ; ++j) mma32(acc[g][j], fb[j], fa[g]);
        }
        ld_advance();
        if ((step + 1) % nk == 0) epilogue(step / nk);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // my stores of slab step+1 are in LDS
        slot_barrier();                                                    // everybody's; slab `step` fully consumed
    }
}

// C = A.B^T.  128x128 block tile, 4 waves (2x2) x 64x64.  K-slab BKT (64 bf16 / 32 f32) is
// double buffered in LDS: one barrier per slab, the slab after next is in flight in registers
// while the current one is multiplied.  The 32x32 blocks are accumulated TRANSPOSED (mfma(B, A)):
// a lane then owns one row of C and 4 consecutive columns per register quad, so the epilogue is
// 8/16-byte row-contiguous stores (and loads of add/gate) instead of 2-byte ones.
// Blocks are renumbered so that each XCD (private L2) works on a contiguous range of tiles.
template <typename T> struct GemmK { static constexpr int BKT = sizeof(T) == 2 ? 64 : 32; };

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, void* __restrict__ Cv, int ldc,
    const float* __restrict__ bias, const T* __restrict__ add, int ldadd, const T* __restrict__ gate, int ldgate,
    int M, int N, int K, int flags) {
    constexpr int CH = ET<T>::CH;
    constexpr int BKT = GemmK<T>::BKT;
    constexpr int CPR = BKT / CH;          // chunks per tile row
    constexpr int LDK = BKT + CH;          // padded LDS row (elements)
    constexpr int NCH = BM * CPR / NTHREADS;
    __shared__ __attribute__((aligned(16))) T As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) T Bs[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int ntn = (N + BN - 1) / BN, ntiles = ntn * ((M + BM - 1) / BM);
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    chunk16 ra[NCH], rb[NCH];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, row = c / CPR, k = k0 + (c % CPR) * CH;
            ra[i] = (m0 + row < M && k < K) ? ld_chunk(A + (size_t)(m0 + row) * lda + k) : zero_chunk();
            rb[i] = (n0 + row < N && k < K) ? ld_chunk(B + (size_t)(n0 + row) * ldb + k) : zero_chunk();
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, row = c / CPR, cc = c % CPR;
            st_chunk(&As[buf][row * LDK + cc * CH], ra[i]);
            st_chunk(&Bs[buf][row * LDK + cc * CH], rb[i]);
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);

    const int nk = (K + BKT - 1) / BKT;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    gload(0);
    sstore(0);
    if (nk > 1) gload(BKT);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int kk = 0; kk < BKT / 16; ++kk) {
            Frag<T> fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                frag_load(fa[i], &As[buf][(wr * 64 + i * 32 + frow) * LDK + kk * 16 + fk]);
                frag_load(fb[i], &Bs[buf][(wc * 64 + i * 32 + frow) * LDK + kk * 16 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    mma32(acc[i][j], fb[j], fa[i]);          // transposed blocks: lane = row of C
                }
        }
        if (kt + 1 < nk) {
            sstore(buf ^ 1);                       // buf^1 was last read in slab kt-1 (barrier since)
            if (kt + 2 < nk) gload((kt + 2) * BKT);
        }
        __syncthreads();
    }

    const bool relu = flags & ME_EPI_RELU;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            epi_block_swapped<T, OUT_F32>(acc[i][j], m0 + wr * 64 + i * 32, n0 + wc * 64 + j * 32, lane, Cv, ldc, bias, add, ldadd,
                                          gate, ldgate, M, N, relu);
}

// dW[n][k] += sum_t A[t][n] * B[t][k].  Tiles are staged TRANSPOSED into LDS
// ([n][t] / [k][t], contraction contiguous) so the fragment reads are the same
// 16-byte reads as in the NT kernel.  Lanes walk t (consecutive LDS addresses)
// during the transposing scatter, so the element writes are conflict free.
constexpr int BT = 32;

template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, float* __restrict__ dW, int lddw,
    float* __restrict__ dbias, int Tn, int N, int K, int t_per_block) {
    constexpr int CH = ET<T>::CH;
    constexpr int LDT = BT + CH;
    constexpr int CPC = 128 / CH;                 // chunks per 128-wide tile row
    constexpr int NCH = BT * CPC / NTHREADS;      // chunks per thread per operand
    __shared__ __attribute__((aligned(16))) T As[128 * LDT];
    __shared__ __attribute__((aligned(16))) T Bs[128 * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int n0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int t_begin = blockIdx.z * t_per_block;
    const int t_end = min(Tn, t_begin + t_per_block);
    if (t_begin >= t_end) return;

    chunk16 ra[NCH], rb[NCH];
    auto gload = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, t = c % BT, cc = (c / BT) * CH;
            const bool tv = t0 + t < t_end;
            ra[i] = (tv && n0 + cc < N) ? ld_chunk(A + (size_t)(t0 + t) * lda + n0 + cc) : zero_chunk();
            rb[i] = (tv && k0 + cc < K) ? ld_chunk(B + (size_t)(t0 + t) * ldb + k0 + cc) : zero_chunk();
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, t = c % BT, cc = (c / BT) * CH;
            const T* ea = reinterpret_cast<const T*>(&ra[i]);
            const T* eb = reinterpret_cast<const T*>(&rb[i]);
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                As[(cc + e) * LDT + t] = ea[e];
                Bs[(cc + e) * LDT + t] = eb[e];
            }
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);
    float bsum = 0.f;
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;
    const int frow = lane & 31, fk = (lane >> 5) * 8;

    gload(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += BT) {
        sstore();
        __syncthreads();
        if (t0 + BT < t_end) gload(t0 + BT);
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int t = 0; t < BT; ++t) bsum += ET<T>::to_f(As[tid * LDT + t]);
        }
#pragma unroll
        for (int kk = 0; kk < BT / 16; ++kk) {
            Frag<T> fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                frag_load(fa[i], &As[(wr * 64 + i * 32 + frow) * LDT + kk * 16 + fk]);
                frag_load(fb[i], &Bs[(wc * 64 + i * 32 + frow) * LDT + kk * 16 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();
    }
    if (do_bias && tid < 128 && n0 + tid < N) atomicAdd(&dbias[n0 + tid], bsum);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wc * 64 + j * 32 + c_col(lane);
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wr * 64 + i * 32 + c_row(r, lane);
                if (row < N) atomicAdd(&dW[(size_t)row * lddw + col], acc[i][j][r]);
            }
        }
}

// bf16 TN kernel, v2.  Tiles keep their NATURAL layout in LDS ([t][n], 16-byte chunk writes) and the
// contraction-contiguous MFMA fragments are produced by the hardware transpose read
// ds_read_b64_tr_b16: inside each 16-lane group lane l supplies the address of 4 contiguous
// elements and lane i receives element i%4 of the chunks of lanes i/4 + 4j (measured on gfx950).
// With lane l pointing at row k0 + l/4, columns 4*(l%4).., lane i gets column i for 4 consecutive
// k: a free 4 x 16 transpose.  Row stride 160 elements (320 B) puts the 8 row segments of a
// 32-lane half on distinct banks.  Double-buffered, BT = 32 tokens per step.
// (A 256x256 direct-to-LDS variant reaches 757 TF in its main loop but loses end to end: the split-K
// flush costs blocks x tile-area f32 atomics either way, and only this 3-blocks-per-CU kernel overlaps
// it with other blocks' MFMA work.  Measured: 173 vs 123 us on dW1.)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_tn16_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb, float* __restrict__ dW, int lddw,
    float* __restrict__ dbias, int Tn, int N, int K, int t_per_block, int tn, int tk, int nsplit) {
    constexpr int LDN_ = 160;                      // LDS row stride (elements)
    constexpr int NCH = BT * 16 / NTHREADS;        // 16-byte chunks per thread per operand (2)
    __shared__ __attribute__((aligned(16))) T As[2][BT * LDN_];
    __shared__ __attribute__((aligned(16))) T Bs[2][BT * LDN_];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    // 1-D grid, XCD-aware order (block b runs on XCD b % 8, each XCD has a private L2): a "pair" =
    // (n-tile, token slab) shares one A panel; its tk k-tiles run back to back on ONE XCD so the
    // panel is fetched from HBM once and hit in L2 by the others.
    const int npairs = tn * nsplit;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / tk) * 8 + xcd, ky = slot % tk;
    if (pair >= npairs) return;
    const int nx = pair % tn, z = pair / tn;
    const int n0 = nx * 128, k0 = ky * 128;
    const int t_begin = z * t_per_block;
    const int t_end = min(Tn, t_begin + t_per_block);
    if (t_begin >= t_end) return;

    chunk16 ra[NCH], rb[NCH];
    auto gload = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, t = c >> 4, cc = (c & 15) * 8;
            const bool tv = t0 + t < t_end;
            ra[i] = (tv && n0 + cc < N) ? ld_chunk(A + (size_t)(t0 + t) * lda + n0 + cc) : zero_chunk();
            rb[i] = (tv && k0 + cc < K) ? ld_chunk(B + (size_t)(t0 + t) * ldb + k0 + cc) : zero_chunk();
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTHREADS, t = c >> 4, cc = (c & 15) * 8;
            st_chunk(&As[buf][t * LDN_ + cc], ra[i]);
            st_chunk(&Bs[buf][t * LDN_ + cc], rb[i]);
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);
    float bsum = 0.f;
    const bool do_bias = dbias != nullptr && ky == 0;
    // transpose-read lane geometry
    const int l16 = lane & 15, rbase = ((lane >> 4) & 1) * 16, h = lane >> 5;
    const int trow = l16 >> 2, tcol = rbase + 4 * (l16 & 3);

    const int nsteps = (t_end - t_begin + BT - 1) / BT;
    gload(t_begin);
    sstore(0);
    if (nsteps > 1) gload(t_begin + BT);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int t = 0; t < BT; ++t) bsum += (float)As[buf][t * LDN_ + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BT / 16; ++kk) {
            Frag<T> fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const T* pa = &As[buf][(kk * 16 + 8 * h + trow) * LDN_ + wr * 64 + i * 32 + tcol];
                const T* pb = &Bs[buf][(kk * 16 + 8 * h + trow) * LDN_ + wc * 64 + i * 32 + tcol];
                fa[i].v = __builtin_shufflevector(lds_tr4(pa), lds_tr4(pa + 4 * LDN_), 0, 1, 2, 3, 4, 5, 6, 7);
                fb[i].v = __builtin_shufflevector(lds_tr4(pb), lds_tr4(pb + 4 * LDN_), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
        }
        if (st + 1 < nsteps) {
            sstore(buf ^ 1);
            if (st + 2 < nsteps) gload(t_begin + (st + 2) * BT);
        }
        // raw barrier: __syncthreads() would also wait (vmcnt(0)) for the loads of the slab after next issued just above
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot_barrier();
    }
    if (do_bias && tid < 128 && n0 + tid < N) atomicAdd(&dbias[n0 + tid], bsum);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wc * 64 + j * 32 + c_col(lane);
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wr * 64 + i * 32 + c_row(r, lane);
                if (row < N) atomicAdd(&dW[(size_t)row * lddw + col], acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// dW[N][K] += A^T B over a token range, 256 x 256 tile, 8 waves (2 x 4, each 128 x 64), bf16.
// Same pipeline as gemm_nt256_kernel: 64-token slabs [64 t][256 cols] of dY and of X are fetched into
// registers a slab ahead (16-byte chunks, 512-byte row segments: two token rows per wave instruction),
// stored to the other LDS buffer during the current slab's MFMAs, one raw s_barrier per slab.  The
// slabs keep their natural [token][column] layout (row stride 576 B: the four token rows a
// ds_read_b64_tr_b16 group touches fall into different bank quarters); the transposed operand
// fragments come out of the transpose-read.  Compared with the 128 x 128 kernel the tile halves the
// operand bytes per FLOP through L1 (813 -> 406 MB per dWqkv launch).
// Requires K % 256 == 0 and dY rows readable up to the next multiple of 256 columns (lda >= that; columns >= N
// only produce rows >= N of the tile, which are dropped); tokens past the range end are stored as zeros.
// ---------------------------------------------------------------------------------------------
constexpr int TN256_LD = 288;                                   // elements per LDS row
constexpr int TN256_BT = 64;                                    // tokens per slab
constexpr int TN256_OP = TN256_BT * TN256_LD * 2;               // bytes per operand slab (36864)
constexpr int TN256_LDS = 4 * TN256_OP;                         // two buffers x two operands (147456)

// One launch serves up to ME_TN_MAX_GROUP products that share the token dimension (the four weight gradients of a
// layer): the work items (token range, tile) of ALL products are numbered together, so the token split is
// #CUs / (total tiles) instead of #CUs / (tiles of one product) -- at the headline shapes 5 ranges instead of 16-64, i.e.
// a quarter of the partial-tile bytes written here and read by the reduce pass, one flush and one launch instead of four.
struct tn_prod {
    const void* A; const void* B; float* dW; float* dbias;
    int lda, ldb, lddw, N, tn, tk, tile0, pad_;
};
struct tn_group { tn_prod p[ME_TN_MAX_GROUP]; int np, ntile, nsplit, t_per_block, Tn, pad_; };

// RAGGED = false: every slab of every range is whole (Tn % t_per_block == 0, t_per_block % 64 == 0): the operand feed has
// no clamp, no zero-fill select and no per-piece address multiply (73 -> ~35 VALU instructions per slab and wave).
template <typename T, bool RAGGED>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(const tn_group G, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][A slab | B slab]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int Tn = G.Tn, nsplit = G.nsplit, t_per_block = G.t_per_block;
    // XCD-aware order (block b runs on XCD b % 8, each XCD has its own L2): the work items g = (token range z, tile)
    // are numbered range-major and XCD x takes the contiguous run [x * per, (x + 1) * per), so the tn x tk tiles of
    // a token range -- which all stream the same dY / X slabs -- sit on one XCD (two at a run boundary) and share
    // the slabs through L2.  (Spreading them over the XCDs read 295 MB from HBM per launch instead of ~135; the launch time did not
    // change -- the re-reads hit the Infinity Cache -- but the fabric traffic halves.)
    const int ntile = G.ntile, total = ntile * nsplit, per = (total + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = xcd * per + slot;
    if (slot >= per || g >= total) return;
    const int z = g / ntile, gtile = g % ntile;
    int pi = 0;
    while (pi + 1 < G.np && gtile >= G.p[pi + 1].tile0) ++pi;
    const T* __restrict__ A = reinterpret_cast<const T*>(G.p[pi].A);
    const T* __restrict__ B = reinterpret_cast<const T*>(G.p[pi].B);
    float* __restrict__ dW = G.p[pi].dW;
    float* __restrict__ dbias = G.p[pi].dbias;
    const int lda = G.p[pi].lda, ldb = G.p[pi].ldb, lddw = G.p[pi].lddw, N = G.p[pi].N, tn = G.p[pi].tn;
    const int tile = gtile - G.p[pi].tile0;
    const int nx = tile % tn, ky = tile / tn;
    const int n0 = nx * 256, k0 = ky * 256;
    const int t_begin = z * t_per_block;
    const int t_end = min(Tn, t_begin + t_per_block);
    if (t_begin >= t_end) return;
    const int nsteps = (t_end - t_begin + TN256_BT - 1) / TN256_BT;

    // ---- operand stream: piece i of the wave = token rows 2 (4 wid + (i & 3)) + {0, 1}, i < 4: dY, else X
    const int prow = lane >> 5, pcol = (lane & 31) * 8;
    const uint32_t lda2 = (uint32_t)lda * 2u, ldb2 = (uint32_t)ldb * 2u;
    const char* Ab = reinterpret_cast<const char*>(A) + (size_t)(n0 + pcol) * 2;
    const char* Bb = reinterpret_cast<const char*>(B) + (size_t)(k0 + pcol) * 2;
    chunk16 R[8];
    int d_slab = 0;                                                       // slab being fetched
    // whole slabs: piece i of slab s sits at wave-uniform base (s, i) + one fixed lane offset
    const uint32_t vA = (uint32_t)prow * lda2, vB = (uint32_t)prow * ldb2;
    const char* sA = Ab + (size_t)(t_begin + wid * 8) * lda2;
    const char* sB = Bb + (size_t)(t_begin + wid * 8) * ldb2;
    auto ld_piece = [&](int i) __attribute__((always_inline)) {
        if constexpr (RAGGED) {
            const int row = (wid * 4 + (i & 3)) * 2 + prow;
            const uint32_t t = (uint32_t)min(t_begin + d_slab * TN256_BT + row, Tn - 1);
            if (i < 4) R[i] = ld_chunk(Ab + t * lda2);
            else R[i] = ld_chunk(Bb + t * ldb2);
        } else {
            if (i < 4) R[i] = ld_chunk(sA + (size_t)(d_slab * TN256_BT + (i & 3) * 2) * lda2 + vA);
            else R[i] = ld_chunk(sB + (size_t)(d_slab * TN256_BT + (i & 3) * 2) * ldb2 + vB);
        }
    };
    char* const st_base = smem + ((wid * 8 + prow) * TN256_LD + pcol) * 2;
    auto st_piece = [&](int i, int slab) __attribute__((always_inline)) {
        char* dst = st_base + (slab & 1) * (2 * TN256_OP) + ((i >> 2) * TN256_OP + (i & 3) * 2 * TN256_LD * 2);
        if constexpr (RAGGED) {
            const int row = (wid * 4 + (i & 3)) * 2 + prow;
            const bool valid = t_begin + slab * TN256_BT + row < t_end;   // false as well for the slab past the end
            st_chunk(dst, valid ? R[i] : zero_chunk());
        } else st_chunk(dst, R[i]);                                       // the slab past the end goes to the free buffer, never read
    };
    auto ld_advance = [&]() __attribute__((always_inline)) { if (d_slab + 1 < nsteps) ++d_slab; };

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero(acc[i][j]);
    // bias gradient (column sums of dY) of the ky == 0 blocks: accumulated from the dY pieces while they pass through the
    // registers (8 columns x the thread's token rows; VALU only -- the previous version re-read every column of the slab
    // from LDS, 64 two-byte reads per thread and slab, in the pipe this kernel is bound by), combined through LDS at the end
    float bs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && ky == 0;
    auto bias_acc = [&](int i, int slab) __attribute__((always_inline)) {
        if constexpr (RAGGED) {
            const int row = (wid * 4 + (i & 3)) * 2 + prow;
            if (!(t_begin + slab * TN256_BT + row < t_end)) return;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t u = R[i].v[w];
            bs8[2 * w] += lo16_f<T>(u);
            bs8[2 * w + 1] += hi16_f<T>(u);
        }
    };

    // transpose-read lane geometry (see frag_load_tr): lane l of a 16-lane group points at token row l / 4,
    // columns 4 (l % 4)..; the group receives 4 consecutive token rows of 16 columns
    const int l16 = lane & 15, h = lane >> 5;
    const int trow = l16 >> 2, tcol = ((lane >> 4) & 1) * 16 + 4 * (l16 & 3);
    const uint32_t fa0 = (uint32_t)(((8 * h + trow) * TN256_LD + wr * 128 + tcol) * 2);
    const uint32_t fb0 = (uint32_t)(TN256_OP + ((8 * h + trow) * TN256_LD + wc * 64 + tcol) * 2);

    // ---- prologue: slab 0 into LDS, slab 1 into the registers
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_piece(i);
    ld_advance();
#pragma unroll
    for (int i = 0; i < 8; ++i) st_piece(i, 0);
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_acc(i, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_piece(i);
    ld_advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot_barrier();

    for (int step = 0; step < nsteps; ++step) {
        const char* buf = smem + (step & 1) * (2 * TN256_OP);
        const bool bias_step = do_bias && step + 1 < nsteps;             // the slab past the end is a re-load of the last one
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            Frag<T> fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T* pa = reinterpret_cast<const T*>(buf + fa0 + (kk * 16 * TN256_LD + i * 32) * 2);
                fa[i].v = __builtin_shufflevector(lds_tr4(pa), lds_tr4(pa + 4 * TN256_LD), 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const T* pb = reinterpret_cast<const T*>(buf + fb0 + (kk * 16 * TN256_LD + j * 32) * 2);
                fb[j].v = __builtin_shufflevector(lds_tr4(pb), lds_tr4(pb + 4 * TN256_LD), 0, 1, 2, 3, 4, 5, 6, 7);
            }
            st_piece(2 * kk, step + 1);
            st_piece(2 * kk + 1, step + 1);
            if (kk < 2 && bias_step) { bias_acc(2 * kk, step + 1); bias_acc(2 * kk + 1, step + 1); }
            ld_piece(2 * kk);
            ld_piece(2 * kk + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32(acc[i][j], fa[i], fb[j]);
        }
        ld_advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // my stores of slab step+1 are in LDS
        slot_barrier();                                                    // everybody's; slab `step` fully consumed
    }
    if (do_bias) {                                                         // block uniform; the slab buffers are free now
        float* red = reinterpret_cast<float*>(smem);                       // [16 (wave, token-row parity)][256 columns]
        float* mine = red + (wid * 2 + prow) * 256 + pcol;
        *reinterpret_cast<f32x4_t*>(mine) = (f32x4_t){bs8[0], bs8[1], bs8[2], bs8[3]};
        *reinterpret_cast<f32x4_t*>(mine + 4) = (f32x4_t){bs8[4], bs8[5], bs8[6], bs8[7]};
        __syncthreads();
        if (tid < 256) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * 256 + tid];
            // with a workspace the column sums of this token range go to its bias region (after all partial tiles) and the
            // reduce kernel adds the ranges in a fixed order -- like dW, db is then reproducible bit for bit
            if (ws) ws[(size_t)gridDim.x * (32 * 512 * 4) + (size_t)blockIdx.x * 256 + tid] = t;
            else if (n0 + tid < N) atomicAdd(&dbias[n0 + tid], t);
        }
    }
    if (ws) {
        // partial tile -> workspace in register order: slot (i, j, q) of thread tid is one 16-byte store, 1 KiB
        // contiguous per wave instruction; tn256_reduce_kernel adds the token ranges in a fixed order
        f32x4_t* w = reinterpret_cast<f32x4_t*>(ws) + (size_t)blockIdx.x * (32 * 512) + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w[((i * 2 + j) * 4 + q) * 512] = (f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wc * 64 + j * 32 + c_col(lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wr * 128 + i * 32 + c_row(r, lane);
                if (row < N) atomicAdd(&dW[(size_t)row * lddw + col], acc[i][j][r]);
            }
        }
}

// dW tile (nx, ky) += sum over the token ranges z of the partial tiles gemm_tn256_kernel left in the
// workspace (67 MB of f32 atomics cost 40-48 us per launch, the same bytes as plain stores + this pass
// ~15 us, and the sum no longer depends on the arrival order).  grid (tn * tk, 32 slots), 512 threads.
__global__ __launch_bounds__(512) void tn256_reduce_kernel(const float* __restrict__ ws, const tn_group G) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    const int gtile = blockIdx.x, nsplit = G.nsplit;
    int pi = 0;
    while (pi + 1 < G.np && gtile >= G.p[pi + 1].tile0) ++pi;
    float* __restrict__ dW = G.p[pi].dW;
    const int lddw = G.p[pi].lddw, N = G.p[pi].N, tn = G.p[pi].tn;
    const int tile = gtile - G.p[pi].tile0;
    const int nx = tile % tn, ky = tile / tn;
    const int slot = blockIdx.y, q = slot & 3, j = (slot >> 2) & 1, i = slot >> 3;
    f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
    const int ntile = G.ntile, per = (ntile * nsplit + 7) / 8;
    auto part = [&](int z) -> f32x4_t {
        const int g = z * ntile + gtile;
        const int blk = (g % per) * 8 + g / per;                              // inverse of the kernel's block order
        return reinterpret_cast<const f32x4_t*>(ws)[((size_t)blk * 32 + slot) * 512 + tid];
    };
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {                                         // 8 loads in flight, fixed summation order
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part(z + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; z < nsplit; ++z) sum += part(z);
    const int col = ky * 256 + wc * 64 + j * 32 + (lane & 31);
    const int row = nx * 256 + wr * 128 + i * 32 + 8 * q + 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (row + e < N) dW[(size_t)(row + e) * lddw + col] += sum[e];
    // bias column sums of the tile column's first tile (ky = 0): the token ranges in the same fixed order
    float* __restrict__ dbias = G.p[pi].dbias;
    if (dbias && ky == 0 && slot == 0 && tid < 256 && nx * 256 + tid < N) {
        const float* wsb = ws + (size_t)per * 8 * (32 * 512 * 4);               // after the partial tiles of all gridDim.x = 8 per blocks
        float t = 0.f;
        for (int z2 = 0; z2 < nsplit; ++z2) {
            const int g = z2 * ntile + gtile;
            t += wsb[(size_t)((g % per) * 8 + g / per) * 256 + tid];
        }
        dbias[nx * 256 + tid] += t;
    }
}

// f32 master -> T copy and/or T transposed copy, 32x32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, int rows, int cols,
                                                             T* __restrict__ dst, int ld_dst, T* __restrict__ dstT,
                                                             int ld_dstT) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        float v = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
        tile[ty + i * 8][tx] = v;
        if (dst && r < rows && c < cols) dst[(size_t)r * ld_dst + c] = ET<T>::from_f(v);
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;    // dstT[c][r]
        if (c < cols && r < rows) dstT[(size_t)c * ld_dstT + r] = ET<T>::from_f(tile[tx][ty + i * 8]);
    }
}

// Multi-tensor variant: one launch refreshes every prepared weight of the model (31 tensors at the headline
// config: 31 launches of ~5 us for 164 MB of traffic that takes ~35 us at bandwidth).  desc = device array of
// me_ct_desc.  Persistent blocks walk 64 x 64 supertiles (all tensors, in descriptor order): 16-byte loads of the f32
// master rows (256 B per row segment), 8-byte stores of the T copy straight from the registers and, through the LDS
// tile, 16-byte stores of the transposed copy (128 B per segment) -- the round-2 version moved one 32 x 32 tile per
// block with 4-byte loads and 2-byte stores (80 k blocks, 2.6 TB/s).  Tensors whose shape or leading dimensions do not
// allow the vector accesses take guarded element accesses on the same walk.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const me_ct_desc* __restrict__ desc, int n) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    int ti = 0, base = 0;
    me_ct_desc dsc = desc[0];
    auto nsup = [](const me_ct_desc& d) { return ((d.rows + 63) / 64) * ((d.cols + 63) / 64); };
    int cnt = nsup(dsc);
    for (int st = blockIdx.x;; st += gridDim.x) {
        while (st >= base + cnt) {
            base += cnt;
            if (++ti >= n) return;
            dsc = desc[ti];
            cnt = nsup(dsc);
        }
        const int local = st - base;
        const int rows = dsc.rows, cols = dsc.cols;
        const int tiles_x = (cols + 63) / 64;
        const int r0 = (local / tiles_x) * 64, c0 = (local % tiles_x) * 64;
        T* dst = reinterpret_cast<T*>(dsc.dst);
        T* dstT = reinterpret_cast<T*>(dsc.dstT);
        const bool vsrc = (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(dsc.src) & 15) == 0;
        const bool vdst = dst && (dsc.ld_dst & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
        {
            const int cl = (tid & 15) * 4, c = c0 + cl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = (tid >> 4) + 16 * i, r = r0 + rl;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (r < rows) {
                    if (vsrc && c + 3 < cols) {
                        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(dsc.src + (size_t)r * cols + c);
                        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = dsc.src[(size_t)r * cols + c + e];
                    }
                    if (dst) {
                        if (vdst && c + 3 < cols) st4<T>(dst + (size_t)r * dsc.ld_dst + c, v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (c + e < cols) dst[(size_t)r * dsc.ld_dst + c + e] = ET<T>::from_f(v[e]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[rl][cl + e] = v[e];
            }
        }
        if (dstT) {
            __syncthreads();
            if (dsc.mode == ME_CT_PACK_REL) {
                // relative table: a 32 x 32 tile (row block eb, column block ib) holds two row images (kk = 2 ib, 2 ib + 1)
                // and the two transposed images (ib, t = 0 / 1) of the packed block -- layout of rel_pack_kernel (me_attn.hip)
                const int KA = cols / 16, DB = (cols + 31) / 32;
                const int img = (tid >> 6) & 1, lane = tid & 63, a = lane & 31, h = lane >> 5, j0 = (tid >> 7) * 4;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const int sr = (sub >> 1) * 32, sc = (sub & 1) * 32;
                    if (r0 + sr >= rows || c0 + sc >= cols) continue;
                    const int eb = (r0 + sr) >> 5, ib = (c0 + sc) >> 5;
                    T* blk = dstT + (size_t)eb * (KA + 2 * DB) * 512;
                    if (2 * ib + img < KA) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            blk[((2 * ib + img) * 64 + lane) * 8 + j0 + j] = ET<T>::from_f(tile[sr + a][sc + img * 16 + h * 8 + j0 + j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        blk[(KA + 2 * ib + img) * 512 + lane * 8 + j0 + j] = ET<T>::from_f(tile[sr + 16 * img + 8 * h + j0 + j][sc + a]);
                }
            } else {
                const bool vT = (dsc.ld_dstT & 7) == 0 && (reinterpret_cast<uintptr_t>(dstT) & 15) == 0;
                const int rg = (tid & 7) * 8, r = r0 + rg;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int cl = (tid >> 3) + 32 * i, c = c0 + cl;      // dstT[c][r .. r + 7]
                    if (c < cols && r < rows) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = tile[rg + j][cl];
                        T* o = dstT + (size_t)c * dsc.ld_dstT + r;
                        if (vT && r + 7 < rows) {
                            st4<T>(o, v[0], v[1], v[2], v[3]);
                            st4<T>(o + 4, v[4], v[5], v[6], v[7]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) if (r + j < rows) o[j] = ET<T>::from_f(v[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();               // the tile is rewritten by the next supertile
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static const bool g_disable_nt256 = getenv("MIDIEMO_NO_NT256") != nullptr;
// CUs the persistent kernels may occupy on the current device: multiProcessorCount minus MIDIEMO_CU_RESERVE (CUs left
// to a concurrent RCCL kernel when the gradient all-reduce overlaps the backward; default 0).  A query, not a
// synchronisation; cached per device; 256 (MI355X) when no device is visible (host-only symbol checks).
static int persistent_cus() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (dev >= 0 && dev < 16 && cached[dev]) return cached[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    static const int reserve = getenv("MIDIEMO_CU_RESERVE") ? atoi(getenv("MIDIEMO_CU_RESERVE")) : 0;
    if (reserve > 0 && reserve < n) n -= reserve;
    if (n > 8) n &= ~7;                                   // whole blocks per XCD round (tile renumbering)
    if (dev >= 0 && dev < 16) cached[dev] = n;
    return n;
}

// shapes the persistent 256 x 256 bf16 kernels take (me_workspace_bytes(ME_WS_RELU_MASK) answers with the same test).
// Few tiles on many CUs (the sliding-window forward of generate(): M = 4 x 1024 rows): a launch costs one whole
// 256 x 256 x K tile regardless, and 128 x 128 tiles finish sooner -- measured at M = 2048 .. 8192, N = 512 / 1007:
// 15.0-16.2 vs 17.1-20.5 us (K = 512), 33.5 vs 45.6 us (K = 2048); below M = 2048 and above 64 tiles the big tile wins
static bool nt256_shape_ok(int M, int N, int K) {
    const long t256 = (long)((N + 255) / 256) * ((M + 255) / 256);
    const bool few_tiles = M >= 2048 && t256 * 4 <= persistent_cus();
    return K % 64 == 0 && M >= 256 && N >= 192 && !g_disable_nt256 && !few_tiles;
}

template <typename T>
int gemm_nt_launch(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                   const void* add, int ldadd, const void* gate, int ldgate, int M, int N, int K, int flags,
                   hipStream_t st, int mask_dir = -1) {
    // mask_dir >= 0 (me_gemm_nt_relu_mask): `gate` is the ReLU sign mask, written (0, with bias + ReLU) or applied (1)
    constexpr int CH = ET<T>::CH;
    if (M <= 0 || N <= 0 || K <= 0) return ME_ERR_BAD_SHAPE;
    if (K % CH || lda % CH || ldb % CH) return ME_ERR_BAD_SHAPE;
    if (!aligned16(A) || !aligned16(B)) return ME_ERR_ALIGNMENT;
    if constexpr (sizeof(T) == 2) {
        // the 256-tile kernel addresses its operands with 32-bit byte offsets
        const bool off32 = (unsigned long long)M * lda * 2ull < (1ull << 32) && (unsigned long long)N * ldb * 2ull < (1ull << 32);
        if (nt256_shape_ok(M, N, K) && off32) {
            static bool attr_set[16] = {false};                               // (one flag array per instantiation of this launcher, i.e. per T)
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
            if (dev < 0 || dev >= 16 || !attr_set[dev]) {                   // the attribute is per device
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, false, 2, 4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, true, 2, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, true, 2, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS);
                if (dev >= 0 && dev < 16) attr_set[dev] = true;
            }
            unsigned g256 = (unsigned)(((N + 255) / 256) * ((M + 255) / 256));
            const unsigned ncu = (unsigned)persistent_cus();
            if (g256 > ncu) g256 = ncu;              // persistent: one block per CU
            // write-out path (EPI): the row paths need 16-byte aligned operand rows, a vector-storable C and N % 8 == 0
            const bool vec_c = (ldc % 8) == 0 && aligned16(C) && (N & 7) == 0;
            int epi = 3;
            if (mask_dir >= 0) {
                // sign-mask write-outs: bf16 rows stored as whole 16-byte chunks, mask blocks of 64 columns, 8-byte aligned words
                if ((flags & ME_EPI_OUT_F32) || add || !gate || !vec_c || (N & 63) || !aligned16(gate)) return ME_ERR_BAD_SHAPE;
                if (mask_dir == 1 && (bias || (flags & ME_EPI_RELU))) return ME_ERR_BAD_SHAPE;
                epi = mask_dir == 0 ? 4 : 5;
            } else if (!add && !gate) epi = 0;
            else if (!(flags & ME_EPI_OUT_F32) && gate && !add && !bias && !(flags & ME_EPI_RELU) && vec_c && (ldgate & 7) == 0 && aligned16(gate)) epi = 1;
            else if (!(flags & ME_EPI_OUT_F32) && add && !gate && vec_c && (ldadd & 7) == 0 && aligned16(add)) epi = 2;
#define ME_NT256(F32, E) gemm_nt256_kernel<T, F32, 2, 4, E><<<g256, 512, NT256_LDS, st>>>((const T*)A, lda, (const T*)B, ldb, C, ldc, bias, \
                                                                                  (const T*)add, ldadd, (const T*)gate, ldgate, M, N, K, flags)
            if (flags & ME_EPI_OUT_F32) { if (epi == 0) ME_NT256(true, 0); else ME_NT256(true, 3); }
            else if (epi == 0) ME_NT256(false, 0);
            else if (epi == 1) ME_NT256(false, 1);
            else if (epi == 2) ME_NT256(false, 2);
            else if (epi == 4) ME_NT256(false, 4);
            else if (epi == 5) ME_NT256(false, 5);
            else ME_NT256(false, 3);
#undef ME_NT256
            return me_launch_status();
        }
    }
    if (mask_dir >= 0) return ME_ERR_BAD_SHAPE;                          // the sign-mask write-outs exist in the 256-tile kernel only
    const unsigned grid = (unsigned)(((N + BN - 1) / BN) * ((M + BM - 1) / BM));
    if (flags & ME_EPI_OUT_F32)
        gemm_nt_kernel<T, true><<<grid, NTHREADS, 0, st>>>((const T*)A, lda, (const T*)B, ldb, C, ldc, bias, (const T*)add, ldadd,
                                                          (const T*)gate, ldgate, M, N, K, flags);
    else
        gemm_nt_kernel<T, false><<<grid, NTHREADS, 0, st>>>((const T*)A, lda, (const T*)B, ldb, C, ldc, bias, (const T*)add, ldadd,
                                                           (const T*)gate, ldgate, M, N, K, flags);
    return me_launch_status();
}

// plan of a (grouped) 256-tile launch: token ranges so that tiles x ranges just fills the chip (one block per CU)
struct tn_item_256 { const void* A; int lda; const void* B; int ldb; float* dW; int lddw; float* dbias; int N, K; };
static void tn256_plan(int Tn, int ntile, int* ns_out, int* tp_out, int* grid_out) {
    int ns = persistent_cus() / ntile;
    if (ns < 1) ns = 1;
    int tp = (Tn + ns - 1) / ns;
    tp = ((tp + TN256_BT - 1) / TN256_BT) * TN256_BT;
    ns = (Tn + tp - 1) / tp;
    *ns_out = ns; *tp_out = tp; *grid_out = ((ntile * ns + 7) / 8) * 8;
}
static bool tn256_eligible(int Tn, int N, int K, int lda, int ldb) {
    static const bool no256 = getenv("MIDIEMO_NO_TN256") != nullptr;
    const bool off32 = (unsigned long long)Tn * lda * 2ull < (1ull << 32) && (unsigned long long)Tn * ldb * 2ull < (1ull << 32);
    const int n256 = ((N + 255) / 256) * 256;
    return !no256 && (N % 256 == 0 || lda >= n256) && K % 256 == 0 && Tn >= 2048 && off32;
}
template <typename T>
static int tn256_group_launch(const tn_item_256* it, int n, int Tn, void* ws_caller, size_t ws_bytes, hipStream_t st) {
    static bool attr_set[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {                       // the attribute is per device
        (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TN256_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TN256_LDS);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    tn_group G;
    int ntile = 0;
    for (int i = 0; i < n; ++i) {
        tn_prod& p = G.p[i];
        p.A = it[i].A; p.B = it[i].B; p.dW = it[i].dW; p.dbias = it[i].dbias;
        p.lda = it[i].lda; p.ldb = it[i].ldb; p.lddw = it[i].lddw; p.N = it[i].N;
        p.tn = (it[i].N + 255) / 256; p.tk = it[i].K / 256; p.tile0 = ntile; p.pad_ = 0;
        ntile += p.tn * p.tk;
    }
    for (int i = n; i < ME_TN_MAX_GROUP; ++i) G.p[i] = G.p[n - 1];
    int ns, tp, grid256;
    tn256_plan(Tn, ntile, &ns, &tp, &grid256);
    G.np = n; G.ntile = ntile; G.nsplit = ns; G.t_per_block = tp; G.Tn = Tn; G.pad_ = 0;
    // partial tiles (256 KB per block) go to the caller's workspace and are summed in a fixed order by
    // tn256_reduce_kernel; without a workspace the blocks accumulate with f32 atomics (order-dependent sum)
    const size_t need = (size_t)grid256 * (32 * 512 * 16 + 1024);             // partial tiles + 256 bias column sums per block
    float* ws = nullptr;
    if (ws_caller && ns > 1) {
        if (ws_bytes < need || !aligned16(ws_caller)) return ME_ERR_WORKSPACE;
        ws = reinterpret_cast<float*>(ws_caller);
    }
    // whole slabs in every range is all the unpredicated feed needs (tp is a multiple of the slab; the last range may be shorter):
    // at T = 32768 and 5 ranges (103 / 103 / 103 / 103 / 100 slabs) round 3 took the predicated kernel for no reason
    if (Tn % TN256_BT == 0) gemm_tn256_kernel<T, false><<<grid256, 512, TN256_LDS, st>>>(G, ws);
    else gemm_tn256_kernel<T, true><<<grid256, 512, TN256_LDS, st>>>(G, ws);
    if (ws) tn256_reduce_kernel<<<dim3(ntile, 32), 512, 0, st>>>(ws, G);
    return me_launch_status();
}

template <typename T>
int gemm_tn_launch(const void* A, int lda, const void* B, int ldb, float* dW, int lddw, float* dbias, int Tn, int N,
                   int K, void* ws_caller, size_t ws_bytes, hipStream_t st) {
    constexpr int CH = ET<T>::CH;
    if (Tn <= 0 || N <= 0 || K <= 0) return ME_ERR_BAD_SHAPE;
    if (lda % CH || ldb % CH || K % CH) return ME_ERR_BAD_SHAPE;
    if (lda < ((N + CH - 1) / CH) * CH) return ME_ERR_BAD_SHAPE;   // padded rows must be readable
    if (!aligned16(A) || !aligned16(B)) return ME_ERR_ALIGNMENT;
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    int nsplit = (1024 + tn * tk - 1) / (tn * tk);
    const int max_split = (Tn + 4 * BT - 1) / (4 * BT);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int t_per = (Tn + nsplit - 1) / nsplit;
    t_per = ((t_per + BT - 1) / BT) * BT;
    nsplit = (Tn + t_per - 1) / t_per;
    dim3 grid(tn, tk, nsplit);
    if constexpr (sizeof(T) == 2) {
        if (tn256_eligible(Tn, N, K, lda, ldb)) {
            tn_item_256 it = {A, lda, B, ldb, dW, lddw, dbias, N, K};
            return tn256_group_launch<T>(&it, 1, Tn, ws_caller, ws_bytes, st);
        }
        const int npairs8 = ((tn * nsplit + 7) / 8) * 8;
        gemm_tn16_kernel<T><<<npairs8 * tk, NTHREADS, 0, st>>>((const T*)A, lda, (const T*)B, ldb, dW, lddw, dbias, Tn, N, K, t_per, tn, tk,
                                                               nsplit);
    } else
        gemm_tn_kernel<T><<<grid, NTHREADS, 0, st>>>((const T*)A, lda, (const T*)B, ldb, dW, lddw, dbias, Tn, N, K, t_per);
    return me_launch_status();
}

}  // namespace

extern "C" {

int me_abi_version(void) { return ME_ABI_VERSION; }

int me_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, const void* add,
               int ldadd, const void* gate, int ldgate, int M, int N, int K, int flags, int dtype, void* stream) {
    me_clear_error();
    if (!A || !B || !C) return ME_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ME_F32) return gemm_nt_launch<float>(A, lda, B, ldb, C, ldc, bias, add, ldadd, gate, ldgate, M, N, K, flags, st);
    if (dtype == ME_BF16) return gemm_nt_launch<bf16_t>(A, lda, B, ldb, C, ldc, bias, add, ldadd, gate, ldgate, M, N, K, flags, st);
    if (dtype == ME_F16) return gemm_nt_launch<f16_t>(A, lda, B, ldb, C, ldc, bias, add, ldadd, gate, ldgate, M, N, K, flags, st);
    return ME_ERR_BAD_DTYPE;
}

int me_gemm_nt_relu_mask(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, void* mask,
                         int M, int N, int K, int dir, int dtype, void* stream) {
    me_clear_error();
    if (!A || !B || !C || !mask) return ME_ERR_NULL;
    if (dir != 0 && dir != 1) return ME_ERR_BAD_SHAPE;
    if (dtype == ME_F32) return ME_ERR_BAD_SHAPE;                                          // the f32 tier keeps the gate operand
    if (dtype == ME_BF16)
        return gemm_nt_launch<bf16_t>(A, lda, B, ldb, C, ldc, dir == 0 ? bias : nullptr, nullptr, 0, mask, 0, M, N, K,
                                      dir == 0 ? ME_EPI_RELU : 0, (hipStream_t)stream, dir);
    if (dtype == ME_F16)
        return gemm_nt_launch<f16_t>(A, lda, B, ldb, C, ldc, dir == 0 ? bias : nullptr, nullptr, 0, mask, 0, M, N, K,
                                     dir == 0 ? ME_EPI_RELU : 0, (hipStream_t)stream, dir);
    return ME_ERR_BAD_DTYPE;
}

// bytes of partial-tile workspace a 256-tile launch over `ntile` tiles needs; 0 = no token split
static size_t tn_ws_bytes_tiles(int Tn, int ntile) {
    if (Tn < 2048 || ntile <= 0) return 0;
    int ns, tp, grid256;
    tn256_plan(Tn, ntile, &ns, &tp, &grid256);
    if (ns <= 1) return 0;
    return (size_t)grid256 * (32 * 512 * 16 + 1024);
}
// ... gemm_tn_launch<bf16> needs for (T, N, K); 0 = the shape runs a kernel without one
static size_t tn_ws_bytes(int Tn, int N, int K) {
    if (K % 256 != 0) return 0;
    return tn_ws_bytes_tiles(Tn, ((N + 255) / 256) * (K / 256));
}

size_t me_workspace_bytes(int op, int M, int N, int K, int dtype) {
    if (op == ME_WS_RELU_MASK)                                            // 64 bytes per 8 rows x 64 columns, rows rounded up to 256; 0 = use the gate operand
        return (dtype != ME_F32 && M > 0 && N > 0 && (N & 63) == 0 && nt256_shape_ok(M, N, K) &&
                (unsigned long long)M * K * 2ull < (1ull << 32) && (unsigned long long)N * K * 2ull < (1ull << 32))
                   ? (size_t)((M + 255) / 256) * 32 * (size_t)(N / 64) * 64 : 0;
    if (op == ME_WS_GEMM_TN) return dtype != ME_F32 ? tn_ws_bytes(M, N, K) : 0;
    if (op == ME_WS_GEMM_TN_GROUP) return dtype != ME_F32 ? tn_ws_bytes_tiles(M, N) : 0;      // N = total 256 x 256 tiles of the group
    if (op == ME_WS_RGA_PT || op == ME_WS_RGA_DGT) {
        // 32 x 32 tiles of the compute type per (batch, head): M = B*H, N = Lp (multiple of 32), K = causal flag
        if (M <= 0 || N <= 0 || (N & 31)) return 0;
        const size_t nq = (size_t)N / 32, es = dtype != ME_F32 ? 2 : 4;
        const size_t tiles = (op == ME_WS_RGA_DGT || K) ? nq * (nq + 1) / 2 : nq * nq;
        return (size_t)M * tiles * 1024 * es;
    }
    if (op == ME_WS_RGA_MT) return (M > 0 && N > 0 && !(N & 31)) ? (size_t)M * (N / 32) * N * sizeof(float) : 0;
    if (op == ME_WS_SUMSQ) return ME_SUMSQ_WS_BYTES;                      // ordered block sums of me_sumsq (zero before the first use)
    if (op == ME_WS_DEC_TOKEN) return (M > 0 && M <= ME_DEC_TOKEN_ROWS && N > 0 && K > 0) ? me_dec_token_ws_bytes(K, N) : 0;   // (M, N, K) = (Mr, d_inner, d)
    if (op == ME_WS_EMBED_BWD) return 1024;                               // frequent-token list of me_embed_bwd (zero before the first use)
    return 0;
}

int me_gemm_tn_acc(const void* A, int lda, const void* B, int ldb, float* dW, int lddw, float* dbias, int T, int N,
                   int K, void* ws, size_t ws_bytes, int dtype, void* stream) {
    me_clear_error();
    if (!A || !B || !dW) return ME_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ME_F32) return gemm_tn_launch<float>(A, lda, B, ldb, dW, lddw, dbias, T, N, K, ws, ws_bytes, st);
    if (dtype == ME_BF16) return gemm_tn_launch<bf16_t>(A, lda, B, ldb, dW, lddw, dbias, T, N, K, ws, ws_bytes, st);
    if (dtype == ME_F16) return gemm_tn_launch<f16_t>(A, lda, B, ldb, dW, lddw, dbias, T, N, K, ws, ws_bytes, st);
    return ME_ERR_BAD_DTYPE;
}

int me_gemm_tn_acc_group(const me_tn_item* items, int n_items, int T, void* ws, size_t ws_bytes, int dtype, void* stream) {
    me_clear_error();
    if (!items) return ME_ERR_NULL;
    if (n_items <= 0 || n_items > ME_TN_MAX_GROUP) return ME_ERR_BAD_SHAPE;
    for (int i = 0; i < n_items; ++i)
        if (!items[i].A || !items[i].B || !items[i].dW) return ME_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype != ME_F32 && dtype != ME_BF16 && dtype != ME_F16) return ME_ERR_BAD_DTYPE;
    bool grouped = dtype != ME_F32 && n_items > 1;
    for (int i = 0; grouped && i < n_items; ++i) {
        const me_tn_item& m = items[i];
        grouped = m.N > 0 && m.K > 0 && m.lda % 8 == 0 && m.ldb % 8 == 0 && aligned16(m.A) && aligned16(m.B) &&
                  tn256_eligible(T, m.N, m.K, m.lda, m.ldb);
    }
    if (grouped) {
        tn_item_256 it[ME_TN_MAX_GROUP];
        for (int i = 0; i < n_items; ++i)
            it[i] = {items[i].A, items[i].lda, items[i].B, items[i].ldb, items[i].dW, items[i].lddw, items[i].dbias, items[i].N, items[i].K};
        return dtype == ME_BF16 ? tn256_group_launch<bf16_t>(it, n_items, T, ws, ws_bytes, st)
                                : tn256_group_launch<f16_t>(it, n_items, T, ws, ws_bytes, st);
    }
    // shapes / types the grouped kernel does not take: one launch per product (same results as me_gemm_tn_acc)
    for (int i = 0; i < n_items; ++i) {
        const me_tn_item& m = items[i];
        const int rc = me_gemm_tn_acc(m.A, m.lda, m.B, m.ldb, m.dW, m.lddw, m.dbias, T, m.N, m.K, ws, ws_bytes, dtype, stream);
        if (rc) return rc;
    }
    return ME_OK;
}

int me_cast_transpose(const float* src, int rows, int cols, void* dst, int ld_dst, void* dstT, int ld_dstT,
                      int dtype, void* stream) {
    me_clear_error();
    if (!src || (!dst && !dstT)) return ME_ERR_NULL;
    if (rows <= 0 || cols <= 0) return ME_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (dtype == ME_F32)
        cast_transpose_kernel<float><<<grid, 256, 0, st>>>(src, rows, cols, (float*)dst, ld_dst, (float*)dstT, ld_dstT);
    else if (dtype == ME_BF16)
        cast_transpose_kernel<bf16_t><<<grid, 256, 0, st>>>(src, rows, cols, (bf16_t*)dst, ld_dst, (bf16_t*)dstT, ld_dstT);
    else if (dtype == ME_F16)
        cast_transpose_kernel<f16_t><<<grid, 256, 0, st>>>(src, rows, cols, (f16_t*)dst, ld_dst, (f16_t*)dstT, ld_dstT);
    else
        return ME_ERR_BAD_DTYPE;
    return me_launch_status();
}

int me_cast_transpose_multi(const me_ct_desc* desc_dev, int n_tensors, int total_tiles, int dtype, void* stream) {
    me_clear_error();
    if (!desc_dev) return ME_ERR_NULL;
    if (n_tensors <= 0 || total_tiles <= 0) return ME_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // persistent: every block walks the 64 x 64 supertiles st = blockIdx.x, + gridDim.x, ... (at most total_tiles of them)
    const unsigned grid = (unsigned)(total_tiles < 2048 ? total_tiles : 2048);
    if (dtype == ME_F32) cast_transpose_multi_kernel<float><<<grid, 256, 0, st>>>(desc_dev, n_tensors);
    else if (dtype == ME_BF16) cast_transpose_multi_kernel<bf16_t><<<grid, 256, 0, st>>>(desc_dev, n_tensors);
    else if (dtype == ME_F16) cast_transpose_multi_kernel<f16_t><<<grid, 256, 0, st>>>(desc_dev, n_tensors);
    else return ME_ERR_BAD_DTYPE;
    return me_launch_status();
}

}  // extern "C"
