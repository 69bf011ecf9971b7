// Device-side helpers shared by the decode kernels (me_decode.hip: one launch per stage; me_decode_token.hip: one
// persistent launch per token).  Everything here is arithmetic the two paths must perform IDENTICALLY -- the per-token
// kernel is tested bit for bit against the per-stage launches.
#pragma once
#include "me_common.h"
#include <type_traits>

namespace {

template <typename T> ME_DEV float round_to(float x) { return ET<T>::to_f(ET<T>::from_f(x)); }
template <typename T> ME_DEV void chunk_to_f32(const chunk16& c, float* f) {
    const T* e = reinterpret_cast<const T*>(&c);
#pragma unroll
    for (int i = 0; i < ET<T>::CH; ++i) f[i] = ET<T>::to_f(e[i]);
}

// a * b + c with the product and the sum rounded SEPARATELY, whatever the contraction pass would do.  The softmax-combine of the
// decode step accumulates its denominator this way in both decode paths: left to -ffp-contract=fast the same source line became
// v_mul + v_add in one kernel and v_fma / v_fmac (first two terms only -- the rest had been paired into v_pk_mul_f32) in the
// other, found by the bit-for-bit test of the two paths.  `#pragma clang fp contract(off)` and __fmul_rn / __fadd_rn do not
// prevent it (the backend fuses under AllowFPOpFusion = Fast regardless of instruction flags; -save-temps builds show different
// code than the real ones): the product goes through an opaque move.
ME_DEV float mul_add_unfused(float a, float b, float c) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p + c;
}

// sum over the G lanes (G = 4, 8, 16) of an aligned lane group; every lane receives the total
template <int G> ME_DEV float group_sum(float v) {
    v += dpp_move<0xB1>(v);                     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);                     // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_move<0x141>(v);        // row_half_mirror
    if (G >= 16) v += dpp_move<0x140>(v);       // row_mirror
    return v;
}

constexpr int DEC_NSMAX = 8;        // key splits the combine prologue unrolls over

// Sum NV per-lane values over the 64 lanes.  v_permlane32_swap / v_permlane16_swap exchange one half of a register
// pair, so each swap + add halves the number of live values (NV -> NV/2 -> NV/4) while summing lane pairs (l, l + 32)
// and (l, l + 16); the remaining NV/4 values are summed inside the 16-lane rows with four DPP adds each.  Result: every
// lane of row r = lane >> 4 holds, in v[i] (i < NV/4), the total of input value i + (NV/4) * r.  2.5 NV instructions
// instead of ~12 NV for NV independent wave reductions.  Inline asm: through __builtin_amdgcn_permlane{32,16}_swap hipcc
// (ROCm 7.2) folds the two results of a swap into one register (it emitted v_add v2, v3, v3 after v_permlane32_swap v3, v2).
template <int NV> ME_DEV void reduce_scatter64(float* v) {
    static_assert(NV % 4 == 0, "NV must be a multiple of 4");
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        float x = v[i], y = v[i + NV / 2];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
        v[i] = x + y;
    }
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
        float x = v[i], y = v[i + NV / 4];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
        v[i] = x + y;
    }
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
        v[i] += dpp_move<0xB1>(v[i]);
        v[i] += dpp_move<0x4E>(v[i]);
        v[i] += dpp_move<0x141>(v[i]);
        v[i] += dpp_move<0x140>(v[i]);
    }
}

template <typename T, int MR, int CW>
ME_DEV void dec_fma_chunks(float (&acc)[CW][MR], const chunk16 (&w)[CW], bool ok, const float* xs, int K, int chc) {
    constexpr int CH = ET<T>::CH;
    float wf[CW][CH];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const T* we = reinterpret_cast<const T*>(&w[c]);
#pragma unroll
        for (int i = 0; i < CH; ++i) wf[c][i] = ok ? ET<T>::to_f(we[i]) : 0.f;
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

// KS = false: every wave owns CW columns and the whole contraction (4 CW columns per block).
// KS = true : the block owns CW columns, wave w contracts over the w-th quarter of K and the four partial results meet in
//             LDS -- long rows (FFN_suf: K = 2048) then spread over as many blocks as the short ones: a cold weight
//             stream is fetched fastest when every CU pulls a few KB (measured: 64 blocks x 32 KB 9.7 us, L2-hot 4.4 us).
// weight rows: every element is read by exactly one wave per token -- streamed with the non-temporal policy (guide, price
// list "nt-weights": issued -> landed -18 %)
ME_DEV chunk16 ld_w(const void* p) {
    chunk16 c;
    c.v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return c;
}
// K / V / E rows of the cache: default policy (round 5, same box: non-temporal here 0.1575 vs 0.159 ms per token, plain weight
// loads 0.1565, both together 0.164 -- noise-level, nothing changed)
ME_DEV chunk16 ld_kv(const void* p) { return ld_chunk(p); }

}  // namespace
