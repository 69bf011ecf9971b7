// Shared device-side helpers for the midiemo gfx950 (CDNA4 / MI355X) kernels.
// wave = 64 lanes; every MFMA tile below is the 32x32 "macro-atom":
//     acc[32x32] += sum_{h in {0,1}, e in 0..7} a(i,h)[e] * b(j,h)[e]
// where lane = (i | h<<5) supplies 8 contraction elements of row i of A and
// lane = (j | h<<5) supplies 8 contraction elements of column j of B.
//   bf16 : one v_mfma_f32_32x32x16_bf16            (8 bf16 per lane, 16 B)
//   f16  : one v_mfma_f32_32x32x16_f16             (8 halves per lane, 16 B; same rate, 10 mantissa bits)
//   f32  : eight v_mfma_f32_32x32x2_f32 (exact f32) (8 floats per lane, 32 B)
// Because A and B always use the same (h,e)->k map, the hardware's own k
// ordering never matters; only the C/D map does:
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/midiemo.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define ME_WAVE 64
#define ME_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------
// C/D layout of the 32x32 accumulator
// ---------------------------------------------------------------------------
ME_DEV int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
ME_DEV int c_col(int lane) { return lane & 31; }

// ---------------------------------------------------------------------------
// element traits
// ---------------------------------------------------------------------------
template <typename T> struct ET;
template <> struct ET<bf16_t> {
    static constexpr int CH = 8;  // elements per 16-byte chunk
    ME_DEV static float to_f(bf16_t x) { return (float)x; }
    ME_DEV static bf16_t from_f(float x) { return (bf16_t)x; }
    ME_DEV static float fexp(float x) { return __expf(x); }
};
template <> struct ET<f16_t> {
    static constexpr int CH = 8;
    ME_DEV static float to_f(f16_t x) { return (float)x; }
    ME_DEV static f16_t from_f(float x) { return (f16_t)x; }
    ME_DEV static float fexp(float x) { return __expf(x); }
};
template <> struct ET<float> {
    static constexpr int CH = 4;
    ME_DEV static float to_f(float x) { return x; }
    ME_DEV static float from_f(float x) { return x; }
    ME_DEV static float fexp(float x) { return expf(x); }
};

// vector types of the two 16-bit element types (every 16-bit kernel is a template over T in {bf16_t, f16_t}: the data
// movement is identical, only the MFMA opcode and the conversions differ)
template <typename T> struct V16;
template <> struct V16<bf16_t> { typedef bf16x4_t x4; typedef bf16x8_t x8; };
template <> struct V16<f16_t> { typedef f16x4_t x4; typedef f16x8_t x8; };

// 16-byte chunk moved as one unit between global memory, registers and LDS
struct __attribute__((aligned(16))) chunk16 { u32x4_t v; };
ME_DEV chunk16 zero_chunk() { chunk16 c; c.v = (u32x4_t){0u, 0u, 0u, 0u}; return c; }
ME_DEV chunk16 ld_chunk(const void* p) { return *reinterpret_cast<const chunk16*>(p); }
ME_DEV void st_chunk(void* p, const chunk16& c) { *reinterpret_cast<chunk16*>(p) = c; }

// ---------------------------------------------------------------------------
// MFMA operand fragment: 8 contraction elements for one row/column
// ---------------------------------------------------------------------------
template <typename T> struct Frag { typename V16<T>::x8 v; };
template <> struct Frag<float> { f32x4_t lo, hi; };

// 8 contiguous elements (16-byte aligned for bf16, 16-byte aligned halves for f32)
template <typename T> ME_DEV void frag_load(Frag<T>& f, const T* p) { f.v = *reinterpret_cast<const typename V16<T>::x8*>(p); }
ME_DEV void frag_load(Frag<float>& f, const float* p) {
    f.lo = *reinterpret_cast<const f32x4_t*>(p);
    f.hi = *reinterpret_cast<const f32x4_t*>(p + 4);
}
// two groups of 4 contiguous elements (used with the accumulator-as-operand k map)
template <typename T> ME_DEV void frag_load_4x2(Frag<T>& f, const T* p0, const T* p1) {
    typedef typename V16<T>::x4 x4;
    x4 a = *reinterpret_cast<const x4*>(p0);
    x4 b = *reinterpret_cast<const x4*>(p1);
    f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
ME_DEV void frag_load_4x2(Frag<float>& f, const float* p0, const float* p1) {
    f.lo = *reinterpret_cast<const f32x4_t*>(p0);
    f.hi = *reinterpret_cast<const f32x4_t*>(p1);
}
// Transposed fragment from a NATURAL LDS tile tile[row][col] (row stride ld elements): the lane's operand
// row/column index is the tile COLUMN cbase + (lane & 31) and its 8 contraction elements are the tile rows
// rA..rA+3 and rB..rB+3.  bf16: two ds_read_b64_tr_b16 -- inside every 16-lane group lane l supplies the
// address of 4 contiguous elements and lane i receives element i%4 of the chunks of lanes i/4 + 4j (measured
// on gfx950), so with lane l pointing at row r + l/4, columns 4*(l%4).., lane i gets column i for 4
// consecutive rows.  f32: plain element gather (the exact tier is not performance critical).
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <typename T> ME_DEV typename V16<T>::x4 lds_tr4(const T* p) {        // one ds_read_b64_tr_b16 (type agnostic: 16-bit elements)
    v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(p));
    return __builtin_bit_cast(typename V16<T>::x4, r);
}
template <typename T> ME_DEV void frag_load_tr(Frag<T>& f, const T* tile, int ld, int rA, int rB, int cbase, int lane) {
    const int l16 = lane & 15;
    const T* p = tile + (l16 >> 2) * ld + cbase + ((lane >> 4) & 1) * 16 + 4 * (l16 & 3);
    f.v = __builtin_shufflevector(lds_tr4(p + rA * ld), lds_tr4(p + rB * ld), 0, 1, 2, 3, 4, 5, 6, 7);
}
ME_DEV void frag_load_tr(Frag<float>& f, const float* tile, int ld, int rA, int rB, int cbase, int lane) {
    const float* p = tile + cbase + (lane & 31);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.lo[e] = p[(rA + e) * ld]; f.hi[e] = p[(rB + e) * ld]; }
}
template <typename T> ME_DEV void frag_zero(Frag<T>& f) { f.v = (typename V16<T>::x8){0, 0, 0, 0, 0, 0, 0, 0}; }
ME_DEV void frag_zero(Frag<float>& f) { f.lo = (f32x4_t){0, 0, 0, 0}; f.hi = f.lo; }
template <typename T> ME_DEV void frag_set(Frag<T>& f, int e, float x) { f.v[e] = (T)x; }
ME_DEV void frag_set(Frag<float>& f, int e, float x) { if (e < 4) f.lo[e] = x; else f.hi[e - 4] = x; }
// the 8 elements of a fragment as one 16-byte (bf16) / two 16-byte (f32) stores
template <typename T> ME_DEV void frag_store(T* p, const Frag<T>& f) { *reinterpret_cast<typename V16<T>::x8*>(p) = f.v; }
ME_DEV void frag_store(float* p, const Frag<float>& f) {
    *reinterpret_cast<f32x4_t*>(p) = f.lo;
    *reinterpret_cast<f32x4_t*>(p + 4) = f.hi;
}
// elements 4 half .. 4 half + 3 of a fragment as one 8-byte (bf16) / 16-byte (f32) store
template <typename T> ME_DEV void frag_store_half(T* p, const Frag<T>& f, int half) {
    typename V16<T>::x4 v = half ? __builtin_shufflevector(f.v, f.v, 4, 5, 6, 7) : __builtin_shufflevector(f.v, f.v, 0, 1, 2, 3);
    *reinterpret_cast<typename V16<T>::x4*>(p) = v;
}
ME_DEV void frag_store_half(float* p, const Frag<float>& f, int half) { *reinterpret_cast<f32x4_t*>(p) = half ? f.hi : f.lo; }
template <typename T> ME_DEV float frag_get(const Frag<T>& f, int e) { return (float)f.v[e]; }
ME_DEV float frag_get(const Frag<float>& f, int e) { return e < 4 ? f.lo[e] : f.hi[e - 4]; }

// accumulator registers [8*t .. 8*t+7] -> operand fragment (k map = c_row of those registers)
template <typename T> ME_DEV void frag_from_acc(Frag<T>& f, const f32x16_t& a, int t) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = (T)a[8 * t + e];
}
ME_DEV void frag_from_acc(Frag<float>& f, const f32x16_t& a, int t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.lo[e] = a[8 * t + e]; f.hi[e] = a[8 * t + 4 + e]; }
}

ME_DEV void mma32(f32x16_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
ME_DEV void mma32(f32x16_t& acc, const Frag<f16_t>& a, const Frag<f16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
}
ME_DEV void mma32(f32x16_t& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[e], b.lo[e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[e], b.hi[e], acc, 0, 0, 0);
}
// four consecutive elements of T from four floats (one 8-byte / 16-byte store)
template <typename T> ME_DEV void st4(T* p, float a, float b, float c, float d) {
    typename V16<T>::x4 v; v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    *reinterpret_cast<typename V16<T>::x4*>(p) = v;
}
template <> ME_DEV void st4<float>(float* p, float a, float b, float c, float d) { *reinterpret_cast<f32x4_t*>(p) = (f32x4_t){a, b, c, d}; }
// the low / high 16-bit half of a dword as a float (element 2 w / 2 w + 1 of a chunk of T)
template <typename T> ME_DEV float lo16_f(uint32_t u);
template <typename T> ME_DEV float hi16_f(uint32_t u);
template <> ME_DEV float lo16_f<bf16_t>(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
template <> ME_DEV float hi16_f<bf16_t>(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <> ME_DEV float lo16_f<f16_t>(uint32_t u) { return (float)__builtin_bit_cast(f16_t, (uint16_t)(u & 0xffffu)); }
template <> ME_DEV float hi16_f<f16_t>(uint32_t u) { return (float)__builtin_bit_cast(f16_t, (uint16_t)(u >> 16)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also makes the compiler wait vmcnt(0), i.e. for
// every global load just issued as a prefetch and for every global store still draining (vmcnt is in-order on
// gfx950); inside pipelined loops only the LDS hand-over needs the barrier.
ME_DEV void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// s_waitcnt vmcnt(0) as a real instruction (the waitcnt pass accounts for it, unlike inline asm).  Placed in a loop's
// preheader it makes "nothing in flight" the entry state, so the conservative merge of entry and back-edge states at the
// loop header keeps the back edge's exact counts instead of draining the prefetches every iteration.
ME_DEV void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
ME_DEV void acc_zero(f32x16_t& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ---------------------------------------------------------------------------
// wave reductions (64 lanes)
// ---------------------------------------------------------------------------
// Wave-wide reductions without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0) per
// level: six serialised LDS round trips per reduction (measured: the LDS unit was busy half of resid_ln_bwd's
// run time).  Here: four DPP levels inside a 16-lane row (quad butterflies, half-row mirror, row mirror: every
// lane then holds its row's total), then the four row totals are combined through v_readlane.
template <int CTRL> ME_DEV float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <typename OP> ME_DEV float wave_reduce(float v, OP op) {
    v = op(v, dpp_move<0xB1>(v));       // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E>(v));       // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141>(v));      // row_half_mirror
    v = op(v, dpp_move<0x140>(v));      // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return op(op(r0, r1), op(r2, r3));
}
// op(v[lane], v[lane ^ 32]) in every lane: v_permlane32_swap exchanges the upper half of one register with the lower
// half of another (gfx950), so two copies of v become {lo, lo} and {hi, hi}
template <typename OP> ME_DEV float half_reduce(float v, OP op) {
    // inline asm: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folds the two results into one when both
    // inputs carry the same value (max(r0, r1) -> r0).  s_nop: VALU write -> permlane read / permlane write -> VALU read.
    unsigned lo = __builtin_bit_cast(unsigned, v), hi = lo;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
    return op(__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi));       // lo = {v.lo, v.lo}, hi = {v.hi, v.hi}
}
ME_DEV float half_sum(float v) { return half_reduce(v, [](float a, float b) { return a + b; }); }
ME_DEV float half_max(float v) { return half_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }
ME_DEV float wave_sum(float v) { return wave_reduce(v, [](float a, float b) { return a + b; }); }
ME_DEV float wave_max(float v) { return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }

// ---------------------------------------------------------------------------
// counter-based dropout RNG: 32-bit avalanche hash of (seed, site, element).
// One hash yields two 16-bit uniforms (two neighbouring elements).
// keep  <=>  u16 >= thr16  with thr16 = round(p * 65536)
// ---------------------------------------------------------------------------
ME_DEV uint32_t me_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
ME_DEV uint32_t me_rng_pair(uint64_t seed, uint32_t site, uint64_t pair_index) {
    uint32_t lo = (uint32_t)pair_index, hi = (uint32_t)(pair_index >> 32);
    uint32_t s = me_hash32((uint32_t)seed ^ (site * 0x9E3779B9U) ^ (hi * 0x85EBCA6BU));
    return me_hash32(lo ^ s ^ (uint32_t)(seed >> 32));
}
// keep-flag of element idx (consistent with the pair generator)
ME_DEV bool me_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr16) {
    uint32_t r = me_rng_pair(seed, site, idx >> 1);
    uint32_t u = (idx & 1) ? (r >> 16) : (r & 0xFFFFu);
    return u >= thr16;
}

// me_dec_token's workspace (me_workspace_bytes(ME_WS_DEC_TOKEN)): a 256-byte control block (launch epoch, blocks done, error
// word) followed by the 8-byte exchange records of ME_DEC_TOKEN_ROWS sequences: s2, s1, att [rows][d]; qkv [rows][3d]; hid [rows][d_inner];
// attention partials [rows * H][nsplit <= 8][dh + 2] <= rows * 8 * (d + d / 16 + 2) (dh >= 32)
#define ME_DEC_TOKEN_ROWS 4
static inline size_t me_dec_token_ws_bytes(int d, int d_inner) {
    return 256 + 8 * ((size_t)ME_DEC_TOKEN_ROWS * d * 6 + (size_t)ME_DEC_TOKEN_ROWS * d_inner + (size_t)ME_DEC_TOKEN_ROWS * 8 * (d + d / 16 + 2));
}

// status helpers for the C-ABI launchers.  hipGetLastError() is sticky per thread: other users
// of the runtime (PyTorch) routinely leave benign non-success codes behind, so every entry point
// clears the state first and only reports errors raised by its own launches.
static inline void me_clear_error() { (void)hipGetLastError(); }
static inline int me_launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess && getenv("MIDIEMO_DEBUG")) fprintf(stderr, "midiemo: launch failed: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? ME_OK : ME_ERR_LAUNCH;
}
