// Helpers shared by the relative-global-attention translation units (me_attn.hip: generic kernels + C entry points,
// me_attn64.hip: the 16-bit / head-dim-64 / causal kernels of the training hot path).
#pragma once
#include "me_common.h"
#include <type_traits>

namespace me_attn {


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
        if (TT::NCH % 256 == 0 || c < TT::NCH) {
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
        // a whole number of 256-chunk rounds: no predicate -- an exec-masked region here splits the step into basic
        // blocks and the s_waitcnt pass then drains vmcnt to 0 at the loop header (every prefetch latency exposed)
        if (TT::NCH % 256 == 0 || c < TT::NCH) {
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
// A dG^T tile [32 E rows m][32 queries q] is stored as the E kernel's two A-operand fragment images: image t = q / 16,
// lane (a = m, h = (q / 8) % 2) holds queries 16 t + 8 h .. + 7 contiguously -- consecutive lanes read consecutive 16-byte
// (bf16) pieces, one contiguous KB per wave load (row-major tiles made every lane its own 64-byte-granule request).
ME_DEV int dg_pos(int m, int q) { return (q >> 4) * 512 + (m + 32 * ((q >> 3) & 1)) * 8 + (q & 7); }
// Probability tiles are stored as the forward's REGISTER IMAGE: row q holds its 32 keys in the order the two lanes
// (q, h = 0 / 1) own them (accumulator registers 0..15 of lane (q, h) = keys 8 g + 4 h + i, g = r / 4, i = r % 4), i.e.
// key -> position 16 h + 4 g + i.  A lane then writes / reads back 16 contiguous elements (a wave: one contiguous tile,
// full cache lines) instead of four 8-byte pieces scattered over 32 rows (measured: +80 us per forward launch).
ME_DEV int p_col(int key) { return ((key >> 2) & 1) * 16 + (key >> 3) * 4 + (key & 3); }

// Streaming accesses (workspace tiles written once and read once or twice much later): non-temporal loads / stores
// keep them from displacing the K / V / Q rows the other blocks re-read through L2.
template <typename V> ME_DEV V nt_load(const V* p) {
    return __builtin_nontemporal_load(p);
}
template <typename V> ME_DEV void nt_store(V v, V* p) {
    __builtin_nontemporal_store(v, p);
}
template <typename T> ME_DEV void frag_load_nt(Frag<T>& f, const T* p) { f.v = nt_load(reinterpret_cast<const typename V16<T>::x8*>(p)); }
ME_DEV void frag_load_nt(Frag<float>& f, const float* p) {
    f.lo = nt_load(reinterpret_cast<const f32x4_t*>(p));
    f.hi = nt_load(reinterpret_cast<const f32x4_t*>(p + 4));
}

// v_exp_f32 without the denormal-range fix-up of exp2f (arguments here are <= 0: tiny results may flush to 0)
ME_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace me_attn

// 16-bit / head dim 64 / causal instantiations (me_attn64.hip); same arguments and workspace formats as the generic launchers
namespace me_attn64 {
template <typename T>
int fwd_launch(const void* qkv, const void* Epk, const uint8_t* key_pad, void* out, float* lse, void* PT, float* MT, int B,
               int L, int H, int M, hipStream_t st);
}
