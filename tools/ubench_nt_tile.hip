// Microbenchmark (round 5): what does a FULL-ROW output tile cost the NT GEMM main loop?
// LayerNorm in the write-out of the N = d products needs every CU to own whole rows of C: a 128 x 512 tile (the accumulators
// fill the register file either way: 8 waves x 128 x 64) instead of 256 x 256.  Per 64-deep slab that is 80 KB of operands
// instead of 64 KB for the same 8.4 MFLOP.  This program runs the SAME register-staged main loop as gemm_nt256_kernel
// (me_gemm.hip: slab images, source-chunk swizzle, 8 x global_load_dwordx4 a slab ahead, ds_write_b128 into the other buffer,
// one raw barrier per slab) for both tile shapes, one tile per block, with a token write-out (one checksum per lane), and
// times them interleaved.  The global sum of C is checked against sum_k (sum_i A_ik)(sum_j B_jk).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench_nt_tile.hip -o gpurun_out/ubench_nt_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(16))) chunk16 { u32x4_t v; };

template <int BMT, int BNT, int WR, int WC>
__global__ __launch_bounds__(512) void nt_tile_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                      float* __restrict__ out, int M, int N, int K) {
    static_assert(WR * WC == 8, "8 waves");
    constexpr int TM = BMT / WR, TN = BNT / WC, AI = TM / 32, BJ = TN / 32;
    constexpr int NPA = BMT / 8, NPB = BNT / 8, PPW = (NPA + NPB) / 8;     // 1 KB pieces per slab: A, B, per wave
    constexpr int BUF = (BMT + BNT) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][A BMT x 128 B | B BNT x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WC, wc = wid % WC;
    const int ntn = N / BNT;
    int t = blockIdx.x;
    const int ntiles = ntn * (M / BMT);
    if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
    const int m0 = (t / ntn) * BMT, n0 = (t % ntn) * BNT;
    const int nk = K / 64;
    const int lrow = lane >> 3;
    u32x4_t R[PPW];
    const uint32_t lda2 = lda * 2u, ldb2 = ldb * 2u;
    const char* Ab = reinterpret_cast<const char*>(A) + (size_t)m0 * lda2;
    const char* Bb = reinterpret_cast<const char*>(B) + (size_t)n0 * ldb2;
    auto ld_piece = [&](int i, int k) __attribute__((always_inline)) {
        const int g = wid * PPW + i;                                       // piece of the slab
        const bool isA = g < NPA;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                       // recompute the address at the issue: nothing hoisted, nothing spilled
        const int r = (isA ? g : g - NPA) * 8 + (ln >> 3);
        const uint32_t col = (uint32_t)((((ln & 7) ^ ((r ^ (r >> 3)) & 7)) << 4) + k * 2);
        R[i] = *reinterpret_cast<const u32x4_t*>(isA ? Ab + ((uint32_t)r * lda2 + col) : Bb + ((uint32_t)r * ldb2 + col));
    };
    auto st_piece = [&](int i, int slab) __attribute__((always_inline)) {
        const int g = wid * PPW + i;
        *reinterpret_cast<u32x4_t*>(smem + (slab & 1) * BUF + g * 1024 + lane * 16) = R[i];      // A pieces first, then B: the image is contiguous
    };
    f32x16_t acc[AI][BJ];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, h = lane >> 5;
    uint32_t pa0, pb0;
    { const int r = wr * TM + frow; pa0 = r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    { const int r = wc * TN + frow; pb0 = BMT * 128 + r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    bf16x8_t fa[AI], fb[BJ];
    auto lfrag = [&](uint32_t bufoff, int kk) __attribute__((always_inline)) {
        const char* ae = smem + ((pa0 + bufoff) ^ (uint32_t)(kk << 5));
        const char* ao = smem + ((pa0 + bufoff) ^ (uint32_t)((kk << 5) ^ 64));
        const char* be = smem + ((pb0 + bufoff) ^ (uint32_t)(kk << 5));
        const char* bo = smem + ((pb0 + bufoff) ^ (uint32_t)((kk << 5) ^ 64));
#pragma unroll
        for (int i = 0; i < AI; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(((i & 1) ? ao : ae) + i * 4096);
#pragma unroll
        for (int j = 0; j < BJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(((j & 1) ? bo : be) + j * 4096);
    };
    int dk = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) ld_piece(i, dk);
    dk = min(dk + 64, K - 64);
#pragma unroll
    for (int i = 0; i < PPW; ++i) st_piece(i, 0);
#pragma unroll
    for (int i = 0; i < PPW; ++i) ld_piece(i, dk);
    dk = min(dk + 64, K - 64);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int step = 0; step < nk; ++step) {
        const uint32_t buf = (uint32_t)(step & 1) * BUF;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            lfrag(buf, kk);
#pragma unroll
            for (int i = 0; i < PPW; ++i) if (i * 4 / PPW == kk) st_piece(i, step + 1);
#pragma unroll
            for (int i = 0; i < PPW; ++i) if (i * 4 / PPW == kk) ld_piece(i, dk);
#pragma unroll
            for (int g = 0; g < AI; ++g)
#pragma unroll
                for (int j = 0; j < BJ; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[g], acc[g][j], 0, 0, 0);
        }
        dk = min(dk + 64, K - 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < BJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

// ---- hand-scheduled 4-wave variant (2 x 2 waves of 128 x 128, one wave per SIMD, accumulators in a[0:255]): the loop body is
//      generated by tools/gen_nt4w.py (tools/nt4w_asm.inc); same slab images / swizzle, LDS [A even][A odd][B even][B odd]
#include "nt4w_asm.inc"
__global__ __launch_bounds__(256) void nt4w_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                   float* __restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int ntn = N / 256;
    int t = blockIdx.x;
    const int ntiles = ntn * (M / 256);
    if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
    const int m0 = (t / ntn) * 256, n0 = (t % ntn) * 256;
    const bool isA = wid < 2;                                             // waves 0, 1 feed the A slab, waves 2, 3 the B slab
    const uint32_t ld2 = (isA ? lda : ldb) * 2u;
    const char* sbase = isA ? reinterpret_cast<const char*>(A) + (size_t)m0 * ld2 : reinterpret_cast<const char*>(B) + (size_t)n0 * ld2;
    const int lrow = lane >> 3;
    uint32_t vo[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int r = ((wid & 1) * 16 + p) * 8 + lrow;
        vo[p] = (uint32_t)r * ld2 + (uint32_t)(((lane & 7) ^ ((r ^ (r >> 3)) & 7)) << 4);
    }
    const uint32_t lwr = (isA ? 0u : 65536u) + (uint32_t)(wid & 1) * 16384u + (uint32_t)lane * 16u;
    const int frow = lane & 31, h = lane >> 5;
    uint32_t lra, lrb;
    { const int r = wr * 128 + frow; lra = r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    { const int r = wc * 128 + frow; lrb = 65536 + r * 128 + ((h ^ ((r ^ (r >> 3)) & 7)) << 4); }
    const int npair = K / 128;
    float osum;
    asm volatile(NT4W_ASM_BODY
                 : [osum] "=&v"(osum)
                 : [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), [vo4] "v"(vo[4]), [vo5] "v"(vo[5]), [vo6] "v"(vo[6]),
                   [vo7] "v"(vo[7]), [vo8] "v"(vo[8]), [vo9] "v"(vo[9]), [vo10] "v"(vo[10]), [vo11] "v"(vo[11]), [vo12] "v"(vo[12]),
                   [vo13] "v"(vo[13]), [vo14] "v"(vo[14]), [vo15] "v"(vo[15]), [lwr] "v"(lwr), [lra] "v"(lra), [lrb] "v"(lrb),
                   [sbase] "s"(sbase), [npair] "s"(npair)
                 : NT4W_ASM_CLOBBERS);
    out[(size_t)blockIdx.x * 512 + tid] = osum;
    out[(size_t)blockIdx.x * 512 + 256 + tid] = 0.f;
}

static float run4w(const bf16_t* A, const bf16_t* B, float* out, int M, int N, int K, int reps, double* sum) {
    const int lds = 131072;
    CK(hipFuncSetAttribute((const void*)nt4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = (M / 256) * (N / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) { nt4w_kernel<<<grid, 256, lds>>>(A, K, B, K, out, M, N, K); CK(hipGetLastError()); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (sum) {
        std::vector<float> h((size_t)grid * 512);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        double t = 0;
        for (float v : h) t += v;
        *sum = t;
    }
    return ms * 1e3f / reps;
}

template <int BMT, int BNT, int WR, int WC>
static float run(const bf16_t* A, const bf16_t* B, float* out, int M, int N, int K, int reps, double* sum) {
    auto kern = nt_tile_kernel<BMT, BNT, WR, WC>;
    const int lds = 2 * (BMT + BNT) * 128;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = (M / BMT) * (N / BNT);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) kern<<<grid, 512, lds>>>(A, K, B, K, out, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (sum) {
        std::vector<float> h((size_t)grid * 512);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        double t = 0;
        for (float v : h) t += v;
        *sum = t;
    }
    return ms * 1e3f / reps;
}

int main() {
    const int M = 32768;
    const int shapes[][2] = {{512, 2048}, {512, 512}, {1536, 512}, {2048, 512}, {512, 1536}};
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        srand(1);
        std::vector<double> ca(K, 0.0), cb(K, 0.0);
        for (size_t i = 0; i < hA.size(); ++i) { float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; hA[i] = (bf16_t)v; ca[i % K] += (double)(float)hA[i]; }
        for (size_t i = 0; i < hB.size(); ++i) { float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; hB[i] = (bf16_t)v; cb[i % K] += (double)(float)hB[i]; }
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += ca[k] * cb[k];
        bf16_t *A, *B; float* out;
        CK(hipMalloc(&A, hA.size() * 2 + 65536)); CK(hipMalloc(&B, hB.size() * 2 + 65536));      // (the 4-wave loop reads two slabs past the end)
        CK(hipMalloc(&out, (size_t)4096 * 512 * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        double s0, s1, s2, s3;
        run4w(A, B, out, M, N, K, 2, &s3);
        run<256, 256, 2, 4>(A, B, out, M, N, K, 2, &s0);
        run<128, 512, 1, 8>(A, B, out, M, N, K, 2, &s1);
        run<128, 512, 2, 4>(A, B, out, M, N, K, 2, &s2);
        std::vector<float> t0, t1, t2, t3;
        for (int r = 0; r < 10; ++r) {
            t0.push_back(run<256, 256, 2, 4>(A, B, out, M, N, K, 8, nullptr));
            t1.push_back(run<128, 512, 1, 8>(A, B, out, M, N, K, 8, nullptr));
            t2.push_back(run<128, 512, 2, 4>(A, B, out, M, N, K, 8, nullptr));
            t3.push_back(run4w(A, B, out, M, N, K, 8, nullptr));
        }
        std::sort(t0.begin(), t0.end()); std::sort(t1.begin(), t1.end()); std::sort(t2.begin(), t2.end()); std::sort(t3.begin(), t3.end());
        const double scale = fabs(ref) + 1e3 * sqrt((double)M * N);
        printf("N %4d K %4d: main loop only, us per launch (median of 10 x 8): 256x256 (2x4 waves) %.1f | 128x512 (1x8 waves of 128x64) %.1f | "
               "128x512 (2x4 waves of 64x128) %.1f | 256x256 hand-scheduled 4 waves of 128x128 %.1f   [sum(C) %.6g %.6g %.6g %.6g ref %.6g: %s]\n", N, K, t0[5], t1[5], t2[5], t3[5], s0, s1, s2, s3, ref,
               (fabs(s0 - ref) < 1e-3 * scale && fabs(s1 - ref) < 1e-3 * scale && fabs(s2 - ref) < 1e-3 * scale && fabs(s3 - ref) < 1e-3 * scale) ? "ok" : "MISMATCH");
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
    }
    return 0;
}
