// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate with nothing else in the loop.
// Decides what "100 %" means for the GEMM main loop on this part (clock/power-limited, operand-data dependent).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o build_tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NW, int LDSREADS>
__global__ __launch_bounds__(NW * 64) void mfma_kernel(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    __shared__ bf16x8 lds[LDSREADS ? 4096 : 1];
    const int tid = threadIdx.x;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 6 + i) & 4095];
    for (int j = 0; j < 2; ++j) b[j] = src[(tid * 6 + 4 + j) & 4095];
    if (LDSREADS) { for (int i = tid; i < 4096; i += NW * 64) lds[i] = src[i]; __syncthreads(); }
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) {
            // 6 ds_read_b128 per 8 MFMAs, like the GEMM phase (conflict-free: consecutive 16-byte slots)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds[(tid + 64 * i + it * 8) & 4095];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = lds[(tid + 64 * (4 + j) + it * 8) & 4095];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int NW, int LDSREADS>
void run(const char* name, const bf16x8* src, float* sink, int ncu) {
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) mfma_kernel<NW, LDSREADS><<<ncu, NW * 64>>>(src, sink, iters);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) mfma_kernel<NW, LDSREADS><<<ncu, NW * 64>>>(src, sink, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n_mfma = (double)reps * ncu * NW * iters * 8;
    const double tf = n_mfma * 32768.0 / (ms * 1e-3) / 1e12;
    const double ns_per_mfma_simd = ms * 1e6 / (n_mfma / (ncu * 4.0));
    printf("%-52s %8.1f TF/s   %6.2f ns per MFMA per SIMD (= %5.1f cyc @2.4 GHz)\n", name, tf, ns_per_mfma_simd, ns_per_mfma_simd * 2.4);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    std::vector<unsigned short> h(4096 * 8);
    bf16x8* src; float* sink;
    CK(hipMalloc(&src, 4096 * 16)); CK(hipMalloc(&sink, 64));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) { for (auto& x : h) x = 0; }
        else { unsigned s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; unsigned e = 120 + ((s >> 20) & 7); x = (unsigned short)(((s >> 31) << 15) | (e << 7) | ((s >> 8) & 127)); } }
        CK(hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice));
        printf("--- operands: %s\n", pass == 0 ? "zeros" : "random bf16 (|x| ~ 2^-7..1)");
        run<4, 0>("4 waves/CU (1 per SIMD), registers only", src, sink, ncu);
        run<8, 0>("8 waves/CU (2 per SIMD), registers only", src, sink, ncu);
        run<4, 1>("4 waves/CU, 6 ds_read_b128 per 8 MFMAs", src, sink, ncu);
        run<8, 1>("8 waves/CU, 6 ds_read_b128 per 8 MFMAs", src, sink, ncu);
    }
    return 0;
}
