/* Plain C consumer of libmidiemo_hip.so: no Python, no torch -- only the HIP runtime for device memory.
 * Builds a tiny bf16 problem, runs me_cast_transpose -> me_gemm_nt (bias + ReLU epilogue) -> me_sumsq on the
 * default stream and checks the results against a host loop.  This is what a non-Python host (the reference is
 * Python; a C / C++ / Go-cgo / JNI host would look the same) has to do to use the library.
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_smoke.c -Iinclude -I/opt/rocm/include -Lmidi-emotion_amd/midiemo \
 *       -L/opt/rocm/lib -lmidiemo_hip -lamdhip64 -lm -o c_abi_smoke        (tests/test_c_abi_example.py does this) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "midiemo.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define ME(x) do { int rc_ = (x); if (rc_ != ME_OK) { printf("midiemo error %d at line %d\n", rc_, __LINE__); return 3; } } while (0)

static float bf16_to_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(void) {
    enum { M = 512, N = 256, K = 128 };
    if (me_abi_version() != ME_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
    float *hA = malloc(sizeof(float) * M * K), *hW = malloc(sizeof(float) * N * K), *hb = malloc(sizeof(float) * N);
    uint32_t s = 12345u;
    for (int i = 0; i < M * K; ++i) { s = s * 1664525u + 1013904223u; hA[i] = ((int)(s >> 20) % 64 - 32) / 64.f; }
    for (int i = 0; i < N * K; ++i) { s = s * 1664525u + 1013904223u; hW[i] = ((int)(s >> 20) % 64 - 32) / 64.f; }   /* exact in bf16 */
    for (int i = 0; i < N; ++i) hb[i] = 0.25f * (i % 5) - 0.5f;
    float *dA32, *dW32, *db, *dss;
    void *dA, *dW, *dC;
    CK(hipMalloc((void**)&dA32, sizeof(float) * M * K)); CK(hipMalloc((void**)&dW32, sizeof(float) * N * K));
    CK(hipMalloc((void**)&db, sizeof(float) * N)); CK(hipMalloc((void**)&dss, sizeof(float)));
    CK(hipMalloc(&dA, 2 * M * K)); CK(hipMalloc(&dW, 2 * N * K)); CK(hipMalloc(&dC, 2 * M * N));
    CK(hipMemcpy(dA32, hA, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW32, hW, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb, sizeof(float) * N, hipMemcpyHostToDevice));
    CK(hipMemset(dss, 0, sizeof(float)));
    /* f32 masters -> bf16 copies (no transposed copy wanted: dstT = NULL) */
    ME(me_cast_transpose(dA32, M, K, dA, K, NULL, 0, ME_BF16, NULL));
    ME(me_cast_transpose(dW32, N, K, dW, K, NULL, 0, ME_BF16, NULL));
    /* C = relu(A . W^T + b), bf16 out */
    ME(me_gemm_nt(dA, K, dW, K, dC, N, db, NULL, 0, NULL, 0, M, N, K, ME_EPI_RELU, ME_BF16, NULL));
    ME(me_sumsq(dA32, (int64_t)M * K, dss, NULL));
    CK(hipDeviceSynchronize());
    uint16_t* hC = malloc(2 * M * N);
    float ss = 0.f;
    CK(hipMemcpy(hC, dC, 2 * M * N, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&ss, dss, sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0, ref_ss = 0.0;
    for (int i = 0; i < M * K; ++i) ref_ss += (double)hA[i] * hA[i];
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += (double)hA[m * K + k] * hW[n * K + k];
            if (acc < 0) acc = 0;
            const double got = bf16_to_f(hC[m * N + n]);
            const double err = fabs(got - acc) / (fabs(acc) + 1.0);
            if (err > worst) worst = err;
        }
    printf("c_abi_smoke: gemm_nt worst rel err %.3e (bf16 output rounding), sumsq %.4f vs %.4f\n", worst, ss, ref_ss);
    if (worst > 5e-3 || fabs(ss - ref_ss) > 1e-3 * ref_ss) { printf("FAILED\n"); return 4; }
    /* error behaviour: bad arguments are reported, not crashed on */
    if (me_gemm_nt(NULL, K, dW, K, dC, N, db, NULL, 0, NULL, 0, M, N, K, 0, ME_BF16, NULL) == ME_OK) { printf("NULL accepted\n"); return 5; }
    if (me_rga_fwd(dA, dW, NULL, dC, dss, 1, 64, 2, 40, 2048, 1, ME_BF16, NULL) == ME_OK) { printf("dh = 40 accepted\n"); return 6; }
    printf("OK\n");
    return 0;
}
