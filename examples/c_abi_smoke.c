/* Plain C consumer of libmidiemo_hip.so: no Python, no torch -- only the HIP runtime for device memory.
 * Builds a tiny bf16 problem, runs me_cast_transpose -> me_gemm_nt (bias + ReLU epilogue) -> me_sumsq and a weight
 * gradient (me_gemm_tn_acc with a caller-owned workspace sized by me_workspace_bytes) on the default stream and
 * checks the results against host loops.  This is what a non-Python host (the reference is
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
    /* caller-owned, zeroed scratch: block sums are added in a fixed order (bit-reproducible result) */
    void* dsw;
    CK(hipMalloc(&dsw, me_workspace_bytes(ME_WS_SUMSQ, 0, 0, 0, ME_F32)));
    CK(hipMemset(dsw, 0, me_workspace_bytes(ME_WS_SUMSQ, 0, 0, 0, ME_F32)));
    ME(me_sumsq(dA32, (int64_t)M * K, dss, dsw, me_workspace_bytes(ME_WS_SUMSQ, 0, 0, 0, ME_F32), NULL));
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
    /* weight gradient dW[N2][K2] += dY[T][N2]^T . X[T][K2] through the caller-owned partial-tile workspace */
    {
        enum { T = 4096, N2 = 256, K2 = 256 };
        float *hY = malloc(sizeof(float) * T * N2), *hX = malloc(sizeof(float) * T * K2), *hG = malloc(sizeof(float) * N2 * K2);
        for (int i = 0; i < T * N2; ++i) { s = s * 1664525u + 1013904223u; hY[i] = ((int)(s >> 20) % 16 - 8) / 16.f; }
        for (int i = 0; i < T * K2; ++i) { s = s * 1664525u + 1013904223u; hX[i] = ((int)(s >> 20) % 16 - 8) / 16.f; }
        float *dY32, *dX32, *dG;
        void *dY, *dX, *dws = NULL;
        CK(hipMalloc((void**)&dY32, sizeof(float) * T * N2)); CK(hipMalloc((void**)&dX32, sizeof(float) * T * K2));
        CK(hipMalloc((void**)&dG, sizeof(float) * N2 * K2)); CK(hipMalloc(&dY, 2 * T * N2)); CK(hipMalloc(&dX, 2 * T * K2));
        CK(hipMemcpy(dY32, hY, sizeof(float) * T * N2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dX32, hX, sizeof(float) * T * K2, hipMemcpyHostToDevice));
        CK(hipMemset(dG, 0, sizeof(float) * N2 * K2));
        ME(me_cast_transpose(dY32, T, N2, dY, N2, NULL, 0, ME_BF16, NULL));
        ME(me_cast_transpose(dX32, T, K2, dX, K2, NULL, 0, ME_BF16, NULL));
        const size_t wsb = me_workspace_bytes(ME_WS_GEMM_TN, T, N2, K2, ME_BF16);
        if (wsb == 0) { printf("expected a workspace requirement for the 256-tile TN kernel\n"); return 7; }
        CK(hipMalloc(&dws, wsb));
        if (me_gemm_tn_acc(dY, N2, dX, K2, dG, K2, NULL, T, N2, K2, dws, wsb / 2, ME_BF16, NULL) != ME_ERR_WORKSPACE) {
            printf("short workspace accepted\n"); return 8;
        }
        ME(me_gemm_tn_acc(dY, N2, dX, K2, dG, K2, NULL, T, N2, K2, dws, wsb, ME_BF16, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hG, dG, sizeof(float) * N2 * K2, hipMemcpyDeviceToHost));
        double w2 = 0.0;
        for (int n = 0; n < N2; n += 17)
            for (int k = 0; k < K2; k += 13) {
                double acc = 0.0;
                for (int t = 0; t < T; ++t) acc += (double)hY[t * N2 + n] * hX[t * K2 + k];
                const double err = fabs(hG[n * K2 + k] - acc) / (fabs(acc) + 1.0);
                if (err > w2) w2 = err;
            }
        printf("c_abi_smoke: gemm_tn_acc (workspace %zu bytes) worst rel err %.3e\n", wsb, w2);
        if (w2 > 1e-4) { printf("FAILED\n"); return 9; }
    }
    /* error behaviour: bad arguments are reported, not crashed on */
    if (me_gemm_nt(NULL, K, dW, K, dC, N, db, NULL, 0, NULL, 0, M, N, K, 0, ME_BF16, NULL) == ME_OK) { printf("NULL accepted\n"); return 5; }
    if (me_rga_fwd(dA, dW, NULL, dC, dss, NULL, NULL, 1, 64, 2, 40, 2048, 1, ME_BF16, NULL) == ME_OK) { printf("dh = 40 accepted\n"); return 6; }
    printf("OK\n");
    return 0;
}
