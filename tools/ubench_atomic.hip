// development aid: cost of accumulating per-wave 32 x 64 f32 partial tiles into a small table with global float atomics
// (the "fold dE into the query kernel" question).  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_atomic.hip -o build_tmp/ubench_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid = BH * 8 blocks of 4 waves; block (bh, qb) does 4 qb + 4 steps; per step every wave adds one 32 x 64 f32 tile
// (8 KB: 32 floats per lane, lane-contiguous 256 B per instruction) to table block (head, eb = ebB - w + kt).
template <int MODE>   // 0: per-wave tile atomics; 1: one tile per block-step (waves pre-combined): 1/4 of the atomics; 2: plain stores (no atomics) of the same bytes
__global__ __launch_bounds__(256) void k(float* dE, int BH, int H, int spread) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int bh = blockIdx.x % BH, qb = 7 - blockIdx.x / BH;
    const int head = spread ? (bh % H) : 0;
    const int nkt = 4 * qb + 4, ebB = 28 - 4 * qb + 3;
    float v = 1.0f + lane;
    for (int kt = 0; kt < nkt; ++kt) {
        const int eb = (ebB - w + kt) & 31;
        float* dst = dE + ((size_t)head * 32 + eb) * 2048;
        if (MODE == 0 || (MODE == 1 && w == (kt & 3))) {
#pragma unroll
            for (int r = 0; r < 32; ++r) atomicAdd(dst + r * 64 + lane, v);
        } else if (MODE == 2) {
            float* d2 = dE + ((size_t)blockIdx.x * 4 + w) * 2048;
#pragma unroll
            for (int r = 0; r < 32; ++r) __builtin_nontemporal_store(v, d2 + r * 64 + lane);
        }
        v += 1.f;
        __syncthreads();
    }
}
int main() {
    const int BH = 256, H = 8;
    float* dE; CK(hipMalloc(&dE, (size_t)2048 * 4 * 2048 * 4));
    CK(hipMemset(dE, 0, (size_t)2048 * 4 * 2048 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode)
        for (int spread = 1; spread >= 0; --spread) {
            float best = 1e9;
            for (int it = 0; it < 5; ++it) {
                CK(hipEventRecord(e0));
                if (mode == 0) k<0><<<BH * 8, 256>>>(dE, BH, H, spread);
                else if (mode == 1) k<1><<<BH * 8, 256>>>(dE, BH, H, spread);
                else k<2><<<BH * 8, 256>>>(dE, BH, H, spread);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("mode %d (%s) heads %s: %.1f us\n", mode, mode == 0 ? "atomics per wave-step, 1.2 GB" : mode == 1 ? "atomics per block-step, 302 MB" : "plain stores 1.2 GB",
                   spread ? "spread (head = bh % 8)" : "all on one head", best * 1e3);
        }
    return 0;
}
