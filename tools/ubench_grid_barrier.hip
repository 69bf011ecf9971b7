// Cost of a grid-wide barrier inside one persistent kernel against the kernel boundary of a graph-replayed launch chain
// (decode: 26 dependent ~5 us launches per token; would one persistent launch with grid barriers be cheaper?).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_grid_barrier.hip -o abl_tmp/ubench_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// flat barrier: one monotonically increasing counter; arrive = atomic add (device scope), wait = spin on a device-scope load
__device__ __forceinline__ void grid_barrier_flat(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);                 // agent scope by default for global atomics
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// floor: relaxed atomics (no L2 write-back / invalidate around them: the data would have to travel with sc1 stores / loads)
__device__ __forceinline__ void grid_barrier_relaxed(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED);
        while (__atomic_load_n(ctr, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_relaxed(unsigned* ctr, int nbar, float* data) {
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        if (threadIdx.x == 0) __builtin_nontemporal_store(acc + i, &data[blockIdx.x]);
        grid_barrier_relaxed(ctr, gridDim.x * (unsigned)(i + 1));
        acc += __builtin_nontemporal_load(&data[(blockIdx.x + 1) % gridDim.x]);
    }
    if (threadIdx.x == 0) data[gridDim.x + blockIdx.x] = acc;
}
// two-level: blocks of a group (blockIdx % G) arrive on their group's counter; the last arriver of a group arrives on the root;
// everybody spins on the root's generation word
__device__ __forceinline__ void grid_barrier_tree(unsigned* grp, unsigned* root, unsigned* gen, int G, unsigned per_group, unsigned it) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x % G;
        const unsigned old = __atomic_fetch_add(&grp[g * 32], 1u, __ATOMIC_ACQ_REL);
        if (old + 1 == per_group * (it + 1)) {
            const unsigned r = __atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL);
            if (r + 1 == (unsigned)G * (it + 1)) __atomic_store_n(gen, it + 1, __ATOMIC_RELEASE);
        }
        while (__atomic_load_n(gen, __ATOMIC_ACQUIRE) < it + 1) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_flat(unsigned* ctr, int nbar, float* data) {
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        // a token of real traffic: every block writes a value the others read after the barrier
        if (threadIdx.x == 0) data[blockIdx.x] = acc + i;
        grid_barrier_flat(ctr, gridDim.x * (unsigned)(i + 1));
        acc += __builtin_nontemporal_load(&data[(blockIdx.x + 1) % gridDim.x]);
    }
    if (threadIdx.x == 0) data[gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_tree(unsigned* grp, unsigned* root, unsigned* gen, int G, int nbar, float* data) {
    float acc = 0.f;
    const unsigned per_group = gridDim.x / G;
    for (int i = 0; i < nbar; ++i) {
        if (threadIdx.x == 0) data[blockIdx.x] = acc + i;
        grid_barrier_tree(grp, root, gen, G, per_group, (unsigned)i);
        acc += __builtin_nontemporal_load(&data[(blockIdx.x + 1) % gridDim.x]);
    }
    if (threadIdx.x == 0) data[gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_stage(float* data, int i) {
    float acc = __builtin_nontemporal_load(&data[(blockIdx.x + 1) % gridDim.x]);
    if (threadIdx.x == 0) data[gridDim.x + blockIdx.x] = acc + i;
}
int main() {
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned* ctr; float* data;
    CK(hipMalloc(&ctr, 1 << 16)); CK(hipMalloc(&data, 1 << 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NB = 520;
    for (int grid : {64, 128, 256}) {
        if (grid > ncu) continue;
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 1 << 16, st));
            CK(hipEventRecord(e0, st));
            k_flat<<<grid, 256, 0, st>>>(ctr, NB, data);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("grid %3d flat barrier: %.3f us per barrier (median of 7 x %d)\n", grid, t[3] * 1e3 / NB, NB);
        t.clear();
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 1 << 16, st));
            CK(hipEventRecord(e0, st));
            k_relaxed<<<grid, 256, 0, st>>>(ctr, NB, data);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("grid %3d flat barrier, relaxed atomics (floor): %.3f us per barrier\n", grid, t[3] * 1e3 / NB);
        for (int G : {8, 16}) {
            t.clear();
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 1 << 16, st));
                CK(hipEventRecord(e0, st));
                k_tree<<<grid, 256, 0, st>>>(ctr, ctr + 4096, ctr + 8192, G, NB, data);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("grid %3d tree barrier (G = %2d): %.3f us per barrier\n", grid, G, t[3] * 1e3 / NB);
        }
        // graph of NB dependent launches
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < NB; ++i) k_stage<<<grid, 256, 0, st>>>(data, i);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        t.clear();
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("grid %3d graph of %d dependent launches: %.3f us per launch\n", grid, NB, t[3] * 1e3 / NB);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
