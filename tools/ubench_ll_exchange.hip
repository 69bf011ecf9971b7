// Cost of an all-to-all exchange between the blocks of ONE persistent kernel when the payload carries its own validity tag
// (8-byte records {value, round}: a single-copy-atomic 64-bit store / load at agent scope, no counter, no fence, no L2
// write-back) -- the floor of a decode stage if the token ran as one launch (decode today: 26 launches x (1.7 us boundary +
// ~4 us dependent chain)).  tools/ubench_grid_barrier.hip measured the counter barrier (3.8-10.8 us at 256 blocks).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_ll_exchange.hip -o abl_tmp/ubench_ll_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void st_rec(unsigned long long* p, float v, unsigned tag) {
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_rec(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_rec_sys(unsigned long long* p, float v, unsigned tag) {
    const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_rec_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// variant 3: system-scope stores and loads;  variant 4: only wave 0 polls (RPT * 4 records per lane), the block meets at a barrier
template <int RPT>
__global__ __launch_bounds__(256) void k_ll2(unsigned long long* buf, int rounds, float* out, unsigned* err, int variant) {
    const int NV = 256 * RPT, G = gridDim.x, own = NV / G;
    float acc = 0.f;
    __shared__ float red[2][4];
    for (int i = 0; i < rounds; ++i) {
        unsigned long long* v = buf + (size_t)(i & 1) * NV;
        if ((int)threadIdx.x < own) {
            if (variant == 3) st_rec_sys(&v[blockIdx.x * own + threadIdx.x], acc * 1e-6f + (float)(i & 7), (unsigned)(i + 1));
            else st_rec(&v[blockIdx.x * own + threadIdx.x], acc * 1e-6f + (float)(i & 7), (unsigned)(i + 1));
        }
        float s = 0.f;
        if (variant == 3) {
            unsigned long long w[RPT];
            unsigned pending = (1u << RPT) - 1u;
            int spins = 0;
            while (pending) {
#pragma unroll
                for (int r = 0; r < RPT; ++r) if (pending >> r & 1) w[r] = ld_rec_sys(&v[r * 256 + threadIdx.x]);
#pragma unroll
                for (int r = 0; r < RPT; ++r) if ((pending >> r & 1) && (unsigned)(w[r] >> 32) == (unsigned)(i + 1)) { pending &= ~(1u << r); s += __uint_as_float((unsigned)w[r]); }
                if (pending && ++spins > (1 << 22)) { atomicAdd(err, 1u); pending = 0; }
            }
        } else if (threadIdx.x < 64) {
            constexpr int R4 = RPT * 4;
            unsigned long long w[R4];
            unsigned pending = R4 >= 32 ? 0xffffffffu : (1u << R4) - 1u;
            int spins = 0;
            while (pending) {
#pragma unroll
                for (int r = 0; r < R4; ++r) if (pending >> r & 1) w[r] = ld_rec(&v[r * 64 + threadIdx.x]);
#pragma unroll
                for (int r = 0; r < R4; ++r) if ((pending >> r & 1) && (unsigned)(w[r] >> 32) == (unsigned)(i + 1)) { pending &= ~(1u << r); s += __uint_as_float((unsigned)w[r]); }
                if (pending && ++spins > (1 << 22)) { atomicAdd(err, 1u); pending = 0; }
            }
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0) red[i & 1][threadIdx.x >> 6] = s;
        __syncthreads();
        acc = red[i & 1][0] + red[i & 1][1] + red[i & 1][2] + red[i & 1][3];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// NV records per round, block b owns NV / grid of them; every thread of every block reads NV / 256 records per round
template <int RPT>                                         // records per thread on the read side (NV = 256 * RPT)
__global__ __launch_bounds__(256) void k_ll(unsigned long long* buf, int rounds, float* out, unsigned* err, int sleep) {
    const int NV = 256 * RPT, G = gridDim.x, own = NV / G;          // own >= 1 (host checks)
    float acc = 0.f;
    for (int i = 0; i < rounds; ++i) {
        unsigned long long* v = buf + (size_t)(i & 1) * NV;
        if ((int)threadIdx.x < own) st_rec(&v[blockIdx.x * own + threadIdx.x], acc * 1e-6f + (float)(i & 7), (unsigned)(i + 1));
        float s = 0.f;
        unsigned long long w[RPT];
        unsigned pending = (1u << RPT) - 1u;
        int spins = 0;
        while (pending) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) if (pending >> r & 1) w[r] = ld_rec(&v[r * 256 + threadIdx.x]);
#pragma unroll
            for (int r = 0; r < RPT; ++r) if ((pending >> r & 1) && (unsigned)(w[r] >> 32) == (unsigned)(i + 1)) {
                pending &= ~(1u << r); s += __uint_as_float((unsigned)w[r]);
            }
            if (pending) { if (sleep) __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) { atomicAdd(err, 1u); pending = 0; } }
        }
        // block-wide use of the data (what a LayerNorm prologue would do): wave reduce + LDS
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        __shared__ float red[2][4];
        if ((threadIdx.x & 63) == 0) red[i & 1][threadIdx.x >> 6] = s;
        __syncthreads();
        acc = red[i & 1][0] + red[i & 1][1] + red[i & 1][2] + red[i & 1][3];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// flags + payload: producers store the payload with agent-scope stores, wait for their acknowledgement (vmcnt 0) and publish
// one flag per block; consumers poll the G flags, then read the payload (two dependent round trips)
template <int RPT>
__global__ __launch_bounds__(256) void k_flag(float* buf, unsigned* flags, int rounds, float* out, unsigned* err) {
    const int NV = 256 * RPT, G = gridDim.x, own = NV / G;
    float acc = 0.f;
    for (int i = 0; i < rounds; ++i) {
        float* v = buf + (size_t)(i & 1) * NV;
        if ((int)threadIdx.x < own) __hip_atomic_store(&v[blockIdx.x * own + threadIdx.x], acc * 1e-6f + (float)(i & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                      // stores acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        for (int f = threadIdx.x; f < G; f += 256)
            while (__hip_atomic_load(&flags[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(i + 1)) if (++spins > (1 << 22)) { atomicAdd(err, 1u); break; }
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RPT; ++r) s += __hip_atomic_load(&v[r * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        __shared__ float red[2][4];
        if ((threadIdx.x & 63) == 0) red[i & 1][threadIdx.x >> 6] = s;
        __syncthreads();
        acc = red[i & 1][0] + red[i & 1][1] + red[i & 1][2] + red[i & 1][3];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <int RPT>
static int run(int grid, hipStream_t st, unsigned long long* buf, float* out, unsigned* err) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 500;
    for (int variant = 0; variant < 5; ++variant) {
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(buf, 0, 1 << 20, st)); CK(hipMemsetAsync(err, 0, 4, st));
            CK(hipEventRecord(e0, st));
            if (variant < 2) k_ll<RPT><<<grid, 256, 0, st>>>(buf, R, out, err, variant);
            else if (variant >= 3) k_ll2<RPT><<<grid, 256, 0, st>>>(buf, R, out, err, variant);
            else k_flag<RPT><<<grid, 256, 0, st>>>((float*)buf, (unsigned*)(buf + 65536), R, out, err);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        unsigned h_err = 0; float h0 = 0, h1 = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h1, out + grid - 1, 4, hipMemcpyDeviceToHost));
        printf("grid %3d  %5d values  %-28s %.3f us per exchange (median of 7 x %d; min %.3f)  timeouts %u  acc %g %g\n", grid, 256 * RPT,
               variant == 0 ? "tagged records" : variant == 1 ? "tagged records + s_sleep" : variant == 2 ? "flags then payload" : variant == 3 ? "tagged, system scope" : "tagged, wave 0 polls", t[3] * 1e3 / R, R, t[0] * 1e3 / R, h_err, h0, h1);
    }
    return 0;
}

int main() {
    unsigned long long* buf; float* out; unsigned* err;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&err, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int grid : {32, 64, 128, 256}) {
        if (run<2>(grid, st, buf, out, err)) return 1;      //  512 values (a [4][128] slice)
        if (run<8>(grid, st, buf, out, err)) return 1;      // 2048 values ([4][512] residual rows)
    }
    return 0;
}
