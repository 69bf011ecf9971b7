// Microbenchmark: how fast can one CU pull a GEMM-shaped operand stream (128-byte row segments at a
// 1 KiB row stride, 64 KiB per slab) out of L2 / HBM?  Decides the operand path of the NT GEMM.
//   mode 0: global_load_lds b128 (LDS-DMA), vmcnt(0)+barrier per slab     (what gemm_nt256 does)
//   mode 1: LDS-DMA, two slabs in flight (wait only for the older one)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128, one slab in flight in registers
//   mode 3: global_load_dwordx4 -> VGPR only (xor-reduced), no LDS
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_ingest.hip -o gpurun_out/ubench_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void ingest_kernel(const char* __restrict__ src, size_t region_rows, int nslabs,
                                                         unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];            // 2 x 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PIECES = 64 / NW;                                          // 1 KiB pieces per wave per slab
    const int lrow = lane >> 3, lch = lane & 7;
    u32x4 acc = {0, 0, 0, 0};
    // slab s of this block: 512 rows (A 256 + B 256) x 128 B; row-tile walks through the region, 6 blocks share a tile
    auto src_of = [&](int s, int piece) -> const char* {
        if (MODE >= 4) {
            // ffn2-like: A [32768 x 2048] bf16 streamed (rows of 4 KiB, k-slab s & 31), each 256-row panel read by the
            // blocks b and b + 8 (same XCD); B [512 x 2048] L2 resident, n-tile (b >> 3) & 1.  First half of the
            // pieces are A rows, second half B rows.
            const int p = wid * PIECES + piece;                              // 0..63, 8 rows each
            const size_t k_off = (size_t)(s & 31) * 128 + (lch ^ lrow) * 16;
            if (p < 32) {
                const size_t panel = ((size_t)(s >> 5) * (gridDim.x / 2) + (blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) % 128;
                return src + (panel * 256 + (size_t)p * 8 + lrow) * 4096 + k_off;
            }
            return src + ((size_t)32768 + ((blockIdx.x >> 3) & 1) * 256 + (size_t)(p - 32) * 8 + lrow) * 4096 + k_off;
        }
        const size_t tile = ((size_t)(s >> 3) * gridDim.x + blockIdx.x) / 6;
        const size_t row = (tile * 512 + (size_t)(wid * PIECES + piece) * 8 + lrow) % region_rows;
        return src + row * 1024 + (size_t)(s & 7) * 128 + (lch ^ lrow) * 16;
    };
    if constexpr (MODE == 0 || MODE == 1 || MODE == 5) {
        auto issue = [&](int s) {
            char* base = smem + (s & 1) * 65536 + wid * PIECES * 1024;
#pragma unroll
            for (int i = 0; i < PIECES; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_of(s, i),
                                                 (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
        };
        issue(0);
        if (MODE == 1) issue(1);
        for (int s = 0; s < nslabs; ++s) {
            if (MODE == 0 || MODE == 5) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (s + 1 < nslabs) issue(s + 1);
            } else {
                if (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                __syncthreads();
                // consume a little so the buffer is really read before being refilled
                acc ^= *reinterpret_cast<const u32x4*>(smem + (s & 1) * 65536 + tid * 16);
                __syncthreads();
                if (s + 2 < nslabs) issue(s + 2);
                else issue(s);                                     // keep the count of outstanding loads uniform
            }
            if (MODE == 0 || MODE == 5) acc ^= *reinterpret_cast<const u32x4*>(smem + (s & 1) * 65536 + tid * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        u32x4 r[PIECES];
        f32x16 macc[8];
        bf16x8 ma, mb;
        if (MODE == 6 || MODE == 7) {
            for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) macc[i][e] = 0.f;
            unsigned sd = tid * 2654435761u + 12345u;
            for (int e = 0; e < 8; ++e) { sd = sd * 1664525u + 1013904223u; ma[e] = (__bf16)(((int)(sd >> 16) & 255) / 256.f - 0.5f); mb[e] = (__bf16)(((int)(sd >> 8) & 255) / 256.f - 0.5f); }
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) r[i] = *reinterpret_cast<const u32x4*>(src_of(0, i));
        for (int s = 0; s < nslabs; ++s) {
            if (MODE == 2 || MODE == 4 || MODE == 6) {
                char* base = smem + (s & 1) * 65536 + wid * PIECES * 1024 + lane * 16;
#pragma unroll
                for (int i = 0; i < PIECES; ++i) *reinterpret_cast<u32x4*>(base + i * 1024) = r[i];
            } else {
#pragma unroll
                for (int i = 0; i < PIECES; ++i) acc ^= r[i];
            }
            const int sn = s + 1 < nslabs ? s + 1 : s;
            if (MODE != 7) {
#pragma unroll
                for (int i = 0; i < PIECES; ++i) r[i] = *reinterpret_cast<const u32x4*>(src_of(sn, i));
            }
            if (MODE == 6 || MODE == 7) {
#pragma unroll
                for (int q = 0; q < 32 * 8 / NW; ++q) macc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, macc[q & 7], 0, 0, 0);
            }
            if (MODE == 2 || MODE == 4 || MODE == 6) {
                __syncthreads();
                acc ^= *reinterpret_cast<const u32x4*>(smem + (s & 1) * 65536 + tid * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < PIECES; ++i) acc ^= r[i];
        if (MODE == 6 || MODE == 7) { float t = 0.f; for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) t += macc[i][e]; if (t == 1.2345f) acc.x ^= 1; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345677u) sink[0] = 1;
}

template <int MODE, int NW>
void run(const char* name, const char* src, size_t region_rows, int nslabs, unsigned* sink, int ncu) {
    CK(hipFuncSetAttribute((const void*)ingest_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) ingest_kernel<MODE, NW><<<ncu, NW * 64, 131072>>>(src, region_rows, nslabs, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) ingest_kernel<MODE, NW><<<ncu, NW * 64, 131072>>>(src, region_rows, nslabs, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)reps * ncu * nslabs * 65536.0;
    const double gbs_cu = bytes / (ms * 1e-3) / ncu / 1e9;
    printf("  [%.1f us per launch] ", ms * 1e3 / reps);
    printf("%-44s region %6.1f MB  %7.1f GB/s/CU  %5.1f B/clk/CU @2.4GHz  aggregate %6.2f TB/s\n", name,
           region_rows * 1024.0 / 1e6, gbs_cu, gbs_cu / 2.4, gbs_cu * ncu / 1e3);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, ncu);
    const size_t max_rows = 256 * 1024;                                      // 256 MiB
    char* src; unsigned* sink;
    CK(hipMalloc(&src, max_rows * 1024)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, max_rows * 1024));
    const int nslabs = 512;
    for (size_t rows : {(size_t)2048, (size_t)32768, max_rows}) {
        run<0, 8>("LDS-DMA, barrier per slab, 8 waves", src, rows, nslabs, sink, ncu);
        run<1, 8>("LDS-DMA, 2 slabs in flight, 8 waves", src, rows, nslabs, sink, ncu);
        run<1, 4>("LDS-DMA, 2 slabs in flight, 4 waves", src, rows, nslabs, sink, ncu);
        run<2, 8>("dwordx4->VGPR->ds_write_b128, 8 waves", src, rows, nslabs, sink, ncu);
        run<2, 4>("dwordx4->VGPR->ds_write_b128, 4 waves", src, rows, nslabs, sink, ncu);
        run<3, 8>("dwordx4->VGPR only, 8 waves", src, rows, nslabs, sink, ncu);
        run<3, 4>("dwordx4->VGPR only, 4 waves", src, rows, nslabs, sink, ncu);
    }
    printf("--- ffn2-like mix: A 134 MB streamed (2 CUs of an XCD share a panel), B 2 MB L2-resident; 1 tile x 32 slabs per launch\n");
    run<5, 8>("mix: LDS-DMA, barrier per slab, 8 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<4, 8>("mix: dwordx4->VGPR->ds_write_b128, 8 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<4, 4>("mix: dwordx4->VGPR->ds_write_b128, 4 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<7, 8>("MFMA only (256 per slab per CU), 8 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<6, 8>("mix: VGPR path + 256 MFMA per slab, 8 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<7, 4>("MFMA only (256 per slab per CU), 4 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<6, 4>("mix: VGPR path + 256 MFMA per slab, 4 waves", src, (size_t)33280 * 4, 32, sink, ncu);
    run<5, 8>("mix x4 slabs: LDS-DMA, 8 waves", src, (size_t)33280 * 4, 128, sink, ncu);
    run<4, 8>("mix x4 slabs: VGPR path, 8 waves", src, (size_t)33280 * 4, 128, sink, ncu);
    return 0;
}
