// Microbenchmark (diagnostic, not product): how fast can every CU stream the SAME L2-resident block (one layer's packed
// weights, 2.6 MB) into registers / LDS?  Sets the weight-stream floor of attn_apply at one workgroup per CU.
//   ./l2_stream [bytes=2621440] [iters=50]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each wave streams its contiguous share in 1 KiB wave-loads, D loads in flight
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_vgpr(const u32x4* __restrict__ w, size_t n16, unsigned* out) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = n16 / WAVES;                 // 16-byte elements per wave
    const u32x4* p = w + (size_t)wave * per_wave + lane;
    u32x4 acc = {0, 0, 0, 0};
    const size_t steps = per_wave / 64 / D;
    for (size_t s = 0; s < steps; ++s) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = p[(s * D + d) * 64];
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
    if (threadIdx.x == 0 && smem[0] == 77) out[0] = 2;
}

// interleaved: consecutive 1 KiB pieces go round-robin over the waves (all waves walk the block together)
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_vgpr_rr(const u32x4* __restrict__ w, size_t n16, unsigned* out) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* p = w + (size_t)wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    const size_t steps = n16 / 64 / WAVES / D;
    for (size_t s = 0; s < steps; ++s) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = p[(s * D + d) * 64 * WAVES];
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
    if (threadIdx.x == 0 && smem[0] == 77) out[0] = 2;
}

// LDS-DMA: global_load_lds_dwordx4 into a ring of D x 1 KiB slots per wave (never read back)
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_ldsdma(const u32x4* __restrict__ w, size_t n16, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t per_wave = n16 / WAVES;
    const u32x4* p = w + (size_t)wave * per_wave + lane;
    char* ring = smem + wave * D * 1024;
    const size_t steps = per_wave / 64 / D;
    for (size_t s = 0; s < steps; ++s) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (s * D + d) * 64),
                                             (__attribute__((address_space(3))) void*)(ring + d * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[threadIdx.x] == 77 && smem[threadIdx.x + 1] == 78 && smem[threadIdx.x + 2] == 79) out[blockIdx.x] = 1;
}

template <typename K>
double time_kernel(K kern, int waves, size_t lds, const u32x4* w, size_t n16, unsigned* out, int grid, int iters) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, w, n16, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, w, n16, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3 / iters;          // us per launch
}

int main(int argc, char** argv) {
    size_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 2621440;
    int iters = argc > 2 ? atoi(argv[2]) : 50;
    bytes = bytes / (16 * 64 * 8 * 32) * (16 * 64 * 8 * 32);
    const size_t n16 = bytes / 16;
    u32x4* w; unsigned* out;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 4096 * 4));
    std::vector<unsigned> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice));
    const size_t lds1 = 100 * 1024;       // > 80 KiB: one workgroup per CU
    printf("block %zu bytes, %d iterations; us per launch and GB/s per CU (one workgroup per CU unless noted)\n", bytes, iters);
#define RUN(NAME, KERN, WAVES, GRID, LDS)                                                                   \
    { double us = time_kernel(KERN, WAVES, LDS, w, n16, out, GRID, iters);                                  \
      printf("%-44s grid %4d: %8.2f us  %7.1f GB/s per WG  %6.2f TB/s chip\n", NAME, GRID, us, bytes / us * 1e-3, bytes / us * 1e-6 * GRID); }
    RUN("vgpr contiguous D=8  4 waves", (stream_vgpr<8, 4>), 4, 256, lds1);
    RUN("vgpr contiguous D=16 4 waves", (stream_vgpr<16, 4>), 4, 256, lds1);
    RUN("vgpr contiguous D=32 4 waves", (stream_vgpr<32, 4>), 4, 256, lds1);
    RUN("vgpr contiguous D=8  8 waves", (stream_vgpr<8, 8>), 8, 256, lds1);
    RUN("vgpr contiguous D=16 8 waves", (stream_vgpr<16, 8>), 8, 256, lds1);
    RUN("vgpr contiguous D=16 16 waves", (stream_vgpr<16, 16>), 16, 256, lds1);
    RUN("vgpr round-robin D=8  4 waves", (stream_vgpr_rr<8, 4>), 4, 256, lds1);
    RUN("vgpr round-robin D=16 4 waves", (stream_vgpr_rr<16, 4>), 4, 256, lds1);
    RUN("vgpr round-robin D=32 4 waves", (stream_vgpr_rr<32, 4>), 4, 256, lds1);
    RUN("vgpr round-robin D=16 8 waves", (stream_vgpr_rr<16, 8>), 8, 256, lds1);
    RUN("lds-dma D=8  4 waves", (stream_ldsdma<8, 4>), 4, 256, lds1);
    RUN("lds-dma D=16 4 waves", (stream_ldsdma<16, 4>), 4, 256, lds1);
    RUN("lds-dma D=16 8 waves", (stream_ldsdma<16, 8>), 8, 256, 8 * 16 * 1024 + 1024);
    RUN("vgpr contiguous D=16 4 waves, 128 WGs", (stream_vgpr<16, 4>), 4, 128, lds1);
    RUN("vgpr contiguous D=16 4 waves, 64 WGs", (stream_vgpr<16, 4>), 4, 64, lds1);
    RUN("vgpr contiguous D=16 4 waves, 8 WGs", (stream_vgpr<16, 4>), 4, 8, lds1);
    RUN("vgpr contiguous D=16 4 waves, 2 WG/CU", (stream_vgpr<16, 4>), 4, 512, 70 * 1024);
    return 0;
}
