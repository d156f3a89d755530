// Diagnostic: cycles per v_mfma_f32_32x32x16_bf16 issued by ONE wave (operands in registers, NA independent accumulators,
// random data), with 1 or 2 waves per SIMD, timed by s_memtime and by the wall clock (=> shader clock under this load).
//   ./mfma32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NA, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const bf16x8* __restrict__ src, int iters, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[((4 + i) * 64 + lane)]; }
    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12 / NA; ++rep)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(rep + i) & 3], b[(rep * 3 + i) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NA, int THREADS>
void run(const char* name, const bf16x8* d, float* out, unsigned long long* cyc, int grid) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) k<NA, THREADS><<<grid, THREADS>>>(d, iters, out, cyc);
    CK(hipEventRecord(e0));
    k<NA, THREADS><<<grid, THREADS>>>(d, iters, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double c = 0;
    for (auto v : h) c += (double)v;
    c /= grid;
    const double per = c / (iters * 12.0);
    printf("%-44s %6.1f counts per MFMA per wave, kernel %7.1f us, counter rate %.2f GHz, %.0f TFLOP/s\n", name, per, ms * 1e3, c / (ms * 1e-3) * 1e-9,
           (double)grid * (THREADS / 64) * iters * 12.0 * 32 * 32 * 16 * 2 / (ms * 1e-3) * 1e-12);
}

int main() {
    std::vector<unsigned short> h(8 * 64 * 8);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));      // random bf16 around +-1
    bf16x8* d; float* out; unsigned long long* cyc;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    run<4, 256>("1 wave/SIMD, 4 accumulators, 256 WGs", d, out, cyc, 256);
    run<2, 256>("1 wave/SIMD, 2 accumulators, 256 WGs", d, out, cyc, 256);
    run<1, 256>("1 wave/SIMD, 1 accumulator,  256 WGs", d, out, cyc, 256);
    run<4, 512>("2 waves/SIMD, 4 accumulators, 256 WGs", d, out, cyc, 256);
    run<4, 256>("1 wave/SIMD, 4 accumulators, 8 WGs (idle chip)", d, out, cyc, 8);
    run<4, 512>("2 waves/SIMD, 4 accumulators, 8 WGs (idle chip)", d, out, cyc, 8);
    return 0;
}
