// Diagnostic: what it costs ONE wave to move 1 KiB from L2 into LDS, per piece, (a) by LDS-DMA (global_load_lds_dwordx4) and
// (b) through registers (global_load_dwordx4 + ds_write_b128, loads issued 4 pieces ahead), alone and beside a wave of the
// same SIMD that issues v_mfma_f32_32x32x16_bf16 back to back.  256 workgroups of 8 waves (4 movers + 4 MFMA waves), every
// mover streams the same 1 MB block (L2 resident).    ./stage_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, bool MFMA>      // MODE 0: LDS-DMA, 1: registers + ds_write
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int pieces, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (wave < 4) {                                  // movers: wave w streams pieces w, w + 4, ... into a 64 KiB ring
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
            for (int p = wave; p < pieces; p += 4) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(p & 1023) * 1024 + 16 * lane),
                                                 (__attribute__((address_space(3))) void*)(smem + (p & 63) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 r[4];
            int p = wave;
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const f32x4*>(src + (size_t)((p + 4 * u) & 1023) * 1024 + 16 * lane);
            for (; p < pieces; p += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    *reinterpret_cast<f32x4*>(smem + ((p + 4 * u) & 63) * 1024 + 16 * lane) = r[u];
                    r[u] = *reinterpret_cast<const f32x4*>(src + (size_t)((p + 16 + 4 * u) & 1023) * 1024 + 16 * lane);
                }
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    } else if (MFMA) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.5f + lane * 0.001f); b[j] = (__bf16)(1.0f - lane * 0.002f); }
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < pieces / 16; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x * 512] += smem[17];
}

template <int MODE, bool MFMA>
void run(const char* name, const char* d, float* out, unsigned long long* cyc) {
    const int pieces = 4096;                         // 1024 per mover wave
    for (int w = 0; w < 3; ++w) k<MODE, MFMA><<<256, 512, 65536>>>(d, pieces, out, cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double c = 0; for (auto v : h) c += (double)v; c /= 256;
    printf("%-58s %6.1f cycles per 1 KiB piece per mover wave (%.1f B/clk per CU from 4 movers)\n", name, c / (pieces / 4), 4.0 * 1024 * (pieces / 4) / c);
}

int main() {
    char* d; float* out; unsigned long long* cyc;
    CK(hipMalloc(&d, 1 << 21)); CK(hipMemset(d, 1, 1 << 21));
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    run<0, false>("LDS-DMA, movers alone", d, out, cyc);
    run<1, false>("registers + ds_write_b128, movers alone", d, out, cyc);
    run<0, true>("LDS-DMA, beside an MFMA wave per SIMD", d, out, cyc);
    run<1, true>("registers + ds_write_b128, beside an MFMA wave per SIMD", d, out, cyc);
    return 0;
}
