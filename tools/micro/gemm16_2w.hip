// Diagnostic: the k-step of csrc/encoder_x3.hip at ONE wave per SIMD (4 waves, 4 feature tiles each) against TWO waves per SIMD
// (8 waves, 2 feature tiles each): same MFMAs, weight bytes and fragments per CU; the partner wave issues MFMAs while a
// wave's own vector-memory / LDS instruction is being issued.   ./gemm16_2w
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NTT = 3;

template <int NF, int WAVES, bool LOADS, bool LDS, int UNR = 1>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void k(const bf16x8* __restrict__ w, int nfrag_per_wave, int ksteps,
                                                                                                    float* out, unsigned long long* cyc) {
    constexpr int F = 2 * NF, R = 4 * F;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, fw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c16 = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < 49152 / 4; i += WAVES * 64) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-3f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(w + (size_t)fw * nfrag_per_wave * 64), 0, nfrag_per_wave * 1024, 0x00020000);
    const int voff = lane * 16;
    auto ld = [&](int frag) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, frag * 1024, 0)); };
    bf16x8 ring[R];
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i] = ld(i);
    f32x4 acc[NF][NTT];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) acc[ft][tt] = f32x4{0, 0, 0, 0};
    bf16x8 xh[2][NTT], xl[2][NTT];
    auto read_x = [&](int set, int chunk) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int off = (16 * tt + c16) * 512 + (((chunk & 31) ^ c16) << 4);
            xh[set][tt] = *reinterpret_cast<const bf16x8*>(smem + off);
            xl[set][tt] = *reinterpret_cast<const bf16x8*>(smem + 24576 + off);
        }
    };
    read_x(0, q); read_x(1, 4 + q);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int pos = R;
#pragma unroll UNR
    for (int g = 0; g < ksteps; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cur = u & 1;
            if (LDS) read_x(cur ^ 1, 4 * (g + u + 1) + q);
#pragma unroll
            for (int ft = 0; ft < NF; ++ft)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const bf16x8& whi = ring[u * F + 2 * ft];
                    const bf16x8& wlo = ring[u * F + 2 * ft + 1];
                    acc[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[cur][tt], acc[ft][tt], 0, 0, 0);
                    acc[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xl[cur][tt], acc[ft][tt], 0, 0, 0);
                    acc[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh[cur][tt], acc[ft][tt], 0, 0, 0);
                }
            if (LOADS) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    int p = pos + u * F + f;
                    if (p >= nfrag_per_wave) p -= nfrag_per_wave;
                    ring[u * F + f] = ld(p);
                }
            }
#pragma unroll
            for (int i = 0; i < 9 * NF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else if (i - 6 < F) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pos += 4 * F;
        if (pos >= nfrag_per_wave) pos -= nfrag_per_wave;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) s += acc[ft][tt][0] + acc[ft][tt][1] + acc[ft][tt][2] + acc[ft][tt][3];
    if (!LOADS) { float t = 0; for (int i = 0; i < R; ++i) t += (float)ring[i][0]; s += t; }
    if (s == 1.2345f) out[blockIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int waves, const bf16x8* w, int nfrag, int ksteps, float* out, unsigned long long* cyc, int grid) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 100 * 1024, 0, w, nfrag, ksteps, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 100 * 1024, 0, w, nfrag, ksteps, out, cyc);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= grid;
    const double us = ms * 1e3 / it;
    printf("%-44s: %8.1f us, %7.0f cycles per k-step per SIMD (floor 576), %.2f GHz\n", name, us, m / ksteps, m / us * 1e-3);
}

int main() {
    bf16x8* w; float* out; unsigned long long* cyc;
    const size_t bytes = (size_t)4 * 640 * 1024;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8 * 1024));
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22 & 0x3ff) + ((i & 1) << 15));
    CK(hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice));
    run("4 waves x 4 tiles: mfma only", (k<4, 4, false, false>), 4, w, 640, 640, out, cyc, 256);
    run("4 waves x 4 tiles: mfma + lds + loads", (k<4, 4, true, true>), 4, w, 640, 640, out, cyc, 256);
    run("4 waves, body unrolled x8 (32 k-steps)", (k<4, 4, true, true, 8>), 4, w, 640, 640, out, cyc, 256);
    run("4 waves, body unrolled x20 (80 k-steps)", (k<4, 4, true, true, 20>), 4, w, 640, 640, out, cyc, 256);
    run("4 waves, body unrolled x40 (160 k-steps)", (k<4, 4, true, true, 40>), 4, w, 640, 640, out, cyc, 256);
    run("8 waves x 2 tiles: mfma only", (k<2, 8, false, false>), 8, w, 320, 640, out, cyc, 256);
    run("8 waves x 2 tiles: mfma + lds", (k<2, 8, false, true>), 8, w, 320, 640, out, cyc, 256);
    run("8 waves x 2 tiles: mfma + loads", (k<2, 8, true, false>), 8, w, 320, 640, out, cyc, 256);
    run("8 waves x 2 tiles: mfma + lds + loads", (k<2, 8, true, true>), 8, w, 320, 640, out, cyc, 256);
    return 0;
}
