// Diagnostic: what one CU sustains on the k-step of csrc/encoder_x3.hip (36 x v_mfma_f32_16x16x32_bf16 + 6 ds_read_b128 +
// 8 x 1 KiB weight loads per wave, 4 waves) with parts switched off.  256 workgroups, one per CU, all streaming the same
// 2.5 MB block.   ./gemm16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NTT = 3, R = 32, NF = 4, F = 8;
#ifdef ASM_MFMA
__device__ __forceinline__ f32x4 MF16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
#else
__device__ __forceinline__ f32x4 MF16(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif

template <bool LOADS, bool LDS, int MF /* mfma per (ft, tt): 3 = split, 1 = hi only, 0 = none */, int WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const bf16x8* __restrict__ w, int nfrag_per_wave, int ksteps,
                                                                                    float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, fw = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < 49152 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-3f;
    __syncthreads();
    const bf16x8* wp = w + (size_t)fw * nfrag_per_wave * 64 + lane;
    bf16x8 ring[R];
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i] = wp[(size_t)i * 64];
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
    for (int g = 0; g < ksteps; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cur = u & 1;
            if (LDS) read_x(cur ^ 1, 4 * (g + u + 1) + q);
            if (WIDE != 2) {
#pragma unroll
            for (int ft = 0; ft < NF; ++ft)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const bf16x8& whi = ring[u * F + 2 * ft];
                    const bf16x8& wlo = ring[u * F + 2 * ft + 1];
                    if (MF == 3) {
                        acc[ft][tt] = MF16(wlo, xh[cur][tt], acc[ft][tt]);
                        acc[ft][tt] = MF16(whi, xl[cur][tt], acc[ft][tt]);
                    }
                    if (MF >= 1) acc[ft][tt] = MF16(whi, xh[cur][tt], acc[ft][tt]);
                }
            } else {
                // source order IS issue order: term-major MFMAs (dependent ones 12 apart), one memory op behind each of the first 14;
                // the refills go into the slots the PREVIOUS k-step consumed (no write-after-read wait inside this k-step)
                const int up = (u + 3) & 3;
#pragma unroll
                for (int i = 0; i < 36; ++i) {
                    const int term = i / 12, ft = (i % 12) / NTT, tt = i % NTT;
                    const bf16x8& whi = ring[u * F + 2 * ft];
                    const bf16x8& wlo = ring[u * F + 2 * ft + 1];
                    acc[ft][tt] = term == 0 ? MF16(wlo, xh[cur][tt], acc[ft][tt]) : term == 1 ? MF16(whi, xl[cur][tt], acc[ft][tt]) : MF16(whi, xh[cur][tt], acc[ft][tt]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (LDS && i < 6) {
                        const int tt2 = i >> 1;
                        const int off = (16 * tt2 + c16) * 512 + ((((4 * (g + u + 1) + q) & 31) ^ c16) << 4);
                        if (i & 1) xl[cur ^ 1][tt2] = *reinterpret_cast<const bf16x8*>(smem + 24576 + off);
                        else xh[cur ^ 1][tt2] = *reinterpret_cast<const bf16x8*>(smem + off);
                    }
                    if (LOADS && i >= 6 && i < 14) {
                        const int f = i - 6;
                        int p = pos + (u - 1) * F + f;                // the slot of the previous k-step gets the fragment R ahead of it
                        if (p >= nfrag_per_wave) p -= nfrag_per_wave;
                        if (p < 0) p += nfrag_per_wave;
                        ring[up * F + f] = wp[(size_t)p * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (WIDE != 2) {
            if (LOADS) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    int p = pos + u * F + f;
                    if (p >= nfrag_per_wave) p -= nfrag_per_wave;
                    ring[u * F + f] = wp[(size_t)p * 64];
                }
            }
            if (WIDE == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pos += 4 * F;
        if (pos >= nfrag_per_wave) pos -= nfrag_per_wave;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
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
void run(const char* name, K kern, const bf16x8* w, int nfrag, int ksteps, float* out, unsigned long long* cyc, int grid) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 100 * 1024, 0, w, nfrag, ksteps, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 100 * 1024, 0, w, nfrag, ksteps, out, cyc);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= grid;
    const double us = ms * 1e3 / it;
    printf("%-40s grid %3d: %8.1f us, %7.0f cycles per k-step (%.2f GHz), 36-MFMA floor 576; weight bytes/clk/CU %.1f\n", name, grid, us, m / ksteps,
           m / us * 1e-3, 4.0 * 8192 / (m / ksteps));
}

int main() {
    const int nfrag = 640, ksteps = 640;         // per wave: 640 KiB region, 640 k-steps x 8 fragments = 8 passes over it
    bf16x8* w; float* out; unsigned long long* cyc;
    CK(hipMalloc(&w, (size_t)4 * nfrag * 1024)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8 * 1024));
    std::vector<unsigned short> h((size_t)4 * nfrag * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22 & 0x3ff) + ((i & 1) << 15));
    CK(hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    run("mfma x3 only", (k<false, false, 3, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("mfma x3 + lds reads", (k<false, true, 3, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("mfma x3 + weight loads", (k<true, false, 3, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("mfma x3 + lds + loads (the k-step)", (k<true, true, 3, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("same, compiler-scheduled", (k<true, true, 3, 1>), w, nfrag, ksteps, out, cyc, 256);
    run("loads + lds only (no mfma)", (k<true, true, 0, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("mfma x1 (12 per k-step) + lds + loads", (k<true, true, 1, 0>), w, nfrag, ksteps, out, cyc, 256);
    run("the k-step, manual source order", (k<true, true, 3, 2>), w, nfrag, ksteps, out, cyc, 256);
    run("manual order, no loads", (k<false, true, 3, 2>), w, nfrag, ksteps, out, cyc, 256);
    run("manual order, no lds", (k<true, false, 3, 2>), w, nfrag, ksteps, out, cyc, 256);
    run("the k-step, 64 workgroups", (k<true, true, 3, 0>), w, nfrag, ksteps, out, cyc, 64);
    run("the k-step, 8 workgroups", (k<true, true, 3, 0>), w, nfrag, ksteps, out, cyc, 8);
    return 0;
}
