// Diagnostic: semantics of v_permlane16_swap / v_permlane32_swap as used by sum_over_q (csrc/encoder_x3.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float sum_over_q(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // -> rows [r0 r0 r2 r2], [r1 r1 r3 r3]
    const unsigned a0 = a[0], a1 = a[1];       // (a bit_cast applied to a vector ELEMENT reads element 0 on hipcc 7.2: copy first)
    const float s = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const unsigned w = __builtin_bit_cast(unsigned, s);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);      // -> [A A A A], [B B B B]
    const unsigned b0 = b[0], b1 = b[1];
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
}
__global__ void k(float* out, unsigned* raw) {
    const int l = threadIdx.x;
    const float v = (float)(1 << (l >> 4)) * 1000.f + (float)(l & 15);     // row q -> 1000 * 2^q + c16
    out[l] = sum_over_q(v);
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
    raw[l] = a[0]; raw[64 + l] = a[1];
    auto b = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
    raw[128 + l] = b[0]; raw[192 + l] = b[1];
}
int main() {
    float* o; unsigned* r;
    hipMalloc(&o, 256); hipMalloc(&r, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, r);
    float h[64]; unsigned hr[256];
    hipMemcpy(h, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 1024, hipMemcpyDeviceToHost);
    printf("sum_over_q (expect 15000 + 4 c16):"); for (int i = 0; i < 64; ++i) printf(" %g", h[i]); printf("\n");
    for (int t = 0; t < 4; ++t) { printf("raw%d:", t); for (int i = 0; i < 64; ++i) printf(" %u", hr[64 * t + i]); printf("\n"); }
    return 0;
}
