// Building blocks of the split-bf16 "x3" kernels on v_mfma_f32_16x16x32_bf16 (csrc/fine_x3.hip; csrc/encoder_x3.hip carries
// its own copy specialised to 3 token tiles): 16-token tiles, swizzled (hi, lo) LDS planes, per-wave weight streams behind a
// buffer descriptor pulled through a register ring, GEMM stages with pinned issue order.
//
// Lane maps (lane l, c16 = l & 15, q = l >> 4; 8 bf16 = 16 B per lane and operand):
//     A[row = c16][k = 8q + j]      B[k = 8q + j][col = c16]      D[row = 4q + reg][col = c16]   reg = 0..3
// An accumulator quad of a lane is four consecutive ROWS of one column, so two quads (rows 4q.., 16 + 4q..) are, as they
// stand, the 8 k-elements of that lane for a following product that contracts over the rows -- as the A operand (transposed
// use) or as the B operand -- provided the other operand orders its k the same way: k slot (q, j) = row 4q + j of the first
// quad's tile for j < 4, of the second quad's tile for j >= 4.
#pragma once
#include "tile_bf16.h"

namespace x3 {

constexpr int R = 16;                           // ring depth in 1 KiB fragments (4 registers each)

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// c += a * b, split cross terms first (small terms first)
__device__ __forceinline__ f32x4 mma16x3(const bf16x8& ahi, const bf16x8& alo, const bf16x8& bhi, const bf16x8& blo, f32x4 c) {
    c = mma16(alo, bhi, c);
    c = mma16(ahi, blo, c);
    return mma16(ahi, bhi, c);
}

// 8 f32 (two accumulator quads) -> (hi, lo) fragment; element j < 4 from a, j >= 4 from b
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 h0, l0, h1, l1;
        split_bf16(a[j], h0, l0);
        split_bf16(b[j], h1, l1);
        hi[j] = h0; lo[j] = l0; hi[4 + j] = h1; lo[4 + j] = l1;
    }
}

// byte offset of 16-byte chunk `chunk` of row 16 tt + c16 (row & 15 == c16) in a swizzled plane
__device__ __forceinline__ int poff(int tt, int c16, int chunk, int rowb) { return (16 * tt + c16) * rowb + ((chunk ^ c16) << 4); }

// four consecutive features f0 .. f0 + 3 (f0 % 4 == 0) of token row 16 tt + c16 -> 8-byte store into both planes
__device__ __forceinline__ void store_quad(const f32x4& v, char* ph, char* pl, int rowb, int tt, int c16, int f0) {
    bf16x4 vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 hh, ll;
        split_bf16(v[j], hh, ll);
        vh[j] = hh; vl[j] = ll;
    }
    const int off = poff(tt, c16, f0 >> 3, rowb) + 2 * (f0 & 7);
    *reinterpret_cast<bf16x4*>(ph + off) = vh;
    *reinterpret_cast<bf16x4*>(pl + off) = vl;
}

// sum over the four 16-lane rows (q) of a wave, result in every lane: v_permlane16_swap + v_permlane32_swap (vector ALU only)
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

struct Ring { bf16x8 s[R]; };

// One wave's weight stream: 1 KiB fragments behind a buffer descriptor; a load is voffset = 16 lane + a scalar fragment
// offset (no vector address arithmetic); fragments beyond the stream's end are dropped by the range check.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct WStream {
    __amdgpu_buffer_rsrc_t rs;
    int voff;
    __device__ __forceinline__ void open(const bf16x8* base_uniform, int nfrags, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(base_uniform), 0, nfrags * 1024, 0x00020000);
        voff = lane * 16;
    }
    __device__ __forceinline__ bf16x8 load(int frag) const {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, frag * 1024, 0));
    }
};

template <int NTT>
struct XFrag { bf16x8 h[NTT], l[NTT]; };

template <int NTT>
__device__ __forceinline__ void read_x(XFrag<NTT>& x, const char* ph, const char* pl, int rowb, int chunk, int c16) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int off = poff(tt, c16, chunk, rowb);
        x.h[tt] = *reinterpret_cast<const bf16x8*>(ph + off);
        x.l[tt] = *reinterpret_cast<const bf16x8*>(pl + off);
    }
}

// One GEMM stage of a wave as a rolled loop: acc[ft][tt] += W(ft) . Act(tt) over NKS k-steps of 32.
//   weights: ring slots (ks * 2 NF + 2 ft + plane) % R in stream order; after a k-step its slots are refilled with the
//            fragments R positions ahead (stream `wa` from position posa = this stage's first fragment + R);
//   activations: swizzled (hi, lo) planes, chunk0 = first 16-byte chunk of k-step 0; read one k-step ahead (the read past
//            the last k-step stays inside the plane's allocation and is discarded);
//   W_IS_A: D[feature][token] (weights are the A operand) / false: D[token][feature].
// The loop body covers whole passes over the ring, so slot indices and the activation double buffer stay static.
//   refills: the first `na` fragments of the stage come from stream `wa` at posa.., the rest from `wb` at posb.. (a stream that
//            runs into another one: the main stream into the next layer's K|V stream)
template <int NTT, int NF, int NKS, bool W_IS_A>
__device__ __forceinline__ void gemm_stage(f32x4 (&acc)[NF][NTT], Ring& ring, const WStream& wa, int posa, const char* ph, const char* pl,
                                           int rowb, int chunk0, int c16, int q, int na = 1 << 30, const WStream* wb = nullptr, int posb = 0) {
    constexpr int F = 2 * NF;
    constexpr int U = (R / F) >= 2 ? (R / F) : 2;
    constexpr int NMFMA = 3 * NF * NTT;
    static_assert(NKS % U == 0 && (U * F) % R == 0, "a loop body must cover whole passes over the ring");
    XFrag<NTT> x[2];
    read_x<NTT>(x[0], ph, pl, rowb, chunk0 + q, c16);
#pragma unroll 1
    for (int it = 0; it < NKS / U; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cur = u & 1;
            const int ks = it * U + u;
            read_x<NTT>(x[cur ^ 1], ph, pl, rowb, chunk0 + 4 * (ks + 1) + q, c16);
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
                const bf16x8& whi = ring.s[(u * F + 2 * ft) % R];
                const bf16x8& wlo = ring.s[(u * F + 2 * ft + 1) % R];
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    acc[ft][tt] = W_IS_A ? mma16x3(whi, wlo, x[cur].h[tt], x[cur].l[tt], acc[ft][tt])
                                         : mma16x3(x[cur].h[tt], x[cur].l[tt], whi, wlo, acc[ft][tt]);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int idx = ks * F + f;
                ring.s[(u * F + f) % R] = (idx < na) ? wa.load(posa + idx) : wb->load(posb + idx - na);
            }
            // issue order: the next k-step's activation reads first (one per MFMA), then the refills one per MFMA
            constexpr int NRD = 2 * NTT, NLD = F;
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else if (i - NRD < NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// per-token moments of this wave's 16 NF features (D[feature][token] accumulators): sum and centred second moment; all q
// groups hold the result.  Two-pass inside the wave; waves are merged later (Chan et al.): one barrier per LayerNorm.
template <int NTT, int NF>
__device__ __forceinline__ void wave_moments(const f32x4 (&m)[NF][NTT], float (&s)[NTT], float (&d2)[NTT]) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        float a = 0.f;
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) a += (m[ft][tt][0] + m[ft][tt][1]) + (m[ft][tt][2] + m[ft][tt][3]);
        a = sum_over_q(a);
        const float mean = a * (1.0f / (16 * NF));
        float b = 0.f;
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = m[ft][tt][r] - mean;
                b += d * d;
            }
        b = sum_over_q(b);
        s[tt] = a;
        d2[tt] = b;
    }
}

}  // namespace x3
