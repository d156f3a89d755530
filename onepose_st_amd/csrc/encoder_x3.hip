// Coarse LoFTR encoder layer, split-bf16 (x = hi + lo, 3 MFMAs per product, f32 accumulate), second-generation mapping.
// Same mathematics as csrc/encoder.hip / encoder_bf16.hip (reference: loftr_module/transformer.py:65-94,146-159,
// linear_attention.py:29-61); what changes is how the layer is laid over a CU:
//
//   * one workgroup per CU, never two: a CU streams an L2-resident block at 114-128 GB/s (tools/micro/l2_stream.hip: one
//     layer's 2.6 MB of hi + lo fragments in 21-23 us), but two workgroups on a CU get 60 GB/s each -- so every weight
//     fragment must enter a CU ONCE and feed all of that CU's tokens from registers;
//   * 16-token MFMA tiles (v_mfma_f32_16x16x32_bf16): a workgroup owns 48 tokens = 3 token tiles, so c2's 11 800 tokens
//     are 246 workgroups on 256 CUs (32-wide tiles quantise to 64 tokens on 113 of the CUs);
//   * 4 waves (one per SIMD); wave fw owns a FEATURE group (64 of 256 outputs, 32 of a 128-wide hidden chunk) of all three
//     token tiles, so a 1 KiB weight fragment feeds 3 x (2 or 1) MFMAs straight from its registers;
//   * the weights of a layer are packed on the host in exactly the order a wave consumes them (packing.pack_coarse_layer_x3):
//     one linear 640 KiB stream per wave, pulled through a 32-fragment register ring that never drains between stages;
//   * epilogues overlap the matrix pipe inside a wave: the hidden chunk c of the MLP is rectified / split / stored while
//     chunk c + 1's GEMM issues (double-buffered hidden planes, one barrier per chunk), LayerNorm 1's normalise + store runs
//     under the x half of the first MLP GEMM, LayerNorms take ONE barrier (per-wave two-pass moments merged by Chan's
//     formula), the residual add and the output store work from the accumulator layout (no LDS staging pass).
//
// Lane maps of v_mfma_f32_16x16x32_bf16 (lane l, c16 = l & 15, q = l >> 4; 8 bf16 = 16 B per lane and operand):
//     A[row = c16][k = 8q + j]      B[k = 8q + j][col = c16]      D[row = 4q + reg][col = c16]   reg = 0..3
// Weights are the A operand (rows = output features) in every per-token GEMM, so an accumulator holds four consecutive
// features of one token per lane; the K/V projection of the fused tail swaps roles (tokens on rows) so that phi(K)^T V
// contracts over the accumulators' ROW index without leaving registers (k order permuted the same way on both operands).
#include "tile_bf16.h"
#include <stdlib.h>

namespace {

constexpr int C = 256, NH = 8;
constexpr int NTT = 3, TOK = 16 * NTT;          // token tiles / tokens per workgroup
constexpr int ROWB = C * 2;                     // plane row pitch (bytes), 32 chunks of 16 B
constexpr int HROWB = 128 * 2;                  // hidden-chunk plane pitch, 16 chunks
constexpr int PLANE = TOK * ROWB;               // 24 576 B
constexpr int HPLANE = TOK * HROWB;             // 12 288 B
#ifndef X3_R
#define X3_R 16
#endif
constexpr int R = X3_R;                        // ring depth in 1 KiB fragments (4 registers each)
constexpr int MAIN_FRAGS = 512, KV_FRAGS = 128; // per wave: Q 64 | merge 64 | MLP 384 ;  next layer's K|V 128
constexpr int KV_PART_FLOATS = NH * 1024 + NH * 32;             // per-tile slab: KV [head][dt][vt][lane][4] f32 + Ksum [head][32]
constexpr int KV_FRAG_BYTES = NH * 2 * 2 * 64 * 16;             // [head][vt][plane][lane][16 B]
constexpr int KV_BLOCK_BYTES = KV_FRAG_BYTES + NH * 32 * 4;     // + Ksum f32
constexpr int LDS_BYTES = 4 * PLANE + 4 * HPLANE + 4 * TOK * 2 * 4;   // X, Y planes (hi, lo) + 2 hidden buffers + LayerNorm scratch

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

// sum over the four 16-lane rows (q) of a wave, result in every lane: v_permlane16_swap + v_permlane32_swap (vector ALU
// only; __shfl_xor is an LDS round trip per step)
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

// One wave's weight stream: 1 KiB fragments behind a buffer descriptor; every load is voffset = 16 lane (one VGPR for the
// whole kernel) + a scalar / immediate fragment offset, so the stream costs no vector address arithmetic.
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

struct XFrag { bf16x8 h[NTT], l[NTT]; };

__device__ __forceinline__ void read_x(XFrag& x, const char* ph, const char* pl, int rowb, int chunk, int c16) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int off = poff(tt, c16, chunk, rowb);
        x.h[tt] = *reinterpret_cast<const bf16x8*>(ph + off);
        x.l[tt] = *reinterpret_cast<const bf16x8*>(pl + off);
    }
}

// One GEMM stage of a wave: acc[ft][tt] += W(ft) . Act(tt) over NKS k-steps of 32.
//   weights: ring slots (ks * 2 NF + 2 ft + plane) % R, consumed in stream order; after a k-step its slots are refilled with
//            the fragments R positions ahead: the first NA of them from stream `wa` at posa.., the rest (up to NREFILL in
//            all) from stream `wb` at posb.. (the main stream runs into the next layer's K|V stream; the tail of the last
//            stream has nothing to fetch);
//   activations: swizzled (hi, lo) planes, chunk0 = first 16-byte chunk of k-step 0; read one k-step ahead;
//   W_IS_A: D[feature][token] (weights are the A operand) / false: D[token][feature];
//   side(ks): a slice of independent epilogue work issued beside k-step ks's MFMAs.
template <int NF, int NKS, bool W_IS_A, int NREFILL, int NA, class Side>
__device__ __forceinline__ void gemm_stage(f32x4 (&acc)[NF][NTT], Ring& ring, const WStream& wa, int posa, const WStream& wb, int posb,
                                           const char* ph, const char* pl, int rowb, int chunk0, int c16, int q, Side&& side) {
    constexpr int S0 = 0;
    constexpr int F = 2 * NF;
    constexpr int NMFMA = 3 * NF * NTT;              // per k-step
    XFrag x[2];
    read_x(x[0], ph, pl, rowb, chunk0 + q, c16);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < NKS) read_x(x[cur ^ 1], ph, pl, rowb, chunk0 + 4 * (ks + 1) + q, c16);
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
            const bf16x8& whi = ring.s[(S0 + ks * F + 2 * ft) % R];
            const bf16x8& wlo = ring.s[(S0 + ks * F + 2 * ft + 1) % R];
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
                acc[ft][tt] = W_IS_A ? mma16x3(whi, wlo, x[cur].h[tt], x[cur].l[tt], acc[ft][tt])
                                     : mma16x3(x[cur].h[tt], x[cur].l[tt], whi, wlo, acc[ft][tt]);
        }
#pragma unroll
        for (int f = 0; f < F; ++f)
            if (ks * F + f < NREFILL)
                ring.s[(S0 + ks * F + f) % R] = (ks * F + f < NA) ? wa.load(posa + ks * F + f) : wb.load(posb + ks * F + f - NA);
        side(ks);
        // issue order inside the k-step, one MFMA at a time: behind each of the first MFMAs one of the next k-step's activation
        // reads (their LDS round trip must be over when this k-step's MFMAs are), then one ring refill per MFMA, and behind
        // every MFMA up to two vector instructions and an LDS write of the side work (what a 16-cycle 16x16x32 slot leaves
        // of the SIMD's issue bandwidth)
        constexpr int NRD = 2 * NTT, NLD = F;
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (i - NRD < NLD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // nothing crosses a k-step
    }
}

// The same stage as a rolled loop (no side work; refills unconditional from ONE stream: a fragment beyond the stream's end is
// dropped by the descriptor's range check).  The body covers U k-steps = whole passes over the ring, so slot indices and the
// activation double-buffer stay static; keeps the kernel's code inside the instruction cache.
template <int NF, int NKS, bool W_IS_A>
__device__ __forceinline__ void gemm_stage_rolled(f32x4 (&acc)[NF][NTT], Ring& ring, const WStream& wa, int posa, const char* ph,
                                                  const char* pl, int rowb, int chunk0, int c16, int q) {
    constexpr int F = 2 * NF;
    constexpr int U = (R / F) >= 2 ? (R / F) : 2;
    constexpr int NMFMA = 3 * NF * NTT;
    static_assert(NKS % U == 0 && (U * F) % R == 0, "a loop body must cover whole passes over the ring");
    XFrag x[2];
    read_x(x[0], ph, pl, rowb, chunk0 + q, c16);
#pragma unroll 1
    for (int it = 0; it < NKS / U; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cur = u & 1;
            const int ks = it * U + u;
            read_x(x[cur ^ 1], ph, pl, rowb, chunk0 + 4 * (ks + 1) + q, c16);          // one k-step past the end: stays inside the plane's row
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
            for (int f = 0; f < F; ++f) ring.s[(u * F + f) % R] = wa.load(posa + ks * F + f);
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

struct NoSide { __device__ __forceinline__ void operator()(int) const {} };

// per-token moments of this wave's 64 features (D[feature][token] accumulators): sum and centred second moment, all q
// groups hold the result.  Two-pass inside the wave; waves are merged later (Chan et al.), so one barrier per LayerNorm.
__device__ __forceinline__ void wave_moments(const f32x4 (&m)[4][NTT], float (&s)[NTT], float (&d2)[NTT]) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        float a = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) a += (m[ft][tt][0] + m[ft][tt][1]) + (m[ft][tt][2] + m[ft][tt][3]);
        a = sum_over_q(a);
        const float mean = a * (1.0f / 64);
        float b = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
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

__device__ __forceinline__ void publish_moments(float* scratch, const float (&s)[NTT], const float (&d2)[NTT], int fw, int c16, int q) {
    if (q == 0) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            float2 v = {s[tt], d2[tt]};
            *reinterpret_cast<float2*>(scratch + (fw * TOK + 16 * tt + c16) * 2) = v;
        }
    }
}

// merge the four waves' moments of token 16 tt + c16 -> mean, 1 / sqrt(var + eps)   (biased variance over 256 features)
__device__ __forceinline__ void merged_stats(const float* scratch, int tt, int c16, float& mean, float& rstd) {
    float sw[4], dw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(scratch + (w * TOK + 16 * tt + c16) * 2);
        sw[w] = v.x; dw[w] = v.y;
    }
    mean = ((sw[0] + sw[1]) + (sw[2] + sw[3])) * (1.0f / C);
    float m2 = (dw[0] + dw[1]) + (dw[2] + dw[3]);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float d = sw[w] * (1.0f / 64) - mean;
        m2 += 64.f * d * d;
    }
    rstd = 1.0f / sqrtf(m2 * (1.0f / C) + 1e-5f);
}

struct EncX3Args {
    const float* x[2];
    float* y[2];
    long long xbs[2], ybs[2];
    int L[2];
    int tiles[2];              // 48-token tiles per stream
    const char* kv[2];         // KV block each stream attends to
    long long kvbs;            // bytes between frames
    float srclen[2];
    const bf16x8* wmain;       // [4 waves][MAIN_FRAGS][64]
    const bf16x8* wkv;         // K|V stream consumed by the tail ([4][KV_FRAGS][64]); NULL: no tail
    const float* ln;           // g1 b1 g2 b2
    float* partial;            // K/V slabs written by the tail: [B][tiles0 + tiles1][KV_PART_FLOATS]
    unsigned long long* stamps;
};

// K|V projection of the 48 tokens in the X planes (heads 2 fw, 2 fw + 1) and their phi(K)^T V / Ksum slab.
// The ring must hold the first R fragments of this wave's K|V stream `wsk`.
__device__ __forceinline__ void kv_tail(Ring& ring, const WStream& wsk, const char* XH, const char* XL, int tok0, int L,
                                        float* __restrict__ out, int fw, int lane) {
    const int c16 = lane & 15, q = lane >> 4;
    f32x4 kk[8][NTT];            // D[token 4q + r][feature c16]: ft 0..3 = K of heads 2fw (0,1), 2fw+1 (2,3); ft 4..7 = V likewise
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) kk[ft][tt] = zero4();
    gemm_stage_rolled<8, 8, false>(kk, ring, wsk, R, XH, XL, ROWB, 0, c16, q);
    const float inv_len = 1.0f / (float)L;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                kk[ft][tt][r] = (tok0 + 16 * tt + 4 * q + r < L) ? elu_plus_one_fast(kk[ft][tt][r]) : 0.f;   // padded tokens drop out
                kk[4 + ft][tt][r] *= inv_len;                                                             // values / v_length
            }
    const f32x4 z4 = zero4();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int head = 2 * fw + hh;
        // fragments over the token (k) axis: k-step 0 = tiles 0, 1; k-step 1 = tile 2 + zeros (k order as the accumulators hold it)
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            split8(kk[2 * hh + t][0], kk[2 * hh + t][1], ah[t][0], al[t][0]);
            split8(kk[2 * hh + t][2], z4, ah[t][1], al[t][1]);
            split8(kk[4 + 2 * hh + t][0], kk[4 + 2 * hh + t][1], bh[t][0], bl[t][0]);
            split8(kk[4 + 2 * hh + t][2], z4, bh[t][1], bl[t][1]);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                f32x4 kvt = zero4();                  // KV[d = 16 dt + 4q + r][v = 16 vt + c16]
                kvt = mma16x3(ah[dt][0], al[dt][0], bh[vt][0], bl[vt][0], kvt);
                kvt = mma16x3(ah[dt][1], al[dt][1], bh[vt][1], bl[vt][1], kvt);
                *reinterpret_cast<f32x4*>(out + ((size_t)((head * 2 + dt) * 2 + vt) * 64 + lane) * 4) = kvt;
            }
            float s = 0.f;                            // Ksum[d = 16 dt + c16]: exact f32 sum over the 48 tokens
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) s += (kk[2 * hh + dt][tt][0] + kk[2 * hh + dt][tt][1]) + (kk[2 * hh + dt][tt][2] + kk[2 * hh + dt][tt][3]);
            s = sum_over_q(s);
            if (q == 0) out[NH * 1024 + head * 32 + 16 * dt + c16] = s;
        }
    }
}

// global f32 rows -> swizzled (hi, lo) planes of the 48-token tile; zero fill beyond L.  load() and store() are separate so
// that other loads can be queued behind the rows before the conversion waits for them.
struct StagedRows {
    static constexpr int ITEMS = TOK * (C / 8) / 256;       // 6 (row, 8-feature chunk) items per thread
    f32x4 v0[ITEMS], v1[ITEMS];
    __device__ __forceinline__ void load(const float* __restrict__ x, int tok0, int L, int tid) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 256 * i, row = it >> 5, ch = it & 31;
            v0[i] = v1[i] = zero4();
            if (tok0 + row < L) {
                const float* src = x + (size_t)(tok0 + row) * C + 8 * ch;
                v0[i] = *reinterpret_cast<const f32x4*>(src);
                v1[i] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
    }
    // stash: exact f32 copy of the tile, [row][64 chunks of 16 B] with chunk ^ (row & 15) (NULL: none)
    __device__ __forceinline__ void store(char* ph, char* pl, char* stash, int tid) const {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 256 * i, row = it >> 5, ch = it & 31;
            bf16x8 vh, vl;
            split8(v0[i], v1[i], vh, vl);
            const int off = row * ROWB + ((ch ^ (row & 15)) << 4);
            *reinterpret_cast<bf16x8*>(ph + off) = vh;
            *reinterpret_cast<bf16x8*>(pl + off) = vl;
            if (stash) {
                *reinterpret_cast<f32x4*>(stash + row * (C * 4) + (((2 * ch) ^ (row & 15)) << 4)) = v0[i];
                *reinterpret_cast<f32x4*>(stash + row * (C * 4) + (((2 * ch + 1) ^ (row & 15)) << 4)) = v1[i];
            }
        }
    }
};

template <bool ONLY_KV>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 1) void enc_x3_kernel(EncX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;
    char* XL = smem + PLANE;
    char* YH = smem + 2 * PLANE;
    char* YL = smem + 3 * PLANE;
    char* HB = smem + 4 * PLANE;                    // hidden buffer b: hi at HB + 2 b HPLANE, lo at + HPLANE
    float* scratch = reinterpret_cast<float*>(smem + 4 * PLANE + 4 * HPLANE);
    const int tid = threadIdx.x, lane = tid & 63, fw = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * TOK;
    const float* xg = a.x[s] + (size_t)b * a.xbs[s];
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    float* slab = a.partial + ((size_t)b * (a.tiles[0] + a.tiles[1]) + tile) * KV_PART_FLOATS;

    Ring ring;
    const int fwu = __builtin_amdgcn_readfirstlane(fw);         // the wave index, provably uniform (scalar stream bases)
    const bool tail = a.wkv != nullptr;
    WStream wsk;                                                 // no tail: zero records, every load of it is dropped by the range check
    wsk.open(tail ? a.wkv + (size_t)fwu * KV_FRAGS * 64 : a.wmain, tail ? KV_FRAGS : 0, lane);
    OPHIP_STAMP(a.stamps, wg, 0);
    if (ONLY_KV) {
        StagedRows rows;
        rows.load(xg, tok0, L, tid);
#pragma unroll
        for (int i = 0; i < R; ++i) ring.s[i] = wsk.load(i);
        __builtin_amdgcn_sched_barrier(0);
        rows.store(XH, XL, nullptr, tid);
        __syncthreads();
        kv_tail(ring, wsk, XH, XL, tok0, L, slab, fw, lane);
        return;
    }
    WStream wsm;
    wsm.open(a.wmain + (size_t)fwu * MAIN_FRAGS * 64, MAIN_FRAGS, lane);
    // loads return in issue order: the activation rows go first (the staging barrier waits for them only), the weight stream
    // and this wave's slices of the attention state stay in flight behind them
    StagedRows rows;
    rows.load(xg, tok0, L, tid);
#pragma unroll
    for (int i = 0; i < R; ++i) ring.s[i] = wsm.load(i);
    // KV^T fragments (A operand, k = d in accumulator order) and Ksum of heads 2fw, 2fw+1
    const char* kvb = a.kv[s] + (size_t)b * a.kvbs;
    bf16x8 kvh[2][2], kvl[2][2];
    f32x4 ksm[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int head = 2 * fw + hh;
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            kvh[hh][vt] = *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((head * 2 + vt) * 2 + 0) * 64 + lane) * 16);
            kvl[hh][vt] = *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((head * 2 + vt) * 2 + 1) * 64 + lane) * 16);
        }
        const float* kp = reinterpret_cast<const float*>(kvb + KV_FRAG_BYTES) + head * 32 + 4 * q;
        ksm[hh][0] = *reinterpret_cast<const f32x4*>(kp);
        ksm[hh][1] = *reinterpret_cast<const f32x4*>(kp + 16);
    }
    // LayerNorm parameters of this lane's features (parked in the accumulator-file half of the registers until needed)
    f32x4 xr[4][NTT], g1[4], b1[4], g2[4], b2[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const int f0 = 64 * fw + 16 * ft + 4 * q;
        g1[ft] = *reinterpret_cast<const f32x4*>(a.ln + f0);
        b1[ft] = *reinterpret_cast<const f32x4*>(a.ln + C + f0);
        g2[ft] = *reinterpret_cast<const f32x4*>(a.ln + 2 * C + f0);
        b2[ft] = *reinterpret_cast<const f32x4*>(a.ln + 3 * C + f0);
    }
    __builtin_amdgcn_sched_barrier(0);
    rows.store(XH, XL, HB, tid);                     // + an exact f32 copy in the (still idle) hidden buffers
    __syncthreads();
    // residual rows in the accumulator layout, exact f32: four features of one token per lane and (ft, tt), read back from the
    // stash (a global re-read in this layout touches 16 rows per instruction and stalls the ring queued behind it)
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
            xr[ft][tt] = *reinterpret_cast<const f32x4*>(HB + (16 * tt + c16) * (C * 4) + (((16 * fw + 4 * ft + q) ^ c16) << 4));
    OPHIP_STAMP(a.stamps, wg, 1);

    // ---- Q projection (heads 2fw, 2fw+1), phi, linear attention from registers -> msg planes (Y) ----------------------
    {
        f32x4 qa[4][NTT];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) qa[ft][tt] = zero4();
        gemm_stage_rolled<4, 8, true>(qa, ring, wsm, 0 + R, XH, XL, ROWB, 0, c16, q);
        OPHIP_STAMP(a.stamps, wg, 2);
        const float S = a.srclen[s];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 p0, p1;
                float den = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p0[r] = elu_plus_one_fast(qa[2 * hh][tt][r]);
                    p1[r] = elu_plus_one_fast(qa[2 * hh + 1][tt][r]);
                    den += p0[r] * ksm[hh][0][r] + p1[r] * ksm[hh][1][r];
                }
                den = sum_over_q(den);
                bf16x8 qh, ql;
                split8(p0, p1, qh, ql);                          // B[k = d][token]: d = 4q + j (j < 4), 16 + 4q + j - 4
                const float z = rcp_fast(den + 1e-6f) * S;
#pragma unroll
                for (int vt = 0; vt < 2; ++vt) {
                    f32x4 num = mma16x3(kvh[hh][vt], kvl[hh][vt], qh, ql, zero4());     // num^T[v][tok] = sum_d KV[d][v] phiQ[tok][d]
#pragma unroll
                    for (int r = 0; r < 4; ++r) num[r] *= z;
                    store_quad(num, YH, YL, ROWB, tt, c16, 64 * fw + 32 * hh + 16 * vt + 4 * q);
                }
            }
    }
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 3);

    // ---- merge, LayerNorm 1 (normalise + store run under the x half of the first MLP GEMM) ------------------------------
    f32x4 m[4][NTT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) m[ft][tt] = zero4();
    gemm_stage_rolled<4, 8, true>(m, ring, wsm, 64 + R, YH, YL, ROWB, 0, c16, q);
    OPHIP_STAMP(a.stamps, wg, 4);
    {
        float sm[NTT], dm[NTT];
        wave_moments(m, sm, dm);
        publish_moments(scratch, sm, dm, fw, c16, q);
    }
    __syncthreads();                                 // moments visible; every wave is done reading the msg planes
    OPHIP_STAMP(a.stamps, wg, 5);

    // ---- MLP: hidden = relu([x, msg] W0^T) in four 128-feature chunks, o += hidden_chunk W2[:, chunk]^T.  Chunk c + 1's
    //      first GEMM carries chunk c's relu / split / store as its side work (hidden buffers alternate) -------------------
    f32x4 o[4][NTT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) o[ft][tt] = zero4();
    auto hidden_store = [&](const f32x4 (&hd)[2][NTT], int buf, int part) {       // part 0..5: one (ft, tt) quad
        const int ft = part / NTT, tt = part % NTT;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(hd[ft][tt][r], 0.f);
        store_quad(v, HB + 2 * buf * HPLANE, HB + (2 * buf + 1) * HPLANE, HROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
    };
    f32x4 hA[2][NTT], hB[2][NTT];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) hA[ft][tt] = hB[ft][tt] = zero4();
    // chunk 0, x half: LayerNorm 1's normalise + split + store of (ft = ks / 2, tokens of all tt) beside k-steps 0..7
    {
        float mean[NTT], rstd[NTT];
        auto ln1_side = [&](int ks) {
            if (ks == 0) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) merged_stats(scratch, tt, c16, mean[tt], rstd[tt]);
            }
            // 12 (ft, tt) quads over k-steps 1..7: two per step, the last two steps one each
            const int first = ks == 0 ? 12 : (ks <= 5 ? 2 * (ks - 1) : 10 + (ks - 6));
            const int count = ks == 0 ? 0 : (ks <= 5 ? 2 : 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int part = first + k;
                if (k < count && part < 12) {
                    const int ft = part / NTT, tt = part % NTT;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (m[ft][tt][r] - mean[tt]) * rstd[tt] * g1[ft][r] + b1[ft][r];
                    store_quad(v, YH, YL, ROWB, tt, c16, 64 * fw + 16 * ft + 4 * q);
                }
            }
        };
        gemm_stage<2, 8, true, 32, 32>(hA, ring, wsm, 128 + R, wsm, 0, XH, XL, ROWB, 0, c16, q, ln1_side);
    }
    __syncthreads();                                 // LayerNorm-1 output (the msg half of the MLP input) is in the Y planes
    OPHIP_STAMP(a.stamps, wg, 6);
    gemm_stage_rolled<2, 8, true>(hA, ring, wsm, 160 + R, YH, YL, ROWB, 0, c16, q);
    OPHIP_STAMP(a.stamps, wg, 7);
    // stream positions: W0c0 128 | W0c1 192 | W2c0 256 | W0c2 288 | W2c1 352 | W0c3 384 | W2c2 448 | W2c3 480
    auto w0_x = [&](f32x4 (&hn)[2][NTT], const f32x4 (&hp)[2][NTT], int pos, int pbuf) {       // x half (k-steps 0..7) + previous chunk's store
        gemm_stage<2, 8, true, 32, 32>(hn, ring, wsm, pos + R, wsm, 0, XH, XL, ROWB, 0, c16, q,
                                      [&](int ks) { if (ks < 6) hidden_store(hp, pbuf, ks); });
    };
    auto w0_y = [&](f32x4 (&hn)[2][NTT], int pos) {
        gemm_stage_rolled<2, 8, true>(hn, ring, wsm, pos + 32 + R, YH, YL, ROWB, 0, c16, q);
    };
    auto w2 = [&](int buf, int pos, auto&& side) {
        gemm_stage<4, 4, true, 32, 32>(o, ring, wsm, pos + R, wsm, 0, HB + 2 * buf * HPLANE, HB + (2 * buf + 1) * HPLANE, HROWB, 0, c16, q, side);
    };
    w0_x(hB, hA, 192, 0); w0_y(hB, 192);             // chunk 1 (+ chunk 0 -> hidden buffer 0)
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 8);
    w2(0, 256, NoSide());
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) hA[ft][tt] = zero4();
    w0_x(hA, hB, 288, 1); w0_y(hA, 288);             // chunk 2 (+ chunk 1 -> buffer 1)
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 9);
    w2(1, 352, NoSide());
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) hB[ft][tt] = zero4();
    w0_x(hB, hA, 384, 0); w0_y(hB, 384);             // chunk 3 (+ chunk 2 -> buffer 0; W2 chunk 0 finished two barriers ago)
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 10);
    w2(0, 448, [&](int ks) {                     // chunk 3 -> buffer 1 beside W2 chunk 2
        if (2 * ks < 6) hidden_store(hB, 1, 2 * ks);
        if (2 * ks + 1 < 6) hidden_store(hB, 1, 2 * ks + 1);
    });
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 11);
    // the last 32 fragments of the main stream: the refills beyond its end pull the head of the next layer's K|V stream (or,
    // without a tail, nothing: that stream has zero records)
    gemm_stage<4, 4, true, 32, 32 - R>(o, ring, wsm, 480 + R, wsk, 0, HB + 2 * HPLANE, HB + 3 * HPLANE, HROWB, 0, c16, q, NoSide());
    OPHIP_STAMP(a.stamps, wg, 12);

    // ---- LayerNorm 2, residual, output rows (and their planes for the fused K|V tail) -----------------------------------
    {
        float so[NTT], dq[NTT];
        wave_moments(o, so, dq);
        publish_moments(scratch, so, dq, fw, c16, q);
    }
    __syncthreads();                                 // also: every wave is done with the X planes (last read: chunk 3, x half)
    float* yg = a.y[s] + (size_t)b * a.ybs[s];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        float mean, rstd;
        merged_stats(scratch, tt, c16, mean, rstd);
        const int tok = tok0 + 16 * tt + c16;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xr[ft][tt][r] + ((o[ft][tt][r] - mean) * rstd * g2[ft][r] + b2[ft][r]);
            if (tok < L) *reinterpret_cast<f32x4*>(yg + (size_t)tok * C + 64 * fw + 16 * ft + 4 * q) = v;
            else v = zero4();                                        // rows beyond L stay zero: they drop out of phi(K) and V
            if (tail) store_quad(v, XH, XL, ROWB, tt, c16, 64 * fw + 16 * ft + 4 * q);
        }
    }
    OPHIP_STAMP(a.stamps, wg, 13);
    if (tail) {
        __syncthreads();
        kv_tail(ring, wsk, XH, XL, tok0, L, slab, fw, lane);
    }
    OPHIP_STAMP(a.stamps, wg, 14);
}

struct KvSumX3Args {
    const float* partial;
    char* kv;              // [B][2][KV_BLOCK_BYTES]
    int tiles[2];
};

constexpr int KVS_G = 16;

// fixed-order sum of the per-tile slabs (four consecutive floats per thread, 16 tile groups per workgroup, groups merged in a
// fixed order); emits KV^T as (hi, lo) bf16 A fragments [head][vt][plane][lane][j = 4 dt + r] and Ksum f32
__global__ __launch_bounds__(1024) void kv_sum_x3_kernel(KvSumX3Args a) {
    __shared__ f32x4 red[KVS_G][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int s = blockIdx.y & 1, b = blockIdx.y >> 1;
    const int ttot = a.tiles[0] + a.tiles[1];
    const int t0 = s ? a.tiles[0] : 0, nt = a.tiles[s];
    const int e = (blockIdx.x * 64 + o) * 4;
    const float* p = a.partial + ((size_t)b * ttot + t0) * KV_PART_FLOATS + e;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int t = g;
    for (; t + 3 * KVS_G < nt; t += 4 * KVS_G) {           // four independent loads in flight
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)t * KV_PART_FLOATS);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + KVS_G) * KV_PART_FLOATS);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 2 * KVS_G) * KV_PART_FLOATS);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 3 * KVS_G) * KV_PART_FLOATS);
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; t < nt; t += KVS_G) acc += *reinterpret_cast<const f32x4*>(p + (size_t)t * KV_PART_FLOATS);
    red[g][o] = acc;
    __syncthreads();
    if (g == 0) {
        f32x4 tot = red[0][o];
#pragma unroll
        for (int k = 1; k < KVS_G; ++k) tot += red[k][o];
        char* blk = a.kv + ((size_t)b * 2 + s) * KV_BLOCK_BYTES;
        if (e < NH * 1024) {
            const int ln = (e >> 2) & 63, vt = (e >> 8) & 1, dt = (e >> 9) & 1, head = e >> 10;
            bf16x4 vh, vl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __bf16 hh, ll;
                split_bf16(tot[r], hh, ll);
                vh[r] = hh; vl[r] = ll;
            }
            const size_t fr = (size_t)(head * 2 + vt) * 2;
            *reinterpret_cast<bf16x4*>(blk + ((fr + 0) * 64 + ln) * 16 + 8 * dt) = vh;
            *reinterpret_cast<bf16x4*>(blk + ((fr + 1) * 64 + ln) * 16 + 8 * dt) = vl;
        } else {
            *reinterpret_cast<f32x4*>(blk + KV_FRAG_BYTES + (size_t)(e - NH * 1024) * 4) = tot;
        }
    }
}


}  // namespace

extern "C" size_t ophip_encoder_x3_workspace_bytes(int B, int L3d, int L2d) {
    const size_t tiles = (size_t)((L3d + TOK - 1) / TOK + (L2d + TOK - 1) / TOK);
    return 2 * (size_t)B * tiles * KV_PART_FLOATS * 4 + (size_t)B * 2 * KV_BLOCK_BYTES + 256;      // two slab sets (ping-pong) + KV
}

extern "C" size_t ophip_encoder_x3_wpack_bytes(void) { return (size_t)4 * (MAIN_FRAGS + KV_FRAGS) * 1024 + 4 * C * 4; }

extern "C" int ophip_encoder_layer_x3(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                      const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                      void* workspace, void* stream_) {
    if (!x3d || !x2d || !y3d || !y2d || !wpack || !workspace) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || L3d < 1 || L2d < 1) return ophip_bad_arg(__func__, "B, L3d, L2d must be >= 1");
    if (slot != 0 && slot != 1) return ophip_bad_arg(__func__, "slot must be 0 or 1");
    if (x3d == y3d || x2d == y2d) return ophip_bad_arg(__func__, "in-place layer is not supported (cross layers read the pre-update streams)");
    if (reinterpret_cast<uintptr_t>(wpack) & 15) return ophip_bad_arg(__func__, "wpack must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const int t3 = (L3d + TOK - 1) / TOK, t2 = (L2d + TOK - 1) / TOK;
    const size_t part_floats = (size_t)B * (t3 + t2) * KV_PART_FLOATS;
    float* partial = reinterpret_cast<float*>(workspace) + (size_t)slot * part_floats;          // this layer's slabs
    float* partial_next = reinterpret_cast<float*>(workspace) + (size_t)(slot ^ 1) * part_floats;
    char* kv = reinterpret_cast<char*>(workspace) + 2 * part_floats * 4;
    kv += (256 - (reinterpret_cast<uintptr_t>(kv) & 255)) & 255;
    // layer block: [main stream 4 x 512 KiB][K|V stream 4 x 128 KiB][g1 b1 g2 b2 f32]   (packing.pack_coarse_layer_x3)
    const bf16x8* wmain = reinterpret_cast<const bf16x8*>(wpack);
    const bf16x8* wkv_own = wmain + (size_t)4 * MAIN_FRAGS * 64;
    const float* ln = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + (size_t)4 * (MAIN_FRAGS + KV_FRAGS) * 1024);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3_kernel<false>), LDS_BYTES, "hipFuncSetAttribute(enc_x3)")) return rc;
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3_kernel<true>), LDS_BYTES, "hipFuncSetAttribute(enc_x3 kv)")) return rc;

    EncX3Args aa;
    aa.x[0] = x3d; aa.x[1] = x2d; aa.y[0] = y3d; aa.y[1] = y2d;
    aa.xbs[0] = aa.ybs[0] = (long long)L3d * C; aa.xbs[1] = aa.ybs[1] = (long long)L2d * C;
    aa.L[0] = L3d; aa.L[1] = L2d; aa.tiles[0] = t3; aa.tiles[1] = t2;
    aa.kvbs = 2LL * KV_BLOCK_BYTES;
    aa.srclen[0] = (float)(is_cross ? L2d : L3d);
    aa.srclen[1] = (float)(is_cross ? L3d : L2d);
    aa.wmain = wmain; aa.ln = ln;
    aa.stamps = ophip_stamp_buffer();
    if (!kv_from_prev) {                             // first layer of a chain: its own K|V slabs
        EncX3Args ka = aa;
        ka.kv[0] = ka.kv[1] = nullptr;
        ka.wkv = wkv_own;
        ka.partial = partial;
        ka.stamps = nullptr;
        OPHIP_LAUNCH("kv_reduce", stream, enc_x3_kernel<true>, dim3(t3 + t2, B), dim3(256), LDS_BYTES, stream, ka);
        OPHIP_CHECK_LAUNCH();
    }
    KvSumX3Args sa;
    sa.partial = partial; sa.kv = kv; sa.tiles[0] = t3; sa.tiles[1] = t2;
    OPHIP_LAUNCH("kv_sum", stream, kv_sum_x3_kernel, dim3(KV_PART_FLOATS / 256, 2 * B), dim3(1024), 0, stream, sa);
    OPHIP_CHECK_LAUNCH();

    aa.kv[0] = kv + (is_cross ? KV_BLOCK_BYTES : 0);
    aa.kv[1] = kv + (is_cross ? 0 : KV_BLOCK_BYTES);
    aa.wkv = nullptr;
    aa.partial = partial_next;
    if (wpack_next) aa.wkv = reinterpret_cast<const bf16x8*>(wpack_next) + (size_t)4 * MAIN_FRAGS * 64;
    OPHIP_LAUNCH("attn_apply", stream, enc_x3_kernel<false>, dim3(t3 + t2, B), dim3(256), LDS_BYTES, stream, aa);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
