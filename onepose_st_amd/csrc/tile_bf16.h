// bf16 / split-bf16 MFMA building blocks (v_mfma_f32_32x32x16_bf16, f32 accumulate) for gfx950.
//
// Precision modes (template int NS):
//   NS = 1  plain bf16 operands                                   (1 MFMA per fragment pair)
//   NS = 3  split bf16: x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi
//           (3 MFMAs; the dropped terms are ~2^-17 relative, i.e. ~f32-grade results at 3/16 of the cost of the
//           exact-f32 matrix instruction, which runs at 1/16 of the bf16 rate on CDNA4)
//
// Lane maps of v_mfma_f32_32x32x16_bf16 (lane l, r = l & 31, h = l >> 5; 8 bf16 = 16 B per lane and operand):
//     A[i = r][k = 8h + j]      B[k = 8h + j][j' = r]      D[row = (reg&3) + 8*(reg>>2) + 4h][col = r]
//
// LDS activation image: one plane per split part, row-major [token][feature] bf16, row pitch = 2 * features bytes,
// 16-byte chunk c of row `row` lives at chunk (c ^ (row & 15)): ds_read_b128 of a fragment column is conflict free
// (rows of a 16-lane group are distinct mod 16) without padding -- the planes of a 64-token tile use all 160 KiB.
//
// An accumulator tile can feed the next product that contracts over its ROW index without touching LDS: regs
// 8s..8s+7 of a lane, converted to bf16, are the fragment of k-step s; the k order inside the step is permuted
// (element j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3)), so the other operand must be packed in that order.
#pragma once
#include "tile.h"

// Wave priority around the matrix loops of the ring GEMMs (OPHIP_GEMM_PRIO=1): two independent workgroups share every SIMD of a CU in the
// fine stage; with the GEMM wave preferred by the issue arbiter its matrix instructions go out back to back and the partner's vector
// work fills the issue slots between them, instead of the older wave winning whatever it runs.
#ifndef OPHIP_GEMM_PRIO
#define OPHIP_GEMM_PRIO 0
#endif
#if OPHIP_GEMM_PRIO
#define OPHIP_GEMM_PRIO_UP() __builtin_amdgcn_s_setprio(OPHIP_GEMM_PRIO)
#define OPHIP_GEMM_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define OPHIP_GEMM_PRIO_UP() do {} while (0)
#define OPHIP_GEMM_PRIO_DOWN() do {} while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// Fast transcendental forms for the bf16 / split-bf16 modes (their error budget is ~1e-5, these are ~1e-6):
// exp via v_exp_f32 (2^x), reciprocal via v_rcp_f32.  The exact-f32 kernels keep libm expf and true division.
__device__ __forceinline__ float elu_plus_one_fast(float x) { return x > 0.f ? x + 1.0f : __expf(x); }
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// lane-uniform-per-lane select of a whole fragment (4 packed registers)
__device__ __forceinline__ bf16x8 select_frag(bool keep, const bf16x8& v, const bf16x8& z) { return keep ? v : z; }

__device__ __forceinline__ bf16x8 zero_bf8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

// c += a * b with the split cross terms (small terms first)
template <int NS>
__device__ __forceinline__ f32x16 mma_bf16(const bf16x8& ahi, const bf16x8& alo, const bf16x8& bhi, const bf16x8& blo, f32x16 c) {
    if (NS == 3) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, c, 0, 0, 0);
}

// byte offset of 16-byte chunk `chunk` of row `row` in a swizzled plane with row pitch `rowb`
// (`swz` = 15 for rows of >= 16 chunks; 7 for 8-chunk rows (128-byte pitch): then rows r and r + 8 share a slot: 2-way)
__device__ __forceinline__ int plane_off(int row, int chunk, int rowb, int swz = 15) { return row * rowb + ((chunk ^ (row & swz)) << 4); }

// registers 8s..8s+7 of an accumulator as a (hi, lo) fragment
template <int NS>
__device__ __forceinline__ void acc_frag(const f32x16& a, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hh, ll;
        split_bf16(a[8 * s + j], hh, ll);
        hi[j] = hh;
        lo[j] = (NS == 3) ? ll : (__bf16)0.f;
    }
}

// Register ring of packed weight fragments, PD k-blocks deep, NT output tiles wide.  fill() issues the first PD k-blocks;
// it is called one phase AHEAD of the GEMM that consumes it (weights do not depend on data), so that the L2 round trip
// of a GEMM's first fragments overlaps the previous phase's MFMAs, epilogue and barrier.
template <int NT, int PD, int NS>
struct WRing {
    bf16x8 hi[PD][NT], lo[PD][NT];
    __device__ __forceinline__ void fill(const bf16x8* __restrict__ whi, const bf16x8* __restrict__ wlo, int tstride) {
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                hi[p][t] = whi[(size_t)t * tstride + (size_t)p * 64];
                lo[p][t] = (NS == 3) ? wlo[(size_t)t * tstride + (size_t)p * 64] : zero_bf8();
            }
    }
};

// registers 8s..8s+7 of an accumulator, each mapped through f(reg, value), as a (hi, lo) fragment
template <int NS, class F>
__device__ __forceinline__ void acc_frag_map(const f32x16& a, int s, F f, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hh, ll;
        split_bf16(f(8 * s + j, a[8 * s + j]), hh, ll);
        hi[j] = hh;
        lo[j] = (NS == 3) ? ll : (__bf16)0.f;
    }
}

// GEMM of a packed weight matrix against an LDS activation tile.
//   W_IS_A = true :  acc[t][tt] += W(tile t)[32 x K] . Act(token tile tt)[32 x K]^T   -> D[feature row][token col]
//   W_IS_A = false:  acc[t][tt] += Act(tt)[32 x K] . W(t)[32 x K]^T                   -> D[token row][feature col]
// whi/wlo point at [first tile of this wave][first k-block][lane]; tstride = fragments between tiles (= KB_total * 64).
// act_hi/act_lo: swizzled planes; chunk0 = first 16-byte chunk of the activation row that k-block 0 reads.
// Weight fragments stream L2 -> VGPR through the ring (compile-time slots): the kernels run one wave per SIMD, so the
// ring is what hides the L2 round trip behind the MFMAs.  KBLOCKS must be a multiple of PD.  `ring` must have been
// filled from the same whi/wlo/tstride.
template <int NT, int TT, int NS, bool W_IS_A, int KBLOCKS, int PD>
__device__ __forceinline__ void gemm_bf16_ring(f32x16 (&acc)[NT][TT], WRing<NT, PD, NS>& ring, const bf16x8* __restrict__ whi,
                                               const bf16x8* __restrict__ wlo, int tstride, const char* act_hi, const char* act_lo,
                                               int rowb, int chunk0, int lane, int swz = 15) {
    static_assert(KBLOCKS % PD == 0 && (PD % 2 == 0 || PD == 1), "k-blocks must be a multiple of the (even) prefetch depth");
    const int r = lane & 31, h = lane >> 5;
    // activation fragments are read one k-block ahead (two register sets), so the LDS round trip of k-block kb + 1
    // runs under the MFMAs of k-block kb even with a single wave on the SIMD
    bf16x8 x_hi[2][TT], x_lo[2][TT];
    auto read_x = [&](int kb, int set) {
        // The swizzled offsets are loop invariants of the caller's layer loop: hoisted out of it, the offsets of every k-block of every
        // GEMM stage stay live across the whole layer (~100 registers in the fine kernels; 87-187 spilled in the two-match form).
        // An opaque copy of the lane's row keeps each offset next to its read: 3 vector-ALU instructions per read, beside 3-12 MFMAs.
        int rr = r;
        asm volatile("" : "+v"(rr));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int off = plane_off(32 * tt + rr, chunk0 + 2 * kb + h, rowb, swz);
            x_hi[set][tt] = *reinterpret_cast<const bf16x8*>(act_hi + off);
            x_lo[set][tt] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(act_lo + off) : zero_bf8();
        }
    };
    read_x(0, 0);
    OPHIP_GEMM_PRIO_UP();
    for (int kb0 = 0; kb0 < KBLOCKS; kb0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int kb = kb0 + p;
            const int cur = p & 1;                  // PD is even or 1: the parity of kb is the parity of p
            if (kb + 1 < KBLOCKS) read_x(kb + 1, cur ^ 1);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
                    acc[t][tt] = W_IS_A ? mma_bf16<NS>(ring.hi[p][t], ring.lo[p][t], x_hi[cur][tt], x_lo[cur][tt], acc[t][tt])
                                        : mma_bf16<NS>(x_hi[cur][tt], x_lo[cur][tt], ring.hi[p][t], ring.lo[p][t], acc[t][tt]);
            // refill this slot with k-block kb + PD (nothing to fetch at the tail)
            if (kb + PD < KBLOCKS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    ring.hi[p][t] = whi[(size_t)t * tstride + (size_t)(kb + PD) * 64];
                    if (NS == 3) ring.lo[p][t] = wlo[(size_t)t * tstride + (size_t)(kb + PD) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the refill here (see tile.h)
        }
    }
    OPHIP_GEMM_PRIO_DOWN();
}

// Same, with the K range split in two halves read from two plane pairs (the [x, msg] concatenation feeding the MLP):
// k-blocks [0, KBLOCKS/2) from (a_hi, a_lo), the rest from (b_hi, b_lo); both with pitch rowb, chunk 0 first.
template <int NT, int TT, int NS, int KBLOCKS, int PD>
__device__ __forceinline__ void gemm_bf16_ring_cat(f32x16 (&acc)[NT][TT], WRing<NT, PD, NS>& ring, const bf16x8* __restrict__ whi,
                                                   const bf16x8* __restrict__ wlo, int tstride, const char* a_hi, const char* a_lo,
                                                   const char* b_hi, const char* b_lo, int rowb, int lane) {
    static_assert(KBLOCKS % (2 * PD) == 0, "each half must be a multiple of the prefetch depth");
    const int r = lane & 31, h = lane >> 5;
    bf16x8 x_hi[2][TT], x_lo[2][TT];
    auto read_x = [&](int kb, int set) {
        const bool second = kb >= KBLOCKS / 2;
        const char* s_hi = second ? b_hi : a_hi;
        const char* s_lo = second ? b_lo : a_lo;
        const int kk = second ? kb - KBLOCKS / 2 : kb;
        int rr = r;                              // (see gemm_bf16_ring: keeps the offset out of the caller's loop-invariant set)
        asm volatile("" : "+v"(rr));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int off = plane_off(32 * tt + rr, 2 * kk + h, rowb);
            x_hi[set][tt] = *reinterpret_cast<const bf16x8*>(s_hi + off);
            x_lo[set][tt] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(s_lo + off) : zero_bf8();
        }
    };
    read_x(0, 0);
    OPHIP_GEMM_PRIO_UP();
    for (int kb0 = 0; kb0 < KBLOCKS; kb0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int kb = kb0 + p;
            const int cur = p & 1;
            if (kb + 1 < KBLOCKS) read_x(kb + 1, cur ^ 1);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) acc[t][tt] = mma_bf16<NS>(ring.hi[p][t], ring.lo[p][t], x_hi[cur][tt], x_lo[cur][tt], acc[t][tt]);
            if (kb + PD < KBLOCKS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    ring.hi[p][t] = whi[(size_t)t * tstride + (size_t)(kb + PD) * 64];
                    if (NS == 3) ring.lo[p][t] = wlo[(size_t)t * tstride + (size_t)(kb + PD) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    OPHIP_GEMM_PRIO_DOWN();
}

template <int NT, int TT, int NS, bool W_IS_A, int KBLOCKS, int PD>
__device__ __forceinline__ void gemm_bf16(f32x16 (&acc)[NT][TT], const bf16x8* __restrict__ whi, const bf16x8* __restrict__ wlo,
                                          int tstride, const char* act_hi, const char* act_lo, int rowb, int chunk0, int lane, int swz = 15) {
    WRing<NT, PD, NS> ring;
    ring.fill(whi, wlo, tstride);
    gemm_bf16_ring<NT, TT, NS, W_IS_A, KBLOCKS, PD>(acc, ring, whi, wlo, tstride, act_hi, act_lo, rowb, chunk0, lane, swz);
}

// Store a D[feature row][token col] accumulator into the planes: lane (token = tok0 + r, half h) owns features
// feat0 + 8g + 4h + {0..3} in regs 4g..4g+3  ->  one 8-byte store per g and plane.
template <int NS>
__device__ __forceinline__ void store_featrow_acc(const f32x16& a, char* phi, char* plo, int rowb, int feat0, int tok0, int lane, int swz = 15) {
    const int r = lane & 31, h = lane >> 5;
    const int row = tok0 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 hh, ll;
            split_bf16(a[4 * g + j], hh, ll);
            vh[j] = hh;
            vl[j] = ll;
        }
        const int off = plane_off(row, (feat0 >> 3) + g, rowb, swz) + 8 * h;
        *reinterpret_cast<bf16x4*>(phi + off) = vh;
        if (NS == 3) *reinterpret_cast<bf16x4*>(plo + off) = vl;
    }
}

// global f32 rows -> swizzled (hi, lo) planes; T rows of C features, zero fill beyond L
template <int NS, int C, int T>
__device__ __forceinline__ void load_rows_to_planes(char* phi, char* plo, const float* __restrict__ x, int tok0, int L, int tid, int nthreads) {
    constexpr int CH = C / 8;       // 16-byte bf16 chunks per row
    for (int i = tid; i < T * CH; i += nthreads) {
        const int row = i / CH, ch = i % CH;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (tok0 + row < L) {
            const float* src = x + (size_t)(tok0 + row) * C + 8 * ch;
            v0 = *reinterpret_cast<const f32x4*>(src);
            v1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 hh, ll;
            split_bf16(v0[j], hh, ll); vh[j] = hh; vl[j] = ll;
            split_bf16(v1[j], hh, ll); vh[4 + j] = hh; vl[4 + j] = ll;
        }
        const int off = plane_off(row, ch, C * 2);
        *reinterpret_cast<bf16x8*>(phi + off) = vh;
        if (NS == 3) *reinterpret_cast<bf16x8*>(plo + off) = vl;
    }
}
