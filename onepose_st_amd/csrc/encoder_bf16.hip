// Coarse LoFTR encoder layer on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, f32 accumulate), plain or split-bf16
// (tile_bf16.h).  Same three launches per layer and the same mathematics as csrc/encoder.hip (reference:
// loftr_module/transformer.py:65-94,146-159, linear_attention.py:29-61); what changes is the mapping to the machine:
//
//   * an attn_apply workgroup is 4 waves (one per SIMD); wave fw owns feature group fw (64 features = heads 2 fw,
//     2 fw + 1) of the TT 32-token MFMA tiles of the workgroup, so a packed weight fragment pulled from L2 feeds TT
//     matrix instructions.  TT = 1 by default: two independent 32-token workgroups per CU whose barrier / epilogue
//     gaps interleave (c2: 369 workgroups, all resident at once).  Weight rings are filled one phase ahead and the
//     activation fragments are read from LDS one k-block ahead (tile_bf16.h);
//   * operands swap roles in attn_apply: A = packed weights, B = activations, so an accumulator holds
//     D[feature][token] -- four consecutive features of one token per lane and register quad.  Epilogues therefore
//     write 8-byte packed bf16 quads into the [token][feature] LDS planes, LayerNorm reduces over registers (plus a
//     64-float cross-wave exchange), and phi(Q) never leaves registers: the Q accumulator is the B operand of
//     KV^T . phi(Q)^T (accumulator-as-operand, k order fixed up in how kv_sum emits KV);
//   * the MLP hidden layer is produced and consumed in four 128-feature chunks, so the three live tiles (x, msg/merge,
//     hidden chunk) fit the 160 KiB LDS exactly in split mode (swizzled planes, no padding).
#include "tile_bf16.h"
#include <stdlib.h>

namespace {

constexpr int C = 256, NH = 8;
constexpr int ROWB = C * 2;                 // plane row pitch (bytes)
constexpr int HROWB = 128 * 2;              // hidden-chunk plane row pitch
constexpr int KB = C / 16, TS = KB * 64;            // K = 256: 16 k-blocks, fragments per tile
constexpr int KB2 = 2 * C / 16, TS2 = KB2 * 64;     // K = 512
constexpr int KV_PART_FLOATS = NH * 2 * 64 * 8 + NH * 32;       // 8448: KV fragments (f32) + Ksum
constexpr int KV_FRAG_BYTES = NH * 2 * 2 * 64 * 16;             // [head][s][plane][lane][16 B] = 32768
constexpr int KV_BLOCK_BYTES = KV_FRAG_BYTES + NH * 32 * 4;     // + Ksum f32 [head][h][16] = 33792
constexpr int W_ELEMS = 10 * C * C;                             // bf16 elements per plane of a layer block

struct KvRedArgs {
    const float* x[2];
    long long xbs[2];
    int L[2];
    int tiles[2];                   // 64-token workgroup tiles per stream
    int slabs[2];                   // 32-token partial slabs per stream (= ceil(L / 32))
    const bf16x8 *w_hi, *w_lo;      // Wkv fragments
    float* partial;                 // [B][slabs0 + slabs1][KV_PART_FLOATS]
    const unsigned char* mask2d;    // MASKED kernels: [B][L[1]] 1 = real cell, 0 = padding of the 2D stream (linear_attention.py:49-53)
};

// K, V projection of TT 32-token tiles (rows 0 .. 32 TT - 1 of the planes) for heads 2 fw, 2 fw + 1, then each tile's
// KV = phi(K)^T V / Ksum partial slab straight from the accumulators (slab tt at out + tt * KV_PART_FLOATS; tiles that
// start beyond L write nothing).  `ring` must be filled from whi / wlo.
template <int NS, int TT, int HW = 2>      // HW = heads per wave
__device__ __forceinline__ void kv_slab_from_planes(WRing<2 * HW, 2, NS>& ring, const bf16x8* whi, const bf16x8* wlo, const char* xh, const char* xl,
                                                    int tok_base, int L, float* out, int fw, int lane, int tstride = TS, int head0 = -1,
                                                    const unsigned char* mk = nullptr) {
    const int r = lane & 31, h = lane >> 5;
    if (head0 < 0) head0 = 2 * fw;
    // D[token][feature]: t < HW -> K of heads head0 + t;  HW + t -> V of the same heads (tiles `tstride` fragments apart)
    f32x16 acc[2 * HW][TT];
#pragma unroll
    for (int t = 0; t < 2 * HW; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[t][tt] = zero16();
    gemm_bf16_ring<2 * HW, TT, NS, false, KB, 2>(acc, ring, whi, wlo, tstride, xh, xl, ROWB, 0, lane);
    const float inv_len = 1.0f / (float)L;
    const bf16x8 zeros = zero_bf8();
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int tb = tok_base + 32 * tt;
        if (tb >= L) break;                                                                                   // wave-uniform
        auto f_k = [&](int reg, float v) {                                                                    // padded tokens (and masked cells: kv_mask) drop out
            const int tok = tb + acc_row(reg, h);
            return (tok < L && (!mk || mk[tok])) ? elu_plus_one_fast(v) : 0.f;
        };
        auto f_v = [&](int reg, float v) { return v * inv_len; };                                             // values / v_length
        float* outt = out + (size_t)tt * KV_PART_FLOATS;
#pragma unroll
        for (int t = 0; t < HW; ++t) {
            f32x16 kv = zero16(), ks = zero16();
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 khi, klo, vhi, vlo;
                acc_frag_map<NS>(acc[t][tt], st, f_k, khi, klo);
                acc_frag_map<NS>(acc[HW + t][tt], st, f_v, vhi, vlo);
                kv = mma_bf16<NS>(khi, klo, vhi, vlo, kv);        // KV[d][v] += sum_tok phi(K)[tok][d] V[tok][v]
                ks = mma_bf16<NS>(khi, klo, ones, zeros, ks);     // Ksum[d] replicated over v
            }
            const int head = head0 + t;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float* o = outt + ((size_t)(head * 2 + st) * 64 + lane) * 8;
                f32x4 v0 = {kv[8 * st], kv[8 * st + 1], kv[8 * st + 2], kv[8 * st + 3]};
                f32x4 v1 = {kv[8 * st + 4], kv[8 * st + 5], kv[8 * st + 6], kv[8 * st + 7]};
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            }
            if (r == 0) {
                float* o = outt + NH * 2 * 64 * 8 + head * 32 + h * 16;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[reg] = ks[reg];
            }
        }
    }
}

template <int NS, int TOK, bool MASKED = false>
__global__ __launch_bounds__(TOK * 8) OPHIP_WAVES_PER_SIMD(1, 2) void kv_reduce_bf16_kernel(KvRedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = NS == 3 ? 2 : 1;
    constexpr int NT_ = TOK * 8;                     // threads: 4 waves per 32-token tile
    char* XH = smem;
    char* XL = smem + (PL - 1) * TOK * ROWB;
    // wave = 4 * tt + fw owns heads 2 fw, 2 fw + 1 (K and V tiles) of token tile tt -> one partial slab per 32 tokens
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fw = wave & 3, tt = wave >> 2;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * TOK;
    WRing<4, 2, NS> ring;
    const bf16x8* whi = a.w_hi + (size_t)(4 * fw) * TS + lane;
    const bf16x8* wlo = a.w_lo + (size_t)(4 * fw) * TS + lane;
    ring.fill(whi, wlo, TS);                         // weights travel while the activation tile is staged
    load_rows_to_planes<NS, C, TOK>(XH, XL, a.x[s] + (size_t)b * a.xbs[s], tok0, L, tid, NT_);
    __syncthreads();
    if (tok0 + 32 * tt >= L) return;                 // second half of a ragged last tile: no tokens, no slab

    float* out = a.partial + ((size_t)b * (a.slabs[0] + a.slabs[1]) + (s ? a.slabs[0] : 0) + (TOK / 32) * lt + tt) * KV_PART_FLOATS;
    const unsigned char* mk = (MASKED && s == 1) ? a.mask2d + (size_t)b * L : nullptr;
    kv_slab_from_planes<NS, 1>(ring, whi, wlo, XH + 32 * tt * ROWB, XL + 32 * tt * ROWB, tok0 + 32 * tt, L, out, fw, lane, TS, -1, mk);
}

struct KvSumBArgs {
    const float* partial;
    char* kv;              // [B][2][KV_BLOCK_BYTES]
    int tiles[2];
};

constexpr int KVS_G = 16;

// fixed-order sum of the per-tile partials; emits KV as (hi, lo) bf16 A-fragments and Ksum as f32
__global__ __launch_bounds__(1024) void kv_sum_bf16_kernel(KvSumBArgs a) {
    __shared__ float red[KVS_G][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int s = blockIdx.y & 1, b = blockIdx.y >> 1;
    const int ttot = a.tiles[0] + a.tiles[1];
    const int t0 = s ? a.tiles[0] : 0, nt = a.tiles[s];
    const int e = blockIdx.x * 64 + o;
    const float* p = a.partial + ((size_t)b * ttot + t0) * KV_PART_FLOATS + e;
    float acc = 0.f;
    for (int t = g; t < nt; t += KVS_G) acc += p[(size_t)t * KV_PART_FLOATS];
    red[g][o] = acc;
    __syncthreads();
    if (g == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < KVS_G; ++q) tot += red[q][o];
        char* blk = a.kv + ((size_t)b * 2 + s) * KV_BLOCK_BYTES;
        if (e < NH * 2 * 64 * 8) {
            const int j = e & 7, ln = (e >> 3) & 63, hs = e >> 9;      // hs = head * 2 + s
            __bf16 hh, ll;
            split_bf16(tot, hh, ll);
            *reinterpret_cast<__bf16*>(blk + ((size_t)(hs * 2 + 0) * 64 + ln) * 16 + j * 2) = hh;
            *reinterpret_cast<__bf16*>(blk + ((size_t)(hs * 2 + 1) * 64 + ln) * 16 + j * 2) = ll;
        } else {
            reinterpret_cast<float*>(blk + KV_FRAG_BYTES)[e - NH * 2 * 64 * 8] = tot;
        }
    }
}

struct AttnBArgs {
    const float* x[2];
    float* y[2];
    long long xbs[2], ybs[2];
    int L[2];
    int tiles[2];
    const char* kv[2];
    long long kvbs;            // bytes
    float srclen[2];
    const bf16x8 *w_hi, *w_lo; // layer block planes (Wq | Wkv | Wm | W0 | W2)
    const float* ln;           // g1 b1 g2 b2
    // fused kv_reduce of the NEXT layer (NULL: not fused): its Wkv fragments and partial slabs
    const bf16x8 *nkv_hi, *nkv_lo;
    float* npartial;
    int slabs[2];
    unsigned long long* stamps;
    const unsigned char* mask2d;   // MASKED kernels: q_mask of the 2D stream (see KvRedArgs)
};

// LayerNorm over the 256 features of a token held as D[feature][token] accumulators by the 4 feature-group waves
// (wave fw: features 64 fw .. 64 fw + 63 of all TT token tiles).  Two-pass; partial sums cross waves through `scratch`
// ([2][4][64] floats).  Contains 2 workgroup barriers.
template <int TT>
__device__ __forceinline__ void layernorm_featrow(f32x16 (&m)[2][TT], const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* scratch, int fw, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s += m[t][tt][reg];
        s += __shfl_xor(s, 32, 64);
        if (h == 0) scratch[fw * 64 + 32 * tt + r] = s;
    }
    __syncthreads();
    float mean[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int tok = 32 * tt + r;
        mean[tt] = ((scratch[tok] + scratch[64 + tok]) + (scratch[128 + tok] + scratch[192 + tok])) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float d = m[t][tt][reg] - mean[tt];
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        if (h == 0) scratch[256 + fw * 64 + tok] = q;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int tok = 32 * tt + r;
        const float var = ((scratch[256 + tok] + scratch[320 + tok]) + (scratch[384 + tok] + scratch[448 + tok])) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = 64 * fw + 32 * t + 8 * g + 4 * h;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + f0);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(beta + f0);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[t][tt][4 * g + j] = (m[t][tt][4 * g + j] - mean[tt]) * rstd * gv[j] + bv[j];
            }
    }
}

template <int NS, int TT, bool MASKED = false>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(TT == 1 ? 2 : 1, TT == 1 ? 2 : 1) void attn_apply_bf16_kernel(AttnBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = NS == 3 ? 2 : 1;
    constexpr int TOK = 32 * TT;
    constexpr int NT_ = 256;
    constexpr int XB = TOK * ROWB, HB = TOK * HROWB;
    char* XH = smem;
    char* XL = smem + (PL - 1) * XB;
    char* YH = smem + PL * XB;
    char* YL = YH + (PL - 1) * XB;
    char* HH = smem + 2 * PL * XB;
    char* HL = HH + (PL - 1) * HB;
    float* scratch = reinterpret_cast<float*>(HH);          // LayerNorm exchange; H is idle whenever a LayerNorm runs
    // 4 waves: wave fw owns feature group fw (features 64 fw .. 64 fw + 63 = heads 2 fw, 2 fw + 1) of all TT token tiles
    const int tid = threadIdx.x, lane = tid & 63, fw = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * TOK;
    const float* xg = a.x[s] + (size_t)b * a.xbs[s];
    const unsigned char* mk = (MASKED && s == 1) ? a.mask2d + (size_t)b * L : nullptr;

    const bf16x8 *wq_hi = a.w_hi + (size_t)(2 * fw) * TS + lane, *wq_lo = a.w_lo + (size_t)(2 * fw) * TS + lane;
    const bf16x8 *wm_hi = a.w_hi + 3 * C * C / 8 + (size_t)(2 * fw) * TS + lane, *wm_lo = a.w_lo + 3 * C * C / 8 + (size_t)(2 * fw) * TS + lane;
    const bf16x8 *w0_hi = a.w_hi + 4 * C * C / 8 + lane, *w0_lo = a.w_lo + 4 * C * C / 8 + lane;
    const bf16x8 *w2_hi = a.w_hi + 8 * C * C / 8 + (size_t)(2 * fw) * TS2 + lane, *w2_lo = a.w_lo + 8 * C * C / 8 + (size_t)(2 * fw) * TS2 + lane;

    WRing<2, 4, NS> rq;
    rq.fill(wq_hi, wq_lo, TS);                       // weights travel while the activation tile is staged
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    OPHIP_STAMP(a.stamps, wg, 0);
    load_rows_to_planes<NS, C, TOK>(XH, XL, xg, tok0, L, tid, NT_);
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 1);

    // ---- Q projection (heads 2fw, 2fw+1), phi, linear attention from registers -----------------------------
    WRing<2, 4, NS> rm;
    {
        f32x16 q[2][TT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) q[t][tt] = zero16();
        gemm_bf16_ring<2, TT, NS, true, KB, 4>(q, rq, wq_hi, wq_lo, TS, XH, XL, ROWB, 0, lane);
        OPHIP_STAMP(a.stamps, wg, 2);
        rm.fill(wm_hi, wm_lo, TS);                   // merge weights: in flight during the attention below
        const char* kvb = a.kv[s] + (size_t)b * a.kvbs;
        const float* ksum = reinterpret_cast<const float*>(kvb + KV_FRAG_BYTES);
        const float S = a.srclen[s];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int head = 2 * fw + t;
            f32x16 num[TT], den[TT];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                num[tt] = zero16(); den[tt] = zero16();
                bool live = true;                    // q_mask: phi(Q) = 0 for a padded cell => its message is 0
                if (MASKED && mk) { const int tok = tok0 + 32 * tt + r; live = tok >= L || mk[tok]; }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) q[t][tt][reg] = live ? elu_plus_one_fast(q[t][tt][reg]) : 0.f;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 kvh = *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((head * 2 + st) * 2 + 0) * 64 + lane) * 16);
                const bf16x8 kvl = (NS == 3) ? *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((head * 2 + st) * 2 + 1) * 64 + lane) * 16) : zero_bf8();
                const float* kp = ksum + head * 32 + h * 16 + 8 * st;
                bf16x8 ksh, ksl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hh2, ll2;
                    split_bf16(kp[j], hh2, ll2);
                    ksh[j] = hh2;
                    ksl[j] = (NS == 3) ? ll2 : (__bf16)0.f;
                }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    bf16x8 qh, ql;
                    acc_frag<NS>(q[t][tt], st, qh, ql);
                    num[tt] = mma_bf16<NS>(kvh, kvl, qh, ql, num[tt]);      // num^T[v][tok] = sum_d KV[d][v] phiQ[tok][d]
                    den[tt] = mma_bf16<NS>(ksh, ksl, qh, ql, den[tt]);      // den[tok] replicated over v
                }
            }
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) num[tt][reg] = num[tt][reg] * rcp_fast(den[tt][reg] + 1e-6f) * S;
                store_featrow_acc<NS>(num[tt], YH, YL, ROWB, 32 * head, 32 * tt, lane);
            }
        }
    }
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 3);
    // ---- merge + LayerNorm 1 -> Y ----------------------------------------------------------------
    WRing<1, 4, NS> r0;
    {
        f32x16 m[2][TT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) m[t][tt] = zero16();
        gemm_bf16_ring<2, TT, NS, true, KB, 4>(m, rm, wm_hi, wm_lo, TS, YH, YL, ROWB, 0, lane);
        OPHIP_STAMP(a.stamps, wg, 4);
        r0.fill(w0_hi + (size_t)fw * TS2, w0_lo + (size_t)fw * TS2, TS2);          // MLP-up weights of chunk 0
        layernorm_featrow<TT>(m, a.ln, a.ln + C, scratch, fw, lane);     // its first barrier also fences the reads of Y above
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) store_featrow_acc<NS>(m[t][tt], YH, YL, ROWB, 64 * fw + 32 * t, 32 * tt, lane);
    }
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 5);
    // ---- MLP: hidden = relu([x, msg] W0^T) in four 128-feature chunks, o += hidden_chunk W2[:, chunk]^T ---------
    f32x16 o[2][TT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) o[t][tt] = zero16();
    for (int c = 0; c < 4; ++c) {
        f32x16 hd[1][TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) hd[0][tt] = zero16();
        const size_t wt = (size_t)(4 * c + fw) * TS2;
        gemm_bf16_ring_cat<1, TT, NS, KB2, 4>(hd, r0, w0_hi + wt, w0_lo + wt, TS2, XH, XL, YH, YL, ROWB, lane);
        OPHIP_STAMP(a.stamps, wg, 6 + 4 * c);
        WRing<2, 4, NS> r2;
        r2.fill(w2_hi + (size_t)(8 * c) * 64, w2_lo + (size_t)(8 * c) * 64, TS2);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) hd[0][tt][reg] = fmaxf(hd[0][tt][reg], 0.f);
            store_featrow_acc<NS>(hd[0][tt], HH, HL, HROWB, 32 * fw, 32 * tt, lane);
        }
        __syncthreads();
        OPHIP_STAMP(a.stamps, wg, 7 + 4 * c);
        gemm_bf16_ring<2, TT, NS, true, 8, 4>(o, r2, w2_hi + (size_t)(8 * c) * 64, w2_lo + (size_t)(8 * c) * 64, TS2, HH, HL, HROWB, 0, lane);
        OPHIP_STAMP(a.stamps, wg, 8 + 4 * c);
        if (c + 1 < 4) r0.fill(w0_hi + wt + (size_t)4 * TS2, w0_lo + wt + (size_t)4 * TS2, TS2);
        __syncthreads();
        OPHIP_STAMP(a.stamps, wg, 9 + 4 * c);
    }
    layernorm_featrow<TT>(o, a.ln + 2 * C, a.ln + 3 * C, scratch, fw, lane);
    OPHIP_STAMP(a.stamps, wg, 22);
    // ---- stage LN2 output as f32 [TOK][256] over the (now dead) X / Y planes, then x + msg with whole-row stores ----
    char* stage = smem;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 32 * tt + r;
                const int ch = 16 * fw + 8 * t + 2 * g + h;          // 16-byte chunk of the f32 row
                f32x4 v = {o[t][tt][4 * g], o[t][tt][4 * g + 1], o[t][tt][4 * g + 2], o[t][tt][4 * g + 3]};
                *reinterpret_cast<f32x4*>(stage + row * (C * 4) + ((ch ^ (row & 15)) << 4)) = v;
            }
    __syncthreads();
    float* yg = a.y[s] + (size_t)b * a.ybs[s];
    const bool fuse = a.nkv_hi != nullptr;
    // next layer's K|V weights start travelling now (consumed after the barrier below)
    WRing<4, 2, NS> rkv;
    const bf16x8* nhi = a.nkv_hi + (size_t)(4 * fw) * TS + lane;
    const bf16x8* nlo = a.nkv_lo + (size_t)(4 * fw) * TS + lane;
    if (fuse) rkv.fill(nhi, nlo, TS);
    // planes of the OUTPUT tile for the fused kv_reduce: behind the f32 stage (= the Y region in split mode)
    char* KH = smem + TOK * C * 4;
    char* KL = KH + (PL - 1) * XB;
    for (int i = tid; i < TOK * (C / 4); i += NT_) {
        const int row = i / (C / 4), ch = i % (C / 4);
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (tok0 + row < L) {
            const f32x4 mv = *reinterpret_cast<const f32x4*>(stage + row * (C * 4) + ((ch ^ (row & 15)) << 4));
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xg + (size_t)(tok0 + row) * C + 4 * ch);
            yv = xv + mv;
            *reinterpret_cast<f32x4*>(yg + (size_t)(tok0 + row) * C + 4 * ch) = yv;
        }
        if (fuse) {                                   // rows beyond L stay zero: they drop out of phi(K) and V
            bf16x4 vh, vl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 hh, ll;
                split_bf16(yv[j], hh, ll);
                vh[j] = hh; vl[j] = ll;
            }
            const int off = plane_off(row, ch >> 1, ROWB) + 8 * (ch & 1);
            *reinterpret_cast<bf16x4*>(KH + off) = vh;
            if (NS == 3) *reinterpret_cast<bf16x4*>(KL + off) = vl;
        }
    }
    OPHIP_STAMP(a.stamps, wg, 30);
    if (fuse) {
        // ---- kv_reduce of the next layer on the tile that is still on chip (saves a launch and a re-read of the streams) ----
        __syncthreads();
        float* out = a.npartial + ((size_t)b * (a.slabs[0] + a.slabs[1]) + (s ? a.slabs[0] : 0) + TT * lt) * KV_PART_FLOATS;
        kv_slab_from_planes<NS, TT>(rkv, nhi, nlo, KH, KL, tok0, L, out, fw, lane, TS, -1, mk);
    }
    OPHIP_STAMP(a.stamps, wg, 31);
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char* what) {
    return ophip_lds_attr(reinterpret_cast<const void*>(kernel), bytes, what);
}

}  // namespace

extern "C" size_t ophip_encoder_bf16_workspace_bytes(int B, int L3d, int L2d) {
    const size_t slabs = (size_t)((L3d + 31) / 32 + (L2d + 31) / 32);
    return 2 * (size_t)B * slabs * KV_PART_FLOATS * 4 + (size_t)B * 2 * KV_BLOCK_BYTES + 256;      // two slab sets (ping-pong) + KV
}

extern "C" size_t ophip_encoder_bf16_wpack_bytes(void) { return (size_t)2 * W_ELEMS * 2 + 4 * C * 4; }

namespace {
int layer_bf16(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
               const void* wpack, const void* wpack_next, int nsplit, int is_cross, int kv_from_prev, int slot,
               void* workspace, const unsigned char* mask2d, void* stream_) {
    if (!x3d || !x2d || !y3d || !y2d || !wpack || !workspace) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || L3d < 1 || L2d < 1) return ophip_bad_arg(__func__, "B, L3d, L2d must be >= 1");
    if (nsplit != 1) return ophip_bad_arg(__func__, "nsplit must be 1: this is the plain-bf16 mode's layer (split-bf16: ophip_encoder_layer_x3w8)");
    if (slot != 0 && slot != 1) return ophip_bad_arg(__func__, "slot must be 0 or 1");
    if (x3d == y3d || x2d == y2d) return ophip_bad_arg(__func__, "in-place layer is not supported (cross layers read the pre-update streams)");
    hipStream_t stream = (hipStream_t)stream_;
    // 32-token workgroups, two per CU (TT = 1); kv_reduce works on 32-token tiles as well
    constexpr int TT = 1;
    const int TOK = 32 * TT;
    const int t3 = (L3d + TOK - 1) / TOK, t2 = (L2d + TOK - 1) / TOK;
    const int s3 = (L3d + 31) / 32, s2 = (L2d + 31) / 32;
    const size_t part_floats = (size_t)B * (s3 + s2) * KV_PART_FLOATS;
    float* partial = reinterpret_cast<float*>(workspace) + (size_t)slot * part_floats;          // this layer's slabs
    float* partial_next = reinterpret_cast<float*>(workspace) + (size_t)(slot ^ 1) * part_floats;
    char* kv = reinterpret_cast<char*>(workspace) + 2 * part_floats * 4;
    kv += (256 - (reinterpret_cast<uintptr_t>(kv) & 255)) & 255;
    // layer block: [hi plane: Wq | Wkv | Wm | W0 | W2][lo plane: same][g1 b1 g2 b2 f32]   (packing.pack_coarse_layer_bf16)
    const bf16x8* w_hi = reinterpret_cast<const bf16x8*>(wpack);
    const bf16x8* w_lo = w_hi + W_ELEMS / 8;
    const float* ln = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + (size_t)2 * W_ELEMS * 2);
    const int PL = nsplit == 3 ? 2 : 1;

    if (!kv_from_prev) {
        KvRedArgs ka;
        ka.x[0] = x3d; ka.x[1] = x2d;
        ka.xbs[0] = (long long)L3d * C; ka.xbs[1] = (long long)L2d * C;
        ka.L[0] = L3d; ka.L[1] = L2d; ka.tiles[0] = s3; ka.tiles[1] = s2; ka.slabs[0] = s3; ka.slabs[1] = s2;
        ka.w_hi = w_hi + C * C / 8; ka.w_lo = w_lo + C * C / 8;
        ka.partial = partial;
        ka.mask2d = mask2d;
        const size_t lds_kv = (size_t)PL * 32 * ROWB;
#define OPHIP_KV_CASE(NS_)                                                                                                       \
        {                                                                                                                        \
            if (mask2d) {                                                                                                        \
                if (int rc = set_lds(kv_reduce_bf16_kernel<NS_, 32, true>, lds_kv, "hipFuncSetAttribute(kv_reduce_bf16 masked)")) return rc; \
                OPHIP_LAUNCH("kv_reduce", stream, (kv_reduce_bf16_kernel<NS_, 32, true>), dim3(s3 + s2, B), dim3(256), lds_kv, stream, ka);  \
            } else {                                                                                                             \
                if (int rc = set_lds(kv_reduce_bf16_kernel<NS_, 32>, lds_kv, "hipFuncSetAttribute(kv_reduce_bf16)")) return rc;  \
                OPHIP_LAUNCH("kv_reduce", stream, (kv_reduce_bf16_kernel<NS_, 32>), dim3(s3 + s2, B), dim3(256), lds_kv, stream, ka); \
            }                                                                                                                    \
        }
        OPHIP_KV_CASE(1)
#undef OPHIP_KV_CASE
        OPHIP_CHECK_LAUNCH();
    }

    KvSumBArgs sa;
    sa.partial = partial; sa.kv = kv; sa.tiles[0] = s3; sa.tiles[1] = s2;
    OPHIP_LAUNCH("kv_sum", stream, kv_sum_bf16_kernel, dim3(KV_PART_FLOATS / 64, 2 * B), dim3(1024), 0, stream, sa);
    OPHIP_CHECK_LAUNCH();

    AttnBArgs aa;
    aa.x[0] = x3d; aa.x[1] = x2d; aa.y[0] = y3d; aa.y[1] = y2d;
    aa.xbs[0] = aa.ybs[0] = (long long)L3d * C; aa.xbs[1] = aa.ybs[1] = (long long)L2d * C;
    aa.L[0] = L3d; aa.L[1] = L2d; aa.tiles[0] = t3; aa.tiles[1] = t2;
    aa.kv[0] = kv + (is_cross ? KV_BLOCK_BYTES : 0);
    aa.kv[1] = kv + (is_cross ? 0 : KV_BLOCK_BYTES);
    aa.kvbs = 2LL * KV_BLOCK_BYTES;
    aa.srclen[0] = (float)(is_cross ? L2d : L3d);
    aa.srclen[1] = (float)(is_cross ? L3d : L2d);
    aa.w_hi = w_hi; aa.w_lo = w_lo; aa.ln = ln;
    aa.nkv_hi = aa.nkv_lo = nullptr; aa.npartial = partial_next; aa.slabs[0] = s3; aa.slabs[1] = s2;
    if (wpack_next) {
        const bf16x8* n_hi = reinterpret_cast<const bf16x8*>(wpack_next);
        aa.nkv_hi = n_hi + C * C / 8;
        aa.nkv_lo = n_hi + W_ELEMS / 8 + C * C / 8;
    }
    aa.stamps = ophip_stamp_buffer();
    aa.mask2d = mask2d;
    // LDS: X, Y planes + hidden chunk; the fused tail needs the output planes behind the 64 KiB f32 stage
    size_t lds_at = (size_t)PL * (2 * TOK * ROWB + TOK * HROWB);
    const size_t lds_fuse = (size_t)TOK * C * 4 + (size_t)PL * TOK * ROWB;
    if (lds_fuse > lds_at) lds_at = lds_fuse;
#define OPHIP_AT_CASE(NS_, TT_)                                                                                                  \
    {                                                                                                                            \
        if (mask2d) {                                                                                                            \
            if (int rc = set_lds(attn_apply_bf16_kernel<NS_, TT_, true>, lds_at, "hipFuncSetAttribute(attn_apply_bf16 masked)")) return rc; \
            OPHIP_LAUNCH("attn_apply", stream, (attn_apply_bf16_kernel<NS_, TT_, true>), dim3(t3 + t2, B), dim3(256), lds_at, stream, aa); \
        } else {                                                                                                                 \
            if (int rc = set_lds(attn_apply_bf16_kernel<NS_, TT_>, lds_at, "hipFuncSetAttribute(attn_apply_bf16)")) return rc;   \
            OPHIP_LAUNCH("attn_apply", stream, (attn_apply_bf16_kernel<NS_, TT_>), dim3(t3 + t2, B), dim3(256), lds_at, stream, aa); \
        }                                                                                                                        \
    }
    OPHIP_AT_CASE(1, 1)
#undef OPHIP_AT_CASE
    OPHIP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int ophip_encoder_layer_bf16(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                        const void* wpack, const void* wpack_next, int nsplit, int is_cross, int kv_from_prev, int slot,
                                        void* workspace, void* stream) {
    return layer_bf16(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, nsplit, is_cross, kv_from_prev, slot, workspace, nullptr, stream);
}

// The same layer with the reference's query_mask (transformer.py:148-159): mask2d [B][L2d], 1 = real cell, 0 = padding.
extern "C" int ophip_encoder_layer_bf16_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                               const void* wpack, const void* wpack_next, int nsplit, int is_cross, int kv_from_prev, int slot,
                                               void* workspace, const unsigned char* mask2d, void* stream) {
    if (!mask2d) return ophip_bad_arg(__func__, "null mask (use ophip_encoder_layer_bf16)");
    return layer_bf16(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, nsplit, is_cross, kv_from_prev, slot, workspace, mask2d, stream);
}
