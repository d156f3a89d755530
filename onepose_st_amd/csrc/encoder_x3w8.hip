// Coarse LoFTR encoder layer, split-bf16, 48-token workgroups of EIGHT waves (two per SIMD).  Same mathematics as the exact-f32 layer
// csrc/encoder.hip (reference: loftr_module/transformer.py:65-94,146-159, linear_attention.py:29-61); K / V partial slabs of
// KV_PART_FLOATS floats per workgroup, summed in fixed order by kv_sum_w8_kernel below.  (Round 2's four-wave kernel, whose file this
// one replaced in round 3, had the same data layout; what the eight-wave form changed is the split of a workgroup's work over waves:)
//
//   * wave fw = head fw owns 32 of the 256 output features (two 16-row feature tiles) of all three token tiles, so every weight
//     fragment still enters the CU once -- but a SIMD now holds two waves, and while one of them is stuck issuing a 1 KiB buffer
//     load (~12 issue cycles next to a 16-cycle 16x16x32 MFMA, tools/micro/gemm16_rate.hip) or runs an epilogue on the vector
//     ALU, its partner keeps the matrix pipe fed (tools/micro/gemm16_2w.hip: 257 against 293 us for the bare k-step loop);
//   * the MLP's hidden layer goes in two 256-wide chunks through ONE hidden buffer (each wave produces 32 hidden features per
//     chunk: with 128-wide chunks a wave would own a single feature tile and read 1 KiB of activations per 3 MFMAs);
//   * per-wave streams are 256 + 64 KiB (packing.pack_coarse_layer_x3w8): Q | merge | W0c0 | W2c0 | W0c1 | W2c1, then the next
//     layer's K|V of head fw.
#include "tile_x3.h"
#include "x3w8_internal.h"
#include <stdlib.h>

namespace {

using namespace x3;

constexpr int C = 256, NH = 8, NW = 8;
constexpr int NTT = 3, TOK = 16 * NTT;
constexpr int ROWB = C * 2;                     // X / Y / hidden plane pitch (512 B, 32 chunks)
constexpr int PLANE = TOK * ROWB;               // 24 576 B
constexpr int MAIN_FRAGS = 256, KV_FRAGS = 64;  // per wave: Q 32 | merge 32 | W0c0 64 | W2c0 32 | W0c1 64 | W2c1 32 ;  K|V 64
constexpr int KV_PART_FLOATS = NH * 1024 + NH * 32;
constexpr int KV_FRAG_BYTES = NH * 2 * 2 * 64 * 16;
constexpr int KV_BLOCK_BYTES = KV_FRAG_BYTES + NH * 32 * 4;
constexpr int LDS_BYTES = 6 * PLANE + NW * TOK * 2 * 4;       // X, Y, hidden planes (hi, lo) + LayerNorm scratch = 150 528 B

struct EncW8Args {
    const float* x[2];
    float* y[2];
    long long xbs[2], ybs[2];
    int L[2];
    int tiles[2];
    const char* kv[2];
    long long kvbs[2];         // per stream: batch stride of its K^T V block in bytes (0: one cached block shared by the whole batch)
    float srclen[2];
    const bf16x8* wmain;       // [8 waves][MAIN_FRAGS][64]
    const bf16x8* wkv;         // [8 waves][KV_FRAGS][64] consumed by the tail; NULL: no tail
    const float* ln;
    float* partial;
    unsigned long long* stamps;
    char* frag[2];             // optional: the output rows also as the similarity kernel's operand fragments (csrc/coarse_match.hip:
    int frag_rows[2];          // frag_planes layout, rows padded to frag_rows = a multiple of 128); NULL: not written
    const unsigned char* mask2d;   // MASKED kernels: [B][L[1]] 1 = real cell, 0 = padding of the 2D stream (linear_attention.py:49-53)
};

__device__ __forceinline__ int stash_off(int row, int chunk) { return row * (C * 4) + ((chunk ^ (row & 15)) << 4); }

// merge the eight waves' moments of token 16 tt + c16 -> mean, 1 / sqrt(var + eps)   (biased variance over 256 features)
__device__ __forceinline__ void merged_stats(const float* scratch, int tt, int c16, float& mean, float& rstd) {
    float sw[NW], dw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(scratch + (w * TOK + 16 * tt + c16) * 2);
        sw[w] = v.x; dw[w] = v.y;
    }
    mean = (((sw[0] + sw[1]) + (sw[2] + sw[3])) + ((sw[4] + sw[5]) + (sw[6] + sw[7]))) * (1.0f / C);
    float m2 = ((dw[0] + dw[1]) + (dw[2] + dw[3])) + ((dw[4] + dw[5]) + (dw[6] + dw[7]));
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float d = sw[w] * (1.0f / 32) - mean;
        m2 += 32.f * d * d;
    }
    rstd = 1.0f / sqrtf(m2 * (1.0f / C) + 1e-5f);
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

// K|V projection of the 48 tokens in the X planes for head fw and its phi(K)^T V / Ksum slab slice
// (mk: the stream's padding mask or NULL -- kv_mask: a padded cell's phi(K) row is zero, and with it its K^T V term)
__device__ __forceinline__ void kv_tail(Ring& ring, const WStream& wsk, const char* XH, const char* XL, int tok0, int L, float* __restrict__ out,
                                        int fw, int lane, const unsigned char* mk = nullptr) {
    // This function is inlined into two kernels (the fused tail of attn_apply and the stand-alone kv_reduce) whose slabs must agree bit for
    // bit (test_encoder_x3_chain_with_fused_kv_tail).  Under hipcc's default -ffp-contract=fast the product `v * inv_len` below may or may
    // not be fused into the subtraction of the (hi, lo) split that follows it, kernel by kernel (round 5: the re-ordered attn_apply took
    // the other choice and the two slabs differed by an ulp of their inputs): no contraction of this function's own arithmetic
#pragma clang fp contract(off)
    const int c16 = lane & 15, q = lane >> 4;
    f32x4 kk[4][NTT];            // D[token 4q + r][feature c16]: ft 0, 1 = K of head fw, ft 2, 3 = V
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) kk[ft][tt] = zero4();
    gemm_stage<NTT, 4, 8, false>(kk, ring, wsk, R, XH, XL, ROWB, 0, c16, q);
    const float inv_len = 1.0f / (float)L;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = tok0 + 16 * tt + 4 * q + r;
                kk[ft][tt][r] = (tok < L && (!mk || mk[tok])) ? elu_plus_one_fast(kk[ft][tt][r]) : 0.f;
                kk[2 + ft][tt][r] *= inv_len;
            }
    const f32x4 z4 = zero4();
    const int head = fw;
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        split8(kk[t][0], kk[t][1], ah[t][0], al[t][0]);
        split8(kk[t][2], z4, ah[t][1], al[t][1]);
        split8(kk[2 + t][0], kk[2 + t][1], bh[t][0], bl[t][0]);
        split8(kk[2 + t][2], z4, bh[t][1], bl[t][1]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            f32x4 kvt = zero4();
            kvt = mma16x3(ah[dt][0], al[dt][0], bh[vt][0], bl[vt][0], kvt);
            kvt = mma16x3(ah[dt][1], al[dt][1], bh[vt][1], bl[vt][1], kvt);
            *reinterpret_cast<f32x4*>(out + ((size_t)((head * 2 + dt) * 2 + vt) * 64 + lane) * 4) = kvt;
        }
        float s = 0.f;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) s += (kk[dt][tt][0] + kk[dt][tt][1]) + (kk[dt][tt][2] + kk[dt][tt][3]);
        s = sum_over_q(s);
        if (q == 0) out[NH * 1024 + head * 32 + 16 * dt + c16] = s;
    }
}

template <bool ONLY_KV, bool MASKED = false>
__global__ __launch_bounds__(512) OPHIP_WAVES_PER_SIMD(2, 2) void enc_x3w8_kernel(EncW8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;
    char* XL = smem + PLANE;
    char* YH = smem + 2 * PLANE;
    char* YL = smem + 3 * PLANE;
    char* HH = smem + 4 * PLANE;
    char* HL = smem + 5 * PLANE;
    float* scratch = reinterpret_cast<float*>(smem + 6 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, fw = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * TOK;
    const float* xg = a.x[s] + (size_t)b * a.xbs[s];
    const unsigned char* mk = (MASKED && s == 1) ? a.mask2d + (size_t)b * L : nullptr;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    float* slab = a.partial + ((size_t)b * (a.tiles[0] + a.tiles[1]) + tile) * KV_PART_FLOATS;
    const int fwu = __builtin_amdgcn_readfirstlane(fw);
    const bool tail = a.wkv != nullptr;
    WStream wsk;
    wsk.open(tail ? a.wkv + (size_t)fwu * KV_FRAGS * 64 : a.wmain, tail ? KV_FRAGS : 0, lane);
    Ring ring;
    OPHIP_STAMP(a.stamps, wg, 0);
    OPHIP_STAMP_REAL(a.stamps, wg, 30);

    // activation rows: 48 x 256 f32, (row, 8-feature chunk) items over 512 threads
    constexpr int ITEMS = TOK * (C / 8) / 512;       // 3
    f32x4 v0[ITEMS], v1[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = tid + 512 * i, row = it >> 5, ch = it & 31;
        v0[i] = v1[i] = zero4();
        if (tok0 + row < L) {
            const float* src = xg + (size_t)(tok0 + row) * C + 8 * ch;
            v0[i] = *reinterpret_cast<const f32x4*>(src);
            v1[i] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    }
    WStream wsm;
    wsm.open(a.wmain + (size_t)fwu * MAIN_FRAGS * 64, MAIN_FRAGS, lane);
    if (ONLY_KV) {
#pragma unroll
        for (int i = 0; i < R; ++i) ring.s[i] = wsk.load(i);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) ring.s[i] = wsm.load(i);
    }
    // attention state of head fw and the LayerNorm parameters of this lane's features
    bf16x8 kvh[2], kvl[2];
    f32x4 ksm[2], g1[2], b1[2], g2[2], b2[2];
    if (!ONLY_KV) {
        const char* kvb = a.kv[s] + (size_t)b * a.kvbs[s];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
            kvh[vt] = *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((fw * 2 + vt) * 2 + 0) * 64 + lane) * 16);
            kvl[vt] = *reinterpret_cast<const bf16x8*>(kvb + ((size_t)((fw * 2 + vt) * 2 + 1) * 64 + lane) * 16);
        }
        const float* kp = reinterpret_cast<const float*>(kvb + KV_FRAG_BYTES) + fw * 32 + 4 * q;
        ksm[0] = *reinterpret_cast<const f32x4*>(kp);
        ksm[1] = *reinterpret_cast<const f32x4*>(kp + 16);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const int f0 = 32 * fw + 16 * ft + 4 * q;
            g1[ft] = *reinterpret_cast<const f32x4*>(a.ln + f0);
            b1[ft] = *reinterpret_cast<const f32x4*>(a.ln + C + f0);
            g2[ft] = *reinterpret_cast<const f32x4*>(a.ln + 2 * C + f0);
            b2[ft] = *reinterpret_cast<const f32x4*>(a.ln + 3 * C + f0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int it = tid + 512 * i, row = it >> 5, ch = it & 31;
        bf16x8 vh, vl;
        split8(v0[i], v1[i], vh, vl);
        const int off = row * ROWB + ((ch ^ (row & 15)) << 4);
        *reinterpret_cast<bf16x8*>(XH + off) = vh;
        *reinterpret_cast<bf16x8*>(XL + off) = vl;
        if (!ONLY_KV) {                               // exact f32 copy for the residual, parked in the idle hidden planes
            *reinterpret_cast<f32x4*>(HH + stash_off(row, 2 * ch)) = v0[i];
            *reinterpret_cast<f32x4*>(HH + stash_off(row, 2 * ch + 1)) = v1[i];
        }
    }
    __syncthreads();
    if (ONLY_KV) {
        kv_tail(ring, wsk, XH, XL, tok0, L, slab, fw, lane, mk);
        return;
    }
    f32x4 xr[2][NTT];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) xr[ft][tt] = *reinterpret_cast<const f32x4*>(HH + stash_off(16 * tt + c16, 8 * fw + 4 * ft + q));
    OPHIP_STAMP(a.stamps, wg, 1);

    // ---- Q projection of head fw, phi, linear attention from registers -> msg planes (Y) --------------------------------
    {
        f32x4 qa[2][NTT];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) qa[ft][tt] = zero4();
        gemm_stage<NTT, 2, 8, true>(qa, ring, wsm, 0 + R, XH, XL, ROWB, 0, c16, q);
        OPHIP_STAMP(a.stamps, wg, 2);
        const float S = a.srclen[s];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            f32x4 p0, p1;
            float den = 0.f;
            bool live = true;                        // q_mask: phi(Q) = 0 for a padded cell => its message is 0
            if (MASKED && mk) { const int tok = tok0 + 16 * tt + c16; live = tok >= L || mk[tok]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0[r] = live ? elu_plus_one_fast(qa[0][tt][r]) : 0.f;
                p1[r] = live ? elu_plus_one_fast(qa[1][tt][r]) : 0.f;
                den += p0[r] * ksm[0][r] + p1[r] * ksm[1][r];
            }
            den = sum_over_q(den);
            bf16x8 qh, ql;
            split8(p0, p1, qh, ql);
            const float z = rcp_fast(den + 1e-6f) * S;
#pragma unroll
            for (int vt = 0; vt < 2; ++vt) {
                f32x4 num = mma16x3(kvh[vt], kvl[vt], qh, ql, zero4());
#pragma unroll
                for (int r = 0; r < 4; ++r) num[r] *= z;
                store_quad(num, YH, YL, ROWB, tt, c16, 32 * fw + 16 * vt + 4 * q);
            }
        }
    }
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 3);

    // ---- merge + LayerNorm 1 -> Y --------------------------------------------------------------------------------------
    {
        f32x4 m[2][NTT];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) m[ft][tt] = zero4();
        gemm_stage<NTT, 2, 8, true>(m, ring, wsm, 32 + R, YH, YL, ROWB, 0, c16, q);
        OPHIP_STAMP(a.stamps, wg, 4);
        float sm[NTT], dm[NTT];
        wave_moments<NTT, 2>(m, sm, dm);
        publish_moments(scratch, sm, dm, fw, c16, q);
        __syncthreads();                             // moments visible; every wave is done reading the msg planes
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            float mean, rstd;
            merged_stats(scratch, tt, c16, mean, rstd);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (m[ft][tt][r] - mean) * rstd * g1[ft][r] + b1[ft][r];
                store_quad(v, YH, YL, ROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
            }
        }
    }
    __syncthreads();
    OPHIP_STAMP(a.stamps, wg, 5);

    // ---- MLP: hidden = relu([x, msg] W0^T) in two 256-feature chunks, o += hidden_chunk W2[:, chunk]^T ------------------------
    f32x4 o[2][NTT];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) o[ft][tt] = zero4();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x4 hd[2][NTT];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) hd[ft][tt] = zero4();
        const int pos = 64 + 96 * c;                 // W0 chunk c: 64 fragments (x half 32, msg half 32), then W2 chunk c: 32
        gemm_stage<NTT, 2, 8, true>(hd, ring, wsm, pos + R, XH, XL, ROWB, 0, c16, q);
        gemm_stage<NTT, 2, 8, true>(hd, ring, wsm, pos + 32 + R, YH, YL, ROWB, 0, c16, q);
        if (c == 1) __syncthreads();                 // every wave is done reading chunk 0 of the hidden planes
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(hd[ft][tt][r], 0.f);
                store_quad(v, HH, HL, ROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
            }
        __syncthreads();
        OPHIP_STAMP(a.stamps, wg, 6 + 2 * c);
        // W2 chunk c; the main stream ends inside chunk 1: its last 16 refills pull the head of the next layer's K|V stream
        if (c == 0) gemm_stage<NTT, 2, 8, true>(o, ring, wsm, pos + 64 + R, HH, HL, ROWB, 0, c16, q);
        else gemm_stage<NTT, 2, 8, true>(o, ring, wsm, pos + 64 + R, HH, HL, ROWB, 0, c16, q, 32 - R, &wsk, 0);
        OPHIP_STAMP(a.stamps, wg, 7 + 2 * c);
    }

    // ---- LayerNorm 2, residual, output rows (and their planes for the fused K|V tail) -----------------------------------
    {
        float so[NTT], dq[NTT];
        wave_moments<NTT, 2>(o, so, dq);
        publish_moments(scratch, so, dq, fw, c16, q);
    }
    __syncthreads();                                 // also: every wave is done with the X planes
    float* yg = a.y[s] + (size_t)b * a.ybs[s];
    const int frag_rows = a.frag_rows[s];
    char* fragp = a.frag[s] ? a.frag[s] + (size_t)b * (frag_rows / 32) * 32768 : nullptr;
    if (fragp && lt == a.tiles[s] - 1) {           // rows between this stream's last workgroup and the 128-row padding: zeros
        const int r0 = a.tiles[s] * TOK;
        for (int i = threadIdx.x; i < (frag_rows - r0) * 64; i += NW * 64) {      // 16-byte chunks: (row, k-step, plane, half)
            const int row = r0 + (i >> 6), c = i & 63, ks = c >> 2, plane = (c >> 1) & 1, half = c & 1;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(fragp + ((size_t)(row >> 5) * 16 + ks) * 2048 + plane * 1024 + 16 * ((row & 31) + 32 * half)) = z;
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        float mean, rstd;
        merged_stats(scratch, tt, c16, mean, rstd);
        const int tok = tok0 + 16 * tt + c16;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = xr[ft][tt][r] + ((o[ft][tt][r] - mean) * rstd * g2[ft][r] + b2[ft][r]);
            if (tok < L) *reinterpret_cast<f32x4*>(yg + (size_t)tok * C + 32 * fw + 16 * ft + 4 * q) = v;
            else v = zero4();
            if (tail) store_quad(v, XH, XL, ROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
            if (fragp && tok < frag_rows) {
                // features 32 fw + 16 ft + 4 q ..+3 of row tok -> k-step 2 fw + ft, half q >> 1, elements 4 (q & 1) ..+3 of lane slot
                // (tok % 32) + 32 (q >> 1) in the 1 KiB fragment of row tile tok / 32; scaled by 1/16 like frag_planes_kernel
                bf16x4 vh, vl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    __bf16 hh, ll;
                    split_bf16(v[r] * 0.0625f, hh, ll);
                    vh[r] = hh; vl[r] = ll;
                }
                char* dst = fragp + ((size_t)(tok >> 5) * 16 + (2 * fw + ft)) * 2048 + 16 * ((tok & 31) + 32 * (q >> 1)) + 8 * (q & 1);
                *reinterpret_cast<bf16x4*>(dst) = vh;
                *reinterpret_cast<bf16x4*>(dst + 1024) = vl;
            }
        }
    }
    OPHIP_STAMP(a.stamps, wg, 10);
    if (tail) {
        __syncthreads();
        kv_tail(ring, wsk, XH, XL, tok0, L, slab, fw, lane, mk);
    }
    OPHIP_STAMP(a.stamps, wg, 11);
    OPHIP_STAMP_REAL(a.stamps, wg, 31);
}

struct KvSumArgs {
    const float* partial;
    char* kv;
    int tiles[2];
    long long kv_bs, kv_ss;    // bytes between the blocks of consecutive batch elements / of the two streams
};

constexpr int KVS_G = 64;          // tile groups per workgroup: a thread sums tiles g, g + 64, g + 128, ... (<= 3 at c2: all its loads in flight at once)
constexpr int KVS_L = 16;          // lanes per group: one workgroup reduces 64 consecutive floats (16 x f32x4) of the 8448 of a slab

// Sum of a stream's K|V partial slabs in a FIXED order (no float atomics: bit-reproducible) -> KV as (hi, lo) bf16 A-operand
// fragments + Ksum f32.  Latency-bound: 264 workgroups (every CU busy) whose threads each issue all their loads at once, instead of
// 66 workgroups looping over the tiles (5.4 -> ~2 us per launch at c2).
__global__ __launch_bounds__(1024) void kv_sum_w8_kernel(KvSumArgs a) {
    __shared__ f32x4 red[KVS_G][KVS_L];
    const int o = threadIdx.x & (KVS_L - 1), g = threadIdx.x / KVS_L;
    const int s = blockIdx.y & 1, b = blockIdx.y >> 1;
    const int ttot = a.tiles[0] + a.tiles[1];
    const int t0 = s ? a.tiles[0] : 0, nt = a.tiles[s];
    if (nt == 0) return;                             // a stream that is not part of this launch (object cache: its block comes from the cache)
    const int e = (blockIdx.x * KVS_L + o) * 4;
    const float* p = a.partial + ((size_t)b * ttot + t0) * KV_PART_FLOATS + e;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int t = g;
    for (; t + 2 * KVS_G < nt; t += 3 * KVS_G) {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(p + (size_t)t * KV_PART_FLOATS);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + KVS_G) * KV_PART_FLOATS);
        const f32x4 u2 = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 2 * KVS_G) * KV_PART_FLOATS);
        acc = ((acc + u0) + u1) + u2;
    }
    for (; t < nt; t += KVS_G) acc += *reinterpret_cast<const f32x4*>(p + (size_t)t * KV_PART_FLOATS);
    red[g][o] = acc;
    __syncthreads();
    if (g < 8) {                                   // 64 -> 8 partial sums, each over groups g, g + 8, ... in that order
        f32x4 tot = red[g][o];
#pragma unroll
        for (int k = 1; k < KVS_G / 8; ++k) tot += red[g + 8 * k][o];
        red[g][o] = tot;
    }
    __syncthreads();
    if (g == 0) {
        f32x4 tot = red[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += red[k][o];
        char* blk = a.kv + (size_t)b * a.kv_bs + (size_t)s * a.kv_ss;
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

extern "C" size_t ophip_encoder_x3w8_workspace_bytes(int B, int L3d, int L2d) {
    const size_t tiles = (size_t)((L3d + TOK - 1) / TOK + (L2d + TOK - 1) / TOK);
    return 2 * (size_t)B * tiles * KV_PART_FLOATS * 4 + (size_t)B * 2 * KV_BLOCK_BYTES + 256;
}

extern "C" size_t ophip_encoder_x3w8_wpack_bytes(void) { return (size_t)NW * (MAIN_FRAGS + KV_FRAGS) * 1024 + 4 * C * 4; }

namespace {
// Object-cache options of a layer call (ophip_frame_enqueue_object / ophip_encoder_object_x3w8; all zero: the plain layer).
//   streams:      bit 0 = the 3D stream's tiles run, bit 1 = the 2D stream's (0 = both).  A stream that is left out has no workgroups, no
//                 K / V slabs and no K^T V sum in this call; the slabs of the streams that DO run are laid out compactly (tile index
//                 inside the launch), so the layer that sums them must be told the same (`partial_streams`).
//   partial_streams: which streams' slabs `partial` of THIS call holds (written by the previous layer's tail or by this call's kv_reduce).
//   x3d_bs:       batch stride of x3d in floats (-1: dense; 0: one object block shared by the batch).
//   kv3d / kv3d_bs: the summed K^T V / Ksum block of the 3D SOURCE for this layer, from the cache (bytes between batch elements; 0 = shared):
//                 the stream that attends to the 3D source reads it instead of the workspace's block.
//   kv_out / kv_out_bs: the 3D stream's summed block goes HERE instead of into the workspace (the cache build).
struct X3Opts {
    int streams = 0;
    int partial_streams = 0;
    long long x3d_bs = -1;
    const void* kv3d = nullptr;
    long long kv3d_bs = 0;
    void* kv_out = nullptr;
    long long kv_out_bs = 0;
};

// kv_mode: 0 = this call projects its own K, V (kv_reduce), sums the slabs (kv_sum) and applies; 1 = the slabs are there (the previous
// layer's fused tail wrote them): kv_sum + apply; 2 = the summed K^T V / Ksum block is there (ophip_encoder_kv_first_x3w8 ran): apply only.
// only_kv: stop after kv_reduce + kv_sum (the body of ophip_encoder_kv_first_x3w8; y3d / y2d / wpack_next unused).
// sum_only: kv_sum of the slabs that are there and nothing else (the cache build's last step).
int layer_x3w8(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
               const void* wpack, const void* wpack_next, int is_cross, int kv_mode, int slot,
               void* workspace, void* stream_, void* frag3d, void* frag2d, const unsigned char* mask2d = nullptr, bool only_kv = false,
               const X3Opts& opt = X3Opts(), bool sum_only = false) {
    const int run = opt.streams ? opt.streams : 3;              // streams whose tiles run in this call
    const int have = opt.partial_streams ? opt.partial_streams : 3;      // streams whose slabs `partial` holds
    const bool r3 = run & 1, r2 = run & 2;
    if ((!sum_only && ((r3 && !x3d) || (r2 && !x2d))) || !wpack || !workspace || (!only_kv && !sum_only && ((r3 && !y3d) || (r2 && !y2d))))
        return ophip_bad_arg(__func__, "null pointer");
    if (is_cross && run != 3 && kv_mode == 0 && !sum_only && (!x3d || !x2d)) return ophip_bad_arg(__func__, "null pointer (a cross layer reads both streams)");
    if (B < 1 || L3d < 1 || L2d < 1) return ophip_bad_arg(__func__, "B, L3d, L2d must be >= 1");
    if (slot != 0 && slot != 1) return ophip_bad_arg(__func__, "slot must be 0 or 1");
    if (kv_mode < 0 || kv_mode > 2) return ophip_bad_arg(__func__, "kv_from_prev must be 0, 1 or 2");
    if (!only_kv && !sum_only && ((r3 && x3d == y3d) || (r2 && x2d == y2d))) return ophip_bad_arg(__func__, "in-place layer is not supported (cross layers read the pre-update streams)");
    if (reinterpret_cast<uintptr_t>(wpack) & 15) return ophip_bad_arg(__func__, "wpack must be 16-byte aligned");
    // a cross layer on ONE stream's rows (LoFTR's sequential cross: image 1 attends to the UPDATED image 0, so each image is its own launch):
    // its K / V come from the OTHER stream, which this call must reduce itself (the previous layer's tail wrote no slabs for that case)
    if (is_cross && run != 3 && !only_kv && !sum_only && kv_mode != 0) return ophip_bad_arg(__func__, "a one-stream cross layer projects its own K / V (kv_from_prev = 0)");
    const int src = (kv_mode == 0 && is_cross && run != 3) ? (run ^ 3) : run;      // streams whose K / V this call reduces
    hipStream_t stream = (hipStream_t)stream_;
    const int t3 = (L3d + TOK - 1) / TOK, t2 = (L2d + TOK - 1) / TOK;
    // the workspace is sized for both streams (ophip_encoder_x3w8_workspace_bytes); a launch that leaves one out uses the head of each slab area
    const size_t part_floats = (size_t)B * (t3 + t2) * KV_PART_FLOATS;
    float* partial = reinterpret_cast<float*>(workspace) + (size_t)slot * part_floats;
    float* partial_next = reinterpret_cast<float*>(workspace) + (size_t)(slot ^ 1) * part_floats;
    char* kv = reinterpret_cast<char*>(workspace) + 2 * part_floats * 4;
    kv += (256 - (reinterpret_cast<uintptr_t>(kv) & 255)) & 255;
    // layer block: [main streams 8 x 256 KiB][K|V streams 8 x 64 KiB][g1 b1 g2 b2 f32]   (packing.pack_coarse_layer_x3w8)
    const bf16x8* wmain = reinterpret_cast<const bf16x8*>(wpack);
    const bf16x8* wkv_own = wmain + (size_t)NW * MAIN_FRAGS * 64;
    const float* ln = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + (size_t)NW * (MAIN_FRAGS + KV_FRAGS) * 1024);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3w8_kernel<false>), LDS_BYTES, "hipFuncSetAttribute(enc_x3w8)")) return rc;
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3w8_kernel<true>), LDS_BYTES, "hipFuncSetAttribute(enc_x3w8 kv)")) return rc;
    if (mask2d) {
        if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3w8_kernel<false, true>), LDS_BYTES, "hipFuncSetAttribute(enc_x3w8 masked)")) return rc;
        if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(enc_x3w8_kernel<true, true>), LDS_BYTES, "hipFuncSetAttribute(enc_x3w8 kv masked)")) return rc;
    }

    // tiles per stream of THIS call's launches (a stream that does not run has none: tile -> stream, token offsets and slab indices follow)
    const int rt3 = r3 ? t3 : 0, rt2 = r2 ? t2 : 0;
    EncW8Args aa;
    aa.x[0] = x3d; aa.x[1] = x2d; aa.y[0] = y3d; aa.y[1] = y2d;
    aa.xbs[0] = opt.x3d_bs >= 0 ? opt.x3d_bs : (long long)L3d * C;
    aa.ybs[0] = (long long)L3d * C; aa.xbs[1] = aa.ybs[1] = (long long)L2d * C;
    aa.L[0] = L3d; aa.L[1] = L2d; aa.tiles[0] = rt3; aa.tiles[1] = rt2;
    aa.kvbs[0] = aa.kvbs[1] = 2LL * KV_BLOCK_BYTES;
    aa.srclen[0] = (float)(is_cross ? L2d : L3d);
    aa.srclen[1] = (float)(is_cross ? L3d : L2d);
    aa.wmain = wmain; aa.ln = ln;
    aa.stamps = ophip_stamp_buffer();
    aa.frag[0] = static_cast<char*>(frag3d); aa.frag[1] = static_cast<char*>(frag2d);
    aa.frag_rows[0] = (L3d + 127) / 128 * 128; aa.frag_rows[1] = (L2d + 127) / 128 * 128;
    aa.mask2d = mask2d;
    if (kv_mode == 0 && !sum_only) {
        EncW8Args ka = aa;
        ka.kv[0] = ka.kv[1] = nullptr;
        ka.frag[0] = ka.frag[1] = nullptr;
        ka.wkv = wkv_own;
        ka.partial = partial;
        ka.stamps = nullptr;
        ka.tiles[0] = (src & 1) ? t3 : 0; ka.tiles[1] = (src & 2) ? t2 : 0;
        const int kt = ka.tiles[0] + ka.tiles[1];
        if (mask2d) { OPHIP_LAUNCH("kv_reduce", stream, (enc_x3w8_kernel<true, true>), dim3(kt, B), dim3(512), LDS_BYTES, stream, ka); }
        else { OPHIP_LAUNCH("kv_reduce", stream, enc_x3w8_kernel<true>, dim3(kt, B), dim3(512), LDS_BYTES, stream, ka); }
        OPHIP_CHECK_LAUNCH();
    }
    if (kv_mode != 2) {
        // the slabs `partial` holds: written by this call's kv_reduce (the source streams) or by the previous layer's tail (`have`)
        const int hs = kv_mode == 0 ? src : have;
        KvSumArgs sa;
        sa.partial = partial; sa.kv = kv; sa.tiles[0] = (hs & 1) ? t3 : 0; sa.tiles[1] = (hs & 2) ? t2 : 0;
        sa.kv_bs = 2LL * KV_BLOCK_BYTES; sa.kv_ss = KV_BLOCK_BYTES;
        if (opt.kv_out) {                                 // the cache build: the 3D stream's block leaves the workspace
            if (hs != 1) return ophip_bad_arg(__func__, "kv_out needs a 3D-only call");
            sa.kv = static_cast<char*>(opt.kv_out); sa.kv_bs = opt.kv_out_bs; sa.kv_ss = 0;
        }
        static_assert(KV_PART_FLOATS % (4 * KVS_L) == 0, "a slab is a whole number of 64-float chunks");
        OPHIP_LAUNCH("kv_sum", stream, kv_sum_w8_kernel, dim3(KV_PART_FLOATS / (4 * KVS_L), 2 * B), dim3(KVS_G * KVS_L), 0, stream, sa);
        OPHIP_CHECK_LAUNCH();
    }
    if (only_kv || sum_only) return 0;
    aa.kv[0] = kv + (is_cross ? KV_BLOCK_BYTES : 0);
    aa.kv[1] = kv + (is_cross ? 0 : KV_BLOCK_BYTES);
    if (opt.kv3d) {                                       // the 3D source's block from the cache: read by the stream that attends to it
        const int rd = is_cross ? 1 : 0;
        aa.kv[rd] = static_cast<const char*>(opt.kv3d);
        aa.kvbs[rd] = opt.kv3d_bs;
    }
    aa.wkv = nullptr;
    aa.partial = partial_next;
    if (wpack_next) aa.wkv = reinterpret_cast<const bf16x8*>(wpack_next) + (size_t)NW * MAIN_FRAGS * 64;
    if (mask2d) { OPHIP_LAUNCH("attn_apply", stream, (enc_x3w8_kernel<false, true>), dim3(rt3 + rt2, B), dim3(512), LDS_BYTES, stream, aa); }
    else { OPHIP_LAUNCH("attn_apply", stream, enc_x3w8_kernel<false>, dim3(rt3 + rt2, B), dim3(512), LDS_BYTES, stream, aa); }
    OPHIP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int ophip_encoder_layer_x3w8(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                        const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                        void* workspace, void* stream) {
    return layer_x3w8(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace, stream, nullptr, nullptr);
}

// The same layer; its output rows are ALSO written as the (hi, lo) bf16 operand fragments of the similarity kernel (scaled by
// 1/16, rows padded with zeros to a multiple of 128: the layout of csrc/coarse_match.hip's frag_planes kernel), so that the last
// encoder layer feeds ophip_coarse_match_conf directly (nsplit | OPHIP_COARSE_PLANES_READY) and that kernel's launch disappears.
extern "C" int ophip_encoder_layer_x3w8_frag(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                             const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                             void* workspace, void* frag3d, void* frag2d, void* stream) {
    if (!frag3d || !frag2d) return ophip_bad_arg(__func__, "null fragment buffer");
    return layer_x3w8(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace, stream, frag3d, frag2d);
}

// The same layer with the reference's query_mask (transformer.py:148-159): mask2d [B][L2d], 1 = real cell, 0 = padding.  The 2D stream's
// padded rows drop out as sources (phi(K) = 0) and get a zero message as queries (phi(Q) = 0); their rows are still updated.
extern "C" int ophip_encoder_layer_x3w8_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                               const void* wpack, const void* wpack_next, int is_cross, int kv_from_prev, int slot,
                                               void* workspace, const unsigned char* mask2d, void* stream) {
    if (!mask2d) return ophip_bad_arg(__func__, "null mask (use ophip_encoder_layer_x3w8)");
    return layer_x3w8(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace, stream, nullptr, nullptr, mask2d);
}

// The K / V half of a layer that projects its own K, V (the first layer of a frame: nothing wrote its slabs): K, V projections of both
// streams -> phi(K)^T V / Ksum slabs (kv_reduce) -> their fixed-order sum (kv_sum) into the workspace's K^T V block.  It reads the layer's
// INPUT rows only, so a pipeline may issue it as soon as those exist -- before, and beside, whatever the previous frame still runs --
// and then call ophip_encoder_layer_x3w8{,_frag,_masked} with kv_from_prev = 2 (same wpack, slot, workspace; mask2d NULL or the layer's mask).
// Bit-identical to the one-call layer (same kernels, same order).  Reference: transformer.py:65-94 (k_proj, v_proj), linear_attention.py:49-57.
extern "C" int ophip_encoder_kv_first_x3w8(const float* x3d, const float* x2d, int B, int L3d, int L2d, const void* wpack, int slot,
                                           void* workspace, const unsigned char* mask2d, void* stream) {
    return layer_x3w8(x3d, x2d, nullptr, nullptr, B, L3d, L2d, wpack, nullptr, 0, 0, slot, workspace, stream, nullptr, nullptr, mask2d, true);
}

// ---- object cache ("kpt_encode and the first 3D self-layer are frame-invariant ... computed once per
//      sequence and cached"; the reference keeps the object block resident, OnePosePlus_inference_dataset.py:157-169) -------------------
// With a first layer of kind "self" the 3D stream's first-layer output depends on the object block and the weights only
// (transformer.py:148-153), and so does the K^T V / Ksum block of those rows as the SOURCE of the second layer (:154-159).
// ophip_encoder_object_x3w8 computes both with the very launches a frame would run -- kv_reduce / kv_sum / attn_apply restricted to the 3D
// stream's workgroups; a workgroup's arithmetic does not depend on which other workgroups share its launch, and kv_sum adds a stream's
// slabs in the same fixed order -- so a frame that uses the cache is bit-identical to one that does not.
//   x3d [Bo][N][256]: the keypoint encoding (ophip_kpt_encode); y3d0 [Bo][N][256]: layer 0's rows; kv1 [Bo][ophip_encoder_x3w8_kv_block_bytes()]:
//   the summed block of layer 1's 3D source; workspace: ophip_encoder_x3w8_workspace_bytes(Bo, N, 1) bytes (scratch).
extern "C" size_t ophip_encoder_x3w8_kv_block_bytes(void) { return KV_BLOCK_BYTES; }

//   masked_frames: 1 = the entry is for frames with a query mask (ophip_frame_enqueue_object with query_mask).  The mask touches the 2D
//   stream only, but a masked frame runs BOTH streams through the masked instantiation of the kernels, which need not round like the plain
//   one (different code, different contraction choices): the entry is then built with that instantiation too (the mask pointer it is given
//   is never dereferenced for 3D workgroups), and stays bit-identical to the uncached masked frame.  Keep one entry per kind.
extern "C" int ophip_encoder_object_x3w8(const float* x3d, int Bo, int N, const void* wpack0, const void* wpack1, void* workspace,
                                         float* y3d0, void* kv1, int masked_frames, void* stream) {
    if (!x3d || !wpack0 || !wpack1 || !workspace || !y3d0 || !kv1) return ophip_bad_arg(__func__, "null pointer");
    if (reinterpret_cast<uintptr_t>(kv1) & 15) return ophip_bad_arg(__func__, "kv1 must be 16-byte aligned");
    const unsigned char* mk = masked_frames ? reinterpret_cast<const unsigned char*>(x3d) : nullptr;      // selects the instantiation; never read (3D tiles only)
    X3Opts o;
    o.streams = 1;
    // layer 0 on the 3D stream alone: its own K / V (kv_reduce + kv_sum), then the layer; the tail leaves layer 1's slabs of these rows
    if (int rc = layer_x3w8(x3d, nullptr, y3d0, nullptr, Bo, N, 1, wpack0, wpack1, 0, 0, 0, workspace, stream, nullptr, nullptr, mk, false, o)) return rc;
    // their fixed-order sum = the 3D source's block of layer 1 (slot 1: where layer 0's tail wrote)
    X3Opts s;
    s.partial_streams = 1;
    s.kv_out = kv1; s.kv_out_bs = KV_BLOCK_BYTES;
    return layer_x3w8(x3d, nullptr, nullptr, nullptr, Bo, N, 1, wpack1, nullptr, 0, 1, 1, workspace, stream, nullptr, nullptr, nullptr, false, s, true);
}

// The layer calls of a frame that uses the cache (csrc/frame.hip).  first: layer 0 on the 2D stream alone (kv_mode as in the plain call: 0
// = with its own K / V half, 2 = ophip_encoder_kv_first_object_x3w8 ran); second: layer 1 on both streams, its 3D rows and the 3D source's
// block from the cache.
int ophip_x3w8_object_first(const float* x2d, float* y2d, int B, int L3d, int L2d, const void* wpack, const void* wpack_next, int kv_mode,
                            void* workspace, const unsigned char* mask2d, bool only_kv, void* stream) {
    X3Opts o;
    o.streams = 2;
    return layer_x3w8(nullptr, x2d, nullptr, y2d, B, L3d, L2d, wpack, wpack_next, 0, kv_mode, 0, workspace, stream, nullptr, nullptr, mask2d, only_kv, o);
}
int ophip_x3w8_object_second(const float* y3d0, long long y3d0_bs, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                             const void* wpack, const void* wpack_next, int is_cross, const void* kv1, long long kv1_bs,
                             void* workspace, void* frag3d, void* frag2d, const unsigned char* mask2d, void* stream) {
    X3Opts o;
    o.partial_streams = 2;
    o.x3d_bs = y3d0_bs;
    o.kv3d = kv1; o.kv3d_bs = kv1_bs;
    return layer_x3w8(y3d0, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, is_cross, 1, 1, workspace, stream, frag3d, frag2d, mask2d, false, o);
}
int ophip_x3w8_layer_bs(const float* x3d, long long x3d_bs, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                        const void* wpack, const void* wpack_next, int is_cross, int kv_mode, int slot,
                        void* workspace, void* frag3d, void* frag2d, const unsigned char* mask2d, bool only_kv, void* stream) {
    X3Opts o;
    o.x3d_bs = x3d_bs;
    return layer_x3w8(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, wpack_next, is_cross, kv_mode, slot, workspace, stream, frag3d, frag2d, mask2d, only_kv, o);
}

// One layer on a SUBSET of the two streams' rows (streams: bit 0 = the first stream, bit 1 = the second; 3 = ophip_encoder_layer_x3w8 with
// kv_from_prev = 0 and no fused tail).  What it is for: LoFTR's cross layers are sequential -- feat0 = layer(feat0, feat1), then
// feat1 = layer(feat1, feat0_NEW) (zju3dv/LoFTR loftr/loftr_module/transformer.py) -- so each image's update is a launch of its own;
// issuing both with the two-stream entry point computed every cross layer twice and threw half of each result away (rounds 3-4,
// onepose_st_amd/loftr.py).  The rows that run are bit-identical to the same rows of the two-stream call.
extern "C" int ophip_encoder_layer_x3w8_streams(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                                const void* wpack, int is_cross, int streams, void* workspace, void* stream) {
    if (streams < 1 || streams > 3) return ophip_bad_arg(__func__, "streams must be 1, 2 or 3");
    X3Opts o;
    o.streams = streams;
    return layer_x3w8(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, nullptr, is_cross, 0, 0, workspace, stream, nullptr, nullptr, nullptr, false, o);
}
