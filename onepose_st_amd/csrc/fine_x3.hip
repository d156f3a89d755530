// Fine-level refinement, split-bf16, second-generation mapping (same mathematics as csrc/fine.hip / fine_bf16.hip; reference:
// loftr_module/fine_preprocess.py:32-55, loftr_module/transformer.py:65-171 with the loftr_fine config,
// utils/fine_matching.py:28-110).
//
// What bounded the first-generation kernel (one match per workgroup, two workgroups per CU): every workgroup pulled the two
// layers' 1.28 MB of weight fragments for 26 live tokens, and two workgroups on a CU get 60 GB/s each.  Here a workgroup
// (4 waves, one per CU) refines MPW = 3 matches = 6 token tiles of 16 rows (match m: rows 32m .. 32m + 24 the 5 x 5 window,
// row 32m + 25 the 3D token, the rest padding); wave fw owns features 32 fw .. 32 fw + 31 (= heads 2fw, 2fw + 1) of all six
// tiles, so a 1 KiB weight fragment from the wave's stream feeds 6 x 3 MFMAs straight from registers and the stream is paid
// once per three matches (tile_x3.h: v_mfma_f32_16x16x32_bf16, register ring behind a buffer descriptor, rolled GEMM stages).
//   * K, V come out as D[token][feature]; per match, head and source set (window | 3D token) KV = phi(K)^T V is ONE MFMA
//     group over the match's 32 token rows, accumulator to operand, the set selected by masking phi(K) on its token rows;
//   * Q comes out as D[feature][token]; two 16-wide heads share the 32-deep contraction of phi(Q) KV block-diagonally, and
//     the KV tile is already the A operand the product needs (its rows are the contraction index);
//   * a token attends to exactly one set: phi(Q) is masked per set on its token (lane) axis, Ksum goes through a 3 KiB LDS
//     table private to the wave, the denominators are exact f32 dot products on the vector ALU;
//   * the residual stream stays in f32 registers across both layers; LayerNorms take one barrier (Chan merge of wave moments).
#include "tile_x3.h"
#include <stdlib.h>

namespace {

using namespace x3;

constexpr int CF = 128, WIN = 25, TOK3D = 25;
constexpr int ROWB = CF * 2;                    // plane pitch (256 B, 16 chunks); the hidden chunk is 128 wide too
constexpr int LAYER_FRAGS = 160;                // per wave and layer: K|V 32 | Q 16 | merge 16 | W0c0 32 | W2c0 16 | W0c1 32 | W2c1 16

struct FineX3Args {
    const float* feat_f; long long fs_b, fs_y, fs_x; int hf, wf;       // channels-last fine map
    const float* desc_f; long long ds_b, ds_c;
    const long long *b_ids, *i_ids, *j_ids;
    const int* count;
    const float* mkq_c;
    const bf16x8* wstream;       // [4 waves][nlayers * LAYER_FRAGS][64]
    const float* ln;             // [nlayers][4][CF]
    int nlayers; unsigned cross_bits;
    int wc, stride;
    float fine_scale;
    float* expec_f; float* mkq_f;
    float* dbg_win; float* dbg_f3;
};

// f32 image [rows][128], 16-byte chunks swizzled by row (32 chunks per row)
__device__ __forceinline__ int stash_off(int row, int chunk) { return row * (CF * 4) + ((chunk ^ (row & 15)) << 4); }

template <int MPW>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 1) void fine_x3_kernel(FineX3Args p) {
    constexpr int NTT = 2 * MPW, TOK = 16 * NTT, PLANE = TOK * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;
    char* XL = smem + PLANE;
    char* YH = smem + 2 * PLANE;
    char* YL = smem + 3 * PLANE;
    char* HH = smem + 4 * PLANE;
    char* HL = smem + 5 * PLANE;
    char* stash = HH;                                               // f32 [TOK][128] = 2 PLANE: start and end only
    float* scratch = reinterpret_cast<float*>(smem + 6 * PLANE);   // LayerNorm moments [4 waves][TOK][2]
    float* kstab = scratch + 4 * TOK * 2;                           // Ksum [MPW][2 sets][8 heads][16]
    const int k0 = MPW * blockIdx.x;
    const int total = *p.count;
    if (k0 >= total) return;
    const int tid = threadIdx.x, lane = tid & 63, fw = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int fwu = __builtin_amdgcn_readfirstlane(fw);
    const int nl = p.nlayers;
    WStream ws;
    ws.open(p.wstream + (size_t)fwu * nl * LAYER_FRAGS * 64, nl * LAYER_FRAGS, lane);
    Ring ring;
#pragma unroll
    for (int i = 0; i < R; ++i) ring.s[i] = ws.load(i);             // (no layers: zero records, nothing is fetched)

    // ---- gather the matches' windows and 3D descriptors into the f32 stash -------------------------------------------
    {
        constexpr int NQ = MPW * WIN * (CF / 4), PER = (NQ + 255) / 256;
        f32x4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            v[u] = zero4();
            if (e < NQ) {
                const int mi = e / (WIN * (CF / 4)), e2 = e % (WIN * (CF / 4)), rr = e2 >> 5, c4 = e2 & 31;
                if (k0 + mi < total) {
                    const int b = (int)p.b_ids[k0 + mi], j = (int)p.j_ids[k0 + mi];
                    const int y = p.stride * (j / p.wc) + rr / 5 - 2, x = p.stride * (j % p.wc) + rr % 5 - 2;
                    if (y >= 0 && y < p.hf && x >= 0 && x < p.wf)
                        v[u] = *reinterpret_cast<const f32x4*>(p.feat_f + (size_t)b * p.fs_b + (size_t)y * p.fs_y + (size_t)x * p.fs_x + 4 * c4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            if (e < NQ) {
                const int mi = e / (WIN * (CF / 4)), e2 = e % (WIN * (CF / 4)), rr = e2 >> 5, c4 = e2 & 31;
                *reinterpret_cast<f32x4*>(stash + stash_off(32 * mi + rr, c4)) = v[u];
            }
        }
        for (int e = tid; e < MPW * CF; e += 256) {                 // the 3D fine descriptor (row 25) of every match
            const int mi = e >> 7, c = e & 127;
            float d = 0.f;
            if (k0 + mi < total) d = p.desc_f[(size_t)p.b_ids[k0 + mi] * p.ds_b + (size_t)c * p.ds_c + p.i_ids[k0 + mi]];
            *reinterpret_cast<float*>(stash + stash_off(32 * mi + TOK3D, c >> 2) + 4 * (c & 3)) = d;
        }
        for (int e = tid; e < MPW * 6 * (CF / 4); e += 256) {       // padding rows 26..31
            const int mi = e / (6 * (CF / 4)), e2 = e % (6 * (CF / 4)), rr = 26 + (e2 >> 5), c4 = e2 & 31;
            *reinterpret_cast<f32x4*>(stash + stash_off(32 * mi + rr, c4)) = zero4();
        }
    }
    __syncthreads();
    // residual stream in registers, D[feature][token]: this wave's 32 features (tiles ft = 0, 1) of all tokens
    f32x4 xres[2][NTT];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            xres[ft][tt] = *reinterpret_cast<const f32x4*>(stash + stash_off(16 * tt + c16, 8 * fw + 4 * ft + q));
            store_quad(xres[ft][tt], XH, XL, ROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
        }
    __syncthreads();

    // token classes on the two axes an accumulator can carry them
    //   rows (D[token][feature], K / V):  tile 2m: rows 4q + r = window tokens 0..15;  tile 2m + 1: 16 + 4q + r
    //   cols (D[feature][token], Q):      tile 2m: all window;  tile 2m + 1: c16 <= 8 window, c16 == 9 the 3D token, else padding
    const bool col_is3d = c16 == 9, col_pad = c16 > 9;

    for (int l = 0; l < nl; ++l) {
        const int base = l * LAYER_FRAGS;
        const float* ln = p.ln + (size_t)l * 4 * CF;
        const bool cross = (p.cross_bits >> l) & 1u;

        // ---- K, V of heads 2fw, 2fw+1 -> per match and set: KV tiles (registers) and Ksum (LDS table of this wave) -------
        f32x4 kvw[MPW][2], kv3[MPW][2];                  // [match][head]: KV[d = 4q + r][v = c16]
        {
            f32x4 kk[4][NTT];                            // D[token][feature]: ft 0, 1 = K of the two heads, 2, 3 = V
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) kk[ft][tt] = zero4();
            gemm_stage<NTT, 4, 4, false>(kk, ring, ws, base + R, XH, XL, ROWB, 0, c16, q);
            const f32x4 z4 = zero4();
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                f32x4 kw[2][2], k3[2], vs[2][2];         // [head][tile]; the 3D token only lives in the second tile
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int tok = 16 * t + 4 * q + r;
                            const float pk = elu_plus_one_fast(kk[hh][2 * mi + t][r]);
                            kw[hh][t][r] = tok < WIN ? pk : 0.f;
                            if (t == 1) k3[hh][r] = tok == TOK3D ? pk : 0.f;
                            const float vv = kk[2 + hh][2 * mi + t][r];
                            vs[hh][t][r] = tok < WIN ? vv * 0.04f : (tok == TOK3D ? vv : 0.f);      // values / v_length (25 | 1)
                        }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    bf16x8 awh, awl, a3h, a3l, bh, bl;
                    split8(kw[hh][0], kw[hh][1], awh, awl);
                    split8(z4, k3[hh], a3h, a3l);
                    split8(vs[hh][0], vs[hh][1], bh, bl);
                    kvw[mi][hh] = mma16x3(awh, awl, bh, bl, zero4());
                    kv3[mi][hh] = mma16x3(a3h, a3l, bh, bl, zero4());
                    float sw = (kw[hh][0][0] + kw[hh][0][1]) + (kw[hh][0][2] + kw[hh][0][3]) + (kw[hh][1][0] + kw[hh][1][1]) + (kw[hh][1][2] + kw[hh][1][3]);
                    float s3 = (k3[hh][0] + k3[hh][1]) + (k3[hh][2] + k3[hh][3]);
                    sw = sum_over_q(sw);
                    s3 = sum_over_q(s3);
                    if (q == 0) {
                        kstab[((mi * 2 + 0) * 8 + 2 * fw + hh) * 16 + c16] = sw;
                        kstab[((mi * 2 + 1) * 8 + 2 * fw + hh) * 16 + c16] = s3;
                    }
                }
            }
        }
        // ---- Q, phi, attention from registers -> msg planes (Y) ----------------------------------------------------------
        {
            f32x4 qa[2][NTT];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) qa[ft][tt] = zero4();
            gemm_stage<NTT, 2, 4, true>(qa, ring, ws, base + 32 + R, XH, XL, ROWB, 0, c16, q);
            const f32x4 z4 = zero4();
            const bf16x8 zf = zero_bf8();
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                // A operands: KV^T of one head in its half of the 32-deep contraction, zeros in the other head's half
                bf16x8 aw[2][2], a3[2][2];               // [head][plane]
                split8(kvw[mi][0], z4, aw[0][0], aw[0][1]);
                split8(z4, kvw[mi][1], aw[1][0], aw[1][1]);
                split8(kv3[mi][0], z4, a3[0][0], a3[0][1]);
                split8(z4, kv3[mi][1], a3[1][0], a3[1][1]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tt = 2 * mi + t;
                    f32x4 p0, p1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p0[r] = elu_plus_one_fast(qa[0][tt][r]);
                        p1[r] = elu_plus_one_fast(qa[1][tt][r]);
                    }
                    // the set this lane's token attends to: self: window tokens -> window, 3D token -> itself; cross: swapped
                    const bool tok3d = t == 1 && col_is3d;
                    const bool use3 = (t == 1 && col_pad) ? false : (cross ? !tok3d : tok3d);
                    const float* kp = kstab + ((mi * 2 + (use3 ? 1 : 0)) * 8 + 2 * fw) * 16 + 4 * q;
                    const f32x4 ks0 = *reinterpret_cast<const f32x4*>(kp), ks1 = *reinterpret_cast<const f32x4*>(kp + 16);
                    float d0 = (p0[0] * ks0[0] + p0[1] * ks0[1]) + (p0[2] * ks0[2] + p0[3] * ks0[3]);
                    float d1 = (p1[0] * ks1[0] + p1[1] * ks1[1]) + (p1[2] * ks1[2] + p1[3] * ks1[3]);
                    d0 = sum_over_q(d0);
                    d1 = sum_over_q(d1);
                    bf16x8 bh, bl;
                    split8(p0, p1, bh, bl);              // B[k][token]: k slot (q, j) = d 4q + j of head 0 (j < 4) / head 1 (j >= 4)
                    f32x4 n0, n1;
                    if (t == 0) {                        // all 16 tokens are window tokens: one set for the whole tile (wave-uniform)
                        if (cross) {
                            n0 = mma16x3(a3[0][0], a3[0][1], bh, bl, zero4());
                            n1 = mma16x3(a3[1][0], a3[1][1], bh, bl, zero4());
                        } else {
                            n0 = mma16x3(aw[0][0], aw[0][1], bh, bl, zero4());
                            n1 = mma16x3(aw[1][0], aw[1][1], bh, bl, zero4());
                        }
                    } else {                             // mixed tile: phi(Q) masked per set on the token (lane) axis, both sets accumulate
                        const bf16x8 bwh = use3 ? zf : bh, bwl = use3 ? zf : bl, b3h = use3 ? bh : zf, b3l = use3 ? bl : zf;
                        n0 = mma16x3(aw[0][0], aw[0][1], bwh, bwl, zero4());
                        n0 = mma16x3(a3[0][0], a3[0][1], b3h, b3l, n0);
                        n1 = mma16x3(aw[1][0], aw[1][1], bwh, bwl, zero4());
                        n1 = mma16x3(a3[1][0], a3[1][1], b3h, b3l, n1);
                    }
                    const float S = use3 ? 1.0f : 25.0f;
                    const float z0 = rcp_fast(d0 + 1e-6f) * S, z1 = rcp_fast(d1 + 1e-6f) * S;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { n0[r] *= z0; n1[r] *= z1; }
                    store_quad(n0, YH, YL, ROWB, tt, c16, 32 * fw + 4 * q);
                    store_quad(n1, YH, YL, ROWB, tt, c16, 32 * fw + 16 + 4 * q);
                }
            }
        }
        __syncthreads();
        // ---- merge + LayerNorm 1 -> Y -------------------------------------------------------------------------------------
        {
            f32x4 m[2][NTT];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) m[ft][tt] = zero4();
            gemm_stage<NTT, 2, 4, true>(m, ring, ws, base + 48 + R, YH, YL, ROWB, 0, c16, q);
            float sm[NTT], dm[NTT];
            wave_moments<NTT, 2>(m, sm, dm);
            if (q == 0) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    float2 v = {sm[tt], dm[tt]};
                    *reinterpret_cast<float2*>(scratch + (fw * TOK + 16 * tt + c16) * 2) = v;
                }
            }
            __syncthreads();                             // moments visible; every wave is done reading the msg planes
            f32x4 g1[2], b1[2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                g1[ft] = *reinterpret_cast<const f32x4*>(ln + 32 * fw + 16 * ft + 4 * q);
                b1[ft] = *reinterpret_cast<const f32x4*>(ln + CF + 32 * fw + 16 * ft + 4 * q);
            }
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float sw[4], dw[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 v = *reinterpret_cast<const float2*>(scratch + (w * TOK + 16 * tt + c16) * 2);
                    sw[w] = v.x; dw[w] = v.y;
                }
                const float mean = ((sw[0] + sw[1]) + (sw[2] + sw[3])) * (1.0f / CF);
                float m2 = (dw[0] + dw[1]) + (dw[2] + dw[3]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float d = sw[w] * (1.0f / 32) - mean;
                    m2 += 32.f * d * d;
                }
                const float rstd = 1.0f / sqrtf(m2 * (1.0f / CF) + 1e-5f);
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
        // ---- MLP: hidden = relu([x, msg] W0^T) in two 128-feature chunks, o += hidden_chunk W2[:, chunk]^T ---------------------
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
            const int pos = base + 64 + 48 * c;          // W0 chunk c: 32 fragments (x half 16, msg half 16), then W2 chunk c: 16
            gemm_stage<NTT, 2, 4, true>(hd, ring, ws, pos + R, XH, XL, ROWB, 0, c16, q);
            gemm_stage<NTT, 2, 4, true>(hd, ring, ws, pos + 16 + R, YH, YL, ROWB, 0, c16, q);
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
            gemm_stage<NTT, 2, 4, true>(o, ring, ws, pos + 32 + R, HH, HL, ROWB, 0, c16, q);
        }
        // ---- LayerNorm 2, residual, new X planes -------------------------------------------------------------------------------
        {
            float so[NTT], dq[NTT];
            wave_moments<NTT, 2>(o, so, dq);
            if (q == 0) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    float2 v = {so[tt], dq[tt]};
                    *reinterpret_cast<float2*>(scratch + (fw * TOK + 16 * tt + c16) * 2) = v;
                }
            }
            __syncthreads();                             // also: every wave is done with the X planes and the hidden planes
            f32x4 g2[2], b2[2];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                g2[ft] = *reinterpret_cast<const f32x4*>(ln + 2 * CF + 32 * fw + 16 * ft + 4 * q);
                b2[ft] = *reinterpret_cast<const f32x4*>(ln + 3 * CF + 32 * fw + 16 * ft + 4 * q);
            }
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float sw[4], dw[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 v = *reinterpret_cast<const float2*>(scratch + (w * TOK + 16 * tt + c16) * 2);
                    sw[w] = v.x; dw[w] = v.y;
                }
                const float mean = ((sw[0] + sw[1]) + (sw[2] + sw[3])) * (1.0f / CF);
                float m2 = (dw[0] + dw[1]) + (dw[2] + dw[3]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float d = sw[w] * (1.0f / 32) - mean;
                    m2 += 32.f * d * d;
                }
                const float rstd = 1.0f / sqrtf(m2 * (1.0f / CF) + 1e-5f);
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xres[ft][tt][r] += (o[ft][tt][r] - mean) * rstd * g2[ft][r] + b2[ft][r];
                    store_quad(xres[ft][tt], XH, XL, ROWB, tt, c16, 32 * fw + 16 * ft + 4 * q);
                }
            }
        }
        __syncthreads();
    }

    // ---- final f32 features to the stash, then correlation -> softmax -> expectation (one wave per match) --------------------
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) *reinterpret_cast<f32x4*>(stash + stash_off(16 * tt + c16, 8 * fw + 4 * ft + q)) = xres[ft][tt];
    __syncthreads();
    if (p.dbg_win) {
        for (int mi = 0; mi < MPW; ++mi) {
            const int k = k0 + mi;
            if (k >= total) break;
            for (int e = tid; e < WIN * CF; e += 256) {
                const int rr = e >> 7, c = e & 127;
                p.dbg_win[(size_t)k * WIN * CF + e] = *reinterpret_cast<const float*>(stash + stash_off(32 * mi + rr, c >> 2) + 4 * (c & 3));
            }
            if (tid < CF) p.dbg_f3[(size_t)k * CF + tid] = *reinterpret_cast<const float*>(stash + stash_off(32 * mi + TOK3D, tid >> 2) + 4 * (tid & 3));
        }
    }
    if (fw < MPW && k0 + fw < total) {
        const int k = k0 + fw, base = 32 * fw;
        float t = -INFINITY;
        if (lane < WIN) {
            float dot = 0.f;
            for (int c4 = 0; c4 < CF / 4; ++c4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(stash + stash_off(base + TOK3D, c4));
                const f32x4 bq = *reinterpret_cast<const f32x4*>(stash + stash_off(base + lane, c4));
                dot += a[0] * bq[0];
                dot += a[1] * bq[1];
                dot += a[2] * bq[2];
                dot += a[3] * bq[3];
            }
            t = dot * 0.08838834764831845f;              // 1 / sqrt(128)
        }
        const float m = wave_max(t);
        const float e = lane < WIN ? expf(t - m) : 0.f;
        const float sum = wave_sum(e);
        const float pr = e / sum;
        const float gx = (float)(lane % 5 - 2) * 0.5f, gy = (float)(lane / 5 - 2) * 0.5f;
        const float ex = wave_sum(pr * gx), ey = wave_sum(pr * gy);
        const float ex2 = wave_sum(pr * gx * gx), ey2 = wave_sum(pr * gy * gy);
        if (lane == 0) {
            const float vx = ex2 - ex * ex, vy = ey2 - ey * ey;
            const float sd = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
            p.expec_f[3 * k] = ex; p.expec_f[3 * k + 1] = ey; p.expec_f[3 * k + 2] = sd;
            p.mkq_f[2 * k] = p.mkq_c[2 * k] + ex * p.fine_scale;
            p.mkq_f[2 * k + 1] = p.mkq_c[2 * k + 1] + ey * p.fine_scale;
        }
    }
}

}  // namespace

extern "C" size_t ophip_fine_x3_wpack_bytes(int nlayers) { return (size_t)nlayers * ((size_t)4 * LAYER_FRAGS * 1024 + 4 * CF * 4); }

extern "C" int ophip_fine_refine_x3(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                                    const float* desc3d_f, long long ds_b, long long ds_c,
                                    const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                                    const float* mkpts_c, const void* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
                                    int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                                    float* dbg_win, float* dbg_f3, void* stream_) {
    if (!feat_f || !desc3d_f || !b_ids || !i_ids || !j_ids || !count || !mkpts_c || !expec_f || !mkpts_f)
        return ophip_bad_arg(__func__, "null pointer");
    if (encoder_enable && (!wpack || nlayers < 1 || nlayers > 32)) return ophip_bad_arg(__func__, "encoder enabled without weights");
    if (fs_c != 1) return ophip_bad_arg(__func__, "the fine map must be channels-last (fs_c == 1); ophip_fine_refine_bf16 takes NCHW");
    if ((reinterpret_cast<uintptr_t>(feat_f) & 15) || (fs_b & 3) || (fs_y & 3) || (fs_x & 3))
        return ophip_bad_arg(__func__, "channels-last feat_f needs 16-byte aligned pixels (strides multiples of 4 floats)");
    if (wpack && (reinterpret_cast<uintptr_t>(wpack) & 15)) return ophip_bad_arg(__func__, "wpack must be 16-byte aligned");
    if ((dbg_win == nullptr) != (dbg_f3 == nullptr)) return ophip_bad_arg(__func__, "dbg_win and dbg_f3 go together");
    if (max_matches <= 0) return 0;
    const int nl = encoder_enable ? nlayers : 0;
    FineX3Args a;
    a.feat_f = feat_f; a.fs_b = fs_b; a.fs_y = fs_y; a.fs_x = fs_x; a.hf = hf; a.wf = wf;
    a.desc_f = desc3d_f; a.ds_b = ds_b; a.ds_c = ds_c;
    a.b_ids = b_ids; a.i_ids = i_ids; a.j_ids = j_ids; a.count = count; a.mkq_c = mkpts_c;
    // block: [streams: 4 waves x nlayers x LAYER_FRAGS KiB][ln: nlayers x (g1 b1 g2 b2) f32]   (packing.pack_fine_layers_x3)
    a.wstream = reinterpret_cast<const bf16x8*>(wpack);
    a.ln = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + (size_t)4 * nlayers * LAYER_FRAGS * 1024);
    a.nlayers = nl; a.cross_bits = cross_bits;
    a.wc = wc; a.stride = stride; a.fine_scale = fine_scale;
    a.expec_f = expec_f; a.mkq_f = mkpts_f; a.dbg_win = dbg_win; a.dbg_f3 = dbg_f3;
    if (nl > 0 && nl != nlayers) return ophip_bad_arg(__func__, "nlayers mismatch");
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int MPW = 3;
    const int grid = (max_matches + MPW - 1) / MPW;
    const size_t lds = (size_t)6 * (2 * MPW * 16) * ROWB + (size_t)4 * (2 * MPW * 16) * 2 * 4 + (size_t)MPW * 2 * 8 * 16 * 4;
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(fine_x3_kernel<MPW>), lds, "hipFuncSetAttribute(fine_x3)")) return rc;
    OPHIP_LAUNCH("fine_refine", stream, fine_x3_kernel<MPW>, dim3(grid), dim3(256), lds, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
