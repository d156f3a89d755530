// Fine stage of the LoFTR 2D-2D matcher behind the object detector: window 9 x 9 on BOTH images,
// two-stream fine transformer (d_model 128, 8 heads, [self, cross]), correlation of image 0's centre token with image 1's window,
// soft-argmax.  Reference call site: src/KeypointFreeSfM/loftr_for_sfm/loftr.py:127-135 (fine_preprocess, loftr_fine,
// fine_matching); the arithmetic lives in the un-vendored submodules/LoFTR (zju3dv/LoFTR, no pin in .gitmodules) and is restated
// from its published definition: "parity unpinned" (oracle/loftr_oracle.py).
//
// A match carries 2 x 81 tokens here (the OnePose++ fine stage: 26), too many for the one-workgroup-per-match kernel of
// fine_bf16.hip, so the stage is batched over ALL matches instead: token rows [K * 81][128] per image in HBM, every linear layer
// one split-bf16 MFMA GEMM over all rows (weights cross L2 -> CU once per 64 rows), the per-match pieces (window gather, linear
// attention, LayerNorm, correlation) small row-parallel kernels.  The detector runs on frame 0 and after a lost track only
// (inference.py:142-173): correctness and the C-ABI boundary matter here, not the last microsecond.
#include "tile_bf16.h"

namespace {

constexpr int CF = 128, NH = 8, DH = CF / NH;

// ---- window gather: 9 x 9 (W x W) windows of a channels-last fine map around coarse cells, zero padding ----------------------
struct GatherArgs {
    const float* feat;          // [hf * wf][CF] channels-last (one image; batched form: [B][hf * wf][CF])
    const long long* ids;       // [K] coarse cell index of every match (i_ids or j_ids)
    int hf, wf, wc, stride, W, K;
    float* out;                 // [K][W * W][CF]
    const long long* b_ids;     // batched form: [K] image of every match; NULL: one image
    long long feat_bs;          // floats between consecutive images (0: every match reads the same image)
};

__global__ __launch_bounds__(256) void fine2_gather_kernel(GatherArgs p) {
    const int k = blockIdx.x;
    const int cell = (int)p.ids[k];
    if (p.b_ids) p.feat += (size_t)p.b_ids[k] * p.feat_bs;
    const int cy = p.stride * (cell / p.wc), cx = p.stride * (cell % p.wc), half = p.W / 2;
    const int items = p.W * p.W * (CF / 4);
    float* dst = p.out + (size_t)k * p.W * p.W * CF;
    for (int e = threadIdx.x; e < items; e += 256) {
        const int rr = e / (CF / 4), c4 = e % (CF / 4);
        const int y = cy + rr / p.W - half, x = cx + rr % p.W - half;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < p.hf && x >= 0 && x < p.wf) v = *reinterpret_cast<const f32x4*>(p.feat + ((size_t)y * p.wf + x) * CF + 4 * c4);
        *reinterpret_cast<f32x4*>(dst + (size_t)rr * CF + 4 * c4) = v;
    }
}

// ---- y[T][NOUT] = act([xa | xb] W^T): split-bf16 MFMA, 64 rows per workgroup, weights through the L2 -> VGPR ring ---------------
struct LinArgs {
    const float* xa;            // [T][KA]
    const float* xb;            // [T][KB_] or NULL (KB_ = 0)
    int T, KA, KB_;
    const bf16x8* w_hi;         // packing.pack_linear_x3: [NOUT / 32][KIN / 16][64 lanes][8] hi plane, then the lo plane
    const bf16x8* w_lo;
    int relu;
    float* y;                   // [T][NOUT]
};

template <int KIN, int NOUT>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void rows_linear_kernel(LinArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 64, ROWB = KIN * 2, KBL = KIN / 16, NT = NOUT / 128;
    char* XH = smem;
    char* XL = smem + ROWS * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok0 = blockIdx.x * ROWS;
    // rows -> (hi, lo) planes; the concatenation [xa | xb] is formed here
    constexpr int CH = KIN / 8;
    for (int i = tid; i < ROWS * CH; i += 256) {
        const int row = i / CH, ch = i % CH, f0 = 8 * ch;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (tok0 + row < p.T) {
            const float* src = f0 < p.KA ? p.xa + (size_t)(tok0 + row) * p.KA + f0 : p.xb + (size_t)(tok0 + row) * p.KB_ + (f0 - p.KA);
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
        const int off = plane_off(row, ch, ROWB);
        *reinterpret_cast<bf16x8*>(XH + off) = vh;
        *reinterpret_cast<bf16x8*>(XL + off) = vl;
    }
    __syncthreads();
    // wave w owns output tiles w, w + 4 (NT of them), both 32-row token tiles
    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }
    constexpr int TS = KBL * 64;
    gemm_bf16<NT, 2, 3, false, KBL, 2>(acc, p.w_hi + (size_t)wave * TS + lane, p.w_lo + (size_t)wave * TS + lane, 4 * TS, XH, XL, ROWB, 0, lane);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = tok0 + 32 * tt + acc_row(reg, h);
                if (row < p.T) {
                    const float v = acc[t][tt][reg];
                    p.y[(size_t)row * NOUT + 32 * (wave + 4 * t) + r] = p.relu ? fmaxf(v, 0.f) : v;
                }
            }
}

// ---- linear attention of one match (loftr/loftr_module/linear_attention.py: Q = elu(q) + 1, K = elu(k) + 1, v / S,
//      KV = sum_s K^T v, Z = 1 / (Q . sum_s K + eps), out = (Q KV) Z S), heads of 16 -----------------------------------------------
struct AttnArgs {
    const float *q, *k, *v;     // [K][L][CF] (queries), [K][S][CF] (source)
    int L, S;
    float* msg;                 // [K][L][CF]
};

__global__ __launch_bounds__(256) void fine2_attention_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);          // [S][CF] phi(k)
    float* vs = ks + p.S * CF;                           // [S][CF] v / S
    float* kv = vs + p.S * CF;                           // [NH][DH][DH]
    float* ksum = kv + NH * DH * DH;                     // [CF]
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* kg = p.k + (size_t)m * p.S * CF;
    const float* vg = p.v + (size_t)m * p.S * CF;
    const float inv_s = 1.0f / (float)p.S;
    for (int e = tid; e < p.S * CF; e += 256) {
        const float kk = kg[e];
        ks[e] = kk > 0.f ? kk + 1.0f : expf(kk);
        vs[e] = vg[e] * inv_s;
    }
    __syncthreads();
    for (int e = tid; e < NH * DH * DH; e += 256) {      // KV[h][d][c] = sum_s K[s][h][d] v[s][h][c]
        const int hh = e / (DH * DH), d = (e / DH) % DH, c = e % DH;
        float a = 0.f;
        for (int s = 0; s < p.S; ++s) a += ks[s * CF + hh * DH + d] * vs[s * CF + hh * DH + c];
        kv[e] = a;
    }
    if (tid < CF) {
        float a = 0.f;
        for (int s = 0; s < p.S; ++s) a += ks[s * CF + tid];
        ksum[tid] = a;
    }
    __syncthreads();
    const float* qg = p.q + (size_t)m * p.L * CF;
    float* og = p.msg + (size_t)m * p.L * CF;
    for (int e = tid; e < p.L * CF; e += 256) {
        const int l = e / CF, f = e % CF, hh = f / DH, c = f % DH;
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float qq = qg[l * CF + hh * DH + d];
            const float ph = qq > 0.f ? qq + 1.0f : expf(qq);
            num += ph * kv[(hh * DH + d) * DH + c];
            den += ph * ksum[hh * DH + d];
        }
        og[e] = num * (1.0f / (den + 1e-6f)) * (float)p.S;
    }
}

// ---- y = (res ? res : 0) + LayerNorm(x) gamma + beta over rows of 128 features, one wave per row -----------------------------------
struct LnArgs {
    const float *x, *gamma, *beta, *res;
    int T;
    float* y;
};

__global__ __launch_bounds__(256) void rows_layernorm_kernel(LnArgs p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.T) return;
    const float2 v = *reinterpret_cast<const float2*>(p.x + (size_t)row * CF + 2 * lane);
    const float mean = wave_sum(v.x + v.y) * (1.0f / CF);
    const float d0 = v.x - mean, d1 = v.y - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / CF);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float2 g = *reinterpret_cast<const float2*>(p.gamma + 2 * lane), b = *reinterpret_cast<const float2*>(p.beta + 2 * lane);
    float2 o = {d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y};
    if (p.res) {
        const float2 rr = *reinterpret_cast<const float2*>(p.res + (size_t)row * CF + 2 * lane);
        o.x += rr.x; o.y += rr.y;
    }
    *reinterpret_cast<float2*>(p.y + (size_t)row * CF + 2 * lane) = o;
}

// ---- FineMatching: centre token of image 0's window against image 1's window -> heat-map expectation ----------------------------
struct MatchArgs {
    const float *f0, *f1;       // [K][WW][CF]
    const float* mk1_c;         // [K][2]
    int K, W;
    float scale;                // (W / 2) * (image height / fine height)
    float* expec;               // [K][3]
    float* mk1_f;               // [K][2]
};

__global__ __launch_bounds__(128) void fine2_match_kernel(MatchArgs p) {
    __shared__ float sim[128];
    __shared__ float cen[CF];
    const int m = blockIdx.x, tid = threadIdx.x, WW = p.W * p.W;
    cen[tid] = p.f0[((size_t)m * WW + WW / 2) * CF + tid];
    __syncthreads();
    float t = -INFINITY;
    if (tid < WW) {
        const float* row = p.f1 + ((size_t)m * WW + tid) * CF;
        float dot = 0.f;
        for (int c = 0; c < CF; ++c) dot += cen[c] * row[c];
        t = dot * 0.08838834764831845f;                   // 1 / sqrt(128)
    }
    sim[tid] = t;
    __syncthreads();
    if (tid < 64) {                                       // one wave finishes: WW <= 81 < 128 -> two elements per lane
        const float a = sim[tid], b = sim[tid + 64];
        const float mx = wave_max(fmaxf(a, b));
        const float ea = tid < WW ? expf(a - mx) : 0.f, eb = tid + 64 < WW ? expf(b - mx) : 0.f;
        const float sum = wave_sum(ea + eb);
        const float pa = ea / sum, pb = eb / sum;
        const float step = 2.0f / (float)(p.W - 1);
        const int ia = tid, ib = tid + 64;
        const float gxa = -1.0f + step * (float)(ia % p.W), gya = -1.0f + step * (float)(ia / p.W);
        const float gxb = -1.0f + step * (float)(ib % p.W), gyb = -1.0f + step * (float)(ib / p.W);
        const float ex = wave_sum(pa * gxa + pb * gxb), ey = wave_sum(pa * gya + pb * gyb);
        const float ex2 = wave_sum(pa * gxa * gxa + pb * gxb * gxb), ey2 = wave_sum(pa * gya * gya + pb * gyb * gyb);
        if (tid == 0) {
            const float vx = ex2 - ex * ex, vy = ey2 - ey * ey;
            p.expec[3 * m] = ex; p.expec[3 * m + 1] = ey;
            p.expec[3 * m + 2] = sqrtf(fmaxf(vx, 1e-10f)) + sqrtf(fmaxf(vy, 1e-10f));
            p.mk1_f[2 * m] = p.mk1_c[2 * m] + ex * p.scale;
            p.mk1_f[2 * m + 1] = p.mk1_c[2 * m + 1] + ey * p.scale;
        }
    }
}

}  // namespace

extern "C" int ophip_fine2_gather(const float* feat_cl, int hf, int wf, const long long* cell_ids, int K, int wc, int stride, int W,
                                  float* out, void* stream_) {
    if (!feat_cl || !cell_ids || !out) return ophip_bad_arg(__func__, "null pointer");
    if (K < 0 || W < 1 || (W & 1) == 0 || W * W > 128 || hf < 1 || wf < 1 || wc < 1 || stride < 1) return ophip_bad_arg(__func__, "bad sizes (odd window <= 11)");
    if (K == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    GatherArgs a{feat_cl, cell_ids, hf, wf, wc, stride, W, K, out, nullptr, 0};
    OPHIP_LAUNCH("fine2_gather", stream, fine2_gather_kernel, dim3(K), dim3(256), 0, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

// the same gather over a BATCH of images: match k reads image b_ids[k] of feat_cl [B][hf * wf][128] (feat_bstride floats apart; 0 = one image
// shared by every match: the detector's query frame).  For the detector's one batched matcher call over all reference views.
extern "C" int ophip_fine2_gather_b(const float* feat_cl, long long feat_bstride, const long long* b_ids, int hf, int wf, const long long* cell_ids,
                                    int K, int wc, int stride, int W, float* out, void* stream_) {
    if (!feat_cl || !cell_ids || !out || !b_ids) return ophip_bad_arg(__func__, "null pointer");
    if (K < 0 || W < 1 || (W & 1) == 0 || W * W > 128 || hf < 1 || wf < 1 || wc < 1 || stride < 1 || feat_bstride < 0) return ophip_bad_arg(__func__, "bad sizes (odd window <= 11)");
    if (K == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    GatherArgs a{feat_cl, cell_ids, hf, wf, wc, stride, W, K, out, b_ids, feat_bstride};
    OPHIP_LAUNCH("fine2_gather", stream, fine2_gather_kernel, dim3(K), dim3(256), 0, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t ophip_rows_linear_wpack_bytes(int kin, int nout) { return (size_t)kin * nout * 4; }

extern "C" int ophip_rows_linear_x3(const float* xa, int ka, const float* xb, int kb, int T, const void* wpack, int nout, int relu,
                                    float* y, void* stream_) {
    if (!xa || !wpack || !y || (kb > 0 && !xb)) return ophip_bad_arg(__func__, "null pointer");
    const int kin = ka + kb;
    if (T < 0 || (ka % 8) || (kb % 8) || !((kin == 128 || kin == 256) && (nout == 128 || nout == 256)))
        return ophip_bad_arg(__func__, "sizes: K in {128, 256} (two sources, multiples of 8), N in {128, 256}");
    if (T == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    LinArgs a{xa, xb, T, ka, kb, reinterpret_cast<const bf16x8*>(wpack), reinterpret_cast<const bf16x8*>(wpack) + (size_t)kin * nout / 8, relu, y};
    const int grid = (T + 63) / 64;
    const size_t lds = (size_t)2 * 64 * kin * 2;
#define OPHIP_LIN_CASE(KIN_, NOUT_)                                                                                              \
    if (kin == KIN_ && nout == NOUT_) {                                                                                          \
        if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(rows_linear_kernel<KIN_, NOUT_>), lds, "hipFuncSetAttribute(rows_linear)")) return rc; \
        OPHIP_LAUNCH("rows_linear", stream, (rows_linear_kernel<KIN_, NOUT_>), dim3(grid), dim3(256), lds, stream, a);          \
    }
    OPHIP_LIN_CASE(128, 128) OPHIP_LIN_CASE(128, 256) OPHIP_LIN_CASE(256, 128) OPHIP_LIN_CASE(256, 256)
#undef OPHIP_LIN_CASE
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_fine2_attention(const float* q, const float* k, const float* v, int K, int L, int S, float* msg, void* stream_) {
    if (!q || !k || !v || !msg) return ophip_bad_arg(__func__, "null pointer");
    if (K < 0 || L < 1 || S < 1 || S > 128) return ophip_bad_arg(__func__, "bad sizes (source length <= 128)");
    if (K == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a{q, k, v, L, S, msg};
    const size_t lds = ((size_t)2 * S * CF + NH * DH * DH + CF) * sizeof(float);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(fine2_attention_kernel), lds, "hipFuncSetAttribute(fine2_attention)")) return rc;
    OPHIP_LAUNCH("fine2_attention", stream, fine2_attention_kernel, dim3(K), dim3(256), lds, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_rows_layernorm128(const float* x, const float* gamma, const float* beta, const float* residual, int T, float* y, void* stream_) {
    if (!x || !gamma || !beta || !y) return ophip_bad_arg(__func__, "null pointer");
    if (T < 0) return ophip_bad_arg(__func__, "T");
    if (T == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    LnArgs a{x, gamma, beta, residual, T, y};
    OPHIP_LAUNCH("rows_layernorm", stream, rows_layernorm_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_fine2_match(const float* f0, const float* f1, const float* mkpts1_c, int K, int W, float scale, float* expec_f,
                                 float* mkpts1_f, void* stream_) {
    if (!f0 || !f1 || !mkpts1_c || !expec_f || !mkpts1_f) return ophip_bad_arg(__func__, "null pointer");
    if (K < 0 || W < 3 || (W & 1) == 0 || W * W > 128) return ophip_bad_arg(__func__, "bad sizes (odd window, 3 .. 11)");
    if (K == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    MatchArgs a{f0, f1, mkpts1_c, K, W, scale, expec_f, mkpts1_f};
    OPHIP_LAUNCH("fine2_match", stream, fine2_match_kernel, dim3(K), dim3(128), 0, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
