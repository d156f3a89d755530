// Fine-level refinement of the coarse matches, one workgroup per match, everything in LDS:
//   gather the 5x5 window of the 1/2-resolution feature map (no 61 MB unfold) + the 3D fine descriptor
//   -> 2-layer LoFTR encoder at d_model 128 (8 heads x 16; window stream 25 tokens, 3D stream 1 token)
//   -> local correlation, softmax heat-map, spatial expectation, std, sub-pixel query keypoint.
// Reference: loftr_module/fine_preprocess.py:32-55 (a9), loftr_module/transformer.py:65-171 with the
// loftr_fine config (a10), utils/fine_matching.py:28-110 (a11).
//
// Token tile of a match: rows 0..24 = window (row-major ky,kx), row 25 = the 3D token, rows 26..31 = padding.
// The same layer weights serve both streams (transformer.py:148-159), so every per-token GEMM runs once on
// the whole tile; only the attention differs by stream:  self: window<-window, 3D<-3D;  cross: window<-3D(old),
// 3D<-window(old).  KV / Ksum of both source sets are formed with accumulator-as-operand MFMAs (tile.h) and
// consumed straight from registers as the B operand of phi(Q) KV.  With D = 16 a 32-wide tile holds two
// heads, so KV tiles are masked to their block diagonal.
#include "tile.h"

namespace {

constexpr int CF = 128;
constexpr int LDF = CF + OPHIP_PAD, LDF2 = 2 * CF + OPHIP_PAD;
constexpr int WIN = 25, TOK3D = 25;
constexpr int LAYER_FLOATS = 3 * CF * CF + CF * CF + 4 * CF * CF + 2 * CF * CF + 4 * CF;   // Wqkv | Wm | W0 | W2 | ln

struct FineArgs {
    const float* feat_f; long long fs_b, fs_c, fs_y, fs_x; int hf, wf;
    const float* desc_f; long long ds_b, ds_c;          // [B][128][N], n-stride 1
    const long long *b_ids, *i_ids, *j_ids;
    const int* count;
    const float* mkq_c;
    const float* wpack;          // nlayers x LAYER_FLOATS
    int nlayers; unsigned cross_bits;                    // bit l set: layer l is "cross"
    int enc_enable;
    int wc, stride;
    float fine_scale;            // (W // 2) * (H_img / H_f)
    const float* qscale;         // [B][2] query_image_scale (h, w factors; fine_matching.py:104) or NULL
    float* expec_f; float* mkq_f;
    float* dbg_win; float* dbg_f3;                       // optional [K][25][128], [K][128]
};

__device__ __forceinline__ f32x4 frag_of(const f32x16& a, int kb) {
    f32x4 v = {a[4 * kb], a[4 * kb + 1], a[4 * kb + 2], a[4 * kb + 3]};
    return v;
}

__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void fine_refine_kernel(FineArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                     // [32][LDF]
    float* P = X + 32 * LDF;             // [32][LDF]
    float* Hh = P + 32 * LDF;            // [32][LDF2]
    const int k = blockIdx.x;
    if (k >= *p.count) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = (int)p.b_ids[k], i3 = (int)p.i_ids[k], j = (int)p.j_ids[k];
    const int cy = p.stride * (j / p.wc), cx = p.stride * (j % p.wc);

    // ---- gather ------------------------------------------------------------------------------
    const float* ff = p.feat_f + (size_t)b * p.fs_b;
    if (p.fs_c == 1) {                   // channels-last memory: 512 B contiguous per pixel
        for (int e = tid; e < WIN * CF; e += 256) {
            const int rr = e >> 7, c = e & 127;
            const int y = cy + rr / 5 - 2, x = cx + rr % 5 - 2;
            float v = 0.f;
            if (y >= 0 && y < p.hf && x >= 0 && x < p.wf) v = ff[(size_t)y * p.fs_y + (size_t)x * p.fs_x + c];
            X[rr * LDF + c] = v;
        }
    } else {                             // NCHW: one thread per (channel, window row): 5 consecutive x
        for (int q = tid; q < CF * 5; q += 256) {
            const int c = q / 5, ky = q % 5;
            const int y = cy + ky - 2;
            const bool yin = (y >= 0 && y < p.hf);
            const float* src = ff + (size_t)c * p.fs_c + (size_t)(yin ? y : 0) * p.fs_y;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int x = cx + kx - 2;
                float v = 0.f;
                if (yin && x >= 0 && x < p.wf) v = src[(size_t)x * p.fs_x];
                X[(ky * 5 + kx) * LDF + c] = v;
            }
        }
    }
    if (tid < CF) X[TOK3D * LDF + tid] = p.desc_f[(size_t)b * p.ds_b + (size_t)tid * p.ds_c + i3];
    for (int e = tid; e < 6 * LDF; e += 256) X[26 * LDF + e] = 0.f;
    __syncthreads();

    constexpr int KB = CF / 8, TS = KB * 64;            // K = 128
    constexpr int KB2 = 2 * CF / 8, TS2 = KB2 * 64;     // K = 256
    const float* xa = X + r * LDF + 4 * h;
    const float* pa = P + r * LDF + 4 * h;
    const float* ha = Hh + r * LDF2 + 4 * h;

    const int nl = p.enc_enable ? p.nlayers : 0;
    for (int l = 0; l < nl; ++l) {
        const float* wl = p.wpack + (size_t)l * LAYER_FLOATS;
        const f32x4* wqkv = reinterpret_cast<const f32x4*>(wl);
        const f32x4* wm = reinterpret_cast<const f32x4*>(wl + 3 * CF * CF);
        const f32x4* w0 = reinterpret_cast<const f32x4*>(wl + 4 * CF * CF);
        const f32x4* w2 = reinterpret_cast<const f32x4*>(wl + 8 * CF * CF);
        const float* ln = wl + 10 * CF * CF;
        const bool cross = (p.cross_bits >> l) & 1u;

        // ---- Q, K, V for this wave's two heads (packed rows per wave: Q | K | V tile) ---------
        f32x16 qkv[3] = {zero16(), zero16(), zero16()};
        gemm_lds_x_packed<3>(qkv, xa, KB, wqkv + (size_t)(3 * wave) * TS + lane, TS);
        f32x16 kw, k3;          // phi(K) restricted to the window rows / to the 3D row
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = acc_row(reg, h);
            qkv[0][reg] = elu_plus_one(qkv[0][reg]);
            const float pk = elu_plus_one(qkv[1][reg]);
            kw[reg] = row < WIN ? pk : 0.f;
            k3[reg] = row == TOK3D ? pk : 0.f;
            // values / v_length of the stream the token belongs to (25 window tokens, 1 3D token)
            qkv[2][reg] = row < WIN ? qkv[2][reg] / 25.0f : (row == TOK3D ? qkv[2][reg] / 1.0f : 0.f);
        }
        acc_to_lds(qkv[0], P, LDF, 32 * wave, lane);
        f32x16 kvw = zero16(), kv3 = zero16(), ksw = zero16(), ks3 = zero16();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            kvw = __builtin_amdgcn_mfma_f32_32x32x2f32(kw[reg], qkv[2][reg], kvw, 0, 0, 0);
            kv3 = __builtin_amdgcn_mfma_f32_32x32x2f32(k3[reg], qkv[2][reg], kv3, 0, 0, 0);
            ksw = __builtin_amdgcn_mfma_f32_32x32x2f32(kw[reg], 1.0f, ksw, 0, 0, 0);
            ks3 = __builtin_amdgcn_mfma_f32_32x32x2f32(k3[reg], 1.0f, ks3, 0, 0, 0);
        }
        // two heads per tile: keep the block diagonal (d and v in the same 16-wide head)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const bool same = (acc_row(reg, h) >> 4) == (r >> 4);
            if (!same) { kvw[reg] = 0.f; kv3[reg] = 0.f; ksw[reg] = 0.f; ks3[reg] = 0.f; }
        }
        __syncthreads();
        f32x16 nw = zero16(), n3 = zero16(), dw = zero16(), d3 = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 aq = *reinterpret_cast<const f32x4*>(pa + 32 * wave + 8 * kb);
            nw = mfma4(aq, frag_of(kvw, kb), nw);
            n3 = mfma4(aq, frag_of(kv3, kb), n3);
            dw = mfma4(aq, frag_of(ksw, kb), dw);
            d3 = mfma4(aq, frag_of(ks3, kb), d3);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const bool is3d = acc_row(reg, h) == TOK3D;
            const bool use_w = cross ? is3d : !is3d;          // which source set this token attends to
            const float num = use_w ? nw[reg] : n3[reg];
            const float den = use_w ? dw[reg] : d3[reg];
            const float S = use_w ? 25.0f : 1.0f;
            nw[reg] = num * (1.0f / (den + 1e-6f)) * S;
        }
        acc_to_lds(nw, Hh, LDF2, 32 * wave, lane);
        __syncthreads();
        // ---- merge + LN1 -----------------------------------------------------------------------
        {
            f32x16 m[1] = {zero16()};
            gemm_lds_x_packed<1>(m, ha, KB, wm + (size_t)wave * TS + lane, TS);
            acc_to_lds(m[0], P, LDF, 32 * wave, lane);
        }
        __syncthreads();
        rows_layernorm<CF, true, false>(P, LDF, ln, ln + CF, 1e-5f, wave, lane);
        __syncthreads();
        // ---- MLP -------------------------------------------------------------------------------
        {
            f32x16 hid[2] = {zero16(), zero16()};
            const f32x4* w0w = w0 + (size_t)(2 * wave) * TS2 + lane;
            gemm_lds_x_packed<2>(hid, xa, KB, w0w, TS2);
            gemm_lds_x_packed<2>(hid, pa, KB, w0w + (size_t)KB * 64, TS2);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) hid[t][reg] = fmaxf(hid[t][reg], 0.f);
                acc_to_lds(hid[t], Hh, LDF2, 64 * wave + 32 * t, lane);
            }
        }
        __syncthreads();
        {
            f32x16 o[1] = {zero16()};
            gemm_lds_x_packed<1>(o, ha, KB2, w2 + (size_t)wave * TS2 + lane, TS2);
            acc_to_lds(o[0], P, LDF, 32 * wave, lane);
        }
        __syncthreads();
        rows_layernorm<CF, true, false>(P, LDF, ln + 2 * CF, ln + 3 * CF, 1e-5f, wave, lane);
        // residual, in place (each wave owns rows 8w..8w+7 here and in the LayerNorm above)
        for (int rr = 0; rr < 8; ++rr) {
            const int row = 8 * wave + rr;
            X[row * LDF + lane] += P[row * LDF + lane];
            X[row * LDF + 64 + lane] += P[row * LDF + 64 + lane];
        }
        __syncthreads();
    }

    if (p.dbg_win) {
        for (int e = tid; e < WIN * CF; e += 256) p.dbg_win[(size_t)k * WIN * CF + e] = X[(e >> 7) * LDF + (e & 127)];
        if (tid < CF) p.dbg_f3[(size_t)k * CF + tid] = X[TOK3D * LDF + tid];
    }

    // ---- fine matching: correlation -> softmax -> expectation (wave 0) ------------------------------
    if (wave == 0) {
        float t = -INFINITY;
        if (lane < WIN) {
            float dot = 0.f;
            for (int c = 0; c < CF; ++c) dot += X[TOK3D * LDF + c] * X[lane * LDF + c];
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
            store_fine_keypoint(p, k, ex, ey);
        }
    }
}

}  // namespace

namespace {
int fine_f32(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
             const float* desc3d_f, long long ds_b, long long ds_c,
             const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
             const float* mkpts_c, const float* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
             int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
             float* dbg_win, float* dbg_f3, const float* query_scale, void* stream_) {
    if (!feat_f || !desc3d_f || !b_ids || !i_ids || !j_ids || !count || !mkpts_c || !expec_f || !mkpts_f)
        return ophip_bad_arg(__func__, "null pointer");
    if (encoder_enable && (!wpack || nlayers < 1 || nlayers > 32)) return ophip_bad_arg(__func__, "encoder enabled without weights");
    if ((dbg_win == nullptr) != (dbg_f3 == nullptr)) return ophip_bad_arg(__func__, "dbg_win and dbg_f3 go together");
    if (max_matches <= 0) return 0;
    FineArgs a;
    a.feat_f = feat_f; a.fs_b = fs_b; a.fs_c = fs_c; a.fs_y = fs_y; a.fs_x = fs_x; a.hf = hf; a.wf = wf;
    a.desc_f = desc3d_f; a.ds_b = ds_b; a.ds_c = ds_c;
    a.b_ids = b_ids; a.i_ids = i_ids; a.j_ids = j_ids; a.count = count; a.mkq_c = mkpts_c;
    a.wpack = wpack; a.nlayers = nlayers; a.cross_bits = cross_bits; a.enc_enable = encoder_enable;
    a.wc = wc; a.stride = stride; a.fine_scale = fine_scale; a.qscale = query_scale;
    a.expec_f = expec_f; a.mkq_f = mkpts_f; a.dbg_win = dbg_win; a.dbg_f3 = dbg_f3;
    const size_t lds = (size_t)32 * (2 * LDF + LDF2) * sizeof(float);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(fine_refine_kernel), lds, "hipFuncSetAttribute(fine_refine)")) return rc;
    OPHIP_LAUNCH("fine_refine", (hipStream_t)stream_, fine_refine_kernel, dim3(max_matches), dim3(256), lds, (hipStream_t)stream_, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int ophip_fine_refine(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                                 const float* desc3d_f, long long ds_b, long long ds_c,
                                 const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                                 const float* mkpts_c, const float* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
                                 int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                                 float* dbg_win, float* dbg_f3, void* stream) {
    return fine_f32(feat_f, fs_b, fs_c, fs_y, fs_x, hf, wf, desc3d_f, ds_b, ds_c, b_ids, i_ids, j_ids, count, max_matches, mkpts_c, wpack, nlayers,
                    cross_bits, encoder_enable, wc, stride, fine_scale, expec_f, mkpts_f, dbg_win, dbg_f3, nullptr, stream);
}

// ophip_fine_refine with query_scale [B][2] = data["query_image_scale"] ((h, w) factors; fine_matching.py:104): the refinement offset of
// match k is scaled by query_scale[b_ids[k]][[1, 0]].
extern "C" int ophip_fine_refine_scaled(const float* feat_f, long long fs_b, long long fs_c, long long fs_y, long long fs_x, int hf, int wf,
                                        const float* desc3d_f, long long ds_b, long long ds_c,
                                        const long long* b_ids, const long long* i_ids, const long long* j_ids, const int* count, int max_matches,
                                        const float* mkpts_c, const float* wpack, int nlayers, unsigned cross_bits, int encoder_enable,
                                        int wc, int stride, float fine_scale, float* expec_f, float* mkpts_f,
                                        float* dbg_win, float* dbg_f3, const float* query_scale, void* stream) {
    if (!query_scale) return ophip_bad_arg(__func__, "null query_scale (use ophip_fine_refine)");
    return fine_f32(feat_f, fs_b, fs_c, fs_y, fs_x, hf, wf, desc3d_f, ds_b, ds_c, b_ids, i_ids, j_ids, count, max_matches, mkpts_c, wpack, nlayers,
                    cross_bits, encoder_enable, wc, stride, fine_scale, expec_f, mkpts_f, dbg_win, dbg_f3, query_scale, stream);
}
