// Encoder inputs: positional-encoding add + NCHW->NLC transpose of the coarse feature map, and the
// 3D keypoint encoding (normalise -> MLP 3->32->64->128->256 with per-point feature norm -> add to the
// coarse 3D descriptors, emitted token-major).
// Reference: utils/position_encoding.py:37-42 + OnePosePlusModel.py:135-140 (a1),
// utils/normalize.py:17-28 (a2), utils/position_encoding.py:54-79 (a3).
#include "tile.h"

namespace {

// ---------------------------------------------------------------------------------------------
// out[b][m][c] = feat[b][c][m] + pe[m][c]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pe_add_transpose_kernel(const float* __restrict__ feat, const float* __restrict__ pe,
                                                               float* __restrict__ out, int Cc, int M) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
    const float* f = feat + (size_t)b * Cc * M;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, m = m0 + tx;
        tile[ty + 8 * k][tx] = (c < Cc && m < M) ? f[(size_t)c * M + m] : 0.f;
    }
    __syncthreads();
    float* o = out + (size_t)b * M * Cc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = m0 + ty + 8 * k, c = c0 + tx;
        if (m < M && c < Cc) o[(size_t)m * Cc + c] = tile[tx][ty + 8 * k] + (pe ? pe[(size_t)m * Cc + c] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// stats[4*b .. 4*b+2] = mean of keypoints3d[b];  stats[4*B] = 0.6 * max extent of batch element 0
// ---------------------------------------------------------------------------------------------
constexpr int KS_THREADS = 1024, KS_WAVES = KS_THREADS / 64;

__global__ __launch_bounds__(KS_THREADS) void kpt_stats_kernel(const float* __restrict__ kpts, long long bs, float* __restrict__ stats, int B, int N) {
    __shared__ float red[KS_WAVES][9];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* k = kpts + (size_t)b * bs;
    float s[3] = {0.f, 0.f, 0.f}, mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
    for (int i = tid; i < N; i += KS_THREADS) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = k[(size_t)i * 3 + d];
            s[d] += v; mn[d] = fminf(mn[d], v); mx[d] = fmaxf(mx[d], v);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        s[d] = wave_sum(s[d]);
        mx[d] = wave_max(mx[d]);
        mn[d] = -wave_max(-mn[d]);
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[wave][d] = s[d]; red[wave][3 + d] = mn[d]; red[wave][6 + d] = mx[d]; }
    }
    __syncthreads();
    if (tid == 0) {
        float ext = 0.f;
        for (int d = 0; d < 3; ++d) {
            float sum = 0.f, lo = INFINITY, hi = -INFINITY;
            for (int w = 0; w < KS_WAVES; ++w) {         // fixed order: the mean is run-to-run reproducible
                sum += red[w][d];
                lo = fminf(lo, red[w][3 + d]);
                hi = fmaxf(hi, red[w][6 + d]);
            }
            stats[4 * b + d] = sum / (float)N;
            ext = fmaxf(ext, hi - lo);
        }
        stats[4 * b + 3] = 0.f;
        if (b == 0) stats[4 * B] = ext * 0.6f;
    }
}

// ---------------------------------------------------------------------------------------------
// keypoint encoder: 32 points per workgroup, MLP tiles on MFMA, per-point norm in LDS
// ---------------------------------------------------------------------------------------------
constexpr int KD = 256;                    // descriptor_dim
constexpr int L0 = 8 + OPHIP_PAD, L1 = 32 + OPHIP_PAD, L2 = 64 + OPHIP_PAD, L3 = 128 + OPHIP_PAD, LO = KD + OPHIP_PAD;

struct KptArgs {
    const float* kpts; long long kbs;     // [B][N][3]
    const float* stats;                   // centres + scaling
    const float* desc; long long dbs;     // [B][256][N]
    const f32x4 *w1, *w2, *w3, *w4;       // packed
    const float *b1, *b2, *b3, *b4;
    float* out;                           // [B][N][256]
    int B, N;
};

__global__ __launch_bounds__(256) void kpt_encode_kernel(KptArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A0 = smem;                       // [32][L0]
    float* A1 = A0 + 32 * L0;               // [32][L1]
    float* A2 = A1 + 32 * L1;               // [32][L2]
    float* A3 = A2 + 32 * L2;               // [32][L3]
    float* Ot = A3 + 32 * L3;               // [32][LO] descriptor tile, point-major; the MLP's output is added in place
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, b = blockIdx.y;
    const float scaling = p.stats[4 * p.B];
    // normalised keypoints, zero-padded to K = 8
    if (tid < 32 * 8) {
        const int row = tid >> 3, d = tid & 7;
        float v = 0.f;
        if (d < 3 && n0 + row < p.N) v = (p.kpts[(size_t)b * p.kbs + (size_t)(n0 + row) * 3 + d] - p.stats[4 * b + d]) / scaling;
        A0[row * L0 + d] = v;
    }
    // descriptor tile: read coalesced along n, written transposed (one tile instead of a channel-major and a point-major one: 65 KB
    // of LDS per workgroup instead of 99, so that the kernel fits beside a fine-stage workgroup on a CU)
    // (all 32 loads of a thread are issued before the first LDS write: as a rolled loop the compiler waited for each load in turn --
    //  32 dependent HBM round trips, 25 of the kernel's 31 us at c2)
    const float* dsc = p.desc + (size_t)b * p.dbs;
    {
        float dv[KD * 32 / 256];
        const int n = tid & 31, cb = tid >> 5;
        const bool live = n0 + n < p.N;
#pragma unroll
        for (int k = 0; k < KD * 32 / 256; ++k) dv[k] = live ? dsc[(size_t)(cb + 8 * k) * p.N + n0 + n] : 0.f;
#pragma unroll
        for (int k = 0; k < KD * 32 / 256; ++k) Ot[n * LO + cb + 8 * k] = dv[k];
    }
    __syncthreads();
    // 3 -> 32
    if (wave == 0) {
        f32x16 acc[1] = {zero16()};
        gemm_lds_x_packed<1, 1>(acc, A0 + r * L0 + 4 * h, 1, p.w1 + lane, 64);
        const float bias = p.b1[r];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[0][reg] += bias;
        acc_to_lds(acc[0], A1, L1, 0, lane);
    }
    __syncthreads();
    rows_layernorm<32, false, true>(A1, L1, nullptr, nullptr, 1e-5f, wave, lane);
    __syncthreads();
    // 32 -> 64
    if (wave < 2) {
        f32x16 acc[1] = {zero16()};
        gemm_lds_x_packed<1>(acc, A1 + r * L1 + 4 * h, 4, p.w2 + (size_t)wave * 4 * 64 + lane, 4 * 64);
        const float bias = p.b2[32 * wave + r];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[0][reg] += bias;
        acc_to_lds(acc[0], A2, L2, 32 * wave, lane);
    }
    __syncthreads();
    rows_layernorm<64, false, true>(A2, L2, nullptr, nullptr, 1e-5f, wave, lane);
    __syncthreads();
    // 64 -> 128
    {
        f32x16 acc[1] = {zero16()};
        gemm_lds_x_packed<1>(acc, A2 + r * L2 + 4 * h, 8, p.w3 + (size_t)wave * 8 * 64 + lane, 8 * 64);
        const float bias = p.b3[32 * wave + r];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[0][reg] += bias;
        acc_to_lds(acc[0], A3, L3, 32 * wave, lane);
    }
    __syncthreads();
    rows_layernorm<128, false, true>(A3, L3, nullptr, nullptr, 1e-5f, wave, lane);
    __syncthreads();
    // 128 -> 256, + bias, + descriptors
    {
        f32x16 acc[2] = {zero16(), zero16()};
        gemm_lds_x_packed<2>(acc, A3 + r * L3 + 4 * h, 16, p.w4 + (size_t)(2 * wave) * 16 * 64 + lane, 16 * 64);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = 64 * wave + 32 * t + r;
            const float bias = p.b4[c];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = acc_row(reg, h);
                Ot[row * LO + c] = Ot[row * LO + c] + (acc[t][reg] + bias);
            }
        }
    }
    __syncthreads();
    float* o = p.out + (size_t)b * p.N * KD;
    for (int rr = 0; rr < 8; ++rr) {
        const int row = 8 * wave + rr;
        if (n0 + row < p.N)
            *reinterpret_cast<f32x4*>(o + (size_t)(n0 + row) * KD + 4 * lane) = *reinterpret_cast<const f32x4*>(Ot + row * LO + 4 * lane);
    }
}

// plain [B][C][L] -> [B][L][C] (keypoint encoding disabled: the encoder still wants token-major input)
__global__ __launch_bounds__(256) void transpose_cl_kernel(const float* __restrict__ in, float* __restrict__ out, int Cc, int L) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int l0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, l = l0 + tx;
        tile[ty + 8 * k][tx] = (c < Cc && l < L) ? in[((size_t)b * Cc + c) * L + l] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = l0 + ty + 8 * k, c = c0 + tx;
        if (l < L && c < Cc) out[((size_t)b * L + l) * Cc + c] = tile[tx][ty + 8 * k];
    }
}

// [B][128][L] -> [B][L][128]: 32 positions x 128 channels per workgroup, 128-byte reads, whole 512-byte rows written
// (the fine feature map goes channels-last once per frame so that a 5x5 window gather reads 25 contiguous rows
// instead of 640 20-byte runs -- measured 328 MB -> 8x over-fetch on the NCHW gather, profiles/r01_pmc_traffic.json)
__global__ __launch_bounds__(256) void transpose_c128_kernel(const float* __restrict__ in, float* __restrict__ out, int L) {
    __shared__ float tile[128][33];
    const int tid = threadIdx.x, l0 = blockIdx.x * 32, b = blockIdx.y;
    const float* src = in + (size_t)b * 128 * L;
    const int lx = tid & 31, cy = tid >> 5;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = cy + 8 * k;
        tile[c][lx] = (l0 + lx < L) ? src[(size_t)c * L + l0 + lx] : 0.f;
    }
    __syncthreads();
    float* dst = out + (size_t)b * L * 128;
    const int p = tid >> 3, q0 = tid & 7;
    if (l0 + p < L) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int q = q0 + 8 * m;              // float4 index inside the 128-channel row
            f32x4 v = {tile[4 * q][p], tile[4 * q + 1][p], tile[4 * q + 2][p], tile[4 * q + 3][p]};
            *reinterpret_cast<f32x4*>(dst + (size_t)(l0 + p) * 128 + 4 * q) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// query crop: bbox [x0, y0, x1, y1) of a full-resolution grayscale frame -> S x S float image in [0, 1]
// (row f-2; local_feature_2D_detector.py:164-190 crop_img_by_bbox = two cv2.warpAffine calls: an integer-shift crop to the
// box, then an isotropic resize by S / box_width about the crop centre with zero border).  One bilinear pass here.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_resize_kernel(const unsigned char* __restrict__ img, int H, int W, int x0, int y0, int wb, int hb,
                                                          int S, float* __restrict__ out) {
    const int X = blockIdx.x * 32 + (threadIdx.x & 31), Y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (X >= S || Y >= S) return;
    const float inv = (float)wb / (float)S;                    // 1 / scale of the second warp
    const float u = ((float)X - 0.5f * (float)S) * inv + 0.5f * (float)wb;
    const float v = ((float)Y - 0.5f * (float)S) * inv + 0.5f * (float)hb;
    const float fu = floorf(u), fv = floorf(v);
    const int i0 = (int)fu, j0 = (int)fv;
    const float a = u - fu, b = v - fv;
    auto px = [&](int i, int j) -> float {                     // the box crop with zeros outside it and outside the frame
        if (i < 0 || i >= wb || j < 0 || j >= hb) return 0.f;
        const int x = x0 + i, y = y0 + j;
        return (x >= 0 && x < W && y >= 0 && y < H) ? (float)img[(size_t)y * W + x] : 0.f;
    };
    const float val = (1.f - b) * ((1.f - a) * px(i0, j0) + a * px(i0 + 1, j0)) + b * ((1.f - a) * px(i0, j0 + 1) + a * px(i0 + 1, j0 + 1));
    out[(size_t)Y * S + X] = fminf(fmaxf(rintf(val), 0.f), 255.f) / 255.0f;      // cv2 writes uint8; astype(float32) / 255 is a true division
}

}  // namespace

extern "C" int ophip_crop_resize_gray(const unsigned char* image, int H, int W, int x0, int y0, int x1, int y1, int S, float* out, void* stream) {
    if (!image || !out) return ophip_bad_arg(__func__, "null pointer");
    if (H < 1 || W < 1 || S < 1 || x1 <= x0 || y1 <= y0) return ophip_bad_arg(__func__, "bad sizes (need x1 > x0, y1 > y0)");
    OPHIP_LAUNCH("crop_resize", (hipStream_t)stream, crop_resize_kernel, dim3((S + 31) / 32, (S + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                 image, H, W, x0, y0, x1 - x0, y1 - y0, S, out);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_pe_add_transpose(const float* feat_nchw, const float* pe_nlc, float* out_nlc, int B, int C, int M, void* stream) {
    if (!feat_nchw || !out_nlc) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || C < 1 || M < 1) return ophip_bad_arg(__func__, "bad sizes");
    OPHIP_LAUNCH("pe_add_transpose", (hipStream_t)stream, pe_add_transpose_kernel, dim3((M + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream,
                       feat_nchw, pe_nlc, out_nlc, C, M);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_transpose_cl(const float* in_bcl, float* out_blc, int B, int C, int L, void* stream) {
    if (!in_bcl || !out_blc) return ophip_bad_arg(__func__, "null pointer");
    if (C == 128) {
        OPHIP_LAUNCH("transpose_cl", (hipStream_t)stream, transpose_c128_kernel, dim3((L + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, in_bcl, out_blc, L);
        OPHIP_CHECK_LAUNCH();
        return 0;
    }
    OPHIP_LAUNCH("transpose_cl", (hipStream_t)stream, transpose_cl_kernel, dim3((L + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, in_bcl, out_blc, C, L);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int ophip_kpt_encode(const float* keypoints3d, long long kpts_bstride, const float* desc_bcn, long long desc_bstride,
                                const float* wpack, float* stats, float* out_bnc, int B, int N, void* stream_) {
    if (!keypoints3d || !desc_bcn || !wpack || !stats || !out_bnc) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || N < 1) return ophip_bad_arg(__func__, "bad sizes");
    hipStream_t stream = (hipStream_t)stream_;
    OPHIP_LAUNCH("kpt_stats", stream, kpt_stats_kernel, dim3(B), dim3(KS_THREADS), 0, stream, keypoints3d, kpts_bstride, stats, B, N);
    OPHIP_CHECK_LAUNCH();
    // packed block (floats): W1[32x8] | W2[64x32] | W3[128x64] | W4[256x128] | b1 | b2 | b3 | b4
    KptArgs a;
    a.kpts = keypoints3d; a.kbs = kpts_bstride; a.stats = stats; a.desc = desc_bcn; a.dbs = desc_bstride;
    const float* w = wpack;
    a.w1 = reinterpret_cast<const f32x4*>(w); w += 32 * 8;
    a.w2 = reinterpret_cast<const f32x4*>(w); w += 64 * 32;
    a.w3 = reinterpret_cast<const f32x4*>(w); w += 128 * 64;
    a.w4 = reinterpret_cast<const f32x4*>(w); w += 256 * 128;
    a.b1 = w; a.b2 = w + 32; a.b3 = w + 96; a.b4 = w + 224;
    a.out = out_bnc; a.B = B; a.N = N;
    const size_t lds = (size_t)(32 * (L0 + L1 + L2 + L3 + LO)) * sizeof(float);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(kpt_encode_kernel), lds, "hipFuncSetAttribute(kpt_encode)")) return rc;
    OPHIP_LAUNCH("kpt_encode", stream, kpt_encode_kernel, dim3((N + 31) / 32, B), dim3(256), lds, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
