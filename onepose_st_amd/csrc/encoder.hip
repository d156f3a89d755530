// Coarse LoFTR encoder layer (d_model 256, 8 heads x 32) for the 3D-point stream and the 2D-grid
// stream of one frame batch -- reference: loftr_module/transformer.py:65-94 (LoFTREncoderLayer),
// :133-171 (stream wiring), loftr_module/linear_attention.py:29-61 (phi(Q) (phi(K)^T V)).
//
// One layer = three launches, both streams in each launch:
//   kv_reduce   (over SOURCE tokens)  K,V projections on MFMA -> phi(K), V/S -> per-tile partial
//               KV_h = phi(K_h)^T V_h (32x32 per head) and Ksum_h, accumulator-as-operand (no LDS)
//   kv_sum      deterministic fixed-order sum of the per-tile partials (no float atomics)
//   attn_apply  (over QUERY tokens)   Q projection -> phi(Q) -> (phi(Q) KV)/(phi(Q) Ksum + eps)
//               -> merge -> LayerNorm -> [x, msg] -> MLP 512->512 ReLU ->256 -> LayerNorm -> x + msg
// Activations are [B, L, 256] f32 row-major in HBM; a workgroup owns 32 tokens, 4 waves split the
// output features; tiles live in LDS between the chained GEMMs (tile.h).
#include "tile.h"

namespace {

constexpr int C = 256;          // d_model
constexpr int NH = 8;           // heads
constexpr int LDX = C + OPHIP_PAD;
constexpr int LDH = 2 * C + OPHIP_PAD;
constexpr int KV_PER_HEAD = 1024 + 32;          // KV fragments + Ksum
constexpr int KV_FLOATS = NH * KV_PER_HEAD;     // 8448 per (batch, stream)

struct KvReduceArgs {
    const float* x[2];
    long long xbs[2];      // batch stride of x in floats
    int L[2];
    int tiles[2];
    const f32x4* wkv;      // packed, rows ordered per wave: [w][K heads 2w,2w+1 | V heads 2w,2w+1]
    float* partial;        // [B][tiles0 + tiles1][KV_FLOATS]
    const unsigned char* mask2d;   // [B][L[1]] 1 = real cell, 0 = padding of the 2D stream (linear_attention.py:49-53); NULL: none
};

__device__ __forceinline__ void load_x_tile(float* lds, const float* __restrict__ x, int tok0, int L, int tid) {
    for (int i = tid; i < OPHIP_TOK * (C / 4); i += 256) {
        const int r = i / (C / 4), c4 = i % (C / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tok0 + r < L) v = *reinterpret_cast<const f32x4*>(x + (size_t)(tok0 + r) * C + 4 * c4);
        *reinterpret_cast<f32x4*>(lds + r * LDX + 4 * c4) = v;
    }
}

__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void kv_reduce_kernel(KvReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * OPHIP_TOK;
    load_x_tile(smem, a.x[s] + (size_t)b * a.xbs[s], tok0, L, tid);
    __syncthreads();

    constexpr int KB = C / 8, TSTRIDE = KB * 64;
    f32x16 acc[4];               // [K head 2w, K head 2w+1, V head 2w, V head 2w+1]
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    gemm_lds_x_packed<4>(acc, smem + r * LDX + 4 * h, KB, a.wkv + (size_t)(4 * wave) * TSTRIDE + lane, TSTRIDE);

    const float flen = (float)L;
    const unsigned char* mk = (s == 1 && a.mask2d) ? a.mask2d + (size_t)b * L : nullptr;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int tok = tok0 + acc_row(reg, h);
            const bool valid = tok < L && (!mk || mk[tok]);          // kv_mask: a padded cell's phi(K) row is zero (so is its K^T V term)
            acc[t][reg] = valid ? elu_plus_one(acc[t][reg]) : 0.f;      // phi(K); padded tokens drop out
            acc[2 + t][reg] = acc[2 + t][reg] / flen;                   // values / v_length
        }
    }
    float* out = a.partial + ((size_t)b * (a.tiles[0] + a.tiles[1]) + tile) * KV_FLOATS;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 kv = zero16(), ks = zero16();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][reg], acc[2 + t][reg], kv, 0, 0, 0);
            ks = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][reg], 1.0f, ks, 0, 0, 0);
        }
        float* o = out + (size_t)(2 * wave + t) * KV_PER_HEAD;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 v = {kv[4 * kb], kv[4 * kb + 1], kv[4 * kb + 2], kv[4 * kb + 3]};
            *reinterpret_cast<f32x4*>(o + (kb * 64 + lane) * 4) = v;
        }
        if (r == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) o[1024 + acc_row(reg, h)] = ks[reg];
        }
    }
}

struct KvSumArgs {
    const float* partial;
    float* kv;             // [B][2][KV_FLOATS]
    int tiles[2];
};

constexpr int KVS_G = 16;      // tile groups per output (1024 threads)

__global__ __launch_bounds__(1024) void kv_sum_kernel(KvSumArgs a) {
    __shared__ float red[KVS_G][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int s = blockIdx.y & 1, b = blockIdx.y >> 1;
    const int ttot = a.tiles[0] + a.tiles[1];
    const int t0 = s ? a.tiles[0] : 0, nt = a.tiles[s];
    const float* p = a.partial + ((size_t)b * ttot + t0) * KV_FLOATS + blockIdx.x * 64 + o;
    float acc = 0.f;
    for (int t = g; t < nt; t += KVS_G) acc += p[(size_t)t * KV_FLOATS];
    red[g][o] = acc;
    __syncthreads();
    if (g == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < KVS_G; ++q) tot += red[q][o];       // fixed order
        a.kv[((size_t)b * 2 + s) * KV_FLOATS + blockIdx.x * 64 + o] = tot;
    }
}

struct AttnArgs {
    const float* x[2];
    float* y[2];
    long long xbs[2], ybs[2];
    int L[2];
    int tiles[2];
    const float* kv[2];    // KV block used by stream s (its own for "self", the other's for "cross")
    long long kvbs;        // batch stride of kv (floats)
    float srclen[2];       // length of the source stream feeding stream s
    const f32x4 *wq, *wm, *w0, *w2;
    const float *g1, *b1, *g2, *b2;
    const unsigned char* mask2d;   // q_mask of the 2D stream (see KvReduceArgs)
};

__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 1) void attn_apply_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                         // [32][LDX] layer input tile
    float* P = X + OPHIP_TOK * LDX;          // [32][LDX] phi(Q) -> merge out -> mlp out
    float* Hh = P + OPHIP_TOK * LDX;         // [32][LDH] attention message -> hidden
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int s = tile >= a.tiles[0] ? 1 : 0;
    const int lt = s ? tile - a.tiles[0] : tile;
    const int L = a.L[s], tok0 = lt * OPHIP_TOK;
    load_x_tile(X, a.x[s] + (size_t)b * a.xbs[s], tok0, L, tid);
    __syncthreads();

    constexpr int KB = C / 8, TS = KB * 64;         // K = 256 GEMMs
    constexpr int KB2 = 2 * C / 8, TS2 = KB2 * 64;  // K = 512 GEMMs
    const float* xa = X + r * LDX + 4 * h;
    const float* pa = P + r * LDX + 4 * h;
    const float* ha = Hh + r * LDH + 4 * h;

    // ---- Q projection, phi(Q) ------------------------------------------------------------
    {
        f32x16 q[2] = {zero16(), zero16()};
        gemm_lds_x_packed<2>(q, xa, KB, a.wq + (size_t)(2 * wave) * TS + lane, TS);
        const unsigned char* mk = (s == 1 && a.mask2d) ? a.mask2d + (size_t)b * L : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int tok = tok0 + acc_row(reg, h);
                const bool live = !mk || tok >= L || mk[tok];        // q_mask: phi(Q) = 0 => message 0 (rows >= L are never stored)
                q[t][reg] = live ? elu_plus_one(q[t][reg]) : 0.f;
            }
            acc_to_lds(q[t], P, LDX, 64 * wave + 32 * t, lane);
        }
    }
    __syncthreads();
    // ---- linear attention: msg = (phi(Q) KV) * 1/(phi(Q).Ksum + eps) * S -----------------------
    {
        const float* kvb = a.kv[s] + (size_t)b * a.kvbs;
        const float S = a.srclen[s];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int head = 2 * wave + t;
            const float* kvh = kvb + (size_t)head * KV_PER_HEAD;
            f32x16 num = zero16(), den = zero16();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x4 aq = *reinterpret_cast<const f32x4*>(pa + 32 * head + 8 * kb);
                const f32x4 bk = *reinterpret_cast<const f32x4*>(kvh + (kb * 64 + lane) * 4);
                const f32x4 bs = *reinterpret_cast<const f32x4*>(kvh + 1024 + 8 * kb + 4 * h);
                num = mfma4(aq, bk, num);
                den = mfma4(aq, bs, den);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) num[reg] = num[reg] * (1.0f / (den[reg] + 1e-6f)) * S;
            acc_to_lds(num, Hh, LDH, 32 * head, lane);
        }
    }
    __syncthreads();
    // ---- merge + LayerNorm 1 -------------------------------------------------------------
    {
        f32x16 m[2] = {zero16(), zero16()};
        gemm_lds_x_packed<2>(m, ha, KB, a.wm + (size_t)(2 * wave) * TS + lane, TS);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_to_lds(m[t], P, LDX, 64 * wave + 32 * t, lane);
    }
    __syncthreads();
    rows_layernorm<C, true, false>(P, LDX, a.g1, a.b1, 1e-5f, wave, lane);
    __syncthreads();
    // ---- MLP: relu([x, msg] W0^T) W2^T ---------------------------------------------------
    {
        f32x16 hid[4] = {zero16(), zero16(), zero16(), zero16()};
        const f32x4* w0 = a.w0 + (size_t)(4 * wave) * TS2 + lane;
        gemm_lds_x_packed<4>(hid, xa, KB, w0, TS2);
        gemm_lds_x_packed<4>(hid, pa, KB, w0 + (size_t)KB * 64, TS2);
        // Hh (merge input) is free: every wave passed the barriers after the merge GEMM
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) hid[t][reg] = fmaxf(hid[t][reg], 0.f);
            acc_to_lds(hid[t], Hh, LDH, 128 * wave + 32 * t, lane);
        }
    }
    __syncthreads();
    {
        f32x16 o[2] = {zero16(), zero16()};
        gemm_lds_x_packed<2>(o, ha, KB2, a.w2 + (size_t)(2 * wave) * TS2 + lane, TS2);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_to_lds(o[t], P, LDX, 64 * wave + 32 * t, lane);   // P (msg) no longer read: all waves passed the barrier above
    }
    __syncthreads();
    rows_layernorm<C, true, false>(P, LDX, a.g2, a.b2, 1e-5f, wave, lane);
    // ---- residual + store (each wave: its own 8 rows, whole 1 KiB rows) --------------------
    float* y = a.y[s] + (size_t)b * a.ybs[s];
    for (int rr = 0; rr < 8; ++rr) {
        const int row = 8 * wave + rr;
        if (tok0 + row < L) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(X + row * LDX + 4 * lane);
            const f32x4 mv = *reinterpret_cast<const f32x4*>(P + row * LDX + 4 * lane);
            *reinterpret_cast<f32x4*>(y + (size_t)(tok0 + row) * C + 4 * lane) = xv + mv;
        }
    }
}

}  // namespace

extern "C" size_t ophip_encoder_workspace_floats(int B, int L3d, int L2d) {
    const size_t tiles = (size_t)((L3d + 31) / 32 + (L2d + 31) / 32);
    return (size_t)B * tiles * KV_FLOATS + (size_t)B * 2 * KV_FLOATS;
}

namespace {
int layer_f32(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
              const float* wpack, int is_cross, float* workspace, const unsigned char* mask2d, void* stream_) {
    if (!x3d || !x2d || !y3d || !y2d || !wpack || !workspace) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || L3d < 1 || L2d < 1) return ophip_bad_arg(__func__, "B, L3d, L2d must be >= 1");
    if (x3d == y3d || x2d == y2d) return ophip_bad_arg(__func__, "in-place layer is not supported (cross layers read the pre-update streams)");
    hipStream_t stream = (hipStream_t)stream_;
    const int t3 = (L3d + 31) / 32, t2 = (L2d + 31) / 32;
    float* partial = workspace;
    float* kv = workspace + (size_t)B * (t3 + t2) * KV_FLOATS;
    // packed layer block (floats): Wq | Wkv | Wm | W0 | W2 | g1 b1 g2 b2   (host: onepose_st_amd/packing.py)
    const float* wq = wpack;
    const float* wkv = wq + C * C;
    const float* wm = wkv + 2 * C * C;
    const float* w0 = wm + C * C;
    const float* w2 = w0 + 4 * C * C;
    const float* ln = w2 + 2 * C * C;

    KvReduceArgs ka;
    ka.x[0] = x3d; ka.x[1] = x2d;
    ka.xbs[0] = (long long)L3d * C; ka.xbs[1] = (long long)L2d * C;
    ka.L[0] = L3d; ka.L[1] = L2d;
    ka.tiles[0] = t3; ka.tiles[1] = t2;
    ka.wkv = reinterpret_cast<const f32x4*>(wkv);
    ka.partial = partial;
    ka.mask2d = mask2d;
    const size_t lds_kv = (size_t)OPHIP_TOK * LDX * sizeof(float);
    OPHIP_LAUNCH("kv_reduce", stream, kv_reduce_kernel, dim3(t3 + t2, B), dim3(256), lds_kv, stream, ka);
    OPHIP_CHECK_LAUNCH();

    KvSumArgs sa;
    sa.partial = partial; sa.kv = kv; sa.tiles[0] = t3; sa.tiles[1] = t2;
    OPHIP_LAUNCH("kv_sum", stream, kv_sum_kernel, dim3(KV_FLOATS / 64, 2 * B), dim3(1024), 0, stream, sa);
    OPHIP_CHECK_LAUNCH();

    AttnArgs aa;
    aa.x[0] = x3d; aa.x[1] = x2d; aa.y[0] = y3d; aa.y[1] = y2d;
    aa.xbs[0] = aa.ybs[0] = (long long)L3d * C; aa.xbs[1] = aa.ybs[1] = (long long)L2d * C;
    aa.L[0] = L3d; aa.L[1] = L2d; aa.tiles[0] = t3; aa.tiles[1] = t2;
    // stream 0 = 3D points, stream 1 = 2D grid; "self": own KV, "cross": the other stream's (transformer.py:148-159)
    aa.kv[0] = kv + (is_cross ? KV_FLOATS : 0);
    aa.kv[1] = kv + (is_cross ? 0 : KV_FLOATS);
    aa.kvbs = 2LL * KV_FLOATS;
    aa.srclen[0] = (float)(is_cross ? L2d : L3d);
    aa.srclen[1] = (float)(is_cross ? L3d : L2d);
    aa.wq = reinterpret_cast<const f32x4*>(wq); aa.wm = reinterpret_cast<const f32x4*>(wm);
    aa.w0 = reinterpret_cast<const f32x4*>(w0); aa.w2 = reinterpret_cast<const f32x4*>(w2);
    aa.g1 = ln; aa.b1 = ln + C; aa.g2 = ln + 2 * C; aa.b2 = ln + 3 * C;
    aa.mask2d = mask2d;
    const size_t lds_attn = (size_t)OPHIP_TOK * (2 * LDX + LDH) * sizeof(float);
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(attn_apply_kernel), lds_attn, "hipFuncSetAttribute(attn_apply)")) return rc;
    OPHIP_LAUNCH("attn_apply", stream, attn_apply_kernel, dim3(t3 + t2, B), dim3(256), lds_attn, stream, aa);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
}  // namespace

extern "C" int ophip_encoder_layer(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                   const float* wpack, int is_cross, float* workspace, void* stream) {
    return layer_f32(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, is_cross, workspace, nullptr, stream);
}

// The same layer with the reference's query_mask (transformer.py:148-159, linear_attention.py:49-53): mask2d [B][L2d], 1 = real cell.
extern "C" int ophip_encoder_layer_masked(const float* x3d, const float* x2d, float* y3d, float* y2d, int B, int L3d, int L2d,
                                          const float* wpack, int is_cross, float* workspace, const unsigned char* mask2d, void* stream) {
    if (!mask2d) return ophip_bad_arg(__func__, "null mask (use ophip_encoder_layer)");
    return layer_f32(x3d, x2d, y3d, y2d, B, L3d, L2d, wpack, is_cross, workspace, mask2d, stream);
}
