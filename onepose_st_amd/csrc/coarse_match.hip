// Coarse matching: N x M dual-softmax confidence matrix + mutual-nearest filter + ordered compaction.
// Reference: utils/coarse_matching.py:76-123 (forward: sim = <A/sqrt(C), Bq/sqrt(C)> / (T + 1e-4),
// conf = softmax(sim, dim=1) * softmax(sim, dim=2)) and :125-242 (get_coarse_match, inference branch).
//
// Launch chain (all f32, deterministic -- no float atomics, every cross-workgroup reduction is a
// fixed-order combine of per-tile partials):
//   sim_stats     128x128 tiles of S = A Bq^T / (256 T') on v_mfma_f32_32x32x2_f32, LDS-staged K chunks;
//                 writes S into the conf buffer and per-tile (max, sum exp) for rows and columns
//   stat_combine  online-softmax merge of the partials -> row (max, sum), column (max, sum)
//   conf          in-place S -> conf, coalesced 16 B/lane streaming (HBM-bound); per row the best
//                 (value, lowest j, tie count) as partials; column maxima by integer atomicMax on the
//                 float bits (max is order independent => still deterministic)
//   select        threshold (strict >), border removal (top/left only: the reference's `-b:0` slices are
//                 empty), mutual test, first-true-j semantics on exact ties, compaction in ascending (b, i)
#include "tile_bf16.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int C = 256;
constexpr int TM = 128, TN = 128, KC = 32, LDT = KC + OPHIP_PAD;

struct SimArgs {
    const float* a;      // [B][N][C]
    const float* bq;     // [B][M][C]
    float* conf;         // [B][N][M]
    float* rowpart;      // [B][ntc][N][2]
    float* colpart;      // [B][ntr][M][2]
    int N, M, ntr, ntc;
    float temp;          // temperature + 1e-4
    unsigned long long* stamps;
};

// exact-f32 mode keeps libm expf; the bf16 modes (error budget ~1e-5) use v_exp_f32
template <bool FAST>
__device__ __forceinline__ float exp_sel(float x) { return FAST ? __expf(x) : expf(x); }

__device__ __forceinline__ void merge_ms(float& m, float& e, float m2, float e2) {
    const float mm = fmaxf(m, m2);
    if (mm == -INFINITY) { m = mm; e = 0.f; return; }
    e = e * expf(m - mm) + e2 * expf(m2 - mm);
    m = mm;
}

__device__ __forceinline__ float half_max(float v) {     // over the 32 lanes that share lane>>5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Shared epilogue of the similarity kernels.  The S tile (128 x 128 f32) is staged through LDS (row pitch 130 floats:
// conflict-free for the two-lanes-per-row / per-column sweeps below), then
//   * stored with whole-row 16-byte accesses (the accumulator layout would need 64 scalar stores per lane),
//   * reduced to per-row and per-column (max, sum exp): two lanes per row (resp. column), each sweeping every other
//     element sequentially -- no cross-lane shuffles except the final pair merge.
// In-kernel stamps had the previous register-level epilogue (640 shuffles per wave) at 63 % of the kernel.
constexpr int SLD = 130;
constexpr size_t SIM_STAGE_BYTES = (size_t)TM * SLD * sizeof(float);        // 66 560

template <bool FAST>
__device__ __forceinline__ void sim_epilogue(f32x16 (&acc)[2][2], const SimArgs& p, float* St, int tid, int i0, int j0, int b) {
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    __syncthreads();                                  // every wave is done reading the operand tiles that St overlays
    const float inv_temp = 1.0f / p.temp;             // fast modes: one reciprocal instead of 64 divisions per lane
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                St[(64 * wr + 32 * x + acc_row(reg, h)) * SLD + 64 * wc + 32 * y + r] = FAST ? acc[x][y][reg] * inv_temp : acc[x][y][reg] / p.temp;
    __syncthreads();
    // ---- S -> conf buffer, whole rows ------------------------------------------------------------
    float* conf = p.conf + (size_t)b * p.N * p.M;
    const bool vec = (p.M & 3) == 0;
#pragma unroll 4
    for (int i = tid; i < TM * (TN / 4); i += 256) {
        const int row = i / (TN / 4), c4 = i % (TN / 4);
        const int gi = i0 + row, gj = j0 + 4 * c4;
        if (gi >= p.N || gj >= p.M) continue;
        const float* src = St + row * SLD + 4 * c4;
        if (vec) {
            f32x4 v = {src[0], src[1], src[2], src[3]};
            *reinterpret_cast<f32x4*>(conf + (size_t)gi * p.M + gj) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gj + e < p.M) conf[(size_t)gi * p.M + gj + e] = src[e];
        }
    }
    // ---- row / column (max, sum exp).  Lanes (2k, 2k + 1) share row (resp. column) k and take its even / odd
    // elements: 64 independent LDS reads into registers (latency overlapped), then max and sum exp from registers ----
    const int idx = tid >> 1, par = tid & 1;
    {
        const int ncol = min(TN, p.M - j0);
        float v[TN / 2];
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) v[q] = (2 * q + par < ncol) ? St[idx * SLD + 2 * q + par] : -INFINITY;
        float m = -INFINITY, e = 0.f;
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) m = fmaxf(m, v[q]);
        if (m != -INFINITY) {
#pragma unroll
            for (int q = 0; q < TN / 2; ++q) e += exp_sel<FAST>(v[q] - m);          // exp(-inf) = 0 for the masked tail
        }
        const float m2 = __shfl_xor(m, 1, 64), e2 = __shfl_xor(e, 1, 64);
        float ma = par ? m2 : m, ea = par ? e2 : e, mb = par ? m : m2, eb = par ? e : e2;      // even lane's part first
        merge_ms(ma, ea, mb, eb);
        if (par == 0 && i0 + idx < p.N) {
            float* o = p.rowpart + (((size_t)b * p.ntc + (j0 / TN)) * p.N + i0 + idx) * 2;
            o[0] = ma; o[1] = ea;
        }
    }
    {
        const int nrow = min(TM, p.N - i0);
        float v[TM / 2];
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) v[q] = (2 * q + par < nrow) ? St[(2 * q + par) * SLD + idx] : -INFINITY;
        float m = -INFINITY, e = 0.f;
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) m = fmaxf(m, v[q]);
        if (m != -INFINITY) {
#pragma unroll
            for (int q = 0; q < TM / 2; ++q) e += exp_sel<FAST>(v[q] - m);
        }
        const float m2 = __shfl_xor(m, 1, 64), e2 = __shfl_xor(e, 1, 64);
        float ma = par ? m2 : m, ea = par ? e2 : e, mb = par ? m : m2, eb = par ? e : e2;
        merge_ms(ma, ea, mb, eb);
        if (par == 0 && j0 + idx < p.M) {
            float* o = p.colpart + (((size_t)b * p.ntr + (i0 / TM)) * p.M + j0 + idx) * 2;
            o[0] = ma; o[1] = ea;
        }
    }
}

__global__ __launch_bounds__(256) void sim_stats_kernel(SimArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* At = reinterpret_cast<float*>(smem_f);            // [TM][LDT]
    float* Bt = At + TM * LDT;                               // [TN][LDT]; the S staging image overlays both afterwards
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int j0 = blockIdx.x * TN, i0 = blockIdx.y * TM, b = blockIdx.z;
    const float* A = p.a + (size_t)b * p.N * C;
    const float* Bq = p.bq + (size_t)b * p.M * C;

    f32x4 ra[4], rb[4];
    auto prefetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c4 = idx & 7;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[u] = (i0 + row < p.N) ? *reinterpret_cast<const f32x4*>(A + (size_t)(i0 + row) * C + kc * KC + 4 * c4) : z;
            rb[u] = (j0 + row < p.M) ? *reinterpret_cast<const f32x4*>(Bq + (size_t)(j0 + row) * C + kc * KC + 4 * c4) : z;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c4 = idx & 7;
            // feat / sqrt(C): 1/16 is a power of two, so (a/16).(b/16) == (a.b)/256 bit for bit
            *reinterpret_cast<f32x4*>(At + row * LDT + 4 * c4) = ra[u] * 0.0625f;
            *reinterpret_cast<f32x4*>(Bt + row * LDT + 4 * c4) = rb[u] * 0.0625f;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();

    prefetch(0);
    stage();
    __syncthreads();
    constexpr int NKC = C / KC;
    for (int kc = 0; kc < NKC; ++kc) {
        if (kc + 1 < NKC) prefetch(kc + 1);
#pragma unroll
        for (int kb = 0; kb < KC / 8; ++kb) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const f32x4*>(At + (64 * wr + 32 * t + r) * LDT + 8 * kb + 4 * h);
                fb[t] = *reinterpret_cast<const f32x4*>(Bt + (64 * wc + 32 * t + r) * LDT + 8 * kb + 4 * h);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mfma4(fa[x], fb[y], acc[x][y]);
        }
        __syncthreads();
        if (kc + 1 < NKC) {
            stage();
            __syncthreads();
        }
    }

    sim_epilogue<false>(acc, p, reinterpret_cast<float*>(smem_f), tid, i0, j0, b);
}

// Same tile on the bf16 matrix pipe.  K chunks of 64 features are staged as (hi, lo) bf16 planes with a 144-byte row
// pitch (128 + 16: conflict-free ds_read_b128 for rows distinct mod 16); 72 KiB of LDS -> two workgroups per CU.
constexpr int SKC = 64, SPITCH = SKC * 2 + 16;

template <int NS>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void sim_stats_bf16_kernel(SimArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PL = NS == 3 ? 2 : 1;
    constexpr int PB = TM * SPITCH;                    // bytes per plane
    char* AH = smem;
    char* AL = smem + (PL - 1) * PB;
    char* BH = smem + PL * PB;
    char* BL = BH + (PL - 1) * PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int j0 = blockIdx.x * TN, i0 = blockIdx.y * TM, b = blockIdx.z;
    const float* A = p.a + (size_t)b * p.N * C;
    const float* Bq = p.bq + (size_t)b * p.M * C;

    f32x4 ra[4][2], rb[4][2];
    auto prefetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c8 = idx & 7;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const float* sa = A + (size_t)(i0 + row) * C + kc * SKC + 8 * c8;
            const float* sb = Bq + (size_t)(j0 + row) * C + kc * SKC + 8 * c8;
            const bool va = i0 + row < p.N, vb = j0 + row < p.M;
            ra[u][0] = va ? *reinterpret_cast<const f32x4*>(sa) : z;
            ra[u][1] = va ? *reinterpret_cast<const f32x4*>(sa + 4) : z;
            rb[u][0] = vb ? *reinterpret_cast<const f32x4*>(sb) : z;
            rb[u][1] = vb ? *reinterpret_cast<const f32x4*>(sb + 4) : z;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c8 = idx & 7;
            bf16x8 ah, al, bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 hh, ll;
                split_bf16(ra[u][j >> 2][j & 3] * 0.0625f, hh, ll);     // feat / sqrt(C), exact power of two
                ah[j] = hh; al[j] = ll;
                split_bf16(rb[u][j >> 2][j & 3] * 0.0625f, hh, ll);
                bh[j] = hh; bl[j] = ll;
            }
            const int off = row * SPITCH + 16 * c8;
            *reinterpret_cast<bf16x8*>(AH + off) = ah;
            *reinterpret_cast<bf16x8*>(BH + off) = bh;
            if (NS == 3) {
                *reinterpret_cast<bf16x8*>(AL + off) = al;
                *reinterpret_cast<bf16x8*>(BL + off) = bl;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    OPHIP_STAMP(p.stamps, wg, 0);
    prefetch(0);
    stage();
    __syncthreads();
    OPHIP_STAMP(p.stamps, wg, 1);
    constexpr int NKC = C / SKC;
    for (int kc = 0; kc < NKC; ++kc) {
        if (kc + 1 < NKC) prefetch(kc + 1);
#pragma unroll
        for (int kb = 0; kb < SKC / 16; ++kb) {
            bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oa = (64 * wr + 32 * t + r) * SPITCH + 16 * (2 * kb + h);
                const int ob = (64 * wc + 32 * t + r) * SPITCH + 16 * (2 * kb + h);
                fah[t] = *reinterpret_cast<const bf16x8*>(AH + oa);
                fbh[t] = *reinterpret_cast<const bf16x8*>(BH + ob);
                fal[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(AL + oa) : zero_bf8();
                fbl[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(BL + ob) : zero_bf8();
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mma_bf16<NS>(fah[x], fal[x], fbh[y], fbl[y], acc[x][y]);
        }
        OPHIP_STAMP(p.stamps, wg, 2 + 3 * kc);
        __syncthreads();
        OPHIP_STAMP(p.stamps, wg, 3 + 3 * kc);
        if (kc + 1 < NKC) {
            stage();
            __syncthreads();
        }
        OPHIP_STAMP(p.stamps, wg, 4 + 3 * kc);
    }
    sim_epilogue<true>(acc, p, reinterpret_cast<float*>(smem), tid, i0, j0, b);
    OPHIP_STAMP(p.stamps, wg, 31);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-pass form of the bf16 modes: S is never stored.  split_planes turns the two encoder outputs into (hi, lo) bf16 planes
// once (scaled by 1/sqrt(C) = 1/16, exact); pass 1 computes the S tiles and keeps only their row / column (max, sum exp)
// partials; after stat_combine, pass 2 recomputes each tile on the matrix pipe, turns it into confidences in registers,
// stores conf_matrix ONCE (whole rows through the LDS image) and leaves per-tile row-best / column-max partials that
// best_combine merges in a fixed order (no atomics).  HBM traffic of the stage: the N x M f32 write plus the 2 x 12 MB of
// planes, against write S + read S + write conf before.  Workgroups are dealt to the XCDs in row bands: the blocks that share
// an XCD (blockIdx % 8) sweep all column tiles of a contiguous range of row tiles, so an XCD's L2 holds its A rows and
// streams B once.
// ---------------------------------------------------------------------------------------------------------------------------
struct SplitArgs {
    const float* x;      // [rows][C]
    char* hi;            // [rows][C] bf16
    char* lo;
    long long rows;
};

__global__ __launch_bounds__(256) void split_planes_kernel(SplitArgs p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one 8-feature chunk per thread
    if (i >= p.rows * (C / 8)) return;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.x + i * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.x + i * 8 + 4);
    bf16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 hh, ll;
        split_bf16(v0[j] * 0.0625f, hh, ll); vh[j] = hh; vl[j] = ll;          // feat / sqrt(C): exact power of two
        split_bf16(v1[j] * 0.0625f, hh, ll); vh[4 + j] = hh; vl[4 + j] = ll;
    }
    *reinterpret_cast<bf16x8*>(p.hi + i * 16) = vh;
    *reinterpret_cast<bf16x8*>(p.lo + i * 16) = vl;
}

struct Sim2Args {
    const char *a_hi, *a_lo;     // [B][N][C] bf16 planes
    const char *b_hi, *b_lo;     // [B][M][C]
    float* conf;                 // [B][N][M]                         (pass 2)
    float* rowpart;              // [B][ntc][N][2] (max, sum exp)      (pass 1)
    float* colpart;              // [B][ntr][M][2]
    const float* rowstat;        // [B][N][2] merged                   (pass 2)
    const float* colstat;        // [B][M][2]
    float* rowbest;              // [B][ntc][N][3] (value, j bits, tie count bits)   (pass 2)
    float* colmaxp;              // [B][ntr][M]                                      (pass 2)
    int N, M, ntr, ntc, per_xcd; // per_xcd: blocks per XCD label (grid.x = 8 * per_xcd)
    float temp;
};

// XCD-aware tile of this block: label x = blockIdx.x % 8 owns row tiles [r0, r1) (sizes differ by at most one) and walks
// them column-major (consecutive blocks of an XCD share the B tile, the whole range shares the A rows); false: no tile
__device__ __forceinline__ bool xcd_tile(const Sim2Args& p, int& ti, int& tj) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = p.ntr / 8, rem = p.ntr % 8;
    const int r0 = x * q + (x < rem ? x : rem), nr = q + (x < rem ? 1 : 0);
    if (nr == 0 || k >= nr * p.ntc) return false;
    tj = k / nr;
    ti = r0 + k % nr;
    return true;
}

// S tile (128 x 128, f32 accumulators) from the planes: K chunks of 64 staged through padded LDS rows (144 B: conflict-free
// ds_read_b128), next chunk prefetched into registers under the MFMAs.  The accumulators hold a.b / 256 (planes are pre-scaled).
template <int NS>
__device__ __forceinline__ void sim_tile_from_planes(f32x16 (&acc)[2][2], const Sim2Args& p, char* smem, int i0, int j0, int b, int tid) {
    constexpr int PL = NS == 3 ? 2 : 1;
    constexpr int PB = TM * SPITCH;
    char* AH = smem;
    char* AL = smem + (PL - 1) * PB;
    char* BH = smem + PL * PB;
    char* BL = BH + (PL - 1) * PB;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const size_t arow = (size_t)b * p.N, brow = (size_t)b * p.M;
    bf16x8 ra[4][PL], rb[4][PL];
    auto prefetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c8 = idx & 7;
            const bool va = i0 + row < p.N, vb = j0 + row < p.M;
            const size_t oa = ((arow + i0 + row) * C + kc * SKC + 8 * c8) * 2, ob = ((brow + j0 + row) * C + kc * SKC + 8 * c8) * 2;
            ra[u][0] = va ? *reinterpret_cast<const bf16x8*>(p.a_hi + oa) : zero_bf8();
            rb[u][0] = vb ? *reinterpret_cast<const bf16x8*>(p.b_hi + ob) : zero_bf8();
            if (NS == 3) {
                ra[u][PL - 1] = va ? *reinterpret_cast<const bf16x8*>(p.a_lo + oa) : zero_bf8();
                rb[u][PL - 1] = vb ? *reinterpret_cast<const bf16x8*>(p.b_lo + ob) : zero_bf8();
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u, row = idx >> 3, c8 = idx & 7;
            const int off = row * SPITCH + 16 * c8;
            *reinterpret_cast<bf16x8*>(AH + off) = ra[u][0];
            *reinterpret_cast<bf16x8*>(BH + off) = rb[u][0];
            if (NS == 3) {
                *reinterpret_cast<bf16x8*>(AL + off) = ra[u][PL - 1];
                *reinterpret_cast<bf16x8*>(BL + off) = rb[u][PL - 1];
            }
        }
    };
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
    prefetch(0);
    stage();
    __syncthreads();
    constexpr int NKC = C / SKC;
    for (int kc = 0; kc < NKC; ++kc) {
        if (kc + 1 < NKC) prefetch(kc + 1);
#pragma unroll
        for (int kb = 0; kb < SKC / 16; ++kb) {
            bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oa = (64 * wr + 32 * t + r) * SPITCH + 16 * (2 * kb + h);
                const int ob = (64 * wc + 32 * t + r) * SPITCH + 16 * (2 * kb + h);
                fah[t] = *reinterpret_cast<const bf16x8*>(AH + oa);
                fbh[t] = *reinterpret_cast<const bf16x8*>(BH + ob);
                fal[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(AL + oa) : zero_bf8();
                fbl[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(BL + ob) : zero_bf8();
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = mma_bf16<NS>(fah[x], fal[x], fbh[y], fbl[y], acc[x][y]);
        }
        __syncthreads();
        if (kc + 1 < NKC) {
            stage();
            __syncthreads();
        }
    }
}

// pass 1: row / column (max, sum exp) partials of the tile; S is not stored
template <int NS>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void sim_pass1_kernel(Sim2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ti, tj;
    if (!xcd_tile(p, ti, tj)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int i0 = ti * TM, j0 = tj * TN, b = blockIdx.y;
    f32x16 acc[2][2];
    sim_tile_from_planes<NS>(acc, p, smem, i0, j0, b, tid);
    float* St = reinterpret_cast<float*>(smem);
    const float inv_temp = 1.0f / p.temp;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                St[(64 * wr + 32 * x + acc_row(reg, h)) * SLD + 64 * wc + 32 * y + r] = acc[x][y][reg] * inv_temp;
    __syncthreads();
    const int idx = tid >> 1, par = tid & 1;
    {
        const int ncol = min(TN, p.M - j0);
        float v[TN / 2];
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) v[q] = (2 * q + par < ncol) ? St[idx * SLD + 2 * q + par] : -INFINITY;
        float m = -INFINITY, e = 0.f;
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) m = fmaxf(m, v[q]);
        if (m != -INFINITY) {
#pragma unroll
            for (int q = 0; q < TN / 2; ++q) e += __expf(v[q] - m);
        }
        const float m2 = __shfl_xor(m, 1, 64), e2 = __shfl_xor(e, 1, 64);
        float ma = par ? m2 : m, ea = par ? e2 : e, mb = par ? m : m2, eb = par ? e : e2;      // even lane's part first
        merge_ms(ma, ea, mb, eb);
        if (par == 0 && i0 + idx < p.N) {
            float* o = p.rowpart + (((size_t)b * p.ntc + tj) * p.N + i0 + idx) * 2;
            o[0] = ma; o[1] = ea;
        }
    }
    {
        const int nrow = min(TM, p.N - i0);
        float v[TM / 2];
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) v[q] = (2 * q + par < nrow) ? St[(2 * q + par) * SLD + idx] : -INFINITY;
        float m = -INFINITY, e = 0.f;
#pragma unroll
        for (int q = 0; q < TM / 2; ++q) m = fmaxf(m, v[q]);
        if (m != -INFINITY) {
#pragma unroll
            for (int q = 0; q < TM / 2; ++q) e += __expf(v[q] - m);
        }
        const float m2 = __shfl_xor(m, 1, 64), e2 = __shfl_xor(e, 1, 64);
        float ma = par ? m2 : m, ea = par ? e2 : e, mb = par ? m : m2, eb = par ? e : e2;
        merge_ms(ma, ea, mb, eb);
        if (par == 0 && j0 + idx < p.M) {
            float* o = p.colpart + (((size_t)b * p.ntr + ti) * p.M + j0 + idx) * 2;
            o[0] = ma; o[1] = ea;
        }
    }
}

// pass 2: the tile again -> conf = softmax over i x softmax over j from the merged statistics (one exponential per element:
// exp((S - cm_j) + (S - rm_i)) / (csum_j rsum_i)), stored once; row best (value, lowest j, ties) and column maximum of the tile
template <int NS>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void sim_pass2_kernel(Sim2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float rstat[TM][2];       // (max, 1 / sum) of the tile's rows
    int ti, tj;
    if (!xcd_tile(p, ti, tj)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int i0 = ti * TM, j0 = tj * TN, b = blockIdx.y;
    if (tid < TM) {
        const int i = i0 + tid;
        const float* q = p.rowstat + ((size_t)b * p.N + min(i, p.N - 1)) * 2;
        rstat[tid][0] = q[0];
        rstat[tid][1] = 1.0f / q[1];
    }
    // this lane's two columns: (max, 1 / sum)
    float cm[2], cinv[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int j = min(j0 + 64 * wc + 32 * y + r, p.M - 1);
        const float* q = p.colstat + ((size_t)b * p.M + j) * 2;
        cm[y] = q[0];
        cinv[y] = 1.0f / q[1];
    }
    f32x16 acc[2][2];
    sim_tile_from_planes<NS>(acc, p, smem, i0, j0, b, tid);           // (its barriers also publish rstat)
    float* St = reinterpret_cast<float*>(smem);
    const float inv_temp = 1.0f / p.temp;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 64 * wr + 32 * x + acc_row(reg, h);
            const float rm = rstat[row][0], rinv = rstat[row][1];
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const float sv = acc[x][y][reg] * inv_temp;
                St[row * SLD + 64 * wc + 32 * y + r] = __expf((sv - cm[y]) + (sv - rm)) * (cinv[y] * rinv);
            }
        }
    __syncthreads();
    // ---- conf_matrix rows, coalesced ---------------------------------------------------------------------------
    float* conf = p.conf + (size_t)b * p.N * p.M;
    const bool vec = (p.M & 3) == 0;
#pragma unroll 4
    for (int i = tid; i < TM * (TN / 4); i += 256) {
        const int row = i / (TN / 4), c4 = i % (TN / 4);
        const int gi = i0 + row, gj = j0 + 4 * c4;
        if (gi >= p.N || gj >= p.M) continue;
        const float* src = St + row * SLD + 4 * c4;
        if (vec) {
            f32x4 v = {src[0], src[1], src[2], src[3]};
            *reinterpret_cast<f32x4*>(conf + (size_t)gi * p.M + gj) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gj + e < p.M) conf[(size_t)gi * p.M + gj + e] = src[e];
        }
    }
    // ---- row best / column maximum of the tile: lanes (2k, 2k + 1) share row (column) k, even / odd elements ----------
    const int idx = tid >> 1, par = tid & 1;
    {
        const int ncol = min(TN, p.M - j0);
        float bv = -1.f;
        int bj = 0x7fffffff, bc = 0;
#pragma unroll 16
        for (int q = 0; q < TN / 2; ++q) {
            const int jj = 2 * q + par;
            if (jj < ncol) {
                const float c = St[idx * SLD + jj];
                if (c > bv) { bv = c; bj = j0 + jj; bc = 1; }
                else if (c == bv) { bc += 1; }                 // ascending sweep: bj already holds the lowest j
            }
        }
        const float v2 = __shfl_xor(bv, 1, 64);
        const int j2 = __shfl_xor(bj, 1, 64), c2 = __shfl_xor(bc, 1, 64);
        if (v2 > bv) { bv = v2; bj = j2; bc = c2; }
        else if (v2 == bv) { bj = min(bj, j2); bc += c2; }
        if (par == 0 && i0 + idx < p.N) {
            float* o = p.rowbest + (((size_t)b * p.ntc + tj) * p.N + i0 + idx) * 3;
            o[0] = bv; o[1] = __int_as_float(bj); o[2] = __int_as_float(bc);
        }
    }
    {
        const int nrow = min(TM, p.N - i0);
        float m = 0.f;                                       // conf >= 0
#pragma unroll 16
        for (int q = 0; q < TM / 2; ++q) {
            const int ii = 2 * q + par;
            if (ii < nrow) m = fmaxf(m, St[ii * SLD + idx]);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        if (par == 0 && j0 + idx < p.M) p.colmaxp[((size_t)b * p.ntr + ti) * p.M + j0 + idx] = m;
    }
}

struct BestArgs {
    const float* rowbest_part;   // [B][ntc][N][3]
    const float* colmaxp;        // [B][ntr][M]
    float* rowbest;              // [B][N][3]
    float* colmax;               // [B][M]
    int N, M, ntr, ntc;
};

// fixed-order merge of the per-tile partials: one thread per row (over the ntc column tiles) or column (over the ntr row tiles)
__global__ __launch_bounds__(256) void best_combine_kernel(BestArgs p) {
    const int g = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (g < p.N) {
        float v = -1.f;
        int j = 0x7fffffff, c = 0;
        for (int t = 0; t < p.ntc; ++t) {
            const float* q = p.rowbest_part + (((size_t)b * p.ntc + t) * p.N + g) * 3;
            const float v2 = q[0];
            const int j2 = __float_as_int(q[1]), c2 = __float_as_int(q[2]);
            if (v2 > v) { v = v2; j = j2; c = c2; }
            else if (v2 == v) { j = min(j, j2); c += c2; }
        }
        float* o = p.rowbest + ((size_t)b * p.N + g) * 3;
        o[0] = v; o[1] = __int_as_float(j); o[2] = __int_as_float(c);
    } else if (g - p.N < p.M) {
        const int jj = g - p.N;
        float m = 0.f;
        for (int t = 0; t < p.ntr; ++t) m = fmaxf(m, p.colmaxp[((size_t)b * p.ntr + t) * p.M + jj]);
        p.colmax[(size_t)b * p.M + jj] = m;
    }
}

struct CombineArgs {
    const float *rowpart, *colpart;
    float *rowstat, *colstat;     // [B][N][2], [B][M][2]
    unsigned* colmax_bits;        // [B][M], cleared here for the conf pass's atomicMax
    int N, M, ntr, ntc;
};

// 8 lanes per row / column: lane q merges partials q, q+8, ... in order, then the 8 are merged by an xor
// butterfly whose operand order is fixed (lower lane first), so the result is deterministic.
__global__ __launch_bounds__(256) void stat_combine_kernel(CombineArgs p) {
    const int gid = blockIdx.x * 32 + (threadIdx.x >> 3), q = threadIdx.x & 7, b = blockIdx.y;
    const bool is_row = gid < p.N;
    const int idx = is_row ? gid : gid - p.N;
    const bool live = gid < p.N + p.M;
    const int np = is_row ? p.ntc : p.ntr, len = is_row ? p.N : p.M;
    const float* part = is_row ? p.rowpart : p.colpart;
    float m = -INFINITY, e = 0.f;
    if (live)
        for (int t = q; t < np; t += 8) {
            const float* v = part + (((size_t)b * np + t) * len + idx) * 2;
            merge_ms(m, e, v[0], v[1]);
        }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), e2 = __shfl_xor(e, o, 64);
        float ma = m, ea = e, mb = m2, eb = e2;
        if (q & o) { ma = m2; ea = e2; mb = m; eb = e; }     // lower lane's value first on both sides
        merge_ms(ma, ea, mb, eb);
        m = ma; e = ea;
    }
    if (live && q == 0) {
        float* o = (is_row ? p.rowstat : p.colstat) + ((size_t)b * len + idx) * 2;
        o[0] = m; o[1] = e;
        if (!is_row) p.colmax_bits[(size_t)b * p.M + idx] = 0u;
    }
}

struct ConfArgs {
    float* conf;
    const float *rowstat, *colstat;
    float* rowbest;          // [B][nspan][N][3]  (value, j as float bits, tie count as float bits)
    unsigned* colmax_bits;   // [B][M] column maxima as float bits (conf >= 0, so unsigned order == float order)
    int N, M, nspan, spanw, nrb;
};

#ifndef OPHIP_CONF_ROWS
#define OPHIP_CONF_ROWS 32
#endif
constexpr int CONF_ROWS = OPHIP_CONF_ROWS;      // rows per workgroup (measured at c2: 16 -> 106 us, 32 -> 90, 48 -> 114, 64 -> 132: fewer, less contended column atomics vs grid fill)
constexpr int CONF_RB = 2;         // rows per pipeline stage (two stages in flight)
constexpr int CONF_U = 3;          // float4 groups per thread per row  => span <= 3072 columns (164 VGPRs, 3 waves per SIMD; 4 -> 212 VGPRs, 2 -> more spans: 87 / 90 / 94 us at c2)

template <bool VEC, bool FAST>
__global__ __launch_bounds__(256) void conf_kernel(ConfArgs p) {
    __shared__ float red_v[4][CONF_ROWS];
    __shared__ int red_j[4][CONF_ROWS];
    __shared__ int red_c[4][CONF_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int span = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    const int jb = span * p.spanw, je = min(p.M, jb + p.spanw);
    const int i0 = rb * CONF_ROWS;
    float* conf = p.conf + (size_t)b * p.N * p.M;
    const float* cst = p.colstat + (size_t)b * p.M * 2;
    const float* rst = p.rowstat + (size_t)b * p.N * 2;

    // the rows' (max, 1 / sum) once, through LDS: a global load inside the row loop is an L2 round trip per row pair
    __shared__ float rstat_s[CONF_ROWS][2];
    if (tid < CONF_ROWS) {
        const int i = min(i0 + tid, p.N - 1);
        rstat_s[tid][0] = rst[2 * i];
        rstat_s[tid][1] = 1.0f / rst[2 * i + 1];
    }
    float cm[CONF_U][4], cinv[CONF_U][4], cbest[CONF_U][4];
#pragma unroll
    for (int u = 0; u < CONF_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + 4 * tid + 1024 * u + e;
            cm[u][e] = (j < je) ? cst[2 * j] : 0.f;
            cinv[u][e] = (j < je) ? 1.0f / cst[2 * j + 1] : 1.f;
            cbest[u][e] = 0.f;
        }
    const int nrows = min(CONF_ROWS, p.N - i0);
    // software pipeline over batches of CONF_RB rows: the loads of batch k + 1 are in flight while batch k is processed
    float s[2][CONF_RB][CONF_U][4];
    auto load_batch = [&](int r0, float (&dst)[CONF_RB][CONF_U][4]) {
#pragma unroll
        for (int q = 0; q < CONF_RB; ++q) {
            const int rr = min(r0 + q, nrows - 1);
            const float* row = conf + (size_t)(i0 + rr) * p.M;
#pragma unroll
            for (int u = 0; u < CONF_U; ++u) {
                const int jq = jb + 4 * tid + 1024 * u;
                if (VEC) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (jq < je) v = *reinterpret_cast<const f32x4*>(row + jq);
                    dst[q][u][0] = v[0]; dst[q][u][1] = v[1]; dst[q][u][2] = v[2]; dst[q][u][3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[q][u][e] = (jq + e < je) ? row[jq + e] : 0.f;
                }
            }
        }
    };
    auto process_batch = [&](int r0, float (&cur)[CONF_RB][CONF_U][4]) {
#pragma unroll
        for (int q = 0; q < CONF_RB; ++q) {
            const int rr = r0 + q;
            if (rr < nrows) {
                const int i = i0 + rr;
                const float rm = rstat_s[rr][0], rinv = rstat_s[rr][1];
                float* row = conf + (size_t)i * p.M;
                float bv = -1.f;
                int bj = 0x7fffffff, bc = 0;
#pragma unroll
                for (int u = 0; u < CONF_U; ++u) {
                    const int jq = jb + 4 * tid + 1024 * u;
                    if (jq < je) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // softmax over the 3D axis (dim=1: column stats) times softmax over the 2D axis (dim=2: row stats)
                            const float c = (exp_sel<FAST>(cur[q][u][e] - cm[u][e]) * cinv[u][e]) * (exp_sel<FAST>(cur[q][u][e] - rm) * rinv);
                            cur[q][u][e] = c;
                            if (jq + e < je) {
                                cbest[u][e] = fmaxf(cbest[u][e], c);
                                if (c > bv) { bv = c; bj = jq + e; bc = 1; }
                                else if (c == bv) { bc += 1; bj = min(bj, jq + e); }
                            }
                        }
                        if (VEC) {
                            f32x4 v = {cur[q][u][0], cur[q][u][1], cur[q][u][2], cur[q][u][3]};
                            *reinterpret_cast<f32x4*>(row + jq) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (jq + e < je) row[jq + e] = cur[q][u][e];
                        }
                    }
                }
                // wave reduce: max value, lowest j among the maxima, number of maxima
                const float wv = wave_max(bv);
                int cj = (bv == wv) ? bj : 0x7fffffff;
                int cc = (bv == wv) ? bc : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    cj = min(cj, __shfl_xor(cj, o, 64));
                    cc += __shfl_xor(cc, o, 64);
                }
                if (lane == 0) { red_v[wave][rr] = wv; red_j[wave][rr] = cj; red_c[wave][rr] = cc; }
            }
        }
    };
    load_batch(0, s[0]);
    __syncthreads();                                  // rstat_s
    for (int r0 = 0; r0 < nrows; r0 += 2 * CONF_RB) {
        if (r0 + CONF_RB < nrows) load_batch(r0 + CONF_RB, s[1]);
        process_batch(r0, s[0]);
        if (r0 + 2 * CONF_RB < nrows) load_batch(r0 + 2 * CONF_RB, s[0]);
        if (r0 + CONF_RB < nrows) process_batch(r0 + CONF_RB, s[1]);
    }
    __syncthreads();
    if (tid < nrows) {
        float v = red_v[0][tid];
        int j = red_j[0][tid], c = red_c[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float v2 = red_v[w][tid];
            if (v2 > v) { v = v2; j = red_j[w][tid]; c = red_c[w][tid]; }
            else if (v2 == v) { j = min(j, red_j[w][tid]); c += red_c[w][tid]; }
        }
        float* o = p.rowbest + (((size_t)b * p.nspan + span) * p.N + i0 + tid) * 3;
        o[0] = v; o[1] = __int_as_float(j); o[2] = __int_as_float(c);
    }
    // column maxima: max is order independent, so an integer atomicMax on the float bits is deterministic
    unsigned* cb = p.colmax_bits + (size_t)b * p.M;
#pragma unroll
    for (int u = 0; u < CONF_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + 4 * tid + 1024 * u + e;
            if (j < je && cbest[u][e] > 0.f) atomicMax(cb + j, __float_as_uint(cbest[u][e]));
        }
}

struct SelectArgs {
    const float* conf;
    const float* rowbest;
    const float* colmax;
    const float* kpts;          // [B][N][3]
    long long kpts_bs;          // batch stride of kpts in floats (0 for a shared object block)
    int B, N, M, nspan, wc, border;
    float thr, scale;
    long long* b_ids; long long* i_ids; long long* j_ids;
    float* mconf; float* mk3d; float* mkq;
    long long* m_bids;          // optional second copy of b_ids (the reference's 'm_bids')
    unsigned char* gt_mask;     // optional mconf == 0 flags (the reference's 'gt_mask')
    int* count;
};

// One workgroup compacts the surviving rows in (b, i) order.  A thread owns SEL_IT rows of a chunk of 1024 * SEL_IT rows:
// all its row-best records are loaded first, then all its column maxima (independent gathers, one memory round trip each
// instead of one per 1024 rows), then one scan over the SEL_IT x 16 wave counts places every match.
constexpr int SEL_IT = 8;

__global__ __launch_bounds__(1024) void select_kernel(SelectArgs p) {
    __shared__ int wcount[SEL_IT * 16];
    __shared__ int wpref[SEL_IT * 16 + 1];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int b = 0; b < p.B; ++b) {
        const float* conf = p.conf + (size_t)b * p.N * p.M;
        const float* cmx = p.colmax + (size_t)b * p.M;
        const float* kpb = p.kpts + (size_t)b * p.kpts_bs;
        for (int ib = 0; ib < p.N; ib += 1024 * SEL_IT) {
            float v[SEL_IT], cm[SEL_IT], kx[SEL_IT], ky[SEL_IT], kz[SEL_IT];
            int j[SEL_IT], c[SEL_IT];
            bool ok[SEL_IT];
#pragma unroll
            for (int it = 0; it < SEL_IT; ++it) {
                const int i = ib + it * 1024 + tid;
                v[it] = -1.f; j[it] = 0x7fffffff; c[it] = 0;
                if (i < p.N) {
                    for (int sp = 0; sp < p.nspan; ++sp) {
                        const float* q = p.rowbest + (((size_t)b * p.nspan + sp) * p.N + i) * 3;
                        const float v2 = q[0];
                        const int j2 = __float_as_int(q[1]), c2 = __float_as_int(q[2]);
                        if (v2 > v[it]) { v[it] = v2; j[it] = j2; c[it] = c2; }
                        else if (v2 == v[it]) { j[it] = min(j[it], j2); c[it] += c2; }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < SEL_IT; ++it) {
                const int i = ib + it * 1024 + tid;
                const bool live = i < p.N && v[it] > p.thr;
                cm[it] = live ? cmx[j[it]] : 0.f;
                kx[it] = live ? kpb[(size_t)i * 3] : 0.f;
                ky[it] = live ? kpb[(size_t)i * 3 + 1] : 0.f;
                kz[it] = live ? kpb[(size_t)i * 3 + 2] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < SEL_IT; ++it) {
                const int i = ib + it * 1024 + tid;
                const float vv = v[it];
                auto inside = [&](int jj) { return (jj / p.wc >= p.border) && (jj % p.wc >= p.border); };
                bool o = false;
                if (i < p.N && vv > p.thr) {
                    o = inside(j[it]) && vv == cm[it];
                    if (!o && c[it] > 1) {
                        // exact tie of the row maximum: the reference takes the first j whose mask is true
                        const float* row = conf + (size_t)i * p.M;
                        for (int jj = j[it] + 1; jj < p.M; ++jj)
                            if (row[jj] == vv && inside(jj) && vv == cmx[jj]) { j[it] = jj; o = true; break; }
                    }
                }
                ok[it] = o;
                const unsigned long long mask = __ballot(o);
                if (lane == 0) wcount[it * 16 + wave] = __popcll(mask);
            }
            __syncthreads();
            if (tid < SEL_IT * 16) {
                int acc = 0;
                for (int u = 0; u < tid; ++u) acc += wcount[u];
                wpref[tid] = acc;
                if (tid == SEL_IT * 16 - 1) wpref[SEL_IT * 16] = acc + wcount[tid];
            }
            __syncthreads();
            const int base = base_s;
#pragma unroll
            for (int it = 0; it < SEL_IT; ++it) {
                const unsigned long long mask = __ballot(ok[it]);
                if (ok[it]) {
                    const int i = ib + it * 1024 + tid;
                    const int pos = base + wpref[it * 16 + wave] + __popcll(mask & ((1ull << lane) - 1ull));
                    p.b_ids[pos] = b; p.i_ids[pos] = i; p.j_ids[pos] = j[it];
                    p.mconf[pos] = v[it];
                    p.mk3d[3 * pos] = kx[it]; p.mk3d[3 * pos + 1] = ky[it]; p.mk3d[3 * pos + 2] = kz[it];
                    p.mkq[2 * pos] = (float)(j[it] % p.wc) * p.scale;
                    p.mkq[2 * pos + 1] = (float)(j[it] / p.wc) * p.scale;
                    if (p.m_bids) p.m_bids[pos] = b;
                    if (p.gt_mask) p.gt_mask[pos] = v[it] == 0.f ? 1 : 0;
                }
            }
            __syncthreads();
            if (tid == 0) base_s = base + wpref[SEL_IT * 16];
            __syncthreads();
        }
    }
    if (tid == 0) *p.count = base_s;
}

inline int conf_nspan(int M) { return (M + 3071) / 3072; }
inline int conf_spanw(int M) { const int ns = conf_nspan(M); return (((M + ns - 1) / ns) + 3) / 4 * 4; }

}  // namespace

extern "C" size_t ophip_coarse_workspace_floats(int B, int N, int M) {
    const size_t ntr = (N + TM - 1) / TM, ntc = (M + TN - 1) / TN;
    size_t f = 0;
    f += (size_t)B * ntc * N * 2;          // rowpart
    f += (size_t)B * ntr * M * 2;          // colpart
    f += (size_t)B * N * 2 + (size_t)B * M * 2;     // rowstat, colstat
    f += (size_t)B * conf_nspan(M) * N * 3;         // rowbest (one-pass form: per span)
    f += (size_t)B * M;                    // colmax (float bits)
    // two-pass form of the bf16 modes: (hi, lo) bf16 planes of both inputs, per-tile row-best / column-max partials
    f += (size_t)B * ((size_t)N + M) * C + 64;
    f += (size_t)B * ntc * N * 3 + (size_t)B * ntr * M + (size_t)B * N * 3;
    return f + 64;
}

namespace {
// parts: 1 = similarity / confidence kernels, 2 = select (threshold, mutual test, compaction), 3 = both
int coarse_impl(int parts, const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                int* count, int nsplit, void* stream_) {
    if (!feat3d || !feat2d || !keypoints3d || !conf || !workspace || !b_ids || !i_ids || !j_ids || !mconf || !mkpts3d || !mkpts_c || !count)
        return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || N < 1 || M < 1 || wc < 1 || M % wc != 0) return ophip_bad_arg(__func__, "bad sizes (need M == hc * wc)");
    if (nsplit != 0 && nsplit != 1 && nsplit != 3) return ophip_bad_arg(__func__, "nsplit must be 0 (exact f32), 1 (bf16) or 3 (split bf16)");
    hipStream_t stream = (hipStream_t)stream_;
    const int ntr = (N + TM - 1) / TM, ntc = (M + TN - 1) / TN, nrb = (N + CONF_ROWS - 1) / CONF_ROWS;
    const int nspan = conf_nspan(M), spanw = conf_spanw(M);
    float* rowpart = workspace;
    float* colpart = rowpart + (size_t)B * ntc * N * 2;
    float* rowstat = colpart + (size_t)B * ntr * M * 2;
    float* colstat = rowstat + (size_t)B * N * 2;
    float* rowbest = colstat + (size_t)B * M * 2;
    float* colmax = rowbest + (size_t)B * nspan * N * 3;

    // Measured at c2 (tools/time_coarse.py, round 2): the two-pass form as built moves 1/3 of the bytes but runs its tile GEMM
    // + epilogue twice -- split 16 + pass1 119 + combine 8 + pass2 181 + best 27 + select 19 = 345 us against 205 us for the
    // one-pass chain (sim_stats 105, conf 85): the 128 x 128 tile kernel spends 3/4 of its time outside the MFMAs (staging,
    // barriers, the LDS sweeps of the epilogue), so recomputing S only pays once that kernel is ~2x leaner.  It stays
    // selectable (OPHIP_COARSE_TWOPASS=1) and parity-tested; the default is the one-pass chain.
    static const bool two_pass_env = getenv("OPHIP_COARSE_TWOPASS") != nullptr;
    const bool two_pass = nsplit != 0 && two_pass_env;
    int sel_nspan = nspan;
    if (two_pass) {
        float* w2s = colmax + (size_t)B * M;
        w2s += (16 - ((reinterpret_cast<uintptr_t>(w2s) >> 2) & 15)) & 15;
        rowbest = reinterpret_cast<float*>(reinterpret_cast<char*>(w2s) + (size_t)B * ((size_t)N + M) * C * 4) + (size_t)B * ntc * N * 3 + (size_t)B * ntr * M;
        sel_nspan = 1;
    }
    if (two_pass && (parts & 1)) {
        // ---- bf16 modes: S recomputed instead of stored (see the kernels above) ------------------------------------------
        float* w2 = colmax + (size_t)B * M;
        w2 += (16 - ((reinterpret_cast<uintptr_t>(w2) >> 2) & 15)) & 15;                 // 64-byte aligned planes
        char* a_hi = reinterpret_cast<char*>(w2);
        char* a_lo = a_hi + (size_t)B * N * C * 2;
        char* b_hi = a_lo + (size_t)B * N * C * 2;
        char* b_lo = b_hi + (size_t)B * M * C * 2;
        float* rowbest_part = reinterpret_cast<float*>(b_lo + (size_t)B * M * C * 2);
        float* colmaxp = rowbest_part + (size_t)B * ntc * N * 3;
        float* rowbest1 = colmaxp + (size_t)B * ntr * M;
        SplitArgs s3{feat3d, a_hi, a_lo, (long long)B * N}, s2{feat2d, b_hi, b_lo, (long long)B * M};
        OPHIP_LAUNCH("split_planes", stream, split_planes_kernel, dim3((unsigned)(((long long)B * N * (C / 8) + 255) / 256)), dim3(256), 0, stream, s3);
        OPHIP_LAUNCH("split_planes", stream, split_planes_kernel, dim3((unsigned)(((long long)B * M * (C / 8) + 255) / 256)), dim3(256), 0, stream, s2);
        OPHIP_CHECK_LAUNCH();
        const int per_xcd = ((ntr + 7) / 8) * ntc;
        Sim2Args pa{a_hi, a_lo, b_hi, b_lo, conf, rowpart, colpart, rowstat, colstat, rowbest_part, colmaxp, N, M, ntr, ntc, per_xcd,
                    (float)(temperature + 1e-4)};
        const size_t tiles = (size_t)(nsplit == 3 ? 4 : 2) * TM * SPITCH;
        const size_t lds = tiles > SIM_STAGE_BYTES ? tiles : SIM_STAGE_BYTES;
        const void* f1 = nsplit == 3 ? reinterpret_cast<const void*>(sim_pass1_kernel<3>) : reinterpret_cast<const void*>(sim_pass1_kernel<1>);
        const void* f2 = nsplit == 3 ? reinterpret_cast<const void*>(sim_pass2_kernel<3>) : reinterpret_cast<const void*>(sim_pass2_kernel<1>);
        if (int rc = ophip_lds_attr(f1, lds, "hipFuncSetAttribute(sim_pass1)")) return rc;
        if (int rc = ophip_lds_attr(f2, lds, "hipFuncSetAttribute(sim_pass2)")) return rc;
        if (nsplit == 3) OPHIP_LAUNCH("sim_pass1", stream, sim_pass1_kernel<3>, dim3(8 * per_xcd, B), dim3(256), lds, stream, pa);
        else OPHIP_LAUNCH("sim_pass1", stream, sim_pass1_kernel<1>, dim3(8 * per_xcd, B), dim3(256), lds, stream, pa);
        OPHIP_CHECK_LAUNCH();
        CombineArgs ca{rowpart, colpart, rowstat, colstat, reinterpret_cast<unsigned*>(colmax), N, M, ntr, ntc};
        OPHIP_LAUNCH("stat_combine", stream, stat_combine_kernel, dim3((N + M + 31) / 32, B), dim3(256), 0, stream, ca);
        OPHIP_CHECK_LAUNCH();
        if (nsplit == 3) OPHIP_LAUNCH("sim_pass2", stream, sim_pass2_kernel<3>, dim3(8 * per_xcd, B), dim3(256), lds, stream, pa);
        else OPHIP_LAUNCH("sim_pass2", stream, sim_pass2_kernel<1>, dim3(8 * per_xcd, B), dim3(256), lds, stream, pa);
        OPHIP_CHECK_LAUNCH();
        BestArgs ba{rowbest_part, colmaxp, rowbest1, colmax, N, M, ntr, ntc};
        OPHIP_LAUNCH("best_combine", stream, best_combine_kernel, dim3((N + M + 255) / 256, B), dim3(256), 0, stream, ba);
        OPHIP_CHECK_LAUNCH();
        if (rowbest1 != rowbest) return ophip_bad_arg(__func__, "internal: workspace layout");
    } else if (parts & 1) {
        SimArgs sa{feat3d, feat2d, conf, rowpart, colpart, N, M, ntr, ntc, (float)(temperature + 1e-4), ophip_stamp_buffer()};
        {
            // dynamic LDS = max(operand tiles, S staging image of the epilogue)
            const size_t tiles = nsplit == 0 ? (size_t)(TM + TN) * LDT * sizeof(float) : (size_t)(nsplit == 3 ? 4 : 2) * TM * SPITCH;
            const size_t lds = tiles > SIM_STAGE_BYTES ? tiles : SIM_STAGE_BYTES;
            const void* fn = nsplit == 0 ? reinterpret_cast<const void*>(sim_stats_kernel)
                           : nsplit == 3 ? reinterpret_cast<const void*>(sim_stats_bf16_kernel<3>) : reinterpret_cast<const void*>(sim_stats_bf16_kernel<1>);
            if (int rc = ophip_lds_attr(fn, lds, "hipFuncSetAttribute(sim_stats)")) return rc;
            if (nsplit == 0) OPHIP_LAUNCH("sim_stats", stream, sim_stats_kernel, dim3(ntc, ntr, B), dim3(256), lds, stream, sa);
            else if (nsplit == 3) OPHIP_LAUNCH("sim_stats", stream, sim_stats_bf16_kernel<3>, dim3(ntc, ntr, B), dim3(256), lds, stream, sa);
            else OPHIP_LAUNCH("sim_stats", stream, sim_stats_bf16_kernel<1>, dim3(ntc, ntr, B), dim3(256), lds, stream, sa);
        }
        OPHIP_CHECK_LAUNCH();
        CombineArgs ca{rowpart, colpart, rowstat, colstat, reinterpret_cast<unsigned*>(colmax), N, M, ntr, ntc};
        OPHIP_LAUNCH("stat_combine", stream, stat_combine_kernel, dim3((N + M + 31) / 32, B), dim3(256), 0, stream, ca);
        OPHIP_CHECK_LAUNCH();
        ConfArgs fa{conf, rowstat, colstat, rowbest, reinterpret_cast<unsigned*>(colmax), N, M, nspan, spanw, nrb};
        const bool vec = M % 4 == 0, fast = nsplit != 0;
        if (vec && fast) OPHIP_LAUNCH("conf", stream, (conf_kernel<true, true>), dim3(nspan, nrb, B), dim3(256), 0, stream, fa);
        else if (vec) OPHIP_LAUNCH("conf", stream, (conf_kernel<true, false>), dim3(nspan, nrb, B), dim3(256), 0, stream, fa);
        else if (fast) OPHIP_LAUNCH("conf", stream, (conf_kernel<false, true>), dim3(nspan, nrb, B), dim3(256), 0, stream, fa);
        else OPHIP_LAUNCH("conf", stream, (conf_kernel<false, false>), dim3(nspan, nrb, B), dim3(256), 0, stream, fa);
        OPHIP_CHECK_LAUNCH();
    }
    if (parts & 2) {
        SelectArgs se{conf, rowbest, colmax, keypoints3d, kpts_bstride, B, N, M, sel_nspan, wc, border_rm, thr, scale,
                      b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count};
        OPHIP_LAUNCH("select", stream, select_kernel, dim3(1), dim3(1024), 0, stream, se);
        OPHIP_CHECK_LAUNCH();
    }
    return 0;
}
}  // namespace

extern "C" int ophip_coarse_match(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                  int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                  float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                  float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                  int* count, int nsplit, void* stream) {
    return coarse_impl(3, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream);
}

// The same stage in two calls, so that a pipeline can put the single-workgroup select kernel (and what follows it) on another
// stream than the similarity / confidence kernels: ophip_coarse_match == ophip_coarse_match_conf then ophip_coarse_match_select
// with identical arguments (the second call only reads conf_matrix and the partials the first left in the workspace).
extern "C" int ophip_coarse_match_conf(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                       int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                       float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                       float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                       int* count, int nsplit, void* stream) {
    return coarse_impl(1, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream);
}

extern "C" int ophip_coarse_match_select(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                         int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                         float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                         float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                         int* count, int nsplit, void* stream) {
    return coarse_impl(2, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream);
}
