// Coarse matching: N x M dual-softmax confidence matrix + mutual-nearest filter + ordered compaction.
// Reference: utils/coarse_matching.py:76-123 (forward: sim = <A/sqrt(C), Bq/sqrt(C)> / (T + 1e-4),
// conf = softmax(sim, dim=1) * softmax(sim, dim=2)) and :125-242 (get_coarse_match, inference branch).
//
// Launch chain (deterministic -- no float atomics, every cross-workgroup reduction is a fixed-order combine of per-tile
// partials):
//   frag_planes   (bf16 modes) both inputs once into (hi, lo) bf16 MFMA operand fragments, scaled by 1/sqrt(C)
//   sim_frag      (bf16 modes) 128x128 tiles of S on v_mfma_f32_32x32x16_bf16, operands by LDS-DMA into a four-buffer ring;
//   sim_stats     (exact-f32 mode) the same tiles on v_mfma_f32_32x32x2_f32 with LDS-staged K chunks;
//                 both write S into the conf buffer and per-tile (max, sum exp) partials for rows and columns
//   stat_combine  online-softmax merge of the partials -> row (max, sum), column (max, sum)
//   conf          in-place S -> conf, coalesced 16 B/lane streaming (HBM-bound); per row the best candidate above the
//                 threshold (value, lowest j, tie count) as partials; column maxima of the candidates by integer atomicMax
//                 on the float bits (max is order independent => still deterministic)
//   select        threshold (strict >), border removal (top/left only: the reference's `-b:0` slices are
//                 empty), mutual test, first-true-j semantics on exact ties, compaction in ascending (b, i)
#include "tile_bf16.h"
#include "onepose_hip.h"
#include "x3w8_internal.h"
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
    const unsigned char* colmask;   // [B][M] 1 = real query cell, 0 = padding: -1e9 is added to its column (coarse_matching.py:108-114); NULL: none
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

// Epilogue of the round-1 similarity kernels (exact-f32 mode; OPHIP_SIM_V1).  The S tile (128 x 128 f32) is staged through LDS
// (row pitch 130 floats: conflict-free for the two-lanes-per-row / per-column sweeps below), then
//   * stored with whole-row 16-byte accesses (the accumulator layout would need 64 scalar stores per lane),
//   * reduced to per-row and per-column (max, sum exp): two lanes per row (resp. column), each sweeping every other
//     element sequentially -- no cross-lane shuffles except the final pair merge.
constexpr int SLD = 130;
constexpr size_t SIM_STAGE_BYTES = (size_t)TM * SLD * sizeof(float);        // 66 560

// row / column (max, sum exp) of the staged tile St (pitch LD floats), masked at the matrix edge.  Lanes (2k, 2k + 1) share row
// (resp. column) k and take its even / odd elements: 64 independent LDS reads into registers (latency overlapped), then max and
// sum exp from registers
template <bool FAST, int LD>
__device__ __forceinline__ void tile_stats_lds(const SimArgs& p, const float* St, int tid, int i0, int j0, int b) {
    const int idx = tid >> 1, par = tid & 1;
    {
        const int ncol = min(TN, p.M - j0);
        float v[TN / 2];
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) v[q] = (2 * q + par < ncol) ? St[idx * LD + 2 * q + par] : -INFINITY;
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
        for (int q = 0; q < TM / 2; ++q) v[q] = (2 * q + par < nrow) ? St[(2 * q + par) * LD + idx] : -INFINITY;
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
    if (p.colmask) {                                  // sim_matrix += -1e9 on the padded query cells' columns
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = j0 + 64 * wc + 32 * y + r;
            if (col < p.M && !p.colmask[(size_t)b * p.M + col]) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) St[(64 * wr + 32 * x + acc_row(reg, h)) * SLD + 64 * wc + 32 * y + r] += -1e9f;
            }
        }
    }
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
    tile_stats_lds<FAST, SLD>(p, St, tid, i0, j0, b);
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

// ---------------------------------------------------------------------------------------------------------------------------
// Fragment-plane form of the bf16 modes (default).  frag_planes turns both encoder outputs once into (hi, lo) bf16 planes,
// scaled by 1/sqrt(C) = 1/16 (exact), laid out as the MFMA operand fragments themselves: for 32-row tile T and k-step s
// (16 features) the 1 KiB block ((T * 16 + s) * 2 + plane) holds lane l's 8 bf16 (row 32 T + (l & 31), features
// 16 s + 8 (l >> 5) ..) at byte 16 l; rows are zero-padded to a multiple of 128.  sim_frag then has no conversion work and no
// register staging in its k-loop: one k-step (16 KiB: 4 + 4 row tiles x 2 planes) arrives by LDS-DMA (global_load_lds_dwordx4,
// one fragment per wave-instruction, lane-linear image = conflict-free ds_read_b128) into a ring of four buffers, three
// k-steps ahead of the 12 MFMAs per wave that consume it (counted vmcnt + raw s_barrier, one barrier per k-step).  The old kernel split every A row 38 times and every
// B row 55 times (once per tile) and spent 3/4 of its time outside the MFMAs (stamps, tools/stamps_sim.py).
//
// Epilogue: the scaled tile is staged through LDS and leaves in ONE pass: whole rows to the conf buffer and, on the way, into
// the (max, sum exp) partials with one exponential per element (tile maximum as the common reference; see the pass).  Tiles cut
// by the matrix edge, and tiles where a row or column sits > 59 below the tile maximum, take the exact per-row / per-column sweep
// instead.  Workgroups are dealt to the XCDs in row bands: the blocks that share an XCD (blockIdx % 8) walk a contiguous range
// of row tiles column-major, so an XCD's L2 keeps its A fragments and streams B once.
// ---------------------------------------------------------------------------------------------------------------------------
struct FragArgs {
    const float* x3;     // [B][N][C]
    const float* x2;     // [B][M][C]
    char* a;             // [B][Npad / 32][16][2][1024]
    char* b;             // [B][Mpad / 32][16][2][1024]
    int N, M, nta, ntb;  // nta = Npad / 32, ntb = Mpad / 32
};

// one wave per fragment pair: block (T, s / 4), wave s % 4; lane l converts the 8 features it will hold as an MFMA operand
// (32 bytes in, 16 + 16 out: each wave writes two whole 1 KiB fragments, the block reads 256 contiguous bytes of each row)
__global__ __launch_bounds__(256) void frag_planes_kernel(FragArgs p) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int T = blockIdx.x >> 2;
    const int s = 4 * (blockIdx.x & 3) + wave;
    const bool is_a = T < p.nta;
    if (!is_a) T -= p.nta;
    const int rows = is_a ? p.N : p.M;
    const float* x = (is_a ? p.x3 : p.x2) + (size_t)b * rows * C;
    char* dst = (is_a ? p.a + (size_t)b * p.nta * 32768 : p.b + (size_t)b * p.ntb * 32768) + (size_t)T * 32768 + (size_t)(2 * s) * 1024 + 16 * lane;
    const int row = 32 * T + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v0 = row < rows ? *reinterpret_cast<const f32x4*>(x + (size_t)row * C + k0) : z;
    const f32x4 v1 = row < rows ? *reinterpret_cast<const f32x4*>(x + (size_t)row * C + k0 + 4) : z;
    bf16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 hh, ll;
        split_bf16(v0[j] * 0.0625f, hh, ll); vh[j] = hh; vl[j] = ll;          // feat / sqrt(C): exact power of two
        split_bf16(v1[j] * 0.0625f, hh, ll); vh[4 + j] = hh; vl[4 + j] = ll;
    }
    *reinterpret_cast<bf16x8*>(dst) = vh;
    *reinterpret_cast<bf16x8*>(dst + 1024) = vl;
}

struct SimFragArgs {
    const char* a;               // fragment planes of feat3d / 16 (see frag_planes_kernel)
    const char* b;               // fragment planes of feat2d / 16
    float* conf;                 // [B][N][M]   (MODE 0 only)
    float* rowpart;              // [B][ntc][N][2] (max, sum exp)
    float* colpart;              // [B][ntr][M][2]
    int N, M, ntr, ntc;
    float temp;
    unsigned long long* stamps;
    // MODE 2 (candidates of the lazy form): merged statistics in, per-(row, column tile) best candidates and column maxima out
    const float* rowstat;        // [B][N][2] (max, sum exp)
    const float* colstat;        // [B][M][2]
    const float* rowlog;         // [B][N] logf(sum exp)
    const float* collog;         // [B][M]
    float* rowbest;              // [B][ntc][N][3] (value, j as float bits, tie count as float bits)
    unsigned* colmax_bits;       // [B][M]
    float thr, logthr_lo;        // strict threshold; logf(thr) - 1e-3: conservative prefilter on the exponent (saves the exponential)
    const unsigned char* colmask;   // see SimArgs
};

// XCD-aware tile of this block: label x = blockIdx.x % 8 owns row tiles [r0, r1) (sizes differ by at most one) and walks
// them column-major (consecutive blocks of an XCD share the B tile, the whole range shares the A rows); false: no tile
__device__ __forceinline__ bool xcd_tile(int ntr, int ntc, int& ti, int& tj) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int q = ntr / 8, rem = ntr % 8;
    const int r0 = x * q + (x < rem ? x : rem), nr = q + (x < rem ? 1 : 0);
    if (nr == 0 || k >= nr * ntc) return false;
    // the LAST column band first: its tiles are cut by the matrix edge (M is rarely a multiple of 128) and take the exact statistics pass,
    // 1.3-2x a plain tile (tools/stamps_sim.py) -- walked last they were each XCD's tail (round 5: the slowest decile of workgroups all ended
    // the launch); walked first they sit under the other rounds
    tj = ntc - 1 - k / nr;
    ti = r0 + k % nr;
    return true;
}

constexpr int FLD = 132;                                             // S staging pitch of sim_frag: rows stay 16-byte aligned
constexpr int FRAG_CHUNK_BYTES = 16384;                              // one k-step: 16 fragments, A 8 KiB, B 8 KiB; four buffers
constexpr size_t SIM_FRAG_STAGE = (size_t)TM * FLD * sizeof(float);  // 67 584
constexpr size_t SIM_FRAG_LDS = SIM_FRAG_STAGE + 4 * 128 * sizeof(float) + 64;

// MODE 0: the scaled S tile goes to the conf buffer and into the (max, sum exp) partials (the eager form: conf_kernel follows);
// MODE 1: partials only, nothing stored (first pass of the lazy form: conf_matrix is not materialised);
// MODE 3: MODE 2 that ALSO stores every confidence: the second pass of the eager form's two-pass variant (large N x M: S is never written
//         and re-read, conf_matrix is written once -- SURVEY 8d's algorithmic traffic; coarse_impl's `two_pass`);
// MODE 2: second pass of the lazy form: the tile is recomputed, turned into confidences with the merged statistics -- the very
//         expression of conf_kernel, so every value is bit-identical to the eager form's -- and only what the selection consumes
//         leaves the chip: per (row, column tile) the best candidate above the threshold (value, lowest j, tie count) and the
//         column maxima.  Reference callers read only the match lists (inference.py:179-180): 134 MB store + 269 MB pass saved.
template <int NS, int MODE>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(1, 2) void sim_frag_kernel(SimFragArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ti, tj;
    if (!xcd_tile(p.ntr, p.ntc, ti, tj)) return;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int i0 = ti * TM, j0 = tj * TN;
    const int nta = 4 * p.ntr, ntb = 4 * p.ntc;
    // this wave's DMA sources: row tile `wave` of the A tile and of the B tile (4 KiB per chunk each, contiguous)
    const char* ga = p.a + ((size_t)b * nta + 4 * ti + wave) * 32768 + 16 * lane;
    const char* gb = p.b + ((size_t)b * ntb + 4 * tj + wave) * 32768 + 16 * lane;
    // one chunk = one k-step (16 features): this wave's A and B row tiles, (hi, lo) fragments: 4 DMAs of 1 KiB
    auto issue = [&](int s, int buf) {
        char* la = smem + buf * FRAG_CHUNK_BYTES + wave * 2048;
        char* lb = la + 8192;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (NS == 1 && f) continue;                              // bf16 mode: hi planes only
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (size_t)(2 * s + f) * 1024),
                                             (__attribute__((address_space(3))) void*)(la + f * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (size_t)(2 * s + f) * 1024),
                                             (__attribute__((address_space(3))) void*)(lb + f * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    OPHIP_STAMP(p.stamps, wg, 0);
    constexpr int NKS = C / 16, G = NS == 3 ? 4 : 2;                 // k-steps; DMAs per chunk and wave
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        // chunk s has landed for this wave (the DMAs of the chunks behind it may stay in flight: counted wait), then for all
        // waves (barrier); every wave is also done reading buffer (s + 3) % 4 = chunk s - 1's (lgkmcnt(0): its reads have
        // returned).  Wait and barrier are ONE asm statement with a memory clobber: the s_barrier builtin alone is no memory
        // barrier to the compiler, which then moves LDS reads / DMA issues across it (seen: rare wrong tiles)
        if (s + 2 < NKS) { if (G == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else if (s + 1 < NKS) { if (G == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 3 < NKS) issue(s + 3, (s + 3) & 3);
        const char* base = smem + (s & 3) * FRAG_CHUNK_BYTES + 16 * lane;
        bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char* pa = base + (2 * wr + t) * 2048;
            const char* pb = base + 8192 + (2 * wc + t) * 2048;
            fah[t] = *reinterpret_cast<const bf16x8*>(pa);
            fbh[t] = *reinterpret_cast<const bf16x8*>(pb);
            fal[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(pa + 1024) : zero_bf8();
            fbl[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(pb + 1024) : zero_bf8();
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = mma_bf16<NS>(fah[x], fal[x], fbh[y], fbl[y], acc[x][y]);
    }
    OPHIP_STAMP(p.stamps, wg, 1);

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    float* St = reinterpret_cast<float*>(smem);
    float* csw = reinterpret_cast<float*>(smem + SIM_FRAG_STAGE);      // [4 (wave)][128] column sums over each wave's 32 rows
    float* wmx = csw + 512;                                            // [4] maximum of each wave's quarter
    int* slow = reinterpret_cast<int*>(wmx + 4);                       // [4] per-wave "needs the exact sweep" flags
    const float inv_temp = 1.0f / p.temp;
    const bool edge = (i0 + TM > p.N) || (j0 + TN > p.M);
    __syncthreads();                                  // every wave is done reading the operand buffers that St overlays
    float vmax = -INFINITY;
    float pad[2] = {0.f, 0.f};                        // sim_matrix += -1e9 on the padded query cells' columns (0 elsewhere: exact)
    if (p.colmask) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = j0 + 64 * wc + 32 * y + r;
            pad[y] = (col < p.M && !p.colmask[(size_t)b * p.M + col]) ? -1e9f : 0.f;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float sv = acc[x][y][reg] * inv_temp + pad[y];
                St[(64 * wr + 32 * x + acc_row(reg, h)) * FLD + 64 * wc + 32 * y + r] = sv;
                vmax = fmaxf(vmax, sv);
            }
    if (MODE >= 2) {
        unsigned long long* bestk = reinterpret_cast<unsigned long long*>(smem + SIM_FRAG_STAGE);     // [128] (value bits << 32) | ~j : max = best value, lowest j
        int* tiecnt = reinterpret_cast<int*>(bestk + 128);                                             // [128]
        if (tid < 128) { bestk[tid] = 0ull; tiecnt[tid] = 0; }
        __syncthreads();
        const int c4 = tid & 31, rg = tid >> 5;
        const float* cst = p.colstat + (size_t)b * p.M * 2;
        const float* rst = p.rowstat + (size_t)b * p.N * 2;
        const float* clg = p.collog + (size_t)b * p.M;
        const float* rlg = p.rowlog + (size_t)b * p.N;
        unsigned* cb = p.colmax_bits + (size_t)b * p.M;
        float* confb = (MODE == 3) ? p.conf + (size_t)b * p.N * p.M : nullptr;
        const bool vec = (p.M & 3) == 0;
        float cmv[4], clv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = min(j0 + 4 * c4 + e, p.M - 1);
            cmv[e] = cst[2 * j]; clv[e] = clg[j];
        }
        // conf = exp(((s - M_c) - log E_c) + ((s - M_r) - log E_r)): conf_kernel<.., FAST>'s expression, term for term
        auto sweep = [&](bool count_ties) {
            bool any = false;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = rg + 8 * it, gi = i0 + row;
                if (gi >= p.N) continue;
                const float rm = rst[2 * gi], rl = rlg[gi];
                const f32x4 v = *reinterpret_cast<const f32x4*>(St + row * FLD + 4 * c4);
                f32x4 cv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gj = j0 + 4 * c4 + e;
                    const float t = ((v[e] - cmv[e]) - clv[e]) + ((v[e] - rm) - rl);
                    if ((MODE == 3 && !count_ties) || (t > p.logthr_lo && gj < p.M)) {
                        const float c = __expf(t);
                        cv[e] = c;
                        if (c > p.thr && gj < p.M) {
                            any = true;
                            const unsigned long long key = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)gj);
                            if (!count_ties) {
                                atomicMax(bestk + row, key);
                                atomicMax(cb + gj, __float_as_uint(c));
                            } else if ((unsigned)(bestk[row] >> 32) == __float_as_uint(c)) {
                                atomicAdd(tiecnt + row, 1);
                            }
                        }
                    }
                }
                if (MODE == 3 && !count_ties) {
                    const int gj = j0 + 4 * c4;
                    if (vec) {
                        if (gj < p.M) __builtin_nontemporal_store(cv, reinterpret_cast<f32x4*>(confb + (size_t)gi * p.M + gj));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gj + e < p.M) confb[(size_t)gi * p.M + gj + e] = cv[e];
                    }
                }
            }
            return any;
        };
        const bool any = sweep(false);
        __syncthreads();
        if (any) sweep(true);                        // candidates are a few per frame and tile: the second look costs nothing
        __syncthreads();
        if (tid < 128 && i0 + tid < p.N) {
            const unsigned long long key = bestk[tid];
            float* o = p.rowbest + (((size_t)b * p.ntc + tj) * p.N + i0 + tid) * 3;
            o[0] = key ? __uint_as_float((unsigned)(key >> 32)) : -1.f;
            o[1] = __int_as_float(key ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0x7fffffff);
            o[2] = __int_as_float(tiecnt[tid]);
        }
        OPHIP_STAMP(p.stamps, wg, 4);
        return;
    }
    const float mw = wave_max_dpp(vmax);
    if (lane == 0) { wmx[wave] = mw; slow[wave] = 0; }
    __syncthreads();
    OPHIP_STAMP(p.stamps, wg, 2);
    // One pass over the staged tile: whole rows go to the conf buffer (16 bytes per lane) and, on the way, into the statistics
    // with ONE exponential per element -- the tile maximum m is the reference of every row and column (any reference >= the true
    // maximum is a valid (max, sum) partial; stat_combine merges references).  A thread owns 4 columns (c4) and 16 rows
    // (tid / 32 + 8 it): column sums accumulate in registers, a row's sum is a 32-lane DPP reduction.
    const float m = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    const bool fast = !edge && (m - m == 0.f);                         // finite reference
    const int c4 = tid & 31, rg = tid >> 5;
    float colacc[4] = {0.f, 0.f, 0.f, 0.f};
    float mine = 1.f;
    {
        float* conf = p.conf + (size_t)b * p.N * p.M;
        const bool vec = (p.M & 3) == 0;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = rg + 8 * it;
            const int gi = i0 + row, gj = j0 + 4 * c4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(St + row * FLD + 4 * c4);
            if (MODE == 0 && gi < p.N && gj < p.M) {
                if (vec) {
                    *reinterpret_cast<f32x4*>(conf + (size_t)gi * p.M + gj) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gj + e < p.M) conf[(size_t)gi * p.M + gj + e] = v[e];
                }
            }
            if (fast) {
                const float e0 = __expf(v[0] - m), e1 = __expf(v[1] - m), e2 = __expf(v[2] - m), e3 = __expf(v[3] - m);
                colacc[0] += e0; colacc[1] += e1; colacc[2] += e2; colacc[3] += e3;
                const float rs = half_sum_dpp((e0 + e1) + (e2 + e3));
                mine = (c4 == it) ? rs : mine;                         // lane c4 < 16 keeps the sum of row rg + 8 c4
            }
        }
    }
    OPHIP_STAMP(p.stamps, wg, 3);
    // a sum below 1e-26 means every term sits > 59 below the reference (a row whose own maximum is that far under the tile's):
    // its partial would lose bits against f32's range, so the workgroup takes the exact per-row / per-column sweep instead
    constexpr float TINY = 1e-26f;
    if (fast) {
        const f32x4 cs = {swap32_sum(colacc[0]), swap32_sum(colacc[1]), swap32_sum(colacc[2]), swap32_sum(colacc[3])};
        if (lane < 32) *reinterpret_cast<f32x4*>(csw + wave * 128 + 4 * c4) = cs;
        if (!__all(mine >= TINY) && lane == 0) slow[wave] = 1;
    }
    __syncthreads();
    float ctot = 1.f;
    if (fast && tid < 128) {
        ctot = (csw[tid] + csw[128 + tid]) + (csw[256 + tid] + csw[384 + tid]);
        if (!(ctot >= TINY)) slow[0] = 1;
    }
    __syncthreads();
    if (fast && (slow[0] | slow[1] | slow[2] | slow[3]) == 0) {
        if (c4 < 16) {
            float* o = p.rowpart + (((size_t)b * p.ntc + tj) * p.N + i0 + rg + 8 * c4) * 2;
            o[0] = m; o[1] = mine;
        }
        if (tid < 128) {
            float* o = p.colpart + (((size_t)b * p.ntr + ti) * p.M + j0 + tid) * 2;
            o[0] = m; o[1] = ctot;
        }
    } else {
        SimArgs q{nullptr, nullptr, nullptr, p.rowpart, p.colpart, p.N, p.M, p.ntr, p.ntc, p.temp, nullptr, nullptr};
        tile_stats_lds<true, FLD>(q, St, tid, i0, j0, b);
    }
    OPHIP_STAMP(p.stamps, wg, 4);
}


// ---- round 5: the same tile with THREE workgroups per CU (opt-in: OPHIP_SIM_TILE=3; see the measurement at its launch site) --------------
// sim_frag_kernel holds 69.7 KB of LDS (four 16 KiB operand buffers, overlaid by the 67.6 KB staging image of the whole tile): two
// workgroups per CU.  Its stamps (profiles/r02-r04) show a k-loop at the matrix pipe's rate and an epilogue of the same length that no
// matrix work covers, and 2 090 tiles on 512 slots are 4.08 rounds that cost 5.  This form keeps the tile, the wave split and the
// arithmetic and cuts the LDS to 51.3 KB -- THREE operand buffers (chunks two k-steps ahead instead of three) and a staging image of HALF
// the tile (64 rows) that the two row halves pass through one after the other -- so that three workgroups share a CU (138 registers: the
// register file allows it): a third wave per SIMD to run its k-loop under the other two's epilogues, and 768 slots (2.72 rounds that cost 3).
//   MODE 0 / 1 as sim_frag_kernel (1: statistics only, the lazy form's first pass).  The fast pass accumulates row and column sums in the
//   association of sim_frag_kernel (rows rg + 8 it in the same order, the same DPP and cross-wave sums), so its statistics are bit-identical;
//   the exact pass (edge tiles, tiles with a row or column > 59 below the tile maximum) merges a column's two halves as (max, sum) partials.
constexpr int F3_BUFS = 3;
constexpr int F3_HALF = TM / 2;                                          // rows per staging pass
constexpr size_t SIM_F3_STAGE = (size_t)F3_HALF * FLD * sizeof(float);   // 33 792
constexpr size_t SIM_F3_LDS = (size_t)F3_BUFS * FRAG_CHUNK_BYTES + 4 * 128 * sizeof(float) + 64;      // 51 264
static_assert(SIM_F3_STAGE <= (size_t)F3_BUFS * FRAG_CHUNK_BYTES, "the half-tile staging image overlays the operand buffers");

template <int NS, int MODE>
__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(3, 3) void sim_frag3_kernel(SimFragArgs p) {
    static_assert(MODE == 0 || MODE == 1, "the candidate pass of the lazy form stays on sim_frag_kernel<NS, 2>");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ti, tj;
    if (!xcd_tile(p.ntr, p.ntc, ti, tj)) return;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int i0 = ti * TM, j0 = tj * TN;
    const int nta = 4 * p.ntr, ntb = 4 * p.ntc;
    const char* ga = p.a + ((size_t)b * nta + 4 * ti + wave) * 32768 + 16 * lane;
    const char* gb = p.b + ((size_t)b * ntb + 4 * tj + wave) * 32768 + 16 * lane;
    auto issue = [&](int s, int buf) {
        char* la = smem + buf * FRAG_CHUNK_BYTES + wave * 2048;
        char* lb = la + 8192;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (NS == 1 && f) continue;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (size_t)(2 * s + f) * 1024),
                                             (__attribute__((address_space(3))) void*)(la + f * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (size_t)(2 * s + f) * 1024),
                                             (__attribute__((address_space(3))) void*)(lb + f * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = zero16();
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    OPHIP_STAMP(p.stamps, wg, 0);
    constexpr int NKS = C / 16, G = NS == 3 ? 4 : 2;
    issue(0, 0);
    issue(1, 1);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        // chunk s has landed for this wave (chunk s + 1 may stay in flight), then for all waves; every wave is done reading buffer
        // (s + 2) % 3 = chunk s - 1's.  One asm statement with a memory clobber (see sim_frag_kernel)
        if (s + 1 < NKS) { if (G == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 2 < NKS) issue(s + 2, (s + 2) % F3_BUFS);
        const char* base = smem + (s % F3_BUFS) * FRAG_CHUNK_BYTES + 16 * lane;
        bf16x8 fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const char* pa = base + (2 * wr + t) * 2048;
            const char* pb = base + 8192 + (2 * wc + t) * 2048;
            fah[t] = *reinterpret_cast<const bf16x8*>(pa);
            fbh[t] = *reinterpret_cast<const bf16x8*>(pb);
            fal[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(pa + 1024) : zero_bf8();
            fbl[t] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(pb + 1024) : zero_bf8();
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = mma_bf16<NS>(fah[x], fal[x], fbh[y], fbl[y], acc[x][y]);
    }
    OPHIP_STAMP(p.stamps, wg, 1);

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    float* St = reinterpret_cast<float*>(smem);                        // [64][FLD]: one row half of the tile at a time
    float* csw = reinterpret_cast<float*>(smem + F3_BUFS * FRAG_CHUNK_BYTES);      // [4 (wave)][128]
    float* wmx = csw + 512;
    int* slow = reinterpret_cast<int*>(wmx + 4);
    const float inv_temp = 1.0f / p.temp;
    const bool edge = (i0 + TM > p.N) || (j0 + TN > p.M);
    float vmax = -INFINITY;
    float pad[2] = {0.f, 0.f};
    if (p.colmask) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int col = j0 + 64 * wc + 32 * y + r;
            pad[y] = (col < p.M && !p.colmask[(size_t)b * p.M + col]) ? -1e9f : 0.f;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float sv = acc[x][y][reg] * inv_temp + pad[y];      // (the scaled value stays in the accumulator: staged once or twice below)
                acc[x][y][reg] = sv;
                vmax = fmaxf(vmax, sv);
            }
    const float mw = wave_max_dpp(vmax);
    if (lane == 0) { wmx[wave] = mw; slow[wave] = 0; }
    __syncthreads();                                  // also: every wave is done reading the operand buffers that St overlays
    OPHIP_STAMP(p.stamps, wg, 2);
    const float m = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    const bool fast = !edge && (m - m == 0.f);
    const int c4 = tid & 31, rg = tid >> 5;
    auto stage_half = [&](int half) {                 // the two waves that hold rows 64 half .. 64 half + 63 write them into St
        if (wr == half) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) St[(32 * x + acc_row(reg, h)) * FLD + 64 * wc + 32 * y + r] = acc[x][y][reg];
        }
    };
    float colacc[4] = {0.f, 0.f, 0.f, 0.f};
    float mine[2] = {1.f, 1.f};
    float* conf = (MODE == 0) ? p.conf + (size_t)b * p.N * p.M : nullptr;
    const bool vec = (p.M & 3) == 0;
    if (MODE == 0 || fast) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            stage_half(half);
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int row = rg + 8 * it;
                const int gi = i0 + F3_HALF * half + row, gj = j0 + 4 * c4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(St + row * FLD + 4 * c4);
                if (MODE == 0 && gi < p.N && gj < p.M) {
                    if (vec) {
                        *reinterpret_cast<f32x4*>(conf + (size_t)gi * p.M + gj) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gj + e < p.M) conf[(size_t)gi * p.M + gj + e] = v[e];
                    }
                }
                if (fast) {
                    const float e0 = __expf(v[0] - m), e1 = __expf(v[1] - m), e2 = __expf(v[2] - m), e3 = __expf(v[3] - m);
                    colacc[0] += e0; colacc[1] += e1; colacc[2] += e2; colacc[3] += e3;
                    const float rs = half_sum_dpp((e0 + e1) + (e2 + e3));
                    mine[half] = (c4 == it) ? rs : mine[half];            // lane c4 < 8 keeps the sum of row 64 half + rg + 8 c4
                }
            }
            __syncthreads();                          // the image is rewritten by the other half (or by the exact pass)
        }
    }
    OPHIP_STAMP(p.stamps, wg, 3);
    constexpr float TINY = 1e-26f;
    if (fast) {
        const f32x4 cs = {swap32_sum(colacc[0]), swap32_sum(colacc[1]), swap32_sum(colacc[2]), swap32_sum(colacc[3])};
        if (lane < 32) *reinterpret_cast<f32x4*>(csw + wave * 128 + 4 * c4) = cs;
        if (!__all(mine[0] >= TINY && mine[1] >= TINY) && lane == 0) slow[wave] = 1;
    }
    __syncthreads();
    float ctot = 1.f;
    if (fast && tid < 128) {
        ctot = (csw[tid] + csw[128 + tid]) + (csw[256 + tid] + csw[384 + tid]);
        if (!(ctot >= TINY)) slow[0] = 1;
    }
    __syncthreads();
    if (fast && (slow[0] | slow[1] | slow[2] | slow[3]) == 0) {
        if (c4 < 8) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float* o = p.rowpart + (((size_t)b * p.ntc + tj) * p.N + i0 + F3_HALF * half + rg + 8 * c4) * 2;
                o[0] = m; o[1] = mine[half];
            }
        }
        if (tid < 128) {
            float* o = p.colpart + (((size_t)b * p.ntr + ti) * p.M + j0 + tid) * 2;
            o[0] = m; o[1] = ctot;
        }
    } else {
        // exact pass, half by half: rows as in tile_stats_lds (two lanes per row, even / odd columns, true row maximum); a column's two
        // halves are (max, sum) partials of their own, merged in row order
        const int idx = tid >> 1, par = tid & 1;
        float cm = -INFINITY, ce = 0.f;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            stage_half(half);
            __syncthreads();
            // (sixteen independent LDS reads at a time into registers, then their maximum resp. their exponentials: the loops are
            //  latency-bound, and the edge tiles of the last column band are the LAST tiles an XCD walks -- a slow exact pass is the kernel's tail)
            if (tid < 2 * F3_HALF) {                  // 64 rows x 2 lanes
                const int ncol = min(TN, p.M - j0);
                const float* rowp = St + idx * FLD + par;
                float mr = -INFINITY, er = 0.f;
#pragma unroll 1
                for (int q0 = 0; q0 < TN / 2; q0 += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = (2 * (q0 + u) + par < ncol) ? rowp[2 * (q0 + u)] : -INFINITY;
#pragma unroll
                    for (int u = 0; u < 16; ++u) mr = fmaxf(mr, v[u]);
                }
                if (mr != -INFINITY) {
#pragma unroll 1
                    for (int q0 = 0; q0 < TN / 2; q0 += 16) {
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = (2 * (q0 + u) + par < ncol) ? rowp[2 * (q0 + u)] : -INFINITY;
#pragma unroll
                        for (int u = 0; u < 16; ++u) er += __expf(v[u] - mr);          // exp(-inf) = 0 for the masked tail
                    }
                }
                const float m2 = __shfl_xor(mr, 1, 64), e2 = __shfl_xor(er, 1, 64);
                float ma = par ? m2 : mr, ea = par ? e2 : er, mb = par ? mr : m2, eb = par ? er : e2;      // even lane's part first
                merge_ms(ma, ea, mb, eb);
                const int gi = i0 + F3_HALF * half + idx;
                if (par == 0 && gi < p.N) {
                    float* o = p.rowpart + (((size_t)b * p.ntc + tj) * p.N + gi) * 2;
                    o[0] = ma; o[1] = ea;
                }
            }
            {                                         // 128 columns x 2 lanes: rows par, par + 2, ... of this half
                const int nrow = min(F3_HALF, p.N - i0 - F3_HALF * half);
                const float* colp = St + par * FLD + idx;
                float mc = -INFINITY, ec = 0.f;
#pragma unroll 1
                for (int q0 = 0; q0 < F3_HALF / 2; q0 += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = (2 * (q0 + u) + par < nrow) ? colp[2 * (q0 + u) * FLD] : -INFINITY;
#pragma unroll
                    for (int u = 0; u < 16; ++u) mc = fmaxf(mc, v[u]);
                }
                if (mc != -INFINITY) {
#pragma unroll 1
                    for (int q0 = 0; q0 < F3_HALF / 2; q0 += 16) {
                        float v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = (2 * (q0 + u) + par < nrow) ? colp[2 * (q0 + u) * FLD] : -INFINITY;
#pragma unroll
                        for (int u = 0; u < 16; ++u) ec += __expf(v[u] - mc);
                    }
                }
                merge_ms(cm, ce, mc, ec);
            }
            __syncthreads();
        }
        const float m2 = __shfl_xor(cm, 1, 64), e2 = __shfl_xor(ce, 1, 64);
        float ma = par ? m2 : cm, ea = par ? e2 : ce, mb = par ? cm : m2, eb = par ? ce : e2;
        merge_ms(ma, ea, mb, eb);
        if (par == 0 && j0 + idx < p.M) {
            float* o = p.colpart + (((size_t)b * p.ntr + ti) * p.M + j0 + idx) * 2;
            o[0] = ma; o[1] = ea;
        }
    }
    OPHIP_STAMP(p.stamps, wg, 4);
}


struct CombineArgs {
    const float *rowpart, *colpart;
    float *rowstat, *colstat;     // [B][N][2], [B][M][2]
    unsigned* colmax_bits;        // [B][M], cleared here for the conf pass's atomicMax
    int N, M, ntr, ntc;
    float *rowlog, *collog;       // optional [B][N], [B][M]: logf(sum exp) for the lazy form's candidate pass (NULL: not written)
    int* lazy_flag;               // optional: count[1], cleared here (select_decide sets it on an exact row tie it cannot resolve without conf)
};

// 8 lanes per row / column: lane q merges partials q, q+8, ... in order, then the 8 are merged by an xor
// butterfly whose operand order is fixed (lower lane first), so the result is deterministic.
__global__ __launch_bounds__(256) void stat_combine_kernel(CombineArgs p) {
    const int gid = blockIdx.x * 32 + (threadIdx.x >> 3), q = threadIdx.x & 7, b = blockIdx.y;
    const bool is_row = gid < p.N;
    const int idx = is_row ? gid : gid - p.N;
    const bool live = gid < p.N + p.M;
    if (p.lazy_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *p.lazy_flag = 0;
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
        float* lg = is_row ? p.rowlog : p.collog;
        if (lg) lg[(size_t)b * len + idx] = logf(e);
    }
}

struct ConfArgs {
    float* conf;
    const float *rowstat, *colstat;
    float* rowbest;          // [B][nspan][N][3]  (value, j as float bits, tie count as float bits)
    unsigned* colmax_bits;   // [B][M] column maxima as float bits (conf >= 0, so unsigned order == float order)
    int N, M, nspan, spanw, nrb;
    float thr;               // match threshold (strict >): entries at or below it are never tracked
    int rows;                // rows per workgroup (a multiple of 2 * CONF_RB, <= CONF_ROWS_MAX)
};

#ifndef OPHIP_CONF_ROWS
#define OPHIP_CONF_ROWS 32
#endif
constexpr int CONF_ROWS = OPHIP_CONF_ROWS;      // default rows per workgroup (measured ALONE at c2 with CONF_RB = 2: 16 -> 106 us, 32 -> 90, 48 -> 114, 64 -> 132: fewer, less contended column atomics vs grid fill)
constexpr int CONF_ROWS_MAX = 512;              // OPHIP_CONF_ROWS_RT (run time): more rows per workgroup = fewer workgroups holding slots beside the fine stage.
                                                // Round 4, c2, in the pipeline (frames/s, 100 steps, interleaved, two boxes): rows 32 / 48 / 64 / 80 / 96 at CONF_RB 2 and 4 all within
                                                // the +-3 % run-to-run spread (1 428-1 472), 128 and 256 worse (the pass, 143 / 275 us alone, becomes the frame's longest kernel)
#ifndef OPHIP_CONF_RB
#define OPHIP_CONF_RB 1
#endif
constexpr int CONF_RB = OPHIP_CONF_RB;          // rows per pipeline stage (two stages in flight): what a workgroup keeps in flight is 2 * CONF_RB * CONF_U KiB per wave
#ifndef OPHIP_CONF_U
#define OPHIP_CONF_U 1
#endif
// float4 groups per thread per row => a span of <= 1024 CONF_U columns.  Alone at c2: 2 -> 94 us (123 VGPRs), 3 -> 90 us (164), 4 -> 87 us (212).
// Beside the previous frame's fine stage (whose waves hold 256 registers, two per SIMD: a wave of this pass enters a SIMD in place of one
// of them) 2 and 3 measure the same within noise (1355 against 1330 frames/s over four runs each, fine stage 292 us in both traces): the
// bytes a wave keeps in flight per register it holds are the same.
// End of round 4: CONF_U 1 and CONF_RB 1 (52 registers instead of 138).  The pass runs in the window where the previous frame's fine stage
// holds every register of every SIMD it sits on and gets the slots that stage's retiring workgroups leave: at 52 registers four of its waves
// fit where one did.  Interleaved whole-process runs, c2, 100 steps, three boxes: 1 438 -> 1 491, 1 484 -> 1 489, 1 510 -> 1 538 (medians of
// 3, 4 and 8 rounds); the intermediate settings (U 1 / RB 2: 63 registers, U 2 / RB 1: 86) fall between.
constexpr int CONF_U = OPHIP_CONF_U;

template <bool VEC, bool FAST>
__global__ __launch_bounds__(256) void conf_kernel(ConfArgs p) {
    // dynamic LDS, p.rows rows: [4][rows] best value | [4][rows] its column | [4][rows] tie count | [rows][2] row statistics
    extern __shared__ __attribute__((aligned(16))) float conf_lds[];
    const int ROWS = p.rows;
    float* red_v = conf_lds;
    int* red_j = reinterpret_cast<int*>(conf_lds + 4 * ROWS);
    int* red_c = reinterpret_cast<int*>(conf_lds + 8 * ROWS);
    float* rstat_s = conf_lds + 12 * ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int span = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    const int jb = span * p.spanw, je = min(p.M, jb + p.spanw);
    const int i0 = rb * ROWS;
    float* conf = p.conf + (size_t)b * p.N * p.M;
    const float* cst = p.colstat + (size_t)b * p.M * 2;
    const float* rst = p.rowstat + (size_t)b * p.N * 2;

    // Row / column statistics once per workgroup (rows through LDS: a global load inside the row loop is an L2 round trip per
    // row pair).  Exact mode: (max, 1 / sum) of the true maxima.  bf16 modes: the merged partials' reference is a tile maximum,
    // possibly far above a row's own, so both its inverse sum and its exponent can leave f32's range when multiplied out; the
    // log form conf = exp((s - M_c - log E_c) + (s - M_r - log E_r)) has every term <= 0 and needs no division at all.
    for (int t = tid; t < ROWS; t += 256) {
        const int i = min(i0 + t, p.N - 1);
        rstat_s[2 * t] = rst[2 * i];
        rstat_s[2 * t + 1] = FAST ? logf(rst[2 * i + 1]) : 1.0f / rst[2 * i + 1];
#pragma unroll
        for (int w = 0; w < 4; ++w) { red_v[w * ROWS + t] = -1.f; red_j[w * ROWS + t] = 0x7fffffff; red_c[w * ROWS + t] = 0; }
    }
    float cm[CONF_U][4], cinv[CONF_U][4], cbest[CONF_U][4];
#pragma unroll
    for (int u = 0; u < CONF_U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + 4 * tid + 1024 * u + e;
            cm[u][e] = (j < je) ? cst[2 * j] : 0.f;
            cinv[u][e] = (j < je) ? (FAST ? logf(cst[2 * j + 1]) : 1.0f / cst[2 * j + 1]) : (FAST ? 0.f : 1.f);
            cbest[u][e] = 0.f;
        }
    const int nrows = min(ROWS, p.N - i0);
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
                    // streamed once in, once out: non-temporal, so that the pass does not wash the L2 of what runs beside it (the previous
                    // frame's fine stage re-reads its 1.3 MB of weights per match pair: 1403 against 1363 frames/s at c2)
                    if (jq < je) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + jq));
                    dst[q][u][0] = v[0]; dst[q][u][1] = v[1]; dst[q][u][2] = v[2]; dst[q][u][3] = v[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[q][u][e] = (jq + e < je) ? row[jq + e] : 0.f;
                }
            }
        }
    };
    // Only entries above the threshold can become matches (get_coarse_match masks conf > thr before the mutual test, and a
    // row / column maximum that matters is itself such an entry), so the running row best, its tie count and the column maxima
    // are tracked for those entries alone: the common element costs one compare, and a wave reduces a row only when one of
    // its lanes saw a candidate.  Rows / columns without a candidate keep (-1, none, 0) / 0, which select reads as "no match".
    auto process_batch = [&](int r0, float (&cur)[CONF_RB][CONF_U][4]) {
#pragma unroll
        for (int q = 0; q < CONF_RB; ++q) {
            const int rr = r0 + q;
            if (rr < nrows) {
                const int i = i0 + rr;
                const float rm = rstat_s[2 * rr], rinv = rstat_s[2 * rr + 1];
                float* row = conf + (size_t)i * p.M;
                bool hit = false;
#pragma unroll
                for (int u = 0; u < CONF_U; ++u) {
                    const int jq = jb + 4 * tid + 1024 * u;
                    if (jq < je) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // softmax over the 3D axis (dim=1: column stats) times softmax over the 2D axis (dim=2: row stats);
                            // the bf16 modes (error budget 1e-5) fold the two exponentials into one (cinv / rinv = log sums there)
                            const float sv = cur[q][u][e];
                            const float c = FAST ? exp_sel<true>(((sv - cm[u][e]) - cinv[u][e]) + ((sv - rm) - rinv))
                                                 : (exp_sel<false>(sv - cm[u][e]) * cinv[u][e]) * (exp_sel<false>(sv - rm) * rinv);
                            cur[q][u][e] = c;
                            hit |= (c > p.thr) && (jq + e < je);
                        }
                        if (VEC) {
                            f32x4 v = {cur[q][u][0], cur[q][u][1], cur[q][u][2], cur[q][u][3]};
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + jq));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (jq + e < je) row[jq + e] = cur[q][u][e];
                        }
                    }
                }
                if (__any(hit)) {
                    float bv = -1.f;
                    int bj = 0x7fffffff, bc = 0;
                    if (hit) {
#pragma unroll
                        for (int u = 0; u < CONF_U; ++u) {
                            const int jq = jb + 4 * tid + 1024 * u;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float c = cur[q][u][e];
                                if (jq + e < je && c > p.thr) {
                                    cbest[u][e] = fmaxf(cbest[u][e], c);
                                    if (c > bv) { bv = c; bj = jq + e; bc = 1; }
                                    else if (c == bv) { bc += 1; bj = min(bj, jq + e); }
                                }
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
                    if (lane == 0) { red_v[wave * ROWS + rr] = wv; red_j[wave * ROWS + rr] = cj; red_c[wave * ROWS + rr] = cc; }
                }
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
    for (int t = tid; t < nrows; t += 256) {
        float v = red_v[t];
        int j = red_j[t], c = red_c[t];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float v2 = red_v[w * ROWS + t];
            if (v2 > v) { v = v2; j = red_j[w * ROWS + t]; c = red_c[w * ROWS + t]; }
            else if (v2 == v) { j = min(j, red_j[w * ROWS + t]); c += red_c[w * ROWS + t]; }
        }
        float* o = p.rowbest + (((size_t)b * p.nspan + span) * p.N + i0 + t) * 3;
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
    int border_mode;            // 0: OnePose++ (top rows / left columns of the query grid only: coarse_matching.py:19-20 slices -b:0 are empty);
                                // 1: LoFTR (all four sides of BOTH grids: loftr/utils/coarse_matching.py mask_border)
    int wi;                     // border_mode 1: width of the i grid (N = hi * wi)
    float thr, scale;
    const float* qscale;        // [B][2] query_image_scale (h, w factors; coarse_matching.py:224) or NULL
    long long* b_ids; long long* i_ids; long long* j_ids;
    float* mconf; float* mk3d; float* mkq;
    long long* m_bids;          // optional second copy of b_ids (the reference's 'm_bids')
    unsigned char* gt_mask;     // optional mconf == 0 flags (the reference's 'gt_mask')
    int* count;
};

// Selection in two small kernels (round 4; the single 8 x 1024-thread kernel of round 3 needed whole CUs, which beside the previous
// frame's fine stage -- two 256-register waves on every SIMD -- it got only after 50-200 us):
//   select_decide: one thread per 3D point (b, i): merges its row-best records, applies threshold, border and the mutual test (and the
//                  exact-tie walk along the stored row, coarse_matching.py:166 "first true j"), writes j or -1 and a count per workgroup;
//   select_place:  the same grid; a workgroup adds up the counts of the workgroups before it (<= a few hundred ints) and writes its own
//                  matches at their final positions -- ascending (b, i), the order of torch.where in the reference (coarse_matching.py:170).
// 256-thread workgroups, < 64 registers, 1 KiB of LDS: they fit wherever one wave slot per SIMD is free.
constexpr int SEL_T = 256;

__global__ __launch_bounds__(SEL_T) void select_decide_kernel(SelectArgs p, int* __restrict__ dec, int* __restrict__ wgcount) {
    __shared__ int wsum[SEL_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long g = (long long)blockIdx.x * SEL_T + tid;
    const long long total = (long long)p.B * p.N;
    int jsel = -1;
    if (g < total) {
        const int b = (int)(g / p.N), i = (int)(g % p.N);
        float v = -1.f;
        int j = 0x7fffffff, c = 0;
        for (int sp0 = 0; sp0 < p.nspan; sp0 += 4) {          // four records in flight (the lazy form has one per column tile: 38 at c2)
            float rv[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sp = min(sp0 + u, p.nspan - 1);
                const float* q = p.rowbest + (((size_t)b * p.nspan + sp) * p.N + i) * 3;
                rv[u][0] = q[0]; rv[u][1] = q[1]; rv[u][2] = q[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (sp0 + u >= p.nspan) break;
                const float v2 = rv[u][0];
                const int j2 = __float_as_int(rv[u][1]), c2 = __float_as_int(rv[u][2]);
                if (v2 > v) { v = v2; j = j2; c = c2; }
                else if (v2 == v) { j = min(j, j2); c += c2; }
            }
        }
        auto inside = [&](int jj) {
            const int jy = jj / p.wc, jx = jj % p.wc;
            if (p.border_mode == 0) return (jy >= p.border) && (jx >= p.border);
            return jy >= p.border && jx >= p.border && jy < p.M / p.wc - p.border && jx < p.wc - p.border;
        };
        bool i_ok = true;
        if (p.border_mode == 1) {
            const int iy = i / p.wi, ix = i % p.wi;
            i_ok = iy >= p.border && ix >= p.border && iy < p.N / p.wi - p.border && ix < p.wi - p.border;
        }
        if (v > p.thr && i_ok) {
            const float* cmx = p.colmax + (size_t)b * p.M;
            bool o = inside(j) && v == cmx[j];
            if (!o && c > 1) {
                // exact tie of the row maximum: the reference takes the first j whose mask is true
                if (p.conf) {
                    const float* row = p.conf + ((size_t)b * p.N + i) * p.M;
                    for (int jj = j + 1; jj < p.M; ++jj)
                        if (row[jj] == v && inside(jj) && v == cmx[jj]) { j = jj; o = true; break; }
                } else {
                    p.count[1] = 1;          // lazy form: the other tied columns were never stored -> the caller re-runs this frame eagerly
                }
            }
            if (o) jsel = j;
        }
        dec[g] = jsel;
    }
    const unsigned long long mask = __ballot(jsel >= 0);
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    if (tid == 0) wgcount[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(SEL_T) void select_place_kernel(SelectArgs p, const int* __restrict__ dec, const int* __restrict__ wgcount) {
    __shared__ int red[SEL_T / 64];
    __shared__ int wpre[SEL_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // matches of all workgroups before this one (integer sums: any order gives the same result)
    int part = 0;
    for (int w = tid; w < (int)blockIdx.x; w += SEL_T) part += wgcount[w];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) red[wave] = part;
    const long long g = (long long)blockIdx.x * SEL_T + tid;
    const long long total = (long long)p.B * p.N;
    const int j = g < total ? dec[g] : -1;
    const unsigned long long mask = __ballot(j >= 0);
    if (lane == 0) wpre[wave] = __popcll(mask);
    __syncthreads();
    const int base = (red[0] + red[1]) + (red[2] + red[3]);
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wpre[w];
    if (j >= 0) {
        const int b = (int)(g / p.N), i = (int)(g % p.N);
        const int pos = base + before + __popcll(mask & ((1ull << lane) - 1ull));
        const float* q = p.rowbest + ((size_t)b * p.nspan * p.N + i) * 3;
        float v = q[0];
        for (int sp = 1; sp < p.nspan; ++sp) v = fmaxf(v, q[(size_t)sp * p.N * 3]);
        const float* kpb = p.kpts + (size_t)b * p.kpts_bs + (size_t)i * 3;
        p.b_ids[pos] = b; p.i_ids[pos] = i; p.j_ids[pos] = j;
        p.mconf[pos] = v;
        p.mk3d[3 * pos] = kpb[0]; p.mk3d[3 * pos + 1] = kpb[1]; p.mk3d[3 * pos + 2] = kpb[2];
        // scale_total = scale * query_image_scale[b][[1, 0]] (f32), then (x, y) * scale_total
        const float sx = p.qscale ? __fmul_rn(p.scale, p.qscale[2 * b + 1]) : p.scale;
        const float sy = p.qscale ? __fmul_rn(p.scale, p.qscale[2 * b]) : p.scale;
        p.mkq[2 * pos] = (float)(j % p.wc) * sx;
        p.mkq[2 * pos + 1] = (float)(j / p.wc) * sy;
        if (p.m_bids) p.m_bids[pos] = b;
        if (p.gt_mask) p.gt_mask[pos] = v == 0.f ? 1 : 0;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *p.count = base + (wpre[0] + wpre[1]) + (wpre[2] + wpre[3]);
}

inline int conf_nspan(int M) { return (M + 1024 * CONF_U - 1) / (1024 * CONF_U); }
inline int conf_spanw(int M) { const int ns = conf_nspan(M); return (((M + ns - 1) / ns) + 3) / 4 * 4; }

// workspace map (floats), shared by the sizing helper, coarse_impl and the fragment-plane accessor
struct CoarseWs {
    size_t rowpart, colpart, rowstat, colstat, rowbest, colmax, rowlog, collog, seldec, selcnt, planes, total;
    int nspan_cap;                                   // row-best records per row: conf_kernel's spans or (lazy form) the column tiles
};
CoarseWs coarse_ws(int B, int N, int M) {
    const size_t ntr = (N + TM - 1) / TM, ntc = (M + TN - 1) / TN;
    CoarseWs w;
    w.nspan_cap = (int)(ntc > (size_t)conf_nspan(M) ? ntc : (size_t)conf_nspan(M));
    size_t f = 0;
    w.rowpart = f; f += (size_t)B * ntc * N * 2;
    w.colpart = f; f += (size_t)B * ntr * M * 2;
    w.rowstat = f; f += (size_t)B * N * 2;
    w.colstat = f; f += (size_t)B * M * 2;
    w.rowbest = f; f += (size_t)B * w.nspan_cap * N * 3;
    w.colmax = f; f += (size_t)B * M;
    w.rowlog = f; f += (size_t)B * N;
    w.collog = f; f += (size_t)B * M;
    w.seldec = f; f += (size_t)B * N;                                   // select_decide: chosen j (int) or -1 per 3D point
    w.selcnt = f; f += ((size_t)B * N + SEL_T - 1) / SEL_T + 4;        // ... and the matches per workgroup
    w.planes = f; f += (size_t)B * (ntr + ntc) * TM * C + 64;      // fragment planes of both inputs (hi + lo bf16 = 4 bytes per element), rows padded to 128
    w.total = f + 64;
    return w;
}

}  // namespace

extern "C" size_t ophip_coarse_workspace_floats(int B, int N, int M) { return coarse_ws(B, N, M).total; }

// Which form the eager confidence matrix takes in the bf16 modes (internal; csrc/frame.hip orders its streams by it):
//   one pass  (default at c2): similarity tiles store S, conf_kernel turns it into confidences in place -- S written, read and rewritten
//             (3 N x M passes over HBM), but the read-modify-write pass needs no matrix pipe and runs beside the previous frame's fine stage;
//   two passes: statistics tiles, then tiles again that write every confidence once (1 N x M pass, 2 x the tile arithmetic).
// OPHIP_COARSE_TWO_PASS = 0 / 1 forces one form; default: two passes from `OPHIP_COARSE_TWO_PASS_MIN` (default 2^27 = 134 M) matrix elements
// per frame on -- BASELINE config 4 (15 000 x 19 200 = 288 M) yes, config 2 (33.6 M) no; see DESIGN.md for the measurement behind it.
bool ophip_coarse_two_pass(int B, int N, int M) {
    (void)B;
    // (read on every call -- two getenv per frame -- so that a test can compare the forms inside one process)
    const char* f = getenv("OPHIP_COARSE_TWO_PASS");
    if (f && f[0]) return atoi(f) != 0;
    const char* e = getenv("OPHIP_COARSE_TWO_PASS_MIN");
    const long long min_elems = e && e[0] ? atoll(e) : (1LL << 27);
    return (long long)N * M >= min_elems;
}

namespace {
// where the fragment planes of the bf16 modes live inside the workspace (behind the partials; 64-byte aligned)
void frag_plane_ptrs(float* workspace, int B, int N, int M, char** a, char** b) {
    const size_t ntr = (N + TM - 1) / TM;
    float* w2 = workspace + coarse_ws(B, N, M).planes;
    w2 += (16 - ((reinterpret_cast<uintptr_t>(w2) >> 2) & 15)) & 15;
    *a = reinterpret_cast<char*>(w2);
    *b = *a + (size_t)B * ntr * 4 * 32768;
}

// parts (bit mask): 1 = similarity + confidence kernels (= 4 | 8), 2 = select (threshold, mutual test, compaction), 3 = the whole stage;
// 4 = the similarity tiles with their statistics only, 8 = the rest of 1 (statistics merge, then conf_kernel -- or the candidate pass of
// the lazy form): a pipeline may put other work between the matrix-bound and the HBM-bound half
int coarse_impl(int parts, int border_mode, int wi, double temp_eps,
                const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                int* count, int nsplit, void* stream_, const unsigned char* qmask = nullptr, const float* qscale = nullptr) {
    if (!feat3d || !feat2d || !keypoints3d || !workspace || !b_ids || !i_ids || !j_ids || !mconf || !mkpts3d || !mkpts_c || !count)
        return ophip_bad_arg(__func__, "null pointer");
    const bool lazy = conf == nullptr;               // conf_matrix not requested: nothing N x M is stored (bf16 modes)
    if (B < 1 || N < 1 || M < 1 || wc < 1 || M % wc != 0) return ophip_bad_arg(__func__, "bad sizes (need M == hc * wc)");
    if (border_mode == 1 && (wi < 1 || N % wi != 0)) return ophip_bad_arg(__func__, "bad sizes (need N == h0c * w0c)");
    const bool planes_ready = (nsplit & OPHIP_COARSE_PLANES_READY) != 0;       // the caller wrote the fragment planes (ophip_coarse_frag_planes)
    nsplit &= ~OPHIP_COARSE_PLANES_READY;
    if (nsplit != 0 && nsplit != 1 && nsplit != 3) return ophip_bad_arg(__func__, "nsplit must be 0 (exact f32), 1 (bf16) or 3 (split bf16)");
    if (planes_ready && nsplit == 0) return ophip_bad_arg(__func__, "fragment planes are an input of the bf16 modes only");
    if (lazy && nsplit == 0) return ophip_bad_arg(__func__, "conf == NULL (lazy conf_matrix) needs a bf16 mode: the exact-f32 mode always materialises it");
    hipStream_t stream = (hipStream_t)stream_;
    // rows per conf workgroup: OPHIP_CONF_ROWS_RT (a multiple of 2 * CONF_RB up to CONF_ROWS_MAX), default CONF_ROWS
    static const int conf_rows = [] {
        const char* e = getenv("OPHIP_CONF_ROWS_RT");
        int r = e ? atoi(e) : CONF_ROWS;
        r = r < 2 * CONF_RB ? 2 * CONF_RB : (r > CONF_ROWS_MAX ? CONF_ROWS_MAX : r);
        return r / (2 * CONF_RB) * (2 * CONF_RB);
    }();
    const int ntr = (N + TM - 1) / TM, ntc = (M + TN - 1) / TN, nrb = (N + conf_rows - 1) / conf_rows;
    const int nspan = conf_nspan(M), spanw = conf_spanw(M);
    const CoarseWs ws = coarse_ws(B, N, M);
    float* rowpart = workspace + ws.rowpart;
    float* colpart = workspace + ws.colpart;
    float* rowstat = workspace + ws.rowstat;
    float* colstat = workspace + ws.colstat;
    float* rowbest = workspace + ws.rowbest;
    float* colmax = workspace + ws.colmax;
    float* rowlog = workspace + ws.rowlog;
    float* collog = workspace + ws.collog;

    // bf16 modes: fragment planes + LDS-DMA tile kernel; exact-f32 mode: the f32-MFMA tile kernel (true maxima, libm)
    // two_pass (eager form, bf16 modes): statistics pass without a store, then a second tile pass that writes every confidence ONCE
    // (sim_frag_kernel<NS, 3>) -- no S store, no read-modify-write pass over the matrix (ophip_coarse_two_pass())
    const bool two_pass = !lazy && nsplit != 0 && ophip_coarse_two_pass(B, N, M);
    const int sel_nspan = (lazy || two_pass) ? ntc : nspan;
    SimFragArgs sf{};
    const int per_xcd = ((ntr + 7) / 8) * ntc;
    const bool do_sim = (parts & 5) != 0, do_conf = (parts & 9) != 0;
    if ((do_sim || do_conf) && nsplit != 0) {
        char *fa_, *fb_;
        frag_plane_ptrs(workspace, B, N, M, &fa_, &fb_);
        if (!planes_ready && do_sim) {
            FragArgs fr{feat3d, feat2d, fa_, fb_, N, M, 4 * ntr, 4 * ntc};
            OPHIP_LAUNCH("frag_planes", stream, frag_planes_kernel, dim3(16 * (ntr + ntc), B), dim3(256), 0, stream, fr);
            OPHIP_CHECK_LAUNCH();
        }
        sf = SimFragArgs{fa_, fb_, conf, rowpart, colpart, N, M, ntr, ntc, (float)(temperature + temp_eps), ophip_stamp_buffer(),
                         rowstat, colstat, rowlog, collog, rowbest, reinterpret_cast<unsigned*>(colmax), thr, logf(thr) - 1e-3f, qmask};
#define OPHIP_SIM_CASE(NS_, MODE_, NAME_)                                                                                          \
        {                                                                                                                          \
            if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(sim_frag_kernel<NS_, MODE_>), SIM_FRAG_LDS, "hipFuncSetAttribute(sim_frag)")) return rc; \
            OPHIP_LAUNCH(NAME_, stream, (sim_frag_kernel<NS_, MODE_>), dim3(8 * per_xcd, B), dim3(256), SIM_FRAG_LDS, stream, sf);    \
        }
        // Default: sim_frag_kernel (two workgroups per CU, four operand buffers).  OPHIP_SIM_TILE=3: sim_frag3_kernel (three per CU).  Measured in
        // round 5: the three-per-CU form moves a tile ~10 % faster (stage kernels of BASELINE config 4: 422 against 441 us,
        // 523 against 563 us) and LOSES in the pipeline -- config 2: -1 %, config 4: 319 against 394 frames/s -- because three of its
        // workgroups take 154 of a CU's 160 KB of LDS and 480 of 512 registers per lane: the next frame's input kernels (the fine map's
        // transpose needs 17 KB of LDS), which used to run BESIDE the similarity tiles, now wait for them.
        const char* tile_env = getenv("OPHIP_SIM_TILE");          // (read per call: tests compare the two tile kernels in one process)
        const bool tile3 = tile_env && tile_env[0] == '3';
#define OPHIP_SIM3_CASE(NS_, MODE_)                                                                                                \
        {                                                                                                                          \
            if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(sim_frag3_kernel<NS_, MODE_>), SIM_F3_LDS, "hipFuncSetAttribute(sim_frag3)")) return rc; \
            OPHIP_LAUNCH("sim_stats", stream, (sim_frag3_kernel<NS_, MODE_>), dim3(8 * per_xcd, B), dim3(256), SIM_F3_LDS, stream, sf);  \
        }
        if (!do_sim) {}
        else if (tile3 && (lazy || two_pass)) { if (nsplit == 3) OPHIP_SIM3_CASE(3, 1) else OPHIP_SIM3_CASE(1, 1) }
        else if (tile3) { if (nsplit == 3) OPHIP_SIM3_CASE(3, 0) else OPHIP_SIM3_CASE(1, 0) }
        else if (lazy || two_pass) { if (nsplit == 3) OPHIP_SIM_CASE(3, 1, "sim_stats") else OPHIP_SIM_CASE(1, 1, "sim_stats") }
        else { if (nsplit == 3) OPHIP_SIM_CASE(3, 0, "sim_stats") else OPHIP_SIM_CASE(1, 0, "sim_stats") }
#undef OPHIP_SIM3_CASE
        OPHIP_CHECK_LAUNCH();
    } else if (do_sim) {
        SimArgs sa{feat3d, feat2d, conf, rowpart, colpart, N, M, ntr, ntc, (float)(temperature + temp_eps), ophip_stamp_buffer(), qmask};
        // dynamic LDS = max(operand tiles, S staging image of the epilogue)
        const size_t tiles = (size_t)(TM + TN) * LDT * sizeof(float);
        const size_t lds = tiles > SIM_STAGE_BYTES ? tiles : SIM_STAGE_BYTES;
        if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(sim_stats_kernel), lds, "hipFuncSetAttribute(sim_stats)")) return rc;
        OPHIP_LAUNCH("sim_stats", stream, sim_stats_kernel, dim3(ntc, ntr, B), dim3(256), lds, stream, sa);
        OPHIP_CHECK_LAUNCH();
    }
    if (do_conf) {
        const bool logs = lazy || two_pass;
        CombineArgs ca{rowpart, colpart, rowstat, colstat, reinterpret_cast<unsigned*>(colmax), N, M, ntr, ntc, logs ? rowlog : nullptr, logs ? collog : nullptr, lazy ? count + 1 : nullptr};
        OPHIP_LAUNCH("stat_combine", stream, stat_combine_kernel, dim3((N + M + 31) / 32, B), dim3(256), 0, stream, ca);
        OPHIP_CHECK_LAUNCH();
    }
    if (do_conf && lazy) {
        // second look at every tile: confidences from the merged statistics, only the candidates leave the chip
        if (nsplit == 3) OPHIP_SIM_CASE(3, 2, "sim_cand") else OPHIP_SIM_CASE(1, 2, "sim_cand")
        OPHIP_CHECK_LAUNCH();
    }
    if (do_conf && two_pass) {
        // the same second look, and every confidence is stored: conf_matrix written once, nothing read back
        if (nsplit == 3) OPHIP_SIM_CASE(3, 3, "sim_conf") else OPHIP_SIM_CASE(1, 3, "sim_conf")
        OPHIP_CHECK_LAUNCH();
    }
#undef OPHIP_SIM_CASE
    if (do_conf && !lazy && !two_pass) {
        ConfArgs fa{conf, rowstat, colstat, rowbest, reinterpret_cast<unsigned*>(colmax), N, M, nspan, spanw, nrb, thr, conf_rows};
        const size_t clds = (size_t)14 * conf_rows * sizeof(float);
        const bool vec = M % 4 == 0, fast = nsplit != 0;
        if (vec && fast) OPHIP_LAUNCH("conf", stream, (conf_kernel<true, true>), dim3(nspan, nrb, B), dim3(256), clds, stream, fa);
        else if (vec) OPHIP_LAUNCH("conf", stream, (conf_kernel<true, false>), dim3(nspan, nrb, B), dim3(256), clds, stream, fa);
        else if (fast) OPHIP_LAUNCH("conf", stream, (conf_kernel<false, true>), dim3(nspan, nrb, B), dim3(256), clds, stream, fa);
        else OPHIP_LAUNCH("conf", stream, (conf_kernel<false, false>), dim3(nspan, nrb, B), dim3(256), clds, stream, fa);
        OPHIP_CHECK_LAUNCH();
    }
    if (parts & 2) {
        SelectArgs se{conf, rowbest, colmax, keypoints3d, kpts_bstride, B, N, M, sel_nspan, wc, border_rm, border_mode, wi, thr, scale, qscale,
                      b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count};      // (conf == NULL: an exact row tie sets count[1])
        int* dec = reinterpret_cast<int*>(workspace + ws.seldec);
        int* wgc = reinterpret_cast<int*>(workspace + ws.selcnt);
        const int nwg = (int)(((long long)B * N + SEL_T - 1) / SEL_T);
        OPHIP_LAUNCH("select", stream, select_decide_kernel, dim3(nwg), dim3(SEL_T), 0, stream, se, dec, wgc);
        OPHIP_CHECK_LAUNCH();
        OPHIP_LAUNCH("select_place", stream, select_place_kernel, dim3(nwg), dim3(SEL_T), 0, stream, se, dec, wgc);
        OPHIP_CHECK_LAUNCH();
    }
    return 0;
}
}  // namespace

extern "C" int ophip_coarse_frag_planes(float* workspace, int B, int N, int M, void** planes3d, void** planes2d) {
    if (!workspace || !planes3d || !planes2d || B < 1 || N < 1 || M < 1) return ophip_bad_arg(__func__, "bad argument");
    char *a, *b;
    frag_plane_ptrs(workspace, B, N, M, &a, &b);
    *planes3d = a; *planes2d = b;
    return 0;
}

extern "C" int ophip_coarse_match(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                  int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                  float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                  float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                  int* count, int nsplit, void* stream) {
    return coarse_impl(3, 0, 0, 1e-4, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
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
    return coarse_impl(1, 0, 0, 1e-4, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream);
}

extern "C" int ophip_coarse_match_select(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                         int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                         float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                         float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                         int* count, int nsplit, void* stream) {
    return coarse_impl(2, 0, 0, 1e-4, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream);
}

// ophip_coarse_match with the reference's optional inputs of padded / resized query images: query_mask [B][M] (1 = real cell, 0 = padding;
// data["query_image_mask"].flatten(-2), coarse_matching.py:108-114: -1e9 is added to the padded cells' columns of the similarity, so
// their confidences are exactly 0) and query_scale [B][2] = data["query_image_scale"] ((h, w) factors, coarse_matching.py:224: mkpts_query_c
// = (x, y) * scale * query_scale[b][[1, 0]]).  Either may be NULL.  parts (bit mask): 3 = the whole stage, 1 / 2 = the _conf / _select halves, 4 / 8 = the two halves of 1 (similarity
// tiles with their statistics | statistics merge + confidence pass), see coarse_impl.
extern "C" int ophip_coarse_match_masked(const float* feat3d, const float* feat2d, const float* keypoints3d, long long kpts_bstride,
                                         int B, int N, int M, int wc, double temperature, float thr, int border_rm, float scale,
                                         float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                         float* mconf, float* mkpts3d, float* mkpts_c, long long* m_bids, unsigned char* gt_mask,
                                         int* count, int nsplit, int parts, const unsigned char* query_mask, const float* query_scale, void* stream) {
    if (parts < 1 || parts > 15) return ophip_bad_arg(__func__, "parts: bit mask of 1 (= 4 | 8: similarity | confidence), 2 (select)");
    return coarse_impl(parts, 0, 0, 1e-4, feat3d, feat2d, keypoints3d, kpts_bstride, B, N, M, wc, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts3d, mkpts_c, m_bids, gt_mask, count, nsplit, stream, query_mask, query_scale);
}

// LoFTR's 2D-2D coarse matching (loftr/utils/coarse_matching.py of the un-vendored submodule, called at
// src/KeypointFreeSfM/loftr_for_sfm/loftr.py:74): the same dual-softmax + mutual-nearest stage between the coarse grids of TWO
// images -- sim = <f0, f1> / C / temperature (no 1e-4 added), border removal on all four sides of both grids.
// feat0 [B][L0][256], feat1 [B][L1][256], L0 = h0c * w0c, L1 = h1c * w1c.  points0 [B][L0][3]: a caller-made table whose row i is
// gathered into mkpts0 for every match (the detector passes (x, y, 0) of cell i times the image / grid scale = mkpts0_c); mkpts1_c
// is computed as (j % w1c, j / w1c) * scale.  Other arguments as ophip_coarse_match (conf may be NULL: lazy form).
extern "C" int ophip_coarse_match_2d(const float* feat0, const float* feat1, const float* points0, long long points_bstride,
                                     int B, int L0, int L1, int w0c, int w1c, double temperature, float thr, int border_rm, float scale,
                                     float* conf, float* workspace, long long* b_ids, long long* i_ids, long long* j_ids,
                                     float* mconf, float* mkpts0, float* mkpts1_c, long long* m_bids, unsigned char* gt_mask,
                                     int* count, int nsplit, void* stream) {
    return coarse_impl(3, 1, w0c, 0.0, feat0, feat1, points0, points_bstride, B, L0, L1, w1c, temperature, thr, border_rm, scale, conf, workspace,
                       b_ids, i_ids, j_ids, mconf, mkpts0, mkpts1_c, m_bids, gt_mask, count, nsplit, stream);
}
