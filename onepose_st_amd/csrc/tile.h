// Shared device-side building blocks for the gfx950 (CDNA4) kernels of the 2D-3D matcher.
//
// Every contraction in the hot path is a chain of "32 tokens x K  times  K x 32 features" tiles
// on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (64-lane wavefront, f32 in, f32
// accumulate, bit-for-bit an fmaf chain -- MI355X guide, "FP32-input MFMA").  Lane maps used
// below (lane l, r = l & 31, h = l >> 5):
//     A operand:  A[i = r][k = h]            B operand:  B[k = h][j = r]
//     C/D:        D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r],  reg = 0..15
//
// A "fragment" is 16 bytes per lane = 4 consecutive k values of one row, covering 8 k per
// fragment pair (h = 0 takes k 0..3, h = 1 takes k 4..7 of the 8-block); fragment element j feeds
// the j-th MFMA of the block.  With this choice
//   * activations are read from a row-major [token][feature] LDS image with one ds_read_b128,
//   * weights W[out][in] are pre-packed on the host into fragment order
//       [out_tile = out/32][kb = in/8][lane = 32*h + (out%32)][j] = W[out][8*kb + 4*h + j]
//     so that a wave's load of one fragment is 1 KiB contiguous (global_load_dwordx4),
//   * an accumulator tile *is already* a fragment stack for a product that contracts over its
//     row index (regs 4s..4s+3 of a lane are rows 8s + 4h + j): used for K^T V and for using
//     KV as the B operand of phi(Q) KV without touching LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OPHIP_WAVE 64
#define OPHIP_TOK 32            // tokens (rows) per workgroup tile
// Occupancy hint: these kernels are LDS-limited to N waves per SIMD; telling the compiler stops its max-occupancy
// scheduler from trading the software prefetch (live registers) for occupancy the launch can never have.
#define OPHIP_WAVES_PER_SIMD(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#define OPHIP_PAD 4             // LDS row padding in floats (16 B): ds_read_b128 conflict-free

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row of accumulator register `reg` for lane-half h
__device__ __forceinline__ constexpr int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma4(const f32x4 a, const f32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

// acc[t] += A(32 x 8*kblocks, from LDS) * Wpacked(tile t)          (one wave)
//   lds_a   : this lane's row base inside the LDS image: &img[(lane&31)*lda + 4*(lane>>5)]
//   wfrag   : packed weights, pointing at [first tile of this wave][kb0][lane]
//   tstride : distance between consecutive output tiles in f32x4 units (= KB_total * 64)
// Weight fragments stream L2 -> VGPR through a register ring PD k-blocks deep (compile-time slots) so that the L2
// round trip hides behind the MFMAs even at one wave per SIMD; kblocks must be a multiple of PD.
template <int NT, int PD = 4>
__device__ __forceinline__ void gemm_lds_x_packed(f32x16 (&acc)[NT], const float* lds_a, int kblocks,
                                                  const f32x4* __restrict__ wfrag, int tstride) {
    f32x4 ring[PD][NT];
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[p][t] = wfrag[(size_t)t * tstride + (size_t)p * 64];
    for (int kb0 = 0; kb0 < kblocks; kb0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int kb = kb0 + p;
            const f32x4 a = *reinterpret_cast<const f32x4*>(lds_a + 8 * kb);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(a, ring[p][t], acc[t]);
            const int kn = (kb + PD < kblocks) ? kb + PD : kblocks - 1;      // tail re-reads the last block (harmless)
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[p][t] = wfrag[(size_t)t * tstride + (size_t)kn * 64];
            __builtin_amdgcn_sched_barrier(0);      // keep the refill here: the scheduler otherwise sinks it next to its use
        }
    }
}

// write an accumulator tile into a row-major LDS image: img[row][col0 + (lane&31)]
__device__ __forceinline__ void acc_to_lds(const f32x16& acc, float* img, int ld, int col0, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) img[acc_row(reg, h) * ld + col0 + r] = acc[reg];
}

// mkpts_query_f = mkpts_query_c + expectation * (W // 2) * scale_f [* query_image_scale[b][[1, 0]]]   (fine_matching.py:104-105).
// fine_scale = (W // 2) * scale_f with W // 2 = 2 (5 x 5 windows).  With a per-image scale the reference multiplies scale_f by it
// first (f32), then the doubled expectation by that product, then adds: the same three roundings here (no contraction).
template <typename Args>
__device__ __forceinline__ void store_fine_keypoint(const Args& p, int k, float ex, float ey) {
    if (!p.qscale) {
        p.mkq_f[2 * k] = p.mkq_c[2 * k] + ex * p.fine_scale;
        p.mkq_f[2 * k + 1] = p.mkq_c[2 * k + 1] + ey * p.fine_scale;
        return;
    }
    const int b = (int)p.b_ids[k];
    const float sf = p.fine_scale * 0.5f;
    p.mkq_f[2 * k] = __fadd_rn(p.mkq_c[2 * k], __fmul_rn(ex * 2.0f, __fmul_rn(sf, p.qscale[2 * b + 1])));
    p.mkq_f[2 * k + 1] = __fadd_rn(p.mkq_c[2 * k + 1], __fmul_rn(ey * 2.0f, __fmul_rn(sf, p.qscale[2 * b])));
}

// phi(x) = elu(x) + 1 with the reference's arithmetic: (exp(x) - 1) + 1 on the negative side
// (torch.nn.functional.elu(x) + 1, linear_attention.py:10-11)
__device__ __forceinline__ float elu_plus_one(float x) { return x > 0.f ? x + 1.0f : (expf(x) - 1.0f) + 1.0f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// vector-ALU cross-lane moves (DPP / permlane swaps: no LDS crossbar round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// over the 32 lanes that share lane >> 5, result in all of them: xor 1, 2 (quad permutes), 4 (half mirror of uniform quads),
// 8 (mirror of uniform halves), 16 (v_permlane16_swap)
__device__ __forceinline__ float half_sum_dpp(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    return __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    u = __builtin_bit_cast(unsigned, v);
    auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned c0 = c[0], c1 = c[1];
    return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}
__device__ __forceinline__ float swap32_sum(float v) {              // v(lane) + v(lane ^ 32)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned c0 = c[0], c1 = c[1];
    return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}

__device__ __forceinline__ float wave_sum_dpp(float v) {               // sum over the 64 lanes, result in all of them (same moves as wave_max_dpp)
    v = half_sum_dpp(v);
    return swap32_sum(v);
}

// Row-wise LayerNorm over an LDS image [32][C] (row stride ld).  Wave w of 4 normalises rows
// 8w..8w+7; two-pass (mean, then centred second moment), biased variance, eps inside the sqrt.
// AFFINE: y = xhat * gamma + beta.  RELU: max(y, 0) (the keypoint encoder's IN + ReLU).
template <int C, bool AFFINE, bool RELU>
__device__ __forceinline__ void rows_layernorm(float* img, int ld, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, int wave, int lane) {
    constexpr int PER = (C + 63) / 64;       // elements per lane
    for (int rr = 0; rr < 8; ++rr) {
        float* row = img + (8 * wave + rr) * ld;
        float v[PER];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int c = lane + 64 * e;
            v[e] = (c < C) ? row[c] : 0.f;
            s += v[e];
        }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int c = lane + 64 * e;
            const float d = (c < C) ? v[e] - mean : 0.f;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int c = lane + 64 * e;
            if (c < C) {
                float y = (v[e] - mean) * rstd;
                if (AFFINE) y = y * gamma[c] + beta[c];
                if (RELU) y = fmaxf(y, 0.f);
                row[c] = y;
            }
        }
    }
}

#define OPHIP_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return ophip_fail(e__, __func__); \
    } while (0)

int ophip_fail(hipError_t e, const char* where);
int ophip_bad_arg(const char* where, const char* what);
// raise a kernel's dynamic-LDS limit (cached per kernel and device)
int ophip_lds_attr(const void* fn, size_t bytes, const char* what);

// Optional per-kernel HIP-event timing (bench.py's roofline leg): when NAME is the kernel selected with ophip_timing_select(), the
// launch goes through hipExtLaunchKernelGGL with a start / stop event pair, which the runtime fills with the dispatch's OWN begin
// and end timestamps -- the kernel's duration as rocprofv3 reports it, with no extra barrier packets on the stream (an
// hipEventRecord pair around the launch read 7 us more than the trace at c2 and cost the stream 2-3 us per pair).
bool ophip_timed_events(const char* name, hipEvent_t* start, hipEvent_t* stop);

// Tracing hook ("roctx ranges around each custom op"; the reference's counterpart is profiler.record_function,
// coarse_matching.py:122,167): with ophip_roctx_enable(1) -- or OPHIP_ROCTX=1 in the environment -- every kernel launch and every stage of
// ophip_frame_enqueue sits in a roctx range named like the launch (rocprofv3 --marker-trace shows them beside the kernel trace).  Off: one
// predictable branch per launch.  libroctx64 is loaded with dlopen, so the library has no link-time dependency on the profiler.
void ophip_range_push(const char* name);
void ophip_range_pop();
struct OphipRange {
    explicit OphipRange(const char* name) { ophip_range_push(name); }
    ~OphipRange() { ophip_range_pop(); }
    OphipRange(const OphipRange&) = delete;
    OphipRange& operator=(const OphipRange&) = delete;
};

// launch + optional event pair; NAME is the string ophip_timing_select() matches
#define OPHIP_LAUNCH(NAME, STREAM, KERNEL, GRID, BLOCK, LDSBYTES, STREAM2, ...)                                  \
    do {                                                                                                         \
        OphipRange range__(NAME);                                                                                \
        hipEvent_t s__ = nullptr, e__ = nullptr;                                                                 \
        if (ophip_timed_events(NAME, &s__, &e__))                                                                \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDSBYTES, STREAM2, s__, e__, 0, __VA_ARGS__);             \
        else                                                                                                     \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDSBYTES, STREAM2, __VA_ARGS__);                             \
    } while (0)

// Diagnostic cycle stamps (NULL in production): ophip_debug_stamps(buf) makes the instrumented kernels record
// s_memtime at phase boundaries, 32 slots per workgroup, written by one lane.  Never read by any kernel.
extern "C" unsigned long long* ophip_stamp_buffer(void);
#define OPHIP_STAMP(buf, wg, slot)                                                        \
    do {                                                                                  \
        if ((buf) && threadIdx.x == 0) (buf)[(size_t)(wg) * 32 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
// the constant 100 MHz counter (s_memrealtime) beside a cycle stamp: delta(s_memtime) / delta(s_memrealtime) x 100 MHz is the
// clock the chip holds inside the kernel (MI355X_MICROARCH.md, "DVFS give-back" item 6)
#define OPHIP_STAMP_REAL(buf, wg, slot)                                                   \
    do {                                                                                  \
        if ((buf) && threadIdx.x == 0) (buf)[(size_t)(wg) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
