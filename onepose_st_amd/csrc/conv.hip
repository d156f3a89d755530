// ResNet-FPN 8->2 image backbone (reference backbone/resnet.py:20-44 BasicBlock, :85-164 ResNetFPN_8_2)
// as implicit-GEMM convolutions on the bf16 matrix pipe, plain or split-bf16 (tile_bf16.h), f32 accumulate.
//
// Data layout: every intermediate feature map lives in HBM channels-last as two bf16 planes (hi = bf16(x),
// lo = bf16(x - hi)), [B][H][W][Cp] with Cp = channels padded to a multiple of 32 (padding channels are exactly 0), so a
// producer splits each value ONCE and all consumers (9 taps x several workgroups) copy 16-byte chunks straight into the
// LDS operand image.  The maps that leave the backbone (1/8 coarse map, 1/2 fine map, and the FPN maps that are only
// bilinearly upsampled) are written as f32 channels-last -- the layout the encoder / fine kernels read.
//
// GEMM view of one convolution: D[cout][pixel] = sum over (cin chunk, tap, 16-channel k-block) W[cout][k] . X[pixel][k],
// A = packed weights (BatchNorm folded in on the host), B = activations.  A wave owns 64 output channels x 128 output
// pixels (2 x 4 MFMA tiles, 128 accumulator registers): per k-block it pulls 4 KiB of weight fragments through the L2 ->
// VGPR ring and 8 KiB of activation fragments from LDS for 24 MFMAs, which keeps both the vector-memory path (64 B/clk per
// CU) and LDS (128 B/clk) at about a third of their rates with 8 waves per CU.  A workgroup is 2 waves = 128 channels of
// a 32 x 4 pixel tile; its input patch (tile + halo, 32 input channels at a time) is staged in an XOR-swizzled LDS image
// (pixel-major, 64 bytes per pixel and plane) so that the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots.
// Epilogue (all optional, in this order): + bias, + residual map, + bilinear x2 upsampling of a coarser f32 map
// (align_corners=True, resnet.py:155-160), + a per-pixel f32 table (the positional encoding of the coarse map), ReLU /
// LeakyReLU, then planes and / or f32 stores.
#include "tile_bf16.h"
#include <stdlib.h>

namespace {

constexpr int CC = 32;               // input channels staged per chunk (2 k-blocks of 16)
constexpr int PIXB = CC * 2;         // bytes per patch pixel and plane
constexpr int TW = 32;               // output pixels per workgroup tile along x (the MFMA column index); TH rows (template)

struct ConvArgs {
    const __bf16 *in_hi, *in_lo;     // [B][Hin][Win][cin_p]
    int Hin, Win, cin_p, ncc;        // ncc = cin_p / 32
    const bf16x8 *w_hi, *w_lo;       // [ctile][ncc][tap][2][64 lanes]
    const float* bias;               // [cout_p]
    int Hout, Wout, cout_p, ctiles;  // ctiles = cout_p / 32
    int act;                         // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
    const __bf16 *res_hi, *res_lo;   // [B][Hout][Wout][cout_p] or NULL
    const float* up;                 // [B][Hup][Wup][cout_p] f32 or NULL
    int Hup, Wup;
    const float* table;              // [Hout][Wout][cout_p] f32 or NULL (same for every batch element)
    __bf16 *out_hi, *out_lo;         // [B][Hout][Wout][cout_p] or NULL
    float* out_f32;                  // [B][Hout][Wout][out_c] or NULL
    int out_c;
    unsigned long long* stamps;
};

__device__ __forceinline__ int patch_off(int p, int c) { return p * PIXB + ((c ^ ((p >> 2) & 3)) << 4); }

template <int KS, int STRIDE, int NS, int TH, int NT, int WR>
__global__ __launch_bounds__(128 * WR) OPHIP_WAVES_PER_SIMD((TH * NT <= 4 && !(TH == 1 && NT == 2)) ? 3 : 2, (TH * NT >= 8) ? 2 : 3) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int ROWS = TH * WR;                    // output rows of the workgroup tile: WR wave rows of TH
    constexpr int NTHR = 128 * WR;
    constexpr int PW = STRIDE * (TW - 1) + KS, PH = STRIDE * (ROWS - 1) + KS, PIX = PW * PH;
    constexpr int WGT = 2 * NT;                      // 32-channel tiles per workgroup (2 wave columns x NT)
    char* LH = smem;
    char* LL = smem + (NS == 3 ? PIX * PIXB : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wr = wave >> 1;         // wave column (channels) and wave row (pixel rows TH wr .. TH wr + TH - 1)
    const int r = lane & 31, h = lane >> 5;
    const int cgroups = (a.ctiles + WGT - 1) / WGT;
    const int b = blockIdx.z / cgroups, cg = blockIdx.z % cgroups;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * ROWS;
    const int ct0 = WGT * cg + NT * wc;              // this wave's first 32-channel tile
    const int S = a.ncc * T * 2;                     // k-blocks in a tile's weight stream
    // tiles beyond the padded channel count (the last group of a 7-tile layer) are computed on a clamped weight tile and
    // dropped in the epilogue: the k-loop stays one straight-line block (a wave-uniform branch around the MFMAs splits it
    // into basic blocks and costs more than the idle tile)
    const bf16x8 *wh[NT], *wl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ct = min(ct0 + t, a.ctiles - 1);
        wh[t] = a.w_hi + (size_t)ct * S * 64 + lane;
        wl[t] = a.w_lo + (size_t)ct * S * 64 + lane;
    }

    // weight ring: slot = parity of the k-block index; two k-blocks ahead
    bf16x8 rh[2][NT], rl[2][NT];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            rh[p][t] = wh[t][(size_t)p * 64];
            rl[p][t] = (NS == 3) ? wl[t][(size_t)p * 64] : zero_bf8();
        }
    f32x16 acc[NT][TH];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int tt = 0; tt < TH; ++tt) acc[t][tt] = zero16();

    const size_t in_b = (size_t)b * a.Hin * a.Win * a.cin_p;
    const int iy0 = y0 * STRIDE - PAD, ix0 = x0 * STRIDE - PAD;
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    OPHIP_STAMP(a.stamps, wg, 0);
    int s = 0;
    for (int cc = 0; cc < a.ncc; ++cc) {
        __syncthreads();                             // everyone is done reading the previous chunk's image
        if (cc < 8) OPHIP_STAMP(a.stamps, wg, 1 + 3 * cc);
        for (int i = tid; i < PIX * 4; i += NTHR) {
            const int p = i >> 2, c = i & 3;
            const int py = p / PW, px = p - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            bf16x8 vh = zero_bf8(), vl = zero_bf8();
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                const size_t g = in_b + ((size_t)iy * a.Win + ix) * a.cin_p + cc * CC + c * 8;
                vh = *reinterpret_cast<const bf16x8*>(a.in_hi + g);
                if (NS == 3) vl = *reinterpret_cast<const bf16x8*>(a.in_lo + g);
            }
            const int off = patch_off(p, c);
            *reinterpret_cast<bf16x8*>(LH + off) = vh;
            if (NS == 3) *reinterpret_cast<bf16x8*>(LL + off) = vl;
        }
        __syncthreads();
        if (cc < 8) OPHIP_STAMP(a.stamps, wg, 2 + 3 * cc);
        // activation fragments are read one k-block ahead (two register sets): the LDS round trip of step i + 1 runs under the
        // MFMAs of step i
        bf16x8 xh[2][TH], xl[2][TH];
        auto read_x = [&](int step, int set) {
            const int tap = step >> 1, kbl = step & 1;
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int tt = 0; tt < TH; ++tt) {
                const int p = (STRIDE * (TH * wr + tt) + dy) * PW + STRIDE * r + dx;
                const int off = patch_off(p, 2 * kbl + h);
                xh[set][tt] = *reinterpret_cast<const bf16x8*>(LH + off);
                xl[set][tt] = (NS == 3) ? *reinterpret_cast<const bf16x8*>(LL + off) : zero_bf8();
            }
        };
        read_x(0, 0);
#pragma unroll
        for (int step = 0; step < 2 * T; ++step) {
            const int kbl = step & 1, cur = step & 1;
            if (step + 1 < 2 * T) read_x(step + 1, cur ^ 1);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int tt = 0; tt < TH; ++tt) acc[t][tt] = mma_bf16<NS>(rh[kbl][t], rl[kbl][t], xh[cur][tt], xl[cur][tt], acc[t][tt]);
            {
                const size_t nx = (size_t)min(s + 2, S - 1) * 64;        // the last two refills re-read the final k-block (unused)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    rh[kbl][t] = wh[t][nx];
                    if (NS == 3) rl[kbl][t] = wl[t][nx];
                }
            }
            ++s;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cc < 8) OPHIP_STAMP(a.stamps, wg, 3 + 3 * cc);
    }

    OPHIP_STAMP(a.stamps, wg, 30);
    // ---- epilogue, one output row (32 pixels x 64 NT channels... of both waves) at a time: accumulators -> f32 stage in LDS
    //      ([pixel][channel], 16-byte chunks XOR-swizzled by the pixel), then every thread finishes 4 consecutive channels of a
    //      pixel so that the residual / upsampling reads and all stores are whole 256- / 512-byte rows ----
    constexpr int WCH = 32 * WGT;                    // channels of the workgroup tile
    constexpr int SROW = WCH * 4;                    // stage row pitch (bytes)
    constexpr int CH16 = WCH / 4;                    // 16-byte chunks per stage row
    char* stage = smem;
    const int cbase = 32 * WGT * cg;                 // first channel of the workgroup tile
    for (int row = 0; row < ROWS; ++row) {
        const int y = y0 + row;
        __syncthreads();                             // the patch image / the previous row's stage is no longer read
        if (y < a.Hout && row / TH == wr) {          // the wave row that owns this output row
#pragma unroll
            for (int tt = 0; tt < TH; ++tt) {
                if (tt != row % TH) continue;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = (32 * (NT * wc + t) + 8 * g + 4 * h) >> 2;
                        const f32x4 v = {acc[t][tt][4 * g], acc[t][tt][4 * g + 1], acc[t][tt][4 * g + 2], acc[t][tt][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(stage + r * SROW + ((ch ^ (r & 15)) << 4)) = v;
                    }
            }
        }
        __syncthreads();
        if (y >= a.Hout) continue;
        float upy_l = 0.f; int uy0 = 0, uy1 = 0;
        if (a.up) {
            const float sy = a.Hout > 1 ? (float)(a.Hup - 1) / (float)(a.Hout - 1) : 0.f;
            const float fy = sy * (float)y;
            uy0 = min((int)fy, a.Hup - 1);
            uy1 = uy0 + (uy0 < a.Hup - 1 ? 1 : 0);
            upy_l = fminf(fmaxf(fy - (float)uy0, 0.f), 1.f);
        }
        for (int i = tid; i < 32 * CH16; i += NTHR) {
            const int pr = i / CH16, q = i - pr * CH16;
            const int x = x0 + pr, c0 = cbase + 4 * q;
            if (x >= a.Wout || c0 >= a.cout_p) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + pr * SROW + ((q ^ (pr & 15)) << 4));
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bv[j];
            const size_t pix = ((size_t)b * a.Hout + y) * a.Wout + x;
            if (a.res_hi) {
                const bf16x4 rhi = *reinterpret_cast<const bf16x4*>(a.res_hi + pix * a.cout_p + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rhi[j];
                if (NS == 3) {
                    const bf16x4 rlo = *reinterpret_cast<const bf16x4*>(a.res_lo + pix * a.cout_p + c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rlo[j];
                }
            }
            if (a.up) {
                const float sx = a.Wout > 1 ? (float)(a.Wup - 1) / (float)(a.Wout - 1) : 0.f;
                const float fx = sx * (float)x;
                const int ux0 = min((int)fx, a.Wup - 1);
                const int ux1 = ux0 + (ux0 < a.Wup - 1 ? 1 : 0);
                const float lx = fminf(fmaxf(fx - (float)ux0, 0.f), 1.f);
                const size_t ub = (size_t)b * a.Hup;
                const f32x4 v00 = *reinterpret_cast<const f32x4*>(a.up + ((ub + uy0) * a.Wup + ux0) * a.cout_p + c0);
                const f32x4 v01 = *reinterpret_cast<const f32x4*>(a.up + ((ub + uy0) * a.Wup + ux1) * a.cout_p + c0);
                const f32x4 v10 = *reinterpret_cast<const f32x4*>(a.up + ((ub + uy1) * a.Wup + ux0) * a.cout_p + c0);
                const f32x4 v11 = *reinterpret_cast<const f32x4*>(a.up + ((ub + uy1) * a.Wup + ux1) * a.cout_p + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] += (1.f - upy_l) * ((1.f - lx) * v00[j] + lx * v01[j]) + upy_l * ((1.f - lx) * v10[j] + lx * v11[j]);
            }
            if (a.table) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(a.table + ((size_t)y * a.Wout + x) * a.cout_p + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += tv[j];
            }
            if (a.act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (a.act == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.01f * v[j];
            }
            if (a.out_hi) {
                bf16x4 vh, vl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __bf16 hh, ll;
                    split_bf16(v[j], hh, ll);
                    vh[j] = hh; vl[j] = ll;
                }
                *reinterpret_cast<bf16x4*>(a.out_hi + pix * a.cout_p + c0) = vh;
                if (NS == 3) *reinterpret_cast<bf16x4*>(a.out_lo + pix * a.cout_p + c0) = vl;
            }
            if (a.out_f32 && c0 < a.out_c) *reinterpret_cast<f32x4*>(a.out_f32 + pix * a.out_c + c0) = v;
        }
    }
    OPHIP_STAMP(a.stamps, wg, 31);
}

// ---------------------------------------------------------------------------------------------
// stem: 7x7 stride-2 convolution of the 1-channel image + folded BatchNorm + ReLU (resnet.py:100-102,140), exact f32 on
// the vector ALU (0.3 % of the backbone's FLOPs; K = 49 is no MFMA shape).  A thread owns one output pixel: its 49
// taps sit in registers, the folded weights are read from LDS as broadcasts, 8 channels at a time.
// ---------------------------------------------------------------------------------------------
struct StemArgs {
    const float* img;                // [B][H][W]
    int H, W, Hout, Wout;
    const float* w;                  // [49][128] folded weights, then bias[128]
    __bf16 *out_hi, *out_lo;         // [B][Hout][Wout][128]
    int nsplit;
};

constexpr int SW = 2 * (TW - 1) + 7, SH = 2 * (8 - 1) + 7;       // 69 x 21 input patch of a 32 x 8 output tile

__global__ __launch_bounds__(256) OPHIP_WAVES_PER_SIMD(3, 4) void stem_kernel(StemArgs a) {
    __shared__ float patch[SH][SW + 1];
    __shared__ __attribute__((aligned(16))) float wl[49 * 128 + 128];      // folded weights [tap][channel] + bias: broadcast reads
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * 8, b = blockIdx.z;
    const float* img = a.img + (size_t)b * a.H * a.W;
    for (int i = tid; i < SH * SW; i += 256) {
        const int py = i / SW, px = i - py * SW;
        const int iy = 2 * y0 - 3 + py, ix = 2 * x0 - 3 + px;
        patch[py][px] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? img[(size_t)iy * a.W + ix] : 0.f;
    }
    for (int i = tid; i < (49 * 128 + 128) / 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(a.w)[i];
    __syncthreads();
    float v[49];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) v[ky * 7 + kx] = patch[2 * ty + ky][2 * tx + kx];
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.Wout || y >= a.Hout) return;
    const size_t pix = ((size_t)b * a.Hout + y) * a.Wout + x;
    for (int cg = 0; cg < 16; ++cg) {                // 8 channels at a time: 49 taps + 8 accumulators keep 3 waves per SIMD resident
        f32x4 acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[q] = *reinterpret_cast<const f32x4*>(wl + 49 * 128 + 8 * cg + 4 * q);
#pragma unroll
        for (int t = 0; t < 49; ++t) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + t * 128 + 8 * cg + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q][j] = fmaf(w4[j], v[t], acc[q][j]);
            }
        }
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 hh, ll;
            split_bf16(fmaxf(acc[j >> 2][j & 3], 0.f), hh, ll);
            vh[j] = hh; vl[j] = ll;
        }
        *reinterpret_cast<bf16x8*>(a.out_hi + pix * 128 + 8 * cg) = vh;
        if (a.nsplit == 3) *reinterpret_cast<bf16x8*>(a.out_lo + pix * 128 + 8 * cg) = vl;
    }
}

template <int KS, int STRIDE, int NS, int TH, int NT, int WR>
int launch_conv_tile(const ConvArgs& a, int B, hipStream_t stream) {
    constexpr int PW = STRIDE * (TW - 1) + KS, PH = STRIDE * (TH * WR - 1) + KS;
    size_t lds = (size_t)(NS == 3 ? 2 : 1) * PW * PH * PIXB;
    const size_t stage = (size_t)32 * 64 * NT * 4;           // one output row of the workgroup tile in f32
    if (stage > lds) lds = stage;
    if (int rc = ophip_lds_attr(reinterpret_cast<const void*>(conv_mfma_kernel<KS, STRIDE, NS, TH, NT, WR>), lds, "hipFuncSetAttribute(conv_mfma)")) return rc;
    const dim3 grid((a.Wout + TW - 1) / TW, (a.Hout + TH * WR - 1) / (TH * WR), B * ((a.ctiles + 2 * NT - 1) / (2 * NT)));
    OPHIP_LAUNCH("conv", stream, (conv_mfma_kernel<KS, STRIDE, NS, TH, NT, WR>), grid, dim3(128 * WR), lds, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}

// Wave tile = 32 NT channels x 32 TH pixels (NT x TH MFMA tiles).  Larger tiles reuse each weight fragment TH times and each
// activation fragment NT times (less L2 / vector-memory / LDS traffic per FLOP) but give fewer, register-heavier waves.
// Measured per layer at 480 x 640 (tools/bench_backbone_hip.py, B = 1 and 4): (2, 2) wins whenever it yields >= 2 waves
// per SIMD; below that the 1/4-resolution maps take (2, 1), the 1/8-resolution maps (1, 1); stride-2 convolutions (large
// input patch per tile) prefer to stay at 2 channel tiles per wave; (4, x) never won and is not built.
template <int KS, int STRIDE, int NS>
int launch_conv(const ConvArgs& a, int B, hipStream_t stream) {
    const long w22 = (long)((a.Wout + TW - 1) / TW) * ((a.Hout + 1) / 2) * B * ((a.ctiles + 3) / 4) * 2;      // waves of the (2, 2) shape
    int th = 2, nt = 2;
    if (KS == 3 && STRIDE == 1) {
        if (w22 < 2000) nt = 1;
        if (2 * w22 < 2000) th = 1;
    } else if (KS == 3) {
        if (w22 < 2000) th = 1;
        if (w22 < 1000) nt = 1;
    } else if (w22 < 2000) {
        th = 1;
    }
    // wave rows per workgroup: 2 (a 4-wave workgroup on a shared patch: halo 1.6x instead of 2.1x at TH = 2) when the map is
    // tall enough to keep the grid full
    const int wrows = (KS == 3 && STRIDE == 1 && w22 >= 2000) ? 2 : 1;
#define OPHIP_CONV_CASE(TH_, NT_, WR_) if (th == TH_ && nt == NT_ && wrows == WR_) return launch_conv_tile<KS, STRIDE, NS, TH_, NT_, WR_>(a, B, stream);
    OPHIP_CONV_CASE(2, 2, 1) OPHIP_CONV_CASE(2, 1, 1) OPHIP_CONV_CASE(1, 2, 1) OPHIP_CONV_CASE(1, 1, 1)
    OPHIP_CONV_CASE(2, 2, 2) OPHIP_CONV_CASE(2, 1, 2) OPHIP_CONV_CASE(1, 2, 2) OPHIP_CONV_CASE(1, 1, 2)
#undef OPHIP_CONV_CASE
    return ophip_bad_arg("ophip_conv2d_bf16", "no kernel for this tile shape");
}

}  // namespace

extern "C" size_t ophip_conv_wpack_bytes(int cin_pad, int cout_pad, int ks) {
    return (size_t)2 * cout_pad * cin_pad * ks * ks * 2 + (size_t)cout_pad * 4;       // hi plane | lo plane | bias f32
}

extern "C" int ophip_conv2d_bf16(const void* in_hi, const void* in_lo, int B, int Hin, int Win, int cin_pad,
                                 const void* wpack, int cout_pad, int ks, int stride, int act,
                                 const void* res_hi, const void* res_lo, const float* up, int Hup, int Wup, const float* table,
                                 void* out_hi, void* out_lo, float* out_f32, int out_c, int nsplit, void* stream_) {
    if (!in_hi || !wpack) return ophip_bad_arg(__func__, "null pointer");
    if (nsplit != 1 && nsplit != 3) return ophip_bad_arg(__func__, "nsplit must be 1 (bf16) or 3 (split bf16)");
    if (nsplit == 3 && (!in_lo || (res_hi && !res_lo) || (out_hi && !out_lo))) return ophip_bad_arg(__func__, "split mode needs the lo planes");
    if (B < 1 || Hin < 1 || Win < 1 || cin_pad < 32 || cin_pad % 32 || cout_pad < 32 || cout_pad % 32) return ophip_bad_arg(__func__, "bad sizes (channels padded to 32)");
    if (!((ks == 3 || ks == 1) && (stride == 1 || stride == 2))) return ophip_bad_arg(__func__, "kernel 1 or 3, stride 1 or 2");
    if (act < 0 || act > 2) return ophip_bad_arg(__func__, "act: 0 none, 1 relu, 2 leaky relu");
    if (!out_hi && !out_f32) return ophip_bad_arg(__func__, "no output");
    if (out_f32 && (out_c < 4 || out_c % 4 || out_c > cout_pad)) return ophip_bad_arg(__func__, "out_c must be a multiple of 4 within cout_pad");
    if (up && (Hup < 1 || Wup < 1)) return ophip_bad_arg(__func__, "bad upsampling source size");
    const int pad = ks / 2;
    ConvArgs a;
    a.in_hi = reinterpret_cast<const __bf16*>(in_hi); a.in_lo = reinterpret_cast<const __bf16*>(in_lo);
    a.Hin = Hin; a.Win = Win; a.cin_p = cin_pad; a.ncc = cin_pad / CC;
    const size_t welems = (size_t)cout_pad * cin_pad * ks * ks;
    a.w_hi = reinterpret_cast<const bf16x8*>(wpack);
    a.w_lo = a.w_hi + welems / 8;
    a.bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + 2 * welems * 2);
    a.Hout = (Hin + 2 * pad - ks) / stride + 1; a.Wout = (Win + 2 * pad - ks) / stride + 1;
    a.cout_p = cout_pad; a.ctiles = cout_pad / 32; a.act = act;
    a.res_hi = reinterpret_cast<const __bf16*>(res_hi); a.res_lo = reinterpret_cast<const __bf16*>(res_lo);
    a.up = up; a.Hup = Hup; a.Wup = Wup; a.table = table;
    a.out_hi = reinterpret_cast<__bf16*>(out_hi); a.out_lo = reinterpret_cast<__bf16*>(out_lo);
    a.out_f32 = out_f32; a.out_c = out_c; a.stamps = ophip_stamp_buffer();
    hipStream_t stream = (hipStream_t)stream_;
    if (nsplit == 3) {
        if (ks == 3 && stride == 1) return launch_conv<3, 1, 3>(a, B, stream);
        if (ks == 3 && stride == 2) return launch_conv<3, 2, 3>(a, B, stream);
        if (ks == 1 && stride == 1) return launch_conv<1, 1, 3>(a, B, stream);
        return launch_conv<1, 2, 3>(a, B, stream);
    }
    if (ks == 3 && stride == 1) return launch_conv<3, 1, 1>(a, B, stream);
    if (ks == 3 && stride == 2) return launch_conv<3, 2, 1>(a, B, stream);
    if (ks == 1 && stride == 1) return launch_conv<1, 1, 1>(a, B, stream);
    return launch_conv<1, 2, 1>(a, B, stream);
}

extern "C" int ophip_stem_conv7(const float* image, int B, int H, int W, const float* wpack, void* out_hi, void* out_lo, int nsplit, void* stream_) {
    if (!image || !wpack || !out_hi || (nsplit == 3 && !out_lo)) return ophip_bad_arg(__func__, "null pointer");
    if (B < 1 || H < 1 || W < 1) return ophip_bad_arg(__func__, "bad sizes");
    if (nsplit != 1 && nsplit != 3) return ophip_bad_arg(__func__, "nsplit must be 1 or 3");
    StemArgs a;
    a.img = image; a.H = H; a.W = W; a.Hout = (H + 6 - 7) / 2 + 1; a.Wout = (W + 6 - 7) / 2 + 1;
    a.w = wpack; a.out_hi = reinterpret_cast<__bf16*>(out_hi); a.out_lo = reinterpret_cast<__bf16*>(out_lo); a.nsplit = nsplit;
    hipStream_t stream = (hipStream_t)stream_;
    OPHIP_LAUNCH("stem", stream, stem_kernel, dim3((a.Wout + TW - 1) / TW, (a.Hout + 7) / 8, B), dim3(256), 0, stream, a);
    OPHIP_CHECK_LAUNCH();
    return 0;
}
