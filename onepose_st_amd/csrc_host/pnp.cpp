// Deterministic host PnP + RANSAC (double precision, no dependencies).  C ABI: include/onepose_pnp.h.
//
// Replaces the call the reference makes after the matcher -- src/utils/metric_utils.py:121-209 `ransac_PnP`, which
// delegates to pycolmap.absolute_pose_estimation (P3P LO-RANSAC + non-linear refinement, max_error_px 7 in
// inference.py:181-189) or cv2.solvePnPRansac (EPnP).  Neither library exists here and the reference pins no pose
// outputs, so this is NOT a restatement of their internals ("parity unpinned"): it is the build's own estimator,
// applied identically to the HIP path's matches and to the oracle's matches so that pose parity is checkable.
//
//   hypotheses : solver 1 (the pycolmap branch of ransac_PnP, metric_utils.py:155-165: COLMAP's estimator is P3P): three
//                correspondences -> Grunert's quartic in the depth ratio s3 / s1 (coefficients by polynomial products of the two
//                eliminated law-of-cosines equations, Ferrari + Newton polish) -> up to four poses by aligning the two point
//                triangles, every root scored on all points like COLMAP does; works on coplanar objects (box faces, cards);
//                solver 0 (the OpenCV branch, :188-196): 6-point DLT on calibrated rays: the normal matrix's block structure
//                reduces it to the smallest eigenvector of a 4x4 Schur complement (inverse iteration), projected to SO(3) by
//                Newton polar iteration, four samples at a time, one per lane of a 4 x double vector; a sample whose DLT is
//                singular (coplanar 3D points) is solved by P3P on its first three points, the root picked by the other three
//   scoring    : reprojection error < threshold (pixels) on a float structure-of-arrays copy (AVX2 + FMA intrinsics in the
//                build pnp.py prefers, clones of one plain loop otherwise; hypotheses that cannot win any more dropped block by
//                block, the first block as long as the misses a hypothesis may afford); a candidate best is re-scored in
//                double, which decides; at least min_iters trials (the
//                reference's pycolmap call runs >= 10 000), then adaptive stopping at the requested confidence
//   refinement : Levenberg-Marquardt on the inliers (6 dof, analytic Jacobian), inlier set re-evaluated once
//   randomness : xorshift64* seeded by the caller -> bit-reproducible
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <semaphore.h>          // (POSIX, not Linux-only: the pool's sem_t is used on every build)
#if defined(__linux__)
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif
#include <vector>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif

namespace {

inline bool env_on(const char* name) {
    const char* v = getenv(name);
    return v && *v && !(v[0] == '0' && !v[1]);
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
    uint64_t next() {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    int below(int n) { return (int)(next() % (uint64_t)n); }
};

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, destroyed); V columns = eigenvectors
template <int N>
void jacobi_eig(double* A, double* V) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j];
        if (off < 1e-30) break;
        for (int p = 0; p < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p * N + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
}

double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

void inv3(const double* m, double* o) {
    const double d = det3(m), id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// nearest rotation of a matrix with positive determinant: Newton iteration R <- (R + R^-T) / 2
bool polar_rotation(double* R, double tol = 1e-15) {
    for (int it = 0; it < 60; ++it) {
        if (std::fabs(det3(R)) < 1e-14) return false;
        double inv[9];
        inv3(R, inv);
        double diff = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double n = 0.5 * (R[i * 3 + j] + inv[j * 3 + i]);
                diff += std::fabs(n - R[i * 3 + j]);
                R[i * 3 + j] = n;
            }
        if (diff < tol) break;
    }
    return true;
}

// eigenvector of the smallest eigenvalue of a symmetric positive semi-definite N x N matrix: inverse iteration on
// A + eps I (Cholesky, three solves from a fixed start).  The DLT matrix of a near-exact minimal sample has one eigenvalue
// ~ 0 far below the rest, so the iteration converges in one or two steps -- what makes the reference's 10 000-trial floor
// affordable (a cyclic Jacobi decomposition of the 12 x 12 normal matrix took ~45 us per hypothesis).
template <int N>
bool smallest_eigvec(const double* A, double trace, double* v) {
    double Lm[N * N];
    const double eps = 1e-13 * (trace > 0 ? trace : 1.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i * N + j] + (i == j ? eps : 0.0);
            for (int k = 0; k < j; ++k) sum -= Lm[i * N + k] * Lm[j * N + k];
            if (i == j) {
                if (!(sum > 0.0)) return false;
                Lm[i * N + i] = std::sqrt(sum);
            } else {
                Lm[i * N + j] = sum / Lm[j * N + j];
            }
        }
    for (int i = 0; i < N; ++i) v[i] = 1.0 + 0.0625 * i;          // fixed, generic start
    for (int it = 0; it < 3; ++it) {
        double y[N];
        for (int i = 0; i < N; ++i) {
            double sum = v[i];
            for (int k = 0; k < i; ++k) sum -= Lm[i * N + k] * y[k];
            y[i] = sum / Lm[i * N + i];
        }
        for (int i = N - 1; i >= 0; --i) {
            double sum = y[i];
            for (int k = i + 1; k < N; ++k) sum -= Lm[k * N + i] * v[k];
            v[i] = sum / Lm[i * N + i];
        }
        double nrm = 0.0;
        for (int i = 0; i < N; ++i) nrm += v[i] * v[i];
        if (!(nrm > 0.0) || !std::isfinite(nrm)) return false;
        nrm = 1.0 / std::sqrt(nrm);
        for (int i = 0; i < N; ++i) v[i] *= nrm;
    }
    return true;
}

// inverse of a symmetric positive definite 4 x 4 matrix (Cholesky); false when not positive definite
bool inv4_spd(const double* A, double* Ai) {
    double Lm[16] = {0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i * 4 + j];
            for (int k = 0; k < j; ++k) sum -= Lm[i * 4 + k] * Lm[j * 4 + k];
            if (i == j) {
                if (!(sum > 1e-300)) return false;
                Lm[i * 4 + i] = std::sqrt(sum);
            } else {
                Lm[i * 4 + j] = sum / Lm[j * 4 + j];
            }
        }
    for (int c = 0; c < 4; ++c) {                      // solve L L^T x = e_c
        double y[4], x[4];
        for (int i = 0; i < 4; ++i) {
            double sum = i == c ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) sum -= Lm[i * 4 + k] * y[k];
            y[i] = sum / Lm[i * 4 + i];
        }
        for (int i = 3; i >= 0; --i) {
            double sum = y[i];
            for (int k = i + 1; k < 4; ++k) sum -= Lm[k * 4 + i] * x[k];
            x[i] = sum / Lm[i * 4 + i];
        }
        for (int i = 0; i < 4; ++i) Ai[i * 4 + c] = x[i];
    }
    return true;
}

// float copy of the correspondences in structure-of-arrays form: the RANSAC loop scores every hypothesis on it with SIMD
// (the winner's inlier mask and count are then re-evaluated in double precision)
struct FastPoints {
    int n = 0;
    std::vector<float> x, y, z, u, v;
};

// inliers and truncated cost of pose (premultiplied by K: 3 x 4 row-major floats) -- the vectorised hot loop of the RANSAC.
// Scored in blocks; after each block the hypothesis is dropped as soon as it can no longer beat the best one (count first, then
// cost: with `rest` points to go it needs cnt + rest >= best_cnt, and on a possible tie a cost below best_cost, while the cost
// only grows).  The decision is exactly the one a full pass would take; returns false when dropped.
//
// Block schedule (round 4, from cycle counts: 2/3 of a reference-policy frame's host time was this function): neither test can fire
// before the hypothesis has MISSED n - best_cnt points, so the first block is exactly that long (at least 16 points): on a clean
// frame (best_cnt = n) a wrong P3P root -- half of all hypotheses -- dies after two vectors instead of a 256-point block, and on a
// frame with 40 % outliers the first thousand points are one loop without the per-block reductions; 256-point blocks after that.
inline int first_block(int n, int best_cnt) {
    int fb = n - best_cnt;
    fb = fb < 16 ? 16 : fb;
    return (fb + 15) & ~15;
}

inline bool keep_scoring(int cnt, float cost, int rest, int best_cnt, float best_cost) {
    const int reach = cnt + rest;                        // the count if every remaining point were an inlier
    if (reach < best_cnt || (reach == best_cnt && !(cost < best_cost))) return false;
    if (cnt <= best_cnt && reach <= best_cnt && !(cost < best_cost)) return false;
    return true;
}

#if defined(__AVX2__) && defined(__FMA__)
// Sixteen points per step on a CPU with AVX-512 (chosen once at load time, OPPNP_NO_AVX512=1 keeps the 256-bit loop): the same
// arithmetic as the loop below, inlier masks in mask registers (counted with a popcount), the last partial vector by a lane mask.
__attribute__((target("avx512f,avx512vl,fma")))
bool score_fast_512(const FastPoints& F, const float* kp, float thr2, int best_cnt, float best_cost, int* cnt_out, float* cost_out) {
    const int n = F.n;
    const float *X = F.x.data(), *Y = F.y.data(), *Z = F.z.data(), *U = F.u.data(), *V = F.v.data();
    const __m512 k0 = _mm512_set1_ps(kp[0]), k1 = _mm512_set1_ps(kp[1]), k2 = _mm512_set1_ps(kp[2]), k3 = _mm512_set1_ps(kp[3]);
    const __m512 k4 = _mm512_set1_ps(kp[4]), k5 = _mm512_set1_ps(kp[5]), k6 = _mm512_set1_ps(kp[6]), k7 = _mm512_set1_ps(kp[7]);
    const __m512 k8 = _mm512_set1_ps(kp[8]), k9 = _mm512_set1_ps(kp[9]), k10 = _mm512_set1_ps(kp[10]), k11 = _mm512_set1_ps(kp[11]);
    const __m512 vthr = _mm512_set1_ps(thr2), veps = _mm512_set1_ps(1e-12f), one = _mm512_set1_ps(1.0f);
    int cnt = 0;
    float cost = 0.f;
    int i1 = 0;
    for (int i0 = 0; i0 < n; i0 = i1) {
        const int len = i0 == 0 ? first_block(n, best_cnt) : 256;
        i1 = i0 + len < n ? i0 + len : n;
        __m512 vs = _mm512_setzero_ps();
        int c_b = 0;
        for (int i = i0; i < i1; i += 16) {
            const __mmask16 live = i + 16 <= i1 ? (__mmask16)0xFFFF : (__mmask16)((1u << (i1 - i)) - 1u);
            const __m512 x = _mm512_maskz_loadu_ps(live, X + i), y = _mm512_maskz_loadu_ps(live, Y + i), z = _mm512_maskz_loadu_ps(live, Z + i);
            const __m512 a = _mm512_fmadd_ps(k0, x, _mm512_fmadd_ps(k1, y, _mm512_fmadd_ps(k2, z, k3)));
            const __m512 b = _mm512_fmadd_ps(k4, x, _mm512_fmadd_ps(k5, y, _mm512_fmadd_ps(k6, z, k7)));
            const __m512 c = _mm512_fmadd_ps(k8, x, _mm512_fmadd_ps(k9, y, _mm512_fmadd_ps(k10, z, k11)));
            const __m512 ic = _mm512_div_ps(one, c);
            const __m512 du = _mm512_fmsub_ps(a, ic, _mm512_maskz_loadu_ps(live, U + i)), dv = _mm512_fmsub_ps(b, ic, _mm512_maskz_loadu_ps(live, V + i));
            const __m512 e2 = _mm512_fmadd_ps(du, du, _mm512_mul_ps(dv, dv));
            const __mmask16 in = _mm512_mask_cmp_ps_mask(_mm512_mask_cmp_ps_mask(live, c, veps, _CMP_GT_OQ), e2, vthr, _CMP_LT_OQ);
            c_b += __builtin_popcount((unsigned)in);
            vs = _mm512_mask_add_ps(vs, live, vs, _mm512_mask_blend_ps(in, vthr, e2));
        }
        cnt += c_b;
        cost += _mm512_reduce_add_ps(vs);
        if (!keep_scoring(cnt, cost, n - i1, best_cnt, best_cost)) return false;
    }
    *cnt_out = cnt;
    *cost_out = cost;
    return true;
}

const bool g_use_512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && !env_on("OPPNP_NO_AVX512");

// The -mavx2 -mfma build (what pnp.py loads on a CPU that has both): eight points per step, 21 vector instructions (the compiler's
// version of the loop below spent 26 plus a dozen broadcasts per block): three-FMA rows, one division, masks summed as integers.
bool score_fast(const FastPoints& F, const float* kp, float thr2, int best_cnt, float best_cost, int* cnt_out, float* cost_out) {
    if (g_use_512) return score_fast_512(F, kp, thr2, best_cnt, best_cost, cnt_out, cost_out);
    const int n = F.n;
    const float *X = F.x.data(), *Y = F.y.data(), *Z = F.z.data(), *U = F.u.data(), *V = F.v.data();
    const __m256 k0 = _mm256_set1_ps(kp[0]), k1 = _mm256_set1_ps(kp[1]), k2 = _mm256_set1_ps(kp[2]), k3 = _mm256_set1_ps(kp[3]);
    const __m256 k4 = _mm256_set1_ps(kp[4]), k5 = _mm256_set1_ps(kp[5]), k6 = _mm256_set1_ps(kp[6]), k7 = _mm256_set1_ps(kp[7]);
    const __m256 k8 = _mm256_set1_ps(kp[8]), k9 = _mm256_set1_ps(kp[9]), k10 = _mm256_set1_ps(kp[10]), k11 = _mm256_set1_ps(kp[11]);
    const __m256 vthr = _mm256_set1_ps(thr2), veps = _mm256_set1_ps(1e-12f), one = _mm256_set1_ps(1.0f);
    int cnt = 0;
    float cost = 0.f;
    int i1 = 0;
    for (int i0 = 0; i0 < n; i0 = i1) {
        const int len = i0 == 0 ? first_block(n, best_cnt) : 256;
        i1 = i0 + len < n ? i0 + len : n;
        __m256i vc = _mm256_setzero_si256();
        __m256 vs = _mm256_setzero_ps();
        auto step = [&](int i, __m256i live) {                       // live: all ones, or the lanes of a last partial vector
            const __m256 x = _mm256_maskload_ps(X + i, live), y = _mm256_maskload_ps(Y + i, live), z = _mm256_maskload_ps(Z + i, live);
            const __m256 a = _mm256_fmadd_ps(k0, x, _mm256_fmadd_ps(k1, y, _mm256_fmadd_ps(k2, z, k3)));
            const __m256 b = _mm256_fmadd_ps(k4, x, _mm256_fmadd_ps(k5, y, _mm256_fmadd_ps(k6, z, k7)));
            const __m256 c = _mm256_fmadd_ps(k8, x, _mm256_fmadd_ps(k9, y, _mm256_fmadd_ps(k10, z, k11)));
            const __m256 ic = _mm256_div_ps(one, c);
            const __m256 du = _mm256_fmsub_ps(a, ic, _mm256_maskload_ps(U + i, live)), dv = _mm256_fmsub_ps(b, ic, _mm256_maskload_ps(V + i, live));
            const __m256 e2 = _mm256_fmadd_ps(du, du, _mm256_mul_ps(dv, dv));
            const __m256 in = _mm256_and_ps(_mm256_cmp_ps(c, veps, _CMP_GT_OQ), _mm256_cmp_ps(e2, vthr, _CMP_LT_OQ));
            vc = _mm256_sub_epi32(vc, _mm256_and_si256(_mm256_castps_si256(in), live));
            vs = _mm256_add_ps(vs, _mm256_and_ps(_mm256_blendv_ps(vthr, e2, in), _mm256_castsi256_ps(live)));
        };
        int i = i0;
        for (; i + 8 <= i1; i += 8) {
            const __m256 x = _mm256_loadu_ps(X + i), y = _mm256_loadu_ps(Y + i), z = _mm256_loadu_ps(Z + i);
            const __m256 a = _mm256_fmadd_ps(k0, x, _mm256_fmadd_ps(k1, y, _mm256_fmadd_ps(k2, z, k3)));
            const __m256 b = _mm256_fmadd_ps(k4, x, _mm256_fmadd_ps(k5, y, _mm256_fmadd_ps(k6, z, k7)));
            const __m256 c = _mm256_fmadd_ps(k8, x, _mm256_fmadd_ps(k9, y, _mm256_fmadd_ps(k10, z, k11)));
            const __m256 ic = _mm256_div_ps(one, c);
            const __m256 du = _mm256_fmsub_ps(a, ic, _mm256_loadu_ps(U + i)), dv = _mm256_fmsub_ps(b, ic, _mm256_loadu_ps(V + i));
            const __m256 e2 = _mm256_fmadd_ps(du, du, _mm256_mul_ps(dv, dv));
            const __m256 in = _mm256_and_ps(_mm256_cmp_ps(c, veps, _CMP_GT_OQ), _mm256_cmp_ps(e2, vthr, _CMP_LT_OQ));
            vc = _mm256_sub_epi32(vc, _mm256_castps_si256(in));
            vs = _mm256_add_ps(vs, _mm256_blendv_ps(vthr, e2, in));
        }
        if (i < i1) {                                                 // only the last block of the list can end inside a vector
            alignas(32) int lanes[8];
            for (int l = 0; l < 8; ++l) lanes[l] = i + l < i1 ? -1 : 0;
            step(i, _mm256_load_si256(reinterpret_cast<const __m256i*>(lanes)));
        }
        __m128i c4 = _mm_add_epi32(_mm256_castsi256_si128(vc), _mm256_extracti128_si256(vc, 1));
        c4 = _mm_add_epi32(c4, _mm_shuffle_epi32(c4, 0x4E));
        c4 = _mm_add_epi32(c4, _mm_shuffle_epi32(c4, 0xB1));
        __m128 s4 = _mm_add_ps(_mm256_castps256_ps128(vs), _mm256_extractf128_ps(vs, 1));
        s4 = _mm_add_ps(s4, _mm_movehl_ps(s4, s4));
        s4 = _mm_add_ss(s4, _mm_shuffle_ps(s4, s4, 1));
        cnt += _mm_cvtsi128_si32(c4);
        cost += _mm_cvtss_f32(s4);
        if (!keep_scoring(cnt, cost, n - i1, best_cnt, best_cost)) return false;
    }
    *cnt_out = cnt;
    *cost_out = cost;
    return true;
}
#else
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
bool score_fast(const FastPoints& F, const float* kp, float thr2, int best_cnt, float best_cost, int* cnt_out, float* cost_out) {
    int cnt = 0;
    float cost = 0.f;
    const float *X = F.x.data(), *Y = F.y.data(), *Z = F.z.data(), *U = F.u.data(), *V = F.v.data();
    int i1 = 0;
    for (int i0 = 0; i0 < F.n; i0 = i1) {
        const int len = i0 == 0 ? first_block(F.n, best_cnt) : 256;
        i1 = i0 + len < F.n ? i0 + len : F.n;
        int c_b = 0;
        float s_b = 0.f;
#pragma omp simd reduction(+ : c_b, s_b)
        for (int i = i0; i < i1; ++i) {
            const float a = kp[0] * X[i] + kp[1] * Y[i] + kp[2] * Z[i] + kp[3];
            const float b = kp[4] * X[i] + kp[5] * Y[i] + kp[6] * Z[i] + kp[7];
            const float c = kp[8] * X[i] + kp[9] * Y[i] + kp[10] * Z[i] + kp[11];
            const float ic = 1.0f / c;
            const float du = a * ic - U[i], dv = b * ic - V[i];
            const float e2 = du * du + dv * dv;
            const bool in = (c > 1e-12f) & (e2 < thr2);
            c_b += in ? 1 : 0;
            s_b += in ? e2 : thr2;
        }
        cnt += c_b;
        cost += s_b;
        if (!keep_scoring(cnt, cost, F.n - i1, best_cnt, best_cost)) return false;
    }
    *cnt_out = cnt;
    *cost_out = cost;
    return true;
}
#endif

struct Problem {
    int n;
    const double* K;          // 3x3
    std::vector<double> ray;  // [n][2] normalised image coordinates
    std::vector<double> px;   // [n][2] pixels
    std::vector<double> X;    // [n][3]
};

// pose = [R | t] row-major 3x4
int count_inliers(const Problem& P, const double* pose, double thr2, unsigned char* mask, double* cost) {
    const double fx = P.K[0], fy = P.K[4], cx = P.K[2], cy = P.K[5], sk = P.K[1];
    int cnt = 0;
    double c = 0.0;
    for (int i = 0; i < P.n; ++i) {
        const double* x = &P.X[3 * i];
        const double xc = pose[0] * x[0] + pose[1] * x[1] + pose[2] * x[2] + pose[3];
        const double yc = pose[4] * x[0] + pose[5] * x[1] + pose[6] * x[2] + pose[7];
        const double zc = pose[8] * x[0] + pose[9] * x[1] + pose[10] * x[2] + pose[11];
        bool in = false;
        if (zc > 1e-12) {
            const double xn = xc / zc, yn = yc / zc;
            const double du = fx * xn + sk * yn + cx - P.px[2 * i], dv = fy * yn + cy - P.px[2 * i + 1];
            const double e2 = du * du + dv * dv;
            in = e2 < thr2;
            c += in ? e2 : thr2;
        } else {
            c += thr2;
        }
        if (mask) mask[i] = in ? 1 : 0;
        cnt += in;
    }
    if (cost) *cost = c;
    return cnt;
}

// ---- four hypotheses at a time -------------------------------------------------------------------------------------------
// The minimal solver is 2/3 of a RANSAC trial (12 of 18 ms per 10 000 trials at 3 000 correspondences), so it runs on four
// samples at once: every quantity is a vector of four doubles (GCC vector extension: one AVX2 register in the -mavx2 build,
// two SSE2 registers otherwise), one lane per sample, all lanes through the same straight-line arithmetic.  Failure tests
// become a lane mask, the polar iteration runs a fixed number of Newton steps.
typedef double vd __attribute__((vector_size(32)));
typedef long long vl __attribute__((vector_size(32)));
inline vd vsplat(double x) { return vd{x, x, x, x}; }
inline vd vsel(vl m, vd a, vd b) { return (vd)(((vl)a & m) | ((vl)b & ~m)); }
inline vd vabs(vd x) { return vsel(x < vsplat(0.0), -x, x); }
inline vd vsqrt(vd x) {
#if defined(__AVX2__) && defined(__FMA__)
    return (vd)_mm256_sqrt_pd((__m256d)x);
#else
    vd r;
    for (int i = 0; i < 4; ++i) r[i] = std::sqrt(x[i]);
    return r;
#endif
}
inline vd vdet3(const vd* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// 6-point DLT on normalised rays -> four poses (idx[lane][k], pose[lane][12]); returns the mask of non-degenerate lanes
int dlt_pose4(const Problem& P, const int (*idx)[6], double (*pose)[12]) {
    constexpr int m = 6;
    const vd zero = vsplat(0.0), one = vsplat(1.0);
    vd X[m][3], rx[m], ry[m];
    for (int k = 0; k < m; ++k)
        for (int l = 0; l < 4; ++l) {
            const int i = idx[l][k];
            X[k][0][l] = P.X[3 * i]; X[k][1][l] = P.X[3 * i + 1]; X[k][2][l] = P.X[3 * i + 2];
            rx[k][l] = P.ray[2 * i]; ry[k][l] = P.ray[2 * i + 1];
        }
    // Hartley-style conditioning of the 3D points
    vd mu[3] = {zero, zero, zero};
    for (int k = 0; k < m; ++k)
        for (int d = 0; d < 3; ++d) mu[d] += X[k][d];
    for (int d = 0; d < 3; ++d) mu[d] /= vsplat((double)m);
    vd sc = zero;
    for (int k = 0; k < m; ++k) {
        vd r2 = zero;
        for (int d = 0; d < 3; ++d) { const vd v = X[k][d] - mu[d]; r2 += v * v; }
        sc += vsqrt(r2);
    }
    vl ok = ~(sc < vsplat(1e-300));
    sc = vsplat(std::sqrt(3.0) * m) / sc;
    // normal matrix of the DLT rows  [Xh 0 -x Xh ; 0 Xh -y Xh]:  A = [[S, 0, -Sx], [0, S, -Sy], [-Sx, -Sy, Sxx + Syy]]  (4x4 blocks)
    vd S[16], Sx[16], Sy[16], Sq[16];
    for (int a = 0; a < 16; ++a) S[a] = Sx[a] = Sy[a] = Sq[a] = zero;
    for (int k = 0; k < m; ++k) {
        const vd Xh[4] = {(X[k][0] - mu[0]) * sc, (X[k][1] - mu[1]) * sc, (X[k][2] - mu[2]) * sc, one};
        const vd x = rx[k], y = ry[k], q2 = x * x + y * y;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const vd o = Xh[a] * Xh[b];
                S[a * 4 + b] += o; Sx[a * 4 + b] += x * o; Sy[a * 4 + b] += y * o; Sq[a * 4 + b] += q2 * o;
            }
    }
    // S^-1 by Cholesky (symmetric positive definite unless the sample is degenerate)
    vd Si[16];
    {
        vd Lm[16];
        for (int a = 0; a < 16; ++a) Lm[a] = zero;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j <= i; ++j) {
                vd sum = S[i * 4 + j];
                for (int k = 0; k < j; ++k) sum -= Lm[i * 4 + k] * Lm[j * 4 + k];
                if (i == j) {
                    ok &= sum > vsplat(1e-300);
                    Lm[i * 4 + i] = vsqrt(sum);
                } else {
                    Lm[i * 4 + j] = sum / Lm[j * 4 + j];
                }
            }
        for (int c = 0; c < 4; ++c) {                      // solve L L^T x = e_c
            vd y[4], x[4];
            for (int i = 0; i < 4; ++i) {
                vd sum = i == c ? one : zero;
                for (int k = 0; k < i; ++k) sum -= Lm[i * 4 + k] * y[k];
                y[i] = sum / Lm[i * 4 + i];
            }
            for (int i = 3; i >= 0; --i) {
                vd sum = y[i];
                for (int k = i + 1; k < 4; ++k) sum -= Lm[k * 4 + i] * x[k];
                x[i] = sum / Lm[i * 4 + i];
            }
            for (int i = 0; i < 4; ++i) Si[i * 4 + c] = x[i];
        }
    }
    // minimise p^T A p over the last projection row p3 (|p3| = 1) with the first two rows eliminated: p1 = S^-1 Sx p3,
    // p2 = S^-1 Sy p3, and p3 = the eigenvector of the smallest eigenvalue of the 4 x 4 Schur complement
    // Sq - Sx S^-1 Sx - Sy S^-1 Sy (all blocks symmetric): a 4 x 4 problem instead of the 12 x 12 one
    vd SiSx[16], SiSy[16], C4[16];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            vd u = zero, v = zero;
            for (int k = 0; k < 4; ++k) { u += Si[a * 4 + k] * Sx[k * 4 + b]; v += Si[a * 4 + k] * Sy[k * 4 + b]; }
            SiSx[a * 4 + b] = u; SiSy[a * 4 + b] = v;
        }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            vd u = Sq[a * 4 + b];
            for (int k = 0; k < 4; ++k) u -= Sx[a * 4 + k] * SiSx[k * 4 + b] + Sy[a * 4 + k] * SiSy[k * 4 + b];
            C4[a * 4 + b] = u;
        }
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b) C4[a * 4 + b] = C4[b * 4 + a] = vsplat(0.5) * (C4[a * 4 + b] + C4[b * 4 + a]);
    vd tr = zero;
    for (int a = 0; a < 4; ++a) tr += C4[a * 4 + a];
    // smallest eigenvector of C4: three inverse iterations on C4 + eps I (Cholesky), fixed generic start
    vd p[12];
    {
        vd Lm[16];
        for (int a = 0; a < 16; ++a) Lm[a] = zero;
        const vd eps = vsplat(1e-13) * vsel(tr > zero, tr, one);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j <= i; ++j) {
                vd sum = C4[i * 4 + j] + (i == j ? eps : zero);
                for (int k = 0; k < j; ++k) sum -= Lm[i * 4 + k] * Lm[j * 4 + k];
                if (i == j) {
                    ok &= sum > zero;
                    Lm[i * 4 + i] = vsqrt(sum);
                } else {
                    Lm[i * 4 + j] = sum / Lm[j * 4 + j];
                }
            }
        vd* v = p + 8;
        for (int i = 0; i < 4; ++i) v[i] = vsplat(1.0 + 0.0625 * i);
        for (int it = 0; it < 3; ++it) {
            vd y[4];
            for (int i = 0; i < 4; ++i) {
                vd sum = v[i];
                for (int k = 0; k < i; ++k) sum -= Lm[i * 4 + k] * y[k];
                y[i] = sum / Lm[i * 4 + i];
            }
            for (int i = 3; i >= 0; --i) {
                vd sum = y[i];
                for (int k = i + 1; k < 4; ++k) sum -= Lm[k * 4 + i] * v[k];
                v[i] = sum / Lm[i * 4 + i];
            }
            vd nrm = zero;
            for (int i = 0; i < 4; ++i) nrm += v[i] * v[i];
            ok &= (nrm > zero) & (nrm < vsplat(1e300));
            nrm = one / vsqrt(nrm);
            for (int i = 0; i < 4; ++i) v[i] *= nrm;
        }
    }
    for (int a = 0; a < 4; ++a) {
        vd u = zero, v = zero;
        for (int k = 0; k < 4; ++k) { u += SiSx[a * 4 + k] * p[8 + k]; v += SiSy[a * 4 + k] * p[8 + k]; }
        p[a] = u; p[4 + a] = v;
    }
    vd M[9] = {p[0], p[1], p[2], p[4], p[5], p[6], p[8], p[9], p[10]};
    vd t[3] = {p[3], p[7], p[11]};
    vd d = vdet3(M);
    ok &= ~(vabs(d) < vsplat(1e-18));
    {
        const vd sg = vsel(d < zero, vsplat(-1.0), one);
        for (vd& v : M) v *= sg;
        for (vd& v : t) v *= sg;
        d *= sg;
    }
    vd s;
    for (int l = 0; l < 4; ++l) s[l] = std::cbrt(d[l]);
    for (vd& v : M) v /= s;
    for (vd& v : t) v /= s;
    // projection to SO(3): Newton iteration R <- (R + R^-T) / 2, six steps (a hypothesis only has to score; the kept one is
    // polished to full precision by the caller).  Quadratic convergence: a sample of inliers starts within a few percent of a rotation
    for (int it = 0; it < 6; ++it) {
        const vd dd = vdet3(M);
        ok &= ~(vabs(dd) < vsplat(1e-14));
        const vd id = one / dd;
        vd inv[9];
        inv[0] = (M[4] * M[8] - M[5] * M[7]) * id; inv[1] = (M[2] * M[7] - M[1] * M[8]) * id; inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
        inv[3] = (M[5] * M[6] - M[3] * M[8]) * id; inv[4] = (M[0] * M[8] - M[2] * M[6]) * id; inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
        inv[6] = (M[3] * M[7] - M[4] * M[6]) * id; inv[7] = (M[1] * M[6] - M[0] * M[7]) * id; inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M[i * 3 + j] = vsplat(0.5) * (M[i * 3 + j] + inv[j * 3 + i]);
    }
    // undo the conditioning: X_n = sc (X - mu)  =>  R X_n + t = (sc R) X + (t - sc R mu)
    int mask = 0;
    for (int l = 0; l < 4; ++l) {
        bool fin = true;
        for (int rI = 0; rI < 3; ++rI) {
            const double trn = t[rI][l] - sc[l] * (M[rI * 3][l] * mu[0][l] + M[rI * 3 + 1][l] * mu[1][l] + M[rI * 3 + 2][l] * mu[2][l]);
            pose[l][rI * 4] = M[rI * 3][l]; pose[l][rI * 4 + 1] = M[rI * 3 + 1][l]; pose[l][rI * 4 + 2] = M[rI * 3 + 2][l];
            pose[l][rI * 4 + 3] = trn / sc[l];
            for (int c = 0; c < 4; ++c) fin = fin && std::isfinite(pose[l][rI * 4 + c]);
        }
        if (ok[l] && fin) mask |= 1 << l;
    }
    return mask;
}


// ---- P3P minimal solver ---------------------------------------------------------------------------------------------------------
// Unit bearings f_i, world points P_i, unknown depths s_i with |s_i f_i - s_j f_j| = |P_i - P_j|.  With u = s2 / s1, v = s3 / s1
// and a, b, c = |P2 P3|, |P1 P3|, |P1 P2|, the three law-of-cosines equations give (Grunert 1841, see Haralick et al. 1994)
//     u = N(v) / D(v),  N = (p - 1) v^2 - 2 p cos(beta) v + p + 1,  D = 2 (cos(gamma) - v cos(alpha)),  p = (a^2 - c^2) / b^2
// and, substituted into the third equation, the quartic  D^2 + N^2 - 2 cos(gamma) N D - (c^2 / b^2) (1 - 2 cos(beta) v + v^2) D^2 = 0,
// whose coefficients are formed here by multiplying the polynomials out (no closed-form coefficient table to get wrong).
inline double cbrt_s(double x) { return std::cbrt(x); }

// largest real root of z^3 + A z^2 + B z + C
double cubic_largest_root(double A, double B, double C) {
    const double a3 = A / 3.0;
    const double P = B - A * a3, Q = 2.0 * a3 * a3 * a3 - a3 * B + C;
    const double disc = 0.25 * Q * Q + P * P * P / 27.0;
    double w;
    if (disc > 0.0) {
        const double sq = std::sqrt(disc);
        w = cbrt_s(-0.5 * Q + sq) + cbrt_s(-0.5 * Q - sq);
    } else {
        const double m = 2.0 * std::sqrt(-P / 3.0);
        double arg = m > 0.0 ? 3.0 * Q / (P * m) : 0.0;
        arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
        w = m * std::cos(std::acos(arg) / 3.0);
    }
    double z = w - a3;
    for (int it = 0; it < 3; ++it) {                       // Newton polish
        const double f = ((z + A) * z + B) * z + C, df = (3.0 * z + 2.0 * A) * z + B;
        if (std::fabs(df) < 1e-300) break;
        z -= f / df;
    }
    return z;
}

// real roots of c4 x^4 + c3 x^3 + c2 x^2 + c1 x + c0 (Ferrari, each root polished on the original polynomial); returns their number
int quartic_real_roots(const double* c, double* roots) {
    if (!(std::fabs(c[4]) > 1e-14 * (std::fabs(c[3]) + std::fabs(c[2]) + std::fabs(c[1]) + std::fabs(c[0]) + 1e-300))) return 0;
    const double a = c[3] / c[4], b = c[2] / c[4], cc = c[1] / c[4], d = c[0] / c[4];
    const double a2 = a * a;
    const double p = b - 0.375 * a2, q = cc - 0.5 * a * b + 0.125 * a2 * a, r = d - 0.25 * a * cc + 0.0625 * a2 * b - (3.0 / 256.0) * a2 * a2;
    double y[4];
    int n = 0;
    const double scale = std::fabs(p) + std::sqrt(std::fabs(r)) + 1e-300;
    if (std::fabs(q) < 1e-12 * scale * std::sqrt(scale)) {                // biquadratic
        double disc = p * p - 4.0 * r;
        if (disc < 0.0) { if (disc > -1e-12 * scale * scale) disc = 0.0; else return 0; }
        const double sq = std::sqrt(disc);
        for (const double w : {0.5 * (-p + sq), 0.5 * (-p - sq)}) {
            if (w < 0.0) continue;
            const double s = std::sqrt(w);
            y[n++] = s; y[n++] = -s;
        }
    } else {
        double z = cubic_largest_root(2.0 * p, p * p - 4.0 * r, -q * q);      // > 0 since q != 0
        if (!(z > 0.0)) return 0;
        const double s = std::sqrt(z), t1 = 0.5 * (p + z - q / s), t2 = 0.5 * (p + z + q / s);
        const double tol = 1e-10 * (z + std::fabs(t1) + std::fabs(t2));
        double d1 = z - 4.0 * t1, d2 = z - 4.0 * t2;
        if (d1 > -tol) { d1 = std::sqrt(d1 > 0.0 ? d1 : 0.0); y[n++] = 0.5 * (-s + d1); y[n++] = 0.5 * (-s - d1); }
        if (d2 > -tol) { d2 = std::sqrt(d2 > 0.0 ? d2 : 0.0); y[n++] = 0.5 * (s + d2); y[n++] = 0.5 * (s - d2); }
    }
    // Newton polish on the original polynomial, the (up to four) roots as the lanes of one vector: three steps are three dependent
    // divisions instead of twelve.  A lane whose derivative vanishes stops moving (the scalar loop's `break`).
    vd x = vsplat(0.0);
    for (int i = 0; i < n; ++i) x[i] = y[i] - 0.25 * a;
    const vd c4 = vsplat(c[4]), c3 = vsplat(c[3]), c2 = vsplat(c[2]), c1 = vsplat(c[1]), c0 = vsplat(c[0]);
    vl live = vsplat(1.0) > vsplat(0.0);
    for (int it = 0; it < 3; ++it) {
        const vd f = (((c4 * x + c3) * x + c2) * x + c1) * x + c0;
        const vd df = ((vsplat(4.0) * c4 * x + vsplat(3.0) * c3) * x + vsplat(2.0) * c2) * x + c1;
        live &= vabs(df) >= vsplat(1e-300);
        x -= vsel(live, f / vsel(live, df, vsplat(1.0)), vsplat(0.0));
    }
    for (int i = 0; i < n; ++i) roots[i] = x[i];
    return n;
}

// right-handed orthonormal frame of a triangle: e1 along Q0 -> Q1, e3 its normal; false when the points are collinear
bool triangle_frame(const double* Q0, const double* Q1, const double* Q2, double* E) {
    double e1[3] = {Q1[0] - Q0[0], Q1[1] - Q0[1], Q1[2] - Q0[2]}, w[3] = {Q2[0] - Q0[0], Q2[1] - Q0[1], Q2[2] - Q0[2]};
    const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    if (!(n1 > 1e-300)) return false;
    for (double& v : e1) v /= n1;
    double e3[3] = {e1[1] * w[2] - e1[2] * w[1], e1[2] * w[0] - e1[0] * w[2], e1[0] * w[1] - e1[1] * w[0]};
    const double n3 = std::sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    const double nw = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(n3 > 1e-9 * nw) || !(nw > 0.0)) return false;
    for (double& v : e3) v /= n3;
    const double e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    for (int i = 0; i < 3; ++i) { E[i * 3] = e1[i]; E[i * 3 + 1] = e2[i]; E[i * 3 + 2] = e3[i]; }
    return true;
}

// rays: three normalised image points (x, y) -> bearings (x, y, 1) / |.|; X: three world points.  poses[k] = [R | t] row-major.
int p3p_poses(const double (*ray)[2], const double (*X)[3], double (*poses)[12]) {
    double f[3][3];
    for (int i = 0; i < 3; ++i) {
        const double inv = 1.0 / std::sqrt(ray[i][0] * ray[i][0] + ray[i][1] * ray[i][1] + 1.0);
        f[i][0] = ray[i][0] * inv; f[i][1] = ray[i][1] * inv; f[i][2] = inv;
    }
    auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto d2 = [](const double* a, const double* b) { return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]); };
    const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
    const double a2 = d2(X[1], X[2]), b2 = d2(X[0], X[2]), c2 = d2(X[0], X[1]);
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
    double EP[9];
    if (!triangle_frame(X[0], X[1], X[2], EP)) return 0;
    const double p = (a2 - c2) / b2, k = c2 / b2;
    // polynomials in v, lowest coefficient first
    const double N[3] = {p + 1.0, -2.0 * p * cb, p - 1.0}, D[2] = {2.0 * cg, -2.0 * ca}, E[3] = {1.0, -2.0 * cb, 1.0};
    const double D2[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
    double c[5] = {D2[0], D2[1], D2[2], 0.0, 0.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i + j] += N[i] * N[j] - k * E[i] * D2[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) c[i + j] -= 2.0 * cg * N[i] * D[j];
    double vs[4];
    const int nr = quartic_real_roots(c, vs);
    if (nr == 0) return 0;
    // every root through the same straight-line arithmetic, one root per lane (a sample has two real roots as a rule, four at most):
    // depths -> camera-frame triangle -> its orthonormal frame -> R = E_C E_P^T, t = C_0 - R X_0.  Rejections become a lane mask.
    const vd one = vsplat(1.0), zero = vsplat(0.0);
    vd v = one;
    vl ok = zero > one;                                                  // all lanes off
    for (int ri = 0; ri < nr; ++ri) {
        v[ri] = vs[ri];
        bool good = vs[ri] > 0.0 && std::isfinite(vs[ri]);
        for (int rj = 0; rj < ri; ++rj) good = good && !(std::fabs(vs[rj] - vs[ri]) < 1e-9 * (1.0 + std::fabs(vs[ri])));      // a double root counted once
        if (good) ok[ri] = -1;
    }
    const vd den = vsplat(D[0]) + vsplat(D[1]) * v;
    ok &= vabs(den) >= vsplat(1e-12);
    const vd u = (vsplat(N[0]) + (vsplat(N[1]) + vsplat(N[2]) * v) * v) / vsel(ok, den, one);
    ok &= u > zero;
    const vd q = one + u * u - vsplat(2.0 * cg) * u;
    ok &= q > vsplat(1e-300);
    const vd s1 = vsqrt(vsplat(c2) / vsel(ok, q, one));
    const vd sd[3] = {s1, u * s1, v * s1};
    vd Cc[3][3];
    for (int i = 0; i < 3; ++i)
        for (int d = 0; d < 3; ++d) Cc[i][d] = sd[i] * vsplat(f[i][d]);
    // triangle_frame on the lanes
    vd e1[3] = {Cc[1][0] - Cc[0][0], Cc[1][1] - Cc[0][1], Cc[1][2] - Cc[0][2]}, w[3] = {Cc[2][0] - Cc[0][0], Cc[2][1] - Cc[0][1], Cc[2][2] - Cc[0][2]};
    const vd n1 = vsqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    ok &= n1 > vsplat(1e-300);
    const vd in1 = one / vsel(ok, n1, one);
    for (vd& t : e1) t *= in1;
    vd e3[3] = {e1[1] * w[2] - e1[2] * w[1], e1[2] * w[0] - e1[0] * w[2], e1[0] * w[1] - e1[1] * w[0]};
    const vd n3 = vsqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    const vd nw = vsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    ok &= (n3 > vsplat(1e-9) * nw) & (nw > zero);
    const vd in3 = one / vsel(ok, n3, one);
    for (vd& t : e3) t *= in3;
    const vd e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    vd ps[12];
    vd fin_sum = zero;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) ps[i * 4 + j] = e1[i] * vsplat(EP[j * 3]) + e2[i] * vsplat(EP[j * 3 + 1]) + e3[i] * vsplat(EP[j * 3 + 2]);
        ps[i * 4 + 3] = Cc[0][i] - (ps[i * 4] * vsplat(X[0][0]) + ps[i * 4 + 1] * vsplat(X[0][1]) + ps[i * 4 + 2] * vsplat(X[0][2]));
        for (int j = 0; j < 4; ++j) fin_sum += vabs(ps[i * 4 + j]);
    }
    ok &= fin_sum < vsplat(1e300);                                       // every entry finite (a NaN or an infinity fails the comparison)
    int n = 0;
    for (int ri = 0; ri < nr; ++ri) {
        if (!ok[ri]) continue;
        for (int e = 0; e < 12; ++e) poses[n][e] = ps[e][ri];
        ++n;
    }
    return n;
}

// ---- P3P on four samples at once ---------------------------------------------------------------------------------------------
// The solver above is one long dependency chain (bearings -> quartic -> resolvent cubic -> roots -> polish -> depths -> two square
// roots and two divisions per triangle frame -> pose): ~900 cycles of latency that a RANSAC trial cannot hide.  Four samples, one per
// lane of a 4 x double vector, go through the same chain together.  Per lane the arithmetic is the scalar solver's, statement by
// statement (same thresholds, same root order: the -s pair before the +s pair); the two places where samples take different paths are
// handled without leaving the lanes: the resolvent cubic's closed form (cbrt or cos / acos by the sign of its discriminant) is
// evaluated per lane and polished on the lanes, and a biquadratic quartic (q = 0: a symmetric configuration) sends its lane through
// the scalar solver.  nsol[l] poses of sample l in poses[l][0 .. nsol[l]).
inline vl vfinite(vd x) { return (x - x) == vsplat(0.0); }

struct VFrame { vd e[9]; vl ok; };                                       // E[i * 3 + k]: component i of axis k, like triangle_frame

inline VFrame triangle_frame4(const vd* Q0, const vd* Q1, const vd* Q2) {
    const vd one = vsplat(1.0), zero = vsplat(0.0);
    VFrame F;
    vd e1[3] = {Q1[0] - Q0[0], Q1[1] - Q0[1], Q1[2] - Q0[2]}, w[3] = {Q2[0] - Q0[0], Q2[1] - Q0[1], Q2[2] - Q0[2]};
    const vd n1 = vsqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    F.ok = n1 > vsplat(1e-300);
    const vd in1 = one / vsel(F.ok, n1, one);
    for (vd& t : e1) t *= in1;
    vd e3[3] = {e1[1] * w[2] - e1[2] * w[1], e1[2] * w[0] - e1[0] * w[2], e1[0] * w[1] - e1[1] * w[0]};
    const vd n3 = vsqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
    const vd nw = vsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    F.ok &= (n3 > vsplat(1e-9) * nw) & (nw > zero);
    const vd in3 = one / vsel(F.ok, n3, one);
    for (vd& t : e3) t *= in3;
    const vd e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    for (int i = 0; i < 3; ++i) { F.e[i * 3] = e1[i]; F.e[i * 3 + 1] = e2[i]; F.e[i * 3 + 2] = e3[i]; }
    return F;
}

void p3p_poses4(const double (*ray)[3][2], const double (*X)[3][3], double (*poses)[4][12], int* nsol) {
    const vd one = vsplat(1.0), zero = vsplat(0.0);
    vd f[3][3], Xw[3][3];
    for (int i = 0; i < 3; ++i) {
        vd rx, ry;
        for (int l = 0; l < 4; ++l) {
            rx[l] = ray[l][i][0]; ry[l] = ray[l][i][1];
            for (int d = 0; d < 3; ++d) Xw[i][d][l] = X[l][i][d];
        }
        const vd inv = one / vsqrt(rx * rx + ry * ry + one);
        f[i][0] = rx * inv; f[i][1] = ry * inv; f[i][2] = inv;
    }
    auto dot = [](const vd* a, const vd* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto d2 = [](const vd* a, const vd* b) { return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]); };
    const vd ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
    const vd a2 = d2(Xw[1], Xw[2]), b2 = d2(Xw[0], Xw[2]), c2 = d2(Xw[0], Xw[1]);
    vl ok = (a2 > zero) & (b2 > zero) & (c2 > zero);
    const VFrame EP = triangle_frame4(Xw[0], Xw[1], Xw[2]);
    ok &= EP.ok;
    const vd b2s = vsel(ok, b2, one);
    const vd p = (a2 - c2) / b2s, k = c2 / b2s;
    const vd two = vsplat(2.0);
    const vd N[3] = {p + one, -two * p * cb, p - one}, D[2] = {two * cg, -two * ca}, E[3] = {one, -two * cb, one};
    const vd D2[3] = {D[0] * D[0], two * D[0] * D[1], D[1] * D[1]};
    vd c[5] = {D2[0], D2[1], D2[2], zero, zero};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i + j] += N[i] * N[j] - k * E[i] * D2[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) c[i + j] -= two * cg * N[i] * D[j];
    // ---- the quartic's real roots: four slots per lane (valid[k]), polished on the original polynomial ---------------------------
    vd y[4] = {zero, zero, zero, zero};
    vl valid[4] = {zero > one, zero > one, zero > one, zero > one};
    {
        vl lead = vabs(c[4]) > vsplat(1e-14) * (vabs(c[3]) + vabs(c[2]) + vabs(c[1]) + vabs(c[0]) + vsplat(1e-300));
        lead &= ok;
        const vd c4s = vsel(lead, c[4], one);
        const vd a = c[3] / c4s, b = c[2] / c4s, cc = c[1] / c4s, d = c[0] / c4s;
        const vd aa = a * a;
        const vd pp = b - vsplat(0.375) * aa, q = cc - vsplat(0.5) * a * b + vsplat(0.125) * aa * a;
        const vd r = d - vsplat(0.25) * a * cc + vsplat(0.0625) * aa * b - vsplat(3.0 / 256.0) * aa * aa;
        const vd scale = vabs(pp) + vsqrt(vabs(r)) + vsplat(1e-300);
        const vl biq = vabs(q) < vsplat(1e-12) * scale * vsqrt(scale);
        const vl gen = lead & ~biq;
        // resolvent cubic z^3 + 2 p z^2 + (p^2 - 4 r) z - q^2: closed form per lane, Newton polish on the lanes
        const vd A = two * pp, B = pp * pp - vsplat(4.0) * r, C = -q * q;
        vd z = one;
        {
            const vd a3 = A / vsplat(3.0);
            const vd P = B - A * a3, Q = two * a3 * a3 * a3 - a3 * B + C;
            const vd disc = vsplat(0.25) * Q * Q + P * P * P / vsplat(27.0);
            for (int l = 0; l < 4; ++l) {
                if (!gen[l]) continue;
                double w;
                if (disc[l] > 0.0) {
                    const double sq = std::sqrt(disc[l]);
                    w = cbrt_s(-0.5 * Q[l] + sq) + cbrt_s(-0.5 * Q[l] - sq);
                } else {
                    const double m = 2.0 * std::sqrt(-P[l] / 3.0);
                    double arg = m > 0.0 ? 3.0 * Q[l] / (P[l] * m) : 0.0;
                    arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
                    w = m * std::cos(std::acos(arg) / 3.0);
                }
                z[l] = w - a3[l];
            }
            vl live = gen;
            for (int it = 0; it < 3; ++it) {
                const vd fz = ((z + A) * z + B) * z + C, dfz = (vsplat(3.0) * z + two * A) * z + B;
                live &= vabs(dfz) >= vsplat(1e-300);
                z -= vsel(live, fz / vsel(live, dfz, one), zero);
            }
        }
        const vl zpos = gen & (z > zero);
        const vd zs = vsel(zpos, z, one);
        const vd s = vsqrt(zs), qs = q / s;
        const vd t1 = vsplat(0.5) * (pp + zs - qs), t2 = vsplat(0.5) * (pp + zs + qs);
        const vd tol = vsplat(1e-10) * (zs + vabs(t1) + vabs(t2));
        const vd d1 = zs - vsplat(4.0) * t1, dd2 = zs - vsplat(4.0) * t2;
        const vl m1 = zpos & (d1 > -tol), m2 = zpos & (dd2 > -tol);
        const vd r1 = vsqrt(vsel(d1 > zero, d1, zero)), r2 = vsqrt(vsel(dd2 > zero, dd2, zero));
        // a sample has two real roots as a rule -- one of the two pairs -- and which one differs from lane to lane: a lane's FIRST existing
        // pair goes to slots 0, 1 and its second, if it has four real roots, to slots 2, 3 (the scalar solver's order within a lane), so
        // that the passes over slots 2 and 3 below are skipped for most groups of four
        const vd y0 = vsplat(0.5) * (-s + r1), y1 = vsplat(0.5) * (-s - r1), y2 = vsplat(0.5) * (s + r2), y3 = vsplat(0.5) * (s - r2);
        y[0] = vsel(m1, y0, y2); y[1] = vsel(m1, y1, y3); valid[0] = valid[1] = m1 | m2;
        y[2] = y2; y[3] = y3; valid[2] = valid[3] = m1 & m2;
        const vd shift = vsplat(0.25) * a;
        for (int kk = 0; kk < 4; ++kk) {
            if (!(valid[kk][0] | valid[kk][1] | valid[kk][2] | valid[kk][3])) continue;
            vd x = y[kk] - shift;
            vl live = valid[kk];
            for (int it = 0; it < 3; ++it) {
                const vd fx = (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
                const vd dfx = ((vsplat(4.0) * c[4] * x + vsplat(3.0) * c[3]) * x + two * c[2]) * x + c[1];
                live &= vabs(dfx) >= vsplat(1e-300);
                x -= vsel(live, fx / vsel(live, dfx, one), zero);
            }
            y[kk] = x;
        }
        // a biquadratic lane: the scalar root finder (roots in its order, into the slots from 0)
        for (int l = 0; l < 4; ++l) {
            if (!(lead[l] && biq[l])) continue;
            const double cl[5] = {c[0][l], c[1][l], c[2][l], c[3][l], c[4][l]};
            double rts[4];
            const int nr = quartic_real_roots(cl, rts);
            for (int kk = 0; kk < 4; ++kk) {
                valid[kk][l] = kk < nr ? -1 : 0;
                if (kk < nr) y[kk][l] = rts[kk];
            }
        }
    }
    // ---- every root slot: depths -> camera-frame triangle -> its frame -> R = E_C E_P^T, t = C_0 - R X_0 ---------------------------
    for (int l = 0; l < 4; ++l) nsol[l] = 0;
    for (int kk = 0; kk < 4; ++kk) {
        const vd v = vsel(valid[kk], y[kk], one);
        vl good = valid[kk] & (y[kk] > zero) & vfinite(y[kk]);
        if (!(good[0] | good[1] | good[2] | good[3])) continue;
        for (int kj = 0; kj < kk; ++kj)                                  // a double root counted once
            good &= ~(valid[kj] & (vabs(y[kj] - v) < vsplat(1e-9) * (one + vabs(v))));
        const vd den = D[0] + D[1] * v;
        good &= vabs(den) >= vsplat(1e-12);
        const vd u = (N[0] + (N[1] + N[2] * v) * v) / vsel(good, den, one);
        good &= u > zero;
        const vd qq = one + u * u - two * u * cg;
        good &= qq > vsplat(1e-300);
        const vd s1 = vsqrt(c2 / vsel(good, qq, one));
        const vd sd[3] = {s1, u * s1, v * s1};
        vd Cc[3][3];
        for (int i = 0; i < 3; ++i)
            for (int d = 0; d < 3; ++d) Cc[i][d] = sd[i] * f[i][d];
        const VFrame EC = triangle_frame4(Cc[0], Cc[1], Cc[2]);
        good &= EC.ok;
        vd ps[12];
        vd mag = zero;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) ps[i * 4 + j] = EC.e[i * 3] * EP.e[j * 3] + EC.e[i * 3 + 1] * EP.e[j * 3 + 1] + EC.e[i * 3 + 2] * EP.e[j * 3 + 2];
            ps[i * 4 + 3] = Cc[0][i] - (ps[i * 4] * Xw[0][0] + ps[i * 4 + 1] * Xw[0][1] + ps[i * 4 + 2] * Xw[0][2]);
            for (int j = 0; j < 4; ++j) mag += vabs(ps[i * 4 + j]);
        }
        good &= mag < vsplat(1e300);                                      // every entry finite
        for (int l = 0; l < 4; ++l) {
            if (!good[l]) continue;
            double* o = poses[l][nsol[l]++];
            for (int e = 0; e < 12; ++e) o[e] = ps[e][l];
        }
    }
}

void rodrigues(const double* w, double* R) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double a = th < 1e-12 ? 1.0 - th * th / 6.0 : std::sin(th) / th;
    const double b = th < 1e-12 ? 0.5 - th * th / 24.0 : (1.0 - std::cos(th)) / (th * th);
    const double Kx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double k2 = 0.0;
            for (int k = 0; k < 3; ++k) k2 += Kx[i * 3 + k] * Kx[k * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0 : 0.0) + a * Kx[i * 3 + j] + b * k2;
        }
}

bool solve6(double* H, double* g, double* dx) {          // Cholesky-free Gaussian elimination with partial pivoting
    double M[6][7];
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) M[i][j] = H[i * 6 + j]; M[i][6] = g[i]; }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int rI = c + 1; rI < 6; ++rI) if (std::fabs(M[rI][c]) > std::fabs(M[piv][c])) piv = rI;
        if (std::fabs(M[piv][c]) < 1e-300) return false;
        if (piv != c) for (int j = 0; j < 7; ++j) { const double tmp = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = tmp; }
        for (int rI = c + 1; rI < 6; ++rI) {
            const double f = M[rI][c] / M[c][c];
            for (int j = c; j < 7; ++j) M[rI][j] -= f * M[c][j];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double s = M[i][6];
        for (int j = i + 1; j < 6; ++j) s -= M[i][j] * dx[j];
        dx[i] = s / M[i][i];
    }
    return true;
}

// Levenberg-Marquardt on the masked points; pose updated in place (left-multiplied rotation increment)
void refine_lm(const Problem& P, const unsigned char* mask, double* pose, int iters) {
    const double fx = P.K[0], fy = P.K[4], sk = P.K[1], cx = P.K[2], cy = P.K[5];
    double lambda = 1e-3;
    // the masked points once as a structure of arrays (padded to a multiple of four with weight 0): every pass below runs four
    // points per step on 4 x double vectors -- the refinement is the serial tail of a frame's pose (1 of its ~1.5 ms of latency)
    std::vector<double> sx, sy, sz, su, sv, sw;
    for (int i = 0; i < P.n; ++i) {
        if (!mask[i]) continue;
        sx.push_back(P.X[3 * i]); sy.push_back(P.X[3 * i + 1]); sz.push_back(P.X[3 * i + 2]);
        su.push_back(P.px[2 * i]); sv.push_back(P.px[2 * i + 1]); sw.push_back(1.0);
    }
    while (sx.size() % 4) { sx.push_back(0.0); sy.push_back(0.0); sz.push_back(1.0); su.push_back(0.0); sv.push_back(0.0); sw.push_back(0.0); }
    const int n4 = (int)sx.size() / 4;
    auto ld = [](const std::vector<double>& v, int g) { vd r; std::memcpy(&r, &v[4 * (size_t)g], sizeof(r)); return r; };
    auto hsum = [](vd v) { return (v[0] + v[1]) + (v[2] + v[3]); };
    const vd zero = vsplat(0.0);
    auto cost_of = [&](const double* ps) {
        vd c = zero;
        for (int g = 0; g < n4; ++g) {
            const vd x = ld(sx, g), y = ld(sy, g), z = ld(sz, g), w = ld(sw, g);
            const vd xc = vsplat(ps[0]) * x + vsplat(ps[1]) * y + vsplat(ps[2]) * z + vsplat(ps[3]);
            const vd yc = vsplat(ps[4]) * x + vsplat(ps[5]) * y + vsplat(ps[6]) * z + vsplat(ps[7]);
            const vd zc = vsplat(ps[8]) * x + vsplat(ps[9]) * y + vsplat(ps[10]) * z + vsplat(ps[11]);
            const vl front = zc > vsplat(1e-12);
            const vd izc = vsplat(1.0) / vsel(front, zc, vsplat(1.0));
            const vd xn = xc * izc, yn = yc * izc;
            const vd du = vsplat(fx) * xn + vsplat(sk) * yn + vsplat(cx) - ld(su, g), dv = vsplat(fy) * yn + vsplat(cy) - ld(sv, g);
            c += w * vsel(front, du * du + dv * dv, vsplat(1e12));
        }
        return hsum(c);
    };
    double cur = cost_of(pose);
    for (int it = 0; it < iters; ++it) {
        double H[36], g[6];
        vd Hv[21], gv[6];
        for (vd& v : Hv) v = zero;
        for (vd& v : gv) v = zero;
        for (int gI = 0; gI < n4; ++gI) {
            const vd x = ld(sx, gI), y = ld(sy, gI), z = ld(sz, gI);
            const vd pc0 = vsplat(pose[0]) * x + vsplat(pose[1]) * y + vsplat(pose[2]) * z + vsplat(pose[3]);
            const vd pc1 = vsplat(pose[4]) * x + vsplat(pose[5]) * y + vsplat(pose[6]) * z + vsplat(pose[7]);
            const vd pc2 = vsplat(pose[8]) * x + vsplat(pose[9]) * y + vsplat(pose[10]) * z + vsplat(pose[11]);
            const vl front = pc2 > vsplat(1e-12);
            const vd w = vsel(front, ld(sw, gI), zero);          // points behind the camera (and the padding) contribute nothing
            const vd iz = vsplat(1.0) / vsel(front, pc2, vsplat(1.0)), xn = pc0 * iz, yn = pc1 * iz;
            const vd ru = w * (vsplat(fx) * xn + vsplat(sk) * yn + vsplat(cx) - ld(su, gI)), rv = w * (vsplat(fy) * yn + vsplat(cy) - ld(sv, gI));
            // d(u,v)/d(pc)
            const vd Ju[3] = {w * vsplat(fx) * iz, w * vsplat(sk) * iz, -w * (vsplat(fx) * xn + vsplat(sk) * yn) * iz};
            const vd Jv[3] = {zero, w * vsplat(fy) * iz, -w * vsplat(fy) * yn * iz};
            // d(pc)/d(w) = -[pc - t]_x ... with the left increment R' = exp(w) R: pc' = exp(w) (pc - t) + t + dt
            const vd q[3] = {pc0 - vsplat(pose[3]), pc1 - vsplat(pose[7]), pc2 - vsplat(pose[11])};
            vd ju[6], jv[6];
            // dW = [[0, q2, -q1], [-q2, 0, q0], [q1, -q0, 0]];  ju[k] = sum_r Ju[r] dW[r][k]
            ju[0] = -Ju[1] * q[2] + Ju[2] * q[1]; ju[1] = Ju[0] * q[2] - Ju[2] * q[0]; ju[2] = -Ju[0] * q[1] + Ju[1] * q[0];
            jv[0] = -Jv[1] * q[2] + Jv[2] * q[1]; jv[1] = Jv[0] * q[2] - Jv[2] * q[0]; jv[2] = -Jv[0] * q[1] + Jv[1] * q[0];
            for (int k = 0; k < 3; ++k) { ju[3 + k] = Ju[k]; jv[3 + k] = Jv[k]; }
            int e = 0;
            for (int a = 0; a < 6; ++a) {
                gv[a] -= ju[a] * ru + jv[a] * rv;
                for (int b2 = a; b2 < 6; ++b2) Hv[e++] += ju[a] * ju[b2] + jv[a] * jv[b2];      // upper triangle
            }
        }
        {
            int e = 0;
            for (int a = 0; a < 6; ++a) {
                g[a] = hsum(gv[a]);
                for (int b2 = a; b2 < 6; ++b2) H[a * 6 + b2] = hsum(Hv[e++]);
            }
        }
        for (int a = 1; a < 6; ++a)
            for (int b2 = 0; b2 < a; ++b2) H[a * 6 + b2] = H[b2 * 6 + a];
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; ++tries) {
            double Hd[36], dx[6];
            std::memcpy(Hd, H, sizeof(H));
            for (int a = 0; a < 6; ++a) Hd[a * 6 + a] *= 1.0 + lambda;
            if (!solve6(Hd, g, dx)) { lambda *= 10.0; continue; }
            double dR[9], np[12];
            rodrigues(dx, dR);
            for (int rI = 0; rI < 3; ++rI) {
                for (int c = 0; c < 3; ++c) np[rI * 4 + c] = dR[rI * 3] * pose[c] + dR[rI * 3 + 1] * pose[4 + c] + dR[rI * 3 + 2] * pose[8 + c];
                // t' = exp(w) t - exp(w) t + t + dt  (rotation about the camera-frame point t keeps pc' = exp(w)(pc - t) + t + dt)
                np[rI * 4 + 3] = pose[rI * 4 + 3] + dx[3 + rI];
            }
            // the increment rotates points about t: pc' = dR (R X) + t + dt  => new translation column is t + dt
            const double nc = cost_of(np);
            if (nc < cur) {
                std::memcpy(pose, np, sizeof(np));
                const double rel = (cur - nc) / (cur + 1e-300);
                cur = nc;
                lambda = lambda > 1e-9 ? lambda * 0.3 : lambda;
                improved = true;
                if (rel < 1e-12) return;       // cost converged to 12 digits: the pose is stationary to ~1e-6 relative
            } else {
                lambda *= 10.0;
            }
        }
        if (!improved) return;
    }
}

}  // namespace

// ---- RANSAC in chunks of CH trials ------------------------------------------------------------------------------------
// Trial t belongs to chunk t / CH and every chunk has its own generator, so the floor of min_iters trials (the reference's
// pycolmap call: 10 000) splits into chunks that a pool solves in parallel; merging the chunks' best candidates in chunk order
// makes the result independent of how many threads took part (and identical to the sequential call).  Chunk 0 runs first and
// its best float score is the rejection bound every other chunk starts from (a function of the data only), so short chunks --
// low latency per frame, work for every thread -- still drop most hypotheses after a fraction of the points.
namespace {

constexpr int CH = 256;
// Chunk 0 is the serial head of a frame's RANSAC (its best float score bounds every other chunk, so they wait for it): it is short.
// A frame's pose latency is head + (floor - CH0) / threads + tail + refinement; with 256 trials the head was a quarter of it.
// The unconditional floor is rounded UP to whole chunks ("at least min_iters trials", as long as max_iters allows): a partial last
// chunk would run in the serial tail (208 of 10 000 trials with CH0 = 64: 0.2 ms of the 0.9 ms a frame's pose spends outside the pool's parallel part).
constexpr int CH0 = 32;
inline int chunk_len(int c) { return c == 0 ? CH0 : CH; }
inline long long chunk_start(int c) { return c == 0 ? 0 : CH0 + (long long)(c - 1) * CH; }

struct FloatBound {
    int cnt = 0;
    float cost = 3.0e38f;
};

struct Candidate {
    double pose[12];
    int cnt = 0;
    double cost = 1e300;
    std::vector<unsigned char> mask;
    bool better_than(const Candidate& o) const { return cnt > o.cnt || (cnt == o.cnt && cost < o.cost); }
};

struct Ransac {
    Problem P;
    FastPoints F;                    // float copy in the caller's order: what chunk 0 scores
    FastPoints G;                    // the same points, those chunk 0's best pose rejects FIRST: what every later chunk scores
    bool reordered = false;
    double K[9];
    double thr2 = 0, confidence = 0.99;
    int n = 0, min_iters = 0, max_iters = 0;
    int solver = 0;                  // 0: 6-point DLT (OpenCV branch), 1: P3P (pycolmap branch)
    unsigned long long seed = 1;

    int sample_size() const { return solver == 1 ? 3 : 6; }

    void setup(const double* K_, const float* pts2d, const float* pts3d, int n_, double reproj, double conf, int min_it, int max_it,
               unsigned long long seed_, int solver_) {
        solver = solver_;
        n = n_; std::memcpy(K, K_, sizeof(K));
        P.n = n; P.K = K;
        P.ray.resize(2 * (size_t)n); P.px.resize(2 * (size_t)n); P.X.resize(3 * (size_t)n);
        double Ki[9];
        inv3(K, Ki);
        F.n = n; F.x.resize(n); F.y.resize(n); F.z.resize(n); F.u.resize(n); F.v.resize(n);
        for (int i = 0; i < n; ++i) {
            const double u = pts2d[2 * i], v = pts2d[2 * i + 1];
            P.px[2 * i] = u; P.px[2 * i + 1] = v;
            const double w = Ki[6] * u + Ki[7] * v + Ki[8];
            P.ray[2 * i] = (Ki[0] * u + Ki[1] * v + Ki[2]) / w;
            P.ray[2 * i + 1] = (Ki[3] * u + Ki[4] * v + Ki[5]) / w;
            for (int d = 0; d < 3; ++d) P.X[3 * i + d] = pts3d[3 * i + d];
            F.x[i] = pts3d[3 * i]; F.y[i] = pts3d[3 * i + 1]; F.z[i] = pts3d[3 * i + 2];
            F.u[i] = pts2d[2 * i]; F.v[i] = pts2d[2 * i + 1];
        }
        thr2 = reproj * reproj; confidence = conf; min_iters = min_it; max_iters = max_it; seed = seed_;
    }
    int floor_trials() const {                       // unconditional trials: min_iters rounded up to whole chunks when max_iters has room for that
        const int f = min_iters < max_iters ? min_iters : max_iters;
        if (f <= 0) return 0;
        const long long up = f <= CH0 ? CH0 : CH0 + (long long)((f - CH0 + CH - 1) / CH) * CH;
        return up <= max_iters ? (int)up : f;
    }
    int full_chunks() const {
        const int floor_ = floor_trials();
        return floor_ < CH0 ? 0 : 1 + (floor_ - CH0) / CH;
    }

    // The scorer drops a hypothesis as soon as it can no longer beat the best one, and a hypothesis that is about as good as the
    // best can only be dropped once it has met the outliers (its count can then no longer exceed the best's and the cost decides).
    // Scored in the caller's order that takes most of the list; with the points chunk 0's best pose rejects moved to the front it
    // takes a block or two.  The order of evaluation does not change a hypothesis' count or (up to float rounding of the screening
    // sum) its cost, and it is a function of the data only (chunk 0 always runs first, alone): results stay independent of the
    // thread count.
    void reorder_after_chunk0(const Candidate& c0);

    int needed_for(int cnt) const {
        const double w = (double)cnt / n, pw = std::pow(w, (double)sample_size());
        if (pw > 1.0 - 1e-12) return 1;
        if (pw > 1e-12) {
            // (clamped BEFORE the conversion: a first hypothesis with a handful of inliers asks for billions of trials, which as an int was
            //  negative -- the adaptive branch then stopped at once and a frame with 40 % outliers came back without a pose)
            const double nd = std::ceil(std::log(1.0 - confidence) / std::log(1.0 - pw));
            return nd < (double)max_iters ? (int)nd : max_iters;
        }
        return max_iters;
    }

    // trials [0, limit) of `chunk`; `floor_in_chunk` of them unconditionally, after that while the global trial index is
    // below what the best candidate so far (`best`: carried in, updated) asks for.  Returns the number of trials run.
    // `bound`: float score (count, cost) a hypothesis has to beat to be looked at -- in: chunk 0's best for the other chunks
    // (fixed by the data, so the result does not depend on scheduling), out: this chunk's best float score
    int run_chunk(int chunk, int limit, int floor_in_chunk, Candidate& best, FloatBound& bound) const {
        Rng rng(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(chunk + 1));
        const FastPoints& FS = (chunk > 0 && reordered) ? G : F;
        std::vector<unsigned char> mask((size_t)n);
        int& best_cnt_f = bound.cnt;
        float& best_cost_f = bound.cost;
        int needed = best.cnt > 0 ? needed_for(best.cnt) : max_iters;
        int it = 0;
        auto wanted = [&](int i) { return i < limit && (i < floor_in_chunk || chunk_start(chunk) + i < needed); };
        // one hypothesis: float score with early rejection against the bound; a new float best is re-scored exactly (double)
        auto consider = [&](double* pose) {
            float kp[12];
            for (int c = 0; c < 4; ++c) {
                kp[c] = (float)(K[0] * pose[c] + K[1] * pose[4 + c] + K[2] * pose[8 + c]);
                kp[4 + c] = (float)(K[4] * pose[4 + c] + K[5] * pose[8 + c]);
                kp[8 + c] = (float)pose[8 + c];
            }
            float cost_f;
            int cnt_f;
            if (!score_fast(FS, kp, (float)thr2, best_cnt_f, best_cost_f, &cnt_f, &cost_f)) return;
            if (cnt_f < best_cnt_f || (cnt_f == best_cnt_f && !(cost_f < best_cost_f))) return;
            best_cnt_f = cnt_f; best_cost_f = cost_f;
            // a new best of this chunk by the float score: its exact (double) count, cost and mask decide what is kept
            Candidate c;
            {                                            // exact rotation before the exact score
                double Rm[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
                if (polar_rotation(Rm))
                    for (int rI = 0; rI < 3; ++rI)
                        for (int cI = 0; cI < 3; ++cI) pose[rI * 4 + cI] = Rm[rI * 3 + cI];
            }
            c.cnt = count_inliers(P, pose, thr2, mask.data(), &c.cost);
            if (c.better_than(best)) {
                std::memcpy(c.pose, pose, sizeof(double) * 12);
                c.mask = mask;
                best = std::move(c);
                needed = needed_for(best.cnt);
            }
        };
        auto draw = [&](int* idx, int m) {
            for (int k = 0; k < m;) {
                const int c = rng.below(n);
                bool dup = false;
                for (int j = 0; j < k; ++j) dup |= idx[j] == c;
                if (!dup) idx[k++] = c;
            }
        };
        auto p3p_of = [&](const int* idx, double (*poses)[12]) {
            double ray[3][2], X3[3][3];
            for (int k = 0; k < 3; ++k) {
                ray[k][0] = P.ray[2 * idx[k]]; ray[k][1] = P.ray[2 * idx[k] + 1];
                for (int d = 0; d < 3; ++d) X3[k][d] = P.X[3 * idx[k] + d];
            }
            return p3p_poses(ray, X3, poses);
        };
        if (solver == 1) {
            // P3P: a trial = one sample of three; every real root of its quartic is a hypothesis scored on all points (COLMAP's P3P
            // estimator does the same), wrong roots die in the scorer's first block
            // four samples per solver call (drawn in trial order from this chunk's generator), their hypotheses then looked at trial by
            // trial exactly as a one-at-a-time loop would: a trial the stopping rule no longer wants is neither counted nor looked at
            while (wanted(it)) {
                const int g = limit - it < 4 ? limit - it : 4;
                double ray4[4][3][2], X4[4][3][3];
                for (int l = 0; l < 4; ++l) {
                    int idx[3];
                    if (l < g) draw(idx, 3);
                    for (int k = 0; k < 3; ++k) {
                        if (l >= g) { std::memcpy(ray4[l][k], ray4[0][k], sizeof(ray4[0][k])); std::memcpy(X4[l][k], X4[0][k], sizeof(X4[0][k])); continue; }
                        ray4[l][k][0] = P.ray[2 * idx[k]]; ray4[l][k][1] = P.ray[2 * idx[k] + 1];
                        for (int d = 0; d < 3; ++d) X4[l][k][d] = P.X[3 * idx[k] + d];
                    }
                }
                double poses4[4][4][12];
                int nsol4[4];
                p3p_poses4(ray4, X4, poses4, nsol4);
                for (int l = 0; l < g; ++l) {
                    if (!wanted(it)) return it;
                    ++it;
                    for (int k = 0; k < nsol4[l]; ++k) consider(poses4[l][k]);
                }
            }
            return it;
        }
        while (wanted(it)) {
            // the minimal solver runs on four samples at once (drawn in trial order from this chunk's generator); the four hypotheses
            // are then looked at one by one in that order, exactly as a one-at-a-time loop would (a trial the stopping rule no longer
            // wants after an earlier one of its group was accepted is not counted and not looked at)
            const int g = limit - it < 4 ? limit - it : 4;
            int idx4[4][6];
            for (int l = 0; l < 4; ++l) {
                if (l >= g) { std::memcpy(idx4[l], idx4[0], sizeof(idx4[0])); continue; }
                draw(idx4[l], 6);
            }
            double pose4[4][12];
            const int okmask = dlt_pose4(P, idx4, pose4);
            for (int l = 0; l < g; ++l) {
                if (!wanted(it)) return it;
                ++it;
                if ((okmask >> l) & 1) { consider(pose4[l]); continue; }
                // singular DLT: the six 3D points are (nearly) coplanar -- a box face, a card.  P3P on the first three, the root
                // chosen by the reprojection error of the other three (normalised coordinates)
                double poses[4][12];
                const int nsol = p3p_of(idx4[l], poses);
                int pick = -1;
                double pick_err = 1e300;
                for (int k = 0; k < nsol; ++k) {
                    double err = 0.0;
                    for (int j = 3; j < 6; ++j) {
                        const double* x = &P.X[3 * idx4[l][j]];
                        const double* ps = poses[k];
                        const double zc = ps[8] * x[0] + ps[9] * x[1] + ps[10] * x[2] + ps[11];
                        if (!(zc > 1e-12)) { err = 1e300; break; }
                        const double du = (ps[0] * x[0] + ps[1] * x[1] + ps[2] * x[2] + ps[3]) / zc - P.ray[2 * idx4[l][j]];
                        const double dv = (ps[4] * x[0] + ps[5] * x[1] + ps[6] * x[2] + ps[7]) / zc - P.ray[2 * idx4[l][j] + 1];
                        err += du * du + dv * dv;
                    }
                    if (err < pick_err) { pick_err = err; pick = k; }
                }
                if (pick >= 0) consider(poses[pick]);
            }
        }
        return it;
    }

    // chunks after the unconditional ones, one at a time, until the confidence criterion or max_iters stops them
    void run_tail(Candidate& best, FloatBound bound, int* iters_run) const {
        int c = full_chunks();
        long long total = chunk_start(c);
        const int rem = (int)(floor_trials() - total);
        while (total < max_iters) {
            const int limit = (int)((max_iters - total) < chunk_len(c) ? (max_iters - total) : chunk_len(c));
            const int ran = run_chunk(c, limit, c == full_chunks() ? rem : 0, best, bound);
            total += ran;
            if (ran < limit) break;
            ++c;
        }
        if (iters_run) *iters_run = (int)total;
    }

    // local optimisation of the winner: LM on the inliers, re-evaluate the inlier set, LM again
    int finish(Candidate& best, double* pose_out, unsigned char* inlier_mask, int* n_inliers) {
        const int need = solver == 1 ? 4 : 6;              // inliers a pose must keep: the sample and, for P3P, the point that picked the root
        if (best.cnt < need) return 1;
        std::vector<unsigned char> mask;
        for (int round = 0; round < 2; ++round) {
            refine_lm(P, best.mask.data(), best.pose, 20);
            mask = best.mask;
            best.cnt = count_inliers(P, best.pose, thr2, best.mask.data(), nullptr);
            if (best.cnt < need || mask == best.mask) break;  // same inlier set: the pose is already its optimum
        }
        std::memcpy(pose_out, best.pose, sizeof(best.pose));
        if (inlier_mask) std::memcpy(inlier_mask, best.mask.data(), (size_t)n);
        if (n_inliers) *n_inliers = best.cnt;
        return best.cnt >= need ? 0 : 1;
    }
};

void Ransac::reorder_after_chunk0(const Candidate& c0) {
    if (c0.cnt <= 0 || (int)c0.mask.size() != n || c0.cnt == n) return;
    G.n = n; G.x.resize(n); G.y.resize(n); G.z.resize(n); G.u.resize(n); G.v.resize(n);
    int k = 0;
    for (int pass = 0; pass < 2; ++pass)                  // the rejected points first, both groups in their original order
        for (int i = 0; i < n; ++i)
            if ((c0.mask[i] != 0) == (pass == 1)) {
                G.x[k] = F.x[i]; G.y[k] = F.y[i]; G.z[k] = F.z[i]; G.u[k] = F.u[i]; G.v[k] = F.v[i];
                ++k;
            }
    reordered = true;
}

const double kIdentPose[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

}  // namespace

extern "C" int oppnp_abi_version(void) { return 3; }          // 3: + oppnp_p3p4

// Fewest correspondences that give a pose: the 6-point DLT needs its sample; P3P + RANSAC (the pycolmap branch, metric_utils.py:155-165)
// needs the three of a sample and ONE more point to pick the root, so frames with 4 or 5 matches still get a pose there.
static int min_points(int solver) { return solver == 1 ? 4 : 6; }

// minimal solver alone (tests): rays = three normalised image points, X = three world points -> up to four poses [R | t]
extern "C" int oppnp_p3p(const double* rays3x2, const double* X3x3, double* poses4x12) {
    if (!rays3x2 || !X3x3 || !poses4x12) return -1;
    return p3p_poses(reinterpret_cast<const double(*)[2]>(rays3x2), reinterpret_cast<const double(*)[3]>(X3x3),
                     reinterpret_cast<double(*)[12]>(poses4x12));
}

extern "C" void oppnp_p3p4(const double* rays4x3x2, const double* X4x3x3, double* poses4x4x12, int* nsol4) {
    p3p_poses4(reinterpret_cast<const double (*)[3][2]>(rays4x3x2), reinterpret_cast<const double (*)[3][3]>(X4x3x3),
               reinterpret_cast<double (*)[4][12]>(poses4x4x12), nsol4);
}

extern "C" int oppnp_ransac(const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px, double confidence,
                            int min_iters, int max_iters, unsigned long long seed, int solver, double* pose_out, unsigned char* inlier_mask,
                            int* n_inliers, int* iters_run) {
    if (!K || !pose_out || (n > 0 && (!pts2d || !pts3d)) || max_iters < 1 || min_iters < 0 || (solver != 0 && solver != 1)) return -1;
    std::memcpy(pose_out, kIdentPose, sizeof(kIdentPose));
    if (inlier_mask && n > 0) std::memset(inlier_mask, 0, (size_t)n);
    if (n_inliers) *n_inliers = 0;
    if (iters_run) *iters_run = 0;
    if (n < min_points(solver)) return 1;                 // too few correspondences: identity pose, no inliers
    Ransac R;
    R.setup(K, pts2d, pts3d, n, reproj_err_px, confidence, min_iters, max_iters, seed, solver);
    Candidate best;
    FloatBound b0;                                        // chunk 0's float best bounds every later chunk
    for (int c = 0; c < R.full_chunks(); ++c) {
        Candidate local;
        FloatBound b = b0;
        R.run_chunk(c, chunk_len(c), chunk_len(c), local, c == 0 ? b0 : b);
        if (c == 0) R.reorder_after_chunk0(local);
        if (local.better_than(best)) best = std::move(local);
    }
    R.run_tail(best, b0, iters_run);
    return R.finish(best, pose_out, inlier_mask, n_inliers);
}

// ---- asynchronous pool: poses are solved on library-owned host threads (no Python / GIL on the per-frame path); the
//      unconditional chunks of a frame are separate tasks, so one frame's 10 000-trial floor spreads over every thread ------
namespace {

struct Result {
    double pose[12];
    int n_inliers, rc;
};

struct Job {
    Ransac R;
    long long ticket = 0;
    std::vector<Candidate> chunk_best;      // one per unconditional chunk
    FloatBound b0;                           // chunk 0's float best: the bound of every other chunk
    int remaining = 0;                       // unconditional chunks not finished yet (guarded by Pool::mu)
};

// The submitting thread is the one that feeds the GPU; the workers run at background priority.  A mutex shared by the two inverts that
// priority: a worker that is descheduled while it holds the lock (a loaded host: other tenants on its CPU) stops the feeder for
// milliseconds -- seen as `finish` at 260-320 us per frame instead of 100 in one 20-step region out of ten, 20 % of the frame rate.  So
// submit takes NO lock the workers take: the job goes onto a lock-free inbox (an atomic singly linked list) and a semaphore counts the
// tasks that exist (inbox entries + queue entries); a worker that has taken a count moves the inbox into the queue under `mu` -- which only
// workers and the joining calls (wait_all, result) hold -- and pops one task.
struct Inbox {
    std::shared_ptr<Job> job;            // null: `ident` (a frame with too few correspondences: identity pose, recorded by a worker)
    int chunk = 0;
    long long ticket = 0;
    Inbox* next = nullptr;
};

struct Pool {
    std::vector<std::thread> workers;
    std::deque<std::pair<std::shared_ptr<Job>, int>> queue;      // (job, chunk); chunk -1: the whole tail (no full chunks); guarded by mu
    std::unordered_map<long long, Result> results;     // by ticket; erased when read
    std::mutex mu;
    std::condition_variable cv_done;
    std::atomic<Inbox*> inbox{nullptr};
    sem_t tasks;                                       // inbox entries + queue entries not yet taken by a worker
    std::atomic<long long> submitted{0};
    long long finished = 0;
    std::atomic<bool> stop{false};

    explicit Pool(int n) {
        sem_init(&tasks, 0, 0);
        for (int i = 0; i < n; ++i) workers.emplace_back([this, i] { run(i); });
    }
    ~Pool() {
        stop.store(true);
        for (size_t i = 0; i < workers.size(); ++i) sem_post(&tasks);
        for (auto& t : workers) t.join();
        for (Inbox* e = inbox.exchange(nullptr); e;) { Inbox* nx = e->next; delete e; e = nx; }
        sem_destroy(&tasks);
    }
    void post(Inbox* e) {                    // any thread, no lock
        Inbox* head = inbox.load(std::memory_order_relaxed);
        do { e->next = head; } while (!inbox.compare_exchange_weak(head, e, std::memory_order_release, std::memory_order_relaxed));
        sem_post(&tasks);
    }
    void drain_inbox_locked() {              // mu held: inbox entries, oldest first, to the back of the queue
        Inbox* e = inbox.exchange(nullptr, std::memory_order_acquire);
        Inbox* rev = nullptr;
        while (e) { Inbox* nx = e->next; e->next = rev; rev = e; e = nx; }
        while (rev) {
            Inbox* nx = rev->next;
            if (rev->job) {
                queue.emplace_back(std::move(rev->job), rev->chunk);
            } else {                         // no pose: identity, recorded at once; the count taken for it is given back as a no-op task
                Result r;
                std::memcpy(r.pose, kIdentPose, sizeof(kIdentPose));
                r.n_inliers = 0; r.rc = 1;
                results[rev->ticket] = r;
                ++finished;
                queue.emplace_back(std::shared_ptr<Job>(), -2);
                cv_done.notify_all();
            }
            delete rev;
            rev = nx;
        }
    }
    void finish_job(Job& job) {              // merge in chunk order, sequential tail, refinement
        Candidate best;
        for (auto& c : job.chunk_best)
            if (c.better_than(best)) best = std::move(c);
        job.R.run_tail(best, job.b0, nullptr);
        Result r;
        std::memcpy(r.pose, kIdentPose, sizeof(kIdentPose));
        r.n_inliers = 0;
        r.rc = job.R.finish(best, r.pose, nullptr, &r.n_inliers);
        {
            std::lock_guard<std::mutex> lk(mu);
            results[job.ticket] = r;
            ++finished;
        }
        cv_done.notify_all();
    }
    // OPPNP_WORKER_CPUS="c0,c1,c2,..." (+ OPPNP_WORKER_CPUS_PER=k, default 1): worker i may run on CPUs [i k, (i + 1) k) of the list (taken
    // modulo its length) instead of everywhere the process may -- background threads that all wake on the same few CPUs are spread by the
    // scheduler's load balancer only slowly (their weight is a tenth of a normal thread's), and a frame's pose then waits for one CPU
    static void pin_worker(int index) {
#if defined(__linux__)
        const char* lst = getenv("OPPNP_WORKER_CPUS");
        if (!lst || !*lst) return;
        std::vector<int> cpus;
        for (const char* p = lst; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            cpus.push_back((int)v);
            p = *end ? end + 1 : end;
        }
        if (cpus.empty()) return;
        const char* per = getenv("OPPNP_WORKER_CPUS_PER");
        const int k = per && atoi(per) > 0 ? atoi(per) : 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int j = 0; j < k; ++j) {
            const int c = cpus[((size_t)index * k + j) % cpus.size()];
            if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
        }
        (void)sched_setaffinity(0, sizeof(set), &set);
#else
        (void)index;
#endif
    }
    void run(int index) {
        pin_worker(index);
#if defined(__linux__)
        // background priority: the thread that feeds the GPU must never wait for a core behind a RANSAC chunk
        // (OPPNP_WORKER_NICE=n: another nice value; OPPNP_WORKER_IDLE=1: SCHED_IDLE -- a worker then runs only on a CPU nothing else wants)
        const char* nv = getenv("OPPNP_WORKER_NICE");
        (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nv && *nv ? atoi(nv) : 10);
        if (env_on("OPPNP_WORKER_IDLE")) {
            struct sched_param sp;
            std::memset(&sp, 0, sizeof(sp));
            (void)sched_setscheduler(0, SCHED_IDLE, &sp);
        }
#endif
        for (;;) {
            std::pair<std::shared_ptr<Job>, int> task;
            while (sem_wait(&tasks) != 0) {}                 // (EINTR)
            {
                std::lock_guard<std::mutex> lk(mu);
                drain_inbox_locked();
                if (queue.empty()) {                          // only the destructor posts without a task
                    if (stop.load()) return;
                    continue;
                }
                task = std::move(queue.front());
                queue.pop_front();
            }
            if (task.second == -2) continue;                  // the place-holder of an identity result
            Job& job = *task.first;
            if (task.second < 0) { finish_job(job); continue; }
            if (task.second == 0) {                  // chunk 0 first, alone: its float best then bounds the others, which start now
                job.R.run_chunk(0, CH0, CH0, job.chunk_best[0], job.b0);
                job.R.reorder_after_chunk0(job.chunk_best[0]);          // before any other chunk of this job exists
                const int nfull = (int)job.chunk_best.size();
                if (nfull > 1) {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        --job.remaining;
                        for (int c = 1; c < nfull; ++c) queue.emplace_back(task.first, c);
                    }
                    for (int c = 1; c < nfull; ++c) sem_post(&tasks);
                    continue;
                }
            } else {
                FloatBound b = job.b0;
                job.R.run_chunk(task.second, CH, CH, job.chunk_best[(size_t)task.second], b);
            }
            bool last;
            {
                std::lock_guard<std::mutex> lk(mu);
                last = --job.remaining == 0;
            }
            if (last) finish_job(job);
        }
    }
};

}  // namespace

extern "C" void* oppnp_pool_create(int threads) { return new Pool(threads < 1 ? 1 : threads); }

extern "C" void oppnp_pool_destroy(void* pool) { delete reinterpret_cast<Pool*>(pool); }

// copies the inputs and returns a ticket (0, 1, 2, ...) at once
extern "C" long long oppnp_pool_submit(void* pool_, const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px,
                                       double confidence, int min_iters, int max_iters, unsigned long long seed, int solver) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    if (!pool || !K || n < 0 || (n > 0 && (!pts2d || !pts3d)) || max_iters < 1 || min_iters < 0 || (solver != 0 && solver != 1)) return -1;
    Inbox* e = new Inbox();
    if (n < min_points(solver)) {                         // no pose: identity, recorded by the worker that takes the entry
        e->ticket = pool->submitted.fetch_add(1);
        const long long ticket = e->ticket;
        pool->post(e);
        return ticket;
    }
    auto job = std::make_shared<Job>();
    job->R.setup(K, pts2d, pts3d, n, reproj_err_px, confidence, min_iters, max_iters, seed, solver);
    const int nfull = job->R.full_chunks();
    job->chunk_best.resize((size_t)nfull);
    job->remaining = nfull;
    const long long ticket = job->ticket = e->ticket = pool->submitted.fetch_add(1);
    e->chunk = nfull == 0 ? -1 : 0;                       // the other chunks follow chunk 0
    e->job = std::move(job);
    pool->post(e);                                        // no lock shared with the workers: see Pool
    return ticket;
}

// blocks until every submitted job is finished; returns their number
extern "C" long long oppnp_pool_wait_all(void* pool_) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    std::unique_lock<std::mutex> lk(pool->mu);
    pool->cv_done.wait(lk, [pool] { return pool->finished == pool->submitted.load(); });
    return pool->finished;
}

extern "C" int oppnp_pool_result(void* pool_, long long ticket, double* pose_out, int* n_inliers) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    std::lock_guard<std::mutex> lk(pool->mu);
    auto itr = pool->results.find(ticket);
    if (itr == pool->results.end()) return -1;          // unknown, unfinished or already read
    const Result r = itr->second;
    pool->results.erase(itr);                            // a long frame loop does not accumulate results
    if (pose_out) std::memcpy(pose_out, r.pose, sizeof(r.pose));
    if (n_inliers) *n_inliers = r.n_inliers;
    return r.rc;
}

// ---- 2D affinity by RANSAC: the detector's box estimate -----------------------------------------------------------------------
// Replaces cv2.estimateAffine2D(mkpts0, mkpts1, method=cv2.RANSAC, ransacReprojThreshold=6) of the reference's match_worker
// (src/local_feature_object_detector/local_feature_2D_detector.py:120-122).  OpenCV is absent: own estimator, same model (six
// parameters, dst = A src + t), same inlier rule (distance below the threshold), OpenCV's defaults as defaults (2 000 trials at
// most, confidence 0.99); minimal samples of three non-collinear points, adaptive stop, least squares on the winning inlier set.
// Deterministic for a seed.  Parity with OpenCV's sampling sequence is unpinned.
namespace {
bool affine_from3(const float* s, const float* d, const int* idx, double* A) {
    const double x0 = s[2 * idx[0]], y0 = s[2 * idx[0] + 1], x1 = s[2 * idx[1]], y1 = s[2 * idx[1] + 1], x2 = s[2 * idx[2]], y2 = s[2 * idx[2] + 1];
    const double det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    const double scale = std::fabs(x1 - x0) + std::fabs(y1 - y0) + std::fabs(x2 - x0) + std::fabs(y2 - y0);
    if (!(std::fabs(det) > 1e-9 * scale * scale) || !(scale > 0.0)) return false;
    for (int r = 0; r < 2; ++r) {
        const double u0 = d[2 * idx[0] + r], u1 = d[2 * idx[1] + r], u2 = d[2 * idx[2] + r];
        const double a = ((u1 - u0) * (y2 - y0) - (u2 - u0) * (y1 - y0)) / det;
        const double b = ((x1 - x0) * (u2 - u0) - (x2 - x0) * (u1 - u0)) / det;
        A[3 * r] = a; A[3 * r + 1] = b; A[3 * r + 2] = u0 - a * x0 - b * y0;
    }
    return true;
}
int affine_inliers(const float* s, const float* d, int n, const double* A, double thr2, unsigned char* mask) {
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const double ex = A[0] * s[2 * i] + A[1] * s[2 * i + 1] + A[2] - d[2 * i], ey = A[3] * s[2 * i] + A[4] * s[2 * i + 1] + A[5] - d[2 * i + 1];
        const bool in = ex * ex + ey * ey < thr2;
        if (mask) mask[i] = in ? 1 : 0;
        cnt += in;
    }
    return cnt;
}
}  // namespace

extern "C" int oppnp_estimate_affine2d(const float* src, const float* dst, int n, double reproj_thr, int max_iters, double confidence,
                                       unsigned long long seed, double* affine2x3, unsigned char* inlier_mask, int* n_inliers) {
    if (!affine2x3 || (n > 0 && (!src || !dst)) || max_iters < 1 || !(reproj_thr > 0.0)) return -1;
    const double ident[6] = {1, 0, 0, 0, 1, 0};
    std::memcpy(affine2x3, ident, sizeof(ident));
    if (inlier_mask && n > 0) std::memset(inlier_mask, 0, (size_t)n);
    if (n_inliers) *n_inliers = 0;
    if (n < 3) return 1;
    Rng rng(seed);
    const double thr2 = reproj_thr * reproj_thr;
    std::vector<unsigned char> mask((size_t)n), best_mask((size_t)n, 0);
    int best = 0, needed = max_iters;
    for (int it = 0; it < needed && it < max_iters; ++it) {
        int idx[3];
        for (int k = 0; k < 3;) {
            const int c = rng.below(n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= idx[j] == c;
            if (!dup) idx[k++] = c;
        }
        double A[6];
        if (!affine_from3(src, dst, idx, A)) continue;
        const int cnt = affine_inliers(src, dst, n, A, thr2, mask.data());
        if (cnt > best) {
            best = cnt;
            best_mask = mask;
            const double w = (double)cnt / n, pw = w * w * w;
            if (pw > 1.0 - 1e-12) needed = 1;
            else if (pw > 1e-12) {
                const double nd = std::ceil(std::log(1.0 - confidence) / std::log(1.0 - pw));      // clamped before the conversion (see Ransac::needed_for)
                needed = nd < (double)max_iters ? (int)nd : max_iters;
            } else needed = max_iters;
        }
    }
    if (best < 3) return 1;
    // least squares on the inliers: normal equations of [x y 1] shared by both rows
    double S[9] = {0}, bu[3] = {0}, bv[3] = {0};
    for (int i = 0; i < n; ++i) {
        if (!best_mask[i]) continue;
        const double p[3] = {src[2 * i], src[2 * i + 1], 1.0};
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) S[a * 3 + b] += p[a] * p[b];
            bu[a] += p[a] * dst[2 * i];
            bv[a] += p[a] * dst[2 * i + 1];
        }
    }
    if (std::fabs(det3(S)) < 1e-12) return 1;
    double Si[9];
    inv3(S, Si);
    for (int a = 0; a < 3; ++a) {
        affine2x3[a] = Si[a * 3] * bu[0] + Si[a * 3 + 1] * bu[1] + Si[a * 3 + 2] * bu[2];
        affine2x3[3 + a] = Si[a * 3] * bv[0] + Si[a * 3 + 1] * bv[1] + Si[a * 3 + 2] * bv[2];
    }
    if (inlier_mask) std::memcpy(inlier_mask, best_mask.data(), (size_t)n);
    if (n_inliers) *n_inliers = best;
    return 0;
}
