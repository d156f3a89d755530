// Deterministic host PnP + RANSAC (double precision, no dependencies).  C ABI: include/onepose_pnp.h.
//
// Replaces the call the reference makes after the matcher -- src/utils/metric_utils.py:121-209 `ransac_PnP`, which
// delegates to pycolmap.absolute_pose_estimation (P3P LO-RANSAC + non-linear refinement, max_error_px 7 in
// inference.py:181-189) or cv2.solvePnPRansac (EPnP).  Neither library exists here and the reference pins no pose
// outputs, so this is NOT a restatement of their internals ("parity unpinned"): it is the build's own estimator,
// applied identically to the HIP path's matches and to the oracle's matches so that pose parity is checkable.
//
//   hypotheses : 6-point DLT on calibrated rays (smallest eigenvector of the 12x12 normal matrix by cyclic Jacobi),
//                projected to SO(3) by Newton polar iteration, cheirality check
//   scoring    : reprojection error < threshold (pixels), adaptive stopping at the requested confidence
//   refinement : Levenberg-Marquardt on the inliers (6 dof, analytic Jacobian), inlier set re-evaluated once
//   randomness : xorshift64* seeded by the caller -> bit-reproducible
#include <cmath>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

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
bool polar_rotation(double* R) {
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
        if (diff < 1e-15) break;
    }
    return true;
}

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

// 6-point DLT on normalised rays -> pose; false when degenerate
bool dlt_pose(const Problem& P, const int* idx, int m, double* pose) {
    // Hartley-style conditioning of the 3D points
    double mu[3] = {0, 0, 0};
    for (int k = 0; k < m; ++k)
        for (int d = 0; d < 3; ++d) mu[d] += P.X[3 * idx[k] + d];
    for (int d = 0; d < 3; ++d) mu[d] /= m;
    double sc = 0.0;
    for (int k = 0; k < m; ++k) {
        double r2 = 0.0;
        for (int d = 0; d < 3; ++d) { const double v = P.X[3 * idx[k] + d] - mu[d]; r2 += v * v; }
        sc += std::sqrt(r2);
    }
    if (sc < 1e-300) return false;
    sc = std::sqrt(3.0) * m / sc;
    double A[144];
    std::memset(A, 0, sizeof(A));
    for (int k = 0; k < m; ++k) {
        const int i = idx[k];
        const double Xh[4] = {(P.X[3 * i] - mu[0]) * sc, (P.X[3 * i + 1] - mu[1]) * sc, (P.X[3 * i + 2] - mu[2]) * sc, 1.0};
        const double x = P.ray[2 * i], y = P.ray[2 * i + 1];
        double r1[12], r2[12];
        for (int j = 0; j < 4; ++j) {
            r1[j] = Xh[j]; r1[4 + j] = 0.0; r1[8 + j] = -x * Xh[j];
            r2[j] = 0.0; r2[4 + j] = Xh[j]; r2[8 + j] = -y * Xh[j];
        }
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) A[a * 12 + b] += r1[a] * r1[b] + r2[a] * r2[b];
    }
    double V[144];
    jacobi_eig<12>(A, V);
    int best = 0;
    for (int j = 1; j < 12; ++j)
        if (A[j * 12 + j] < A[best * 12 + best]) best = j;
    double p[12];
    for (int j = 0; j < 12; ++j) p[j] = V[j * 12 + best];
    double M[9] = {p[0], p[1], p[2], p[4], p[5], p[6], p[8], p[9], p[10]};
    double t[3] = {p[3], p[7], p[11]};
    double d = det3(M);
    if (std::fabs(d) < 1e-18) return false;
    if (d < 0) { for (double& v : M) v = -v; for (double& v : t) v = -v; d = -d; }
    const double s = std::cbrt(d);
    for (double& v : M) v /= s;
    for (double& v : t) v /= s;
    if (!polar_rotation(M)) return false;
    // undo the conditioning: X_n = sc (X - mu)  =>  R X_n + t = (sc R) X + (t - sc R mu)
    for (int rI = 0; rI < 3; ++rI) {
        const double tr = t[rI] - sc * (M[rI * 3] * mu[0] + M[rI * 3 + 1] * mu[1] + M[rI * 3 + 2] * mu[2]);
        pose[rI * 4] = M[rI * 3]; pose[rI * 4 + 1] = M[rI * 3 + 1]; pose[rI * 4 + 2] = M[rI * 3 + 2];
        pose[rI * 4 + 3] = tr / sc;
    }
    return true;
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
    const double fx = P.K[0], fy = P.K[4], sk = P.K[1];
    double lambda = 1e-3;
    auto cost_of = [&](const double* ps) {
        double c = 0.0;
        for (int i = 0; i < P.n; ++i) {
            if (!mask[i]) continue;
            const double* x = &P.X[3 * i];
            const double xc = ps[0] * x[0] + ps[1] * x[1] + ps[2] * x[2] + ps[3];
            const double yc = ps[4] * x[0] + ps[5] * x[1] + ps[6] * x[2] + ps[7];
            const double zc = ps[8] * x[0] + ps[9] * x[1] + ps[10] * x[2] + ps[11];
            if (zc <= 1e-12) { c += 1e12; continue; }
            const double xn = xc / zc, yn = yc / zc;
            const double du = fx * xn + sk * yn + P.K[2] - P.px[2 * i], dv = fy * yn + P.K[5] - P.px[2 * i + 1];
            c += du * du + dv * dv;
        }
        return c;
    };
    double cur = cost_of(pose);
    for (int it = 0; it < iters; ++it) {
        double H[36], g[6];
        std::memset(H, 0, sizeof(H));
        std::memset(g, 0, sizeof(g));
        for (int i = 0; i < P.n; ++i) {
            if (!mask[i]) continue;
            const double* x = &P.X[3 * i];
            const double pc[3] = {pose[0] * x[0] + pose[1] * x[1] + pose[2] * x[2] + pose[3],
                                  pose[4] * x[0] + pose[5] * x[1] + pose[6] * x[2] + pose[7],
                                  pose[8] * x[0] + pose[9] * x[1] + pose[10] * x[2] + pose[11]};
            if (pc[2] <= 1e-12) continue;
            const double iz = 1.0 / pc[2], xn = pc[0] * iz, yn = pc[1] * iz;
            const double ru = fx * xn + sk * yn + P.K[2] - P.px[2 * i], rv = fy * yn + P.K[5] - P.px[2 * i + 1];
            // d(u,v)/d(pc)
            const double Ju[3] = {fx * iz, sk * iz, -(fx * xn + sk * yn) * iz};
            const double Jv[3] = {0.0, fy * iz, -fy * yn * iz};
            // d(pc)/d(w) = -[pc - t]_x ... with the left increment R' = exp(w) R: pc' = exp(w) (pc - t) + t + dt
            const double q[3] = {pc[0] - pose[3], pc[1] - pose[7], pc[2] - pose[11]};
            const double dW[3][3] = {{0, q[2], -q[1]}, {-q[2], 0, q[0]}, {q[1], -q[0], 0}};
            double ju[6], jv[6];
            for (int k = 0; k < 3; ++k) {
                ju[k] = Ju[0] * dW[0][k] + Ju[1] * dW[1][k] + Ju[2] * dW[2][k];
                jv[k] = Jv[0] * dW[0][k] + Jv[1] * dW[1][k] + Jv[2] * dW[2][k];
                ju[3 + k] = Ju[k];
                jv[3 + k] = Jv[k];
            }
            for (int a = 0; a < 6; ++a) {
                g[a] -= ju[a] * ru + jv[a] * rv;
                for (int b2 = a; b2 < 6; ++b2) H[a * 6 + b2] += ju[a] * ju[b2] + jv[a] * jv[b2];      // upper triangle
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

extern "C" int oppnp_abi_version(void) { return 1; }

extern "C" int oppnp_ransac(const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px, double confidence,
                            int min_iters, int max_iters, unsigned long long seed, double* pose_out, unsigned char* inlier_mask,
                            int* n_inliers, int* iters_run) {
    if (!K || !pose_out || (n > 0 && (!pts2d || !pts3d))) return -1;
    static const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::memcpy(pose_out, ident, sizeof(ident));
    if (inlier_mask && n > 0) std::memset(inlier_mask, 0, (size_t)n);
    if (n_inliers) *n_inliers = 0;
    if (iters_run) *iters_run = 0;
    if (n < 6) return 1;                                  // too few correspondences: identity pose, no inliers
    Problem P;
    P.n = n; P.K = K;
    P.ray.resize(2 * (size_t)n); P.px.resize(2 * (size_t)n); P.X.resize(3 * (size_t)n);
    double Ki[9];
    inv3(K, Ki);
    for (int i = 0; i < n; ++i) {
        const double u = pts2d[2 * i], v = pts2d[2 * i + 1];
        P.px[2 * i] = u; P.px[2 * i + 1] = v;
        const double w = Ki[6] * u + Ki[7] * v + Ki[8];
        P.ray[2 * i] = (Ki[0] * u + Ki[1] * v + Ki[2]) / w;
        P.ray[2 * i + 1] = (Ki[3] * u + Ki[4] * v + Ki[5]) / w;
        for (int d = 0; d < 3; ++d) P.X[3 * i + d] = pts3d[3 * i + d];
    }
    const double thr2 = reproj_err_px * reproj_err_px;
    Rng rng(seed);
    std::vector<unsigned char> mask((size_t)n), best_mask((size_t)n, 0);
    double best_pose[12];
    std::memcpy(best_pose, ident, sizeof(ident));
    int best_cnt = 0;
    double best_cost = 1e300;
    int needed = max_iters, it = 0;
    for (; it < max_iters && (it < min_iters || it < needed); ++it) {
        int idx[6];
        for (int k = 0; k < 6;) {
            const int c = rng.below(n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= idx[j] == c;
            if (!dup) idx[k++] = c;
        }
        double pose[12];
        if (!dlt_pose(P, idx, 6, pose)) continue;
        double cost;
        const int cnt = count_inliers(P, pose, thr2, mask.data(), &cost);
        if (cnt > best_cnt || (cnt == best_cnt && cost < best_cost)) {
            best_cnt = cnt; best_cost = cost;
            std::memcpy(best_pose, pose, sizeof(pose));
            best_mask = mask;
            const double w = (double)cnt / n;
            const double pw = std::pow(w, 6.0);
            if (pw > 1.0 - 1e-12) needed = 1;
            else if (pw > 1e-12) needed = (int)std::ceil(std::log(1.0 - confidence) / std::log(1.0 - pw));
        }
    }
    if (iters_run) *iters_run = it;
    if (best_cnt < 6) return 1;
    // local optimisation: LM on the inliers, re-evaluate the inlier set, LM again
    for (int round = 0; round < 2; ++round) {
        refine_lm(P, best_mask.data(), best_pose, 20);
        mask = best_mask;
        best_cnt = count_inliers(P, best_pose, thr2, best_mask.data(), nullptr);
        if (best_cnt < 6 || mask == best_mask) break;     // same inlier set: the pose is already its optimum
    }
    std::memcpy(pose_out, best_pose, sizeof(best_pose));
    if (inlier_mask) std::memcpy(inlier_mask, best_mask.data(), (size_t)n);
    if (n_inliers) *n_inliers = best_cnt;
    return best_cnt >= 6 ? 0 : 1;
}

// ---- asynchronous pool: poses are solved on library-owned host threads (no Python / GIL on the per-frame path) --------
namespace {

struct Job {
    double K[9];
    std::vector<float> p2, p3;
    double thr, conf;
    int min_it, max_it;
    unsigned long long seed;
    long long ticket;
};
struct Result {
    double pose[12];
    int n_inliers, rc;
};

struct Pool {
    std::vector<std::thread> workers;
    std::deque<Job> queue;
    std::vector<Result> results;          // indexed by ticket
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    long long submitted = 0, finished = 0;
    bool stop = false;

    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers.emplace_back([this] { run(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [this] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                job = std::move(queue.front());
                queue.pop_front();
            }
            Result r;
            const int n = (int)(job.p2.size() / 2);
            r.rc = oppnp_ransac(job.K, job.p2.data(), job.p3.data(), n, job.thr, job.conf, job.min_it, job.max_it, job.seed, r.pose,
                                nullptr, &r.n_inliers, nullptr);
            {
                std::lock_guard<std::mutex> lk(mu);
                results[(size_t)job.ticket] = r;
                ++finished;
            }
            cv_done.notify_all();
        }
    }
};

}  // namespace

extern "C" void* oppnp_pool_create(int threads) { return new Pool(threads < 1 ? 1 : threads); }

extern "C" void oppnp_pool_destroy(void* pool) { delete reinterpret_cast<Pool*>(pool); }

// copies the inputs and returns a ticket (0, 1, 2, ...) at once
extern "C" long long oppnp_pool_submit(void* pool_, const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px,
                                       double confidence, int min_iters, int max_iters, unsigned long long seed) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    if (!pool || !K || n < 0 || (n > 0 && (!pts2d || !pts3d))) return -1;
    Job job;
    std::memcpy(job.K, K, sizeof(job.K));
    job.p2.assign(pts2d, pts2d + 2 * (size_t)n);
    job.p3.assign(pts3d, pts3d + 3 * (size_t)n);
    job.thr = reproj_err_px; job.conf = confidence; job.min_it = min_iters; job.max_it = max_iters; job.seed = seed;
    {
        std::lock_guard<std::mutex> lk(pool->mu);
        job.ticket = pool->submitted++;
        pool->results.resize((size_t)pool->submitted);
        pool->queue.push_back(std::move(job));
    }
    pool->cv_work.notify_one();
    return pool->submitted - 1;
}

// blocks until every submitted job is finished; returns their number
extern "C" long long oppnp_pool_wait_all(void* pool_) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    std::unique_lock<std::mutex> lk(pool->mu);
    pool->cv_done.wait(lk, [pool] { return pool->finished == pool->submitted; });
    return pool->finished;
}

extern "C" int oppnp_pool_result(void* pool_, long long ticket, double* pose_out, int* n_inliers) {
    Pool* pool = reinterpret_cast<Pool*>(pool_);
    std::lock_guard<std::mutex> lk(pool->mu);
    if (ticket < 0 || ticket >= (long long)pool->results.size()) return -1;
    const Result& r = pool->results[(size_t)ticket];
    if (pose_out) std::memcpy(pose_out, r.pose, sizeof(r.pose));
    if (n_inliers) *n_inliers = r.n_inliers;
    return r.rc;
}
