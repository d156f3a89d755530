/* libonepose_pnp.so -- host (CPU) PnP + RANSAC used after the 2D-3D matcher.  C ABI, no dependencies.
 *
 * Replaces the reference's `ransac_PnP` (src/utils/metric_utils.py:121-209; called at inference.py:181-189 with
 * pnp_reprojection_error = 7, and at :328-336), which delegates to pycolmap / OpenCV.  north_star keeps PnP on the
 * host.  The estimator is the build's own (P3P or 6-point DLT hypotheses, RANSAC with the reference branch's trial policy,
 * Levenberg-Marquardt refinement; deterministic for a given seed) -- see onepose_st_amd/csrc_host/pnp.cpp.
 * `solver`: 1 = P3P minimal samples, every root scored (what pycolmap.absolute_pose_estimation runs: metric_utils.py:155-165,
 * the `use_pycolmap_ransac=True` branch); 0 = 6-point DLT with a P3P fallback for coplanar samples (the OpenCV branch, :188-196).
 */
#ifndef ONEPOSE_PNP_H
#define ONEPOSE_PNP_H
#ifdef __cplusplus
extern "C" {
#endif

int oppnp_abi_version(void);

/* K: 3x3 intrinsics, row-major doubles.  pts2d [n][2] pixels, pts3d [n][3] (floats, as produced by the matcher).
 * pose_out: 3x4 row-major [R | t] (world -> camera).  inlier_mask: n bytes (may be NULL).
 * Returns 0 on success, 1 when no pose was found (fewer than 6 correspondences / inliers: pose_out = identity, like the
 * reference's cv2.error branch), -1 on invalid arguments. */
int oppnp_ransac(const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px, double confidence,
                 int min_iters, int max_iters, unsigned long long seed, int solver, double* pose_out, unsigned char* inlier_mask,
                 int* n_inliers, int* iters_run);

/* The P3P minimal solver alone: rays3x2 = three normalised image points (x, y) = K^-1 (u, v, 1), X3x3 = their world points;
 * poses4x12 receives up to four [R | t] (row-major); returns their number. */
int oppnp_p3p(const double* rays3x2, const double* X3x3, double* poses4x12);
/* The same solver on four samples at once (what the RANSAC loop of `solver` 1 calls; ABI 3): rays4x3x2 / X4x3x3 = four samples as above,
 * poses4x4x12 receives up to four poses per sample, nsol4 their numbers.  Same roots, same order, same poses as oppnp_p3p to rounding. */
void oppnp_p3p4(const double* rays4x3x2, const double* X4x3x3, double* poses4x4x12, int* nsol4);

/* Asynchronous pool: library-owned host threads (background priority) solve poses while the caller keeps feeding the GPU (the
 * per-frame path has no Python in it).  submit copies its inputs and returns a ticket 0, 1, 2, ... without taking any lock the
 * workers take (it may be called from several threads); wait_all blocks until every submitted pose is solved; result reads one
 * (return value as oppnp_ransac; -1 for a ticket that is unknown, unfinished or already read). */
void* oppnp_pool_create(int threads);
void oppnp_pool_destroy(void* pool);
long long oppnp_pool_submit(void* pool, const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px,
                            double confidence, int min_iters, int max_iters, unsigned long long seed, int solver);
long long oppnp_pool_wait_all(void* pool);
int oppnp_pool_result(void* pool, long long ticket, double* pose_out, int* n_inliers);

/* The detector's 2D affinity: replaces cv2.estimateAffine2D(src, dst, method=RANSAC, ransacReprojThreshold=thr)
 * (src/local_feature_object_detector/local_feature_2D_detector.py:120-122; OpenCV defaults: 2 000 trials at most, confidence 0.99).
 * src, dst [n][2] floats; affine2x3 row-major [a b tx; c d ty] with dst = A src + t; inlier_mask n bytes (may be NULL).
 * Returns 0, 1 when no model was found (fewer than 3 points / inliers: affine = identity), -1 on invalid arguments. */
int oppnp_estimate_affine2d(const float* src, const float* dst, int n, double reproj_thr, int max_iters, double confidence,
                            unsigned long long seed, double* affine2x3, unsigned char* inlier_mask, int* n_inliers);

#ifdef __cplusplus
}
#endif
#endif
