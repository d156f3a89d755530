/* libonepose_pnp.so -- host (CPU) PnP + RANSAC used after the 2D-3D matcher.  C ABI, no dependencies.
 *
 * Replaces the reference's `ransac_PnP` (src/utils/metric_utils.py:121-209; called at inference.py:181-189 with
 * pnp_reprojection_error = 7, and at :328-336), which delegates to pycolmap / OpenCV.  north_star keeps PnP on the
 * host.  The estimator is the build's own (6-point DLT hypotheses, adaptive RANSAC, Levenberg-Marquardt refinement;
 * deterministic for a given seed) -- see onepose_st_amd/csrc_host/pnp.cpp.
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
                 int min_iters, int max_iters, unsigned long long seed, double* pose_out, unsigned char* inlier_mask,
                 int* n_inliers, int* iters_run);

/* Asynchronous pool: library-owned host threads solve poses while the caller keeps feeding the GPU (the per-frame path
 * has no Python in it).  submit copies its inputs and returns a ticket 0, 1, 2, ...; wait_all blocks until every
 * submitted pose is solved; result reads one (return value as oppnp_ransac). */
void* oppnp_pool_create(int threads);
void oppnp_pool_destroy(void* pool);
long long oppnp_pool_submit(void* pool, const double* K, const float* pts2d, const float* pts3d, int n, double reproj_err_px,
                            double confidence, int min_iters, int max_iters, unsigned long long seed);
long long oppnp_pool_wait_all(void* pool);
int oppnp_pool_result(void* pool, long long ticket, double* pose_out, int* n_inliers);

#ifdef __cplusplus
}
#endif
#endif
