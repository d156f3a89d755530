"""Host PnP + RANSAC (C++): recovers planted poses, rejects outliers, is deterministic, handles degenerate input."""
import numpy as np
import pytest

from onepose_st_amd.pnp import PnPPool, ransac_PnP


def _scene(n, seed, noise_px=0.0, outlier_frac=0.0):
    rng = np.random.default_rng(seed)
    X = (rng.random((n, 3)) - 0.5) * np.array([0.2, 0.14, 0.1])
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = 0.5
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    t = np.array([0.01, -0.02, 0.45])
    K = np.array([[1216.0, 0, 320.0], [0, 1216.0, 240.0], [0, 0, 1]])
    pc = X @ R.T + t
    uv = (pc[:, :2] / pc[:, 2:]) * 1216.0 + np.array([320.0, 240.0])
    uv += noise_px * rng.normal(size=uv.shape)
    n_out = int(outlier_frac * n)
    if n_out:
        uv[:n_out] = rng.random((n_out, 2)) * np.array([640.0, 480.0])
    return K, uv.astype(np.float32), X.astype(np.float32), R, t, n_out


def _rot_err(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra @ Rb.T) - 1) / 2, -1, 1)))


def test_exact_data_recovers_pose():
    K, uv, X, R, t, _ = _scene(500, 0)
    pose, homo, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7)
    assert _rot_err(pose[:, :3], R) < 1e-3 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 1e-5
    assert len(inl) == 500 and homo.shape == (4, 4) and np.allclose(homo[3], [0, 0, 0, 1])


def test_noise_and_outliers():
    K, uv, X, R, t, n_out = _scene(2000, 1, noise_px=0.5, outlier_frac=0.3)
    pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7)
    assert _rot_err(pose[:, :3], R) < 0.05 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 2e-3
    assert len(inl) >= 0.95 * (2000 - n_out) and (inl >= n_out).mean() > 0.97
    pose2, _, inl2 = ransac_PnP(K, uv, X, pnp_reprojection_error=7)                  # deterministic
    assert np.array_equal(pose, pose2) and np.array_equal(inl, inl2)


def test_scale_argument_and_degenerate_inputs():
    K, uv, X, R, t, _ = _scene(300, 2, noise_px=0.2)
    p1, _, _ = ransac_PnP(K, uv, X, scale=1)
    p2, _, _ = ransac_PnP(K, uv, X, scale=1000)                                      # inference.py:181-189 passes scale=1000
    assert np.allclose(p1[:, :3], p2[:, :3], atol=1e-6) and np.allclose(p1[:, 3], p2[:, 3], rtol=1e-4, atol=1e-7)
    pose, homo, inl = ransac_PnP(K, uv[:4], X[:4])                                   # too few points: identity, no inliers
    assert np.array_equal(pose, np.eye(4)[:3]) and len(inl) == 0
    pose, _, inl = ransac_PnP(K, np.zeros((0, 2)), np.zeros((0, 3)))
    assert np.array_equal(pose, np.eye(4)[:3]) and len(inl) == 0
    with pytest.raises(ValueError):
        ransac_PnP(K, uv[:10], X[:9])


def test_pose_is_insensitive_at_matcher_noise_level():
    """1e-4 px perturbations of the 2D keypoints (the HIP path's deviation from the oracle) move the pose by < 1e-6 rel."""
    K, uv, X, R, t, _ = _scene(2800, 3, noise_px=0.7)
    p1, _, _ = ransac_PnP(K, uv, X, pnp_reprojection_error=7)
    uv2 = uv + np.float32(1e-4) * np.random.default_rng(9).normal(size=uv.shape).astype(np.float32)
    p2, _, _ = ransac_PnP(K, uv2, X, pnp_reprojection_error=7)
    assert np.abs(p1[:, :3] - p2[:, :3]).max() < 1e-6 and np.linalg.norm(p1[:, 3] - p2[:, 3]) / np.linalg.norm(p1[:, 3]) < 1e-6


def test_async_pool_matches_sync_calls():
    """both trial policies of the reference's ransac_PnP; the pool splits a frame's unconditional trials over its threads and
    must return the sequential call's pose bit for bit, for any thread count; a ticket is read once"""
    scenes = [_scene(400 + 50 * i, 10 + i, noise_px=0.3, outlier_frac=0.1) for i in range(6)]
    K = scenes[0][0]
    for policy, pycolmap_branch, min_iters in (("adaptive", False, None), ("reference", True, 700)):     # 700 trials: two full 256-trial chunks + a partial one
        for threads in (1, 3):
            pool = PnPPool(K, threads=threads, pnp_reprojection_error=7, policy=policy, min_iters=min_iters)
            tickets = [pool.submit(s[1], s[2]) for s in scenes]
            assert tickets == list(range(6)) and pool.wait_all() == 6
            for tk, s in zip(tickets, scenes):
                pose, n_in, rc = pool.result(tk)
                ref, _, inl = ransac_PnP(K, s[1], s[2], pnp_reprojection_error=7, use_pycolmap_ransac=pycolmap_branch, min_iters=min_iters)
                assert rc == 0 and np.array_equal(pose, ref) and n_in == len(inl)
            assert pool.result(tickets[0])[2] == -1          # results are handed out once (a long frame loop does not accumulate them)
            pool.close()


def test_trial_policy_follows_the_reference_branches():
    from onepose_st_amd.pnp import trial_policy
    assert trial_policy(True) == (10000, 1000000)            # pycolmap branch: min_num_trials / max_num_trials (metric_utils.py:155-165)
    assert trial_policy(False) == (4, 10000)                 # cv2 branch: iterationsCount 10 000, adaptive (metric_utils.py:188-196)
    assert trial_policy(True, min_iters=64) == (64, 1000000)
    K, uv, X, R, t, _ = _scene(600, 4, noise_px=0.5, outlier_frac=0.2)
    pa, _, ia = ransac_PnP(K, uv, X, pnp_reprojection_error=7)
    pr, _, ir = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True, min_iters=2000)
    assert np.abs(pa[:, :3] - pr[:, :3]).max() < 1e-4 and abs(len(ia) - len(ir)) <= 2      # more trials, the same pose to refinement accuracy
