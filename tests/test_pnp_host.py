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


@pytest.mark.parametrize("n", [4, 5])
def test_p3p_branch_gives_a_pose_from_four_or_five_matches(n):
    """pycolmap's P3P + RANSAC needs a sample of three and one more point to pick the root, so the pycolmap branch
    (use_pycolmap_ransac=True, what inference.py:181-189 passes) returns a pose for 4 and 5 matches; the 6-point branch cannot."""
    K, uv, X, R, t, _ = _scene(40, 5)
    pose, homo, inl = ransac_PnP(K, uv[:n], X[:n], pnp_reprojection_error=7, use_pycolmap_ransac=True)
    assert len(inl) == n
    assert _rot_err(pose[:, :3], R) < 1e-2 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 1e-3
    pool = PnPPool(K, threads=2, policy="reference")
    tk = pool.submit(uv[:n], X[:n])
    pool.wait_all()
    pp, n_in, rc = pool.result(tk)
    assert rc == 0 and n_in == n and np.allclose(pp, pose, atol=1e-9)
    pose6, _, inl6 = ransac_PnP(K, uv[:n], X[:n], use_pycolmap_ransac=False)          # 6-point DLT branch: identity, no inliers
    assert np.array_equal(pose6, np.eye(4)[:3]) and len(inl6) == 0
    pose3, _, inl3 = ransac_PnP(K, uv[:3], X[:3], use_pycolmap_ransac=True)           # three points: no way to pick a root
    assert np.array_equal(pose3, np.eye(4)[:3]) and len(inl3) == 0


@pytest.mark.parametrize("pycolmap_branch", [False, True])
def test_outlier_heavy_frame_keeps_its_pose(pycolmap_branch):
    """40 % outliers with the outliers FIRST in the list (the first hypotheses then have a handful of inliers): the adaptive branch used to
    turn "billions of trials needed" into a negative int, stop at once and return no pose (found by the c1_hard fixture)."""
    K, uv, X, R, t, n_out = _scene(300, 6, noise_px=0.8, outlier_frac=0.4)
    pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=pycolmap_branch)
    assert len(inl) >= 0.95 * (300 - n_out) and (inl >= n_out).mean() > 0.97
    assert _rot_err(pose[:, :3], R) < 0.3 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 5e-3


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


def _pose(rng, ang=None):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = rng.uniform(0.1, 2.5) if ang is None else ang
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx, np.array([0.05 * rng.normal(), 0.05 * rng.normal(), 0.45])


def test_p3p_minimal_solver_contains_the_planted_pose():
    """Grunert quartic + triangle alignment: on exact data one of the (at most four) roots is the planted pose; every returned
    pose is a rotation that reprojects its three points exactly"""
    from onepose_st_amd.pnp import p3p
    rng = np.random.default_rng(0)
    hit, total_roots, sharp = 0, 0, 0
    for trial in range(500):
        R, t = _pose(rng)
        X = (rng.random((3, 3)) - 0.5) * 0.2
        pc = X @ R.T + t
        rays = pc[:, :2] / pc[:, 2:]
        sols = p3p(rays, X)
        assert len(sols) <= 4
        total_roots += len(sols)
        for ps in sols:
            Rs, ts = ps[:, :3], ps[:, 3]
            assert np.abs(Rs @ Rs.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(Rs) - 1) < 1e-9
            q = X @ Rs.T + ts
            # (a root next to a double root of the quartic is only as sharp as the square root of the rounding: still far below a pixel)
            assert np.abs(q[:, :2] / q[:, 2:] - rays).max() < 1e-4 and (q[:, 2] > 0).all()
            sharp += np.abs(q[:, :2] / q[:, 2:] - rays).max() < 1e-9
        if sols and min(np.abs(ps[:, :3] - R).max() + np.abs(ps[:, 3] - t).max() for ps in sols) < 1e-6:
            hit += 1
    assert hit >= 490, hit                   # a handful of samples sit on a double root of the quartic (lost to rounding: RANSAC draws another)
    assert 500 <= total_roots <= 2000 and sharp >= 0.97 * total_roots
    assert p3p(np.zeros((3, 2)), np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2.0]])) == []          # collinear points: no pose


def _planar_scene(n, seed, noise_px=0.0, outlier_frac=0.0):
    """every 3D point on ONE face of a box (z = const in the object frame): what a 6-point DLT cannot solve"""
    rng = np.random.default_rng(seed)
    X = np.concatenate([(rng.random((n, 2)) - 0.5) * np.array([0.2, 0.14]), np.full((n, 1), 0.05)], 1)
    R, t = _pose(rng, ang=0.6)
    K = np.array([[1216.0, 0, 320.0], [0, 1216.0, 240.0], [0, 0, 1]])
    pc = X @ R.T + t
    uv = (pc[:, :2] / pc[:, 2:]) * 1216.0 + np.array([320.0, 240.0]) + noise_px * rng.normal(size=(n, 2))
    n_out = int(outlier_frac * n)
    if n_out:
        uv[:n_out] = rng.random((n_out, 2)) * np.array([640.0, 480.0])
    return K, uv.astype(np.float32), X.astype(np.float32), R, t, n_out


@pytest.mark.parametrize("pycolmap_branch", [True, False])
def test_coplanar_scene_recovers_the_planted_pose(pycolmap_branch):
    """box faces / cards are what OnePose tracks: the P3P branch (and the DLT branch through its P3P fallback) must solve them"""
    K, uv, X, R, t, n_out = _planar_scene(1500, 21, noise_px=0.4, outlier_frac=0.25)
    kw = dict(min_iters=1000) if pycolmap_branch else {}
    pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=pycolmap_branch, **kw)
    assert _rot_err(pose[:, :3], R) < 0.1 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 3e-3
    assert len(inl) >= 0.95 * (1500 - n_out) and (inl >= n_out).mean() > 0.97
    pose2, _, inl2 = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=pycolmap_branch, **kw)
    assert np.array_equal(pose, pose2) and np.array_equal(inl, inl2)                       # deterministic


def test_p3p_branch_on_a_general_scene_and_pool_thread_independence():
    K, uv, X, R, t, n_out = _scene(2000, 31, noise_px=0.5, outlier_frac=0.3)
    pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True, min_iters=1500)
    assert _rot_err(pose[:, :3], R) < 0.05 and np.linalg.norm(pose[:, 3] - t) / np.linalg.norm(t) < 2e-3
    for threads in (1, 4):
        pool = PnPPool(K, threads=threads, pnp_reprojection_error=7, policy="reference", min_iters=1500)
        tk = pool.submit(uv, X)
        pool.wait_all()
        p2, n_in, rc = pool.result(tk)
        assert rc == 0 and np.array_equal(p2, pose) and n_in == len(inl)
        pool.close()


def test_reference_policy_cost_is_reported():
    """ms per 10 000 trials of the pycolmap-branch policy (P3P, every root scored) at the matcher's match count"""
    import time
    K, uv, X, R, t, _ = _scene(3000, 41, noise_px=0.5, outlier_frac=0.05)
    t0 = time.perf_counter()
    pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True)
    ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True, solver="dlt6")
    ms6 = 1e3 * (time.perf_counter() - t0)
    print(f"10 000 trials at 3 000 correspondences, single thread: P3P {ms:.1f} ms, 6-point DLT {ms6:.1f} ms")
    assert _rot_err(pose[:, :3], R) < 0.05 and len(inl) > 2700 and ms < 2000


def test_scorer_widths_pick_the_same_pose():
    """The screening pass exists twice in the AVX2 build (256-bit, and 512-bit on a CPU with AVX-512; OPPNP_NO_AVX512=1 keeps the first):
    the two sum a hypothesis' float cost in different orders, the double re-score decides either way -- same inlier set, same pose to
    rounding, on a clean, a noisy and an outlier-heavy frame and on both trial policies.  (On a CPU without AVX-512 both runs take the
    256-bit loop and the test is trivially true.)"""
    import json
    import os
    import subprocess
    import sys
    code = ("import json, numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_pnp_host import _scene; from onepose_st_amd.pnp import ransac_PnP\n"
            "out = []\n"
            "for n, seed, noise, of in ((2800, 3, 0.2, 0.0), (1500, 5, 0.5, 0.4), (300, 7, 1.0, 0.1), (37, 9, 0.3, 0.2)):\n"
            "    K, uv, X, R, t, _ = _scene(n, seed, noise_px=noise, outlier_frac=of)\n"
            "    for branch in (True, False):\n"
            "        pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=branch, min_iters=2000 if branch else None)\n"
            "        out.append([pose.tolist(), inl.tolist()])\n"
            "print(json.dumps(out))\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    res = []
    for no512 in ("0", "1"):
        env = dict(os.environ, OPPNP_NO_AVX512=no512)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    for (pa, ia), (pb, ib) in zip(*res):
        assert ia == ib
        assert np.allclose(np.array(pa), np.array(pb), rtol=0, atol=1e-9)


def test_four_lane_p3p_equals_the_one_sample_solver():
    """oppnp_p3p4 (what the RANSAC loop of the pycolmap branch calls) against oppnp_p3p, sample by sample: same number of poses, same order,
    entries equal to rounding -- on exact data, noisy data (the roots a RANSAC sees), symmetric (biquadratic) and degenerate samples
    mixed into one group of four."""
    from onepose_st_amd.pnp import p3p, p3p4
    rng = np.random.default_rng(5)
    n_pose = 0
    diffs = []
    for group in range(400):
        rays, Xs = np.zeros((4, 3, 2)), np.zeros((4, 3, 3))
        for l in range(4):
            R, t = _pose(rng)
            X = (rng.random((3, 3)) - 0.5) * 0.2
            kind = (group + l) % 7
            if kind == 5:                                     # isosceles triangle seen head-on: the quartic's odd coefficients vanish
                X = np.array([[-0.05, 0.0, 0.0], [0.05, 0.0, 0.0], [0.0, 0.08, 0.0]])
                R, t = np.eye(3), np.array([0.0, 0.0, 0.5])
            if kind == 6 and group % 3 == 0:
                X[2] = 2 * X[1] - X[0]                        # collinear: no pose
            pc = X @ R.T + t
            rays[l] = pc[:, :2] / pc[:, 2:] + (1e-3 * rng.normal(size=(3, 2)) if kind in (1, 3) else 0.0)
            Xs[l] = X
        four = p3p4(rays, Xs)
        for l in range(4):
            one = p3p(rays[l], Xs[l])
            assert len(one) == len(four[l]), (group, l, len(one), len(four[l]))
            for pa, pb in zip(one, four[l]):
                diffs.append(float(np.abs(pa - pb).max()))
                n_pose += 1
    diffs = np.sort(np.array(diffs))
    print(f"{n_pose} poses: median |diff| {np.median(diffs):.1e}, 95 % below {diffs[int(0.95 * n_pose)]:.1e}, 99 % below {diffs[int(0.99 * n_pose)]:.1e}, max {diffs[-1]:.1e}")
    # (a root next to a double root of the quartic is only as sharp as the square root of the rounding, in either solver)
    assert n_pose > 2000 and np.median(diffs) < 1e-14 and diffs[int(0.95 * n_pose)] < 1e-9 and diffs[-1] < 1e-4


def test_pool_submit_from_several_threads_with_empty_frames_and_early_close():
    """The pool's submit path takes no lock its workers take (a lock-free inbox + a counting semaphore: a descheduled background worker must
    never stop the thread that feeds the GPU).  Three threads submit 600 small frames between them, every seventh with too few matches
    (identity pose, rc 1, recorded by a worker); every ticket is unique, every job finishes, results equal the sequential call; a pool
    closed with work still queued shuts down cleanly."""
    import threading
    scenes = [_scene(60 + 5 * i, 50 + i, noise_px=0.3, outlier_frac=0.1) for i in range(5)]
    K = scenes[0][0]
    ref = [ransac_PnP(K, s[1], s[2], pnp_reprojection_error=7, use_pycolmap_ransac=True, min_iters=300) for s in scenes]
    pool = PnPPool(K, threads=4, pnp_reprojection_error=7, policy="reference", min_iters=300)
    got = {}
    lock = threading.Lock()

    def feed(tid):
        for i in range(200):
            which = (i + tid) % 7
            if which >= 5:
                tk = pool.submit(np.zeros((2, 2), np.float32), np.zeros((2, 3), np.float32))
            else:
                tk = pool.submit(scenes[which][1], scenes[which][2])
            with lock:
                assert tk not in got
                got[tk] = which
    ths = [threading.Thread(target=feed, args=(t,)) for t in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert pool.wait_all() == 600 and sorted(got) == list(range(600))
    for tk, which in got.items():
        pose, n_in, rc = pool.result(tk)
        if which >= 5:
            assert rc == 1 and n_in == 0 and np.array_equal(pose, np.eye(4)[:3])
        else:
            assert rc == 0 and np.array_equal(pose, ref[which][0]) and n_in == len(ref[which][2])
    pool.close()
    pool = PnPPool(K, threads=2, pnp_reprojection_error=7, policy="reference", min_iters=20000)
    for s in scenes * 4:
        pool.submit(s[1], s[2])
    pool.close()                                          # 20 frames of 20 000 trials queued: the workers finish what is queued, then exit
