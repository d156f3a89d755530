"""Host PnP + RANSAC with the reference's ``ransac_PnP`` signature (``src/utils/metric_utils.py:121-209``), backed by
``libonepose_pnp.so`` (C++, ``csrc_host/pnp.cpp``) through ctypes.  ``ctypes`` releases the GIL during the call, so a
pipeline can solve frame t's pose on a host thread while the GPU matches frame t + 1 (``bench.py``)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_LIB_PATH = os.path.join(_LIB_DIR, "libonepose_pnp.so")
_lib = None


def _cpu_has_avx2_fma() -> bool:
    try:
        flags = next(ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")).split()
        return "avx2" in flags and "fma" in flags
    except (OSError, StopIteration):
        return False


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("OPPNP_LIB") or _LIB_PATH                  # OPPNP_LIB: exactly this build (A/B runs, tools/box.sh ab:...)
        fast = os.path.join(_LIB_DIR, "libonepose_pnp_avx2.so")          # same source built with -mavx2 -mfma
        if path == os.path.join(_LIB_DIR, "libonepose_pnp.so") and os.path.exists(fast) and _cpu_has_avx2_fma():
            path = fast
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run __graft_entry__.build()")
        lib = ctypes.CDLL(path)
        lib.oppnp_ransac.restype = ctypes.c_int
        lib.oppnp_ransac.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.oppnp_p3p.restype = ctypes.c_int
        lib.oppnp_p3p.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oppnp_pool_create.restype = ctypes.c_void_p
        lib.oppnp_pool_create.argtypes = [ctypes.c_int]
        lib.oppnp_pool_destroy.argtypes = [ctypes.c_void_p]
        lib.oppnp_pool_submit.restype = ctypes.c_longlong
        lib.oppnp_pool_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int]
        lib.oppnp_pool_wait_all.restype = ctypes.c_longlong
        lib.oppnp_pool_wait_all.argtypes = [ctypes.c_void_p]
        lib.oppnp_pool_result.restype = ctypes.c_int
        lib.oppnp_pool_result.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.oppnp_estimate_affine2d.restype = ctypes.c_int
        lib.oppnp_estimate_affine2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                                ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.oppnp_p3p4.restype = None
        lib.oppnp_p3p4.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        if lib.oppnp_abi_version() != 3:
            raise RuntimeError("libonepose_pnp.so ABI version mismatch")
        _lib = lib
    return _lib


# trial policy of the two branches of the reference's ``ransac_PnP`` (``metric_utils.py:155-165`` / ``:188-196``)
POLICY = {
    "reference": dict(min_iters=10000, max_iters=1000000),     # pycolmap: min_num_trials 10 000, max_num_trials 1 000 000, confidence 0.99
    "adaptive": dict(min_iters=4, max_iters=10000),            # cv2.solvePnPRansac: iterationsCount 10 000, stops at the confidence
}


SOLVER = {"dlt6": 0, "p3p": 1}


def minimal_solver(use_pycolmap_ransac: bool, solver=None) -> int:
    """pycolmap's ``absolute_pose_estimation`` samples three correspondences (P3P) and scores every root; the OpenCV branch keeps
    the build's 6-point DLT (with a P3P fallback on coplanar samples).  ``solver``: "p3p" / "dlt6" overrides."""
    if solver is None:
        return SOLVER["p3p" if use_pycolmap_ransac else "dlt6"]
    return SOLVER[solver]


def p3p(rays, X):
    """The minimal solver alone: ``rays [3, 2]`` normalised image points, ``X [3, 3]`` world points -> list of ``[3, 4]`` poses."""
    lib = load()
    r = np.ascontiguousarray(rays, dtype=np.float64).reshape(3, 2)
    x = np.ascontiguousarray(X, dtype=np.float64).reshape(3, 3)
    out = np.zeros((4, 12), dtype=np.float64)
    n = lib.oppnp_p3p(r.ctypes.data, x.ctypes.data, out.ctypes.data)
    return [out[i].reshape(3, 4).copy() for i in range(max(n, 0))]


def p3p4(rays, X):
    """Four samples through the four-lane solver the RANSAC loop uses: ``rays [4, 3, 2]``, ``X [4, 3, 3]`` -> four lists of ``[3, 4]`` poses."""
    lib = load()
    r = np.ascontiguousarray(rays, dtype=np.float64).reshape(4, 3, 2)
    x = np.ascontiguousarray(X, dtype=np.float64).reshape(4, 3, 3)
    out = np.zeros((4, 4, 12), dtype=np.float64)
    ns = np.zeros(4, dtype=np.int32)
    lib.oppnp_p3p4(r.ctypes.data, x.ctypes.data, out.ctypes.data, ns.ctypes.data)
    return [[out[l, i].reshape(3, 4).copy() for i in range(int(ns[l]))] for l in range(4)]


def trial_policy(use_pycolmap_ransac: bool, min_iters=None, max_iters=None) -> tuple:
    pol = POLICY["reference" if use_pycolmap_ransac else "adaptive"]
    return (pol["min_iters"] if min_iters is None else int(min_iters), pol["max_iters"] if max_iters is None else int(max_iters))


def ransac_PnP(K, pts_2d, pts_3d, scale=1, pnp_reprojection_error=5, img_hw=None, use_pycolmap_ransac=False,
               confidence=0.99, min_iters=None, max_iters=None, seed=1, solver=None):
    """-> ``(pose [3,4], pose_homo [4,4], inliers [k] int64)`` like the reference.  ``scale`` multiplies the 3D points
    for the solve and divides the translation afterwards (the reference's OpenCV branch, ``metric_utils.py:186,200``).
    ``use_pycolmap_ransac`` selects the trial policy of the reference's branch: ``True`` (what ``inference.py:181-189``
    passes) runs at least 10 000 trials like ``pycolmap.absolute_pose_estimation(min_num_trials=10000, max_num_trials=1e6)``,
    ``False`` stops adaptively at the confidence within 10 000 trials like ``cv2.solvePnPRansac``; ``min_iters`` /
    ``max_iters`` override either.  The minimal solver follows the branch too (:func:`minimal_solver`: P3P / 6-point DLT).
    The estimator itself is the build's own (pose parity vs pycolmap / OpenCV is unpinned);
    ``img_hw`` is accepted for signature compatibility."""
    lib = load()
    min_iters, max_iters = trial_policy(bool(use_pycolmap_ransac), min_iters, max_iters)
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
    p2 = np.ascontiguousarray(np.asarray(pts_2d, dtype=np.float32).reshape(-1, 2))
    p3 = np.ascontiguousarray(np.asarray(pts_3d, dtype=np.float32).reshape(-1, 3) * np.float32(scale))
    n = p2.shape[0]
    if p3.shape[0] != n:
        raise ValueError("pts_2d and pts_3d must have the same length")
    pose = np.zeros((3, 4), dtype=np.float64)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_in, iters = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib.oppnp_ransac(K.ctypes.data, p2.ctypes.data, p3.ctypes.data, n, float(pnp_reprojection_error), float(confidence),
                          int(min_iters), int(max_iters), int(seed), minimal_solver(bool(use_pycolmap_ransac), solver), pose.ctypes.data, mask.ctypes.data,
                          ctypes.byref(n_in), ctypes.byref(iters))
    if rc < 0:
        raise ValueError("oppnp_ransac: invalid arguments")
    pose[:, 3] /= scale
    pose_homo = np.concatenate([pose, np.array([[0.0, 0.0, 0.0, 1.0]])], axis=0)
    inliers = np.nonzero(mask[:n])[0].astype(np.int64) if rc == 0 else np.array([], dtype=np.int64)
    return pose, pose_homo, inliers


def estimate_affine2d(src, dst, ransac_reproj_threshold=6.0, max_iters=2000, confidence=0.99, seed=1):
    """``cv2.estimateAffine2D(src, dst, method=cv2.RANSAC, ransacReprojThreshold=...)`` as the reference's detector calls it
    (``local_feature_2D_detector.py:120-122``): -> ``(affine [2, 3] float64 or None, inliers [n, 1] uint8)``."""
    lib = load()
    s = np.ascontiguousarray(np.asarray(src, dtype=np.float32).reshape(-1, 2))
    d = np.ascontiguousarray(np.asarray(dst, dtype=np.float32).reshape(-1, 2))
    if s.shape != d.shape:
        raise ValueError("src and dst must have the same shape")
    n = s.shape[0]
    A = np.zeros((2, 3), dtype=np.float64)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_in = ctypes.c_int(0)
    rc = lib.oppnp_estimate_affine2d(s.ctypes.data, d.ctypes.data, n, float(ransac_reproj_threshold), int(max_iters), float(confidence),
                                     int(seed), A.ctypes.data, mask.ctypes.data, ctypes.byref(n_in))
    if rc < 0:
        raise ValueError("oppnp_estimate_affine2d: invalid arguments")
    return (A if rc == 0 else None), mask[:n].reshape(-1, 1)


class PnPPool:
    """Library-owned worker threads: ``submit`` copies the matches and returns a ticket immediately."""

    def __init__(self, K, threads=3, pnp_reprojection_error=7, confidence=0.99, min_iters=None, max_iters=None, seed=1, policy="reference",
                 solver=None):
        """``policy``: "reference" = the pycolmap branch's trial floor (what the reference's inference loop runs), "adaptive" = the
        OpenCV branch; a frame's unconditional trials are split over all pool threads (same pose for any thread count)."""
        self._lib = load()
        min_iters, max_iters = trial_policy(policy == "reference", min_iters, max_iters)
        self._K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
        self._args = (float(pnp_reprojection_error), float(confidence), int(min_iters), int(max_iters), int(seed),
                      minimal_solver(policy == "reference", solver))
        self._pool = ctypes.c_void_p(self._lib.oppnp_pool_create(int(threads)))

    def submit(self, pts_2d, pts_3d) -> int:
        p2 = np.ascontiguousarray(pts_2d, dtype=np.float32)
        p3 = np.ascontiguousarray(pts_3d, dtype=np.float32)
        return int(self._lib.oppnp_pool_submit(self._pool, self._K.ctypes.data, p2.ctypes.data, p3.ctypes.data, p2.shape[0], *self._args))

    def wait_all(self) -> int:
        return int(self._lib.oppnp_pool_wait_all(self._pool))

    def result(self, ticket: int):
        pose = np.zeros((3, 4), dtype=np.float64)
        n_in = ctypes.c_int(0)
        rc = self._lib.oppnp_pool_result(self._pool, int(ticket), pose.ctypes.data, ctypes.byref(n_in))
        return pose, n_in.value, rc

    def close(self):
        if self._pool:
            self._lib.oppnp_pool_destroy(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
