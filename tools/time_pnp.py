"""Diagnostic (CPU only): host time of the reference-policy RANSAC (onepose_st_amd.pnp) on c2-sized synthetic matches -- one call on one
thread, and the pool's frames/s at a given thread count.  OPPNP_LIB=<path> selects a library build, OPPNP_NO_AVX512=1 the 256-bit scorer.
    python tools/time_pnp.py [threads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd.pnp import PnPPool, ransac_PnP  # noqa: E402

rng = np.random.default_rng(0)
K = np.array([[608., 0, 160], [0, 608, 120], [0, 0, 1]])


def case(n, noise, out_frac):
    X = rng.uniform(-0.1, 0.1, (n, 3)).astype(np.float32)
    ang = 0.5
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.02, -0.01, 0.6])
    uv = (X @ R.T + t) @ K.T
    uv = uv[:, :2] / uv[:, 2:]
    uv += rng.normal(0, noise, uv.shape)
    m = rng.random(n) < out_frac
    uv[m] = rng.uniform(0, 320, (int(m.sum()), 2))
    return uv.astype(np.float32), X


threads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
try:
    print("cpu:", next(ln for ln in open("/proc/cpuinfo") if ln.startswith("model name")).split(":", 1)[1].strip(),
          "| lib:", os.environ.get("OPPNP_LIB", "(tree)"), "| no512:", os.environ.get("OPPNP_NO_AVX512", "0"))
except Exception:
    pass
for name, (n, noise, of) in {"c2 clean": (2800, 0.2, 0.0), "c2 hard": (1400, 0.3, 0.4), "c1": (300, 0.3, 0.1)}.items():
    uv, X = case(n, noise, of)
    ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True)
    reps = 8
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        pose, _, inl = ransac_PnP(K, uv, X, pnp_reprojection_error=7, use_pycolmap_ransac=True)
        ts.append(time.perf_counter() - t0)
    pool = PnPPool(K, threads=threads, pnp_reprojection_error=7, policy="reference")
    for _ in range(8):
        pool.submit(uv, X)
    pool.wait_all()
    t0 = time.perf_counter()
    tk = [pool.submit(uv, X) for _ in range(64)]
    pool.wait_all()
    rate = 64 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    pool.submit(uv, X)
    pool.wait_all()
    lat = time.perf_counter() - t0
    pool.close()
    print(f"{name:9s} n={n}: one thread {min(ts) * 1e3:6.2f} ms (median {np.median(ts) * 1e3:6.2f}), inliers {len(inl)}; pool of {threads}: {rate:7.0f} frames/s, "
          f"one frame alone {lat * 1e3:.2f} ms")
