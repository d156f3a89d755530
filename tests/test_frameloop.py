"""Row f-2: input formats and the per-sequence loop around the matcher.

The reference's crop arithmetic lives in cv2 (``getAffineTransform`` / ``warpAffine``, opencv 4.4.0.46 per requirements.txt),
which is not installed here: the geometry is checked against a restatement of ``data_utils.get_affine_transform`` that solves
the same three-point system, the pixels against a numpy bilinear resampling of the same two warps (parity unpinned at the cv2
boundary: its fixed-point interpolation is not reproduced).
"""
import numpy as np
import pytest
import torch

from onepose_st_amd import frameloop as fl


# ---- restatement of data_utils.py:17-62 (rot = 0) with the affine solved from the three point pairs -------------------------
def _third(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _affine_from_points(src, dst):
    A = np.concatenate([src, np.ones((3, 1))], axis=1).astype(np.float64)
    return np.linalg.solve(A, dst.astype(np.float64)).T            # [2, 3]


def _ref_affine(center, scale, out_wh):
    src_w, dst_w, dst_h = scale[0], out_wh[0], out_wh[1]
    src, dst = np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + np.array([0, src_w * -0.5])
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    src[2], dst[2] = _third(src[0], src[1]), _third(dst[0], dst[1])
    return _affine_from_points(src, dst)


def _ref_two_stage(bbox, K, S):
    x0, y0, x1, y1 = bbox

    def stage(box, Kc, shape_hw):
        center = np.array([(box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0])
        scale = np.array([box[2] - box[0], box[3] - box[1]])
        t = np.concatenate([_ref_affine(center, scale, [shape_hw[1], shape_hw[0]]), [[0, 0, 1]]], axis=0)
        return (t @ np.concatenate([Kc, np.zeros((3, 1))], axis=1))[:3, :3], t

    K1, t1 = stage(bbox, K, (y1 - y0, x1 - x0))
    K2, t2 = stage([0, 0, x1 - x0, y1 - y0], K1, (S, S))
    return K2, t2 @ t1


@pytest.mark.parametrize("bbox", [(100, 50, 420, 300), (37, 211, 180, 260), (-20, -5, 300, 480)])
def test_crop_geometry_matches_two_stage_affine(bbox):
    K = np.array([[1400.0, 0, 960.0], [0, 1405.0, 540.0], [0, 0, 1]])
    K_ref, t_ref = _ref_two_stage(bbox, K, 512)
    K_crop, trans = fl.crop_geometry(bbox, K, 512)
    np.testing.assert_allclose(trans, t_ref, rtol=1e-6, atol=1e-4)          # the reference builds its point sets in float32
    np.testing.assert_allclose(K_crop, K_ref, rtol=1e-6, atol=1e-3)


def test_project_bbox():
    K = np.array([[800.0, 0, 320.0], [0, 800.0, 240.0], [0, 0, 1]])
    pose = np.concatenate([np.eye(3), [[0.05], [-0.02], [1.0]]], axis=1)
    corners = 0.1 * np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], dtype=np.float64)
    uv = []
    for c in corners:
        p = K @ (pose[:, :3] @ c + pose[:, 3])
        uv.append(p[:2] / p[2])
    uv = np.array(uv)
    want = np.array([uv[:, 0].min(), uv[:, 1].min(), uv[:, 0].max(), uv[:, 1].max()]).astype(np.int32)
    assert np.array_equal(fl.project_bbox(K, pose, corners), want)
    assert np.array_equal(fl.project_bbox(K, np.concatenate([pose, [[0, 0, 0, 1]]]), corners), want)      # [4, 4] pose


def test_object_block_roundtrip_and_truncation(tmp_path):
    g = np.random.default_rng(0)
    N = 50
    kp, d, sc = g.normal(size=(N, 3)), g.normal(size=(128, N)), g.random((N, 1))
    dc, scc = g.normal(size=(256, N)), g.random((N, 1))
    path = str(tmp_path / "anno_3d_average.npz")
    fl.save_object_block(path, kp, d, sc, dc, scc)
    blk = fl.load_object_block(path, "cpu")
    assert blk["num_3d_orig"] == N
    assert tuple(blk["keypoints3d"].shape) == (1, N, 3) and tuple(blk["descriptors3d_db"].shape) == (1, 128, N)
    assert tuple(blk["descriptors3d_coarse_db"].shape) == (1, 256, N) and blk["keypoints3d"].dtype == torch.float32
    np.testing.assert_allclose(blk["keypoints3d"][0].numpy(), kp.astype(np.float32))
    np.testing.assert_allclose(blk["descriptors3d_coarse_db"][0].numpy(), dc.astype(np.float32))
    # larger than shape3d: one torch.randint draw (with replacement) indexes every array alike (data_utils.py:222-246)
    cut = fl.load_object_block(path, "cpu", shape3d=20, generator=torch.Generator().manual_seed(3))
    idx = torch.randint(N, (20,), generator=torch.Generator().manual_seed(3)).numpy()
    np.testing.assert_allclose(cut["keypoints3d"][0].numpy(), kp.astype(np.float32)[idx])
    np.testing.assert_allclose(cut["descriptors3d_db"][0].numpy(), d.astype(np.float32)[:, idx])
    np.testing.assert_allclose(cut["scores3d_coarse"].numpy(), scc.astype(np.float32)[idx])
    same = fl.load_object_block(path, "cpu", shape3d=N)          # not larger: untouched
    assert torch.equal(same["keypoints3d"], blk["keypoints3d"])
    np.savez(str(tmp_path / "bad.npz"), keypoints3d=kp)
    np.savez(str(tmp_path / "bad_coarse.npz"), descriptors3d=dc, scores3d=scc)
    with pytest.raises(KeyError):
        fl.load_object_block(str(tmp_path / "bad.npz"), "cpu")


def _numpy_crop(frame, bbox, S):
    x0, y0, x1, y1 = [int(v) for v in bbox]
    wb, hb = x1 - x0, y1 - y0
    H, W = frame.shape
    crop = np.zeros((hb + 2, wb + 2), np.float32)                 # one pixel of zero border on every side
    for j in range(hb):
        for i in range(wb):
            if 0 <= y0 + j < H and 0 <= x0 + i < W:
                crop[j + 1, i + 1] = frame[y0 + j, x0 + i]
    inv = np.float32(wb) / np.float32(S)
    X = np.arange(S, dtype=np.float32)
    u = (X - np.float32(0.5 * S)) * inv + np.float32(0.5 * wb)
    v = (X - np.float32(0.5 * S)) * inv + np.float32(0.5 * hb)
    out = np.zeros((S, S), np.float32)
    for Y in range(S):
        fv = np.floor(v[Y]); b = v[Y] - fv; j0 = int(fv)
        for Xi in range(S):
            fu = np.floor(u[Xi]); a = u[Xi] - fu; i0 = int(fu)

            def px(i, j):
                return crop[j + 1, i + 1] if (-1 <= i <= wb and -1 <= j <= hb) else 0.0
            val = (1 - b) * ((1 - a) * px(i0, j0) + a * px(i0 + 1, j0)) + b * ((1 - a) * px(i0, j0 + 1) + a * px(i0 + 1, j0 + 1))
            out[Y, Xi] = np.float32(min(max(np.rint(val), 0), 255)) / np.float32(255)
    return out


class _FakeModel:
    """stands in for the matcher: projects a few object points with the true pose of the current frame into the crop"""

    def __init__(self, pts3d, poses, K):
        self.pts3d, self.poses, self.K, self.t, self.fail_at, self.trans = pts3d, poses, K, 0, set(), None

    def __call__(self, data):
        pose = self.poses[self.t]
        cam = pose[:, :3] @ self.pts3d.T + pose[:, 3:4]
        uv = self.K @ cam
        uv = (uv[:2] / uv[2:]).T
        uvc = (self.trans @ np.concatenate([uv, np.ones((len(uv), 1))], axis=1).T).T[:, :2]
        n = 4 if self.t in self.fail_at else len(uv)              # too few matches -> PnP cannot reach 20 inliers
        data["mkpts_3d_db"] = torch.tensor(self.pts3d[:n], dtype=torch.float32)
        data["mkpts_query_f"] = torch.tensor(uvc[:n], dtype=torch.float32)
        self.t += 1


def test_sequence_runner_control_flow():
    """detector on frame 0, previous-pose boxes afterwards, re-detection after a frame with < 20 inliers (inference.py:143-173);
    poses come back through the crop intrinsics (K_crop) exactly."""
    g = np.random.default_rng(1)
    K = np.array([[900.0, 0, 320.0], [0, 900.0, 240.0], [0, 0, 1]])
    pts = g.uniform(-0.08, 0.08, size=(200, 3))
    bbox3d = 0.1 * np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], dtype=np.float64)
    poses = []
    for t in range(5):
        a = 0.05 * t
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        poses.append(np.concatenate([R, [[0.01 * t], [0.0], [0.8]]], axis=1))
    fake = _FakeModel(pts, poses, K)
    fake.fail_at = {2}
    calls = []

    def detector(frame, t):
        calls.append(t)
        return fl.project_bbox(K, poses[t], bbox3d)               # a perfect detector

    class Runner(fl.SequenceRunner):                              # tell the fake model which crop the loop chose
        pass

    def crop_fn(dev_frame, bbox, S):
        fake.trans = fl.crop_geometry(bbox, K, S)[1]
        return torch.zeros(1, 1, S, S)

    block = {"keypoints3d": torch.zeros(1, 200, 3), "descriptors3d_db": torch.zeros(1, 128, 200), "descriptors3d_coarse_db": torch.zeros(1, 256, 200)}
    runner = Runner(fake, block, K, bbox3d, detector, crop_fn=crop_fn)
    recs = runner.run([np.zeros((480, 640), np.uint8)] * 5)
    assert calls == [0, 3]                                        # frame 0, and frame 3 because frame 2 lost the track
    assert [r["redetected"] for r in recs] == [True, False, False, True, False]
    for t, r in enumerate(recs):
        if t == 2:
            assert len(r["inliers"]) < fl.MIN_INLIERS
            continue
        assert len(r["inliers"]) >= 150
        np.testing.assert_allclose(r["pose"], poses[t], atol=2e-4)
    # frame 1's box is the projection of the 3D box with frame 0's ESTIMATED pose
    assert np.array_equal(recs[1]["bbox"], fl.project_bbox(K, recs[0]["pose"], bbox3d))


@pytest.mark.gpu
def test_crop_kernel_vs_numpy():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    g = np.random.default_rng(2)
    frame = g.integers(0, 256, size=(97, 131), dtype=np.uint8)
    dev = torch.device("cuda:0")
    fd = torch.from_numpy(frame).to(dev)
    for bbox, S in (((10, 20, 90, 70), 64), ((-7, -3, 60, 40), 48), ((100, 60, 140, 110), 32), ((0, 0, 131, 97), 40)):
        got = fl.crop_query(fd, bbox, S)[0, 0].cpu().numpy()
        want = _numpy_crop(frame, bbox, S)
        # identical up to the fused multiply-adds of the device code flipping a grey level that sits exactly on .5
        lg, lw = np.rint(got * 255).astype(int), np.rint(want * 255).astype(int)
        assert np.max(np.abs(lg - lw)) <= 1 and np.mean(lg != lw) < 0.01, bbox
        assert np.array_equal(got[lg == lw], want[lg == lw])          # same level -> bit-identical float (true division by 255)


@pytest.mark.gpu
def test_sequence_runner_on_device_runs_the_real_model():
    """plumbing on the device: uploads, GPU crop, the real matcher (HIP backbone included) and PnP; random frames carry no
    object, so every frame ends with an empty inlier set and asks the detector again"""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from onepose_st_amd.config import default_config
    from onepose_st_amd.model import OnePosePlus_model
    from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict
    dev = torch.device("cuda:0")
    cfg = default_config()
    sd = make_synthetic_state_dict(0, cfg)
    model = OnePosePlus_model(cfg).eval()
    model.load_state_dict(sd)
    model.to(dev)
    obj = make_synthetic_inputs(sd, n_points=300, image_hw=(64, 64), n_plant=0, seed=4, config=cfg)
    block = {k: obj[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    K = np.array([[500.0, 0, 160.0], [0, 500.0, 120.0], [0, 0, 1]])
    bbox3d = 0.1 * np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], dtype=np.float64)
    g = np.random.default_rng(5)
    frames = [g.integers(0, 256, size=(240, 320), dtype=np.uint8) for _ in range(3)]
    calls = []

    def detector(frame, t):
        calls.append(t)
        return [40, 30, 200, 190]

    recs = fl.SequenceRunner(model, block, K, bbox3d, detector, crop_size=128).run(frames)
    assert calls == [0, 1, 2] and len(recs) == 3
    for r in recs:
        assert r["pose"].shape == (3, 4) and len(r["inliers"]) == 0 and r["K_crop"].shape == (3, 3)
