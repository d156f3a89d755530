"""Row f-3 without a GPU: what the reference tree itself holds about the LoFTR detector -- the constants of
``loftr_for_onepose_plus_cfg.py:10-50``, the ``state_dict`` layout ``build_2D_match_model`` loads, the control flow of
``match_worker`` / ``detect_by_matching`` (``local_feature_2D_detector.py:89-162``) -- pinned on the oracle's restatement and on the
host code of the product (affine RANSAC in C++, box rules, view sampling).  The LoFTR arithmetic itself lives in an un-vendored
submodule: parity unpinned (oracle/loftr_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import loftr_oracle as lo
from onepose_st_amd import detector, loftr
from onepose_st_amd.pnp import estimate_affine2d
from onepose_st_amd.synthetic import make_synthetic_loftr_state_dict


def test_config_constants_of_the_reference():
    cfg = loftr.default_cfg
    assert cfg == lo.loftr_default_cfg()
    assert cfg["fine_window_size"] == 9 and cfg["resolution"] == (8, 2) and cfg["fine_concat_coarse_feat"] is False
    assert cfg["coarse"]["layer_names"] == ["self", "cross"] * 4 and cfg["coarse"]["d_model"] == 256 and cfg["coarse"]["nhead"] == 8
    assert cfg["coarse"]["temp_bug_fix"] is False
    assert cfg["match_coarse"]["thr"] == 0.2 and cfg["match_coarse"]["border_rm"] == 2 and cfg["match_coarse"]["dsmax_temperature"] == 0.1
    assert cfg["fine"]["layer_names"] == ["self", "cross"] and cfg["fine"]["d_model"] == 128


def test_state_dict_layout_and_strict_load():
    m = loftr.LoFTR_for_OnePose_Plus()
    sd = make_synthetic_loftr_state_dict(0)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    keys = list(sd.keys())
    assert sum(k.startswith("backbone.") for k in keys) == 107
    assert sum(k.startswith("loftr_coarse.layers.") for k in keys) == 8 * 10 and sum(k.startswith("loftr_fine.layers.") for k in keys) == 2 * 10
    assert tuple(sd["loftr_coarse.layers.7.mlp.0.weight"].shape) == (512, 512) and tuple(sd["loftr_fine.layers.1.mlp.2.weight"].shape) == (128, 256)
    assert not any("pos_encoding" in k for k in keys)              # the table is a non-persistent buffer in LoFTR
    with pytest.raises(Exception):                                  # no CPU fallback
        m.eval()({"image0": torch.zeros(1, 1, 64, 64), "image1": torch.zeros(1, 1, 64, 64)})
    with pytest.raises(NotImplementedError):
        loftr.LoFTR_for_OnePose_Plus({**loftr.default_cfg, "coarse": {**loftr.default_cfg["coarse"], "temp_bug_fix": True}})


def test_oracle_matcher_on_planted_features():
    """the restated LoFTR on planted backbone outputs (image 1 = image 0 moved by (2, 1) coarse cells): matches exist, their
    displacement is the shift, and the published rules hold on the result (all-sides border, mutual nearest, strict threshold)"""
    from tests.loftr_helpers import oracle_hook, planted_pair
    sd = make_synthetic_loftr_state_dict(0)
    img = torch.zeros(1, 1, 96, 128)
    with torch.no_grad():
        out = lo.loftr_forward(sd, lo.loftr_default_cfg(), img, img, feature_hook=oracle_hook(planted_pair((96, 128))))
    K = len(out["i_ids"])
    assert K >= 40 and out["fine_f0"].shape == (K, 81, 128) and out["expec_f"].shape == (K, 3)
    d = (out["mkpts1_f"] - out["mkpts0_f"]).numpy()
    assert np.abs(np.median(d, axis=0) - np.array([16.0, 8.0])).max() < 0.5
    h0, w0 = out["hw0_c"]
    i, j = out["i_ids"], out["j_ids"]
    for ids in (i, j):
        assert bool(((ids // w0 >= 2) & (ids // w0 < h0 - 2) & (ids % w0 >= 2) & (ids % w0 < w0 - 2)).all())
    conf = out["conf_matrix"][0]
    assert torch.equal(conf[i, j], conf.max(dim=1)[0][i]) and torch.equal(conf[i, j], conf.max(dim=0)[0][j]) and bool((conf[i, j] > 0.2).all())
    assert torch.equal(out["mkpts0_f"], out["mkpts0_c"])


def test_affine_ransac_against_the_oracle_restatement():
    rng = np.random.default_rng(0)
    src = rng.random((400, 2)) * np.array([640, 480])
    A = np.array([[0.8, -0.3, 50.0], [0.25, 0.9, -20.0]])
    dst = src @ A[:, :2].T + A[:, 2] + 0.7 * rng.normal(size=(400, 2))
    dst[:120] = rng.random((120, 2)) * np.array([640, 480])
    a, inl = estimate_affine2d(src, dst, 6.0)
    a_o, inl_o = lo.estimate_affine2d(src, dst, 6.0, iters=500)
    assert inl.shape == (400, 1) and inl.dtype == np.uint8
    assert np.abs(a - A).max() < 0.5 and np.abs(a - a_o).max() < 0.5
    assert inl[120:].mean() > 0.98 and inl[:120].mean() < 0.1 and abs(int(inl.sum()) - int(inl_o.sum())) <= 6
    a2, inl2 = estimate_affine2d(src, dst, 6.0)
    assert np.array_equal(a, a2) and np.array_equal(inl, inl2)                         # deterministic
    none, z = estimate_affine2d(src[:2], dst[:2])
    assert none is None and z.sum() == 0
    col = np.stack([np.arange(10.0), np.arange(10.0)], 1)                              # collinear: no affinity
    assert estimate_affine2d(col, col + 1.0)[0] is None


def test_detector_rules():
    # view sampling: every len // n_ref_view-th image starting at index 1
    assert detector.sample_reference_views(150, 15) == list(range(1, 150, 10))
    assert detector.sample_reference_views(31, 15) == list(range(1, 31, 2))
    with pytest.raises(ValueError):
        detector.sample_reference_views(10, 15)
    # box from an affinity: corners truncated to int32 (toward zero), min / max
    A = np.array([[0.5, 0.0, 10.7], [0.0, 0.5, -3.2]])
    assert lo.box_from_affine(A, (480, 640)).tolist() == [10, -3, 330, 236]
    assert lo.fallback_box((480, 640)).tolist() == [320 - 500, 240 - 500, 320 + 500, 240 + 500]
    # the first view among equals wins (stable sort with reverse=True)
    res = [{"inliers": np.ones(5), "bbox": "a"}, {"inliers": np.ones(9), "bbox": "b"}, {"inliers": np.ones(9), "bbox": "c"}, {"inliers": np.empty(0), "bbox": "d"}]
    assert lo.pick_detection(res) == "b"


class _FakeMatcher(torch.nn.Module):
    """stands in for LoFTR: replays prepared match lists per reference view (the detector's host logic needs no GPU)"""

    def __init__(self, per_view):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.per_view, self.calls = per_view, 0

    def forward(self, data):
        V = data["image0"].shape[0]
        if V > 1:                                    # the detector's batched call: every view's matches, b_ids = the view, ascending
            self.calls += 1
            self.batched_calls = getattr(self, "batched_calls", 0) + 1
            data["b_ids"] = torch.cat([torch.full((len(self.per_view[v][0]),), v, dtype=torch.int64) for v in range(V)])
            data["mkpts0_f"] = torch.cat([torch.tensor(self.per_view[v][0], dtype=torch.float32) for v in range(V)])
            data["mkpts1_f"] = torch.cat([torch.tensor(self.per_view[v][1], dtype=torch.float32) for v in range(V)])
            return
        m0, m1 = self.per_view[self.calls % len(self.per_view)]
        self.calls += 1
        data["mkpts0_f"], data["mkpts1_f"] = torch.tensor(m0, dtype=torch.float32), torch.tensor(m1, dtype=torch.float32)


def test_detector_control_flow_with_a_replayed_matcher():
    rng = np.random.default_rng(1)
    H, W = 240, 320
    src = rng.random((200, 2)) * np.array([W, H])
    A = np.array([[0.6, 0.0, 40.0], [0.0, 0.6, 30.0]])
    good = (src, src @ A[:, :2].T + A[:, 2] + 0.3 * rng.normal(size=src.shape))
    few = (src[:4], src[:4])                                                         # < 6 matches: fallback box
    noisy = (src, rng.random((200, 2)) * np.array([W, H]))                           # no consistent affinity: few inliers
    det = detector.LocalFeatureObjectDetector(_FakeMatcher([few, noisy, good]), [np.zeros((H, W), np.uint8)] * 3, device="cpu")
    query = torch.zeros(1, 1, H, W)
    res = det.match_worker(query)                        # ONE matcher call over the three views ...
    assert det.matcher.calls == 1 and det.matcher.batched_calls == 1
    det.matcher.calls = 0
    loop = det.match_worker(query, batched=False)        # ... gives the votes of the reference's view-by-view loop
    assert det.matcher.calls == 3
    for k in range(3):
        assert np.array_equal(res[k]["bbox"], loop[k]["bbox"]) and np.array_equal(np.asarray(res[k]["inliers"]), np.asarray(loop[k]["inliers"]))
    assert res[0]["bbox"].tolist() == [W // 2 - 500, H // 2 - 500, W // 2 + 500, H // 2 + 500] and res[0]["inliers"].sum() == 0
    assert res[2]["inliers"].sum() > 190 and res[1]["inliers"].sum() < 30
    want = lo.box_from_affine(lo.affine_lstsq(good[0], good[1]), (H, W))
    assert np.abs(res[2]["bbox"] - want).max() <= 1
    det.matcher.calls = 0
    assert np.array_equal(det.detect_by_matching(query), res[2]["bbox"])
    assert det.previous_pose_detect(np.array([[500.0, 0, 160], [0, 500.0, 120], [0, 0, 1]]), np.eye(4)[:3] + np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1.0]]),
                                    0.1 * np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], dtype=np.float64)).shape == (4,)
