"""Row f-3 on the MI355X: the LoFTR-specific kernels (csrc/loftr_fine.hip, ophip_coarse_match_2d) against the CPU restatement of the
published LoFTR definition (oracle/loftr_oracle.py; parity unpinned: the submodule is not vendored), the whole matcher against it on
planted backbone features and through the real backbone, and the detector end to end.  All calls go through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import loftr_oracle as lo
from oracle import onepose_oracle as orc
from onepose_st_amd import detector, hip, loftr, packing
from onepose_st_amd.synthetic import make_synthetic_loftr_state_dict
from tests.loftr_helpers import device_hook, oracle_hook, planted_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    hip.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lsd():
    return make_synthetic_loftr_state_dict(0)


@pytest.fixture(scope="module")
def matcher(lsd, dev):
    m = loftr.LoFTR_for_OnePose_Plus().eval()
    m.load_state_dict(lsd, strict=True)
    return m.to(dev)


def close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("ka,kb,nout,relu", [(128, 0, 128, False), (128, 128, 256, True), (256, 0, 128, False), (128, 0, 256, True), (64, 64, 128, False)])
@pytest.mark.parametrize("T", [1, 81, 1000])
def test_rows_linear_x3(dev, ka, kb, nout, relu, T):
    g = torch.Generator().manual_seed(T + ka)
    xa, xb = torch.randn(T, ka, generator=g), (torch.randn(T, kb, generator=g) if kb else None)
    w = torch.randn(nout, ka + kb, generator=g) / (ka + kb) ** 0.5
    ref = torch.cat([xa, xb], 1) @ w.T if kb else xa @ w.T
    if relu:
        ref = ref.relu()
    wp = packing.pack_linear_x3(w).to(dev)
    assert wp.numel() == hip.load().ophip_rows_linear_wpack_bytes(ka + kb, nout)
    y = torch.full((T, nout), float("nan"), device=dev)
    da, db = xa.to(dev), (xb.to(dev) if kb else None)
    hip.call("ophip_rows_linear_x3", hip.ptr(da), ka, hip.ptr(db), kb, T, hip.ptr(wp, None), nout, int(relu), hip.ptr(y), hip.stream_handle())
    close(y, ref, 2e-4, 2e-5, "split-bf16 GEMM")
    with pytest.raises(ValueError):
        hip.call("ophip_rows_linear_x3", hip.ptr(da), 96, None, 0, T, hip.ptr(wp, None), nout, 0, hip.ptr(y), hip.stream_handle())


def test_fine2_gather_attention_layernorm_match(dev, lsd):
    g = torch.Generator().manual_seed(9)
    hc, wc, W, K = 7, 9, 9, 23
    hf, wf = 4 * hc, 4 * wc
    feat = torch.randn(1, 128, hf, wf, generator=g)
    cells = torch.randint(0, hc * wc, (K,), generator=g)
    cells[:4] = torch.tensor([0, wc - 1, (hc - 1) * wc, hc * wc - 1])                 # corners: zero padding of the unfold
    ref_w = lo.fine_windows(feat, torch.zeros(K, dtype=torch.long), cells, (hc, wc), W)
    cl = feat[0].permute(1, 2, 0).reshape(-1, 128).contiguous().to(dev)
    out = torch.full((K, W * W, 128), float("nan"), device=dev)
    dc = cells.to(dev)
    hip.call("ophip_fine2_gather", hip.ptr(cl), hf, wf, hip.ptr(dc, torch.int64), K, wc, 4, W, hip.ptr(out), hip.stream_handle())
    assert torch.equal(out.cpu(), ref_w)
    # linear attention, 8 heads of 16, queries of one stream against another source
    q, k, v = (torch.randn(K, 81, 128, generator=g) for _ in range(3))
    ref_a = orc.linear_attention(q.view(K, 81, 8, 16), k.view(K, 81, 8, 16), v.view(K, 81, 8, 16)).reshape(K, 81, 128)
    msg = torch.empty(K, 81, 128, device=dev)
    dq, dk, dv = q.to(dev), k.to(dev), v.to(dev)
    hip.call("ophip_fine2_attention", hip.ptr(dq), hip.ptr(dk), hip.ptr(dv), K, 81, 81, hip.ptr(msg), hip.stream_handle())
    close(msg, ref_a, 1e-4, 1e-5, "linear attention")
    # LayerNorm with and without the residual
    x, res = torch.randn(K * 81, 128, generator=g) * 3 + 1, torch.randn(K * 81, 128, generator=g)
    gam, bet = lsd["loftr_fine.layers.0.norm2.weight"], lsd["loftr_fine.layers.0.norm2.bias"]
    ref_l = torch.nn.functional.layer_norm(x, (128,), gam, bet, 1e-5)
    y = torch.empty(K * 81, 128, device=dev)
    dx, dr, dg, db = x.to(dev), res.to(dev), gam.to(dev), bet.to(dev)
    hip.call("ophip_rows_layernorm128", hip.ptr(dx), hip.ptr(dg), hip.ptr(db), None, K * 81, hip.ptr(y), hip.stream_handle())
    close(y, ref_l, 1e-5, 1e-5, "LayerNorm")
    hip.call("ophip_rows_layernorm128", hip.ptr(dx), hip.ptr(dg), hip.ptr(db), hip.ptr(dr), K * 81, hip.ptr(y), hip.stream_handle())
    close(y, res + ref_l, 1e-5, 1e-5, "residual + LayerNorm")
    # fine matching: centre token of image 0 against image 1's window
    f0, f1 = torch.randn(K, 81, 128, generator=g), torch.randn(K, 81, 128, generator=g)
    f1[:, 50] = 2.0 * f0[:, 40]                                                        # a peak one pixel right and down of the centre
    mk1 = torch.rand(K, 2, generator=g) * 100
    ref_m = lo.fine_matching(f0, f1, mk1, mk1, (hc * 8, wc * 8), (hf, wf))
    expec, mk1f = torch.empty(K, 3, device=dev), torch.empty(K, 2, device=dev)
    d0, d1, dm = f0.to(dev), f1.to(dev), mk1.to(dev)
    hip.call("ophip_fine2_match", hip.ptr(d0), hip.ptr(d1), hip.ptr(dm), K, W, 4 * 2.0, hip.ptr(expec), hip.ptr(mk1f), hip.stream_handle())
    close(expec[:, :2], ref_m["expec_f"][:, :2], 1e-4, 1e-5, "expectation")
    close(expec[:, 2], ref_m["expec_f"][:, 2], 1e-3, 1e-3, "std (ill-conditioned for peaked heat maps)")
    close(mk1f, ref_m["mkpts1_f"], 1e-5, 1e-4, "mkpts1_f")


def test_coarse_match_2d_against_the_published_rules(dev):
    """temperature exactly 0.1, threshold 0.2, border removal on all four sides of BOTH grids, mutual nearest, ascending order"""
    g = torch.Generator().manual_seed(4)
    h0, w0, h1, w1 = 9, 11, 8, 13
    L0, L1 = h0 * w0, h1 * w1
    f0, f1 = torch.randn(1, L0, 256, generator=g) * 1.5, torch.randn(1, L1, 256, generator=g)
    cell0 = lambda y, x: y * w0 + x
    cell1 = lambda y, x: y * w1 + x
    plant = [((4, 5), (3, 6)), ((2, 2), (5, 10)), ((6, 8), (2, 2)),            # interior pairs: kept
             ((1, 5), (4, 4)), ((4, 1), (4, 5)), ((7, 5), (4, 6)), ((4, 9), (4, 7)),        # image-0 side in the border (top, left, bottom, right): dropped
             ((3, 3), (6, 4)), ((3, 4), (3, 11)), ((5, 5), (1, 6)), ((5, 6), (3, 1))]       # image-1 side in the border: dropped
    for (a, b) in plant:
        f1[0, cell1(*b)] = f0[0, cell0(*a)] * 1.5
    cfg = lo.loftr_default_cfg()["match_coarse"]
    ref = lo.coarse_matching(f0, f1, (h0, w0), (h1, w1), (h0 * 8, w0 * 8), cfg)
    assert sorted(ref["i_ids"].tolist()) == sorted(cell0(*a) for a, _ in plant[:3])
    ii = torch.arange(L0)
    pts0 = torch.stack([(ii % w0).float() * 8, (ii // w0).float() * 8, torch.zeros(L0)], 1)[None].contiguous().to(dev)
    d0, d1 = f0.to(dev), f1.to(dev)
    conf = torch.empty(1, L0, L1, device=dev)
    ws = torch.empty(hip.load().ophip_coarse_workspace_floats(1, L0, L1), device=dev)
    ids = [torch.empty(L0, dtype=torch.int64, device=dev) for _ in range(3)]
    mconf, mk0, mk1 = torch.empty(L0, device=dev), torch.empty(L0, 3, device=dev), torch.empty(L0, 2, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for conf_buf in (conf, None):                                               # eager and lazy form
        hip.call("ophip_coarse_match_2d", hip.ptr(d0), hip.ptr(d1), hip.ptr(pts0), 0, 1, L0, L1, w0, w1, 0.1, 0.2, 2, 8.0, hip.ptr(conf_buf), hip.ptr(ws),
                 *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk0), hip.ptr(mk1), None, None, hip.ptr(cnt, torch.int32), 3,
                 hip.stream_handle())
        K = int(cnt[0])
        assert K == 3 and ids[1][:K].tolist() == ref["i_ids"].tolist() and ids[2][:K].tolist() == ref["j_ids"].tolist()
        close(mconf[:K], ref["mconf"], 1e-3, 1e-5)
        assert torch.equal(mk0[:K, :2].cpu(), ref["mkpts0_c"]) and torch.equal(mk1[:K].cpu(), ref["mkpts1_c"])
    close(conf, ref["conf_matrix"], 1e-3, 1e-6, "conf_matrix (similarity / 0.1 exactly)")


def _run(m, dev, img0, img1, hook=None):
    m.feature_hook = hook
    data = {"image0": img0.to(dev), "image1": img1.to(dev)}
    m(data, _debug=True)
    m.feature_hook = None
    return data


def test_matcher_on_planted_features_against_the_oracle(matcher, lsd, dev):
    H, W = 96, 128
    pair = planted_pair((H, W))
    img = torch.zeros(1, 1, H, W)
    with torch.no_grad():
        ref = lo.loftr_forward(lsd, lo.loftr_default_cfg(), img, img, feature_hook=oracle_hook(pair))
    data = _run(matcher, dev, img, img, device_hook(pair, dev))
    K = len(ref["i_ids"])
    assert K >= 40 and float((ref["mconf"] - 0.2).abs().min()) > 0.05            # every reference confidence far from the threshold
    assert data["i_ids"].tolist() == ref["i_ids"].tolist() and data["j_ids"].tolist() == ref["j_ids"].tolist()
    assert tuple(data["hw0_c"]) == (12, 16) and tuple(data["hw1_f"]) == (48, 64) and data["bs"] == 1
    close(data["_feat_c0"], ref["feat_c0"], 2e-3, 1e-3, "coarse rows of image 0 after 8 layers (sequential cross)")
    close(data["_feat_c1"], ref["feat_c1"], 2e-3, 1e-3, "coarse rows of image 1")
    close(data["mconf"], ref["mconf"], 2e-3, 1e-5)
    assert torch.equal(data["mkpts0_c"].cpu(), ref["mkpts0_c"]) and torch.equal(data["mkpts1_c"].cpu(), ref["mkpts1_c"])
    close(data["_fine_f0"], ref["fine_f0"], 1e-3, 5e-4, "fine transformer, image 0 windows")
    close(data["_fine_f1"], ref["fine_f1"], 1e-3, 5e-4, "fine transformer, image 1 windows")
    close(data["expec_f"][:, :2], ref["expec_f"][:, :2], 1e-3, 2e-4)
    close(data["mkpts1_f"], ref["mkpts1_f"], 1e-4, 2e-3)
    assert torch.equal(data["mkpts0_f"], data["mkpts0_c"])
    d = (data["mkpts1_f"] - data["mkpts0_f"]).cpu().numpy()
    assert np.abs(np.median(d, axis=0) - np.array([16.0, 8.0])).max() < 0.5


def test_matcher_through_the_real_backbone_and_the_empty_path(matcher, lsd, dev):
    """random weights + random images: (almost) no matches; the dual-softmax matrix through backbone + 8 layers still has to agree
    with the oracle, and K = 0 returns the reference's empty shapes"""
    g = torch.Generator().manual_seed(8)
    img0, img1 = torch.rand(1, 1, 64, 96, generator=g), torch.rand(1, 1, 64, 96, generator=g)
    with torch.no_grad():
        ref = lo.loftr_forward(lsd, lo.loftr_default_cfg(), img0, img1)
    data = _run(matcher, dev, img0, img1)
    close(data["conf_matrix"].max(dim=2)[0], ref["conf_matrix"].max(dim=2)[0], 2e-2, 1e-6, "row maxima of conf_matrix")
    if len(ref["i_ids"]) == 0 and len(data["i_ids"]) == 0:
        assert data["expec_f"].shape == (0, 3) and data["mkpts0_f"].shape == (0, 2) and data["mkpts1_f"].shape == (0, 2)
    for k in ("mask0", "scale0", "mkpts0_c"):
        with pytest.raises(NotImplementedError):
            matcher({"image0": img0.to(dev), "image1": img1.to(dev), k: torch.zeros(1)})


def test_detector_end_to_end_on_the_device(matcher, dev):
    """three reference views; the query carries view 1's content moved by (2, 1) coarse cells: that view collects the inliers, its
    corners moved by (16, 8) px are the box, the crop and K_crop follow the reference's crop_img_by_bbox geometry"""
    H, W = 96, 128
    views = [np.full((H, W), 10 * (k + 1), dtype=np.uint8) for k in range(3)]
    det = detector.LocalFeatureObjectDetector(matcher, views)
    pairs = {0: planted_pair((H, W), (0, 0), seed=20, noise=3.0),                 # view 0: unrelated content (noise swamps the copy)
             1: planted_pair((H, W), (2, 1), seed=21),
             2: planted_pair((H, W), (0, 0), seed=22, noise=3.0)}
    calls = {"n": 0}

    def hook(fc0, ff0, fc1, ff1):
        if fc0.shape[0] == 3:                        # the batched call: all three views at once, every pair with its own planted query
            outs = [device_hook(pairs[k], dev)(None, None, None, None) for k in range(3)]
            return (torch.cat([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.cat([o[2] for o in outs]), torch.stack([o[3] for o in outs]))
        k = calls["n"] % 3
        calls["n"] += 1
        return device_hook(pairs[k], dev)(fc0, ff0, fc1, ff1)
    matcher.feature_hook = hook
    try:
        frame = np.zeros((H, W), dtype=np.uint8)
        res = det.match_worker(torch.zeros(1, 1, H, W, device=dev))           # ONE batched matcher call over the three views, one read-back
        assert calls["n"] == 0
        loop = det.match_worker(torch.zeros(1, 1, H, W, device=dev), batched=False)      # the reference's view-by-view form: the same votes
        assert calls["n"] == 3
        for k in range(3):
            assert np.array_equal(res[k]["bbox"], loop[k]["bbox"]) and np.array_equal(np.asarray(res[k]["inliers"]), np.asarray(loop[k]["inliers"])), k
        assert res[1]["inliers"].sum() >= 40 and res[1]["inliers"].sum() > max(res[0]["inliers"].sum(), res[2]["inliers"].sum())
        assert np.abs(res[1]["bbox"] - np.array([16, 8, W + 16, H + 8])).max() <= 1
        calls["n"] = 0
        Kmat = np.array([[300.0, 0, 64], [0, 300.0, 48], [0, 0, 1]])
        bbox, crop, K_crop, trans = det.detect(frame, Kmat, crop_size=64)
        assert np.array_equal(bbox, res[1]["bbox"]) and crop.shape == (1, 1, 64, 64) and crop.is_cuda
        s = 64 / (bbox[2] - bbox[0])
        assert abs(K_crop[0, 0] - 300.0 * s) < 1e-9 and trans.shape == (3, 3)
        calls["n"] = 0
        assert np.array_equal(det(frame, 0), bbox)                                # the SequenceRunner hook
    finally:
        matcher.feature_hook = None
