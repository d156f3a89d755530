"""GPU parity of the padded / resized query image inputs (``query_image_mask`` / ``query_image_scale``): the reference's branches at
``OnePosePlusModel.py:156-158``, ``transformer.py:148-159``, ``linear_attention.py:49-53``, ``coarse_matching.py:108-114,224`` and
``fine_matching.py:104``.  Stage level against the oracle, whole path against the golden captured from the reference itself
(``tests/golden/b2_masked_scaled_feature_boundary.npz``, made by ``make_golden.py --only-masked``)."""
import os

import numpy as np
import pytest
import torch

from oracle import onepose_oracle as orc
from onepose_st_amd import hip, packing
from onepose_st_amd.synthetic import make_synthetic_inputs
from tests.test_gpu_parity import TOL, _check_against, _model, _planted_features, close, dev, model, to_dev  # noqa: F401  (dev, model: fixtures)

pytestmark = pytest.mark.gpu


def _mask(B, L2, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.rand(B, L2, generator=g) > 0.3
    m[:, 0] = True
    if L2 > 40:
        m[0, 17:40] = False                      # a whole run of padded cells (covers whole MFMA row groups)
    return m


@pytest.mark.parametrize("cross", [0, 1])
@pytest.mark.parametrize("B,L3,L2", [(1, 64, 32), (2, 70, 45), (3, 49, 97), (1, 1000, 1200)])
@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
def test_encoder_layer_masked(sd, dev, mode, cross, B, L3, L2):
    """one encoder layer with the 2D stream's padding mask, in the three arithmetic modes"""
    g = torch.Generator().manual_seed(2)
    x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
    mask = _mask(B, L2, 11)
    p = "loftr_coarse.layers.2."
    if cross:
        r2, r3 = orc.encoder_layer(sd, p, x2, x3, 8, x_mask=mask), orc.encoder_layer(sd, p, x3, x2, 8, source_mask=mask)
    else:
        r2, r3 = orc.encoder_layer(sd, p, x2, x2, 8, mask, mask), orc.encoder_layer(sd, p, x3, x3, 8)
    unmasked2 = orc.encoder_layer(sd, p, x2, x3 if cross else x2, 8)
    assert (unmasked2 - r2).abs().max() > 1e-2           # the mask matters in this case
    d3, d2, dm = x3.to(dev), x2.to(dev), mask.to(dev).view(torch.uint8)
    y3, y2 = torch.full_like(d3, float("nan")), torch.full_like(d2, float("nan"))
    P, S = hip.ptr, hip.stream_handle()
    lib = hip.load()
    if mode == "f32":
        w = packing.pack_coarse_layer(sd, p).to(dev)
        ws = torch.empty(lib.ophip_encoder_workspace_floats(B, L3, L2), device=dev)
        hip.call("ophip_encoder_layer_masked", P(d3), P(d2), P(y3), P(y2), B, L3, L2, P(w), cross, P(ws), P(dm, torch.uint8), S)
        tol = dict(rtol=1e-4, atol=2e-5)
    elif mode == "bf16x3":
        w = packing.pack_coarse_layer_x3w8(sd, p).to(dev)
        ws = torch.empty(lib.ophip_encoder_x3w8_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
        hip.call("ophip_encoder_layer_x3w8_masked", P(d3), P(d2), P(y3), P(y2), B, L3, L2, P(w, None), None, cross, 0, 0, P(ws, None),
                 P(dm, torch.uint8), S)
        tol = dict(rtol=3e-4, atol=1e-4)
    else:
        w = packing.pack_coarse_layer_bf16(sd, p).to(dev)
        ws = torch.empty(lib.ophip_encoder_bf16_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
        hip.call("ophip_encoder_layer_bf16_masked", P(d3), P(d2), P(y3), P(y2), B, L3, L2, P(w, None), None, 1, cross, 0, 0, P(ws, None),
                 P(dm, torch.uint8), S)
        tol = dict(rtol=5e-2, atol=5e-2)
    print(f"{mode} cross={cross} B={B} L=({L3},{L2}): max abs err 3D {(y3.cpu() - r3).abs().max():.3e} 2D {(y2.cpu() - r2).abs().max():.3e}")
    close(y3, r3, msg="3D stream", **tol)
    close(y2, r2, msg="2D stream", **tol)


def test_masked_entry_points_refuse_null(dev):
    x = torch.zeros(1, 32, 256, device=dev)
    y = torch.zeros_like(x)
    w = torch.zeros(16, device=dev)
    with pytest.raises(ValueError):
        hip.call("ophip_encoder_layer_masked", hip.ptr(x), hip.ptr(x), hip.ptr(y), hip.ptr(y), 1, 32, 32, hip.ptr(w), 0, hip.ptr(w), None,
                 hip.stream_handle())


@pytest.mark.parametrize("nsplit,rtol", [(0, 1e-4), (3, 1e-3)])
@pytest.mark.parametrize("B,N,hc,wc", [(1, 300, 10, 13), (2, 333, 12, 17), (1, 1000, 30, 40)])
def test_coarse_match_masked_scaled_vs_oracle(dev, B, N, hc, wc, nsplit, rtol):
    """coarse matching with a column mask and per-image scales: indices and the scaled coarse keypoints bit-exact, padded columns of
    conf_matrix exactly zero; the lazy form (no conf buffer) agrees bit for bit with the eager one"""
    M = hc * wc
    f3, f2 = _planted_features(B, N, hc, wc, 5, min(N, M) // 2)
    kp = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(4))
    mask = torch.ones(B, hc, wc, dtype=torch.bool)
    mask[0, :, wc - 3:] = False
    mask[B - 1, hc - 2:, :] = False
    qs = torch.tensor([[1.25, 1.5], [0.8, 1.0], [2.0, 3.0]])[:B].contiguous()
    conf_ref = orc.dual_softmax_confidence(f3, f2, 0.08, mask_query=mask.flatten(1))
    want = orc.coarse_match_select(conf_ref, (hc, wc), (hc * 8, wc * 8), kp, 0.1, 2, query_image_scale=qs)
    assert len(want["i_ids"]) > 20
    outs = []
    for lazy in ([False] if nsplit == 0 else [False, True]):
        cap = B * N
        P = hip.ptr
        d3, d2, dk = f3.to(dev), f2.to(dev), kp.to(dev)
        dm, dq = mask.flatten(1).to(dev).view(torch.uint8).contiguous(), qs.to(dev)
        conf = None if lazy else torch.full((B, N, M), float("nan"), device=dev)
        ws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev)
        ids = torch.zeros(4, cap, dtype=torch.int64, device=dev)
        mconf, mk3, mkc = torch.zeros(cap, device=dev), torch.zeros(cap, 3, device=dev), torch.zeros(cap, 2, device=dev)
        gtm, cnt = torch.zeros(cap, dtype=torch.bool, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
        hip.call("ophip_coarse_match_masked", P(d3), P(d2), P(dk), N * 3, B, N, M, wc, 0.08, 0.1, 2, 8.0, P(conf), P(ws),
                 P(ids[0], torch.int64), P(ids[1], torch.int64), P(ids[2], torch.int64), P(mconf), P(mk3), P(mkc), P(ids[3], torch.int64),
                 P(gtm, torch.bool), P(cnt, torch.int32), nsplit, 3, P(dm, torch.uint8), P(dq), hip.stream_handle())
        torch.cuda.synchronize()
        K = int(cnt[0])
        assert int(cnt[1]) == 0
        outs.append((ids[:, :K].cpu(), mconf[:K].cpu(), mkc[:K].cpu()))
        assert K == len(want["i_ids"])
        np.testing.assert_array_equal(ids[0, :K].cpu().numpy(), want["b_ids"].numpy())
        np.testing.assert_array_equal(ids[1, :K].cpu().numpy(), want["i_ids"].numpy())
        np.testing.assert_array_equal(ids[2, :K].cpu().numpy(), want["j_ids"].numpy())
        np.testing.assert_array_equal(mkc[:K].cpu().numpy(), want["mkpts_query_c"].numpy())          # bit-exact: same three roundings
        np.testing.assert_allclose(mconf[:K].cpu().numpy(), want["mconf"].numpy(), rtol=rtol, atol=1e-6)
        if conf is not None:
            dead = ~mask.flatten(1)
            assert float(conf.cpu().transpose(1, 2)[dead].abs().max()) == 0.0
            np.testing.assert_allclose(conf.cpu().numpy(), conf_ref.numpy(), rtol=rtol, atol=1e-6)
    if len(outs) == 2:
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)


def _masked_inputs(sd, cfg, g):
    i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
    i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
    both = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
    both["image_hw"] = i0["image_hw"]
    both["query_image_mask"] = torch.from_numpy(g["query_image_mask"])
    both["query_image_scale"] = torch.from_numpy(g["query_image_scale"])
    return both


def _run_masked(m, both, dev):
    d = to_dev(both, dev)
    data = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "query_image_mask", "query_image_scale")}
    m.forward_features(data, d["feat_c"], d["feat_f"], both["image_hw"])
    return data


def test_b2_masked_scaled_against_reference_golden(model, sd, cfg, dev, golden_dir):
    """the whole path with both optional inputs against the golden captured from the reference's forward"""
    g = np.load(os.path.join(golden_dir, "b2_masked_scaled_feature_boundary.npz"))
    both = _masked_inputs(sd, cfg, g)
    data = _run_masked(model, both, dev)
    n_sa = _check_against(data, g, model.precision, label=f"b2m_{model.precision}", want_rowmax=g["conf_rowmax"])
    if model.precision == "f32":
        assert n_sa == 0
    conf = data["conf_matrix"]
    dead = ~both["query_image_mask"].flatten(1)
    assert float(conf.cpu().transpose(1, 2)[dead].abs().max()) == 0.0                 # padded cells: confidence exactly 0
    assert not dead[data["b_ids"].cpu(), data["j_ids"].cpu()].any()
    rt_conf = TOL[model.precision][2]
    np.testing.assert_allclose(conf.max(dim=2)[0][0].cpu().numpy(), g["conf_rowmax"], rtol=rt_conf, atol=1e-6)
    np.testing.assert_allclose(conf.max(dim=2)[0][1].cpu().numpy(), g["conf_rowmax_b1"], rtol=rt_conf, atol=1e-6)
    np.testing.assert_allclose(conf.max(dim=1)[0][0].cpu().numpy(), g["conf_colmax"], rtol=rt_conf, atol=1e-6)
    # the scales act per batch element: without them the keypoints differ, the indices do not
    plain = dict(both)
    del plain["query_image_scale"]
    d = to_dev(plain, dev)
    data2 = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "query_image_mask")}
    model.forward_features(data2, d["feat_c"], d["feat_f"], both["image_hw"])
    assert torch.equal(data2["j_ids"], data["j_ids"]) and not torch.equal(data2["mkpts_query_f"], data["mkpts_query_f"])
    # wrong shapes are refused before anything is launched
    bad = dict(data)
    bad["query_image_mask"] = data["query_image_mask"][:, :-1]
    with pytest.raises(ValueError):
        model.forward_features(bad, d["feat_c"], d["feat_f"], both["image_hw"])


def test_masked_frame_goes_through_the_one_call_path_bit_identically(sd, cfg, dev, golden_dir):
    """a masked / scaled frame takes the default one-call frame path (ophip_frame_enqueue_padded, kept-back fine stage and all) and gives
    the stage-by-stage path's outputs bit for bit -- also in a pipeline with unmasked frames before and after it"""
    import copy
    from onepose_st_amd import ops
    g = np.load(os.path.join(golden_dir, "b2_masked_scaled_feature_boundary.npz"))
    both = _masked_inputs(sd, cfg, g)
    c = copy.deepcopy(cfg)
    c["hip_frame_call"] = False
    staged = _run_masked(_model(sd, c, dev, "bf16x3"), both, dev)
    m = _model(sd, cfg, dev, "bf16x3")
    before = ops.CALLS["frame_enqueue"]
    one = _run_masked(m, both, dev)
    assert ops.CALLS["frame_enqueue"] == before + 1
    keys = ("b_ids", "i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "expec_f", "mkpts_3d_db", "conf_matrix")
    for k in keys:
        assert torch.equal(one[k], staged[k]), k
    # pipeline: unmasked, masked, unmasked in flight together
    d = to_dev(both, dev)
    plain = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    alone = dict(plain)
    m.forward_features(alone, d["feat_c"], d["feat_f"], both["image_hw"])
    datas = [dict(plain), {**plain, "query_image_mask": d["query_image_mask"], "query_image_scale": d["query_image_scale"]}, dict(plain)]
    pend = [m.enqueue_features(x, d["feat_c"], d["feat_f"], both["image_hw"], host_copy=True) for x in datas]
    for p in (pend[2], pend[0], pend[1]):
        p.finish()
    for k in keys[:-1]:
        assert torch.equal(datas[0][k], alone[k]) and torch.equal(datas[2][k], alone[k]), k
        assert torch.equal(datas[1][k], one[k]), k
    assert not torch.equal(alone["j_ids"], one["j_ids"])


def test_b2_masked_lazy_and_bf16_modes(sd, cfg, dev, golden_dir):
    """lazy conf_matrix with a mask == the eager form bit for bit; the plain-bf16 mode keeps the match set up to margin-free matches"""
    import copy
    g = np.load(os.path.join(golden_dir, "b2_masked_scaled_feature_boundary.npz"))
    both = _masked_inputs(sd, cfg, g)
    eager = _run_masked(_model(sd, cfg, dev, "bf16x3"), both, dev)
    c = copy.deepcopy(cfg)
    c["hip_conf_matrix"] = "lazy"
    lazy = _run_masked(_model(sd, c, dev, "bf16x3"), both, dev)
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "expec_f"):
        assert torch.equal(eager[k], lazy[k]), k
    assert torch.equal(lazy["conf_matrix"].materialize(), eager["conf_matrix"])
    b16 = _run_masked(_model(sd, cfg, dev, "bf16"), both, dev)
    got = set(zip(b16["b_ids"].tolist(), b16["i_ids"].tolist(), b16["j_ids"].tolist()))
    want = set(zip(g["b_ids"].tolist(), g["i_ids"].tolist(), g["j_ids"].tolist()))
    print(f"bf16 masked: K={len(got)} reference K={len(want)} symmetric difference={len(got ^ want)}")
    assert len(got ^ want) <= max(2, 0.03 * len(want))
