"""CPU restatement of the LoFTR 2D-2D matcher and of the detector rules built on it -- TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module.  SURVEY.md section 8f-3: the reference's object detector
(``src/local_feature_object_detector/local_feature_2D_detector.py:40-280``) runs ``LoFTR_for_OnePose_Plus``
(``src/KeypointFreeSfM/loftr_for_sfm/loftr.py:16-167``) whose sub-modules are imported from ``submodules/LoFTR/src/loftr`` --
an un-vendored git submodule (``.gitmodules`` names https://github.com/zju3dv/LoFTR, no pinned commit, empty directory in
``/root/reference``).  Its arithmetic is therefore restated here from the PUBLISHED definition of zju3dv/LoFTR (``src/loftr/``:
``loftr_module/transformer.py``, ``linear_attention.py``, ``fine_preprocess.py``, ``utils/coarse_matching.py``,
``utils/fine_matching.py``, ``utils/position_encoding.py``), anchored on what the reference tree itself holds:

* the call sequence and the dict keys of ``loftr.py:33-135`` (this file follows it statement by statement),
* the constants of ``loftr_for_onepose_plus_cfg.py:10-50`` (window 9, 8 coarse layers, thr 0.2, temperature 0.1, border 2,
  ``temp_bug_fix`` False),
* the detector's control flow ``local_feature_2D_detector.py:89-144`` (< 6 matches, affine RANSAC threshold 6, integer box).

**Parity unpinned**: no fixture of the reference pins these outputs and the submodule cannot be run.  Differences from the
OnePose++ in-tree twin (``src/models/OnePosePlus``) that this restatement takes from the LoFTR definition: the ``cross`` layers
update image 0 first and image 1 then attends to the UPDATED image 0; the similarity is divided by the temperature itself
(no ``+ 1e-4``); ``mask_border`` clears all four sides of both grids; the fine windows are taken on both images.
``cv2.estimateAffine2D`` (absent here) is restated as a plain RANSAC over 3-point affinities + least squares on the inliers:
same model and threshold rule, not OpenCV's sampling sequence.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import onepose_oracle as orc


def loftr_default_cfg() -> dict:
    """``loftr_for_onepose_plus_cfg.py:10-50`` through ``lower_config``."""
    return {
        "backbone_type": "ResNetFPN", "resolution": (8, 2), "fine_window_size": 9, "fine_concat_coarse_feat": False,
        "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]},
        "coarse": {"d_model": 256, "d_ffn": 256, "nhead": 8, "layer_names": ["self", "cross"] * 4, "attention": "linear", "temp_bug_fix": False},
        "match_coarse": {"thr": 0.2, "border_rm": 2, "match_type": "dual_softmax", "dsmax_temperature": 0.1, "skh_iters": 3,
                         "skh_init_bin_score": 1.0, "skh_prefilter": True, "train_coarse_percent": 0.4, "train_pad_num_gt_min": 200},
        "fine": {"d_model": 128, "d_ffn": 128, "nhead": 8, "layer_names": ["self", "cross"] * 1, "attention": "linear"},
    }


def transformer_two_images(sd: dict, prefix: str, layer_names: list, nhead: int, feat0: torch.Tensor, feat1: torch.Tensor):
    """LoFTR ``LocalFeatureTransformer.forward``: self -> both streams on themselves; cross -> image 0 against image 1, THEN
    image 1 against the already updated image 0 (the statements are sequential in the published code)."""
    for i, name in enumerate(layer_names):
        p = f"{prefix}.layers.{i}."
        if name == "self":
            feat0 = orc.encoder_layer(sd, p, feat0, feat0, nhead)
            feat1 = orc.encoder_layer(sd, p, feat1, feat1, nhead)
        elif name == "cross":
            feat0 = orc.encoder_layer(sd, p, feat0, feat1, nhead)
            feat1 = orc.encoder_layer(sd, p, feat1, feat0, nhead)
        else:
            raise KeyError(name)
    return feat0, feat1


def coarse_matching(feat_c0, feat_c1, hw0_c, hw1_c, hw0_i, cfg) -> dict:
    """LoFTR ``CoarseMatching.forward`` + ``get_coarse_match`` (dual-softmax branch, inference)."""
    C = feat_c0.shape[-1]
    f0, f1 = feat_c0 / C ** 0.5, feat_c1 / C ** 0.5
    sim = torch.einsum("nlc,nsc->nls", f0, f1) / cfg["dsmax_temperature"]
    conf = F.softmax(sim, 1) * F.softmax(sim, 2)
    B = conf.shape[0]
    h0, w0 = hw0_c
    h1, w1 = hw1_c
    mask = (conf > cfg["thr"]).view(B, h0, w0, h1, w1).clone()
    b = cfg["border_rm"]
    if b > 0:
        mask[:, :b] = False
        mask[:, :, :b] = False
        mask[:, :, :, :b] = False
        mask[:, :, :, :, :b] = False
        mask[:, -b:] = False
        mask[:, :, -b:] = False
        mask[:, :, :, -b:] = False
        mask[:, :, :, :, -b:] = False
    mask = mask.view(B, h0 * w0, h1 * w1)
    mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = hw0_i[0] / h0
    mk0 = torch.stack([i_ids % w0, i_ids // w0], dim=1).float() * scale
    mk1 = torch.stack([j_ids % w1, j_ids // w1], dim=1).float() * scale
    return {"conf_matrix": conf, "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "mconf": mconf, "mkpts0_c": mk0, "mkpts1_c": mk1}


def fine_windows(feat_f, ids_b, ids_cell, hw_c, W):
    """LoFTR ``FinePreprocess``: ``F.unfold(kernel W, stride = hf / hc, padding W // 2)`` -> ``[n, l, ww, c]`` -> the matches' cells."""
    stride = feat_f.shape[2] // hw_c[0]
    C = feat_f.shape[1]
    u = F.unfold(feat_f, kernel_size=(W, W), stride=stride, padding=W // 2)               # [n, c * ww, l]
    u = u.view(feat_f.shape[0], C, W * W, -1).permute(0, 3, 2, 1)                          # n l ww c
    return u[ids_b, ids_cell]


def fine_matching(feat_f0, feat_f1, mkpts0_c, mkpts1_c, hw0_i, hw0_f) -> dict:
    """LoFTR ``FineMatching.forward`` (kornia's spatial_expectation2d / create_meshgrid restated: normalised grid linspace(-1, 1, W))."""
    M, WW, C = feat_f0.shape
    W = int(math.sqrt(WW))
    scale = hw0_i[0] / hw0_f[0]
    if M == 0:
        return {"expec_f": torch.empty(0, 3), "mkpts0_f": mkpts0_c, "mkpts1_f": mkpts1_c}
    picked = feat_f0[:, WW // 2, :]
    sim = torch.einsum("mc,mrc->mr", picked, feat_f1)
    heat = torch.softmax(sim / C ** 0.5, dim=1)
    xs = torch.linspace(-1, 1, W)
    gx = xs.repeat(W)                      # x varies fastest (row-major window)
    gy = xs.repeat_interleave(W)
    ex, ey = (heat * gx).sum(1), (heat * gy).sum(1)
    var_x = (heat * gx ** 2).sum(1) - ex ** 2
    var_y = (heat * gy ** 2).sum(1) - ey ** 2
    std = torch.sqrt(torch.clamp(var_x, min=1e-10)) + torch.sqrt(torch.clamp(var_y, min=1e-10))
    coords = torch.stack([ex, ey], 1)
    return {"expec_f": torch.cat([coords, std[:, None]], 1), "mkpts0_f": mkpts0_c, "mkpts1_f": mkpts1_c + coords * (W // 2) * scale}


def loftr_forward(sd: dict, cfg: dict, image0: torch.Tensor, image1: torch.Tensor, feature_hook=None) -> dict:
    """``LoFTR_for_OnePose_Plus.forward`` (``loftr.py:33-135``) without masks / scales / provided coarse matches.
    ``feature_hook(f0 [1, L0, 256], ff0 [1, 128, hf, wf], f1, ff1)`` may replace the backbone-output tensors (coarse rows after the
    positional encoding): random-weight backbones give no matches, tests plant them there."""
    out = {"bs": image0.size(0), "hw0_i": tuple(image0.shape[2:]), "hw1_i": tuple(image1.shape[2:])}
    fc0, ff0 = orc.backbone_8_2(sd, image0)
    fc1, ff1 = orc.backbone_8_2(sd, image1)
    out.update({"hw0_c": tuple(fc0.shape[2:]), "hw1_c": tuple(fc1.shape[2:]), "hw0_f": tuple(ff0.shape[2:]), "hw1_f": tuple(ff1.shape[2:])})
    pe = orc.position_table(cfg["coarse"]["d_model"])          # temp_bug_fix False: the floor-division table
    f0, f1 = orc.pe_add_flatten(fc0, pe), orc.pe_add_flatten(fc1, pe)
    if feature_hook is not None:
        f0, ff0, f1, ff1 = feature_hook(f0, ff0, f1, ff1)
    f0, f1 = transformer_two_images(sd, "loftr_coarse", cfg["coarse"]["layer_names"], cfg["coarse"]["nhead"], f0, f1)
    out["feat_c0"], out["feat_c1"] = f0, f1
    out.update(coarse_matching(f0, f1, out["hw0_c"], out["hw1_c"], out["hw0_i"], cfg["match_coarse"]))
    W = cfg["fine_window_size"]
    w0 = fine_windows(ff0, out["b_ids"], out["i_ids"], out["hw0_c"], W)
    w1 = fine_windows(ff1, out["b_ids"], out["j_ids"], out["hw1_c"], W)
    if w0.size(0) != 0:
        w0, w1 = transformer_two_images(sd, "loftr_fine", cfg["fine"]["layer_names"], cfg["fine"]["nhead"], w0, w1)
    out["fine_f0"], out["fine_f1"] = w0, w1
    out.update(fine_matching(w0, w1, out["mkpts0_c"], out["mkpts1_c"], out["hw0_i"], out["hw0_f"]))
    return out


# ----------------------------------------------------------------------------------------------
# detector rules (local_feature_2D_detector.py:89-162)
# ----------------------------------------------------------------------------------------------
def affine_from_3(src, dst):
    """exact affinity through three correspondences: ``[a | t]`` (2 x 3) with ``dst = a src + t``; None when collinear"""
    A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
    if abs(np.linalg.det(A)) < 1e-12:
        return None
    return np.linalg.solve(A, np.asarray(dst, np.float64)).T


def affine_lstsq(src, dst):
    A = np.concatenate([np.asarray(src, np.float64), np.ones((len(src), 1))], 1)
    sol, *_ = np.linalg.lstsq(A, np.asarray(dst, np.float64), rcond=None)
    return sol.T


def estimate_affine2d(src, dst, thr=6.0, iters=2000, seed=0):
    """RANSAC over 3-point affinities, inlier = |A p - q| < thr, least squares on the best inlier set (the model, threshold rule and
    defaults of ``cv2.estimateAffine2D(method=RANSAC, ransacReprojThreshold=6)``: maxIters 2000)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    n = len(src)
    rng = np.random.default_rng(seed)
    best = None
    for _ in range(iters):
        idx = rng.choice(n, 3, replace=False)
        A = affine_from_3(src[idx], dst[idx])
        if A is None:
            continue
        err = np.linalg.norm(src @ A[:, :2].T + A[:, 2] - dst, axis=1)
        inl = err < thr
        if best is None or inl.sum() > best.sum():
            best = inl
    if best is None or best.sum() < 3:
        return None, np.zeros(n, np.uint8)
    A = affine_lstsq(src[best], dst[best])
    return A, best.astype(np.uint8)


def box_from_affine(affine, img0_hw):
    """``match_worker``: the four corners of the reference view through the affinity, truncated to int32, min / max box"""
    H, W = img0_hw
    corners = np.array([[0, 0, 1], [W, 0, 1], [0, H, 1], [W, H, 1]], np.float64).T
    box = (affine @ corners).T.astype(np.int32)
    lt, rb = box.min(axis=0), box.max(axis=0)
    return np.array([lt[0], lt[1], rb[0], rb[1]])


def fallback_box(query_hw):
    H, W = query_hw
    cx, cy = W // 2, H // 2
    return np.array([cx - 500, cy - 500, cx + 500, cy + 500])


def pick_detection(results: list):
    """``detect_by_matching``: the view with the most inliers wins; Python's stable sort keeps the FIRST among equals"""
    order = sorted(range(len(results)), key=lambda k: results[k]["inliers"].sum(), reverse=True)
    return results[order[0]]["bbox"]
