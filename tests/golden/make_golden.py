"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs ``/root/reference``; the GPU box never sees the
reference):

    python tests/golden/make_golden.py

The reference's model files are imported on CPU with behaviour-neutral import shims
for packages that are not installed here: ``loguru`` (logging),
``timm`` (a ``register_model`` decorator), ``src.utils.profiler`` (PassThroughProfiler,
avoids pytorch_lightning).  ``kornia`` 0.4.1 is not installed either; its two helper
functions used by ``fine_matching.py:86-87`` are restated below from their published
definition -- the fine-matching goldens are therefore "parity unpinned at the kornia
boundary" (the arithmetic is a 25-term weighted sum over an exactly representable
grid).

What is stored (all small): match indices, confidences, keypoints, fine expectations,
per-stage float64 checksums + probe values, and checksums of the seeded inputs so that
generator drift is detected.  Inputs and weights come from the build's own seeded
generators (``onepose_st_amd.synthetic``); the reference model loads the synthetic
``state_dict`` with ``strict=True`` which also pins the key layout.
"""
from __future__ import annotations

import contextlib
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _install_shims():
    lg = types.ModuleType("loguru")
    lg.logger = logging.getLogger("ref")
    sys.modules["loguru"] = lg
    for name in ("timm", "timm.models", "timm.models.registry"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["timm.models.registry"].register_model = lambda f: f
    prof = types.ModuleType("src.utils.profiler")

    class PassThroughProfiler:
        @contextlib.contextmanager
        def record_function(self, name):
            yield

        profile = record_function

    prof.PassThroughProfiler = PassThroughProfiler
    sys.modules["src.utils.profiler"] = prof

    # kornia 0.4.1 restatement
    def create_meshgrid(height, width, normalized_coordinates=True, device=None):
        xs = torch.linspace(0, width - 1, width, device=device)
        ys = torch.linspace(0, height - 1, height, device=device)
        if normalized_coordinates:
            xs = (xs / (width - 1) - 0.5) * 2
            ys = (ys / (height - 1) - 0.5) * 2
        base = torch.stack(torch.meshgrid([xs, ys], indexing="ij")).transpose(1, 2)   # 2 x H x W
        return base.unsqueeze(0).permute(0, 2, 3, 1)                                   # 1 x H x W x 2 (x, y)

    def spatial_expectation2d(inp, normalized_coordinates=True):
        b, c, h, w = inp.shape
        grid = create_meshgrid(h, w, normalized_coordinates, inp.device).to(inp.dtype)
        pos_x = grid[..., 0].reshape(-1)
        pos_y = grid[..., 1].reshape(-1)
        flat = inp.view(b, c, -1)
        ex = torch.sum(pos_x * flat, -1, keepdim=True)
        ey = torch.sum(pos_y * flat, -1, keepdim=True)
        return torch.cat([ex, ey], -1).view(b, c, 2)

    for name in ("kornia", "kornia.geometry", "kornia.geometry.subpix", "kornia.geometry.subpix.dsnt",
                 "kornia.utils", "kornia.utils.grid"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["kornia.geometry.subpix.dsnt"].spatial_expectation2d = spatial_expectation2d
    sys.modules["kornia.geometry.subpix"].dsnt = sys.modules["kornia.geometry.subpix.dsnt"]
    sys.modules["kornia.utils.grid"].create_meshgrid = create_meshgrid


def load_reference_model(cfg, sd):
    _install_shims()
    sys.path.insert(0, REF)
    from src.models.OnePosePlus.OnePosePlusModel import OnePosePlus_model  # noqa: E402

    model = OnePosePlus_model(cfg).eval()
    model.load_state_dict(sd, strict=True)
    return model


def checksum(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), (d * d).sum().item(), d.abs().max().item()], dtype=np.float64)


def probe(t: torch.Tensor, rows=(0, 1, 2, 3), cols=8) -> np.ndarray:
    """first rows x first cols of the last two dims of batch element 0"""
    x = t.detach()[0]
    r = [i for i in rows if i < x.shape[0]]
    return x[r][:, :cols].float().numpy().copy()


class _FeatureStub(torch.nn.Module):
    """Stands in for ``model.backbone`` so that the reference's own ``forward`` runs
    from given backbone-output tensors (the north_star boundary)."""

    def __init__(self, feat_c, feat_f):
        super().__init__()
        self.feat_c, self.feat_f = feat_c, feat_f

    def forward(self, x):
        return [self.feat_c, self.feat_f]


def run_feature_case(model, inp, extra=None):
    """Reference forward from the feature boundary, with hooks capturing stage outputs; ``extra``: further ``data`` keys
    (``query_image_mask``, ``query_image_scale``)."""
    caps = {"coarse": [], "fine": [], "kpt": [], "pe": [], "fine_in": []}
    hooks = []
    for lyr in model.loftr_coarse.layers:
        hooks.append(lyr.register_forward_hook(lambda m, i, o: caps["coarse"].append(o.detach().clone())))
    for lyr in model.loftr_fine.layers:
        hooks.append(lyr.register_forward_hook(lambda m, i, o: caps["fine"].append(o.detach().clone())))
    hooks.append(model.kpt_3d_pos_encoding.register_forward_hook(lambda m, i, o: caps["kpt"].append(o.detach().clone())))
    hooks.append(model.dense_pos_encoding.register_forward_hook(lambda m, i, o: caps["pe"].append(o.detach().clone())))
    hooks.append(model.fine_preprocess.register_forward_hook(lambda m, i, o: caps["fine_in"].append([t.detach().clone() for t in o])))
    real_backbone = model.backbone
    model.backbone = _FeatureStub(inp["feat_c"], inp["feat_f"])
    H, W = inp["image_hw"]
    data = {
        "query_image": torch.zeros(inp["feat_c"].size(0), 1, H, W),
        "keypoints3d": inp["keypoints3d"],
        "descriptors3d_db": inp["descriptors3d_db"],
        "descriptors3d_coarse_db": inp["descriptors3d_coarse_db"],
    }
    data.update(extra or {})
    with torch.no_grad():
        model(data)
    model.backbone = real_backbone
    for h in hooks:
        h.remove()
    return data, caps


def pack_feature_case(inp, data, caps) -> dict:
    g = {}
    for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f"):
        g["in_" + k] = checksum(inp[k])
    g["planted_i"] = inp["planted_i"].numpy().astype(np.int32)
    g["planted_j"] = inp["planted_j"].numpy().astype(np.int32)
    g["pose_gt"] = inp["pose_gt"].numpy()
    g["K"] = inp["K"].numpy()
    # stage checksums
    pe_out = caps["pe"][0].flatten(2).transpose(1, 2)
    g["pe_out_cs"], g["pe_out_probe"] = checksum(pe_out), probe(pe_out)
    kp = caps["kpt"][0].transpose(1, 2)                                  # [B,N,C]
    g["kpt_out_cs"], g["kpt_out_probe"] = checksum(kp), probe(kp)
    # coarse layers: hook order per layer index = (2D stream, 3D stream)  (transformer.py:148-159)
    for li in range(len(caps["coarse"]) // 2):
        d2, d3 = caps["coarse"][2 * li], caps["coarse"][2 * li + 1]
        g[f"coarse{li}_2d_cs"], g[f"coarse{li}_2d_probe"] = checksum(d2), probe(d2)
        g[f"coarse{li}_3d_cs"], g[f"coarse{li}_3d_probe"] = checksum(d3), probe(d3)
    conf = data["conf_matrix"]
    g["conf_cs"] = checksum(conf)
    g["conf_rowmax"] = conf.max(dim=2)[0][0].numpy()
    g["conf_colmax"] = conf.max(dim=1)[0][0].numpy()
    for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
        g[k] = data[k].numpy().astype(np.int32)
    for k in ("mconf", "mkpts_3d_db", "mkpts_query_c", "expec_f", "mkpts_query_f"):
        g[k] = data[k].numpy().astype(np.float32)
    g["gt_mask"] = data["gt_mask"].numpy()
    if caps["fine_in"]:
        f3i, wini = caps["fine_in"][0]
        g["fine_in_f3_cs"], g["fine_in_win_cs"] = checksum(f3i), checksum(wini)
    for li in range(len(caps["fine"]) // 2):
        w2, f3 = caps["fine"][2 * li], caps["fine"][2 * li + 1]
        g[f"fine{li}_win_cs"], g[f"fine{li}_win_probe"] = checksum(w2), probe(w2)
        g[f"fine{li}_f3_cs"] = checksum(f3)
        g[f"fine{li}_f3_probe"] = f3[:4, 0, :8].numpy().copy()
    return g


def masked_case_inputs(sd, cfg):
    """Case E's inputs (shared with tests/test_gpu_parity.py through the fixture's checksums): the ragged B = 2 frame pair of case B
    with a padded-image mask per element (element 0: the right 4 coarse columns are padding, element 1: the bottom 3 coarse rows) and
    different (h, w) image scales, as ``OnePosePlus_dataset.py:283,329`` produces for ``img_pad`` / ``img_resize`` datasets."""
    from onepose_st_amd.synthetic import make_synthetic_inputs
    i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
    i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
    both = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
    both["image_hw"] = i0["image_hw"]
    both["planted_i"], both["planted_j"] = i0["planted_i"], i0["planted_j"]
    both["pose_gt"], both["K"] = i0["pose_gt"], i0["K"]
    hc, wc = 96 // 8, 136 // 8
    mask = torch.ones(2, hc, wc, dtype=torch.bool)
    mask[0, :, wc - 4:] = False
    mask[1, hc - 3:, :] = False
    scale = torch.tensor([[1.25, 1.5], [0.8, 1.0]], dtype=torch.float32)
    return both, {"query_image_mask": mask, "query_image_scale": scale}


def main():
    from onepose_st_amd.config import default_config
    from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict

    torch.set_num_threads(4)
    cfg = default_config()
    sd = make_synthetic_state_dict(seed=0, config=cfg)
    model = load_reference_model(cfg, sd)
    if "--only-masked" in sys.argv:
        make_masked_case(model, sd, cfg)
        return
    if "--only-hard" in sys.argv:
        make_hard_case(model, sd, cfg)
        return
    print("reference model loaded; state_dict tensors:", len(sd), "params:", sum(v.numel() for k, v in sd.items()
                                                                                  if "running" not in k and "num_batches" not in k))

    # ---- case A: c1-size planted frame at the feature boundary (B=1) -------------------------
    inp = make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg)
    data, caps = run_feature_case(model, inp)
    g = pack_feature_case(inp, data, caps)
    print("case c1: K =", len(g["i_ids"]), "planted =", len(g["planted_i"]),
          "planted-correct =", int(np.isin(g["i_ids"].astype(np.int64) * 100000 + g["j_ids"],
                                           g["planted_i"].astype(np.int64) * 100000 + g["planted_j"]).sum()))
    np.savez_compressed(os.path.join(HERE, "c1_feature_boundary.npz"), **g)

    # ---- case B: small ragged frame, B=2 (second element = another frame of the same object) ---
    i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
    i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
    both = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
    both["image_hw"] = i0["image_hw"]
    both["planted_i"], both["planted_j"] = i0["planted_i"], i0["planted_j"]
    both["pose_gt"], both["K"] = i0["pose_gt"], i0["K"]
    data, caps = run_feature_case(model, both)
    g = pack_feature_case(both, data, caps)
    print("case b2: K =", len(g["i_ids"]), "per batch:", np.bincount(g["b_ids"], minlength=2))
    np.savez_compressed(os.path.join(HERE, "b2_ragged_feature_boundary.npz"), **g)

    # ---- case C: full forward incl. backbone on a random image (K = 0 path) -------------------
    gi = torch.Generator().manual_seed(5)
    img = torch.rand(1, 1, 64, 96, generator=gi)
    obj = make_synthetic_inputs(sd, n_points=200, image_hw=(64, 96), n_plant=0, seed=4, config=cfg)
    data = {"query_image": img, "keypoints3d": obj["keypoints3d"], "descriptors3d_db": obj["descriptors3d_db"],
            "descriptors3d_coarse_db": obj["descriptors3d_coarse_db"]}
    feats = []
    h = model.backbone.register_forward_hook(lambda m, i, o: feats.extend(t.detach().clone() for t in o))
    with torch.no_grad():
        model(data)
    h.remove()
    g = {"image_cs": checksum(img), "feat_c_cs": checksum(feats[0]), "feat_f_cs": checksum(feats[1]),
         "feat_c_probe": feats[0][0, :4, 0, :8].numpy().copy(),
         "conf_cs": checksum(data["conf_matrix"]), "conf_rowmax": data["conf_matrix"].max(dim=2)[0][0].numpy(),
         "K": np.array([len(data["i_ids"])], dtype=np.int32)}
    for k in ("mconf", "mkpts_3d_db", "mkpts_query_c", "expec_f", "mkpts_query_f"):
        g[k + "_shape"] = np.array(data[k].shape, dtype=np.int32)
    print("case full-forward: K =", int(g["K"][0]), {k: tuple(data[k].shape) for k in ("expec_f", "mkpts_query_f")})
    np.savez_compressed(os.path.join(HERE, "full_forward_empty.npz"), **g)

    # ---- case D: full forward incl. backbone WITH matches: the reference's own backbone runs on a synthetic image and a forward
    #      hook adds the planted feature maps to its two outputs (a random image alone matches nothing) --------------------------------
    gi = torch.Generator().manual_seed(6)
    inp = make_synthetic_inputs(sd, n_points=500, image_hw=(128, 160), n_plant=180, seed=7, config=cfg)
    img = torch.rand(1, 1, 128, 160, generator=gi)
    data = {"query_image": img, "keypoints3d": inp["keypoints3d"], "descriptors3d_db": inp["descriptors3d_db"],
            "descriptors3d_coarse_db": inp["descriptors3d_coarse_db"]}
    feats = []

    def plant(m, i, o):
        feats.extend(t.detach().clone() for t in o)
        return [o[0] + inp["feat_c"], o[1] + inp["feat_f"]]
    h = model.backbone.register_forward_hook(plant)
    with torch.no_grad():
        model(data)
    h.remove()
    conf = data["conf_matrix"]
    g = {"image_cs": checksum(img), "feat_c_cs": checksum(feats[0]), "feat_f_cs": checksum(feats[1]),
         "delta_c_cs": checksum(inp["feat_c"]), "delta_f_cs": checksum(inp["feat_f"]),
         "conf_cs": checksum(conf), "conf_rowmax": conf.max(dim=2)[0][0].numpy(), "conf_colmax": conf.max(dim=1)[0][0].numpy(),
         "planted_i": inp["planted_i"].numpy().astype(np.int32), "planted_j": inp["planted_j"].numpy().astype(np.int32),
         "pose_gt": inp["pose_gt"].numpy(), "K": inp["K"].numpy()}
    for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
        g[k] = data[k].numpy().astype(np.int32)
    for k in ("mconf", "mkpts_3d_db", "mkpts_query_c", "expec_f", "mkpts_query_f"):
        g[k] = data[k].numpy().astype(np.float32)
    mc = g["mconf"]
    print("case full-forward planted: K =", len(g["i_ids"]), "planted =", len(g["planted_i"]),
          "min |mconf - thr| =", float(np.abs(mc - 0.1).min()) if len(mc) else None,
          "row maxima within 1e-3 of thr:", int((np.abs(g["conf_rowmax"] - 0.1) < 1e-3).sum()))
    np.savez_compressed(os.path.join(HERE, "full_forward_planted.npz"), **g)

    make_masked_case(model, sd, cfg)
    make_hard_case(model, sd, cfg)


def make_hard_case(model, sd, cfg):
    # ---- case F: c1-size LOW-MARGIN / OUTLIER frame (synthetic.HARD_PROFILE): descriptor noise U(0.1, 4.0) per planted pair, 35 % of the pairs
    #      planted away from their projection.  The reference's confidences spread over (0, 1) with dozens of row maxima around the 0.1
    #      threshold (coarse_matching.py:145-166 decides real frames there), and ~40 % of its matches are geometrically wrong -------------
    from onepose_st_amd.synthetic import HARD_PROFILE, make_synthetic_inputs
    inp = make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg, **HARD_PROFILE)
    data, caps = run_feature_case(model, inp)
    g = pack_feature_case(inp, data, caps)
    g["wrong_i"] = inp["wrong_i"].numpy().astype(np.int32)
    g["wrong_j"] = inp["wrong_j"].numpy().astype(np.int32)
    key = lambda i, j: i.astype(np.int64) * 100000 + j
    got = key(g["i_ids"], g["j_ids"])
    rm = g["conf_rowmax"]
    print("case c1_hard: K =", len(got), "correct =", int(np.isin(got, key(g["planted_i"], g["planted_j"])).sum()),
          "planted wrong and matched =", int(np.isin(got, key(g["wrong_i"], g["wrong_j"])).sum()),
          "row maxima in (0.05, 0.3):", int(((rm > 0.05) & (rm < 0.3)).sum()), "min |mconf - thr| =", float(np.abs(g["mconf"] - 0.1).min()),
          "row maxima within 1e-3 of thr:", int((np.abs(rm - 0.1) < 1e-3).sum()))
    np.savez_compressed(os.path.join(HERE, "c1_hard_feature_boundary.npz"), **g)


def make_masked_case(model, sd, cfg):
    # ---- case E: query_image_mask + query_image_scale (coarse_matching.py:108-114,224; fine_matching.py:104; transformer.py:148-159)
    both, extra = masked_case_inputs(sd, cfg)
    data, caps = run_feature_case(model, both, extra)
    g = pack_feature_case(both, data, caps)
    g["query_image_mask"] = extra["query_image_mask"].numpy()
    g["query_image_scale"] = extra["query_image_scale"].numpy()
    conf = data["conf_matrix"]
    g["conf_rowmax_b1"] = conf.max(dim=2)[0][1].numpy()
    dead = ~extra["query_image_mask"].flatten(1)
    print("case masked: K =", len(g["i_ids"]), "per batch:", np.bincount(g["b_ids"], minlength=2),
          "max conf in padded columns:", float(conf.transpose(1, 2)[dead].max()),
          "matches in padded cells:", int(dead[data["b_ids"], data["j_ids"]].sum()))
    np.savez_compressed(os.path.join(HERE, "b2_masked_scaled_feature_boundary.npz"), **g)


if __name__ == "__main__":
    main()
