"""Seeded synthetic weights and inputs for the 2D-3D matcher hot path.

There are no checkpoints or demo data in the container, so parity
and benchmarks run on build-owned, seeded, CPU-deterministic generators:

* :func:`make_synthetic_state_dict` -- a full ``state_dict`` with the reference's key
  layout (195 tensors), xavier-uniform matrices, mildly
  non-trivial LayerNorm affines so affine bugs are visible.
* :func:`make_synthetic_inputs` -- geometrically consistent 2D-3D data: 3D points on an
  object, a camera pose, their projections; a subset of points is *planted* into the
  coarse (1/8) and fine (1/2) query feature maps so that the matcher returns
  non-degenerate matches (i.i.d. random inputs give K=0) and host
  PnP can recover the pose.

Nothing here imports ``oracle/`` (the generator also feeds ``bench.py``); the little
math it needs (sinusoid table, keypoint MLP) is evaluated with the product's own host
helpers in :mod:`onepose_st_amd.host_math`.
"""
from __future__ import annotations

import math

import torch

from .backbone import build_backbone
from .config import default_config, encoder_layer_names
from . import host_math


def _xavier(gen, out_f, in_f):
    bound = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound


def _encoder_state(gen, prefix, d_model, n_layers, sd):
    for i in range(n_layers):
        p = f"{prefix}.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "merge"):
            sd[p + name + ".weight"] = _xavier(gen, d_model, d_model)
        sd[p + "mlp.0.weight"] = _xavier(gen, 2 * d_model, 2 * d_model)
        sd[p + "mlp.2.weight"] = _xavier(gen, d_model, 2 * d_model)
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = 1.0 + 0.05 * torch.randn(d_model, generator=gen)
            sd[p + nm + ".bias"] = 0.05 * torch.randn(d_model, generator=gen)


def make_synthetic_state_dict(seed: int = 0, config: dict | None = None, backbone: bool = True) -> dict:
    """Seeded weights with the reference key layout (``inference_OnePosePlus.py:30-40``
    loads such a dict with ``strict=True``)."""
    cfg = config or default_config()
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}
    if backbone:
        torch_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        bb = build_backbone(cfg["loftr_backbone"])
        torch.random.set_rng_state(torch_state)
        for k, v in bb.state_dict().items():
            v = v.clone()
            if k.endswith("running_mean"):
                v = 0.05 * torch.randn(v.shape, generator=gen)
            elif k.endswith("running_var"):
                v = 1.0 + 0.1 * torch.rand(v.shape, generator=gen)
            sd["backbone." + k] = v
    ke = cfg["keypoints_encoding"]
    chans = [3] + list(ke["keypoints_encoder"]) + [ke["descriptor_dim"]]
    for li, (cin, cout) in enumerate(zip(chans[:-1], chans[1:])):
        sd[f"kpt_3d_pos_encoding.encoder.{3 * li}.weight"] = _xavier(gen, cout, cin)
        last = li == len(chans) - 2
        b = torch.zeros(cout) if last else (torch.rand(cout, generator=gen) * 2 - 1) / math.sqrt(cin)
        sd[f"kpt_3d_pos_encoding.encoder.{3 * li}.bias"] = b
    _encoder_state(gen, "loftr_coarse", cfg["loftr_coarse"]["d_model"],
                   len(encoder_layer_names(cfg["loftr_coarse"])), sd)
    _encoder_state(gen, "loftr_fine", cfg["loftr_fine"]["d_model"],
                   len(encoder_layer_names(cfg["loftr_fine"])), sd)
    return {k: v.float().contiguous() for k, v in sd.items()}


def make_synthetic_loftr_state_dict(seed: int = 0, n_coarse: int = 8, n_fine: int = 2) -> dict:
    """Seeded weights with the key layout of ``LoFTR_for_OnePose_Plus`` (``loftr_for_sfm/loftr.py:16-31``: ``backbone.*``,
    ``loftr_coarse.layers.{0..7}.*``, ``loftr_fine.layers.{0,1}.*``; what ``build_2D_match_model`` loads strictly)."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}
    torch_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    bb = build_backbone({"type": "ResNetFPN", "resolution": [8, 2],
                         "resnetfpn": {"block_type": "BasicBlock", "initial_dim": 128, "block_dims": [128, 196, 256], "output_layers": [3, 1]}})
    torch.random.set_rng_state(torch_state)
    for k, v in bb.state_dict().items():
        v = v.clone()
        if k.endswith("running_mean"):
            v = 0.05 * torch.randn(v.shape, generator=gen)
        elif k.endswith("running_var"):
            v = 1.0 + 0.1 * torch.rand(v.shape, generator=gen)
        sd["backbone." + k] = v
    _encoder_state(gen, "loftr_coarse", 256, n_coarse, sd)
    _encoder_state(gen, "loftr_fine", 128, n_fine, sd)
    return {k: v.float().contiguous() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------

def _rotation(gen):
    """A moderate random rotation (axis-angle, <= ~35 degrees)."""
    axis = torch.randn(3, generator=gen, dtype=torch.float64)
    axis = axis / axis.norm()
    ang = float(torch.rand(1, generator=gen, dtype=torch.float64)) * 0.6
    kx = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]], dtype=torch.float64)
    return torch.eye(3, dtype=torch.float64) + math.sin(ang) * kx + (1 - math.cos(ang)) * (kx @ kx)


def make_synthetic_inputs(state_dict: dict, n_points: int = 1000, image_hw=(240, 320), n_plant: int = 600,
                          seed: int = 1, config: dict | None = None, noise: float = 0.1, frame: int = 0,
                          noise_hi: float | None = None, wrong_frac: float = 0.0) -> dict:
    """Feature-boundary inputs for one frame (B=1).

    Returns a dict with the model inputs ``keypoints3d [1,N,3]``, ``descriptors3d_db
    [1,128,N]``, ``descriptors3d_coarse_db [1,256,N]``, the backbone-output boundary
    tensors ``feat_c [1,256,H/8,W/8]`` / ``feat_f [1,128,H/2,W/2]`` (what the hot path
    consumes in place of ``query_image``), ``image_hw``, and ground truth
    (``K`` intrinsics, ``pose_gt [3,4]``, ``planted_i``/``planted_j`` index pairs).

    Planting happens *after* the additive encodings so that the matcher sees matching
    descriptors at the encoder input: ``q2d[j] = enc3d[i] + noise`` then
    ``feat_c = q2d - PE`` (``OnePosePlusModel.py:135-154``).  ``frame`` reseeds only the
    per-frame part (pose, cell permutation, 2D maps); the 3D object block depends on
    ``seed`` alone, as in a sequence of the demo (``inference.py:102-112``).

    Low-margin / outlier frames (``HARD_PROFILE``): ``noise_hi`` draws every planted pair's descriptor noise from
    U(noise, noise_hi) -- the reference's confidences then populate the whole (0, 1) range, with tens to hundreds of row maxima
    around the 0.1 threshold instead of all at ~1.0 -- and ``wrong_frac`` of the pairs are planted in a cell that is NOT the
    point's projection (a confident match that is geometrically wrong: a PnP outlier; returned as ``wrong_i`` / ``wrong_j``).
    The defaults consume exactly the random numbers they always did: every committed fixture stays valid.
    """
    cfg = config or default_config()
    H, W = image_hw
    assert H % 8 == 0 and W % 8 == 0
    hc, wc, hf, wf = H // 8, W // 8, H // 2, W // 2
    C, Cf = cfg["loftr_coarse"]["d_model"], cfg["loftr_fine"]["d_model"]
    g_obj = torch.Generator().manual_seed(seed)
    g_frm = torch.Generator().manual_seed(seed * 7919 + 104729 * (frame + 1))

    # object block (frame invariant)
    kpts = (torch.rand(n_points, 3, generator=g_obj) - 0.5) * torch.tensor([0.20, 0.14, 0.10])
    desc_c = torch.randn(1, C, n_points, generator=g_obj)
    desc_f = torch.randn(1, Cf, n_points, generator=g_obj)

    # camera: object ~0.45 m in front, focal chosen so the object fills most of the frame
    R = _rotation(g_frm)
    t = torch.tensor([0.0, 0.0, 0.45], dtype=torch.float64) + 0.01 * torch.randn(3, generator=g_frm, dtype=torch.float64)
    f = 1.9 * W
    K = torch.tensor([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], dtype=torch.float64)
    pc = kpts.double() @ R.T + t
    uv = (pc[:, :2] / pc[:, 2:3]) * f + torch.tensor([W / 2.0, H / 2.0], dtype=torch.float64)

    # coarse cell by rounding (the fine window is centred on pixel 8*j, fine_preprocess.py:40-46)
    jx = torch.round(uv[:, 0] / 8).long()
    jy = torch.round(uv[:, 1] / 8).long()
    inside = (jx >= 0) & (jx < wc) & (jy >= 0) & (jy < hc)
    order = torch.randperm(n_points, generator=g_frm)
    taken = torch.zeros(hc * wc, dtype=torch.bool)
    taken_f = set()        # fine pixels must be unique too (windows of neighbouring cells overlap)
    pi, pj, pdx, pdy = [], [], [], []
    for i in order.tolist():
        if len(pi) >= n_plant:
            break
        if not bool(inside[i]):
            continue
        j = int(jy[i]) * wc + int(jx[i])
        if taken[j]:
            continue
        fx, fy = int(torch.round(uv[i, 0] / 2)), int(torch.round(uv[i, 1] / 2))
        dx, dy = fx - 4 * int(jx[i]), fy - 4 * int(jy[i])
        if abs(dx) > 2 or abs(dy) > 2 or not (0 <= fx < wf and 0 <= fy < hf) or (fy, fx) in taken_f:
            continue
        taken[j] = True
        taken_f.add((fy, fx))
        pi.append(i), pj.append(j), pdx.append(dx), pdy.append(dy)
    n_wrong = int(wrong_frac * len(pi)) if wrong_frac > 0 else 0
    wi, wj = [], []
    if n_wrong:
        # the first n_wrong pairs (the list is in random order) move to free cells drawn at random: the descriptor match stays, the geometry goes
        free = torch.nonzero(~taken).flatten()
        free = free[torch.randperm(len(free), generator=g_frm)][:n_wrong].tolist()
        for k, jn in enumerate(free):
            taken[pj[k]] = False
            wi.append(pi[k]), wj.append(jn)
            pj[k], pdx[k], pdy[k] = jn, 0, 0
            taken[jn] = True
    pi_t, pj_t = torch.tensor(pi, dtype=torch.long), torch.tensor(pj, dtype=torch.long)
    sig = None
    if noise_hi is not None and pi:          # per-pair noise level
        sig = noise + (noise_hi - noise) * torch.rand(len(pi), generator=g_frm)

    # 3D descriptors after the keypoint encoding, 2D sequence after the sinusoid
    kn = host_math.normalize_keypoints3d(kpts[None])
    enc3d = desc_c + host_math.keypoint_mlp(state_dict, kn).transpose(1, 2)      # [1,C,N]
    q2d = torch.randn(hc * wc, C, generator=g_frm)
    q2d[pj_t] = enc3d[0, :, pi_t].T + (noise if sig is None else sig[:, None]) * torch.randn(len(pi), C, generator=g_frm)
    pe = host_math.sinusoid_table(C, hc, wc)                                     # [C,hc,wc]
    feat_c = (q2d.T.reshape(C, hc, wc) - pe)[None].contiguous()

    feat_f = torch.randn(1, Cf, hf, wf, generator=g_frm)
    if pi:
        fy = 4 * (pj_t // wc) + torch.tensor(pdy)
        fx = 4 * (pj_t % wc) + torch.tensor(pdx)
        feat_f[0, :, fy, fx] = desc_f[0, :, pi_t] + (noise if sig is None else sig[None, :]) * torch.randn(Cf, len(pi), generator=g_frm)

    return {
        "keypoints3d": kpts[None].contiguous(),
        "descriptors3d_db": desc_f.contiguous(),
        "descriptors3d_coarse_db": desc_c.contiguous(),
        "feat_c": feat_c,
        "feat_f": feat_f.contiguous(),
        "image_hw": (H, W),
        "K": K,
        "pose_gt": torch.cat([R, t[:, None]], dim=1),
        "planted_i": pi_t[len(wi):],          # geometrically correct pairs
        "planted_j": pj_t[len(wi):],
        "wrong_i": torch.tensor(wi, dtype=torch.long),
        "wrong_j": torch.tensor(wj, dtype=torch.long),
    }


CONFIG_SIZES = {
    # BASELINE.json configs: name -> (N, (H, W), planted)
    "c1": (1000, (240, 320), 600),
    "c2": (7000, (480, 640), 3000),
    "c4": (15000, (960, 1280), 6000),
    # low-margin / outlier variants of c1 and c2 (HARD_PROFILE below; not a BASELINE config: the parity and PnP stress case)
    "c1_hard": (1000, (240, 320), 600),
    "c2_hard": (7000, (480, 640), 3000),
}

# make_synthetic_inputs keyword arguments of the "*_hard" workloads: per-pair descriptor noise U(0.1, 4.0) (reference confidences spread
# over (0, 1), many around the 0.1 threshold of coarse_matching.py:145) and 35 % of the pairs planted away from their projection
# (confident but wrong: what ransac_PnP, metric_utils.py:155-165, has to reject)
HARD_PROFILE = dict(noise=0.1, noise_hi=4.0, wrong_frac=0.35)


def workload_kwargs(name: str) -> dict:
    return dict(HARD_PROFILE) if name.endswith("_hard") else {}
