"""ORACLE -- test infrastructure, not product code.

CPU (torch, fp32) restatement of the reference's 2D-3D matching hot path
(``OnePosePlus_model.forward`` of mizeller/OnePose_ST, rows a1-a12 of SURVEY.md
section 8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / baseline; the
product path (``onepose_st_amd``) never routes through it and fails loudly if the HIP
library is missing.

Pinning: the reference ships no tests or golden vectors for this path (SURVEY section
4).  The oracle is therefore pinned against outputs of the reference itself, produced
in the build container by importing the reference's model files on CPU
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every stage).  Two arithmetic boundaries are outside ``/root/reference`` and are
restated from their published definitions -- parity unpinned there:
``kornia==0.4.1`` ``dsnt.spatial_expectation2d`` / ``create_meshgrid``
(``fine_matching.py:86-87``; :func:`spatial_expectation_5x5`), and pycolmap / OpenCV PnP
(``metric_utils.py:121-209``; not part of this file).

All functions are written functionally over a ``state_dict`` with the reference's key
layout; every function names the reference lines it follows (paths relative to
``/root/reference/src/models/OnePosePlus/``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# a1  positional encoding of the coarse feature map
# ----------------------------------------------------------------------------------------------

def position_table(d_model: int, max_shape=(256, 256)) -> torch.Tensor:
    """utils/position_encoding.py:13-35 -- ``pe [1, d_model, 256, 256]``.
    ``(-math.log(10000.0) / d_model // 2)`` floor-divides to -1.0 (quirk kept)."""
    pe = torch.zeros((d_model, *max_shape))
    y_position = torch.ones(max_shape).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones(max_shape).cumsum(1).float().unsqueeze(0)
    div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
    div_term = div_term[:, None, None]
    pe[0::4, :, :] = torch.sin(x_position * div_term)
    pe[1::4, :, :] = torch.cos(x_position * div_term)
    pe[2::4, :, :] = torch.sin(y_position * div_term)
    pe[3::4, :, :] = torch.cos(y_position * div_term)
    return pe.unsqueeze(0)


def pe_add_flatten(feat_c: torch.Tensor, pe: torch.Tensor) -> torch.Tensor:
    """position_encoding.py:37-42 + OnePosePlusModel.py:135-140:
    ``x + pe[:, :, :h, :w]`` then ``'n c h w -> n (h w) c'``."""
    x = feat_c + pe[:, :, : feat_c.size(2), : feat_c.size(3)]
    return x.flatten(2).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
# a2 / a3  3D keypoint normalisation and encoding
# ----------------------------------------------------------------------------------------------

def normalize_3d_keypoints(kpts: torch.Tensor) -> torch.Tensor:
    """utils/normalize.py:17-28 (extent of batch element 0 only, mean per element)."""
    width, height, length = kpts[0].max(dim=0).values - kpts[0].min(dim=0).values
    center = torch.mean(kpts, dim=-2)
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height, one * length])[None]
    scaling = size.max(1, keepdim=True).values * 0.6
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def keypoint_encode(sd: dict, kpts_n: torch.Tensor, descriptors: torch.Tensor) -> torch.Tensor:
    """utils/position_encoding.py:54-79: ``descriptors [B,C,N] + MLP(kpts)ᵀ``.
    ``nn.InstanceNorm1d(c)`` on a ``[B, N, c]`` tensor treats N as channels, i.e. it
    normalises each point over its c features (``F.instance_norm``, biased var,
    eps 1e-5, no affine, instance statistics also in eval)."""
    x = kpts_n
    layer_ids = sorted({int(k.split(".")[2]) for k in sd if k.startswith("kpt_3d_pos_encoding.encoder.")})
    for n, li in enumerate(layer_ids):
        p = f"kpt_3d_pos_encoding.encoder.{li}."
        x = F.linear(x, sd[p + "weight"], sd[p + "bias"])
        if n < len(layer_ids) - 1:
            x = F.relu(F.instance_norm(x, eps=1e-5))
    return descriptors + x.transpose(2, 1)


# ----------------------------------------------------------------------------------------------
# a5 / a6  encoder layer with linear attention
# ----------------------------------------------------------------------------------------------

def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, eps: float = 1e-6,
                     q_mask: torch.Tensor | None = None, kv_mask: torch.Tensor | None = None) -> torch.Tensor:
    """loftr_module/linear_attention.py:29-61.  q ``[B,L,H,D]``, k, v ``[B,S,H,D]``; ``q_mask [B,L]`` / ``kv_mask [B,S]``
    (``:49-53``) zero the padded rows of phi(Q) resp. phi(K) and V; ``v_length`` stays the padded length."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    v_length = v.size(1)
    v = v / v_length
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
    return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * v_length).contiguous()


def encoder_layer(sd: dict, p: str, x: torch.Tensor, source: torch.Tensor, nhead: int,
                  x_mask: torch.Tensor | None = None, source_mask: torch.Tensor | None = None) -> torch.Tensor:
    """loftr_module/transformer.py:65-94 (dropout 0, no rezero, LayerNorm eps 1e-5)."""
    bs, C = x.size(0), x.size(2)
    dim = C // nhead
    q = F.linear(x, sd[p + "q_proj.weight"]).view(bs, -1, nhead, dim)
    k = F.linear(source, sd[p + "k_proj.weight"]).view(bs, -1, nhead, dim)
    v = F.linear(source, sd[p + "v_proj.weight"]).view(bs, -1, nhead, dim)
    msg = linear_attention(q, k, v, q_mask=x_mask, kv_mask=source_mask)
    msg = F.linear(msg.view(bs, -1, C), sd[p + "merge.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], dim=2), sd[p + "mlp.0.weight"])), sd[p + "mlp.2.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    return x + msg


def feature_transformer(sd: dict, prefix: str, layer_names: list, nhead: int,
                        desc3d: torch.Tensor, desc2d: torch.Tensor, trace: list | None = None,
                        query_mask: torch.Tensor | None = None):
    """loftr_module/transformer.py:133-171: ``desc3d [B,C,L] -> [B,L,C]`` then the
    layers; both streams of a layer read the *pre-update* tensors.  ``trace`` (optional
    list) receives ``(desc3d, desc2d)`` after every layer.  ``query_mask [B, P]`` (``:148-159``): the 2D stream's rows are masked as
    queries AND as sources (self: both; cross: ``x_mask`` for the 2D update, ``source_mask`` for the 3D update)."""
    d3 = torch.einsum("bdn->bnd", desc3d)
    d2 = desc2d
    for i, name in enumerate(layer_names):
        p = f"{prefix}.layers.{i}."
        if name == "self":
            d2, d3 = encoder_layer(sd, p, d2, d2, nhead, query_mask, query_mask), encoder_layer(sd, p, d3, d3, nhead)
        elif name == "cross":
            d2, d3 = encoder_layer(sd, p, d2, d3, nhead, x_mask=query_mask), encoder_layer(sd, p, d3, d2, nhead, source_mask=query_mask)
        else:
            raise NotImplementedError
        if trace is not None:
            trace.append((d3, d2))
    return d3, d2


# ----------------------------------------------------------------------------------------------
# a7 / a8  coarse matching
# ----------------------------------------------------------------------------------------------

def dual_softmax_confidence(feat3d: torch.Tensor, feat2d: torch.Tensor, temperature: float,
                            mask_query: torch.Tensor | None = None) -> torch.Tensor:
    """utils/coarse_matching.py:101-115 (``sqrt_feat_dim`` normaliser); ``mask_query [B, S]`` (``:108-114``) adds -1e9 to the
    columns of padded query cells."""
    a = feat3d / feat3d.shape[-1] ** 0.5
    b = feat2d / feat2d.shape[-1] ** 0.5
    sim = torch.einsum("nlc,nsc->nls", a, b) / (temperature + 1e-4)
    if mask_query is not None:
        pad = torch.zeros_like(sim)
        pad[~mask_query.bool()[:, None, :].expand_as(sim)] = -1e9
        sim = sim + pad
    return F.softmax(sim, 1) * F.softmax(sim, 2)


def coarse_match_select(conf: torch.Tensor, hw_c, hw_i, keypoints3d: torch.Tensor, thr: float, border_rm: int,
                        query_image_scale: torch.Tensor | None = None) -> dict:
    """utils/coarse_matching.py:125-242, inference branch (``self.training`` False, no
    ``mask0``); ``query_image_scale [B, 2]`` (h, w factors, ``:224``) rescales the coarse keypoints per batch element.  ``mask_border`` (``:10-20``) slices
    ``-b:0`` for the bottom/right edges, which is empty: only the top ``b`` rows and the
    left ``b`` columns are removed."""
    B, N, M = conf.shape
    h, w = int(hw_c[0]), int(hw_c[1])
    mask = (conf > thr).view(B, N, h, w).clone()
    b = border_rm
    mask[:, :, :b] = False
    mask[:, :, :, :b] = False
    mask[:, :, -b:0] = False
    mask[:, :, :, -b:0] = False
    mask = mask.view(B, N, M)
    mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = hw_i[0] / hw_c[0]
    scale_total = scale * query_image_scale[b_ids][:, [1, 0]] if query_image_scale is not None else scale
    mk_q = torch.stack([j_ids % w, j_ids // w], dim=1) * scale_total
    mk_3d = keypoints3d[b_ids, i_ids]
    keep = mconf != 0
    return {
        "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
        "gt_mask": mconf == 0, "m_bids": b_ids[keep],
        "mkpts_3d_db": mk_3d[keep], "mkpts_query_c": mk_q[keep], "mconf": mconf[keep],
    }


# ----------------------------------------------------------------------------------------------
# a9  fine windows
# ----------------------------------------------------------------------------------------------

def fine_windows(feat_f: torch.Tensor, desc3d_fine: torch.Tensor, b_ids, i_ids, j_ids, hw_c, W: int = 5):
    """loftr_module/fine_preprocess.py:32-55: unfold W x W, stride ``h_f // h_c``,
    padding ``W//2``; gather the matched cells; 3D fine descriptors ``[B,C,N]`` gathered
    at ``(b, i)``.  Returns ``feat3d [K,C,1]``, ``windows [K,WW,C]``."""
    C = feat_f.size(1)
    if b_ids.shape[0] == 0:
        return torch.empty(0, C, 1), torch.empty(0, W * W, C)
    stride = feat_f.size(2) // int(hw_c[0])
    unf = F.unfold(feat_f, kernel_size=(W, W), stride=stride, padding=W // 2)     # [B, C*WW, M]
    unf = unf.view(feat_f.size(0), C, W * W, -1).permute(0, 3, 2, 1)             # n l ww c
    f3 = desc3d_fine.permute(0, 2, 1)[b_ids, i_ids, :].unsqueeze(-1)
    return f3, unf[b_ids, j_ids]


# ----------------------------------------------------------------------------------------------
# a11  fine matching
# ----------------------------------------------------------------------------------------------

def spatial_expectation_5x5(heatmap: torch.Tensor, W: int):
    """Restatement of kornia 0.4.1 ``create_meshgrid(W, W, normalized_coordinates=True)``
    and ``dsnt.spatial_expectation2d(heatmap[None], True)[0]`` (not vendored in the
    reference): grid ``xs = (linspace(0, W-1, W) / (W-1) - 0.5) * 2``,
    stacked as (x, y), expectation = sum(grid * p).  heatmap ``[K, W, W]``."""
    xs = (torch.linspace(0, W - 1, W) / (W - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(xs, xs, indexing="ij")
    grid = torch.stack([gx, gy], dim=-1).reshape(1, -1, 2)                       # [1, WW, 2] (x, y)
    p = heatmap.reshape(heatmap.size(0), -1, 1)
    coords = torch.sum(grid * p, dim=1)                                          # [K, 2]
    return coords, grid


def fine_match(feat3d: torch.Tensor, windows: torch.Tensor, mkpts_query_c: torch.Tensor, hw_i, hw_f,
               query_image_scale: torch.Tensor | None = None, b_ids: torch.Tensor | None = None) -> dict:
    """utils/fine_matching.py:28-110, ``heatmap`` type.  feat3d ``[K,L,C]`` (L odd, centre
    token picked), windows ``[K,WW,C]``; ``query_image_scale [B, 2]`` with the matches' ``b_ids`` (``:104``)."""
    K, WW, C = windows.shape
    W = int(math.sqrt(WW))
    scale = hw_i[0] / hw_f[0]
    if K == 0:
        return {"expec_f": torch.empty(0, 3), "mkpts_query_f": mkpts_query_c}
    L = feat3d.shape[1]
    picked = feat3d[:, L // 2, :]
    sim = torch.einsum("mc,mrc->mr", picked, windows)
    heatmap = torch.softmax((1.0 / C ** 0.5) * sim, dim=1).view(-1, W, W)
    coords, grid = spatial_expectation_5x5(heatmap, W)
    var = torch.sum(grid ** 2 * heatmap.view(-1, WW, 1), dim=1) - coords ** 2
    std = torch.sum(torch.sqrt(torch.clamp(var, min=1e-10)), -1)
    expec = torch.cat([coords, std.unsqueeze(1)], -1)
    if query_image_scale is not None:
        scale = scale * query_image_scale[b_ids][:, [1, 0]]
    mk_f = mkpts_query_c + (coords * (W // 2) * scale)[: len(mkpts_query_c)]
    return {"expec_f": expec, "mkpts_query_f": mk_f}


# ----------------------------------------------------------------------------------------------
# backbone (outside the north_star path; stock convolutions) -- resnet.py:20-44,141-164
# ----------------------------------------------------------------------------------------------

def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _block(sd, p, x, stride):
    y = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + "bn2.", F.conv2d(y, sd[p + "conv2.weight"], None, 1, 1))
    if stride != 1:
        x = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], None, stride, 0))
    return F.relu(x + y)


def backbone_8_2(sd: dict, image: torch.Tensor, prefix: str = "backbone."):
    """Eval-mode ResNetFPN_8_2 from the state dict; returns (feat 1/8, feat 1/2)."""
    g = lambda k: sd[prefix + k]
    s = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x0 = F.relu(_bn(s, "bn1.", F.conv2d(image, s["conv1.weight"], None, 2, 3)))
    x1 = _block(s, "layer1.1.", _block(s, "layer1.0.", x0, 1), 1)
    x2 = _block(s, "layer2.1.", _block(s, "layer2.0.", x1, 2), 1)
    x3 = _block(s, "layer3.1.", _block(s, "layer3.0.", x2, 2), 1)
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)

    def head(p, t):
        t = F.conv2d(t, s[p + "0.weight"], None, 1, 1)
        t = F.leaky_relu(_bn(s, p + "1.", t))
        return F.conv2d(t, s[p + "3.weight"], None, 1, 1)

    x3o = F.conv2d(x3, s["layer3_outconv.weight"])
    x2o = head("layer2_outconv2.", F.conv2d(x2, s["layer2_outconv.weight"]) + up(x3o))
    x1o = head("layer1_outconv2.", F.conv2d(x1, s["layer1_outconv.weight"]) + up(x2o))
    return x3o, x1o


# ----------------------------------------------------------------------------------------------
# a12  the whole forward
# ----------------------------------------------------------------------------------------------

def forward_from_features(sd: dict, cfg: dict, data: dict, feat_c: torch.Tensor, feat_f: torch.Tensor,
                          image_hw, trace: dict | None = None) -> dict:
    """OnePosePlusModel.py:133-203 from the backbone-output boundary on.  Returns a new
    dict with every key the reference writes into ``data``."""
    out = {"bs": feat_c.size(0), "q_hw_i": torch.Size(image_hw), "q_hw_c": feat_c.shape[2:], "q_hw_f": feat_f.shape[2:]}
    cc, cf = cfg["loftr_coarse"], cfg["loftr_fine"]
    names_c = list(cc["layer_names"]) * cc["layer_iter_n"]
    names_f = list(cf["layer_names"]) * cf["layer_iter_n"]

    if cfg["positional_encoding"]["enable"]:
        q2d = pe_add_flatten(feat_c, position_table(cc["d_model"], tuple(cfg["positional_encoding"]["pos_emb_shape"])))
    else:
        q2d = feat_c.flatten(2).transpose(1, 2).contiguous()
    desc_in = data["descriptors3d_coarse_db"] if "descriptors3d_coarse_db" in data else data["descriptors3d_db"]
    if cfg["keypoints_encoding"]["enable"]:
        d3 = keypoint_encode(sd, normalize_3d_keypoints(data["keypoints3d"]), desc_in)
    else:
        d3 = desc_in
    if trace is not None:
        trace["q2d_in"], trace["d3_in"], trace["coarse_layers"] = q2d, d3, []
    query_mask = data["query_image_mask"].flatten(-2) if "query_image_mask" in data else None      # OnePosePlusModel.py:156-158
    qscale = data.get("query_image_scale")
    d3, q2d = feature_transformer(sd, "loftr_coarse", names_c, cc["nhead"], d3, q2d,
                                  None if trace is None else trace["coarse_layers"], query_mask=query_mask)

    cm = cfg["coarse_matching"]
    conf = dual_softmax_confidence(d3, q2d, cm["dual_softmax"]["temperature"], mask_query=query_mask)
    out["conf_matrix"] = conf
    out.update(coarse_match_select(conf, out["q_hw_c"], out["q_hw_i"], data["keypoints3d"], cm["thr"], cm["border_rm"], qscale))

    if not cfg["fine_matching"]["enable"]:
        out["mkpts_query_f"] = out["mkpts_query_c"]
        return out

    out["W"] = cf["window_size"]
    f3, win = fine_windows(feat_f, data["descriptors3d_db"], out["b_ids"], out["i_ids"], out["j_ids"], out["q_hw_c"], cf["window_size"])
    if trace is not None:
        trace["fine_f3_in"], trace["fine_win_in"] = f3, win
    if win.size(0) != 0 and cf["enable"]:
        f3, win = feature_transformer(sd, "loftr_fine", names_f, cf["nhead"], f3, win)
    else:
        f3 = torch.einsum("bdn->bnd", f3)
    if trace is not None:
        trace["fine_f3_out"], trace["fine_win_out"] = f3, win
    out.update(fine_match(f3, win, out["mkpts_query_c"], out["q_hw_i"], out["q_hw_f"], qscale, out["b_ids"]))
    return out


def forward(sd: dict, cfg: dict, data: dict) -> dict:
    """OnePosePlusModel.py:95-203 including the backbone on ``data['query_image']``."""
    feat_c, feat_f = backbone_8_2(sd, data["query_image"])
    return forward_from_features(sd, cfg, data, feat_c, feat_f, tuple(data["query_image"].shape[2:]))
