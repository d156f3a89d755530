"""Small host-side (CPU, torch) helpers of the product.

* :func:`sinusoid_table` builds the fixed 2-D positional table the model uploads to
  HBM once (the reference keeps it as a non-persistent buffer,
  ``utils/position_encoding.py:20-35``).  It is evaluated with the same torch CPU ops
  as the reference so the table bits agree, including the reference's floor-division
  quirk in ``div_term``.
* :func:`normalize_keypoints3d` / :func:`keypoint_mlp` are used by the synthetic input
  generator only (to plant matches behind the additive encodings); the product's
  keypoint encoding itself is the HIP kernel ``ophip_kpt_encode``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sinusoid_table(d_model: int, h: int, w: int, max_shape=(256, 256)) -> torch.Tensor:
    """``pe[:, :h, :w]`` of ``PositionEncodingSine`` as ``[d_model, h, w]`` float32.

    ``-math.log(10000.0) / d_model // 2`` parses as ``(-ln(1e4)/d_model) // 2`` which is
    ``-1.0`` for every d_model > 4.6, so ``div_term = exp(-(0, 2, 4, ...))``
    (``position_encoding.py:25-28``); positions are 1-based cumsums (``:23-24``).
    """
    if h > max_shape[0] or w > max_shape[1]:
        raise ValueError(f"feature map {h}x{w} exceeds pos_emb_shape {tuple(max_shape)}")
    ones = torch.ones(h, w)
    y_pos = ones.cumsum(0).float().unsqueeze(0)
    x_pos = ones.cumsum(1).float().unsqueeze(0)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
    div = div[:, None, None]
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(x_pos * div)
    pe[1::4] = torch.cos(x_pos * div)
    pe[2::4] = torch.sin(y_pos * div)
    pe[3::4] = torch.cos(y_pos * div)
    return pe


def normalize_keypoints3d(kpts: torch.Tensor) -> torch.Tensor:
    """Centre per batch element, scale by 0.6 x the largest extent of batch element 0
    (``utils/normalize.py:17-28``)."""
    ext = kpts[0].max(dim=0).values - kpts[0].min(dim=0).values
    centre = kpts.mean(dim=-2, keepdim=True)
    return (kpts - centre) / (ext.max() * 0.6)


def keypoint_mlp(sd: dict, kn: torch.Tensor) -> torch.Tensor:
    """The 3->32->64->128->256 MLP of ``KeypointEncoding_linear`` on ``[B,N,3]`` ->
    ``[B,N,256]``.  ``nn.InstanceNorm1d`` applied to a ``[B,N,c]`` tensor normalises each
    point over its feature axis (biased variance, eps 1e-5, no affine) --
    ``position_encoding.py:62-79`` and SURVEY section 8a row a3."""
    x = kn
    idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith("kpt_3d_pos_encoding.encoder.")})
    for n, li in enumerate(idx):
        x = F.linear(x, sd[f"kpt_3d_pos_encoding.encoder.{li}.weight"], sd[f"kpt_3d_pos_encoding.encoder.{li}.bias"])
        if n < len(idx) - 1:
            mu = x.mean(dim=-1, keepdim=True)
            var = x.var(dim=-1, unbiased=False, keepdim=True)
            x = F.relu((x - mu) / torch.sqrt(var + 1e-5))
    return x
