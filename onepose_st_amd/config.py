"""Model configuration for the OnePose++ 2D-3D matcher hot path.

``default_config()`` restates the ``model.OnePosePlus`` block of the reference's
``configs/experiment/inference_demo.yaml:11-95`` as a plain nested dict -- the exact
object ``OnePosePlus_model(config)`` receives in the reference
(``src/models/OnePosePlus/OnePosePlusModel.py:25``).  Only the keys the hot path reads
are validated by :func:`validate_config`; unsupported values raise the same exception
types the reference raises (``NotImplementedError`` / ``ValueError``).
"""
from __future__ import annotations

import copy

_DEFAULT = {
    "loftr_backbone": {
        "type": "ResNetFPN",
        "resolution": [8, 2],
        "resnetfpn": {
            "block_type": "BasicBlock",
            "initial_dim": 128,
            "block_dims": [128, 196, 256],
            "output_layers": [3, 1],
        },
        "pretrained": None,
        "pretrained_fix": False,
    },
    "interpol_type": "bilinear",
    "keypoints_encoding": {
        "enable": True,
        "type": "mlp_linear",
        "descriptor_dim": 256,
        "keypoints_encoder": [32, 64, 128],
        "norm_method": "instancenorm",
    },
    "positional_encoding": {"enable": True, "pos_emb_shape": [256, 256]},
    "loftr_coarse": {
        "type": "LoFTR",
        "d_model": 256,
        "d_ffm": 128,
        "nhead": 8,
        "layer_names": ["self", "cross"],
        "layer_iter_n": 3,
        "dropout": 0.0,
        "attention": "linear",
        "norm_method": "layernorm",
        "kernel_fn": "elu + 1",
        "d_kernel": 16,
        "redraw_interval": 2,
        "rezero": None,
        "final_proj": False,
    },
    "coarse_matching": {
        "type": "dual-softmax",
        "thr": 0.1,
        "feat_norm_method": "sqrt_feat_dim",
        "border_rm": 2,
        "dual_softmax": {"temperature": 0.08},
        "train": {
            "train_padding": True,
            "train_coarse_percent": 0.3,
            "train_pad_num_gt_min": 200,
        },
    },
    "loftr_fine": {
        "enable": True,
        "window_size": 5,
        "coarse_layer_norm": False,
        "type": "LoFTR",
        "d_model": 128,
        "nhead": 8,
        "layer_names": ["self", "cross"],
        "layer_iter_n": 1,
        "dropout": 0.0,
        "attention": "linear",
        "norm_method": "layernorm",
        "kernel_fn": "elu + 1",
        "d_kernel": 16,
        "redraw_interval": 2,
        "rezero": None,
        "final_proj": False,
    },
    "fine_matching": {"enable": True, "type": "s2d", "s2d": {"type": "heatmap"}},
}


def default_config() -> dict:
    """A fresh deep copy of the demo model config (``pretrained`` backbone disabled:
    the LoFTR checkpoint is not shipped)."""
    return copy.deepcopy(_DEFAULT)


def encoder_layer_names(enc_cfg: dict) -> list:
    """``list(layer_names) * layer_iter_n`` -- reference ``transformer.py:106``."""
    return list(enc_cfg["layer_names"]) * int(enc_cfg["layer_iter_n"])


def validate_config(cfg: dict) -> None:
    """Reject config values the HIP path does not implement, with the reference's
    exception types (``OnePosePlusModel.py:46-50``, ``coarse_matching.py:62-66``,
    ``transformer.py:55-56,120-121,187-201``, ``position_encoding.py:75-76``,
    ``fine_matching.py:71-75``, ``backbone/__init__.py:7-14``)."""
    bb = cfg["loftr_backbone"]
    if bb["type"] != "ResNetFPN":
        raise ValueError("loftr_backbone.type must be 'ResNetFPN'")
    if list(bb["resolution"]) != [8, 2]:
        raise NotImplementedError("only the 8->2 ResNetFPN is implemented")
    if bb["resnetfpn"]["block_type"] != "BasicBlock":
        raise NotImplementedError("only BasicBlock is implemented")
    ke = cfg["keypoints_encoding"]
    if ke["enable"]:
        if ke["type"] != "mlp_linear":
            raise NotImplementedError("keypoints_encoding.type")
        if ke["norm_method"] != "instancenorm":
            raise NotImplementedError("keypoints_encoding.norm_method (HIP path: instancenorm)")
    for name in ("loftr_coarse", "loftr_fine"):
        enc = cfg[name]
        if enc["type"] != "LoFTR":
            raise ValueError(f"{name}.type")
        if enc["attention"] != "linear":
            raise NotImplementedError(f"{name}.attention (HIP path: linear)")
        if enc["kernel_fn"] != "elu + 1":
            raise ValueError(f"{name}.kernel_fn")
        if enc["norm_method"] != "layernorm":
            raise NotImplementedError(f"{name}.norm_method (HIP path: layernorm)")
        if enc["rezero"] is not None:
            raise NotImplementedError(f"{name}.rezero")
        if enc["final_proj"]:
            raise NotImplementedError(f"{name}.final_proj")
        for ln in enc["layer_names"]:
            if ln not in ("self", "cross"):
                raise NotImplementedError(f"{name}.layer_names entry {ln!r}")
        if enc["redraw_interval"] is not None and enc["redraw_interval"] % 2 != 0:
            raise AssertionError("redraw_interval must be divisible by 2")
    cm = cfg["coarse_matching"]
    if cm["type"] != "dual-softmax":
        raise NotImplementedError("coarse_matching.type")
    if cm["feat_norm_method"] != "sqrt_feat_dim":
        raise NotImplementedError("coarse_matching.feat_norm_method (HIP path: sqrt_feat_dim)")
    if cfg["fine_matching"]["enable"] and cfg["fine_matching"]["s2d"]["type"] != "heatmap":
        raise NotImplementedError("fine_matching.s2d.type")
