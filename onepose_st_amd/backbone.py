"""ResNet-FPN 8->2 image backbone: the PARAMETER HOLDER under the reference's keys, and the PyTorch-ROCm (MIOpen) path of the
exact-f32 debug mode.  The default (bf16-pipe) modes run the backbone on this repo's own HIP convolution kernels
(``csrc/conv.hip`` through ``backbone_hip.py``, SURVEY section 8(f)-1 / DESIGN.md section 4b), which read the weights held here.

This module exists so that the drop-in class accepts the reference's full ``state_dict`` (``backbone.*`` keys, 107
tensors) and the reference's ``query_image`` input.  Architecture and parameter names
follow ``src/models/OnePosePlus/backbone/resnet.py:20-44,85-164`` (BasicBlock x2 per
stage, dims 128/196/256, FPN with bilinear ``align_corners=True`` upsampling); the
code is written functionally over small ``nn.Module`` holders so the key layout is the
only thing shared with the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride, 0, bias=False)


class _Residual(nn.Module):
    """Two 3x3 conv+BN with an identity / strided-1x1 shortcut (resnet.py:20-44)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.conv2 = _c3(cin, cout, stride), _c3(cout, cout)
        self.bn1, self.bn2 = nn.BatchNorm2d(cout), nn.BatchNorm2d(cout)
        self.downsample = None if stride == 1 else nn.Sequential(_c1(cin, cout, stride), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        s = x if self.downsample is None else self.downsample(x)
        return F.relu(s + y)


def _head(cin, cmid, cout):
    return nn.Sequential(_c3(cin, cmid), nn.BatchNorm2d(cmid), nn.LeakyReLU(), _c3(cmid, cout))


class ResNetFPN_8_2(nn.Module):
    """Returns ``[feat_1/8 (C=256), feat_1/2 (C=128)]`` for ``output_layers=[3, 1]``."""

    def __init__(self, config: dict):
        super().__init__()
        d0 = config["initial_dim"]
        d1, d2, d3 = config["block_dims"]
        self.output_layers = list(config["output_layers"])
        self.conv1 = nn.Conv2d(1, d0, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.layer1 = nn.Sequential(_Residual(d0, d1, 1), _Residual(d1, d1, 1))
        self.layer2 = nn.Sequential(_Residual(d1, d2, 2), _Residual(d2, d2, 1))
        self.layer3 = nn.Sequential(_Residual(d2, d3, 2), _Residual(d3, d3, 1))
        self.layer3_outconv = _c1(d3, d3)
        self.layer2_outconv = _c1(d2, d3)
        self.layer2_outconv2 = _head(d3, d3, d2)
        self.layer1_outconv = _c1(d1, d2)
        self.layer1_outconv2 = _head(d2, d2, d1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x0 = F.relu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)
        x3o = self.layer3_outconv(x3)
        x2o = self.layer2_outconv2(self.layer2_outconv(x2) + up(x3o))
        x1o = self.layer1_outconv2(self.layer1_outconv(x1) + up(x2o))
        feats = [x, x1o, x2o, x3o]
        return [feats[i] for i in self.output_layers]


def build_backbone(config: dict) -> nn.Module:
    """``backbone/__init__.py:7-14``."""
    if config["type"] != "ResNetFPN":
        raise ValueError("loftr_backbone.type")
    if list(config["resolution"]) != [8, 2]:
        raise NotImplementedError("loftr_backbone.resolution")
    return ResNetFPN_8_2(config["resnetfpn"])
