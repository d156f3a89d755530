"""ResNet-FPN 8->2 backbone on the hand-written gfx950 convolution kernels (``csrc/conv.hip``).

SURVEY.md section 8(f)-1.  Same arithmetic as ``backbone.ResNetFPN_8_2`` (the reference's
``src/models/OnePosePlus/backbone/resnet.py:85-164`` in eval mode) with BatchNorm folded into
the convolution weights on the host; 23 launches per batch through the C ABI
(``ophip_stem_conv7``, ``ophip_conv2d_bf16``).  Intermediate maps are channels-last bf16 plane
pairs, the two outputs are float32 channels-last:

* ``feat_c  [B, H/8 * W/8, 256]``  (+ the positional-encoding table when one is passed: row a1 fused into the last epilogue)
* ``feat_f  [B, H/2 * W/2, 128]``  (the layout ``ophip_fine_refine*`` gathers from)

There is no fallback: without the HIP library or a HIP device this raises.
"""
from __future__ import annotations

import torch

from . import hip, packing

# (name, conv weight key, BatchNorm prefix or None)
_CONVS = [("stem", "conv1.weight", "bn1.")]
for _l, _n in (("layer1", 2), ("layer2", 2), ("layer3", 2)):
    for _i in range(_n):
        _CONVS += [(f"{_l}.{_i}.c1", f"{_l}.{_i}.conv1.weight", f"{_l}.{_i}.bn1."),
                   (f"{_l}.{_i}.c2", f"{_l}.{_i}.conv2.weight", f"{_l}.{_i}.bn2.")]
_CONVS += [("layer2.0.ds", "layer2.0.downsample.0.weight", "layer2.0.downsample.1."),
           ("layer3.0.ds", "layer3.0.downsample.0.weight", "layer3.0.downsample.1."),
           ("l3out", "layer3_outconv.weight", None), ("l2out", "layer2_outconv.weight", None),
           ("l2out2.0", "layer2_outconv2.0.weight", "layer2_outconv2.1."), ("l2out2.3", "layer2_outconv2.3.weight", None),
           ("l1out", "layer1_outconv.weight", None),
           ("l1out2.0", "layer1_outconv2.0.weight", "layer1_outconv2.1."), ("l1out2.3", "layer1_outconv2.3.weight", None)]

NONE, RELU, LEAKY = 0, 1, 2


def pack_backbone(sd: dict, device) -> dict:
    """``sd``: the backbone's own ``state_dict`` (keys without the ``backbone.`` prefix)."""
    blocks = {}
    for name, wkey, bn in _CONVS:
        w, b = packing.fold_bn(sd[wkey], sd, bn)
        if name == "stem":
            if tuple(w.shape) != (128, 1, 7, 7):
                raise NotImplementedError("HIP backbone is specialised for initial_dim 128, 7x7 stem on a 1-channel image")
            blocks[name] = packing.pack_stem(w, b).to(device)
        else:
            blocks[name] = (packing.pack_conv_bf16(w, b).to(device), w.shape[1], w.shape[0], w.shape[2])
    return blocks


class _Planes:
    """A channels-last feature map as (hi, lo) bf16 planes."""

    def __init__(self, B, H, W, c, device, split):
        self.H, self.W, self.c = H, W, c
        self.hi = torch.empty(B, H, W, c, dtype=torch.bfloat16, device=device)
        self.lo = torch.empty(B, H, W, c, dtype=torch.bfloat16, device=device) if split else None


class HipBackbone:
    def __init__(self, precision: str = "bf16x3"):
        if precision not in ("bf16x3", "bf16"):
            raise ValueError("the HIP backbone runs in 'bf16x3' (split-bf16) or 'bf16' arithmetic")
        self.nsplit = 3 if precision == "bf16x3" else 1

    def _conv(self, blocks, name, x, stride=1, act=NONE, res=None, up=None, table=None, planes=True, f32_channels=0):
        wpack, cin, cout, ks = blocks[name]
        cip, cop = packing.pad32(cin), packing.pad32(cout)
        if x.c != cip:
            raise ValueError(f"{name}: input has {x.c} padded channels, the weights expect {cip}")
        B, dev = x.hi.shape[0], x.hi.device
        Ho, Wo = (x.H + 2 * (ks // 2) - ks) // stride + 1, (x.W + 2 * (ks // 2) - ks) // stride + 1
        out = _Planes(B, Ho, Wo, cop, dev, self.nsplit == 3) if planes else None
        o32 = torch.empty(B, Ho, Wo, f32_channels, dtype=torch.float32, device=dev) if f32_channels else None
        P = hip.ptr
        hip.call("ophip_conv2d_bf16", P(x.hi, None), P(x.lo, None), B, x.H, x.W, cip, P(wpack, None), cop, ks, stride, act,
                 P(res.hi, None) if res is not None else None, P(res.lo, None) if res is not None else None,
                 P(up) if up is not None else None, up.shape[1] if up is not None else 0, up.shape[2] if up is not None else 0,
                 P(table) if table is not None else None,
                 P(out.hi, None) if out is not None else None, P(out.lo, None) if out is not None else None,
                 P(o32) if o32 is not None else None, f32_channels, self.nsplit, hip.stream_handle())
        return out if planes else o32

    @torch.no_grad()
    def forward(self, blocks: dict, image: torch.Tensor, pe_table: torch.Tensor | None = None):
        """``image [B, 1, H, W]`` float32 on the HIP device, H and W multiples of 8.  Returns ``(feat_c, feat_f)`` as
        described in the module docstring; ``pe_table [H/8 * W/8, 256]`` is added to ``feat_c`` when given."""
        if not image.is_cuda:
            raise hip.HipLibraryError("the HIP backbone needs device tensors (no CPU fallback)")
        hip.load()
        B, ci, H, W = image.shape
        if ci != 1 or H % 8 or W % 8:
            raise ValueError("query_image must be [B, 1, H, W] with H, W multiples of 8")
        img = image if (image.dtype == torch.float32 and image.is_contiguous()) else image.float().contiguous()
        dev = img.device
        split = self.nsplit == 3
        x0 = _Planes(B, H // 2, W // 2, 128, dev, split)
        hip.call("ophip_stem_conv7", hip.ptr(img), B, H, W, hip.ptr(blocks["stem"]), hip.ptr(x0.hi, None), hip.ptr(x0.lo, None),
                 self.nsplit, hip.stream_handle())
        x = x0
        for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
            for i in range(2):
                s = stride if i == 0 else 1
                t = self._conv(blocks, f"{layer}.{i}.c1", x, stride=s, act=RELU)
                short = x
                if s != 1:
                    short = self._conv(blocks, f"{layer}.{i}.ds", x, stride=s)
                x = self._conv(blocks, f"{layer}.{i}.c2", t, act=RELU, res=short)
            if layer == "layer1":
                x1 = x
            elif layer == "layer2":
                x2 = x
        x3 = x
        C3 = blocks["l3out"][2]
        x3o = self._conv(blocks, "l3out", x3, planes=False, f32_channels=C3)                 # pure map: the FPN upsamples it
        feat_c = x3o
        if pe_table is not None:
            feat_c = self._conv(blocks, "l3out", x3, table=pe_table, planes=False, f32_channels=C3)
        p2 = self._conv(blocks, "l2out", x2, up=x3o)
        q2 = self._conv(blocks, "l2out2.0", p2, act=LEAKY)
        x2o = self._conv(blocks, "l2out2.3", q2, planes=False, f32_channels=packing.pad32(blocks["l2out2.3"][2]))
        p1 = self._conv(blocks, "l1out", x1, up=x2o)
        q1 = self._conv(blocks, "l1out2.0", p1, act=LEAKY)
        feat_f = self._conv(blocks, "l1out2.3", q1, planes=False, f32_channels=blocks["l1out2.3"][2])
        hc, wc, hf, wf = H // 8, W // 8, H // 2, W // 2
        return feat_c.view(B, hc * wc, C3), feat_f.view(B, hf * wf, feat_f.shape[3])
