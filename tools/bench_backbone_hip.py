"""Diagnostic: per-layer device time of the HIP backbone (csrc/conv.hip) at 480x640, with achieved TFLOP/s per convolution
(algorithmic FLOPs; split-bf16 issues 3x that in MFMAs).  Not part of the product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import backbone_hip  # noqa: E402
from onepose_st_amd.backbone_hip import HipBackbone, pack_backbone  # noqa: E402
from onepose_st_amd.config import default_config  # noqa: E402
from onepose_st_amd.synthetic import make_synthetic_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    H, W = 480, 640
    dev = torch.device("cuda:0")
    cfg = default_config()
    sd = make_synthetic_state_dict(0, cfg)
    bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    blocks = pack_backbone(bsd, dev)
    img = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    bb = HipBackbone(prec)
    for _ in range(3):
        bb.forward(blocks, img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        bb.forward(blocks, img)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / n
    print(f"HIP backbone {prec} B={B}: {total:.3f} ms per batch ({total / B:.3f} ms per frame)")

    # per-layer: events around every conv call
    rec = []
    orig = HipBackbone._conv

    def timed(self, blocks_, name, x, stride=1, **kw):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = orig(self, blocks_, name, x, stride=stride, **kw)
        b.record()
        _, cin, cout, ks = blocks_[name]
        ho, wo = (x.H + 2 * (ks // 2) - ks) // stride + 1, (x.W + 2 * (ks // 2) - ks) // stride + 1
        rec.append((name, a, b, 2.0 * B * ho * wo * cin * cout * ks * ks, f"{cin}->{cout} k{ks} s{stride} @{ho}x{wo}"))
        return out

    HipBackbone._conv = timed
    acc = {}
    for _ in range(5):
        rec.clear()
        bb.forward(blocks, img)
        torch.cuda.synchronize()
        for i, (name, a, b, fl, desc) in enumerate(rec):
            acc.setdefault(i, [name, desc, fl, []])[3].append(a.elapsed_time(b))
    HipBackbone._conv = orig
    tot_fl = 0.0
    for i in sorted(acc):
        name, desc, fl, ts = acc[i]
        t = sorted(ts)[len(ts) // 2]
        tot_fl += fl
        print(f"{name:12s} {desc:34s} {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s")
    print(f"conv FLOPs {tot_fl / 1e9:.1f} G -> {tot_fl / total / 1e9:.1f} TFLOP/s over the whole backbone")


if __name__ == "__main__":
    main()
