"""Diagnostic: time the ResNet-FPN backbone on stock PyTorch-ROCm / MIOpen in several storage formats.

Not part of the product path; prints ms per 480x640 frame and the deviation of each variant from fp32 NCHW.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd.backbone import build_backbone  # noqa: E402
from onepose_st_amd.config import default_config  # noqa: E402
from onepose_st_amd.synthetic import make_synthetic_state_dict  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    cfg = default_config()
    sd = make_synthetic_state_dict(0, cfg)
    bb = build_backbone(cfg["loftr_backbone"])
    bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
    bb = bb.eval().to(dev)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(B, 1, 480, 640, generator=g).to(dev)
    with torch.no_grad():
        ref = bb(img)
        print(f"fp32 NCHW           {timed(lambda: bb(img)):8.2f} ms", flush=True)
        torch.backends.cudnn.benchmark = True
        print(f"fp32 NCHW benchmark {timed(lambda: bb(img)):8.2f} ms", flush=True)
        bcl = bb.to(memory_format=torch.channels_last)
        icl = img.contiguous(memory_format=torch.channels_last)
        out = bcl(icl)
        print(f"fp32 NHWC           {timed(lambda: bcl(icl)):8.2f} ms   max|d| {max((a - b).abs().max().item() for a, b in zip(out, ref)):.2e}", flush=True)
        for dt in (torch.bfloat16, torch.float16):
            import copy
            bh = copy.deepcopy(bb).to(dt).to(memory_format=torch.channels_last)
            ih = icl.to(dt)
            out = bh(ih)
            err = max(((a.float() - b).abs().max() / b.abs().max()).item() for a, b in zip(out, ref))
            print(f"{str(dt):19s} {timed(lambda: bh(ih)):8.2f} ms   rel max err {err:.2e}", flush=True)
            bn = copy.deepcopy(bb).to(dt)
            inn = img.to(dt)
            print(f"{str(dt):14s}NCHW {timed(lambda: bn(inn)):8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
