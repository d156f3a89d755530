"""Planted backbone-output features for the LoFTR tests: a random-weight backbone gives no matches (like i.i.d. inputs for the
2D-3D matcher), so the tests replace the backbone's outputs -- coarse rows after the positional encoding and the
fine maps -- by random features in which image 1 is image 0 moved by whole coarse cells."""
import torch


def planted_pair(hw, shift_cells=(2, 1), seed=3, noise=0.1):
    """-> (x0 [1, L, 256], g0 [1, 128, hf, wf], x1, g1); image-1 cell (y + dy, x + dx) carries image-0 cell (y, x); shift = (dx, dy)"""
    H, W = hw
    hc, wc = H // 8, W // 8
    dx, dy = shift_cells
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(1, hc * wc, 256, generator=g)
    x1 = torch.randn(1, hc * wc, 256, generator=g)
    for y in range(hc):
        for x in range(wc):
            y1, x1_ = y + dy, x + dx
            if 0 <= y1 < hc and 0 <= x1_ < wc:
                x1[0, y1 * wc + x1_] = x0[0, y * wc + x] + noise * torch.randn(256, generator=g)
    g0 = torch.randn(1, 128, H // 2, W // 2, generator=g)
    g1 = torch.roll(g0, shifts=(4 * dy, 4 * dx), dims=(2, 3)) + 0.5 * noise * torch.randn(1, 128, H // 2, W // 2, generator=g)
    return x0, g0, x1, g1


def oracle_hook(pair):
    return lambda f0, ff0, f1, ff1: (pair[0], pair[1], pair[2], pair[3])


def device_hook(pair, dev):
    """the product's hook sees channels-last fine maps [hf * wf, 128]"""
    x0, g0, x1, g1 = pair
    cl = lambda g: g[0].permute(1, 2, 0).reshape(-1, 128).contiguous().to(dev)
    return lambda fc0, ff0, fc1, ff1: (x0.to(dev), cl(g0), x1.to(dev), cl(g1))
