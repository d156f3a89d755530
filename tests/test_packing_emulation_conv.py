"""Backbone convolution (csrc/conv.hip) on the lane-level MFMA emulation, in exact arithmetic and without a GPU: BatchNorm
folding, the (chunk, tap, k-block) weight-fragment order of ``packing.pack_conv_bf16``, the swizzled patch image, the
per-tap activation fragment addresses (stride 1 and 2, 3x3 and 1x1) and the accumulator -> (channel, pixel) map."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from onepose_st_amd import packing
from tests import mfma_emul as E

CC, PIXB, TW = 32, 64, 32


def patch_off(p, c):
    return p * PIXB + ((c ^ ((p >> 2) & 3)) << 4)


def emulate_conv(x, w, ks, stride, TH, NT):
    """x [cin_p, H, W] float64 (padded channels zero), w [cout, cin, ks, ks] float64 -> [cout_p, Ho, Wo] as the kernel computes it"""
    cout, cin = w.shape[:2]
    cip, cop = packing.pad32(cin), packing.pad32(cout)
    H, W = x.shape[1:]
    pad, T = ks // 2, ks * ks
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    ncc, ctiles = cip // CC, cop // 32
    S = ncc * T * 2
    # the packed stream, with the packing's own permutation applied to the exact values
    wp = np.zeros((cop, cip, T))
    wp[:cout, :cin] = w.reshape(cout, cin, T)
    wm = torch.from_numpy(wp).view(cop, cip // 32, 2, 16, T).permute(0, 1, 4, 2, 3).reshape(cop, cip * T)
    frags = wm.view(cop // 32, 32, cip * T // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1, 8).numpy()      # pack_linear_frag16
    PW, PH = stride * (TW - 1) + ks, stride * (TH - 1) + ks
    out = np.full((cop, Ho, Wo), np.nan)
    WGT = 2 * NT
    for cg in range((ctiles + WGT - 1) // WGT):
        for by in range((Ho + TH - 1) // TH):
            for bx in range((Wo + TW - 1) // TW):
                x0, y0 = bx * TW, by * TH
                iy0, ix0 = y0 * stride - pad, x0 * stride - pad
                for wc in range(2):
                    ct0 = WGT * cg + NT * wc
                    acc = [[np.zeros((64, 16)) for _ in range(TH)] for _ in range(NT)]
                    s = 0
                    for cc in range(ncc):
                        img = np.full(PW * PH * PIXB // 2, np.nan)              # one plane, 2 bytes per element
                        for p in range(PW * PH):
                            py, px = divmod(p, PW)
                            iy, ix = iy0 + py, ix0 + px
                            for c in range(4):
                                v = x[cc * CC + 8 * c: cc * CC + 8 * c + 8, iy, ix] if (0 <= iy < H and 0 <= ix < W) else np.zeros(8)
                                o = patch_off(p, c) // 2
                                img[o:o + 8] = v
                        for tap in range(T):
                            dy, dx = divmod(tap, ks)
                            for kbl in range(2):
                                for tt in range(TH):
                                    xf = np.stack([img[patch_off((stride * tt + dy) * PW + stride * E.R[l] + dx, 2 * kbl + E.H[l]) // 2:][:8]
                                                   for l in range(64)])
                                    for t in range(NT):
                                        if ct0 + t < ctiles:
                                            wf = frags[((ct0 + t) * S + s) * 64:((ct0 + t) * S + s) * 64 + 64]
                                            acc[t][tt] = E.mfma_32x32x16(wf, xf, acc[t][tt])
                                s += 1
                    for t in range(NT):
                        if ct0 + t >= ctiles:
                            continue
                        for tt in range(TH):
                            for l in range(64):
                                for reg in range(16):
                                    ch, yy, xx = 32 * (ct0 + t) + E.ROWS[l, reg], y0 + tt, x0 + E.R[l]
                                    if yy < Ho and xx < Wo:
                                        out[ch, yy, xx] = acc[t][tt][l, reg]
    return out


@pytest.mark.parametrize("cin,cout,H,W,ks,stride,TH,NT", [
    (40, 48, 5, 35, 3, 1, 2, 2),      # two channel chunks, two x tiles (second ragged), ragged last row block
    (32, 96, 6, 9, 3, 2, 2, 1),       # stride 2, one tile per wave: 3 channel tiles over 2 workgroup groups
    (40, 40, 4, 33, 1, 1, 1, 2),      # 1x1
    (32, 32, 7, 8, 1, 2, 4, 1),       # 1x1 stride 2 (shortcut), 4-row tiles
])
def test_conv_index_arithmetic(cin, cout, H, W, ks, stride, TH, NT):
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(1, cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, ks, ks, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=stride, padding=ks // 2)[0].numpy()
    xp = np.zeros((packing.pad32(cin), H, W))
    xp[:cin] = x[0].numpy()
    got = emulate_conv(xp, w.numpy(), ks, stride, TH, NT)
    np.testing.assert_allclose(got[:cout], ref, atol=1e-9)
    assert np.all(got[cout:] == 0.0)                 # padded output channels come out exactly zero


def test_fold_bn_and_block_layout():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(48, 40, 3, 3, generator=g)
    sd = {"bn.weight": torch.rand(48, generator=g) + 0.5, "bn.bias": torch.randn(48, generator=g),
          "bn.running_mean": torch.randn(48, generator=g), "bn.running_var": torch.rand(48, generator=g) + 0.5}
    wf, bf = packing.fold_bn(w, sd, "bn.")
    x = torch.randn(2, 40, 6, 7, generator=g)
    ref = F.batch_norm(F.conv2d(x, w, padding=1), sd["bn.running_mean"], sd["bn.running_var"], sd["bn.weight"], sd["bn.bias"], False, 0.0, 1e-5)
    torch.testing.assert_close(F.conv2d(x, wf, bf, padding=1), ref, rtol=1e-5, atol=1e-4)      # f32 rounding of sums of ~360 O(1) terms
    blk = packing.pack_conv_bf16(wf, bf)
    cip, cop = 64, 64
    assert blk.dtype == torch.uint8 and blk.numel() == 2 * cop * cip * 9 * 2 + cop * 4
    bias = blk[2 * cop * cip * 9 * 2:].view(torch.float32)
    torch.testing.assert_close(bias[:48], bf)
    assert float(bias[48:].abs().max()) == 0.0
    hi = blk[:cop * cip * 9 * 2].view(torch.bfloat16).float()
    lo = blk[cop * cip * 9 * 2:2 * cop * cip * 9 * 2].view(torch.bfloat16).float()
    # hi + lo reproduces every weight to ~2^-17 relative; the multiset of magnitudes is that of the folded weights
    got = torch.sort((hi + lo).abs())[0][-48 * 40 * 9:]
    want = torch.sort(wf.abs().reshape(-1))[0]
    torch.testing.assert_close(got, want, rtol=2e-5, atol=1e-7)
    w0, b0 = packing.fold_bn(w, sd, None)
    assert torch.equal(w0, w) and float(b0.abs().max()) == 0.0
    stem = packing.pack_stem(torch.arange(128 * 49, dtype=torch.float32).view(128, 1, 7, 7), torch.arange(128, dtype=torch.float32))
    assert stem.numel() == 49 * 128 + 128 and float(stem[5 * 128 + 3]) == 3 * 49 + 5 and float(stem[49 * 128 + 7]) == 7.0
