"""GPU parity of the hand-written backbone convolutions through the C ABI.

Each convolution is compared with the same op on plain PyTorch fp32 (``F.conv2d`` / ``F.interpolate`` on the device); the
whole backbone is compared with the CPU oracle's ``backbone_8_2`` (reference ``backbone/resnet.py:85-164``) on the same
seeded weights and image.  Tolerances: split-bf16 (``bf16x3``) products carry ~2^-17 relative error per term and f32
accumulation -> 2e-5 of the output scale per layer, 1e-4 over the 22-convolution stack; plain bf16 only has to be close
(2e-2), it is not the parity mode.
"""
import pytest
import torch
import torch.nn.functional as F

from onepose_st_amd import hip, packing
from onepose_st_amd.backbone import build_backbone
from onepose_st_amd.backbone_hip import HipBackbone, pack_backbone
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs
from onepose_st_amd.synthetic import make_synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    hip.load()
    return torch.device("cuda:0")


def to_planes(x_nchw, cpad):
    """f32 NCHW -> (hi, lo) bf16 channels-last planes with zero channel padding"""
    B, C, H, W = x_nchw.shape
    x = torch.zeros(B, H, W, cpad, device=x_nchw.device)
    x[..., :C] = x_nchw.permute(0, 2, 3, 1)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def from_planes(hi, lo, C):
    return (hi.float() + lo.float())[..., :C].permute(0, 3, 1, 2)


def run_conv(dev, x, w, bias, stride, act, res=None, up=None, table=None, nsplit=3, want_f32=True):
    cout, cin, ks, _ = w.shape
    cip, cop = packing.pad32(cin), packing.pad32(cout)
    B, _, H, W = x.shape
    xh, xl = to_planes(x, cip)
    wp = packing.pack_conv_bf16(w.cpu(), bias.cpu()).to(dev)
    assert wp.numel() == hip.load().ophip_conv_wpack_bytes(cip, cop, ks)
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    oh = torch.full((B, Ho, Wo, cop), 7.0, dtype=torch.bfloat16, device=dev)
    ol = torch.full((B, Ho, Wo, cop), 7.0, dtype=torch.bfloat16, device=dev)
    o32 = torch.full((B, Ho, Wo, cout), 7.0, device=dev) if want_f32 else None
    rh = rl = None
    if res is not None:
        rh, rl = to_planes(res, cop)
    upp = None
    if up is not None:
        upp = torch.zeros(B, up.shape[2], up.shape[3], cop, device=dev)
        upp[..., :cout] = up.permute(0, 2, 3, 1)
    tab = None
    if table is not None:
        tab = torch.zeros(Ho, Wo, cop, device=dev)
        tab[..., :cout] = table.permute(1, 2, 0)
    P = hip.ptr
    hip.call("ophip_conv2d_bf16", P(xh, None), P(xl, None), B, H, W, cip, P(wp, None), cop, ks, stride, act,
             P(rh, None), P(rl, None), P(upp), upp.shape[1] if upp is not None else 0, upp.shape[2] if upp is not None else 0, P(tab),
             P(oh, None), P(ol, None), P(o32), cout if want_f32 else 0, nsplit, hip.stream_handle())
    torch.cuda.synchronize()
    got_planes = from_planes(oh, ol if nsplit == 3 else torch.zeros_like(ol), cout)
    if cop > cout:
        assert float(oh[..., cout:].float().abs().max()) == 0.0          # padding channels stay exactly zero
    return got_planes, (o32.permute(0, 3, 1, 2) if want_f32 else None)


def torch_conv(x, w, bias, stride, act, res=None, up=None, table=None):
    y = F.conv2d(x, w, bias, stride=stride, padding=w.shape[2] // 2)
    if res is not None:
        y = y + res
    if up is not None:
        y = y + F.interpolate(up, size=y.shape[2:], mode="bilinear", align_corners=True)
    if table is not None:
        y = y + table[None]
    return {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.01)}[act](y)


CASES = [
    # B, cin, cout, H, W, ks, stride, act, res, up, table
    (1, 128, 128, 24, 40, 3, 1, 1, False, False, False),
    (2, 128, 128, 13, 37, 3, 1, 1, True, False, False),      # ragged tile edges + residual
    (1, 128, 196, 24, 40, 3, 2, 1, False, False, False),     # stride 2, 196 -> padded 224 (7 channel tiles: a 1-tile wave)
    (1, 196, 196, 12, 20, 3, 1, 2, False, False, False),     # cin 196 padded, LeakyReLU
    (1, 128, 196, 24, 40, 1, 2, 0, False, False, False),     # 1x1 stride-2 shortcut
    (1, 196, 256, 12, 20, 1, 1, 0, False, True, False),      # 1x1 + bilinear x2 top-down add
    (1, 256, 256, 6, 10, 1, 1, 0, False, False, True),       # 1x1 + positional-encoding table
    (1, 256, 196, 12, 20, 3, 1, 0, False, False, False),
    (2, 196, 128, 9, 33, 3, 1, 0, False, False, False),
]


@pytest.mark.parametrize("nsplit,tol", [(3, 2e-5), (1, 2e-2)])
@pytest.mark.parametrize("case", CASES)
def test_conv_vs_torch(dev, case, nsplit, tol):
    B, cin, cout, H, W, ks, stride, act, use_res, use_up, use_tab = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).to(dev)
    bias = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(B, cout, Ho, Wo, generator=g).to(dev) if use_res else None
    up = torch.randn(B, cout, Ho // 2, Wo // 2, generator=g).to(dev) if use_up else None
    table = torch.randn(cout, Ho, Wo, generator=g).to(dev) if use_tab else None
    if nsplit == 1:      # plain-bf16 mode sees bf16-rounded inputs; compare like with like
        x = x.to(torch.bfloat16).float()
    got_p, got_f = run_conv(dev, x, w, bias, stride, act, res, up, table, nsplit)
    res_ref = res
    if res is not None and nsplit == 1:
        res_ref = res.to(torch.bfloat16).float()
    ref = torch_conv(x, w, bias, stride, act, res_ref, up, table)
    scale = float(ref.abs().max())
    assert float((got_f - ref).abs().max()) <= tol * scale
    # planes hold the same values rounded to hi + lo (16 mantissa bits) / hi only
    assert float((got_p - ref).abs().max()) <= (tol + (2e-5 if nsplit == 3 else 8e-3)) * scale


def test_stem_vs_torch(dev):
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 1, 40, 72, generator=g).to(dev)
    w = (torch.randn(128, 1, 7, 7, generator=g) / 7.0).to(dev)
    bias = (0.1 * torch.randn(128, generator=g)).to(dev)
    wp = packing.pack_stem(w.cpu(), bias.cpu()).to(dev)
    oh = torch.empty(2, 20, 36, 128, dtype=torch.bfloat16, device=dev)
    ol = torch.empty_like(oh)
    hip.call("ophip_stem_conv7", hip.ptr(img), 2, 40, 72, hip.ptr(wp), hip.ptr(oh, None), hip.ptr(ol, None), 3, hip.stream_handle())
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(img, w, bias, stride=2, padding=3))
    got = from_planes(oh, ol, 128)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.fixture(scope="module")
def backbone_setup(dev):
    cfg = default_config()
    sd = make_synthetic_state_dict(0, cfg)
    bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    g = torch.Generator().manual_seed(11)
    # non-trivial BatchNorm statistics so that the folding is exercised
    for k in list(bsd):
        if k.endswith("running_mean"):
            bsd[k] = 0.1 * torch.randn(bsd[k].shape, generator=g)
        elif k.endswith("running_var"):
            bsd[k] = 0.5 + torch.rand(bsd[k].shape, generator=g)
        elif k.endswith(".bias") and "bn" in k:
            bsd[k] = 0.1 * torch.randn(bsd[k].shape, generator=g)
    return cfg, bsd


@pytest.mark.parametrize("B,H,W", [(1, 240, 320), (2, 64, 104)])
def test_backbone_vs_oracle(dev, backbone_setup, B, H, W):
    from oracle import onepose_oracle as oracle
    cfg, bsd = backbone_setup
    g = torch.Generator().manual_seed(H)
    img = torch.rand(B, 1, H, W, generator=g)
    ref_c, ref_f = oracle.backbone_8_2({"backbone." + k: v for k, v in bsd.items()}, img)
    blocks = pack_backbone(bsd, dev)
    fc, ff = HipBackbone("bf16x3").forward(blocks, img.to(dev))
    torch.cuda.synchronize()
    got_c = fc.view(B, H // 8, W // 8, 256).permute(0, 3, 1, 2).cpu()
    got_f = ff.view(B, H // 2, W // 2, 128).permute(0, 3, 1, 2).cpu()
    for got, ref, name in ((got_c, ref_c, "coarse"), (got_f, ref_f, "fine")):
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err <= 1e-4, f"{name} map: relative max error {err:.2e}"


def test_backbone_matches_torch_module_and_pe_fusion(dev, backbone_setup):
    cfg, bsd = backbone_setup
    bb = build_backbone(cfg["loftr_backbone"])
    bb.load_state_dict(bsd)
    bb = bb.eval().to(dev)
    img = torch.rand(1, 1, 96, 128, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        ref_c, ref_f = bb(img)
    blocks = pack_backbone(bsd, dev)
    pe = torch.randn(12 * 16, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    fc, ff = HipBackbone("bf16x3").forward(blocks, img, pe_table=pe)
    torch.cuda.synchronize()
    want_c = ref_c.flatten(2).transpose(1, 2) + pe[None]
    assert float((fc - want_c).abs().max()) <= 1e-4 * float(want_c.abs().max())
    assert float((ff - ref_f.flatten(2).transpose(1, 2)).abs().max()) <= 1e-4 * float(ref_f.abs().max())


def test_model_forward_hip_backbone_vs_miopen_backbone(dev):
    """``model(data)`` with the reference's image input: the HIP backbone (default) against the same model with the
    backbone on PyTorch-ROCm / MIOpen fp32.  A random image gives no matches, so the comparison is on the confidence
    matrix (products of two softmaxes, values ~1e-3: relative 2e-3 covers the exp amplification of 1e-5 feature errors)."""
    cfg = default_config()
    sd = make_synthetic_state_dict(0, cfg)
    img = torch.rand(2, 1, 96, 128, generator=torch.Generator().manual_seed(9)).to(dev)
    obj = make_synthetic_inputs(sd, n_points=300, image_hw=(96, 128), n_plant=0, seed=4, config=cfg)
    outs = []
    for hip_bb in (True, False):
        c = dict(cfg)
        c["hip_backbone"] = hip_bb
        m = OnePosePlus_model(c).eval()
        m.load_state_dict(sd)
        m = m.to(dev)
        assert m.hip_backbone == hip_bb
        data = {"query_image": img, "keypoints3d": obj["keypoints3d"].to(dev).expand(2, -1, -1),
                "descriptors3d_db": obj["descriptors3d_db"].to(dev).expand(2, -1, -1),
                "descriptors3d_coarse_db": obj["descriptors3d_coarse_db"].to(dev).expand(2, -1, -1)}
        assert m(data) is None
        outs.append(data)
    a, b = outs
    assert tuple(a["q_hw_c"]) == (12, 16) and tuple(a["q_hw_f"]) == (48, 64) and a["bs"] == 2
    assert torch.equal(a["i_ids"], b["i_ids"]) and torch.equal(a["j_ids"], b["j_ids"])
    torch.testing.assert_close(a["conf_matrix"], b["conf_matrix"], rtol=2e-3, atol=1e-9)
