"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path through the C ABI against the oracle
on the same seeded inputs, against the committed reference goldens, and size-independent properties at
BASELINE's full size.

Tolerances (north_star: bit-exact coarse match indices; pose / keypoints within 1e-4 relative).  Indices are compared
for exact equality in every mode.  Floats, by matrix arithmetic (model.precision):
  f32     exact-f32 MFMA (fmaf chain): differs from the CPU reference by summation order only: rtol 1e-4 / atol 2e-5
  bf16x3  split-bf16 (default): products carry ~2^-17 relative error: keypoints rtol 1e-4 / atol 5e-4 px, mconf rtol 5e-4
          (a confidence is exp(logit): its relative error is the absolute error of a logit of magnitude ~12);
          measured at c2: 3e-5 px and 4e-5
  bf16    plain bf16: only the match SET is compared (reported; planted matches must survive), floats at 5e-2
"""
import copy
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import onepose_oracle as orc
from onepose_st_amd import hip, host_math, packing
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.pnp import ransac_PnP
from onepose_st_amd.synthetic import make_synthetic_inputs

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    hip.load()
    return torch.device("cuda:0")


def _model(sd, cfg, dev, precision):
    c = copy.deepcopy(cfg)
    c["hip_precision"] = precision
    m = OnePosePlus_model(c).eval()
    m.load_state_dict(sd, strict=True)
    return m.to(dev)


@pytest.fixture(scope="module", params=["bf16x3", "f32"])
def model(request, sd, cfg, dev):
    return _model(sd, cfg, dev, request.param)


# float tolerances of the whole path by precision: (keypoints rtol, keypoints atol [px], mconf rtol)
TOL = {"f32": (RTOL, ATOL, 1e-4), "bf16x3": (1e-4, 5e-4, 5e-4)}


def close(a, b, rtol=RTOL, atol=ATOL, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=msg)


def to_dev(inp, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}


# ------------------------------------------------------------------------------------------------
# stage level
# ------------------------------------------------------------------------------------------------

def test_pe_add_transpose_exact(dev):
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 256, 7, 9, generator=g)
    ref = orc.pe_add_flatten(feat, orc.position_table(256))
    pe = host_math.sinusoid_table(256, 7, 9).flatten(1).t().contiguous().to(dev)
    out = torch.empty(2, 63, 256, device=dev)
    fd = feat.to(dev)
    hip.call("ophip_pe_add_transpose", hip.ptr(fd), hip.ptr(pe), hip.ptr(out), 2, 256, 63, hip.stream_handle())
    assert torch.equal(out.cpu(), ref)           # one f32 add per element on identical table bits


@pytest.mark.parametrize("B,N,shared", [(1, 77, False), (2, 1000, False), (3, 33, True)])
def test_kpt_encode(sd, dev, B, N, shared):
    g = torch.Generator().manual_seed(1)
    kp = torch.randn(1 if shared else B, N, 3, generator=g) * torch.tensor([0.3, 0.1, 0.2])
    desc = torch.randn(1 if shared else B, 256, N, generator=g)
    kp_full, desc_full = kp.expand(B, -1, -1), desc.expand(B, -1, -1)
    ref = orc.keypoint_encode(sd, orc.normalize_3d_keypoints(kp_full), desc_full).transpose(1, 2)
    w = packing.pack_keypoint_encoder(sd).to(dev)
    stats = torch.empty(4 * B + 4, device=dev)
    out = torch.empty(B, N, 256, device=dev)
    kd, dd = kp.to(dev), desc.to(dev)
    hip.call("ophip_kpt_encode", hip.ptr(kd), 0 if shared else kd.stride(0), hip.ptr(dd), 0 if shared else dd.stride(0),
             hip.ptr(w), hip.ptr(stats), hip.ptr(out), B, N, hip.stream_handle())
    close(out, ref, msg="kpt_encode")


@pytest.mark.parametrize("cross", [0, 1])
@pytest.mark.parametrize("B,L3,L2", [(1, 64, 32), (2, 70, 45), (1, 1000, 1200)])
def test_encoder_layer(sd, dev, cross, B, L3, L2):
    g = torch.Generator().manual_seed(2)
    x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
    p = "loftr_coarse.layers.2."
    if cross:
        r2, r3 = orc.encoder_layer(sd, p, x2, x3, 8), orc.encoder_layer(sd, p, x3, x2, 8)
    else:
        r2, r3 = orc.encoder_layer(sd, p, x2, x2, 8), orc.encoder_layer(sd, p, x3, x3, 8)
    w = packing.pack_coarse_layer(sd, p).to(dev)
    ws = torch.empty(hip.load().ophip_encoder_workspace_floats(B, L3, L2), device=dev)
    d3, d2 = x3.to(dev), x2.to(dev)
    y3, y2 = torch.full_like(d3, float("nan")), torch.full_like(d2, float("nan"))
    hip.call("ophip_encoder_layer", hip.ptr(d3), hip.ptr(d2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w), cross,
             hip.ptr(ws), hip.stream_handle())
    close(y3, r3, msg="3D stream")
    close(y2, r2, msg="2D stream")
    with pytest.raises(ValueError):           # in-place is refused (cross layers read pre-update streams)
        hip.call("ophip_encoder_layer", hip.ptr(d3), hip.ptr(d2), hip.ptr(d3), hip.ptr(y2), B, L3, L2, hip.ptr(w), cross,
                 hip.ptr(ws), hip.stream_handle())


@pytest.mark.parametrize("nsplit,rtol,atol", [(1, 5e-2, 5e-2)])
@pytest.mark.parametrize("cross", [0, 1])
@pytest.mark.parametrize("B,L3,L2", [(1, 64, 32), (2, 70, 45), (1, 1000, 1200)])
def test_encoder_layer_bf16(sd, dev, nsplit, rtol, atol, cross, B, L3, L2):
    """plain-bf16 mode's layer (csrc/encoder_bf16.hip): sanity-checked against the f32 oracle (8-bit mantissas: ~1e-2
    relative); the split-bf16 layer is ophip_encoder_layer_x3w8 (next test) and nsplit = 3 is refused here."""
    g = torch.Generator().manual_seed(2)
    x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
    p = "loftr_coarse.layers.2."
    if cross:
        r2, r3 = orc.encoder_layer(sd, p, x2, x3, 8), orc.encoder_layer(sd, p, x3, x2, 8)
    else:
        r2, r3 = orc.encoder_layer(sd, p, x2, x2, 8), orc.encoder_layer(sd, p, x3, x3, 8)
    w = packing.pack_coarse_layer_bf16(sd, p).to(dev)
    assert w.numel() == hip.load().ophip_encoder_bf16_wpack_bytes()
    ws = torch.empty(hip.load().ophip_encoder_bf16_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
    d3, d2 = x3.to(dev), x2.to(dev)
    y3, y2 = torch.full_like(d3, float("nan")), torch.full_like(d2, float("nan"))
    hip.call("ophip_encoder_layer_bf16", hip.ptr(d3), hip.ptr(d2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), None, nsplit, cross, 0, 0,
             hip.ptr(ws, None), hip.stream_handle())
    e3 = (y3.cpu() - r3).abs().max().item()
    e2 = (y2.cpu() - r2).abs().max().item()
    print(f"nsplit={nsplit} cross={cross} B={B} L=({L3},{L2}): max abs err 3D {e3:.3e} 2D {e2:.3e}")
    close(y3, r3, rtol=rtol, atol=atol, msg="3D stream")
    close(y2, r2, rtol=rtol, atol=atol, msg="2D stream")
    with pytest.raises(ValueError):
        hip.call("ophip_encoder_layer_bf16", hip.ptr(d3), hip.ptr(d2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), None, 3, cross, 0, 0,
                 hip.ptr(ws, None), hip.stream_handle())


X3_KERNELS = {"x3w8": ("ophip_encoder_layer_x3w8", "ophip_encoder_x3w8_wpack_bytes", "ophip_encoder_x3w8_workspace_bytes", "pack_coarse_layer_x3w8")}


@pytest.mark.parametrize("kern", ["x3w8"])
@pytest.mark.parametrize("cross", [0, 1])
@pytest.mark.parametrize("B,L3,L2", [(1, 64, 32), (2, 70, 45), (1, 1000, 1200), (3, 49, 97)])
def test_encoder_layer_x3(sd, dev, cross, B, L3, L2, kern):
    """split-bf16 layer on 16-token tiles / per-wave weight streams (csrc/encoder_x3w8.hip, the default): tracks the f32 oracle to ~1e-4
    at ragged sizes (tokens not a multiple of 48, several frames)."""
    g = torch.Generator().manual_seed(2)
    x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
    p = "loftr_coarse.layers.2."
    if cross:
        r2, r3 = orc.encoder_layer(sd, p, x2, x3, 8), orc.encoder_layer(sd, p, x3, x2, 8)
    else:
        r2, r3 = orc.encoder_layer(sd, p, x2, x2, 8), orc.encoder_layer(sd, p, x3, x3, 8)
    entry, wbytes, wsbytes, packer = X3_KERNELS[kern]
    w = getattr(packing, packer)(sd, p).to(dev)
    assert w.numel() == getattr(hip.load(), wbytes)()
    ws = torch.empty(getattr(hip.load(), wsbytes)(B, L3, L2), dtype=torch.uint8, device=dev)
    d3, d2 = x3.to(dev), x2.to(dev)
    y3, y2 = torch.full_like(d3, float("nan")), torch.full_like(d2, float("nan"))
    hip.call(entry, hip.ptr(d3), hip.ptr(d2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), None, cross, 0, 0,
             hip.ptr(ws, None), hip.stream_handle())
    e3 = (y3.cpu() - r3).abs().max().item()
    e2 = (y2.cpu() - r2).abs().max().item()
    print(f"{kern} cross={cross} B={B} L=({L3},{L2}): max abs err 3D {e3:.3e} 2D {e2:.3e}")
    close(y3, r3, rtol=3e-4, atol=1e-4, msg="3D stream")
    close(y2, r2, rtol=3e-4, atol=1e-4, msg="2D stream")
    with pytest.raises(ValueError):
        hip.call(entry, hip.ptr(d3), hip.ptr(d2), hip.ptr(d3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), None, cross, 0, 0,
                 hip.ptr(ws, None), hip.stream_handle())


@pytest.mark.parametrize("cross", [0, 1])
@pytest.mark.parametrize("B,L3,L2", [(1, 64, 32), (2, 70, 145)])
def test_encoder_layer_on_one_stream_equals_the_two_stream_rows(sd, dev, cross, B, L3, L2):
    """ophip_encoder_layer_x3w8_streams: the layer on ONE stream's rows (LoFTR's sequential cross layers: one launch per image; a cross layer
    then reduces the K / V of the OTHER stream itself) -- bit for bit the rows of the two-stream call, the other stream's output untouched."""
    g = torch.Generator().manual_seed(12)
    x3, x2 = torch.randn(B, L3, 256, generator=g).to(dev), torch.randn(B, L2, 256, generator=g).to(dev)
    w = packing.pack_coarse_layer_x3w8(sd, "loftr_coarse.layers.1.").to(dev)
    ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
    y3, y2 = torch.empty_like(x3), torch.empty_like(x2)
    hip.call("ophip_encoder_layer_x3w8", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), None, cross, 0, 0,
             hip.ptr(ws, None), hip.stream_handle())
    for streams, (ya, yb) in ((1, (y3, None)), (2, (None, y2))):
        o3, o2 = torch.full_like(x3, float("nan")), torch.full_like(x2, float("nan"))
        hip.call("ophip_encoder_layer_x3w8_streams", hip.ptr(x3), hip.ptr(x2), hip.ptr(o3) if streams & 1 else None, hip.ptr(o2) if streams & 2 else None,
                 B, L3, L2, hip.ptr(w, None), cross, streams, hip.ptr(ws, None), hip.stream_handle())
        torch.cuda.synchronize()
        if streams & 1:
            assert torch.equal(o3, y3) and bool(torch.isnan(o2).all())
        else:
            assert torch.equal(o2, y2) and bool(torch.isnan(o3).all())
    with pytest.raises(ValueError):
        hip.call("ophip_encoder_layer_x3w8_streams", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None), cross, 0,
                 hip.ptr(ws, None), hip.stream_handle())


@pytest.mark.parametrize("B,N,hc,wc", [(1, 100, 5, 15), (2, 333, 13, 20), (1, 1000, 30, 40)])
def test_encoder_layer_writes_the_similarity_fragments(sd, dev, B, N, hc, wc):
    """ophip_encoder_layer_x3w8_frag: the layer's output rows, also as the similarity kernel's (hi, lo) operand fragments inside the
    coarse workspace (rows padded with zeros to 128).  Coarse matching fed that way (nsplit | OPHIP_COARSE_PLANES_READY) must give
    bit for bit what it gives when its own frag_planes kernel converts the same rows -- at sizes that are no multiple of 48 or 128."""
    import ctypes
    M = hc * wc
    g = torch.Generator().manual_seed(9)
    x3, x2 = torch.randn(B, N, 256, generator=g).to(dev), torch.randn(B, M, 256, generator=g).to(dev)
    kp = torch.randn(B, N, 3, generator=g).to(dev)
    p = "loftr_coarse.layers.5."
    w = packing.pack_coarse_layer_x3w8(sd, p).to(dev)
    ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, N, M), dtype=torch.uint8, device=dev)
    cap = B * N

    def coarse(flag, cws, y3, y2):
        conf = torch.empty(B, N, M, device=dev)
        ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(3)]
        mconf, mk3, mkc = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        hip.call("ophip_coarse_match", hip.ptr(y3), hip.ptr(y2), hip.ptr(kp), kp.stride(0), B, N, M, wc, 0.08, 0.1, 2, 8.0, hip.ptr(conf), hip.ptr(cws),
                 *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(cnt, torch.int32),
                 3 | flag, hip.stream_handle())
        K = int(cnt.item())
        return conf, [t[:K].clone() for t in ids], mconf[:K].clone()

    y3a, y2a = torch.empty_like(x3), torch.empty_like(x2)
    hip.call("ophip_encoder_layer_x3w8", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3a), hip.ptr(y2a), B, N, M, hip.ptr(w, None), None, 1, 0, 0,
             hip.ptr(ws, None), hip.stream_handle())
    cws_a = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev)
    ref = coarse(0, cws_a, y3a, y2a)

    y3b, y2b = torch.empty_like(x3), torch.empty_like(x2)
    cws_b = torch.full((hip.load().ophip_coarse_workspace_floats(B, N, M),), float("nan"), device=dev)      # padding must be written, not assumed
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    hip.call("ophip_coarse_frag_planes", hip.ptr(cws_b), B, N, M, ctypes.byref(pa), ctypes.byref(pb))
    hip.call("ophip_encoder_layer_x3w8_frag", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3b), hip.ptr(y2b), B, N, M, hip.ptr(w, None), None, 1, 0, 0,
             hip.ptr(ws, None), pa, pb, hip.stream_handle())
    got = coarse(0x100, cws_b, y3b, y2b)
    assert torch.equal(y3a, y3b) and torch.equal(y2a, y2b)
    assert torch.equal(ref[0], got[0]) and all(torch.equal(a, b) for a, b in zip(ref[1], got[1])) and torch.equal(ref[2], got[2])


@pytest.mark.parametrize("kern", ["x3w8"])
@pytest.mark.parametrize("B,L3,L2", [(1, 100, 75), (2, 333, 260)])
def test_encoder_x3_chain_with_fused_kv_tail(sd, dev, B, L3, L2, kern):
    """three chained layers (self, cross, self): layers 1.. take their K|V slabs from the previous launch's fused tail
    (kv_from_prev = 1, slab sets ping-pong); result vs the oracle's layers and vs the same chain run stand-alone."""
    g = torch.Generator().manual_seed(5)
    x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
    names = ["self", "cross", "self"]
    r3, r2 = x3, x2
    for li, nm in enumerate(names):
        p = f"loftr_coarse.layers.{li}."
        if nm == "cross":
            r2, r3 = orc.encoder_layer(sd, p, r2, r3, 8), orc.encoder_layer(sd, p, r3, r2, 8)
        else:
            r2, r3 = orc.encoder_layer(sd, p, r2, r2, 8), orc.encoder_layer(sd, p, r3, r3, 8)
    entry, _, wsbytes, packer = X3_KERNELS[kern]
    ws_ = [getattr(packing, packer)(sd, f"loftr_coarse.layers.{li}.").to(dev) for li in range(3)]
    ws = torch.empty(getattr(hip.load(), wsbytes)(B, L3, L2), dtype=torch.uint8, device=dev)

    def chain(fused):
        a3, a2 = x3.to(dev), x2.to(dev)
        b3, b2 = torch.full_like(a3, float("nan")), torch.full_like(a2, float("nan"))
        for li, nm in enumerate(names):
            nxt = ws_[li + 1] if (fused and li + 1 < 3) else None
            hip.call(entry, hip.ptr(a3), hip.ptr(a2), hip.ptr(b3), hip.ptr(b2), B, L3, L2, hip.ptr(ws_[li], None),
                     hip.ptr(nxt, None), 1 if nm == "cross" else 0, 1 if (fused and li > 0) else 0, (li & 1) if fused else 0,
                     hip.ptr(ws, None), hip.stream_handle())
            a3, b3, a2, b2 = b3, a3, b2, a2
        return a3.clone(), a2.clone()
    f3, f2 = chain(True)
    s3, s2 = chain(False)
    assert torch.equal(f3, s3) and torch.equal(f2, s2)          # the fused tail computes exactly what the stand-alone K|V launch does
    close(f3, r3, rtol=5e-4, atol=2e-4, msg="3D stream after 3 layers")
    close(f2, r2, rtol=5e-4, atol=2e-4, msg="2D stream after 3 layers")


def _coarse_match(dev, f3, f2, kp, wc, thr=0.1, border=2, temp=0.08, scale=8.0, nsplit=0):
    B, N, _ = f3.shape
    M = f2.shape[1]
    cap = B * N
    # conf_matrix AND the workspace (tile statistics, fragment planes, row-best records) start as NaN: a tile the similarity kernel
    # skipped or wrote from a ring buffer it read too early (csrc/coarse_match.hip, the wait + barrier of its k-loop) stays visible
    conf = torch.full((B, N, M), float("nan"), device=dev)
    ws = torch.full((hip.load().ophip_coarse_workspace_floats(B, N, M),), float("nan"), device=dev)
    ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(3)]
    mconf, mk3, mkc = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    m_bids, gt_mask = torch.full((cap,), -1, dtype=torch.int64, device=dev), torch.full((cap,), 7, dtype=torch.uint8, device=dev)
    kd, f3d, f2d = kp.to(dev), f3.to(dev), f2.to(dev)      # keep the device copies alive across the call
    hip.call("ophip_coarse_match", hip.ptr(f3d), hip.ptr(f2d), hip.ptr(kd), kd.stride(0), B, N, M, wc,
             temp, thr, border, scale, hip.ptr(conf), hip.ptr(ws), *[hip.ptr(t, torch.int64) for t in ids],
             hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), hip.ptr(m_bids, torch.int64), hip.ptr(gt_mask, torch.uint8),
             hip.ptr(cnt, torch.int32), nsplit, hip.stream_handle())
    K = int(cnt.item())
    assert bool(torch.isfinite(conf).all()), "conf_matrix has elements no kernel wrote"
    assert torch.equal(m_bids[:K], ids[0][:K]) and bool((m_bids[K:] == -1).all())      # second copy of b_ids, nothing past K
    assert torch.equal(gt_mask[:K].bool(), mconf[:K] == 0) and bool((gt_mask[K:] == 7).all())
    return conf, [t[:K] for t in ids], mconf[:K], mk3[:K], mkc[:K]


def _planted_features(B, N, hc, wc, seed, n_plant):
    g = torch.Generator().manual_seed(seed)
    M = hc * wc
    f3, f2 = torch.randn(B, N, 256, generator=g), torch.randn(B, M, 256, generator=g)
    for b in range(B):
        cells = torch.randperm(M, generator=g)[:n_plant]
        f2[b, cells] = f3[b, :n_plant] * 1.5 + 0.1 * torch.randn(n_plant, 256, generator=g)
    return f3 * 1.5, f2


@pytest.mark.parametrize("nsplit,rtol", [(0, 1e-4), (3, 1e-3), (1, 0.5)])
@pytest.mark.parametrize("B,N,hc,wc", [(1, 300, 10, 13), (2, 129, 9, 9), (1, 1000, 30, 40)])
def test_coarse_match_vs_oracle(dev, B, N, hc, wc, nsplit, rtol):
    """similarity GEMM in exact f32 (0), split-bf16 (3) or bf16 (1).  conf = exp(.) products, so its relative error
    is the absolute error of the logits (|sim| up to ~40 here): 1e-4 / 1e-3 / O(0.1)."""
    f3, f2 = _planted_features(B, N, hc, wc, 3, min(N, hc * wc) // 2)
    kp = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(4))
    ref_conf = orc.dual_softmax_confidence(f3, f2, 0.08)
    ref = orc.coarse_match_select(ref_conf, (hc, wc), (hc * 8, wc * 8), kp, 0.1, 2)
    conf, (b_ids, i_ids, j_ids), mconf, mk3, mkc = _coarse_match(dev, f3, f2, kp, wc, nsplit=nsplit)
    close(conf, ref_conf, rtol=rtol, atol=1e-7, msg="conf_matrix")
    assert len(ref["i_ids"]) > 10
    for got, want in ((b_ids, ref["b_ids"]), (i_ids, ref["i_ids"]), (j_ids, ref["j_ids"])):
        assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    close(mconf, ref["mconf"], rtol=rtol, atol=1e-6)
    assert torch.equal(mk3.cpu(), ref["mkpts_3d_db"]) and torch.equal(mkc.cpu(), ref["mkpts_query_c"])


@pytest.mark.parametrize("nsplit", [3, 1])
@pytest.mark.parametrize("B,N,hc,wc", [(1, 300, 10, 13), (2, 129, 9, 9), (1, 1000, 30, 40), (1, 1408, 24, 32)])
def test_coarse_match_tile_kernels_and_two_pass_form_agree(dev, monkeypatch, B, N, hc, wc, nsplit):
    """The bf16 modes' similarity stage exists in two tile kernels (sim_frag_kernel, the default; sim_frag3_kernel, OPHIP_SIM_TILE=3: three
    workgroups per CU, faster per tile and slower in the pipeline, kept as the measured alternative) and two eager forms (one tile pass that stores S + the in-place conversion pass;
    OPHIP_COARSE_TWO_PASS=1: statistics pass + a second tile pass that writes every confidence once -- the default from 2^27 matrix elements
    on, BASELINE config 4).  Match lists must be identical in all four; conf_matrix bit-identical between the two forms on the same tile
    kernel (same statistics, conf_kernel's own expression) and between the tile kernels on interior tiles (same association of every sum);
    tiles cut by the matrix edge take each kernel's exact pass, whose column partials are merged in a different order (last bits)."""
    f3, f2 = _planted_features(B, N, hc, wc, 3, min(N, hc * wc) // 2)
    kp = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(4))
    out = {}
    for tile in ("3", "2"):
        for two in ("0", "1"):
            monkeypatch.setenv("OPHIP_SIM_TILE", tile)
            monkeypatch.setenv("OPHIP_COARSE_TWO_PASS", two)
            conf, ids, mconf, mk3, mkc = _coarse_match(dev, f3, f2, kp, wc, nsplit=nsplit)
            out[tile, two] = (conf.clone(), [t.clone() for t in ids], mconf.clone(), mkc.clone())
    base = out["3", "0"]
    assert len(base[1][1]) > 10
    for key, (conf, ids, mconf, mkc) in out.items():
        for a, b in zip(ids, base[1]):
            assert torch.equal(a, b), key
        assert torch.equal(mkc, base[3]), key
    for tile in ("3", "2"):
        assert torch.equal(out[tile, "0"][0], out[tile, "1"][0]), tile          # the two forms on one tile kernel: every confidence bit for bit
        assert torch.equal(out[tile, "0"][2], out[tile, "1"][2]), tile
    close(out["2", "0"][0], base[0], rtol=1e-5, atol=0, msg="conf_matrix of the two tile kernels")
    M = hc * wc
    ni, nj = (N // 128) * 128, (M // 128) * 128                                  # the interior tiles: statistics in the same association
    if ni and nj and N % 128 == 0 and M % 128 == 0:
        assert torch.equal(out["2", "0"][0], base[0])


@pytest.mark.parametrize("nsplit,rtol", [(0, 1e-4), (3, 1e-3)])
def test_coarse_match_wide_logit_range(dev, nsplit, rtol):
    """A few very strong pairs (logit ~110) among ordinary ones (|logit| < 10) in interior 128 x 128 tiles: the bf16 modes' tile
    kernel takes the tile maximum as the reference of its row / column sums, so the other rows' partials would underflow -- those
    tiles must fall back to the exact sweep, and the merged statistics (references far above a row's own maximum) must not
    overflow in the conf pass (log form).  N = M = 512: 4 x 4 tiles, none cut by the matrix edge."""
    hc, wc, N = 16, 32, 512
    g = torch.Generator().manual_seed(11)
    f3 = torch.randn(1, N, 256, generator=g) * 1.5
    f2 = torch.randn(1, hc * wc, 256, generator=g)
    rows = torch.randperm(N, generator=g)[:24]
    cells = torch.randperm(hc * wc, generator=g)[:24]
    f2[0, cells] = f3[0, rows] * 4.0
    kp = torch.randn(1, N, 3, generator=g)
    ref_conf = orc.dual_softmax_confidence(f3, f2, 0.08)
    ref = orc.coarse_match_select(ref_conf, (hc, wc), (hc * 8, wc * 8), kp, 0.1, 2)
    conf, (b_ids, i_ids, j_ids), mconf, mk3, mkc = _coarse_match(dev, f3, f2, kp, wc, nsplit=nsplit)
    assert bool(torch.isfinite(conf).all())
    close(conf, ref_conf, rtol=rtol, atol=1e-7, msg="conf_matrix")
    assert len(ref["i_ids"]) >= 10
    assert torch.equal(i_ids.cpu(), ref["i_ids"]) and torch.equal(j_ids.cpu(), ref["j_ids"])
    close(mconf, ref["mconf"], rtol=rtol, atol=1e-6)


def test_coarse_match_tie_and_border_semantics(dev):
    """Exact ties: two identical 2D cells, the first of them in the removed border.  The reference mask
    (coarse_matching.py:145-166) is false at the border copy and true at the interior copy, so `mask.max(dim=2)`
    returns the interior j; a plain arg-max-then-test would drop the match.  Also: bottom/right borders stay."""
    hc, wc, N = 8, 9, 40
    g = torch.Generator().manual_seed(5)
    f3 = torch.randn(1, N, 256, generator=g) * 1.5
    f2 = torch.randn(1, hc * wc, 256, generator=g)
    cell = lambda y, x: y * wc + x
    f2[0, cell(0, 4)] = f3[0, 0] * 1.5          # row 0: border copy (top row) ...
    f2[0, cell(3, 4)] = f3[0, 0] * 1.5          # ... and an identical interior copy -> exact tie
    f2[0, cell(7, 8)] = f3[0, 1] * 1.5          # bottom-right corner is NOT removed
    f2[0, cell(5, 1)] = f3[0, 2] * 1.5          # left border: removed
    f2[0, cell(4, 4)] = f3[0, 3] * 1.5          # ordinary interior match
    kp = torch.zeros(1, N, 3)
    ref_conf = orc.dual_softmax_confidence(f3, f2, 0.08)
    ref = orc.coarse_match_select(ref_conf, (hc, wc), (64, 72), kp, 0.1, 2)
    assert ref["i_ids"].tolist() == [0, 1, 3] and ref["j_ids"].tolist() == [cell(3, 4), cell(7, 8), cell(4, 4)]
    conf, (b_ids, i_ids, j_ids), mconf, _, mkc = _coarse_match(dev, f3, f2, kp, wc)
    assert conf[0, 0, cell(0, 4)].item() == conf[0, 0, cell(3, 4)].item()
    assert i_ids.tolist() == [0, 1, 3] and j_ids.tolist() == ref["j_ids"].tolist()
    assert torch.equal(mkc.cpu(), ref["mkpts_query_c"])


def test_fine_pair_kernel_is_bit_identical_to_the_one_match_kernel(sd, dev):
    """fine_pair_kernel (two matches per workgroup, shared weight fragments, hidden planes overlaid on X / Y) runs the same arithmetic
    per match as the one-match kernel it replaces: outputs bit for bit, at an odd match count (the last workgroup holds one match),
    NCHW and channels-last maps, both bf16 modes.  The one-match kernel stays reachable through OPHIP_FINE_PAIR=0 for this test."""
    import subprocess, sys, json, tempfile
    code = r'''
import ctypes, os, sys, torch, hashlib, json
sys.path.insert(0, os.getcwd())
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0"); hip.load()
B, N, hc, wc, K, cap = 2, 500, 12, 14, 301, 320
hf, wf = 4 * hc, 4 * wc
g = torch.Generator().manual_seed(11)
feat = torch.randn(B, 128, hf, wf, generator=g).to(dev)
desc = torch.randn(B, 128, N, generator=g).to(dev)
b_ids = torch.sort(torch.randint(0, B, (cap,), generator=g))[0].to(dev)
i_ids = torch.randint(0, N, (cap,), generator=g).to(dev)
j_ids = torch.randint(0, hc * wc, (cap,), generator=g).to(dev)
j_ids[:4] = torch.tensor([0, wc - 1, (hc - 1) * wc, hc * wc - 1])
mkc = (torch.stack([j_ids % wc, j_ids // wc], 1) * 8.0).float().contiguous()
cnt = torch.tensor([K], dtype=torch.int32, device=dev)
w = packing.pack_fine_layers_bf16(sd, "loftr_fine.layers.", 2).to(dev)
out = {}
for cl in (False, True):
    ff = feat.contiguous(memory_format=torch.channels_last) if cl else feat
    for ns in (3, 1):
        expec = torch.full((cap, 3), float("nan"), device=dev); mkf = torch.full((cap, 2), float("nan"), device=dev)
        dw, d3 = torch.zeros(cap, 25, 128, device=dev), torch.zeros(cap, 128, device=dev)
        hip.call("ophip_fine_refine_bf16", hip.ptr(ff), ff.stride(0), ff.stride(1), ff.stride(2), ff.stride(3), hf, wf, hip.ptr(desc), desc.stride(0), desc.stride(1),
                 hip.ptr(b_ids, torch.int64), hip.ptr(i_ids, torch.int64), hip.ptr(j_ids, torch.int64), hip.ptr(cnt, torch.int32), cap, hip.ptr(mkc),
                 hip.ptr(w, None), 2, ctypes.c_uint(2), 1, ns, wc, 4, 4.0, hip.ptr(expec), hip.ptr(mkf), hip.ptr(dw), hip.ptr(d3), hip.stream_handle())
        torch.cuda.synchronize()
        assert torch.isfinite(expec[:K]).all() and torch.isnan(expec[K:]).all()
        h = hashlib.sha256()
        for t in (expec[:K], mkf[:K], dw[:K], d3[:K]):
            h.update(t.cpu().numpy().tobytes())
        out[f"cl{int(cl)}_ns{ns}"] = h.hexdigest()
print(json.dumps(out))
'''
    res = {}
    for pair in ("1", "0"):
        env = dict(os.environ, OPHIP_FINE_PAIR=pair)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
        assert p.returncode == 0, p.stderr[-3000:]
        res[pair] = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["1"] == res["0"], (res["1"], res["0"])


def test_coarse_match_empty(dev):
    g = torch.Generator().manual_seed(6)
    f3, f2 = torch.randn(1, 50, 256, generator=g), torch.randn(1, 48, 256, generator=g)
    conf, (b_ids, i_ids, j_ids), mconf, mk3, mkc = _coarse_match(dev, f3, f2, torch.zeros(1, 50, 3), 8)
    assert len(i_ids) == 0 and mk3.shape == (0, 3) and mkc.shape == (0, 2)
    close(conf, orc.dual_softmax_confidence(f3, f2, 0.08), atol=1e-7)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_fine_refine_vs_oracle(sd, cfg, dev, channels_last, mode):
    B, N, hc, wc = 2, 90, 6, 7
    hf, wf = hc * 4, wc * 4
    g = torch.Generator().manual_seed(7)
    feat_f = torch.randn(B, 128, hf, wf, generator=g)
    desc = torch.randn(B, 128, N, generator=g)
    K = 37                                       # odd: the last bf16 workgroup holds a single match
    b_ids = torch.sort(torch.randint(0, B, (K,), generator=g))[0]
    i_ids = torch.randint(0, N, (K,), generator=g)
    j_ids = torch.randint(0, hc * wc, (K,), generator=g)
    j_ids[:4] = torch.tensor([0, wc - 1, (hc - 1) * wc, hc * wc - 1])        # corners: zero padding of the unfold
    mkc = torch.stack([j_ids % wc, j_ids // wc], 1) * 8.0
    f3, win = orc.fine_windows(feat_f, desc, b_ids, i_ids, j_ids, (hc, wc), 5)
    f3o, wino = orc.feature_transformer(sd, "loftr_fine", ["self", "cross"], 8, f3, win)
    ref = orc.fine_match(f3o, wino, mkc, (hc * 8, wc * 8), (hf, wf))
    ff = feat_f.to(dev)
    if channels_last:
        ff = ff.contiguous(memory_format=torch.channels_last)
        assert ff.stride(1) == 1
    cap = 64
    pad = lambda t: torch.cat([t, torch.zeros(cap - K, *t.shape[1:], dtype=t.dtype)]).to(dev)
    bd, idd, jd, mkd = pad(b_ids), pad(i_ids), pad(j_ids), pad(mkc)
    cnt = torch.tensor([K], dtype=torch.int32, device=dev)
    expec = torch.full((cap, 3), float("nan"), device=dev)
    mkf = torch.full((cap, 2), float("nan"), device=dev)
    dw, d3 = torch.empty(cap, 25, 128, device=dev), torch.empty(cap, 128, device=dev)
    dd = desc.to(dev)
    head = (hip.ptr(ff), ff.stride(0), ff.stride(1), ff.stride(2), ff.stride(3), hf, wf, hip.ptr(dd), dd.stride(0), dd.stride(1),
            hip.ptr(bd, torch.int64), hip.ptr(idd, torch.int64), hip.ptr(jd, torch.int64), hip.ptr(cnt, torch.int32), cap, hip.ptr(mkd))
    tail = (wc, 4, 4.0, hip.ptr(expec), hip.ptr(mkf), hip.ptr(dw), hip.ptr(d3), hip.stream_handle())
    if mode == "f32":
        w = torch.cat([packing.pack_fine_layer(sd, f"loftr_fine.layers.{i}.") for i in range(2)]).to(dev)
        hip.call("ophip_fine_refine", *head, hip.ptr(w), 2, ctypes.c_uint(2), 1, *tail)
        rt, at = RTOL, ATOL
    else:
        w = packing.pack_fine_layers_bf16(sd, "loftr_fine.layers.", 2).to(dev)
        assert w.numel() == hip.load().ophip_fine_bf16_wpack_bytes(2)
        hip.call("ophip_fine_refine_bf16", *head, hip.ptr(w, None), 2, ctypes.c_uint(2), 1, 3 if mode == "bf16x3" else 1, *tail)
        rt, at = (5e-4, 2e-4) if mode == "bf16x3" else (1e-1, 1e-1)
    print(f"{mode}: fine encoder max abs err window {(dw[:K].cpu() - wino).abs().max().item():.3e}, "
          f"3D {(d3[:K].cpu() - f3o[:, 0]).abs().max().item():.3e}, mkpts_f {(mkf[:K].cpu() - ref['mkpts_query_f']).abs().max().item():.3e} px")
    close(dw[:K], wino, rtol=rt, atol=at, msg="fine encoder, window stream")
    close(d3[:K], f3o[:, 0], rtol=rt, atol=at, msg="fine encoder, 3D stream")
    close(expec[:K, :2], ref["expec_f"][:, :2], rtol=rt, atol=at, msg="expec_f xy")
    close(expec[:K, 2], ref["expec_f"][:, 2], rtol=max(rt, 1e-3), atol=max(at, 1e-3), msg="expec_f std (ill-conditioned, see test_oracle_golden)")
    close(mkf[:K], ref["mkpts_query_f"], rtol=rt, atol=4 * at, msg="mkpts_query_f")
    assert torch.isnan(expec[K:]).all()          # surplus workgroups / the empty half of the last one write nothing


# ------------------------------------------------------------------------------------------------
# whole path: goldens captured from the reference, and the oracle at larger sizes
# ------------------------------------------------------------------------------------------------

def _run_features(model, inp, dev, **kw):
    d = to_dev(inp, dev)
    data = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    model.forward_features(data, d["feat_c"], d["feat_f"], inp["image_hw"], **kw)
    return data


# matches set aside at the confidence threshold, per test label: printed by conftest.pytest_terminal_summary as ONE line
# ("borderline_set_aside: {...}") so that the use of the hatch below is visible in a `pytest -q` tail
BORDERLINE = {}
# half-width of the window around the (strict) threshold inside which a match may fall on either side in another arithmetic,
# relative to the threshold: exact-f32 mode has no window at all; split-bf16 confidences measure <= 4e-5 relative off the oracle's
# (printed by every full-size test), the window is 5x that
BORDER_WINDOW = {"f32": 0.0, "bf16x3": 2e-4}


def _check_against(data, want, precision="f32", thr=0.1, label=None, want_rowmax=None, kp_tol=None):
    """Indices bit-exact wherever that is well defined: a match whose confidence sits within BORDER_WINDOW of the (strict) threshold
    may fall on either side of it in another arithmetic ("hard parts"); such borderline matches -- and only
    those -- are set aside, COUNTED (returned and recorded under `label`), and everything else must agree exactly.  Exact-f32 mode
    has no window: any difference fails.  `want_rowmax` (the oracle's / reference's row maxima of conf_matrix) lets the check also
    require that the other side's value of a set-aside row lies inside the window."""
    rt, at, rt_conf = TOL[precision]
    if kp_tol is not None:                        # (keypoint tolerances of a run whose fine stage is in another arithmetic; indices and confidences as usual)
        rt, at = kp_tol
    win = BORDER_WINDOW.get(precision, 0.0) * thr
    got_bi = torch.stack([data["b_ids"], data["i_ids"]], 1).cpu().numpy()
    want_bi = np.stack([np.asarray(want["b_ids"]), np.asarray(want["i_ids"])], 1)
    gk = {(int(b), int(i)): k for k, (b, i) in enumerate(got_bi)}
    wk = {(int(b), int(i)): k for k, (b, i) in enumerate(want_bi)}
    border = sorted(set(gk) ^ set(wk))
    if label is not None:
        BORDERLINE[label] = len(border)
    if border:
        assert win > 0.0, f"[{precision}] exact mode: match sets differ at {border}"
        gm, wm = data["mconf"].cpu().numpy(), np.asarray(want["mconf"])
        for key in border:
            c = gm[gk[key]] if key in gk else wm[wk[key]]
            assert abs(float(c) - thr) < win, f"match {key} differs with confidence {c}: not a threshold case (window {win:.1e})"
            if want_rowmax is not None and key[0] == 0:
                assert abs(float(np.asarray(want_rowmax)[key[1]]) - thr) < win, f"match {key}: the reference's row maximum is not a threshold case"
        assert len(border) <= max(1, len(wk) // 1000)
        print(f"[{precision}] {len(border)} borderline match(es) at the confidence threshold set aside: {border}")
        gsel = np.array([k for key, k in sorted(gk.items()) if key in wk], dtype=np.int64)
        wsel = np.array([k for key, k in sorted(wk.items()) if key in gk], dtype=np.int64)
        data = {k: (v[torch.as_tensor(gsel, device=v.device)] if (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == len(gk)) else v) for k, v in data.items()}
        want = {k: (np.asarray(v)[wsel] if np.asarray(v).shape[:1] == (len(wk),) else v) for k, v in want.items()}
    for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
        assert data[k].dtype == torch.int64
        np.testing.assert_array_equal(data[k].cpu().numpy(), np.asarray(want[k]), err_msg=k)        # bit-exact in every mode
    assert data["gt_mask"].dtype == torch.bool and not bool(data["gt_mask"].any())
    for k in ("mkpts_3d_db", "mkpts_query_c"):                                                       # pure gathers: exact
        assert data[k].dtype == torch.float32
        np.testing.assert_array_equal(data[k].cpu().numpy(), np.asarray(want[k]), err_msg=k)
    np.testing.assert_allclose(data["mconf"].cpu().numpy(), np.asarray(want["mconf"]), rtol=rt_conf, atol=1e-6, err_msg="mconf")
    np.testing.assert_allclose(data["mkpts_query_f"].cpu().numpy(), np.asarray(want["mkpts_query_f"]), rtol=rt, atol=at, err_msg="mkpts_query_f")
    np.testing.assert_allclose(data["expec_f"][:, :2].cpu().numpy(), np.asarray(want["expec_f"])[:, :2], rtol=rt, atol=at)
    np.testing.assert_allclose(data["expec_f"][:, 2].cpu().numpy(), np.asarray(want["expec_f"])[:, 2], rtol=1e-3, atol=2e-3 if kp_tol is None else max(2e-3, at))
    err_px = float(np.abs(data["mkpts_query_f"].cpu().numpy() - np.asarray(want["mkpts_query_f"])).max()) if len(want["mconf"]) else 0.0
    err_conf = float(np.abs(data["mconf"].cpu().numpy() / np.asarray(want["mconf"]) - 1).max()) if len(want["mconf"]) else 0.0
    print(f"[{precision}] K={len(want['mconf'])}: max |mkpts_query_f err| = {err_px:.2e} px, max mconf rel err = {err_conf:.2e}")
    return len(border)


def _pose_parity(data, ref_mk3d, ref_mkf, inp, label, n_set_aside=0):
    """north_star: pose R|t within 1e-4 relative.  The same deterministic host PnP (onepose_st_amd.pnp) is applied to the
    HIP path's matches and to the reference / oracle matches (the reference's pycolmap / OpenCV solvers are not
    available: pose parity is pinned to equal inputs -> equal estimator)."""
    Kc = inp["K"].numpy()
    pose_g, _, inl_g = ransac_PnP(Kc, data["mkpts_query_f"].cpu().numpy(), data["mkpts_3d_db"].cpu().numpy(), pnp_reprojection_error=7)
    pose_r, _, inl_r = ransac_PnP(Kc, np.asarray(ref_mkf), np.asarray(ref_mk3d), pnp_reprojection_error=7)
    dR = np.abs(pose_g[:, :3] - pose_r[:, :3]).max()
    dt = np.linalg.norm(pose_g[:, 3] - pose_r[:, 3]) / np.linalg.norm(pose_r[:, 3])
    gt = inp["pose_gt"].numpy()
    ang = np.degrees(np.arccos(np.clip((np.trace(pose_g[:, :3] @ gt[:, :3].T) - 1) / 2, -1, 1)))
    print(f"[{label}] pose parity: max |dR| = {dR:.2e}, |dt|/|t| = {dt:.2e}; vs planted pose: {ang:.3f} deg, "
          f"{np.linalg.norm(pose_g[:, 3] - gt[:, 3]) * 1000:.2f} mm; inliers {len(inl_g)}/{len(inl_r)}")
    assert dR < 1e-4 and dt < 1e-4
    assert abs(len(inl_g) - len(inl_r)) <= n_set_aside and ang < 0.5 and np.linalg.norm(pose_g[:, 3] - gt[:, 3]) < 5e-3      # (only a set-aside threshold match may differ)


def test_c1_against_reference_golden(model, sd, cfg, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "c1_feature_boundary.npz"))
    inp = make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg)
    data = _run_features(model, inp, dev)
    n_sa = _check_against(data, g, model.precision, label=f"c1_{model.precision}", want_rowmax=g["conf_rowmax"])
    if model.precision == "f32":
        assert n_sa == 0                                   # exact mode against the reference golden: no hatch
    conf = data["conf_matrix"]
    assert conf.shape == (1, 1000, 1200) and conf.dtype == torch.float32
    rt_conf = TOL[model.precision][2]
    np.testing.assert_allclose(conf.max(dim=2)[0][0].cpu().numpy(), g["conf_rowmax"], rtol=rt_conf, atol=1e-6)
    np.testing.assert_allclose(conf.max(dim=1)[0][0].cpu().numpy(), g["conf_colmax"], rtol=rt_conf, atol=1e-6)
    assert data["bs"] == 1 and tuple(data["q_hw_c"]) == (30, 40) and tuple(data["q_hw_f"]) == (120, 160) and data["W"] == 5
    _pose_parity(data, g["mkpts_3d_db"], g["mkpts_query_f"], inp, f"c1 {model.precision} vs reference golden", n_sa)


def test_b2_ragged_against_reference_golden(model, sd, cfg, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "b2_ragged_feature_boundary.npz"))
    i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
    i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
    both = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
    both["image_hw"] = i0["image_hw"]
    data = _run_features(model, both, dev)
    _check_against(data, g, model.precision, label=f"b2_{model.precision}")
    # shared object block passed as an expand() (stride-0 batch) gives the same answer
    exp = dict(both)
    for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db"):
        exp[k] = i0[k].expand(2, *i0[k].shape[1:])
    data2 = _run_features(model, exp, dev)
    for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f"):
        assert torch.equal(data[k], data2[k]), k


def test_full_forward_with_backbone_empty_path(model, sd, cfg, dev, golden_dir):
    """reference API: model(data) with query_image; random image => K = 0 (fine_preprocess.py:34-37,
    fine_matching.py:46-55).  The backbone runs on PyTorch-ROCm/MIOpen, so tolerances are looser."""
    g = np.load(os.path.join(golden_dir, "full_forward_empty.npz"))
    img = torch.rand(1, 1, 64, 96, generator=torch.Generator().manual_seed(5))
    obj = make_synthetic_inputs(sd, n_points=200, image_hw=(64, 96), n_plant=0, seed=4, config=cfg)
    data = {"query_image": img.to(dev), "keypoints3d": obj["keypoints3d"].to(dev), "descriptors3d_db": obj["descriptors3d_db"].to(dev),
            "descriptors3d_coarse_db": obj["descriptors3d_coarse_db"].to(dev)}
    assert model(data) is None
    assert len(data["i_ids"]) == 0
    for k in ("mconf", "mkpts_3d_db", "mkpts_query_c", "expec_f", "mkpts_query_f"):
        assert tuple(data[k].shape) == tuple(g[k + "_shape"]), k
    np.testing.assert_allclose(data["conf_matrix"].max(dim=2)[0][0].cpu().numpy(), g["conf_rowmax"], rtol=5e-2, atol=1e-6)


def test_full_forward_with_backbone_planted_against_reference_golden(model, sd, cfg, dev, golden_dir):
    """Backbone + K > 0 against the REFERENCE: its own backbone ran on this synthetic image and a forward hook added the planted
    feature maps to the backbone's two outputs (tests/golden/make_golden.py, case D: 180 matches, every confidence > 0.05 away from
    the threshold).  Here: the HIP convolution backbone (exact-f32 mode: MIOpen fp32) -> + the same maps -> rows a1-a11."""
    g = np.load(os.path.join(golden_dir, "full_forward_planted.npz"))
    img = torch.rand(1, 1, 128, 160, generator=torch.Generator().manual_seed(6))
    inp = make_synthetic_inputs(sd, n_points=500, image_hw=(128, 160), n_plant=180, seed=7, config=cfg)
    data = {k: inp[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    data.update({"bs": 1, "q_hw_i": torch.Size((128, 160))})
    if model.hip_backbone:
        fc, ff = model.backbone_features(img.to(dev))                 # channels-last memory, positional encoding already added (linear)
        fc.add_(inp["feat_c"].to(dev)), ff.add_(inp["feat_f"].to(dev))
        model.enqueue_features(data, fc, ff, _pe_applied=True).finish()
    else:
        with torch.no_grad():
            fc, ff = model.backbone(img.to(dev))
        model.forward_features(data, fc + inp["feat_c"].to(dev), ff + inp["feat_f"].to(dev))
    assert len(g["i_ids"]) == 180
    n_sa = _check_against(data, g, model.precision, label=f"full_forward_{model.precision}", want_rowmax=g["conf_rowmax"])
    assert n_sa == 0                                                   # every reference confidence is >= 0.05 away from the threshold
    rt = {"f32": 1e-3, "bf16x3": 2e-3}[model.precision]                # 22 convolution layers in front of the path
    np.testing.assert_allclose(data["conf_matrix"].max(dim=2)[0][0].cpu().numpy(), g["conf_rowmax"], rtol=rt, atol=1e-6)
    np.testing.assert_allclose(data["conf_matrix"].max(dim=1)[0][0].cpu().numpy(), g["conf_colmax"], rtol=rt, atol=1e-6)
    _pose_parity(data, g["mkpts_3d_db"], g["mkpts_query_f"], inp, f"full forward {model.precision} vs reference golden")


def test_c2_full_size_against_oracle_and_properties(model, sd, cfg, dev):
    """BASELINE config c2 (7000 x 4800): indices bit-exact against the oracle, plus size-independent properties
    of the result checked on the device: mutual-nearest, threshold, border, ordering, planted recall."""
    inp = make_synthetic_inputs(sd, n_points=7000, image_hw=(480, 640), n_plant=3000, seed=1, config=cfg)
    data = _run_features(model, inp, dev)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = orc.forward_from_features(sd, cfg, inp, inp["feat_c"], inp["feat_f"], inp["image_hw"])
    K = len(ref["i_ids"])
    assert K > 2000
    n_sa = _check_against(data, {k: ref[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mconf", "mkpts_3d_db", "mkpts_query_c",
                                                             "mkpts_query_f", "expec_f")}, model.precision, label=f"c2_{model.precision}",
                          want_rowmax=ref["conf_matrix"].max(dim=2)[0][0].numpy())
    _pose_parity(data, ref["mkpts_3d_db"].numpy(), ref["mkpts_query_f"].numpy(), inp, f"c2 {model.precision} vs oracle", n_sa)
    conf, i, j = data["conf_matrix"][0], data["i_ids"], data["j_ids"]
    v = conf[i, j]
    assert torch.equal(v, data["mconf"])
    assert bool((v > 0.1).all()) and torch.equal(v, conf.max(dim=1)[0][i]) and torch.equal(v, conf.max(dim=0)[0][j])
    assert bool(((j // 80) >= 2).all()) and bool(((j % 80) >= 2).all())
    assert bool((i[1:] > i[:-1]).all())
    assert bool((conf.sum(dim=1) <= 1.0 + 1e-4).all())               # product of two softmaxes
    planted = set(zip(inp["planted_i"].tolist(), inp["planted_j"].tolist()))
    got = set(zip(i.tolist(), j.tolist()))
    assert len(got & planted) >= 0.95 * len(got)
    # determinism: a second run is bit-identical (no float atomics anywhere)
    data2 = _run_features(model, inp, dev)
    for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f", "conf_matrix"):
        assert torch.equal(data[k], data2[k]), k


@pytest.mark.parametrize("precision", ["bf16"])
def test_c1_bf16_modes_against_reference_golden(sd, cfg, dev, golden_dir, precision):
    """The bf16-pipe encoder inside the whole path, against the goldens captured from the reference.
    split-bf16: indices bit-exact, floats at the f32 tolerances x5.  plain bf16: report the index mismatch count
    (margin-free matches may flip) and check floats on the common matches at 5e-2."""
    g = np.load(os.path.join(golden_dir, "c1_feature_boundary.npz"))
    m = _model(sd, cfg, dev, precision)
    inp = make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg)
    data = _run_features(m, inp, dev)
    got = set(zip(data["i_ids"].tolist(), data["j_ids"].tolist()))
    want = set(zip(g["i_ids"].tolist(), g["j_ids"].tolist()))
    print(f"{precision}: K={len(got)} reference K={len(want)} symmetric difference={len(got ^ want)}")
    if precision == "bf16x3":
        assert got == want
        np.testing.assert_array_equal(data["j_ids"].cpu().numpy(), g["j_ids"])
        np.testing.assert_allclose(data["mconf"].cpu().numpy(), g["mconf"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(data["mkpts_query_f"].cpu().numpy(), g["mkpts_query_f"], rtol=1e-4, atol=1e-3)
    else:
        assert len(got ^ want) <= 0.02 * len(want)
        planted = set(zip(g["planted_i"].tolist(), g["planted_j"].tolist()))
        assert (want & planted) <= got            # every planted (margin) match survives bf16


def test_c4_large_object_against_oracle(sd, cfg, dev):
    """BASELINE config c4: 15 000 3D points x 19 200 cells (960 x 1280 image): the 1.15 GB conf_matrix case.
    Indices bit-exact against the oracle, keypoints at the bf16x3 tolerance, pose parity, mutual-NN on the device."""
    m = _model(sd, cfg, dev, "bf16x3")
    inp = make_synthetic_inputs(sd, n_points=15000, image_hw=(960, 1280), n_plant=6000, seed=2, config=cfg)
    data = _run_features(m, inp, dev)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = orc.forward_from_features(sd, cfg, inp, inp["feat_c"], inp["feat_f"], inp["image_hw"])
    assert len(ref["i_ids"]) > 4000 and data["conf_matrix"].shape == (1, 15000, 19200)
    n_sa = _check_against(data, {k: ref[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mconf", "mkpts_3d_db", "mkpts_query_c",
                                                             "mkpts_query_f", "expec_f")}, "bf16x3", label="c4_bf16x3",
                          want_rowmax=ref["conf_matrix"].max(dim=2)[0][0].numpy())
    _pose_parity(data, ref["mkpts_3d_db"].numpy(), ref["mkpts_query_f"].numpy(), inp, "c4 bf16x3 vs oracle", n_sa)
    conf, i, j = data["conf_matrix"][0], data["i_ids"], data["j_ids"]
    v = conf[i, j]
    assert torch.equal(v, data["mconf"]) and torch.equal(v, conf.max(dim=1)[0][i]) and torch.equal(v, conf.max(dim=0)[0][j])
    del conf, data


def test_c3_batch32_matches_single_frame_runs(sd, cfg, dev):
    """BASELINE config c3: 32 frames per call sharing one 3D block (expand(), stride-0 batch), 7000 x 4800, coarse + fine.
    Every frame of the batch must equal its own B = 1 run bit for bit (frames never interact: no cross-frame state,
    fixed-order reductions), ids must come out in ascending (b, i) order, conf_matrix is [32, 7000, 4800] (4.3 GB)."""
    m = _model(sd, cfg, dev, "bf16x3")
    frames = [make_synthetic_inputs(sd, n_points=7000, image_hw=(480, 640), n_plant=3000, seed=1, config=cfg, frame=f) for f in range(4)]
    singles = [_run_features(m, fr, dev) for fr in frames]
    B = 32
    order = [k % 4 for k in range(B)]
    batch = {"feat_c": torch.cat([frames[k]["feat_c"] for k in order]), "feat_f": torch.cat([frames[k]["feat_f"] for k in order]),
             "image_hw": frames[0]["image_hw"]}
    for key in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db"):
        batch[key] = frames[0][key].expand(B, *frames[0][key].shape[1:])
    d = to_dev(batch, dev)
    data = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    m.forward_features(data, d["feat_c"], d["feat_f"], batch["image_hw"])
    assert data["conf_matrix"].shape == (B, 7000, 4800)
    b_ids = data["b_ids"]
    key = b_ids * 7000 + data["i_ids"]
    assert bool((key[1:] > key[:-1]).all())
    for b in range(B):
        sel = b_ids == b
        one = singles[order[b]]
        for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "mkpts_3d_db", "expec_f"):
            assert torch.equal(data[k][sel], one[k]), (b, k)
    assert torch.equal(data["conf_matrix"][5], singles[order[5]]["conf_matrix"][0])
    # the same batch with the object cache: the shared block's first-layer rows are computed ONCE (one row set, stride 0), bit-identical
    keep = {k: data[k].clone() for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f")}
    del data, singles
    torch.cuda.empty_cache()
    ccfg = copy.deepcopy(cfg)
    ccfg["hip_cache_object"] = True
    mc = _model(sd, ccfg, dev, "bf16x3")
    # (expanded ON the device: stride-0 views of one resident block, what bench.py's config-3 leg passes; `.to(dev)` of a host-side expand
    #  above materialised 32 copies, which the plain run treats as 32 objects)
    data = {k: frames[0][k].to(dev).expand(B, *frames[0][k].shape[1:]) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    mc.forward_features(data, d["feat_c"], d["feat_f"], batch["image_hw"])
    assert mc._obj_cache["y3d0"].shape[0] == 1
    for k, v in keep.items():
        assert torch.equal(data[k], v), ("object cache", k)
    del data


def test_fine_disabled_and_encoder_disabled(sd, cfg, dev):
    inp = make_synthetic_inputs(sd, n_points=300, image_hw=(96, 136), n_plant=100, seed=9, config=cfg)
    c2 = copy.deepcopy(cfg)
    c2["fine_matching"]["enable"] = False
    m = OnePosePlus_model(c2).eval()
    m.load_state_dict(sd)
    m.to(dev)
    data = _run_features(m, inp, dev)
    assert "expec_f" not in data and torch.equal(data["mkpts_query_f"], data["mkpts_query_c"])       # OnePosePlusModel.py:174-181
    c3 = copy.deepcopy(cfg)
    c3["loftr_fine"]["enable"] = False
    m3 = OnePosePlus_model(c3).eval()
    m3.load_state_dict(sd)
    m3.to(dev)
    d3 = _run_features(m3, inp, dev)
    with torch.no_grad():
        ref = orc.forward_from_features(sd, c3, inp, inp["feat_c"], inp["feat_f"], inp["image_hw"])
    assert torch.equal(d3["i_ids"].cpu(), ref["i_ids"])
    close(d3["mkpts_query_f"], ref["mkpts_query_f"])


def test_pipelined_frames_match_sequential_runs(model, sd, cfg, dev):
    """Several frames in flight (fine stage and read-back on side streams, frame t + 1's input kernels under frame t's
    refinement), finished out of order: every frame is bit-identical to its own stand-alone run, with the fine overlap on and
    off, and the host copies of the read-back block equal the device tensors."""
    frames = [make_synthetic_inputs(sd, n_points=1500, image_hw=(160, 224), n_plant=500, seed=21, config=cfg, frame=f) for f in range(4)]
    obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    feats = [(f["feat_c"].to(dev), f["feat_f"].to(dev)) for f in frames]
    keys = ("i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "expec_f", "mkpts_3d_db", "m_bids", "gt_mask")
    alone = []
    for fc, ff in feats:
        d = dict(obj)
        model.enqueue_features(d, fc, ff, frames[0]["image_hw"]).finish()
        torch.cuda.synchronize()
        alone.append({k: d[k].clone() for k in keys})
    assert len(alone[0]["i_ids"]) > 300 and not torch.equal(alone[0]["j_ids"], alone[1]["j_ids"])
    for rep in range(3):
        pend, datas = [], []
        for fc, ff in feats:
            d = dict(obj)
            datas.append(d)
            pend.append(model.enqueue_features(d, fc, ff, frames[0]["image_hw"], host_copy=True))
        for idx in (2, 0, 3, 1):
            pend[idx].finish()
        for d, p, ref in zip(datas, pend, alone):
            for k in keys:
                assert torch.equal(d[k], ref[k]), (rep, k)
            K = len(ref["i_ids"])
            assert p.host["K"] == K
            np.testing.assert_array_equal(p.host["mkpts_2d"], ref["mkpts_query_f"].cpu().numpy())
            np.testing.assert_array_equal(p.host["mkpts_3d_db"], ref["mkpts_3d_db"].cpu().numpy())
    saved = model.overlap_fine
    try:
        model.overlap_fine = False
        d = dict(obj)
        model.enqueue_features(d, *feats[1], frames[0]["image_hw"]).finish()
        for k in keys:
            assert torch.equal(d[k], alone[1][k]), k
    finally:
        model.overlap_fine = saved


def test_dropped_pending_frame_does_not_disturb_the_next(model, sd, cfg, dev):
    """A frame enqueued and then dropped without finish() (an exception in a pipeline): its side-stream work is waited for
    before its buffers and pinned block are released, and the following frames are still bit-identical to stand-alone runs."""
    frames = [make_synthetic_inputs(sd, n_points=900, image_hw=(128, 160), n_plant=300, seed=31, config=cfg, frame=f) for f in range(2)]
    obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    feats = [(f["feat_c"].to(dev), f["feat_f"].to(dev)) for f in frames]
    keys = ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f")
    ref = dict(obj)
    model.enqueue_features(ref, *feats[1], frames[0]["image_hw"]).finish()
    ref = {k: ref[k].clone() for k in keys}
    for rep in range(3):
        dropped = model.enqueue_features(dict(obj), *feats[0], frames[0]["image_hw"], host_copy=True)
        d = dict(obj)
        nxt = model.enqueue_features(d, *feats[1], frames[0]["image_hw"], host_copy=True)
        del dropped                                           # never finished
        nxt.finish()
        for k in keys:
            assert torch.equal(d[k], ref[k]), (rep, k)
    p = model.enqueue_features(dict(obj), *feats[0], frames[0]["image_hw"])
    p.close()
    p.close()                                                 # idempotent
    assert p.done


def test_input_kernels_on_the_side_stream_are_bit_identical(sd, cfg, dev):
    """enqueue_features(inputs_ready=True): PE / transposes / keypoint encoding run on a side stream ahead of the work queued on
    the compute stream (bench.py's pipeline); frames pipelined that way equal their plain forward_features() runs bit for bit."""
    frames = [make_synthetic_inputs(sd, n_points=1100, image_hw=(128, 192), n_plant=400, seed=43, config=cfg, frame=f) for f in range(4)]
    obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    keys = ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f", "conf_matrix")
    m = _model(sd, cfg, dev, "bf16x3")
    feats = [(f["feat_c"].to(dev), f["feat_f"].to(dev)) for f in frames]
    torch.cuda.synchronize()
    plain = []
    for f, (fc, ff) in zip(frames, feats):
        d = dict(obj)
        m.forward_features(d, fc, ff, f["image_hw"])
        plain.append({k: d[k].clone() for k in keys})
    pend, outs = [], []
    for f, (fc, ff) in zip(frames, feats):                       # all four in flight before the first finish()
        d = dict(obj)
        pend.append((d, m.enqueue_features(d, fc, ff, f["image_hw"], inputs_ready=True)))
    for d, p in pend:
        p.finish()
        outs.append(d)
    for a, b in zip(plain, outs):
        assert len(a["i_ids"]) > 200
        for k in keys:
            assert torch.equal(a[k], b[k]), k


def test_frame_call_equals_the_stage_by_stage_path(sd, cfg, dev):
    """config["hip_frame_call"] (default on): the frame as ONE C call (csrc/frame.hip, one device block) against the same frame issued
    stage by stage from Python -- every output bit for bit, NCHW and channels-last fine maps, with and without the input stream,
    and interleaved with stage-by-stage frames on the same stream."""
    keys = ("b_ids", "i_ids", "j_ids", "m_bids", "gt_mask", "mconf", "mkpts_3d_db", "mkpts_query_c", "mkpts_query_f", "expec_f", "conf_matrix")
    c_on, c_off = copy.deepcopy(cfg), copy.deepcopy(cfg)
    c_on["hip_frame_call"], c_off["hip_frame_call"] = True, False
    models = []
    for c in (c_on, c_off):
        m = OnePosePlus_model(c).eval()
        m.load_state_dict(sd, strict=True)
        models.append(m.to(dev))
    fast, slow = models
    assert fast.frame_call and not slow.frame_call
    frames = [make_synthetic_inputs(sd, n_points=900, image_hw=(128, 160), n_plant=350, seed=47, config=cfg, frame=f) for f in range(3)]
    obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    for f in frames:
        fc, ff = f["feat_c"].to(dev), f["feat_f"].to(dev)
        for ffv in (ff, ff.contiguous(memory_format=torch.channels_last)):
            a, b, c = dict(obj), dict(obj), dict(obj)
            slow.forward_features(a, fc, ffv, f["image_hw"])
            fast.forward_features(b, fc, ffv, f["image_hw"])
            torch.cuda.synchronize()
            fast.enqueue_features(c, fc, ffv, f["image_hw"], inputs_ready=True, host_copy=True).finish()
            assert len(a["i_ids"]) > 150
            for k in keys:
                assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert fast._frame_plans and not slow._frame_plans


def test_deferred_fine_stage_is_bit_identical_to_the_in_order_frame(sd, cfg, dev):
    """ophip_frame_enqueue keeps frame t's fine stage back until frame t + 1's encoder and similarity tiles are queued (it then runs
    beside frame t + 1's HBM-bound confidence pass); OPHIP_FRAME_DEFER_FINE=0 launches it behind its own selection.  Both orders, a
    pipeline three frames deep with the input kernels on their side stream (40 frames: the ring of 16 event sets comes round twice), a
    frame finished with no successor and three dropped unfinished: every output bit for bit the same."""
    import subprocess, sys, json
    code = r'''
import os, sys, torch, hashlib, json
sys.path.insert(0, os.getcwd())
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg); dev = torch.device("cuda:0")
m = OnePosePlus_model(cfg).eval(); m.load_state_dict(sd, strict=True); m.to(dev)
frames = [make_synthetic_inputs(sd, n_points=1500, image_hw=(160, 224), n_plant=500, seed=21, config=cfg, frame=f) for f in range(4)]
obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
feats = [(f["feat_c"].to(dev), f["feat_f"].to(dev)) for f in frames]
keys = ("i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "expec_f", "mkpts_3d_db")
torch.cuda.synchronize()
out = []
def digest(d, p):
    h = hashlib.sha256()
    for k in keys:
        h.update(d[k].cpu().numpy().tobytes())
    h.update(p.host["mkpts_2d"].tobytes())
    return h.hexdigest()
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    inflight = []
    for i in range(40):                          # crosses the ring of 16 event sets twice
        d = dict(obj)
        inflight.append((d, m.enqueue_features(d, *feats[i % 4], frames[0]["image_hw"], host_copy=True, inputs_ready=True)))
        if i in (5, 21, 22):
            inflight.pop()                       # dropped unfinished (its fine stage is still kept back at this point)
        if len(inflight) >= 3:
            d0, p0 = inflight.pop(0); p0.finish(); out.append(digest(d0, p0))
    while inflight:
        d0, p0 = inflight.pop(0); p0.finish(); out.append(digest(d0, p0))
    d = dict(obj)
    p = m.enqueue_features(d, *feats[2], frames[0]["image_hw"], host_copy=True)       # alone: finished with no successor
    p.finish(); out.append(digest(d, p))
torch.cuda.synchronize()
print(json.dumps(out))
'''
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, OPHIP_FRAME_DEFER_FINE=mode)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
        assert p.returncode == 0, p.stderr[-3000:]
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(res["1"]) == 38 and len(set(res["1"])) == 4          # four distinct frames, cycled
    assert res["1"] == res["0"]


def test_object_cache_is_bit_identical(sd, cfg, dev, monkeypatch):
    """config["hip_cache_object"]: what depends on the resident object block and the weights alone -- the keypoint encoding (rows a2 + a3),
    the first encoder layer's 3D rows and the K^T V | Ksum block of those rows as the second layer's source (transformer.py:148-159) -- is
    computed once and re-used by the following frames: same results bit for bit as the uncached model, at both cache depths, pipelined with
    the input kernels on their side stream, for a batch that shares one object (computed ONCE, stride 0), for a batch of distinct objects and
    for a padded frame; a changed block (or an in-place edit) rebuilds the entry."""
    import copy
    frames = [make_synthetic_inputs(sd, n_points=1100, image_hw=(128, 192), n_plant=400, seed=41, config=cfg, frame=f) for f in range(3)]
    obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    keys = ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f", "conf_matrix")
    plain = OnePosePlus_model(copy.deepcopy(cfg)).eval()
    plain.load_state_dict(sd, strict=True)
    plain.to(dev)
    ccfg = copy.deepcopy(cfg)
    ccfg["hip_cache_object"] = True
    cached = OnePosePlus_model(ccfg).eval()
    cached.load_state_dict(sd, strict=True)
    cached.to(dev)
    from onepose_st_amd import ops
    n_calls = ops.CALLS["frame_enqueue"]
    singles = []
    for f in frames:
        a, b = dict(obj), dict(obj)
        plain.forward_features(a, f["feat_c"].to(dev), f["feat_f"].to(dev), f["image_hw"])
        cached.forward_features(b, f["feat_c"].to(dev), f["feat_f"].to(dev), f["image_hw"])
        assert len(a["i_ids"]) > 200
        for k in keys:
            assert torch.equal(a[k], b[k]), k
        singles.append({k: a[k].clone() for k in keys})
    assert ops.CALLS["frame_enqueue"] == n_calls + 6            # the cache MISS (first frame) takes the one-call frame path like the hits
    entry = cached._obj_cache
    assert entry is not None and entry["y3d0"] is not None and entry["kv1"] is not None       # first layer "self": the deep entry
    assert tuple(entry["y3d0"].shape) == (1, 1100, 256)
    d = dict(obj)
    cached.forward_features(d, frames[0]["feat_c"].to(dev), frames[0]["feat_f"].to(dev), frames[0]["image_hw"])
    assert cached._obj_cache is entry                           # same block: the cached entry was used
    # the cached rows ARE the uncached first layer's 3D rows: ophip_encoder_object_x3w8 against the plain layer call on both streams
    W = cached._weights(dev)
    lib = hip.load()
    P = hip.ptr
    x2 = torch.randn(1, 24 * 16, 256, device=dev)
    y3, y2 = torch.empty_like(entry["x3d"]), torch.empty_like(x2)
    ws = torch.empty(lib.ophip_encoder_x3w8_workspace_bytes(1, 1100, 24 * 16), device=dev, dtype=torch.uint8)
    hip.call("ophip_encoder_layer_x3w8", P(entry["x3d"]), P(x2), P(y3), P(y2), 1, 1100, 24 * 16, P(W["coarse_x3"][0], None), P(W["coarse_x3"][1], None),
             0, 0, 0, P(ws, None), hip.stream_handle())
    torch.cuda.synchronize()
    assert torch.equal(y3, entry["y3d0"])
    # the encoding alone (depth 1), and both depths pipelined with the input kernels (and the first layer's K / V half) on the side stream
    for depth in ("1", "2"):
        monkeypatch.setenv("OPHIP_OBJECT_CACHE_DEPTH", depth)
        cached._obj_cache = None
        pend, outs = [], []
        for f in frames:
            dd = dict(obj)
            pend.append((dd, cached.enqueue_features(dd, f["feat_c"].to(dev), f["feat_f"].to(dev), f["image_hw"], inputs_ready=True)))
        for dd, p in pend:
            p.finish()
            outs.append(dd)
        assert (cached._obj_cache["y3d0"] is None) == (depth == "1")
        for f, dd in zip(frames, outs):
            a = dict(obj)
            plain.forward_features(a, f["feat_c"].to(dev), f["feat_f"].to(dev), f["image_hw"])
            for k in keys:
                assert torch.equal(a[k], dd[k]), (depth, k)
    monkeypatch.delenv("OPHIP_OBJECT_CACHE_DEPTH")
    # a batch of three frames sharing ONE object block (stride-0 expand: one entry of one row set) and a batch of two distinct objects
    B = 3
    fc = torch.cat([f["feat_c"] for f in frames]).to(dev)
    ff = torch.cat([f["feat_f"] for f in frames]).to(dev)
    shared = {k: v.expand(B, *v.shape[1:]) for k, v in obj.items()}
    cached._obj_cache = None
    a, b = dict(shared), dict(shared)
    plain.forward_features(a, fc, ff, frames[0]["image_hw"])
    cached.forward_features(b, fc, ff, frames[0]["image_hw"])
    assert cached._obj_cache["y3d0"].shape[0] == 1 and len(a["i_ids"]) > 600
    for k in keys + ("b_ids",):
        assert torch.equal(a[k], b[k]), ("shared", k)
    assert plain._obj_cache is None                             # the plain model does a shared batch's object-only work once per CALL and keeps nothing
    for bi in range(B):                                         # ... and every frame of the batch equals its own single-frame run
        sel = a["b_ids"] == bi
        for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f"):
            assert torch.equal(a[k][sel], singles[bi][k]), ("shared vs single", bi, k)
        assert torch.equal(a["conf_matrix"][bi], singles[bi]["conf_matrix"][0])
    other = make_synthetic_inputs(sd, n_points=1100, image_hw=(128, 192), n_plant=400, seed=77, config=cfg, frame=0)
    two = {k: torch.cat([frames[0][k], other[k]]).to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    fc2 = torch.cat([frames[0]["feat_c"], other["feat_c"]]).to(dev)
    ff2 = torch.cat([frames[0]["feat_f"], other["feat_f"]]).to(dev)
    a, b = dict(two), dict(two)
    plain.forward_features(a, fc2, ff2, frames[0]["image_hw"])
    cached.forward_features(b, fc2, ff2, frames[0]["image_hw"])
    assert cached._obj_cache["y3d0"].shape[0] == 2
    for k in keys + ("b_ids",):
        assert torch.equal(a[k], b[k]), ("two objects", k)
    # a padded query image (query_image_mask): the mask touches the 2D stream only, the cached 3D rows stay valid
    hc, wc = 128 // 8, 192 // 8
    qm = torch.ones(1, hc, wc, dtype=torch.bool)
    qm[:, :, wc - 5:] = False
    a, b = dict(obj), dict(obj)
    a["query_image_mask"] = b["query_image_mask"] = qm.to(dev)
    plain.forward_features(a, frames[2]["feat_c"].to(dev), frames[2]["feat_f"].to(dev), frames[2]["image_hw"])
    cached.forward_features(b, frames[2]["feat_c"].to(dev), frames[2]["feat_f"].to(dev), frames[2]["image_hw"])
    assert len(a["i_ids"]) > 100
    for k in keys:
        assert torch.equal(a[k], b[k]), ("masked", k)
    entry = cached._obj_cache
    obj["keypoints3d"].mul_(1.5)                               # in-place edit bumps the version: rebuild
    a, b = dict(obj), dict(obj)
    plain.forward_features(a, frames[1]["feat_c"].to(dev), frames[1]["feat_f"].to(dev), frames[1]["image_hw"])
    cached.forward_features(b, frames[1]["feat_c"].to(dev), frames[1]["feat_f"].to(dev), frames[1]["image_hw"])
    assert cached._obj_cache is not entry
    for k in keys:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("size", ["ragged_b2", "c1", "c2"])
def test_lazy_conf_matrix_is_bit_identical_to_the_eager_form(sd, cfg, dev, size):
    """config["hip_conf_matrix"] = "lazy": conf_matrix is never stored (two passes over the similarity tiles, candidates only);
    every output the reference's inference callers read -- indices, mconf, keypoints, expec_f -- equals the eager form bit for bit,
    on the one-call frame path and on the stage-by-stage path; data["conf_matrix"] materialises on first use to the eager matrix"""
    from onepose_st_amd.model import LazyConfMatrix
    if size == "ragged_b2":
        i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
        i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
        inp = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
        inp["image_hw"] = i0["image_hw"]
    elif size == "c1":
        inp = make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg)
    else:
        inp = make_synthetic_inputs(sd, n_points=7000, image_hw=(480, 640), n_plant=3000, seed=1, config=cfg)
    eager = _model(sd, cfg, dev, "bf16x3")
    cl = copy.deepcopy(cfg)
    cl["hip_conf_matrix"] = "lazy"
    lazy = _model(sd, cl, dev, "bf16x3")
    keys = ("b_ids", "i_ids", "j_ids", "m_bids", "mconf", "mkpts_3d_db", "mkpts_query_c", "mkpts_query_f", "expec_f", "gt_mask")
    want = _run_features(eager, inp, dev)
    assert want["i_ids"].numel() > 50
    for frame_call in (True, False):
        lazy.frame_call = frame_call
        got = _run_features(lazy, inp, dev)
        for k in keys:
            assert torch.equal(got[k], want[k]), (k, frame_call)
        cm = got["conf_matrix"]
        assert isinstance(cm, LazyConfMatrix) and tuple(cm.shape) == tuple(want["conf_matrix"].shape) and cm._t is None
        if size != "c2" or frame_call:
            assert torch.equal(cm[0, :7], want["conf_matrix"][0, :7]) and cm._t is not None            # indexing materialises
            assert torch.equal(torch.max(cm, dim=2)[0], want["conf_matrix"].max(dim=2)[0])             # torch functions too
            assert torch.equal(cm.materialize(), want["conf_matrix"])
    with pytest.raises(ValueError):
        _model(sd, cl, dev, "f32")                    # the exact mode always materialises


def test_lazy_conf_matrix_exact_tie_raises_the_rerun_flag(dev):
    """the one thing the lazy selection cannot decide without the stored row: a row maximum tied exactly between two columns whose
    first column fails the mask (here: it lies in the removed border, the construction of test_coarse_match_tie_and_border_semantics).
    The C entry point then raises count[1] (OnePosePlus_model re-runs such a frame with a conf buffer); without a tie count[1] = 0 and
    the match lists equal the eager form's."""
    hc, wc, N = 8, 9, 40
    M = hc * wc
    g = torch.Generator().manual_seed(5)
    f3 = torch.randn(1, N, 256, generator=g) * 1.5
    f2 = torch.randn(1, M, 256, generator=g)
    cell = lambda y, x: y * wc + x
    f2[0, cell(7, 8)] = f3[0, 1] * 1.5
    f2[0, cell(4, 4)] = f3[0, 3] * 1.5
    kp = torch.zeros(1, N, 3)

    def lazy_call(f2_):
        d3, d2, dk = f3.to(dev), f2_.to(dev), kp.to(dev)
        ws = torch.empty(hip.load().ophip_coarse_workspace_floats(1, N, M), device=dev)
        ids = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3)]
        mconf, mk3, mkc = torch.empty(N, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 2, device=dev)
        cnt = torch.full((4,), 7, dtype=torch.int32, device=dev)
        hip.call("ophip_coarse_match", hip.ptr(d3), hip.ptr(d2), hip.ptr(dk), 0, 1, N, M, wc, 0.08, 0.1, 2, 8.0, None, hip.ptr(ws),
                 *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(cnt, torch.int32), 3,
                 hip.stream_handle())
        K = int(cnt[0])
        return K, int(cnt[1]), ids[1][:K].cpu(), ids[2][:K].cpu(), mconf[:K].cpu()
    # no tie: flag 0, lists equal the eager form's
    conf_e, ids_e, mconf_e, _, _ = _coarse_match(dev, f3, f2, kp, wc, nsplit=3)
    K, flag, i_l, j_l, m_l = lazy_call(f2)
    assert flag == 0 and K == len(ids_e[1]) == 2 and torch.equal(i_l, ids_e[1].cpu()) and torch.equal(j_l, ids_e[2].cpu()) and torch.equal(m_l, mconf_e.cpu())
    # border copy + identical interior copy of row 0's descriptor: exact tie, first column masked out
    f2t = f2.clone()
    f2t[0, cell(0, 4)] = f3[0, 0] * 1.5
    f2t[0, cell(3, 4)] = f3[0, 0] * 1.5
    conf_t, ids_t, _, _, _ = _coarse_match(dev, f3, f2t, kp, wc, nsplit=3)
    assert conf_t[0, 0, cell(0, 4)].item() == conf_t[0, 0, cell(3, 4)].item() and ids_t[2].tolist() == [cell(3, 4), cell(7, 8), cell(4, 4)]
    K, flag, i_l, j_l, _ = lazy_call(f2t)
    assert flag == 1                                   # "run this frame again with a conf buffer"
    with pytest.raises(ValueError):                    # the exact-f32 mode has no lazy form
        d3 = f3.to(dev)
        hip.call("ophip_coarse_match", hip.ptr(d3), hip.ptr(f2.to(dev)), hip.ptr(kp.to(dev)), 0, 1, N, M, wc, 0.08, 0.1, 2, 8.0, None,
                 hip.ptr(torch.empty(hip.load().ophip_coarse_workspace_floats(1, N, M), device=dev)),
                 *[hip.ptr(torch.empty(N, dtype=torch.int64, device=dev), torch.int64) for _ in range(3)],
                 hip.ptr(torch.empty(N, device=dev)), hip.ptr(torch.empty(N, 3, device=dev)), hip.ptr(torch.empty(N, 2, device=dev)), None, None,
                 hip.ptr(torch.zeros(4, dtype=torch.int32, device=dev), torch.int32), 0, hip.stream_handle())


def test_custom_ops_match_the_c_abi(sd, cfg, dev):
    """torch.ops.onepose_hip.* (onepose_st_amd/ops.py) forward to the same C symbols the model runs by default: the eight-wave
    encoder layer (plain and fragment-writing form), coarse matching and the fused fine stage through the ops equal the direct
    C-ABI calls bit for bit"""
    from onepose_st_amd import ops  # noqa: F401  (registers the library)
    g = torch.Generator().manual_seed(3)
    B, L3, hc, wc = 1, 130, 8, 12
    L2 = hc * wc
    x3, x2 = torch.randn(B, L3, 256, generator=g).to(dev), torch.randn(B, L2, 256, generator=g).to(dev)
    w = packing.pack_coarse_layer_x3w8(sd, "loftr_coarse.layers.0.").to(dev)
    ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
    y3, y2 = torch.empty_like(x3), torch.empty_like(x2)
    torch.ops.onepose_hip.encoder_layer_x3w8(x3, x2, y3, y2, w, None, False, False, 0, ws)
    z3, z2 = torch.empty_like(x3), torch.empty_like(x2)
    hip.call("ophip_encoder_layer_x3w8", hip.ptr(x3), hip.ptr(x2), hip.ptr(z3), hip.ptr(z2), B, L3, L2, hip.ptr(w, None), None, 0, 0, 0,
             hip.ptr(ws, None), hip.stream_handle())
    assert torch.equal(y3, z3) and torch.equal(y2, z2)
    # the fragment-writing form: same rows, and coarse matching fed from the planes gives the same matches
    cws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, L3, L2), device=dev)
    f3, f2 = torch.empty_like(x3), torch.empty_like(x2)
    torch.ops.onepose_hip.encoder_layer_x3w8_frag(x3, x2, f3, f2, w, None, False, False, 0, ws, cws)
    assert torch.equal(f3, z3) and torch.equal(f2, z2)
    kp = torch.randn(B, L3, 3, generator=g).to(dev)
    q2 = y2.clone()
    q2[0, 30:70] = y3[0, :40]                                   # plant matches away from the top / left border
    conf, b_ids, i_ids, j_ids, mconf, mk3, mkc, count = torch.ops.onepose_hip.coarse_match(y3, q2, kp, wc, 0.08, 0.1, 0, 8.0, 3)
    K = int(count.item())
    assert K >= 30 and conf.shape == (B, L3, L2) and torch.all(i_ids[:K][1:] > i_ids[:K][:-1])
    ref = _coarse_match(dev, y3.cpu(), q2.cpu(), kp.cpu(), wc, border=0, nsplit=3)
    assert torch.equal(conf, ref[0]) and torch.equal(i_ids[:K], ref[1][1]) and torch.equal(j_ids[:K], ref[1][2]) and torch.equal(mconf[:K], ref[2])
    # fused fine stage: gather + 2 fine layers + correlation + soft-argmax
    hf, wf = 4 * hc, 4 * wc
    ff = torch.randn(B, 128, hf, wf, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dd = torch.randn(B, 128, L3, generator=g).to(dev)
    wf_ = packing.pack_fine_layers_bf16(sd, "loftr_fine.layers.", 2).to(dev)
    expec, mkf = torch.ops.onepose_hip.fine_refine_bf16(ff, dd, b_ids, i_ids, j_ids, count, mkc, wf_, 2, 2, True, 3, wc, 4, 4.0)
    e2, m2 = torch.full_like(expec, float("nan")), torch.full_like(mkf, float("nan"))
    cap = b_ids.numel()
    hip.call("ophip_fine_refine_bf16", hip.ptr(ff), ff.stride(0), ff.stride(1), ff.stride(2), ff.stride(3), hf, wf, hip.ptr(dd), dd.stride(0), dd.stride(1),
             hip.ptr(b_ids, torch.int64), hip.ptr(i_ids, torch.int64), hip.ptr(j_ids, torch.int64), hip.ptr(count, torch.int32), cap, hip.ptr(mkc),
             hip.ptr(wf_, None), 2, ctypes.c_uint(2), 1, 3, wc, 4, 4.0, hip.ptr(e2), hip.ptr(m2), None, None, hip.stream_handle())
    assert torch.equal(expec[:K], e2[:K]) and torch.equal(mkf[:K], m2[:K]) and torch.isfinite(mkf[:K]).all()
    with pytest.raises(ValueError):
        torch.ops.onepose_hip.encoder_layer_x3w8(x3, x2, y3, y2, w, None, False, False, 0, ws[:64])
    with pytest.raises((ValueError, RuntimeError, NotImplementedError)):          # no CPU implementation behind the ops
        torch.ops.onepose_hip.encoder_layer_x3w8(x3.cpu(), x2.cpu(), y3.cpu(), y2.cpu(), w.cpu(), None, False, False, 0, ws.cpu())


def test_model_default_path_dispatches_through_the_frame_op(sd, cfg, dev):
    """OnePosePlus_model's default (bf16x3) forward goes through torch.ops.onepose_hip.frame_enqueue: one custom-op call per frame"""
    from onepose_st_amd import ops
    m = _model(sd, cfg, dev, "bf16x3")
    inp = make_synthetic_inputs(sd, n_points=300, image_hw=(64, 96), n_plant=90, seed=4, config=cfg)
    n_plans, n_calls = len(ops._frame_plans), ops.CALLS["frame_enqueue"]
    for _ in range(3):
        data = {k: inp[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
        m.forward_features(data, inp["feat_c"].to(dev), inp["feat_f"].to(dev), inp["image_hw"])
        assert data["i_ids"].numel() > 20
    assert len(ops._frame_plans) == n_plans + 1               # one plan per (model, input shape), registered once
    assert ops.CALLS["frame_enqueue"] == n_calls + 3           # one op call per frame


@pytest.mark.gpu
def test_more_than_eight_input_shapes_through_one_model(sd, cfg, dev):
    """A model keeps at most eight frame plans and drops the oldest; plan ids are never reused (ops.register_frame_plan), so cycling ten
    shapes and coming back to the first and to the one registered eighth gives the same bits as a fresh model on each shape."""
    from onepose_st_amd import ops
    m = _model(sd, cfg, dev, "bf16x3")
    shapes = [(200 + 16 * k, (64, 96 + 8 * k)) for k in range(10)]
    inputs = [make_synthetic_inputs(sd, n_points=n, image_hw=hw, n_plant=60, seed=20 + k, config=cfg) for k, (n, hw) in enumerate(shapes)]
    first = [_run_features(m, inp, dev) for inp in inputs]
    assert len(m._frame_plans) == 8 and len(m._plan_ids) == 8 and m._plan_ids <= set(ops._frame_plans)
    for k in (0, 7, 9, 1):                                       # 0 and 1 were dropped and come back under new ids; 7 and 9 are still resident
        again = _run_features(m, inputs[k], dev)
        fresh = _run_features(_model(sd, cfg, dev, "bf16x3"), inputs[k], dev)
        for key in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f"):
            assert torch.equal(again[key], first[k][key]) and torch.equal(again[key], fresh[key]), (k, key)
        assert again["i_ids"].numel() > 20


@pytest.mark.gpu
def test_a_failed_enqueue_leaves_the_pipeline_usable(sd, cfg, dev):
    """ophip_frame_enqueue fails in the middle of a frame (a NULL weight block for coarse layer 3: the input kernels and three encoder
    layers are already queued): the call returns the error, the slot goes back clean, and twenty more frames through the same streams
    -- pipelined, three in flight -- give the bits of the frames before the failure."""
    from onepose_st_amd import ops
    m = _model(sd, cfg, dev, "bf16x3")
    inp = make_synthetic_inputs(sd, n_points=500, image_hw=(96, 128), n_plant=150, seed=31, config=cfg)
    d = to_dev(inp, dev)
    obj = {k: d[k] for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    want = _run_features(m, inp, dev)
    pend = [m.enqueue_features(dict(obj), d["feat_c"], d["feat_f"], inp["image_hw"]) for _ in range(2)]      # two frames in flight
    (plan,) = [p for p in m._frame_plans.values()]
    desc = plan[0]
    saved = desc.w_coarse[3]
    desc.w_coarse[3] = None
    with pytest.raises(ValueError):
        m.enqueue_features(dict(obj), d["feat_c"], d["feat_f"], inp["image_hw"])
    desc.w_coarse[3] = saved
    for _ in range(20):
        pend.append(m.enqueue_features(dict(obj), d["feat_c"], d["feat_f"], inp["image_hw"]))
        if len(pend) >= 3:
            got = pend.pop(0).finish()
            for key in ("i_ids", "j_ids", "mconf", "mkpts_query_f"):
                assert torch.equal(got[key], want[key]), key
    while pend:
        got = pend.pop(0).finish()
        assert torch.equal(got["i_ids"], want["i_ids"]) and torch.equal(got["mkpts_query_f"], want["mkpts_query_f"])


def _hard_inputs(sd, cfg, size):
    from onepose_st_amd.synthetic import CONFIG_SIZES, HARD_PROFILE
    n, hw, plant = CONFIG_SIZES[size]
    return make_synthetic_inputs(sd, n_points=n, image_hw=hw, n_plant=plant, seed=1, config=cfg, **HARD_PROFILE)


@pytest.mark.gpu
def test_c1_hard_against_reference_golden(model, sd, cfg, dev, golden_dir):
    """Low-margin / outlier frame (synthetic.HARD_PROFILE) against the fixture the REFERENCE produced (make_golden.py --only-hard): its
    confidences cover (0, 1) with 65 row maxima in (0.05, 0.3) and 39 % of its 303 matches are geometrically wrong.  Same bar as c1:
    indices bit-exact, at most a threshold-window match set aside and counted; the pose from the HIP matches equals the pose from the
    reference's matches (own RANSAC on both; 118 outliers to reject)."""
    g = np.load(os.path.join(golden_dir, "c1_hard_feature_boundary.npz"))
    inp = _hard_inputs(sd, cfg, "c1_hard")
    data = _run_features(model, inp, dev)
    n_sa = _check_against(data, g, model.precision, label=f"c1_hard_{model.precision}", want_rowmax=g["conf_rowmax"])
    if model.precision == "f32":
        assert n_sa == 0
    rt_conf = TOL[model.precision][2]
    np.testing.assert_allclose(data["conf_matrix"].max(dim=2)[0][0].cpu().numpy(), g["conf_rowmax"], rtol=rt_conf, atol=1e-6)
    _pose_parity(data, g["mkpts_3d_db"], g["mkpts_query_f"], inp, f"c1_hard {model.precision} vs reference golden", n_sa)


@pytest.mark.gpu
def test_c2_hard_full_size_against_oracle(model, sd, cfg, dev):
    """The same profile at BASELINE config 2's size (7000 x 4800), against the CPU oracle: ~1 500 matches whose confidences spread over
    (0, 1), hundreds of row maxima around the threshold, ~40 % outliers for the pose.  Reports (pytest's last line and stdout) how many
    matches sat inside the threshold window and were set aside."""
    inp = _hard_inputs(sd, cfg, "c2_hard")
    data = _run_features(model, inp, dev)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = orc.forward_from_features(sd, cfg, inp, inp["feat_c"], inp["feat_f"], inp["image_hw"])
    rowmax = ref["conf_matrix"].max(dim=2)[0][0].numpy()
    K = len(ref["i_ids"])
    planted = set(zip(inp["planted_i"].tolist(), inp["planted_j"].tolist()))
    wrong = 1.0 - sum(p in planted for p in zip(ref["i_ids"].tolist(), ref["j_ids"].tolist())) / max(K, 1)
    near = int(((rowmax > 0.05) & (rowmax < 0.3)).sum())
    print(f"c2_hard: K = {K}, {100 * wrong:.0f} % of the oracle's matches are wrong, {near} row maxima in (0.05, 0.3), "
          f"min |mconf - thr| = {float(np.abs(ref['mconf'].numpy() - 0.1).min()):.2e}")
    assert K > 800 and 0.25 <= wrong <= 0.55 and near >= 200
    n_sa = _check_against(data, {k: ref[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mconf", "mkpts_3d_db", "mkpts_query_c",
                                                             "mkpts_query_f", "expec_f")}, model.precision, label=f"c2_hard_{model.precision}",
                          want_rowmax=rowmax)
    _pose_parity(data, ref["mkpts_3d_db"].numpy(), ref["mkpts_query_f"].numpy(), inp, f"c2_hard {model.precision} vs oracle", n_sa)
    conf, i, j = data["conf_matrix"][0], data["i_ids"], data["j_ids"]
    v = conf[i, j]
    assert torch.equal(v, data["mconf"]) and bool((v > 0.1).all())
    assert torch.equal(v, conf.max(dim=1)[0][i]) and torch.equal(v, conf.max(dim=0)[0][j])          # mutual nearest, on the device's own matrix


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c1", "c1_hard", "c2", "c2_hard", "b2"])
def test_fine_stage_in_plain_bf16_keeps_indices_and_pose(sd, cfg, dev, golden_dir, monkeypatch, case):
    """`hip_fine_precision = "bf16"` / OPHIP_FINE_PRECISION=bf16: the fine stage on plain bf16 operands (one matrix instruction per product), the
    coarse stage unchanged.  What north_star names must hold exactly as in the default mode -- match indices bit-exact, pose R|t within 1e-4 of
    the pose from the reference's (golden) / the oracle's matches -- while the sub-pixel keypoints are allowed 0.08 px (measured: <= 0.05) instead
    of the default mode's 1e-4 relative.  The frames are the reference-generated goldens (c1, c1_hard) and the full-size oracle cases (c2, c2_hard)."""
    monkeypatch.setenv("OPHIP_FINE_PRECISION", "bf16")
    m = _model(sd, cfg, dev, "bf16x3")
    if case == "b2":                              # the ragged two-frame batch of the reference goldens: indices and keypoints (no pose: two frames, one list)
        g = np.load(os.path.join(golden_dir, "b2_ragged_feature_boundary.npz"))
        i0 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=0)
        i1 = make_synthetic_inputs(sd, n_points=333, image_hw=(96, 136), n_plant=120, seed=3, config=cfg, frame=1)
        both = {k: torch.cat([i0[k], i1[k]], 0) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db", "feat_c", "feat_f")}
        both["image_hw"] = i0["image_hw"]
        data = _run_features(m, both, dev)
        assert _check_against(data, g, "bf16x3", label="b2_fine_bf16", kp_tol=(0.0, 0.08)) == 0
        return
    if case in ("c1", "c1_hard"):
        g = dict(np.load(os.path.join(golden_dir, f"{case}_feature_boundary.npz")))
        inp = (make_synthetic_inputs(sd, n_points=1000, image_hw=(240, 320), n_plant=600, seed=1, config=cfg) if case == "c1" else _hard_inputs(sd, cfg, "c1_hard"))
        rowmax = g["conf_rowmax"]
    else:
        inp = (make_synthetic_inputs(sd, n_points=7000, image_hw=(480, 640), n_plant=3000, seed=1, config=cfg) if case == "c2" else _hard_inputs(sd, cfg, "c2_hard"))
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        with torch.no_grad():
            ref = orc.forward_from_features(sd, cfg, inp, inp["feat_c"], inp["feat_f"], inp["image_hw"])
        g = {k: ref[k].numpy() for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mconf", "mkpts_3d_db", "mkpts_query_c", "mkpts_query_f", "expec_f")}
        rowmax = ref["conf_matrix"].max(dim=2)[0][0].numpy()
    data = _run_features(m, inp, dev)
    n_sa = _check_against(data, g, "bf16x3", label=f"{case}_fine_bf16", want_rowmax=rowmax, kp_tol=(0.0, 0.08))
    err = float(np.abs(data["mkpts_query_f"].cpu().numpy() - np.asarray(g["mkpts_query_f"])[:len(data["mkpts_query_f"])]).max()) if n_sa == 0 else float("nan")
    print(f"[{case}, fine stage in plain bf16] K = {len(data['i_ids'])}, max |mkpts_query_f - reference| = {err:.3e} px")
    _pose_parity(data, g["mkpts_3d_db"], g["mkpts_query_f"], inp, f"{case}, fine stage in plain bf16", n_sa)
    monkeypatch.delenv("OPHIP_FINE_PRECISION")
    again = _run_features(m, inp, dev)                                   # the knob is read per call: the default form is back
    assert float(np.abs(again["mkpts_query_f"].cpu().numpy() - np.asarray(g["mkpts_query_f"])[:len(again["mkpts_query_f"])]).max()) < 2e-3 or n_sa


@pytest.mark.gpu
def test_profiler_and_roctx_ranges_on_the_one_call_path(sd, cfg, dev):
    """SURVEY section 5's hook on the DEFAULT path: a profiler object no longer takes the model off the one-call frame path, its
    record_function sees the reference's two scope names (coarse_matching.py:122,167) once per frame, and with the roctx hook enabled
    the library opens ranges around the frame's stages and kernels -- results bit-identical to the unprofiled frame."""
    import contextlib
    from onepose_st_amd import ops
    from onepose_st_amd.model import OnePosePlus_model

    class Recorder:
        def __init__(self):
            self.names = []

        @contextlib.contextmanager
        def record_function(self, name):
            self.names.append(name)
            yield

    inp = make_synthetic_inputs(sd, n_points=300, image_hw=(64, 96), n_plant=90, seed=4, config=cfg)
    plain = _run_features(_model(sd, cfg, dev, "bf16x3"), inp, dev)
    rec = Recorder()
    c2 = dict(cfg)
    c2["hip_precision"] = "bf16x3"
    m = OnePosePlus_model(c2, profiler=rec).eval()
    m.load_state_dict(sd, strict=True)
    m.to(dev)
    lib = hip.load()
    n_calls, n_ranges = ops.CALLS["frame_enqueue"], lib.ophip_roctx_ranges()
    hip.call("ophip_roctx_enable", 1)
    try:
        got = _run_features(m, inp, dev)
    finally:
        hip.call("ophip_roctx_enable", 0)
    assert ops.CALLS["frame_enqueue"] == n_calls + 1                      # still ONE op call for the frame
    assert rec.names == ["LoFTR/coarse-matching/get_coarse_match", "LoFTR/coarse-matching/get_coarse_match/argmax-conf"]
    assert lib.ophip_roctx_ranges() - n_ranges >= 20                      # frame + 3 stages + every kernel launch of the frame
    for key in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f"):
        assert torch.equal(got[key], plain[key]), key
    n_off = lib.ophip_roctx_ranges()
    _run_features(m, inp, dev)
    assert lib.ophip_roctx_ranges() == n_off                              # off again: no range is opened
