"""Randomised parity sweep on the GPU (not part of the test suite: run by hand, `python tools/fuzz_parity.py [seed] [cases]`).

Feature-boundary frames of random ragged sizes against the CPU oracle (indices bit-exact, floats within the bf16x3 tolerances
of tests/test_gpu_parity.py) and random convolution shapes / epilogues against F.conv2d."""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict
from oracle import onepose_oracle as orc
from tests import test_gpu_backbone as tb

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = random.Random(seed)
dev = torch.device("cuda:0")
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg)
model = OnePosePlus_model(cfg).eval(); model.load_state_dict(sd); model.to(dev)
torch.set_num_threads(8)
bad = 0
for c in range(cases):
    hc, wc = rng.randint(6, 24), rng.randint(6, 28)
    N = rng.randint(40, 900)
    B = rng.choice([1, 1, 2, 3])
    frames = [make_synthetic_inputs(sd, n_points=N, image_hw=(8 * hc, 8 * wc), n_plant=rng.randint(0, min(N, hc * wc) // 2), seed=seed * 100 + c, config=cfg, frame=f) for f in range(B)]
    inp = {k: torch.cat([f[k] for f in frames]) for k in ("feat_c", "feat_f")}
    obj = {k: frames[0][k].expand(B, *frames[0][k].shape[1:]).contiguous() for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    extra = {}
    if rng.random() < 0.5:          # padded / resized query images: a bottom / right padding band per element and per-image scales
        mask = torch.ones(B, hc, wc, dtype=torch.bool)
        for b in range(B):
            if rng.random() < 0.5:
                mask[b, hc - rng.randint(1, hc // 3):, :] = False
            else:
                mask[b, :, wc - rng.randint(1, wc // 3):] = False
        extra["query_image_mask"] = mask
        if rng.random() < 0.7:
            extra["query_image_scale"] = torch.tensor([[rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0)] for _ in range(B)], dtype=torch.float32)
    obj.update(extra)
    with torch.no_grad():
        ref = orc.forward_from_features(sd, cfg, obj, inp["feat_c"], inp["feat_f"], (8 * hc, 8 * wc))
    data = {k: v.to(dev) for k, v in obj.items()}
    model.forward_features(data, inp["feat_c"].to(dev), inp["feat_f"].to(dev), (8 * hc, 8 * wc))
    ok = all(torch.equal(data[k].cpu(), ref[k]) for k in ("b_ids", "i_ids", "j_ids"))
    K = len(ref["i_ids"])
    if ok and K:
        ok = bool(torch.allclose(data["mkpts_query_f"].cpu(), ref["mkpts_query_f"], rtol=1e-4, atol=5e-4)) and \
             bool(torch.allclose(data["mconf"].cpu(), ref["mconf"], rtol=5e-4, atol=1e-6))
    if ok and K and "query_image_scale" in extra:
        ok = bool(torch.equal(data["mkpts_query_c"].cpu(), ref["mkpts_query_c"]))
    print(f"frame case {c}: B={B} N={N} grid {hc}x{wc} K={K} {'masked ' if extra else ''}{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
for c in range(cases * 2):
    cin, cout = rng.choice([8, 32, 40, 100, 128, 196, 256]), rng.choice([8, 36, 64, 128, 196, 256])
    ks, stride = rng.choice([1, 3]), rng.choice([1, 2])
    H, W, B = rng.randint(3, 40), rng.randint(3, 70), rng.choice([1, 2])
    act = rng.choice([0, 1, 2])
    g = torch.Generator().manual_seed(seed * 1000 + c)
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).to(dev)
    bias = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(B, cout, Ho, Wo, generator=g).to(dev) if rng.random() < 0.4 else None
    up = torch.randn(B, cout, max(1, Ho // 2), max(1, Wo // 2), generator=g).to(dev) if rng.random() < 0.3 else None
    got_p, got_f = tb.run_conv(dev, x, w, bias, stride, act, res, up, None, 3)
    ref = tb.torch_conv(x, w, bias, stride, act, res, up, None)
    err = float((got_f - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
    ok = err <= 3e-5
    print(f"conv case {c}: {cin}->{cout} k{ks} s{stride} {H}x{W} B={B} act={act} res={res is not None} up={up is not None} err {err:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
