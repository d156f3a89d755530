"""The coarse encoder's slab sum inside the consuming launch (csrc/encoder_x3w8.hip ``fused_kv_sum``, layers 1 .. n - 1 of a chain whose
workgroups all fit the chip) against the separate ``kv_sum`` launch it replaces (``OPHIP_ENC_FUSED_KVSUM=0``): the sums are taken in the
same association, so every output of a frame must be bit-identical, at a size that uses 246 of the 256 CUs (c2), at a small size, with a
padded query image, and for a batch that does NOT fit (the fused form must then not be used).  Each variant runs in its own process (the
switch is read once per process).  Reference of what is computed: loftr_module/linear_attention.py:49-57 (the K^T V / K sums over the
source tokens)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CODE = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
from onepose_st_amd import hip
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict
dev = torch.device("cuda:0")
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg)
m = OnePosePlus_model(cfg).eval(); m.load_state_dict(sd); m.to(dev)
out = {}
for name, n, hw, pl, B, masked in (("c2", 7000, (480, 640), 3000, 1, False), ("small", 600, (96, 136), 200, 1, False),
                                   ("small_b3", 600, (96, 136), 200, 3, False), ("small_masked", 600, (96, 136), 200, 1, True),
                                   ("b8", 1500, (160, 224), 500, 8, False)):
    inp = make_synthetic_inputs(sd, n, hw, pl, seed=3, config=cfg)
    data = {k: inp[k].to(dev).expand(B, *inp[k].shape[1:]) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    fc, ff = inp["feat_c"].to(dev).expand(B, -1, -1, -1).contiguous(), inp["feat_f"].to(dev).expand(B, -1, -1, -1).contiguous()
    if masked:
        qm = torch.ones(B, fc.shape[2], fc.shape[3], dtype=torch.bool, device=dev); qm[:, :, -3:] = False
        data["query_image_mask"] = qm
    for rep in range(2):
        d = dict(data)
        m.forward_features(d, fc, ff, hw)
    h = hashlib.sha256()
    for k in ("i_ids", "j_ids", "mconf", "mkpts_query_f", "expec_f", "conf_matrix"):
        h.update(d[k].cpu().numpy().tobytes())
    out[name] = [h.hexdigest(), int(d["i_ids"].numel())]
out["timeouts"] = hip.load().ophip_encoder_sync_timeouts()
print(json.dumps(out))
""" % REPO


@pytest.mark.gpu
def test_fused_slab_sum_is_bit_identical_to_the_kv_sum_launch():
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, OPHIP_ENC_FUSED_KVSUM=flag)
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["1"]["timeouts"] == 0 and res["0"]["timeouts"] == 0
    for name in ("c2", "small", "small_b3", "small_masked", "b8"):
        assert res["1"][name] == res["0"][name], name
        assert res["1"][name][1] > 50, (name, res["1"][name])          # the frames do produce matches
