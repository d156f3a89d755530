"""Host-side boundary checks that need no GPU: state_dict contract, config errors, C-ABI exports,
and that the product path refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from onepose_st_amd import hip
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_state_dict_contract(sd, cfg):
    """195 tensors, 10 226 480 parameters, strict load both ways."""
    m = OnePosePlus_model(cfg)
    own = m.state_dict()
    assert set(own) == set(sd)
    assert len(own) == 195
    for k, v in own.items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd, strict=True)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 10226480
    assert "dense_pos_encoding.pe" not in own          # non-persistent buffer in the reference


@pytest.mark.parametrize("path,value,exc", [
    (("loftr_backbone", "type"), "VGG", ValueError),
    (("loftr_backbone", "resolution"), [16, 4], NotImplementedError),
    (("keypoints_encoding", "type"), "mlp_conv", NotImplementedError),
    (("coarse_matching", "type"), "sinkhorn", NotImplementedError),
    (("loftr_coarse", "layer_names"), ["self", "other"], NotImplementedError),
    (("loftr_coarse", "type"), "foo", ValueError),
    (("fine_matching", "s2d"), {"type": "regress"}, NotImplementedError),
])
def test_unsupported_config_raises_like_reference(path, value, exc):
    cfg = default_config()
    d = cfg
    for k in path[:-1]:
        d = d[k]
    d[path[-1]] = value
    with pytest.raises(exc):
        OnePosePlus_model(cfg)


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "onepose_hip.h")).read()
    declared = set(re.findall(r"\b(ophip_\w+)\s*\(", header))
    assert declared == set(hip.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(hip.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().ophip_abi_version() == 1
    assert hip.load().ophip_encoder_workspace_floats(1, 7000, 4800) == (219 + 150 + 2) * 8448


def test_no_cpu_fallback(sd, cfg):
    m = OnePosePlus_model(cfg).eval()
    m.load_state_dict(sd)
    data = {"keypoints3d": torch.zeros(1, 40, 3), "descriptors3d_db": torch.zeros(1, 128, 40),
            "descriptors3d_coarse_db": torch.zeros(1, 256, 40)}
    with pytest.raises(hip.HipLibraryError):
        m.forward_features(data, torch.zeros(1, 256, 4, 4), torch.zeros(1, 128, 16, 16), (32, 32))
    m.train()
    with pytest.raises(NotImplementedError):
        m({"query_image": torch.zeros(1, 1, 32, 32)})


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(REPO, "onepose_st_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), os.path.join(root, f)


def test_custom_ops_are_registered_without_a_cpu_implementation():
    """north_star: the kernels as PyTorch custom ops -- torch.ops.onepose_hip.* exist with the documented schemas and, like the
    rest of the product path, have no CPU implementation"""
    import pytest
    import torch
    from onepose_st_amd import ops
    for name in ops.OPS:
        op = getattr(torch.ops.onepose_hip, name)
        assert str(op.default._schema).startswith(f"onepose_hip::{name}(")
    with pytest.raises(NotImplementedError):
        torch.ops.onepose_hip.pe_add_transpose(torch.zeros(1, 256, 2, 2), None, torch.zeros(1, 4, 256))
    with pytest.raises(NotImplementedError):
        torch.ops.onepose_hip.coarse_match(torch.zeros(1, 8, 256), torch.zeros(1, 4, 256), torch.zeros(1, 8, 3), 2, 0.08, 0.1, 2, 8.0, 3)
