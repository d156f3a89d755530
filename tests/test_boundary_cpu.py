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


def test_fine_precision_key_is_validated_and_process_wide(cfg, monkeypatch):
    """config["hip_fine_precision"]: "bf16x3" | "bf16" (the fine stage alone on plain bf16 operands); the library reads OPHIP_FINE_PRECISION on every
    fine-stage launch, so giving the key sets it for the process, leaving it out touches nothing."""
    import copy
    import os
    monkeypatch.delenv("OPHIP_FINE_PRECISION", raising=False)
    OnePosePlus_model(copy.deepcopy(cfg))
    assert "OPHIP_FINE_PRECISION" not in os.environ
    c = copy.deepcopy(cfg)
    c["hip_fine_precision"] = "fp8"
    with pytest.raises(ValueError):
        OnePosePlus_model(c)
    c["hip_fine_precision"] = "bf16"
    monkeypatch.setenv("OPHIP_FINE_PRECISION", "bf16x3")          # (so that monkeypatch restores the variable after the test)
    OnePosePlus_model(c)
    assert os.environ["OPHIP_FINE_PRECISION"] == "bf16"


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "onepose_hip.h")).read()
    declared = set(re.findall(r"\b(ophip_\w+)\s*\(", header))
    assert declared == set(hip.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(hip.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().ophip_abi_version() == hip.ABI_VERSION == 4
    header = open(os.path.join(REPO, "include", "onepose_hip.h")).read()
    assert "#define OPHIP_ABI_VERSION 4" in header          # header, library and binding carry the same number
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


def test_frame_plan_ids_are_never_reused():
    """ops.register_frame_plan: a model drops its oldest plan once it holds more than eight input shapes; the id of a dropped plan must
    never come back while other plans are live (it used to be len(dict) + 1: register 8, drop #1, register -> a second #8 that replaced
    the live plan 8, and the next frame of that shape was launched with another shape's sizes and offsets)."""
    import gc
    from onepose_st_amd import ops
    made = []
    for k in range(8):
        d, L = hip.FrameDesc(), hip.FrameLayout()
        d.N = 100 + k
        made.append((ops.register_frame_plan(d, L, keep_alive=(k,)), d))
    ops.drop_frame_plan(made[0][0])
    d9, L9 = hip.FrameDesc(), hip.FrameLayout()
    d9.N = 999
    pid9 = ops.register_frame_plan(d9, L9)
    ids = [p for p, _ in made] + [pid9]
    assert len(set(ids)) == 9 and pid9 > max(p for p, _ in made)
    for pid, d in made[1:]:
        assert ops._frame_plans[pid][0] is d and ops._frame_plans[pid][0].N == d.N        # plan 8 (and every other live plan) untouched
    assert made[0][0] not in ops._frame_plans
    with pytest.raises(ValueError):                                                        # a stale id raises, it cannot launch anything
        ops._frame_enqueue(made[0][0], None, None, None, [0, 0, 0, 0], None, None, None, None, None, 0, 0, 0, 0, 0)
    ops.drop_frame_plans(ids)
    # a collected model releases its plans (and with them the packed weight blocks the registry kept alive)
    m = OnePosePlus_model(default_config())
    keep = m._plan_ids
    pid = ops.register_frame_plan(hip.FrameDesc(), hip.FrameLayout(), keep_alive=("weights",))
    keep.add(pid)
    del m
    gc.collect()
    assert pid not in ops._frame_plans


def test_frame_layout_is_aligned_and_disjoint():
    """ophip_frame_layout (host arithmetic only, runs without a GPU): the one-call frame path carves every intermediate and output of a
    frame out of one device block -- regions 256-byte aligned, pairwise disjoint, inside `total`; optional regions absent when not asked for."""
    import ctypes
    d = hip.FrameDesc()
    d.B, d.N, d.M, d.hc, d.wc, d.hf, d.wf, d.cf = 2, 1000, 1200, 30, 40, 120, 160, 128
    d.n_coarse, d.n_fine = 6, 2
    sizes = {"x2d": 2 * 1200 * 256 * 4, "ffcl": 2 * 120 * 160 * 128 * 4, "x3d": 2 * 1000 * 256 * 4, "y3d": 2 * 1000 * 256 * 4, "y2d": 2 * 1200 * 256 * 4,
             "z3d": 2 * 1000 * 256 * 4, "conf": 2 * 1000 * 1200 * 4, "result": 16 + 28 * 2000, "i_ids": 16000, "j_ids": 16000, "m_bids": 16000,
             "gt_mask": 2000, "mconf": 8000, "mkc": 16000, "expec": 24000, "stats": 48}
    for transpose_fine, external in ((1, 0), (0, 1), (0, 0)):
        L = hip.FrameLayout()
        hip.call("ophip_frame_layout", ctypes.byref(d), transpose_fine, external, ctypes.byref(L))
        assert L.result_bytes == 16 + 28 * 2000
        assert (L.ffcl != 0) == bool(transpose_fine) and (L.z3d != 0) == bool(external)
        assert (L.x3d == 0) == bool(external) or L.x2d == 0          # x2d is the first region (offset 0)
        regions = []
        for name, nbytes in sizes.items():
            off = getattr(L, name)
            if name in ("ffcl", "z3d") and off == 0:
                continue
            if name == "x3d" and external:
                continue
            assert off % 256 == 0 and off + nbytes <= L.total, name
            regions.append((off, off + nbytes, name))
        for name in ("enc_ws", "cws"):
            regions.append((getattr(L, name), getattr(L, name) + 1, name))
        regions.sort()
        for (a0, a1, an), (b0, b1, bn) in zip(regions, regions[1:]):
            assert a1 <= b0, (an, bn)
    d.M = 1201                                                      # hc * wc != M
    with pytest.raises(ValueError):
        hip.call("ophip_frame_layout", ctypes.byref(d), 0, 0, ctypes.byref(hip.FrameLayout()))


def test_packed_weights_follow_the_parameters_without_walking_the_module_tree_every_frame(sd, cfg):
    """model._weights: the device-side weight blocks are re-packed when a parameter changes; the per-frame check reads (data_ptr, _version) of
    a cached parameter list (the module-tree walk cost a third of an enqueue's host time), rebuilt after ``load_state_dict`` / ``_apply``,
    every ``_PARAM_REWALK`` frames, and as soon as a Parameter object or a sub-module of the matcher was replaced (identity check of the edges)."""
    m = OnePosePlus_model(cfg).eval()
    m.load_state_dict(sd)
    cpu = torch.device("cpu")
    w0 = m._weights(cpu)
    walks = []
    real = m.named_parameters
    m.named_parameters = lambda *a, **k: (walks.append(1), real(*a, **k))[1]
    for _ in range(10):
        assert m._weights(cpu) is w0                                   # unchanged model: the cached blocks ...
    assert len(walks) == 0                                             # ... and no walk of the module tree
    p = m.loftr_coarse.layers[0].q_proj.weight
    with torch.no_grad():
        p.mul_(2.0)                                                    # in place: seen on the next frame through the cached list
    w1 = m._weights(cpu)
    assert w1 is not w0 and not torch.equal(w1["coarse_x3"][0], w0["coarse_x3"][0])
    assert m._weights(cpu) is w1
    m.load_state_dict(sd)                                              # copies in place AND drops the cached list
    w2 = m._weights(cpu)
    assert w2 is not w1 and torch.equal(w2["coarse_x3"][0], w0["coarse_x3"][0])
    m.loftr_coarse.layers[0].q_proj.weight = torch.nn.Parameter(p.detach() * 3.0)       # module surgery: a replaced Parameter object ...
    w3 = m._weights(cpu)
    assert w3 is not w2 and not torch.equal(w3["coarse_x3"][0], w2["coarse_x3"][0])      # ... is seen on the very next frame (identity of the tree's edges)
    import copy
    new_layer = copy.deepcopy(m.loftr_coarse.layers[1])
    with torch.no_grad():
        new_layer.merge.weight.mul_(0.5)
    m.loftr_coarse.layers[1] = new_layer                                                 # a replaced sub-module as well
    w4 = m._weights(cpu)
    assert w4 is not w3 and not torch.equal(w4["coarse_x3"][1], w3["coarse_x3"][1])
    assert m._weights(cpu) is w4
    m.double()                                                         # _apply drops the list as well
    assert m.__dict__["_param_cache"] is None
