"""``torch.ops.onepose_hip.*``: the hot-path kernels registered as PyTorch custom ops (``torch.library``) over the C ABI.

SURVEY.md section 8b names this shape for the replacement (device tensors in, device tensors out, errors as Python
exceptions); the ops below are thin: each checks its tensors and forwards raw device pointers to ``libonepose_hip.so``
(``include/onepose_hip.h``) on the current HIP stream.  ``OnePosePlus_model`` itself calls the C ABI directly (same
symbols, no dispatcher hop on the per-frame path); these ops are for callers that compose the stages themselves, e.g. in
place of ``LocalFeatureTransformer.forward`` (``loftr_module/transformer.py:133-171``) or ``CoarseMatching.forward``
(``utils/coarse_matching.py:76-242``).  CUDA (= HIP on ROCm) dispatch key only: there is no CPU implementation.
"""
from __future__ import annotations

import torch

from . import hip

_lib = torch.library.Library("onepose_hip", "DEF")

_lib.define("pe_add_transpose(Tensor feat_nchw, Tensor? pe_nlc, Tensor(a!) out_nlc) -> ()")
_lib.define("kpt_encode(Tensor keypoints3d, Tensor desc_bcn, Tensor wpack, Tensor(a!) out_bnc) -> ()")
_lib.define("encoder_layer_x3(Tensor x3d, Tensor x2d, Tensor(a!) y3d, Tensor(b!) y2d, Tensor wpack, Tensor? wpack_next, "
            "bool is_cross, bool kv_from_prev, int slot, Tensor(c!) workspace) -> ()")
_lib.define("coarse_match(Tensor feat3d, Tensor feat2d, Tensor keypoints3d, int wc, float temperature, float thr, int border_rm, "
            "float scale, int nsplit) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor on the HIP device")
    return t


def _pe_add_transpose(feat_nchw, pe_nlc, out_nlc):
    B, C, h, w = feat_nchw.shape
    hip.call("ophip_pe_add_transpose", hip.ptr(_f32(feat_nchw, "feat_nchw")), hip.ptr(pe_nlc), hip.ptr(_f32(out_nlc, "out_nlc")), B, C, h * w,
             hip.stream_handle())


def _kpt_encode(keypoints3d, desc_bcn, wpack, out_bnc):
    B, N, _ = keypoints3d.shape
    stats = torch.empty(4 * B + 4, device=keypoints3d.device, dtype=torch.float32)
    hip.call("ophip_kpt_encode", hip.ptr(_f32(keypoints3d, "keypoints3d")), keypoints3d.stride(0), hip.ptr(_f32(desc_bcn, "desc_bcn")),
             desc_bcn.stride(0), hip.ptr(wpack), hip.ptr(stats), hip.ptr(_f32(out_bnc, "out_bnc")), B, N, hip.stream_handle())


def _encoder_layer_x3(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace):
    B, L3, _ = x3d.shape
    L2 = x2d.shape[1]
    need = hip.load().ophip_encoder_x3_workspace_bytes(B, L3, L2)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"workspace: {need} bytes needed (ophip_encoder_x3_workspace_bytes)")
    hip.call("ophip_encoder_layer_x3", hip.ptr(_f32(x3d, "x3d")), hip.ptr(_f32(x2d, "x2d")), hip.ptr(_f32(y3d, "y3d")), hip.ptr(_f32(y2d, "y2d")),
             B, L3, L2, hip.ptr(wpack, None), hip.ptr(wpack_next, None), int(is_cross), int(kv_from_prev), int(slot),
             hip.ptr(workspace, None), hip.stream_handle())


def _coarse_match(feat3d, feat2d, keypoints3d, wc, temperature, thr, border_rm, scale, nsplit):
    B, N, _ = feat3d.shape
    M = feat2d.shape[1]
    dev = feat3d.device
    cap = B * N
    conf = torch.empty(B, N, M, device=dev)
    ws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev)
    ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(3)]
    mconf, mk3, mkc = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    hip.call("ophip_coarse_match", hip.ptr(_f32(feat3d, "feat3d")), hip.ptr(_f32(feat2d, "feat2d")), hip.ptr(_f32(keypoints3d, "keypoints3d")),
             keypoints3d.stride(0), B, N, M, int(wc), float(temperature), float(thr), int(border_rm), float(scale), hip.ptr(conf), hip.ptr(ws),
             *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(count, torch.int32),
             int(nsplit), hip.stream_handle())
    return conf, ids[0], ids[1], ids[2], mconf, mk3, mkc, count


_lib.impl("pe_add_transpose", _pe_add_transpose, "CUDA")
_lib.impl("kpt_encode", _kpt_encode, "CUDA")
_lib.impl("encoder_layer_x3", _encoder_layer_x3, "CUDA")
_lib.impl("coarse_match", _coarse_match, "CUDA")

OPS = ("pe_add_transpose", "kpt_encode", "encoder_layer_x3", "coarse_match")
