"""``torch.ops.onepose_hip.*``: the hot-path kernels registered as PyTorch custom ops (``torch.library``) over the C ABI.

SURVEY.md section 8b names this shape for the replacement (device tensors in, device tensors out, errors as Python
exceptions).  Each op checks its tensors and forwards raw device pointers to ``libonepose_hip.so``
(``include/onepose_hip.h``) on the current HIP stream.  What the survey sketches as eight ops exists here as the kernels
that were actually built -- fused further than the sketch:

    survey name                         op here
    pe_add_transpose                    pe_add_transpose
    kpt_encode                          kpt_encode
    kv_reduce + attn_apply              encoder_layer_x3w8 / encoder_layer_x3w8_frag   (one LoFTREncoderLayer on both streams;
                                        the next layer's K/V reduce is fused into attn_apply's tail)
    coarse_match                        coarse_match
    fine_gather + fine_encoder +        fine_refine_bf16   (one kernel: window gather, 2 fine layers, correlation,
      fine_corr_expect                                      soft-argmax)
    --                                  frame_enqueue   (rows a1-a11 of one frame in ONE call: what ``OnePosePlus_model``
                                                         dispatches through on its default path)

CUDA (= HIP on ROCm) dispatch key only: there is no CPU implementation, a CPU tensor raises.
Reference: ``loftr_module/transformer.py:65-171``, ``utils/coarse_matching.py:76-242``, ``loftr_module/fine_preprocess.py:32-55``,
``utils/fine_matching.py:28-110``, ``OnePosePlusModel.py:115-203``.
"""
from __future__ import annotations

import ctypes
import itertools

import torch

from . import hip

_lib = torch.library.Library("onepose_hip", "DEF")

_lib.define("pe_add_transpose(Tensor feat_nchw, Tensor? pe_nlc, Tensor(a!) out_nlc) -> ()")
_lib.define("kpt_encode(Tensor keypoints3d, Tensor desc_bcn, Tensor wpack, Tensor(a!) out_bnc) -> ()")
_lib.define("encoder_layer_x3w8(Tensor x3d, Tensor x2d, Tensor(a!) y3d, Tensor(b!) y2d, Tensor wpack, Tensor? wpack_next, "
            "bool is_cross, bool kv_from_prev, int slot, Tensor(c!) workspace) -> ()")
_lib.define("encoder_layer_x3w8_frag(Tensor x3d, Tensor x2d, Tensor(a!) y3d, Tensor(b!) y2d, Tensor wpack, Tensor? wpack_next, "
            "bool is_cross, bool kv_from_prev, int slot, Tensor(c!) workspace, Tensor(d!) coarse_workspace) -> ()")
_lib.define("coarse_match(Tensor feat3d, Tensor feat2d, Tensor keypoints3d, int wc, float temperature, float thr, int border_rm, "
            "float scale, int nsplit) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("fine_refine_bf16(Tensor feat_f_cl, Tensor desc3d_f, Tensor b_ids, Tensor i_ids, Tensor j_ids, Tensor count, Tensor mkpts_c, "
            "Tensor wpack, int nlayers, int cross_bits, bool encoder_enable, int nsplit, int wc, int stride, float fine_scale) -> (Tensor, Tensor)")
_lib.define("frame_enqueue(int plan, Tensor(a!) block, Tensor feat_c, Tensor feat_f, int[] fine_strides, Tensor keypoints3d, "
            "Tensor desc3d_c, Tensor desc3d_f, Tensor? x3d_external, Tensor(b!) host_dst, int host_bytes, "
            "int s_main, int s_prep, int s_fine, int s_copy, Tensor? query_mask=None, Tensor? query_scale=None, "
            "Tensor? y3d0=None, Tensor? kv1=None, int object_ready=0) -> int")


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor on the HIP device")
    return t


def _bstride(t: torch.Tensor) -> int:
    return 0 if t.shape[0] == 1 or t.stride(0) == 0 else t.stride(0)


def _pe_add_transpose(feat_nchw, pe_nlc, out_nlc):
    B, C, h, w = feat_nchw.shape
    hip.call("ophip_pe_add_transpose", hip.ptr(_f32(feat_nchw, "feat_nchw")), hip.ptr(pe_nlc), hip.ptr(_f32(out_nlc, "out_nlc")), B, C, h * w,
             hip.stream_handle())


def _kpt_encode(keypoints3d, desc_bcn, wpack, out_bnc):
    B, N, _ = keypoints3d.shape
    stats = torch.empty(4 * B + 4, device=keypoints3d.device, dtype=torch.float32)
    hip.call("ophip_kpt_encode", hip.ptr(_f32(keypoints3d, "keypoints3d")), keypoints3d.stride(0), hip.ptr(_f32(desc_bcn, "desc_bcn")),
             desc_bcn.stride(0), hip.ptr(wpack), hip.ptr(stats), hip.ptr(_f32(out_bnc, "out_bnc")), B, N, hip.stream_handle())


def _layer_args(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace):
    B, L3, _ = x3d.shape
    L2 = x2d.shape[1]
    need = hip.load().ophip_encoder_x3w8_workspace_bytes(B, L3, L2)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"workspace: {need} bytes needed (ophip_encoder_x3w8_workspace_bytes)")
    return (hip.ptr(_f32(x3d, "x3d")), hip.ptr(_f32(x2d, "x2d")), hip.ptr(_f32(y3d, "y3d")), hip.ptr(_f32(y2d, "y2d")),
            B, L3, L2, hip.ptr(wpack, None), hip.ptr(wpack_next, None), int(is_cross), int(kv_from_prev), int(slot), hip.ptr(workspace, None))


def _encoder_layer_x3w8(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace):
    hip.call("ophip_encoder_layer_x3w8", *_layer_args(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace),
             hip.stream_handle())


def _encoder_layer_x3w8_frag(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace, coarse_workspace):
    """the LAST encoder layer: its rows also land in ``coarse_workspace`` as the similarity kernel's operand fragments
    (then ``coarse_match(..., nsplit | 0x100)``)"""
    B, L3, L2 = x3d.shape[0], x3d.shape[1], x2d.shape[1]
    if coarse_workspace.dtype != torch.float32 or coarse_workspace.numel() < hip.load().ophip_coarse_workspace_floats(B, L3, L2):
        raise ValueError("coarse_workspace: ophip_coarse_workspace_floats(B, N, M) float32 elements needed")
    p3, p2 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.call("ophip_coarse_frag_planes", hip.ptr(coarse_workspace), B, L3, L2, ctypes.byref(p3), ctypes.byref(p2))
    hip.call("ophip_encoder_layer_x3w8_frag", *_layer_args(x3d, x2d, y3d, y2d, wpack, wpack_next, is_cross, kv_from_prev, slot, workspace),
             p3, p2, hip.stream_handle())


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
             _bstride(keypoints3d), B, N, M, int(wc), float(temperature), float(thr), int(border_rm), float(scale), hip.ptr(conf), hip.ptr(ws),
             *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(count, torch.int32),
             int(nsplit), hip.stream_handle())
    return conf, ids[0], ids[1], ids[2], mconf, mk3, mkc, count


def _fine_refine_bf16(feat_f_cl, desc3d_f, b_ids, i_ids, j_ids, count, mkpts_c, wpack, nlayers, cross_bits, encoder_enable, nsplit,
                      wc, stride, fine_scale):
    """``feat_f_cl``: the fine map as ``[B, C, hf, wf]`` over channels-last memory (``stride(1) == 1``) or plain NCHW;
    -> ``(expec_f [cap, 3], mkpts_query_f [cap, 2])``, valid up to the device-side ``count``."""
    if feat_f_cl.dtype != torch.float32 or not feat_f_cl.is_cuda:
        raise ValueError("feat_f_cl: expected float32 on the HIP device")
    B, C, hf, wf = feat_f_cl.shape
    cap = b_ids.numel()
    dev = feat_f_cl.device
    expec, mkf = torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
    hip.call("ophip_fine_refine_bf16", hip.ptr(feat_f_cl), feat_f_cl.stride(0), feat_f_cl.stride(1), feat_f_cl.stride(2), feat_f_cl.stride(3), hf, wf,
             hip.ptr(_f32(desc3d_f, "desc3d_f")), _bstride(desc3d_f), desc3d_f.stride(1),
             hip.ptr(b_ids, torch.int64), hip.ptr(i_ids, torch.int64), hip.ptr(j_ids, torch.int64), hip.ptr(count, torch.int32), cap,
             hip.ptr(_f32(mkpts_c, "mkpts_c")), hip.ptr(wpack, None), int(nlayers), ctypes.c_uint(int(cross_bits)), int(bool(encoder_enable)), int(nsplit),
             int(wc), int(stride), float(fine_scale), hip.ptr(expec), hip.ptr(mkf), None, None, hip.stream_handle())
    return expec, mkf


CALLS = {"frame_enqueue": 0}          # how many frames went through the op (tests assert the model's default path uses it)

# frame plans: the (ophip_frame_desc, ophip_frame_layout_t) pair of a model + input shape, registered by the model once and named
# by an integer in the op call (an op schema cannot carry a C struct).  Ids come from a counter that never goes back: a dropped plan's
# id is never handed out again, so a stale id can only raise, never launch a frame with another plan's sizes and offsets.
_frame_plans = {}
_plan_ids = itertools.count(1)


def register_frame_plan(desc: "hip.FrameDesc", layout: "hip.FrameLayout", keep_alive=()) -> int:
    pid = next(_plan_ids)
    _frame_plans[pid] = (desc, layout, keep_alive)
    return pid


def drop_frame_plan(pid: int):
    _frame_plans.pop(pid, None)


def drop_frame_plans(pids):
    """release every plan of ``pids`` (a model's finalizer: its packed weight blocks leave HBM with it)"""
    for pid in list(pids):
        _frame_plans.pop(pid, None)


def _frame_enqueue(plan, block, feat_c, feat_f, fine_strides, keypoints3d, desc3d_c, desc3d_f, x3d_external, host_dst, host_bytes,
                   s_main, s_prep, s_fine, s_copy, query_mask=None, query_scale=None, y3d0=None, kv1=None, object_ready=0):
    """rows a1-a11 of one frame (``ophip_frame_enqueue_object``; ``query_mask [B, M]`` uint8 / ``query_scale [B, 2]`` float32: the
    reference's optional inputs of padded / resized query images); returns the wait ticket (``ophip_frame_wait``).
    Object cache (``ophip_object_cache``): ``x3d_external [Bo, N, 256]`` the keypoint encoding, ``y3d0 [Bo, N, 256]`` the first encoder
    layer's 3D rows and ``kv1 [Bo, kv_block_bytes]`` (uint8) layer 1's block of the 3D source (both or neither), ``Bo`` = 1 (one object
    shared by the batch) or B; ``object_ready``: raw handle of the event behind the kernels that wrote them (0: they are complete)."""
    if plan not in _frame_plans:
        raise ValueError(f"frame plan {plan} is not registered (ops.register_frame_plan)")
    d, L = _frame_plans[plan][:2]
    if block.numel() * block.element_size() < L.total:
        raise ValueError(f"block: {L.total} bytes needed (ophip_frame_layout)")
    if host_dst.is_cuda or not host_dst.is_pinned():
        raise ValueError("host_dst must be pinned host memory")
    fs = list(fine_strides)
    if len(fs) != 4:
        raise ValueError("fine_strides: 4 element strides (b, c, y, x)")
    slot = ctypes.c_int(-1)
    vp = ctypes.c_void_p
    CALLS["frame_enqueue"] += 1
    if query_mask is not None and (query_mask.dtype != torch.uint8 or tuple(query_mask.shape) != (d.B, d.M) or not query_mask.is_contiguous()):
        raise ValueError("query_mask: contiguous uint8 [B, M]")
    if query_scale is not None and (query_scale.dtype != torch.float32 or tuple(query_scale.shape) != (d.B, 2) or not query_scale.is_contiguous()):
        raise ValueError("query_scale: contiguous float32 [B, 2]")
    oc = None
    if x3d_external is not None or y3d0 is not None:
        if (y3d0 is None) != (kv1 is None):
            raise ValueError("y3d0 and kv1 go together")
        oc = hip.ObjectCache()
        for t, name in ((x3d_external, "x3d_external"), (y3d0, "y3d0")):
            if t is not None and (t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.dim() != 3
                                  or t.shape[0] not in (1, d.B) or tuple(t.shape[1:]) != (d.N, 256)):
                raise ValueError(f"{name}: contiguous float32 [1 or B, N, 256] on the HIP device")
        if x3d_external is not None:
            oc.x3d, oc.x3d_bs = x3d_external.data_ptr(), (0 if (x3d_external.shape[0] == 1 and d.B > 1) else d.N * 256)
        if y3d0 is not None:
            kvb = hip.load().ophip_encoder_x3w8_kv_block_bytes()
            if kv1.dtype != torch.uint8 or not kv1.is_cuda or not kv1.is_contiguous() or tuple(kv1.shape) != (y3d0.shape[0], kvb):
                raise ValueError(f"kv1: contiguous uint8 [{y3d0.shape[0]}, {kvb}] on the HIP device")
            shared = y3d0.shape[0] == 1 and d.B > 1
            oc.y3d0, oc.y3d0_bs = y3d0.data_ptr(), (0 if shared else d.N * 256)
            oc.kv1, oc.kv1_bs = kv1.data_ptr(), (0 if shared else kvb)
        oc.ready = int(object_ready) or None
    hip.call("ophip_frame_enqueue_object", ctypes.byref(d), ctypes.byref(L), vp(block.data_ptr()),
             hip.ptr(feat_c), hip.ptr(feat_f), fs[0], fs[1], fs[2], fs[3], hip.ptr(keypoints3d), _bstride(keypoints3d),
             hip.ptr(desc3d_c), _bstride(desc3d_c), hip.ptr(desc3d_f), _bstride(desc3d_f), desc3d_f.stride(1), ctypes.byref(oc) if oc is not None else None,
             hip.ptr(query_mask, torch.uint8), hip.ptr(query_scale), vp(host_dst.data_ptr()), int(host_bytes), vp(s_main), vp(s_prep) if s_prep else None, vp(s_fine), vp(s_copy), ctypes.byref(slot))
    return slot.value


_lib.impl("pe_add_transpose", _pe_add_transpose, "CUDA")
_lib.impl("kpt_encode", _kpt_encode, "CUDA")
_lib.impl("encoder_layer_x3w8", _encoder_layer_x3w8, "CUDA")
_lib.impl("encoder_layer_x3w8_frag", _encoder_layer_x3w8_frag, "CUDA")
_lib.impl("coarse_match", _coarse_match, "CUDA")
_lib.impl("fine_refine_bf16", _fine_refine_bf16, "CUDA")
_lib.impl("frame_enqueue", _frame_enqueue, "CUDA")

OPS = ("pe_add_transpose", "kpt_encode", "encoder_layer_x3w8", "encoder_layer_x3w8_frag", "coarse_match", "fine_refine_bf16", "frame_enqueue")
