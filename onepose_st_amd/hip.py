"""ctypes binding of ``libonepose_hip.so`` (include/onepose_hip.h).

The product path has no CPU fallback: if the shared library cannot be loaded, or a
tensor is not a contiguous float32/int64 CUDA(HIP) tensor, this module raises.
PyTorch is used for device memory and streams only; every kernel is reached through
the C ABI with raw device pointers.
"""
from __future__ import annotations

import ctypes
import os

import torch

_LIB_PATH = os.environ.get("OPHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libonepose_hip.so")   # OPHIP_LIB: A/B builds
_lib = None
ABI_VERSION = 4         # include/onepose_hip.h OPHIP_ABI_VERSION: what FrameDesc / FrameLayout / _SIGNATURES below are written for

c_f = ctypes.c_void_p      # device float*
c_i = ctypes.c_int
c_ll = ctypes.c_longlong



class FrameDesc(ctypes.Structure):
    """``ophip_frame_desc`` (include/onepose_hip.h)"""
    _fields_ = [("B", c_i), ("N", c_i), ("M", c_i), ("hc", c_i), ("wc", c_i), ("hf", c_i), ("wf", c_i), ("cf", c_i),
                ("lazy_conf", c_i), ("n_coarse", c_i), ("coarse_cross_bits", ctypes.c_uint),
                ("n_fine", c_i), ("fine_cross_bits", ctypes.c_uint), ("fine_encoder_enable", c_i),
                ("border_rm", c_i),
                ("thr", ctypes.c_float), ("scale_c", ctypes.c_float), ("fine_scale", ctypes.c_float),
                ("temperature", ctypes.c_double),
                ("pe", ctypes.c_void_p), ("w_kpt", ctypes.c_void_p), ("w_coarse", ctypes.c_void_p * 16), ("w_fine", ctypes.c_void_p)]


class FrameLayout(ctypes.Structure):
    """``ophip_frame_layout_t``: byte offsets inside the frame's device block"""
    _fields_ = [(n, ctypes.c_size_t) for n in ("total", "result_bytes", "x2d", "ffcl", "x3d", "y3d", "y2d", "z3d", "stats", "enc_ws", "conf", "cws",
                                               "result", "i_ids", "j_ids", "m_bids", "gt_mask", "mconf", "mkc", "expec", "feat3d_out", "feat2d_out")]


class ObjectCache(ctypes.Structure):
    """``ophip_object_cache``: per-object buffers of the frame-invariant encoder work (keypoint encoding, first layer's 3D rows, layer 1's
    K^T V | Ksum block of the 3D source) + the event behind the kernels that wrote them"""
    _fields_ = [("x3d", ctypes.c_void_p), ("x3d_bs", c_ll), ("y3d0", ctypes.c_void_p), ("y3d0_bs", c_ll),
                ("kv1", ctypes.c_void_p), ("kv1_bs", c_ll), ("ready", ctypes.c_void_p)]


_SIGNATURES = {
    "ophip_abi_version": (c_i, []),
    "ophip_build_stamp": (ctypes.c_char_p, []),
    "ophip_last_error": (ctypes.c_char_p, []),
    "ophip_roctx_enable": (c_i, [c_i]),
    "ophip_roctx_ranges": (c_ll, []),
    "ophip_device_info": (c_i, [ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.c_char_p, c_i]),
    "ophip_timing_select": (c_i, [ctypes.c_char_p]),
    "ophip_timing_read": (c_i, [ctypes.POINTER(c_i), ctypes.POINTER(ctypes.c_double)]),
    "ophip_timing_every": (c_i, [c_i]),
    "ophip_debug_stamps": (c_i, [ctypes.c_void_p]),
    "ophip_frame_layout": (c_i, [ctypes.POINTER(FrameDesc), c_i, c_i, ctypes.POINTER(FrameLayout)]),
    "ophip_frame_enqueue": (c_i, [ctypes.POINTER(FrameDesc), ctypes.POINTER(FrameLayout), ctypes.c_void_p,
                                  c_f, c_f, c_ll, c_ll, c_ll, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_ll, c_f,
                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.POINTER(c_i)]),
    "ophip_frame_enqueue_padded": (c_i, [ctypes.POINTER(FrameDesc), ctypes.POINTER(FrameLayout), ctypes.c_void_p,
                                         c_f, c_f, c_ll, c_ll, c_ll, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_ll, c_f, c_f, c_f,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(c_i)]),
    "ophip_frame_enqueue_object": (c_i, [ctypes.POINTER(FrameDesc), ctypes.POINTER(FrameLayout), ctypes.c_void_p,
                                         c_f, c_f, c_ll, c_ll, c_ll, c_ll, c_f, c_ll, c_f, c_ll, c_f, c_ll, c_ll, ctypes.POINTER(ObjectCache), c_f, c_f,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(c_i)]),
    "ophip_encoder_x3w8_kv_block_bytes": (ctypes.c_size_t, []),
    "ophip_encoder_object_x3w8": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_frame_wait": (c_i, [c_i]),
    "ophip_frame_order_after_fine": (c_i, [ctypes.c_void_p]),
    "ophip_pe_add_transpose": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_void_p]),
    "ophip_transpose_cl": (c_i, [c_f, c_f, c_i, c_i, c_i, ctypes.c_void_p]),
    "ophip_kpt_encode": (c_i, [c_f, c_ll, c_f, c_ll, c_f, c_f, c_f, c_i, c_i, ctypes.c_void_p]),
    "ophip_encoder_workspace_floats": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "ophip_encoder_layer": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_i, c_f, ctypes.c_void_p]),
    "ophip_encoder_bf16_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "ophip_encoder_bf16_wpack_bytes": (ctypes.c_size_t, []),
    "ophip_encoder_layer_bf16": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_encoder_x3w8_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "ophip_encoder_x3w8_wpack_bytes": (ctypes.c_size_t, []),
    "ophip_encoder_layer_x3w8": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_encoder_layer_x3w8_streams": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_encoder_kv_first_x3w8": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_i, c_f, c_f, ctypes.c_void_p]),
    "ophip_encoder_layer_x3w8_frag": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_f, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ophip_coarse_frag_planes": (c_i, [c_f, c_i, c_i, c_i, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "ophip_coarse_workspace_floats": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "ophip_coarse_match": (c_i, [c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_float, c_i, ctypes.c_float,
                                 c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_coarse_match_conf": (c_i, [c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_float, c_i, ctypes.c_float,
                                 c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_coarse_match_select": (c_i, [c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_float, c_i, ctypes.c_float,
                                 c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_fine_refine": (c_i, [c_f, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_f, c_ll, c_ll, c_f, c_f, c_f, c_f, c_i,
                                c_f, c_f, c_i, ctypes.c_uint, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_f, ctypes.c_void_p]),
    "ophip_fine_bf16_wpack_bytes": (ctypes.c_size_t, [c_i]),
    "ophip_fine_refine_bf16": (c_i, [c_f, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_f, c_ll, c_ll, c_f, c_f, c_f, c_f, c_i,
                                     c_f, c_f, c_i, ctypes.c_uint, c_i, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_f, ctypes.c_void_p]),
    "ophip_encoder_layer_masked": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_i, c_f, c_f, ctypes.c_void_p]),
    "ophip_encoder_layer_bf16_masked": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, ctypes.c_void_p]),
    "ophip_encoder_layer_x3w8_masked": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_f, c_f, ctypes.c_void_p]),
    "ophip_coarse_match_masked": (c_i, [c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_float, c_i, ctypes.c_float,
                                        c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, ctypes.c_void_p]),
    "ophip_fine_refine_scaled": (c_i, [c_f, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_f, c_ll, c_ll, c_f, c_f, c_f, c_f, c_i,
                                       c_f, c_f, c_i, ctypes.c_uint, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]),
    "ophip_fine_refine_bf16_scaled": (c_i, [c_f, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_f, c_ll, c_ll, c_f, c_f, c_f, c_f, c_i,
                                            c_f, c_f, c_i, ctypes.c_uint, c_i, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]),
    "ophip_coarse_match_2d": (c_i, [c_f, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_float, c_i, ctypes.c_float,
                                    c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_fine2_gather": (c_i, [c_f, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_fine2_gather_b": (c_i, [c_f, c_ll, c_f, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_rows_linear_wpack_bytes": (ctypes.c_size_t, [c_i, c_i]),
    "ophip_rows_linear_x3": (c_i, [c_f, c_i, c_f, c_i, c_i, c_f, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_fine2_attention": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
    "ophip_rows_layernorm128": (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, ctypes.c_void_p]),
    "ophip_fine2_match": (c_i, [c_f, c_f, c_f, c_i, c_i, ctypes.c_float, c_f, c_f, ctypes.c_void_p]),
    "ophip_conv_wpack_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "ophip_stem_conv7": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_i, ctypes.c_void_p]),
    "ophip_conv2d_bf16": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_f,
                                c_f, c_f, c_f, c_i, c_i, ctypes.c_void_p]),
    "ophip_crop_resize_gray": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class HipLibraryError(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and return the ctypes handle; raises ``HipLibraryError`` when the
    library is missing -- build it with ``python -c 'import __graft_entry__ as g; g.build()'``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryError(f"{_LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.ophip_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"libonepose_hip.so ABI version {lib.ophip_abi_version()}, this binding is written for {ABI_VERSION} "
                              "(include/onepose_hip.h OPHIP_ABI_VERSION): rebuild with __graft_entry__.build()")
    _lib = lib
    return lib


def _check(rc: int, name: str):
    if rc != 0:
        msg = load().ophip_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{name}: {msg}")
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def ptr(t: torch.Tensor | None, dtype=torch.float32):
    """Device pointer of a tensor of the given dtype (``dtype=None``: any, e.g. packed byte blocks)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("the HIP path needs device tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def stream_handle():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args):
    rc = getattr(load(), name)(*args)
    _check(rc, name)


def build_stamp() -> str:
    """16 hex digits identifying the sources ``libonepose_hip.so`` was built from (``ophip_build_stamp``)."""
    return load().ophip_build_stamp().decode()


def device_info() -> dict:
    cu, lds = c_i(0), c_i(0)
    buf = ctypes.create_string_buffer(64)
    call("ophip_device_info", ctypes.byref(cu), ctypes.byref(lds), buf, 64)
    return {"cu_count": cu.value, "lds_per_block": lds.value, "arch": buf.value.decode()}


def timing_select(kernel_name: str, every: int = 1):
    """Bracket every ``every``-th launch of ``kernel_name`` with HIP events ("" switches timing off)."""
    call("ophip_timing_every", int(every))
    call("ophip_timing_select", kernel_name.encode())


def timing_read():
    """-> (launches, total device milliseconds) since the last select/read."""
    n, ms = c_i(0), ctypes.c_double(0.0)
    call("ophip_timing_read", ctypes.byref(n), ctypes.byref(ms))
    return n.value, ms.value
