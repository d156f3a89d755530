"""Drop-in ``OnePosePlus_model`` whose hot path runs on hand-written gfx950 kernels.

Mirrors the reference interface (``src/models/OnePosePlus/OnePosePlusModel.py:24-203``):

* ``OnePosePlus_model(config, profiler=None, debug=False)`` with the ``model.OnePosePlus``
  config block; unsupported values raise ``NotImplementedError`` / ``ValueError`` as the
  reference does.
* the same ``state_dict`` key layout (195 tensors), so ``build_model`` /
  ``load_state_dict(strict=True)`` of a reference checkpoint works
  (``src/inference/inference_OnePosePlus.py:30-40``).
* ``model(data) -> None`` mutating ``data`` with exactly the keys, dtypes and shapes the
  reference writes.

The ``nn`` sub-modules below only *hold parameters* under the reference's names; the
arithmetic of rows a1-a11 is done by ``libonepose_hip.so`` through its C ABI
(``include/onepose_hip.h``).  The ResNet-FPN backbone runs on the
hand-written convolution kernels as well (``backbone_hip.py``; ``config["hip_backbone"] = False``
or the exact-f32 mode keep it on PyTorch-ROCm / MIOpen).  There is no CPU fallback: calling the
model without the HIP library or with CPU tensors raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
import os
import weakref

import torch
import torch.nn as nn

from . import hip, host_math, ops, packing
from .backbone import build_backbone
from .backbone_hip import HipBackbone, pack_backbone
from .config import encoder_layer_names, validate_config


class _LayerParams(nn.Module):
    """Parameter holder with the key layout of ``LoFTREncoderLayer`` (transformer.py:29-52)."""

    def __init__(self, d):
        super().__init__()
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.merge = nn.Linear(d, d, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d, bias=False), nn.Identity(), nn.Linear(2 * d, d, bias=False))
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _EncoderParams(nn.Module):
    """``LocalFeatureTransformer`` parameter holder (transformer.py:100-131)."""

    def __init__(self, enc_cfg):
        super().__init__()
        self.layer_names = encoder_layer_names(enc_cfg)
        self.d_model, self.nhead = enc_cfg["d_model"], enc_cfg["nhead"]
        self.layers = nn.ModuleList([_LayerParams(self.d_model) for _ in self.layer_names])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class _KeypointEncoderParams(nn.Module):
    """``KeypointEncoding_linear`` parameter holder: Linear at indices 0, 3, 6, 9
    (position_encoding.py:62-79; the norm / ReLU slots carry no parameters)."""

    def __init__(self, inp_dim, feature_dim, layers):
        super().__init__()
        chans = [inp_dim] + list(layers) + [feature_dim]
        mods = []
        for i in range(1, len(chans)):
            mods.append(nn.Linear(chans[i - 1], chans[i], bias=True))
            if i < len(chans) - 1:
                mods += [nn.Identity(), nn.Identity()]
        self.encoder = nn.Sequential(*mods)
        nn.init.constant_(self.encoder[-1].bias, 0.0)


_stream_objs = {}


def _current_stream(dev):
    """``torch.cuda.current_stream(dev)`` without its ~8 us of Python (lazy-init and device-index helpers) on the per-frame path: the raw
    handle of the current stream is one C call, the Stream object that wraps it is looked up by (device, handle)."""
    idx = dev.index
    if idx is None:
        return torch.cuda.current_stream(dev)
    raw = torch._C._cuda_getCurrentRawStream(idx)
    st = _stream_objs.get((idx, raw))
    if st is None:
        st = torch.cuda.current_stream(dev)
        if len(_stream_objs) >= 64:
            _stream_objs.clear()
        _stream_objs[(idx, raw)] = st
    return st


class _NullProfiler:
    _scope = contextlib.nullcontext()

    def record_function(self, name):
        return self._scope


class OnePosePlus_model(nn.Module):
    def __init__(self, config, profiler=None, debug=False):
        super().__init__()
        validate_config(config)
        self.config = config
        self.profiler = profiler or _NullProfiler()
        self.debug = debug
        cc, cf = config["loftr_coarse"], config["loftr_fine"]
        if cc["d_model"] != 256 or cc["nhead"] != 8:
            raise NotImplementedError("HIP coarse encoder is specialised for d_model=256, nhead=8")
        if cf["d_model"] != 128 or cf["nhead"] != 8 or cf["window_size"] != 5:
            raise NotImplementedError("HIP fine stage is specialised for d_model=128, nhead=8, window 5")

        self.backbone = build_backbone(config["loftr_backbone"])
        self.kpt_3d_pos_encoding = None
        if config["keypoints_encoding"]["enable"]:
            ke = config["keypoints_encoding"]
            if ke["descriptor_dim"] != 256 or list(ke["keypoints_encoder"]) != [32, 64, 128]:
                raise NotImplementedError("HIP keypoint encoder is specialised for 3->32->64->128->256")
            self.kpt_3d_pos_encoding = _KeypointEncoderParams(3, ke["descriptor_dim"], ke["keypoints_encoder"])
        self.loftr_coarse = _EncoderParams(cc)
        self.loftr_fine = _EncoderParams(cf)
        self._pe_enable = bool(config["positional_encoding"]["enable"])
        self._pe_shape = tuple(config["positional_encoding"]["pos_emb_shape"])
        # matrix arithmetic of the HIP kernels:
        #   "bf16x3" (default) split-bf16 MFMA, 3 x v_mfma_f32_32x32x16_bf16 per product, f32 accumulate: ~f32-grade
        #   "f32"    exact-f32 MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain), 1/16 of the bf16 rate
        #   "bf16"   plain bf16 operands: fastest, ~1e-2 relative on activations
        self.precision = str(config.get("hip_precision", os.environ.get("OPHIP_PRECISION", "bf16x3")))
        if self.precision not in ("f32", "bf16x3", "bf16"):
            raise ValueError(f"hip_precision {self.precision!r}: expected 'f32', 'bf16x3' or 'bf16'")
        # the FINE stage's arithmetic when the path runs in "bf16x3": "bf16x3" (default: every keypoint within 1e-4 relative of the reference's)
        # or "bf16" -- plain bf16 operands, a third of the stage's matrix work: match indices unchanged (the fine stage only moves a match's
        # sub-pixel offset), keypoints within 0.05 px, pose within 1e-5 of the default's (tests/test_gpu_parity.py::test_fine_stage_in_plain_bf16...).
        # The library reads the switch from the environment on every fine-stage launch, so the key is PROCESS-WIDE: giving it sets
        # OPHIP_FINE_PRECISION for every model of the process; leaving it out keeps whatever the environment says (default "bf16x3").
        fp = config.get("hip_fine_precision")
        if fp is not None:
            if str(fp) not in ("bf16x3", "bf16"):
                raise ValueError(f"hip_fine_precision {fp!r}: expected 'bf16x3' or 'bf16'")
            os.environ["OPHIP_FINE_PRECISION"] = str(fp)
        # backbone on the HIP convolution kernels (bf16 pipe modes only; exact-f32 mode keeps MIOpen's fp32 convolutions)
        self.hip_backbone = bool(config.get("hip_backbone", True)) and self.precision != "f32"
        self.lazy_reruns = 0          # lazy conf_matrix: frames re-run eagerly because of an exact row tie (PendingFrame.finish)
        self._packed = None          # (key, dict of device weight blocks)
        self._packed_bb = None       # (key, backbone conv blocks)
        # fine stage (+ result read-back) on a second HIP stream: frame t's refinement then overlaps frame t + 1's input
        # kernels (PE add, transposes, keypoint encoding, backbone); frame t + 1's encoder waits for it, so the two
        # MFMA-heavy stages never share the chip
        self.overlap_fine = bool(config.get("hip_overlap_fine", True))
        self._fine_streams = {}      # (device, compute stream) -> (fine stream, last fine-done event)
        self._prep_streams = {}      # (device, compute stream) -> input-kernel stream (enqueue_features(inputs_ready=True))
        self._pe_cache = {}          # (h, w, device) -> [M, C] device table
        # per-object cache of the frame-invariant keypoint encoding (rows a2 + a3); off by default so that a forward always does
        # all of its work unless the caller opts in (bench.py reports both)
        self.cache_object = bool(config.get("hip_cache_object", False))
        self._obj_cache = None
        # conf_matrix: "eager" (default) stores the N x M matrix every frame like the reference; "lazy" never stores it (the
        # inference callers read only the match lists): ``data["conf_matrix"]`` is then a :class:`LazyConfMatrix` that materialises on
        # first use.  Match indices and confidences are bit-identical in both forms.
        self.conf_matrix_mode = str(config.get("hip_conf_matrix", "eager"))
        if self.conf_matrix_mode not in ("eager", "lazy"):
            raise ValueError(f"hip_conf_matrix {self.conf_matrix_mode!r}: expected 'eager' or 'lazy'")
        if self.conf_matrix_mode == "lazy" and self.precision == "f32":
            raise ValueError("hip_conf_matrix = 'lazy' needs a bf16 arithmetic mode (the exact-f32 mode always materialises conf_matrix)")
        # the default path as ONE C call per frame (csrc/frame.hip: same kernels, same streams, one device block per frame) instead of
        # ~27 ctypes calls + ~20 allocations + ~10 stream / event operations from Python: 0.39 -> ~0.1 ms of host time per frame
        self.frame_call = bool(config.get("hip_frame_call", os.environ.get("OPHIP_FRAME_CALL", "1") != "0"))
        self._frame_plans = {}
        # ids this model holds in ops._frame_plans (whose entries keep the packed weight blocks alive): released when the model is collected
        self._plan_ids = set()
        weakref.finalize(self, ops.drop_frame_plans, self._plan_ids)
        self._frame_call_pending = set()     # compute streams whose last frame went through the C entry point

        pretrained = config["loftr_backbone"]["pretrained"]
        if pretrained is not None:
            # OnePosePlusModel.py:78-93: take the `backbone.*` entries of a LoFTR checkpoint
            ckpt = torch.load(pretrained, map_location="cpu", weights_only=True)["state_dict"]
            sub = {k[k.find("backbone") + len("backbone") + 1:]: v for k, v in ckpt.items() if "backbone" in k}
            self.backbone.load_state_dict(sub)
            if config["loftr_backbone"]["pretrained_fix"]:
                for p in self.backbone.parameters():
                    p.requires_grad = False

    # ------------------------------------------------------------------------------------------
    # device-side weight blocks (re-packed when parameters change or move)
    # ------------------------------------------------------------------------------------------
    _PARAM_REWALK = 64          # frames between two full walks of the module tree in _weights() (a backstop: see _matcher_params)

    def _matcher_params(self):
        """The parameters whose packed copies the kernels read, as a list that is NOT rebuilt on every frame: walking the module tree
        (``named_parameters``: 145 tensors below ~40 modules) cost ~100 us of a 690 us frame period -- a third of the host time of an
        enqueue -- for a question whose answer changes when a user edits the model.  The list is rebuilt every ``_PARAM_REWALK`` frames,
        whenever ``_apply`` (``.to`` / ``.cuda`` / ``.float``) or ``load_state_dict`` ran, and whenever the tree itself changed: beside the
        list the cache keeps every (module, name, Parameter) and (parent, name, sub-module) edge it was built from and re-checks their
        identity on each frame (~190 dictionary look-ups, ~15 us), so a replaced Parameter (``layer.q_proj.weight = nn.Parameter(...)``,
        ``load_state_dict(assign=True)`` on a sub-module, ``parametrize``) or a replaced sub-module is seen on the very next frame; a
        parameter that is written in place or moved is caught by the (data_ptr, _version) key of ``_weights``."""
        c = self.__dict__.get("_param_cache")
        if c is not None and c[1] > 0:
            for mod, name, p in c[2]:
                if mod._parameters.get(name) is not p:
                    c = None
                    break
            else:
                for parent, name, child in c[3]:
                    if parent._modules.get(name) is not child:
                        c = None
                        break
        if c is None or c[1] <= 0:
            pedges, medges, plist = [], [], []
            stack = [(n, m) for n, m in self._modules.items() if n != "backbone" and m is not None]
            medges += [(self, n, m) for n, m in stack]
            while stack:
                _, mod = stack.pop()
                for n, p in mod._parameters.items():
                    if p is not None:
                        pedges.append((mod, n, p))
                for n, ch in mod._modules.items():
                    if ch is not None:
                        medges.append((mod, n, ch))
                        stack.append((n, ch))
            plist = [p for n, p in self.named_parameters() if not n.startswith("backbone.")]
            c = self.__dict__["_param_cache"] = [plist, self._PARAM_REWALK, pedges, medges]
        c[1] -= 1
        return c[0]

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_param_cache"] = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_param_cache"] = None
        return super().load_state_dict(*args, **kwargs)

    def _weights(self, device):
        params = self._matcher_params()
        key = (str(device), self.precision, tuple([p.data_ptr() for p in params]), tuple([p._version for p in params]))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        self.__dict__["_param_cache"] = None              # a change: the next frame walks the tree again
        params = self._matcher_params()
        key = (str(device), self.precision, tuple([p.data_ptr() for p in params]), tuple([p._version for p in params]))
        sd = {k: v for k, v in self.state_dict().items() if not k.startswith("backbone.")}
        blocks = {
            "coarse": [packing.pack_coarse_layer(sd, f"loftr_coarse.layers.{i}.").to(device)
                       for i in range(len(self.loftr_coarse.layer_names))],
            "fine": torch.cat([packing.pack_fine_layer(sd, f"loftr_fine.layers.{i}.")
                               for i in range(len(self.loftr_fine.layer_names))]).to(device),
        }
        if self.kpt_3d_pos_encoding is not None:
            blocks["kpt"] = packing.pack_keypoint_encoder(sd).to(device)
        if self.precision == "bf16":
            blocks["coarse_bf16"] = [packing.pack_coarse_layer_bf16(sd, f"loftr_coarse.layers.{i}.").to(device)
                                     for i in range(len(self.loftr_coarse.layer_names))]
        if self.precision == "bf16x3":
            blocks["coarse_x3"] = [packing.pack_coarse_layer_x3w8(sd, f"loftr_coarse.layers.{i}.").to(device)
                                   for i in range(len(self.loftr_coarse.layer_names))]
        if self.precision != "f32":
            blocks["fine_bf16"] = packing.pack_fine_layers_bf16(sd, "loftr_fine.layers.", len(self.loftr_fine.layer_names)).to(device)
        self._packed = (key, blocks)
        return blocks

    def _backbone_blocks(self, device):
        tensors = list(self.backbone.state_dict().values())
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed_bb is None or self._packed_bb[0] != key:
            self._packed_bb = (key, pack_backbone(self.backbone.state_dict(), device))
        return self._packed_bb[1]

    def _pe_table(self, h, w, device):
        k = (h, w, str(device))
        if k not in self._pe_cache:
            C = self.config["loftr_coarse"]["d_model"]
            pe = host_math.sinusoid_table(C, h, w, self._pe_shape)              # [C, h, w]
            self._pe_cache[k] = pe.flatten(1).t().contiguous().to(device)        # [M, C]
        return self._pe_cache[k]

    # ------------------------------------------------------------------------------------------
    def forward(self, data):
        """Same contract as the reference ``forward`` (OnePosePlusModel.py:95-203)."""
        self.enqueue(data).finish()

    def enqueue(self, data, host_copy=False):
        """``forward`` without the final wait: backbone + rows a1-a11 enqueued on the current stream; returns the
        :class:`PendingFrame` (see :meth:`enqueue_features`)."""
        if self.training:
            raise NotImplementedError("the HIP path implements inference; call .eval() (training padding "
                                      "of coarse_matching.py:177-217 is out of scope)")
        if "mask0" in data:
            raise NotImplementedError("data['mask0']: not implemented (the reference raises as well, coarse_matching.py:149-152)")
        data.update({"bs": data["query_image"].size(0), "q_hw_i": data["query_image"].shape[2:]})
        img = data["query_image"]
        if self.hip_backbone:
            feat_c, feat_f = self.backbone_features(img)
            return self.enqueue_features(data, feat_c, feat_f, host_copy=host_copy, _pe_applied=True)
        with torch.no_grad():
            feat_c, feat_f = self.backbone(img)
        return self.enqueue_features(data, feat_c, feat_f, host_copy=host_copy)

    def backbone_features(self, img):
        """Row f-1 on the HIP convolution kernels: ``[B, 1, H, W]`` image -> ``(feat_c, feat_f)`` with the reference's
        ``[B, C, h, w]`` shapes over channels-last memory; the positional encoding (row a1) is already added to ``feat_c``."""
        if not self.hip_backbone:
            raise RuntimeError("hip_backbone is disabled for this model (exact-f32 mode or config['hip_backbone'] = False)")
        if not img.is_cuda:
            raise hip.HipLibraryError("OnePosePlus_model runs on the HIP device only (no CPU fallback): move the "
                                      "model and its inputs to 'cuda'")
        B, _, H, W = img.shape
        pe = self._pe_table(H // 8, W // 8, img.device) if self._pe_enable else None
        fc, ff = HipBackbone(self.precision).forward(self._backbone_blocks(img.device), img, pe)
        return (fc.view(B, H // 8, W // 8, fc.shape[2]).permute(0, 3, 1, 2), ff.view(B, H // 2, W // 2, ff.shape[2]).permute(0, 3, 1, 2))

    def forward_features(self, data, feat_c, feat_f, image_hw=None, want_fine_debug=False):
        """The north_star path: everything after the backbone.  ``feat_c [B,256,H/8,W/8]``,
        ``feat_f [B,128,H/2,W/2]`` (any strides); fills ``data`` like :meth:`forward`."""
        self.enqueue_features(data, feat_c, feat_f, image_hw, want_fine_debug).finish()

    @torch.no_grad()
    def enqueue_features(self, data, feat_c, feat_f, image_hw=None, want_fine_debug=False, host_copy=False, _pe_applied=False,
                         inputs_ready=False, _force_eager=False):
        """Enqueue the whole path for one batch on the current stream WITHOUT synchronising and return a
        :class:`PendingFrame`; ``.finish()`` waits for that frame only (an event, not the stream) and fills ``data``.
        A pipeline enqueues frame t + 1 before finishing frame t, so the GPU never idles on the host
        (``bench.py``).  The fine stage runs on a second stream (``config["hip_overlap_fine"]``, default on): the inputs
        must stay unmodified until ``finish()``.  ``host_copy=True`` also queues the D2H of the match buffers into pinned memory, which
        ``finish()`` exposes as numpy arrays (``pending.host``) for host PnP.  ``inputs_ready=True``: the caller guarantees that
        ``feat_c``, ``feat_f`` and the object block are complete in device memory (nothing that writes them is still queued), so
        the input kernels may run on a side stream ahead of the work already queued on the current one."""
        if not feat_c.is_cuda:
            raise hip.HipLibraryError("OnePosePlus_model runs on the HIP device only (no CPU fallback): move the "
                                      "model and its inputs to 'cuda'")
        hip.load()
        lib_call, P = hip.call, hip.ptr
        cfg = self.config
        dev = feat_c.device
        if image_hw is not None:
            data.update({"bs": feat_c.size(0), "q_hw_i": torch.Size(image_hw)})
        B, C, hc, wc = feat_c.shape
        hf, wf = feat_f.shape[2:]
        M = hc * wc
        data.update({"q_hw_c": feat_c.shape[2:], "q_hw_f": feat_f.shape[2:]})
        W = self._weights(dev)
        kpts = data["keypoints3d"]
        desc_fine = data["descriptors3d_db"]
        desc_in = data["descriptors3d_coarse_db"] if "descriptors3d_coarse_db" in data else desc_fine
        N = kpts.shape[1]
        f32 = dict(device=dev, dtype=torch.float32)

        def dense(t):
            return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()

        def bstride(t):          # batch stride in elements; expanded (shared) object blocks give 0
            return 0 if t.shape[0] == 1 or t.stride(0) == 0 else t.stride(0)

        def per_frame(t):        # [B or 1, ...] dense per frame, without materialising an expand()
            if t.stride(0) == 0 and t.shape[0] > 1:
                t = t[:1]
            return dense(t)

        kpts_d, desc_in_d, desc_fine_d = per_frame(kpts), per_frame(desc_in), per_frame(desc_fine)
        for t, name in ((kpts_d, "keypoints3d"), (desc_in_d, "descriptors3d"), (desc_fine_d, "descriptors3d_db")):
            if t.shape[0] not in (1, B):
                raise ValueError(f"{name}: batch {t.shape[0]} does not match the query batch {B}")

        # optional inputs of padded / resized query images (OnePosePlusModel.py:104,156-158; datasets with img_pad / img_resize)
        qmask = qscale = None
        if "query_image_mask" in data:
            qm = data["query_image_mask"]
            if tuple(qm.shape) != (B, hc, wc):
                raise ValueError(f"query_image_mask: expected shape {(B, hc, wc)} (the coarse grid), got {tuple(qm.shape)}")
            qmask = (qm if qm.dtype == torch.bool else qm != 0).to(dev).contiguous().view(torch.uint8).view(B, M)
        if "query_image_scale" in data:
            qs = data["query_image_scale"]
            if tuple(qs.shape) != (B, 2):
                raise ValueError(f"query_image_scale: expected shape {(B, 2)} ((h, w) factors per image), got {tuple(qs.shape)}")
            qscale = qs.to(device=dev, dtype=torch.float32).contiguous()
        PM = lambda: P(qmask, torch.uint8)

        main = _current_stream(dev)
        S = ctypes.c_void_p(main.cuda_stream)          # (the stage-by-stage path below launches on it)
        fkey = (str(dev), main.cuda_stream)
        lazy = self.conf_matrix_mode == "lazy" and not _force_eager
        # a lazy frame whose selection meets an exact row tie it cannot resolve without the stored row is run again with conf_matrix
        rerun = (lambda: self.enqueue_features(data, feat_c, feat_f, image_hw, want_fine_debug, host_copy, _pe_applied, False, True)) if lazy else None
        if (self.frame_call and self.precision == "bf16x3" and self.overlap_fine and not _pe_applied and not want_fine_debug
                and not self.debug and bool(cfg["fine_matching"]["enable"])
                and (self.kpt_3d_pos_encoding is not None or B == 1 or desc_in_d.shape[0] == B)
                and len(self.loftr_coarse.layer_names) <= 16):
            x3d_ext = None
            shared_batch = B > 1 and kpts_d.shape[0] == 1 and desc_in_d.shape[0] == 1
            if self.cache_object or shared_batch:
                # the reference keeps the object block resident across frames (OnePosePlus_inference_dataset.py:157-169): what depends on it
                # and the weights alone is computed once per (object tensors, weights) and handed to the frame call -- the keypoint encoding
                # (rows a2 + a3) and, with a first layer of kind "self", that layer's 3D rows and the K^T V | Ksum block of those rows as the
                # second layer's source (transformer.py:148-159; ophip_encoder_object_x3w8: the frame's own launches on the 3D stream's
                # workgroups, so a cached frame is bit-identical).  One entry serves the whole batch when it shares one object (config 3).
                # A batch whose frames share ONE object block (stride-0 expand: BASELINE config 3) takes the same route WITHOUT the cache flag:
                # the object-only work is then done once per CALL instead of once per frame of the batch -- nothing is kept for the next call
                # (`keep=False`), so a forward still does all of its own work.
                x3d_ext = self._object_cache_entry(kpts_d, desc_in_d, W, dev, B, N, main, masked=qmask is not None, keep=self.cache_object)
            return self._enqueue_frame_call(data, feat_c, feat_f, kpts_d, desc_in_d, desc_fine_d, x3d_ext, W, dev, main, fkey,
                                            B, N, M, hc, wc, hf, wf, host_copy, inputs_ready, lazy, rerun, qmask, qscale)
        if fkey in self._frame_call_pending:                          # order this frame's encoder behind the C path's last fine stage
            self._frame_call_pending.discard(fkey)
            lib_call("ophip_frame_order_after_fine", ctypes.c_void_p(main.cuda_stream))
        # The input kernels (a1-a3 + the fine map's transpose) depend on nothing but the caller's tensors.  With ``inputs_ready`` the
        # caller states that those tensors are complete (no producer still queued on this stream): the kernels then go to a side
        # stream and start at once -- on the CUs the 246-workgroup encoder of the previous frame leaves idle and beside its coarse
        # stage -- instead of behind everything queued before; the encoder waits for their event.
        prep_ctx, sprep = contextlib.nullcontext(), None
        if inputs_ready and self.overlap_fine and not _pe_applied:
            if fkey not in self._prep_streams:
                if len(self._prep_streams) >= 16:
                    self._prep_streams.pop(next(iter(self._prep_streams)))
                self._prep_streams[fkey] = torch.cuda.Stream(device=dev)
            sprep = self._prep_streams[fkey]
            prep_ctx = torch.cuda.stream(sprep)
        with prep_ctx:
            S_in = hip.stream_handle()
            # ---- a1: positional encoding + flatten ------------------------------------------------
            if _pe_applied:          # the HIP backbone wrote [B, M, C] with the encoding added; the encoder may reuse that buffer
                x2d = feat_c.permute(0, 2, 3, 1).reshape(B, M, C)
                if x2d.data_ptr() != feat_c.data_ptr() or not x2d.is_contiguous():
                    raise ValueError("internal: the HIP backbone's coarse map must be dense channels-last")
            else:
                x2d = torch.empty(B, M, C, **f32)
                pe = self._pe_table(hc, wc, dev) if self._pe_enable else None
                lib_call("ophip_pe_add_transpose", P(dense(feat_c)), P(pe), P(x2d), B, C, M, S_in)
            # ---- fine map to channels-last (input kernel of the fine stage; here so that it runs under the previous frame's
            #      fine stage instead of in front of this frame's) ------------------------------------------------------
            ff = ff_strides = None
            if bool(cfg["fine_matching"]["enable"]):
                ff = feat_f if feat_f.dtype == torch.float32 else feat_f.float()
                if ff.stride(1) == 1:                      # channels-last memory already: strides as they are
                    ff_strides = (ff.stride(0), 1, ff.stride(2), ff.stride(3))
                else:                                      # NCHW: one streaming transpose, then every window pixel is a 512-byte row
                    ff = ff.contiguous()
                    ff_cl = torch.empty(B, hf * wf, ff.shape[1], **f32)
                    lib_call("ophip_transpose_cl", P(ff), P(ff_cl), B, ff.shape[1], hf * wf, S_in)
                    ff, ff_strides = ff_cl, (hf * wf * ff_cl.shape[2], 1, wf * ff_cl.shape[2], ff_cl.shape[2])
            # ---- a2 + a3: keypoint encoding (frame-invariant: depends on the object block only) -------------------
            x3d = None
            ckey = None
            if self.cache_object:
                # the reference keeps the object block resident across frames (OnePosePlus_inference_dataset.py:157-169); its encoding
                # is recomputed only when the tensors (or the weights) change
                ent = self._object_cache_entry(kpts_d, desc_in_d, W, dev, B, N, torch.cuda.current_stream(dev))
                torch.cuda.current_stream(dev).wait_event(ent["ev"])
                x3d = ent["x3d"] if ent["x3d"].shape[0] == B else ent["x3d"].expand(B, -1, -1).contiguous()
            if x3d is None:
                x3d = torch.empty(B, N, C, **f32)
                if self.kpt_3d_pos_encoding is not None:
                    stats = torch.empty(4 * B + 4, **f32)
                    lib_call("ophip_kpt_encode", P(kpts_d), bstride(kpts_d), P(desc_in_d), bstride(desc_in_d), P(W["kpt"]),
                             P(stats), P(x3d), B, N, S_in)
                else:
                    src = desc_in_d if desc_in_d.shape[0] == B else desc_in_d.expand(B, -1, -1).contiguous()
                    lib_call("ophip_transpose_cl", P(src), P(x3d), B, C, N, S_in)
            if sprep is not None:
                prep_done = torch.cuda.Event()
                prep_done.record()
        if sprep is not None:
            main.wait_event(prep_done)
            for t in (x2d, x3d, ff):                     # allocated from s_prep's pool, used on the compute / fine streams from here on
                if t is not None:
                    t.record_stream(main)
        # ---- a4-a6: coarse encoder ----------------------------------------------------------------
        y3d, y2d = torch.empty_like(x3d), torch.empty_like(x2d)
        z3d = torch.empty_like(x3d) if self.cache_object else x3d          # a cached encoding is read-only: ping-pong between y and z

        def wait_previous_fine():                                 # the previous frame's fine stage is done
            if self.overlap_fine and fkey in self._fine_streams and self._fine_streams[fkey][1] is not None:
                main.wait_event(self._fine_streams[fkey][1])
        if self.precision == "f32":
            wait_previous_fine()
            ws = torch.empty(hip.load().ophip_encoder_workspace_floats(B, N, M), **f32)
            for li, name in enumerate(self.loftr_coarse.layer_names):
                if qmask is None:
                    lib_call("ophip_encoder_layer", P(x3d), P(x2d), P(y3d), P(y2d), B, N, M, P(W["coarse"][li]),
                             1 if name == "cross" else 0, P(ws), S)
                else:
                    lib_call("ophip_encoder_layer_masked", P(x3d), P(x2d), P(y3d), P(y2d), B, N, M, P(W["coarse"][li]),
                             1 if name == "cross" else 0, P(ws), PM(), S)
                x3d, y3d, x2d, y2d = y3d, (z3d if li == 0 else x3d), y2d, x2d
        elif self.precision == "bf16x3":
            # 16-token tiles, one eight-wave workgroup per CU, per-wave weight streams (csrc/encoder_x3w8.hip)
            entry = "ophip_encoder_layer_x3w8"
            ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, N, M), device=dev, dtype=torch.uint8)
            names_c = self.loftr_coarse.layer_names
            for li, name in enumerate(names_c):
                nxt = W["coarse_x3"][li + 1] if li + 1 < len(names_c) else None
                if li == 0:
                    wait_previous_fine()
                lib_call(entry + ("_masked" if qmask is not None else ""), P(x3d), P(x2d), P(y3d), P(y2d), B, N, M, P(W["coarse_x3"][li], None), P(nxt, None),
                         1 if name == "cross" else 0, 1 if li > 0 else 0, li & 1, P(ws, None), *((PM(),) if qmask is not None else ()), S)
                x3d, y3d, x2d, y2d = y3d, (z3d if li == 0 else x3d), y2d, x2d
        else:
            nsplit = 1                                            # plain-bf16 mode
            ws = torch.empty(hip.load().ophip_encoder_bf16_workspace_bytes(B, N, M), device=dev, dtype=torch.uint8)
            names_c = self.loftr_coarse.layer_names
            for li, name in enumerate(names_c):
                # layer li's attn_apply also emits layer li+1's K/V partial slabs from the on-chip output tile
                nxt = W["coarse_bf16"][li + 1] if li + 1 < len(names_c) else None
                if li == 0:
                    wait_previous_fine()                          # attn_apply, the roofline kernel, never shares the chip with fine
                lib_call("ophip_encoder_layer_bf16" + ("_masked" if qmask is not None else ""), P(x3d), P(x2d), P(y3d), P(y2d), B, N, M,
                         P(W["coarse_bf16"][li], None), P(nxt, None),
                         nsplit, 1 if name == "cross" else 0, 1 if li > 0 else 0, li & 1, P(ws, None), *((PM(),) if qmask is not None else ()), S)
                x3d, y3d, x2d, y2d = y3d, (z3d if li == 0 else x3d), y2d, x2d
        if self.debug:
            data["_feat3d_c"], data["_feat2d_c"] = x3d, x2d
        # ---- a7 + a8: coarse matching -----------------------------------------------------------
        cm = cfg["coarse_matching"]
        cap = B * N
        conf = None if lazy else torch.empty(B, N, M, **f32)
        cws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), **f32)
        i64 = dict(device=dev, dtype=torch.int64)
        # what the host reads back (count, b_ids, 3D points, refined 2D points) lives in one block: one D2H copy per frame
        blob, count, b_ids, mk3d, mk2d = _result_block(dev, cap)
        i_ids, j_ids, m_bids = torch.empty(cap, **i64), torch.empty(cap, **i64), torch.empty(cap, **i64)
        gt_mask = torch.empty(cap, device=dev, dtype=torch.bool)
        fine_on = bool(cfg["fine_matching"]["enable"])
        mconf = torch.empty(cap, **f32)
        mkc = torch.empty(cap, 2, **f32) if fine_on else mk2d
        scale = data["q_hw_i"][0] / hc
        cm_args = (P(x3d), P(x2d), P(kpts_d), bstride(kpts_d), B, N, M, wc,
                   float(cm["dual_softmax"]["temperature"]), float(cm["thr"]), int(cm["border_rm"]), float(scale),
                   P(conf), P(cws), P(b_ids, torch.int64), P(i_ids, torch.int64), P(j_ids, torch.int64),
                   P(mconf), P(mk3d), P(mkc), P(m_bids, torch.int64), P(gt_mask, torch.bool), P(count, torch.int32),
                   {"f32": 0, "bf16": 1, "bf16x3": 3}[self.precision])
        # the single-workgroup select kernel goes with the fine stage onto the side stream (it only feeds that stage and the
        # read-back): the next frame's input kernels then run beside it instead of behind it
        split_select = fine_on and self.overlap_fine
        padded = qmask is not None or qscale is not None
        with self.profiler.record_function("LoFTR/coarse-matching/get_coarse_match"):
            if padded:
                lib_call("ophip_coarse_match_masked", *cm_args, 1 if split_select else 3, PM() if qmask is not None else None, P(qscale), S)
            else:
                lib_call("ophip_coarse_match_conf" if split_select else "ophip_coarse_match", *cm_args, S)
        data["conf_matrix"] = conf if not lazy else LazyConfMatrix(x3d, x2d, float(cm["dual_softmax"]["temperature"]),
                                                                     {"f32": 0, "bf16": 1, "bf16x3": 3}[self.precision], main, qmask)

        fine_ctx = contextlib.nullcontext()
        if fine_on and self.overlap_fine:
            if fkey not in self._fine_streams:
                if len(self._fine_streams) >= 16:
                    self._fine_streams.pop(next(iter(self._fine_streams)))
                self._fine_streams[fkey] = [torch.cuda.Stream(device=dev), None]
            sfine = self._fine_streams[fkey][0]
            coarse_done = torch.cuda.Event()
            coarse_done.record(main)
            sfine.wait_event(coarse_done)
            fine_ctx = torch.cuda.stream(sfine)
        keep = [desc_fine_d, W, qmask, qscale]      # inputs of the side-stream kernels stay referenced until finish()
        keep += [x3d, x2d, cws, conf]
        with fine_ctx:
            S = hip.stream_handle()
            if split_select and padded:
                lib_call("ophip_coarse_match_masked", *cm_args, 2, PM() if qmask is not None else None, P(qscale), S)
            elif split_select:
                lib_call("ophip_coarse_match_select", *cm_args, S)
            if fine_on:
                # ---- a9-a11: fine refinement (grid sized by capacity, device-side count: no sync yet) ----
                cf = cfg["loftr_fine"]
                expec = torch.empty(cap, 3, **f32)
                mkf = mk2d
                dbg_w = torch.empty(cap, 25, 128, **f32) if want_fine_debug else None
                dbg_3 = torch.empty(cap, 128, **f32) if want_fine_debug else None
                names_f = self.loftr_fine.layer_names
                cross_bits = sum(1 << i for i, n in enumerate(names_f) if n == "cross")
                keep.append(ff)
                stride = hf // hc
                fine_scale = (cf["window_size"] // 2) * (data["q_hw_i"][0] / hf)
                max_matches = cap                                # one match per 3D point at most: the grid covers every possible K (surplus workgroups exit)
                if self.precision == "f32":
                    lib_call("ophip_fine_refine" + ("_scaled" if qscale is not None else ""), P(ff), *ff_strides, hf, wf,
                             P(desc_fine_d), bstride(desc_fine_d), desc_fine_d.stride(1),
                             P(b_ids, torch.int64), P(i_ids, torch.int64), P(j_ids, torch.int64), P(count, torch.int32), max_matches,
                             P(mkc), P(W["fine"]), len(names_f), ctypes.c_uint(cross_bits), 1 if cf["enable"] else 0,
                             wc, stride, float(fine_scale), P(expec), P(mkf), P(dbg_w), P(dbg_3), *((P(qscale),) if qscale is not None else ()), S)
                else:
                    lib_call("ophip_fine_refine_bf16" + ("_scaled" if qscale is not None else ""), P(ff), *ff_strides, hf, wf,
                             P(desc_fine_d), bstride(desc_fine_d), desc_fine_d.stride(1),
                             P(b_ids, torch.int64), P(i_ids, torch.int64), P(j_ids, torch.int64), P(count, torch.int32), max_matches,
                             P(mkc), P(W["fine_bf16"], None), len(names_f), ctypes.c_uint(cross_bits), 1 if cf["enable"] else 0,
                             3 if self.precision == "bf16x3" else 1,
                             wc, stride, float(fine_scale), P(expec), P(mkf), P(dbg_w), P(dbg_3), *((P(qscale),) if qscale is not None else ()), S)

            if fine_on and self.overlap_fine:
                fine_done = torch.cuda.Event()
                fine_done.record()
                self._fine_streams[fkey][1] = fine_done
            pend = PendingFrame(self, data, dev, B, N, M, cap, fine_on, want_fine_debug,
                                dict(blob=blob, b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, mconf=mconf, mk3d=mk3d, mkc=mkc, count=count,
                                     m_bids=m_bids, gt_mask=gt_mask,
                                     expec=expec if fine_on else None, mkf=mkf if fine_on else None,
                                     dbg_w=dbg_w if fine_on else None, dbg_3=dbg_3 if fine_on else None, keep=keep), host_copy)
            pend._rerun = rerun
        return pend


    def _object_cache_entry(self, kpts_d, desc_in_d, W, dev, B, N, stream, masked=False, keep=True):
        """The object's cache entry ``{"x3d", "y3d0", "kv1", "ev"}`` (``ophip_object_cache``), built on a miss on ``stream`` with the kernels a
        frame would run.  Keyed on the object tensors' storage + version and the packed weights; ``Bo`` = 1 rows when the batch shares one
        object block (stride-0 expand / batch-1 tensors under a larger query batch), else B."""
        lib_call, P = hip.call, hip.ptr
        shared = B > 1 and kpts_d.shape[0] == 1 and desc_in_d.shape[0] == 1
        Bo = 1 if shared else B
        names = self.loftr_coarse.layer_names
        deep = (self.precision == "bf16x3" and len(names) >= 2 and names[0] == "self"
                and os.environ.get("OPHIP_OBJECT_CACHE_DEPTH", "2") != "1")
        # (masked: frames with a query_image_mask run both streams through the masked instantiation of the layer kernel; their entry is built
        #  with that instantiation, see ophip_encoder_object_x3w8)
        ckey = (str(dev), Bo, N, kpts_d.data_ptr(), kpts_d._version, desc_in_d.data_ptr(), desc_in_d._version, id(W), deep, bool(masked) and deep)
        ent = self._obj_cache
        if keep and ent is not None and ent["key"] == ckey:
            return ent
        f32 = dict(device=dev, dtype=torch.float32)
        with torch.cuda.stream(stream):
            x3d = torch.empty(Bo, N, 256, **f32)
            if self.kpt_3d_pos_encoding is not None:
                stats = torch.empty(4 * Bo + 4, **f32)
                lib_call("ophip_kpt_encode", P(kpts_d), 0 if kpts_d.shape[0] == 1 else kpts_d.stride(0), P(desc_in_d),
                         0 if desc_in_d.shape[0] == 1 else desc_in_d.stride(0), P(W["kpt"]), P(stats), P(x3d), Bo, N, hip.stream_handle())
            else:
                src = desc_in_d if desc_in_d.shape[0] == Bo else desc_in_d.expand(Bo, -1, -1).contiguous()
                lib_call("ophip_transpose_cl", P(src), P(x3d), Bo, 256, N, hip.stream_handle())
            y3d0 = kv1 = None
            if deep:
                lib = hip.load()
                y3d0 = torch.empty(Bo, N, 256, **f32)
                kv1 = torch.empty(Bo, lib.ophip_encoder_x3w8_kv_block_bytes(), device=dev, dtype=torch.uint8)
                ws = torch.empty(lib.ophip_encoder_x3w8_workspace_bytes(Bo, N, 1), device=dev, dtype=torch.uint8)
                lib_call("ophip_encoder_object_x3w8", P(x3d), Bo, N, P(W["coarse_x3"][0], None), P(W["coarse_x3"][1], None), P(ws, None),
                         P(y3d0), P(kv1, None), 1 if masked else 0, hip.stream_handle())
            ev = torch.cuda.Event()
            ev.record(stream)
        ent = {"key": ckey, "x3d": x3d, "y3d0": y3d0, "kv1": kv1, "ev": ev, "keep": (kpts_d, desc_in_d, W)}      # the key's tensors stay alive with the entry
        if keep:
            self._obj_cache = ent
        return ent

    def flush(self):
        """End of a sequence: no further frame is coming on the current stream, so the last frame's kept-back fine stage (it would otherwise
        go out with the next ``enqueue`` or at its ``finish()``) is launched now, behind its own selection.  Optional -- ``finish()`` does it
        as well, only later (after the frames before it have been waited for)."""
        dev = torch.cuda.current_device()
        main = torch.cuda.current_stream()
        fkey = (str(torch.device("cuda", dev)), main.cuda_stream)
        if fkey in self._frame_call_pending:
            hip.call("ophip_frame_order_after_fine", ctypes.c_void_p(main.cuda_stream))

    def _side_stream(self, table, fkey, dev):
        st = table.get(fkey)
        if st is None:
            if len(table) >= 16:
                table.pop(next(iter(table)))
            st = table[fkey] = torch.cuda.Stream(device=dev)
        return st

    def _enqueue_frame_call(self, data, feat_c, feat_f, kpts_d, desc_in_d, desc_fine_d, x3d_ext, W, dev, main, fkey,
                            B, N, M, hc, wc, hf, wf, host_copy, inputs_ready, lazy=False, rerun=None, qmask=None, qscale=None):
        """The whole frame through ``ophip_frame_enqueue`` (csrc/frame.hip): one device block, one C call."""
        cfg = self.config
        fc = feat_c if (feat_c.dtype == torch.float32 and feat_c.is_contiguous()) else feat_c.float().contiguous()
        ff = feat_f if feat_f.dtype == torch.float32 else feat_f.float()
        transpose_fine = ff.stride(1) != 1
        if transpose_fine:
            ff = ff.contiguous()
        cf_ch = ff.shape[1]
        cm, lf = cfg["coarse_matching"], cfg["loftr_fine"]
        img_h = data["q_hw_i"][0]
        ext_mode = 0 if x3d_ext is None else (2 if x3d_ext["y3d0"] is not None else 1)      # ophip_frame_layout's external_x3d
        pkey = (str(dev), B, N, M, hc, wc, hf, wf, cf_ch, bool(transpose_fine), ext_mode, int(img_h), id(W), bool(lazy))
        plan = self._frame_plans.get(pkey)
        if plan is None:
            d = hip.FrameDesc()
            d.B, d.N, d.M, d.hc, d.wc, d.hf, d.wf, d.cf = B, N, M, hc, wc, hf, wf, cf_ch
            d.lazy_conf = 1 if lazy else 0
            names_c, names_f = self.loftr_coarse.layer_names, self.loftr_fine.layer_names
            d.n_coarse, d.coarse_cross_bits = len(names_c), sum(1 << i for i, n in enumerate(names_c) if n == "cross")
            d.n_fine, d.fine_cross_bits = len(names_f), sum(1 << i for i, n in enumerate(names_f) if n == "cross")
            d.fine_encoder_enable = 1 if lf["enable"] else 0
            d.border_rm = int(cm["border_rm"])
            d.thr, d.scale_c = float(cm["thr"]), float(img_h / hc)
            d.fine_scale = float((lf["window_size"] // 2) * (img_h / hf))
            d.temperature = float(cm["dual_softmax"]["temperature"])
            pe = self._pe_table(hc, wc, dev) if self._pe_enable else None
            d.pe = pe.data_ptr() if pe is not None else None
            d.w_kpt = W["kpt"].data_ptr() if self.kpt_3d_pos_encoding is not None else None
            for li in range(len(names_c)):
                d.w_coarse[li] = W["coarse_x3"][li].data_ptr()
            d.w_fine = W["fine_bf16"].data_ptr()
            L = hip.FrameLayout()
            hip.call("ophip_frame_layout", ctypes.byref(d), 1 if transpose_fine else 0, ext_mode, ctypes.byref(L))
            if len(self._frame_plans) >= 8:
                old_id = self._frame_plans.pop(next(iter(self._frame_plans)))[2]
                self._plan_ids.discard(old_id)
                ops.drop_frame_plan(old_id)
            pid = ops.register_frame_plan(d, L, keep_alive=(W, pe))
            self._plan_ids.add(pid)
            plan = self._frame_plans[pkey] = (d, L, pid, (W, pe))      # W and the table stay alive with the pointers, here and in the registry
        d, L, plan_id = plan[:3]

        prev = self._fine_streams.get(fkey)
        if prev is not None and prev[1] is not None:                      # a stage-by-stage frame before this one: order behind its fine stage
            main.wait_event(prev[1])
            prev[1] = None
        if fkey not in self._fine_streams:
            if len(self._fine_streams) >= 16:
                self._fine_streams.pop(next(iter(self._fine_streams)))
            self._fine_streams[fkey] = [torch.cuda.Stream(device=dev, priority=int(os.environ.get("OPHIP_FINE_PRIO", "0"))), None]
        sfine = self._fine_streams[fkey][0]
        sprep = self._side_stream(self._prep_streams, fkey, dev) if inputs_ready else None
        scopy = self._side_stream(PendingFrame._copy_streams, (dev, main.cuda_stream), dev)
        cap = B * N
        if sprep is not None:
            # the input kernels write into the block on s_prep AHEAD of everything queued on the compute stream, so the block must come
            # from s_prep's pool (the caching allocator orders reuse on the allocating stream only: a block freed on the compute stream
            # with work still pending there could otherwise be overwritten early); the other streams that touch it are recorded, so
            # that it goes back to the pool only after their work is done
            with torch.cuda.stream(sprep):
                blob = torch.empty(L.total, dtype=torch.uint8, device=dev)
            for st in (main, sfine, scopy):
                blob.record_stream(st)
        else:
            blob = torch.empty(L.total, dtype=torch.uint8, device=dev)
            for st in (sfine, scopy):
                blob.record_stream(st)
        nbytes = int(L.result_bytes) if host_copy else 16
        pin = PendingFrame._take_pin((cap, bool(host_copy)), nbytes)
        if transpose_fine:
            fs = (0, 0, 0, 0)                                             # filled in by the callee for its channels-last copy
        else:
            fs = (ff.stride(0), 1, ff.stride(2), ff.stride(3))
        try:
            # the whole frame through ONE custom op (torch.ops.onepose_hip.frame_enqueue -> ophip_frame_enqueue).  The reference's two
            # profiler scopes (coarse_matching.py:122,167) enclose it -- get_coarse_match and its argmax are inside this call; a profiler no
            # longer takes the model off the one-call path -- and the library adds a roctx range per stage and kernel (OPHIP_ROCTX=1).
            with self.profiler.record_function("LoFTR/coarse-matching/get_coarse_match"), \
                    self.profiler.record_function("LoFTR/coarse-matching/get_coarse_match/argmax-conf"):
                oc = x3d_ext or {"x3d": None, "y3d0": None, "kv1": None, "ev": None}
                slot = torch.ops.onepose_hip.frame_enqueue(plan_id, blob, fc, ff, list(fs), kpts_d, desc_in_d, desc_fine_d, oc["x3d"], pin, nbytes,
                                                           main.cuda_stream, sprep.cuda_stream if sprep is not None else 0,
                                                           sfine.cuda_stream, scopy.cuda_stream, qmask, qscale, oc["y3d0"], oc["kv1"],
                                                           oc["ev"].cuda_event if oc["ev"] is not None else 0)
        except Exception:
            # part of the frame may be queued on the side streams already: nothing may touch the block or the pinned buffer again
            # before those streams are idle
            torch.cuda.synchronize(dev)
            PendingFrame._pinned_pool.setdefault((cap, bool(host_copy)), []).append(pin)
            raise
        self._frame_call_pending.add(fkey)
        if lazy:
            f3 = blob[L.feat3d_out:L.feat3d_out + 4 * B * N * 256].view(torch.float32).view(B, N, 256)
            f2 = blob[L.feat2d_out:L.feat2d_out + 4 * B * M * 256].view(torch.float32).view(B, M, 256)
            data["conf_matrix"] = LazyConfMatrix(f3, f2, float(cm["dual_softmax"]["temperature"]), 3, main, qmask)
        else:
            data["conf_matrix"] = blob[L.conf:L.conf + 4 * B * N * M].view(torch.float32).view(B, N, M)
        keep = [fc, ff, kpts_d, desc_in_d, desc_fine_d, x3d_ext, W, qmask, qscale]
        pend = PendingFrame._from_block(self, data, dev, B, N, M, cap, blob, L, slot, pin, host_copy, keep)
        pend._rerun = rerun
        return pend

class LazyConfMatrix:
    """``data["conf_matrix"]`` of the lazy form (``config["hip_conf_matrix"] = "lazy"``): the N x M dual-softmax matrix is not stored
    by the frame; this object keeps the encoder's final rows and computes it on first use -- same kernels, same bits as the eager
    form (``ophip_coarse_match_conf`` with a conf buffer).  ``shape`` / ``dtype`` / ``device`` answer without materialising; indexing,
    attribute access (``.max``, ``.cpu`` ...) and ``torch.*`` functions materialise first (``__torch_function__``).
    Reference: ``utils/coarse_matching.py:115``; its readers at inference time: none (``inference.py:179-180``)."""

    def __init__(self, feat3d, feat2d, temperature, nsplit, stream, query_mask=None):
        self._f3, self._f2, self._temp, self._nsplit, self._stream, self._qmask = feat3d, feat2d, temperature, nsplit, stream, query_mask
        self._t = None
        self.shape = torch.Size((feat3d.shape[0], feat3d.shape[1], feat2d.shape[1]))
        self.dtype, self.device = torch.float32, feat3d.device

    def materialize(self) -> torch.Tensor:
        if self._t is None:
            B, N, M = self.shape
            dev = self.device
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(self._stream)             # the frame's encoder has produced the rows
            conf = torch.empty(B, N, M, device=dev, dtype=torch.float32)
            ws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev, dtype=torch.float32)
            cap = B * N
            ids = torch.empty(3 * cap, dtype=torch.int64, device=dev)
            fl = torch.empty(6 * cap + 3, dtype=torch.float32, device=dev)
            cnt = torch.zeros(4, dtype=torch.int32, device=dev)
            P = hip.ptr
            args = (P(self._f3), P(self._f2), P(fl[:3]), 0, B, N, M, M, self._temp, 0.5, 0, 1.0,
                    P(conf), P(ws), P(ids[:cap], torch.int64), P(ids[cap:2 * cap], torch.int64), P(ids[2 * cap:], torch.int64),
                    P(fl[3:3 + cap]), P(fl[3 + cap:3 + 4 * cap]), P(fl[3 + 4 * cap:3 + 6 * cap]), None, None, P(cnt, torch.int32), self._nsplit)
            if self._qmask is None:
                hip.call("ophip_coarse_match_conf", *args, hip.stream_handle())
            else:
                hip.call("ophip_coarse_match_masked", *args, 1, P(self._qmask, torch.uint8), None, hip.stream_handle())
            self._t = conf
            self._f3 = self._f2 = None
        return self._t

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 3

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __getattr__(self, name):                      # anything a tensor has and this object does not: materialise, then delegate
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __repr__(self):
        return f"LazyConfMatrix(shape={tuple(self.shape)}, materialised={self._t is not None})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(a):
            if isinstance(a, LazyConfMatrix):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(conv(x) for x in a)
            return a
        return func(*conv(tuple(args)), **{k: conv(v) for k, v in (kwargs or {}).items()})


def _result_views(blob, cap):
    """(blob, count int32[1], b_ids int64[cap], mkpts3d f32[cap,3], mkpts2d f32[cap,2]) views of one byte block."""
    o_b, o_3, o_2 = 16, 16 + 8 * cap, 16 + 20 * cap
    return (blob, blob[:4].view(torch.int32), blob[o_b:o_3].view(torch.int64),
            blob[o_3:o_2].view(torch.float32).view(cap, 3), blob[o_2:o_2 + 8 * cap].view(torch.float32).view(cap, 2))


def _result_block(dev, cap):
    return _result_views(torch.empty(16 + 28 * cap, dtype=torch.uint8, device=dev), cap)


class PendingFrame:
    """A batch whose kernels are enqueued but whose match count has not been read yet."""

    _pinned_pool = {}
    _copy_streams = {}          # (device, compute stream) -> side stream for the result read-back

    def __init__(self, model, data, dev, B, N, M, cap, fine_on, want_dbg, bufs, host_copy):
        self.model, self.data, self.dev = model, data, dev
        self.B, self.N, self.M, self.cap, self.fine_on, self.want_dbg = B, N, M, cap, fine_on, want_dbg
        self.bufs = bufs
        self.host = None
        key = (cap, bool(host_copy))
        nbytes = bufs["blob"].numel() if host_copy else 16
        self._pin = PendingFrame._take_pin(key, nbytes)
        self._key, self._host_copy = key, bool(host_copy)
        self._slot = None
        self._rerun = None
        # the D2H of the result block runs on a side stream behind an event: on the compute stream the PCIe round trip
        # (~40 us per frame) would sit between this frame's last kernel and the next frame's first one
        main = torch.cuda.current_stream(dev)
        side = PendingFrame._copy_streams.get((dev, main.cuda_stream))
        if side is None:
            if len(PendingFrame._copy_streams) >= 16:          # keyed on raw stream handles: drop the oldest instead of growing for ever
                PendingFrame._copy_streams.pop(next(iter(PendingFrame._copy_streams)))
            side = PendingFrame._copy_streams[(dev, main.cuda_stream)] = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        self.event = torch.cuda.Event()
        with torch.cuda.stream(side):
            self._pin.copy_(bufs["blob"][:nbytes], non_blocking=True)      # bufs (kept until finish) outlive the copy
            self.event.record(side)
        self.done = False

    @staticmethod
    def _take_pin(key, nbytes):
        pool = PendingFrame._pinned_pool.setdefault(key, [])
        return pool.pop() if pool else torch.empty(nbytes, dtype=torch.uint8).pin_memory()

    @classmethod
    def _from_block(cls, model, data, dev, B, N, M, cap, blob, layout, slot, pin, host_copy, keep):
        """A frame enqueued by ``ophip_frame_enqueue``: outputs are views into its device block, made at ``finish()``."""
        self = cls.__new__(cls)
        self.model, self.data, self.dev = model, data, dev
        self.B, self.N, self.M, self.cap, self.fine_on, self.want_dbg = B, N, M, cap, True, False
        self.bufs, self.host = None, None
        self._block, self._layout, self._slot, self._keep = blob, layout, slot, keep
        self._pin, self._key, self._host_copy = pin, (cap, bool(host_copy)), bool(host_copy)
        self.event = None
        self._rerun = None
        self.done = False
        return self

    def wait(self):
        """Block until this frame's results (and its read-back) are complete; ``finish()`` calls it."""
        self._wait()

    def _wait(self):
        if self._slot is not None:
            hip.call("ophip_frame_wait", self._slot)
        else:
            self.event.synchronize()

    def _block_views(self):
        blob, L, cap = self._block, self._layout, self.cap
        res = blob[L.result:L.result + L.result_bytes]
        _, count, b_ids, mk3d, mk2d = _result_views(res, cap)

        def v(off, nbytes, dt):
            return blob[off:off + nbytes].view(dt)
        return dict(blob=res, b_ids=b_ids, i_ids=v(L.i_ids, 8 * cap, torch.int64), j_ids=v(L.j_ids, 8 * cap, torch.int64),
                    mconf=v(L.mconf, 4 * cap, torch.float32), mk3d=mk3d, mkc=v(L.mkc, 8 * cap, torch.float32).view(cap, 2), count=count,
                    m_bids=v(L.m_bids, 8 * cap, torch.int64), gt_mask=v(L.gt_mask, cap, torch.bool),
                    expec=v(L.expec, 12 * cap, torch.float32).view(cap, 3), mkf=mk2d, dbg_w=None, dbg_3=None, keep=self._keep)

    def _release_pin(self):
        if self._pin is not None:
            pool = PendingFrame._pinned_pool.setdefault(self._key, [])
            if len(pool) < 8:                       # a few frames in flight at most: do not hoard pinned memory
                pool.append(self._pin)
            self._pin = None

    def close(self):
        """Abandon the frame: wait until the side streams (fine stage, read-back) are done with its buffers, then release them.
        Called by ``__del__``, so a frame dropped without ``finish()`` (an exception in the pipeline) cannot hand device blocks
        that are still being read or written back to the caching allocator."""
        if self.done or self._pin is None:
            return
        try:
            self._wait()
        finally:
            self._release_pin()
            self.done = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def finish(self, on_host=None):
        """Wait for this frame (event), read K, fill ``data`` exactly like the reference's ``forward``.  ``on_host(host)``: called with the
        frame's host-side matches (``host_copy=True``: K, 3D points, refined 2D points, b_ids) as soon as they are read -- before the
        device-side views of ``data`` are made -- so that a pipeline hands them to its pose solver ~50 us earlier."""
        if self.done:
            return self.data
        self._wait()                                # the one host wait of the frame
        K = int(self._pin[:4].view(torch.int32)[0])
        if self._rerun is not None and int(self._pin[4:8].view(torch.int32)[0]) != 0:
            # lazy conf_matrix and an exact tie of a row maximum whose first column failed the mutual test: which column the reference
            # takes next can only be read from the stored row -> this frame again, eagerly (identical arithmetic, conf_matrix stored)
            self._release_pin()
            self.done = True
            self.model.lazy_reruns += 1
            again = self._rerun()
            again.finish(on_host)
            self.host = again.host
            return self.data
        B, N, M, cap = self.B, self.N, self.M, self.cap
        if self._host_copy:
            _, _, hb, h3, h2 = _result_views(self._pin, cap)
            self.host = {"K": K, "mkpts_3d_db": h3[:K].numpy().copy(), "mkpts_2d": h2[:K].numpy().copy(), "b_ids": hb[:K].numpy().copy()}
            if on_host is not None:
                on_host(self.host)
        self._release_pin()
        if self.bufs is None:
            self.bufs = self._block_views()
        bf = self.bufs
        data, dev = self.data, self.dev
        b_ids, i_ids, j_ids = bf["b_ids"][:K], bf["i_ids"][:K], bf["j_ids"][:K]
        mconf, mk3d, mkc = bf["mconf"][:K], bf["mk3d"][:K], bf["mkc"][:K]
        data.update({
            "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids,
            "gt_mask": bf["gt_mask"][:K], "m_bids": bf["m_bids"][:K],
            "mkpts_3d_db": mk3d, "mkpts_query_c": mkc, "mconf": mconf,
        })
        self.done = True
        if not self.fine_on:
            data.update({"mkpts_3d_db": data["mkpts_3d_db"], "mkpts_query_f": data["mkpts_query_c"]})
            return data
        data.update({"W": self.model.config["loftr_fine"]["window_size"]})
        if K == 0:
            data.update({"expec_f": torch.empty(0, 3, device=dev), "mkpts_3d_db": mk3d, "mkpts_query_f": mkc})
            return data
        data.update({"expec_f": bf["expec"][:K], "mkpts_3d_db": mk3d, "mkpts_query_f": bf["mkf"][:K]})
        if self.want_dbg:
            data["_fine_win"], data["_fine_f3"] = bf["dbg_w"][:K], bf["dbg_3"][:K]
        return data


def build_model(model_configs, ckpt_path) -> OnePosePlus_model:
    """``src/inference/inference_OnePosePlus.py:30-40``: strip the ``matcher.`` prefix, strict
    load, eval.  The checkpoint is read with ``weights_only=True`` (nothing from the file is
    executed)."""
    model = OnePosePlus_model(model_configs)
    state_dict = torch.load(ckpt_path, map_location="cpu", weights_only=True)["state_dict"]
    for k in list(state_dict.keys()):
        state_dict[k.replace("matcher.", "")] = state_dict.pop(k)
    model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model
