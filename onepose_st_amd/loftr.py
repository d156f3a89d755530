"""``LoFTR_for_OnePose_Plus`` -- the 2D-2D matcher behind the object detector on the HIP kernels.

Mirrors the interface of ``src/KeypointFreeSfM/loftr_for_sfm/loftr.py:16-167``: ``LoFTR_for_OnePose_Plus(config,
enable_fine_matching=True)``, ``state_dict`` keys ``backbone.*`` / ``loftr_coarse.layers.{0..7}.*`` / ``loftr_fine.layers.{0,1}.*``
(what ``build_2D_match_model`` loads with ``strict=True``, ``local_feature_2D_detector.py:24-37``), ``forward(data)`` mutating
``data`` (``image0``, ``image1`` in; ``hw*_i/c/f``, ``b_ids``, ``i_ids``, ``j_ids``, ``mconf``, ``mkpts0_c``, ``mkpts1_c``,
``expec_f``, ``mkpts0_f``, ``mkpts1_f``, ``conf_matrix`` out).  ``default_cfg`` restates ``loftr_for_onepose_plus_cfg.py:10-50``.

The reference imports the arithmetic from ``submodules/LoFTR`` (un-vendored): it is taken here from the published zju3dv/LoFTR
definition (``oracle/loftr_oracle.py`` restates it on the CPU; parity unpinned).  Kernels:

* backbone ``ResNetFPN_8_2``: ``backbone_hip.HipBackbone`` (``ophip_conv2d_bf16``), positional encoding in the last epilogue;
* 8 coarse layers ``[self, cross] x 4``: ``ophip_encoder_layer_x3w8``.  LoFTR's ``cross`` is sequential (image 1 attends to the
  UPDATED image 0), so a cross layer is two ONE-stream launches (``ophip_encoder_layer_x3w8_streams``): image 0's rows against image 1,
  then image 1's rows against the new image 0 (rounds 3-4 ran the two-stream kernel twice and discarded half of each result);
* batch: ``image0 [V, 1, H, W]`` against ``image1 [1 or V, 1, H, W]`` in ONE call -- the detector's ~15 reference views against one query
  frame (the query's backbone features are computed once); ``b_ids`` names the pair of every match;
* dual-softmax + mutual-nearest between the two grids: ``ophip_coarse_match_2d`` (temperature exactly 0.1, all-sides border);
* fine stage, window 9 on both images, batched over all matches: ``csrc/loftr_fine.hip``.

Masks / scales / provided coarse matches (``mask0``, ``scale0``, ``mkpts0_c`` inputs) and the feature-extraction kwargs are the
SfM pipeline's (out of scope) and raise ``NotImplementedError``.  No CPU fallback.
"""
from __future__ import annotations

import copy
import ctypes

import torch
import torch.nn as nn

from . import hip, host_math, packing
from .backbone import build_backbone
from .backbone_hip import HipBackbone, pack_backbone

default_cfg = {
    "backbone_type": "ResNetFPN", "resolution": (8, 2), "fine_window_size": 9, "fine_concat_coarse_feat": False,
    "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]},
    "coarse": {"d_model": 256, "d_ffn": 256, "nhead": 8, "layer_names": ["self", "cross"] * 4, "attention": "linear", "temp_bug_fix": False},
    "match_coarse": {"thr": 0.2, "border_rm": 2, "match_type": "dual_softmax", "dsmax_temperature": 0.1, "skh_iters": 3,
                     "skh_init_bin_score": 1.0, "skh_prefilter": True, "train_coarse_percent": 0.4, "train_pad_num_gt_min": 200},
    "fine": {"d_model": 128, "d_ffn": 128, "nhead": 8, "layer_names": ["self", "cross"] * 1, "attention": "linear"},
}


class _Layer(nn.Module):
    """parameter holder with the key layout of LoFTR's ``LoFTREncoderLayer``"""

    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj = (nn.Linear(d, d, bias=False) for _ in range(3))
        self.merge = nn.Linear(d, d, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d, bias=False), nn.Identity(), nn.Linear(2 * d, d, bias=False))
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer_names = list(cfg["layer_names"])
        self.d_model, self.nhead = cfg["d_model"], cfg["nhead"]
        self.layers = nn.ModuleList([_Layer(self.d_model) for _ in self.layer_names])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class LoFTR_for_OnePose_Plus(nn.Module):
    def __init__(self, config=None, enable_fine_matching=True):
        super().__init__()
        config = copy.deepcopy(default_cfg) if config is None else config
        self.config = config
        self.enable_fine_matching = enable_fine_matching
        if config["backbone_type"] != "ResNetFPN" or tuple(config["resolution"]) != (8, 2):
            raise NotImplementedError("LoFTR backbone: ResNetFPN 8 -> 2 only")
        cc, cf, mc = config["coarse"], config["fine"], config["match_coarse"]
        if cc["d_model"] != 256 or cc["nhead"] != 8 or cf["d_model"] != 128 or cf["nhead"] != 8:
            raise NotImplementedError("HIP kernels are specialised for d_model 256 / 128 with 8 heads")
        if cc["attention"] != "linear" or cf["attention"] != "linear":
            raise NotImplementedError("attention: linear")
        if mc["match_type"] != "dual_softmax":
            raise NotImplementedError("match_coarse.match_type: dual_softmax")
        if cc["temp_bug_fix"]:
            raise NotImplementedError("temp_bug_fix: the reference's detector config runs the original (floor-division) position table")
        if config["fine_concat_coarse_feat"]:
            raise NotImplementedError("fine_concat_coarse_feat")
        W = int(config["fine_window_size"])
        if W % 2 == 0 or W * W > 128:
            raise ValueError("fine_window_size must be odd and at most 11")
        for n in list(cc["layer_names"]) + list(cf["layer_names"]):
            if n not in ("self", "cross"):
                raise KeyError(n)
        self.backbone = build_backbone({"type": "ResNetFPN", "resolution": [8, 2],
                                        "resnetfpn": {"block_type": "BasicBlock", "initial_dim": config["resnetfpn"]["initial_dim"],
                                                      "block_dims": list(config["resnetfpn"]["block_dims"]), "output_layers": [3, 1]}})
        self.loftr_coarse = _Encoder(cc)
        self.loftr_fine = _Encoder(cf)
        self._packed = None
        self._pe = {}
        # optional ``hook(fc0 [1, L0, 256], ff0 [hf0 * wf0, 128], fc1, ff1) -> the same four``: the backbone-output boundary (coarse rows
        # with the positional encoding added, fine maps channels-last) -- the counterpart of a forward hook on the reference's backbone.
        # (a batched call hands it ``fc0 [V, L0, 256], ff0 [V, hf0 * wf0, 128], fc1 [1 or V, ...], ff1`` and takes the same back)
        self.feature_hook = None

    # ------------------------------------------------------------------------------------------
    def _blocks(self, device):
        params = list(self.parameters()) + list(self.buffers())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is None or self._packed[0] != key:
            sd = self.state_dict()
            bb = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
            fine = []
            for i in range(len(self.loftr_fine.layer_names)):
                p = f"loftr_fine.layers.{i}."
                fine.append({n: packing.pack_linear_x3(sd[p + k]).to(device) for n, k in
                             (("q", "q_proj.weight"), ("k", "k_proj.weight"), ("v", "v_proj.weight"), ("m", "merge.weight"),
                              ("w0", "mlp.0.weight"), ("w2", "mlp.2.weight"))})
                fine[-1].update({n: sd[p + n.replace("_", ".")].detach().float().contiguous().to(device)
                                 for n in ("norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias")})
            self._packed = (key, {
                "backbone": pack_backbone(bb, device),
                "coarse": [packing.pack_coarse_layer_x3w8(sd, f"loftr_coarse.layers.{i}.").to(device) for i in range(len(self.loftr_coarse.layer_names))],
                "fine": fine,
            })
        return self._packed[1]

    def _pe_table(self, h, w, device):
        k = (h, w, str(device))
        if k not in self._pe:
            pe = host_math.sinusoid_table(self.config["coarse"]["d_model"], h, w, (256, 256))        # the floor-division table (temp_bug_fix False)
            self._pe[k] = pe.flatten(1).t().contiguous().to(device)
        return self._pe[k]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, data, **kwargs):
        if self.training:
            raise NotImplementedError("inference only")
        if kwargs.get("extract_coarse_feature") or kwargs.get("extract_fine_feature"):
            raise NotImplementedError("feature extraction kwargs belong to the SfM pipeline (out of scope)")
        for k in ("mask0", "mask1", "scale0", "scale1", "mkpts0_c", "mkpts1_c"):
            if k in data:
                raise NotImplementedError(f"'{k}' input: not used by the object detector (local_feature_2D_detector.py:93-94)")
        img0, img1 = data["image0"], data["image1"]
        if not img0.is_cuda or not img1.is_cuda:
            raise hip.HipLibraryError("LoFTR_for_OnePose_Plus runs on the HIP device only (no CPU fallback)")
        V = img0.size(0)
        if img1.size(0) not in (1, V):
            raise ValueError(f"image1: batch {img1.size(0)} against image0's {V} (expected 1 -- one query for every pair -- or {V})")
        shared1 = img1.size(0) == 1 and V > 1          # one query image against V views: its features are computed once
        hip.load()
        call, P, S = hip.call, hip.ptr, hip.stream_handle()
        dev = img0.device
        data.update({"bs": V, "hw0_i": img0.shape[2:], "hw1_i": img1.shape[2:]})
        Wb = self._blocks(dev)
        bbk = HipBackbone("bf16x3")

        def features(img):
            H, W = img.shape[2:]
            fc, ff = bbk.forward(Wb["backbone"], img, self._pe_table(H // 8, W // 8, dev))
            return fc, ff, (H // 8, W // 8), (H // 2, W // 2)
        if img0.shape[2:] == img1.shape[2:]:
            fc, ff, hwc, hwf = features(torch.cat([img0, img1], 0))
            fc0, fc1, ff0, ff1 = fc[:V], fc[V:], ff[:V], ff[V:]
            hw0_c = hw1_c = hwc
            hw0_f = hw1_f = hwf
        else:
            fc0, ff0, hw0_c, hw0_f = features(img0)
            fc1, ff1, hw1_c, hw1_f = features(img1)
        data.update({"hw0_c": torch.Size(hw0_c), "hw1_c": torch.Size(hw1_c), "hw0_f": torch.Size(hw0_f), "hw1_f": torch.Size(hw1_f)})
        if self.feature_hook is not None:
            # one pair: the fine maps without the batch axis (the hook's form since round 3); a batch: everything with it
            if V == 1:
                fc0, f0h, fc1, f1h = self.feature_hook(fc0, ff0[0], fc1, ff1[0])
            else:
                fc0, f0h, fc1, f1h = self.feature_hook(fc0, ff0, fc1, ff1)
            ff0, ff1 = (f0h[None] if f0h.dim() == 2 else f0h), (f1h[None] if f1h.dim() == 2 else f1h)
            if fc0.shape[0] != V or fc1.shape[0] not in (1, V) or ff0.shape[0] != V or ff1.shape[0] != fc1.shape[0]:
                raise ValueError("feature_hook: batch sizes of the returned features do not match the call")
            shared1 = fc1.shape[0] == 1 and V > 1
        L0, L1 = hw0_c[0] * hw0_c[1], hw1_c[0] * hw1_c[1]

        # ---- coarse transformer: self = one two-stream launch; cross = two one-stream launches (sequential semantics) -------------------
        x0 = fc0.contiguous()
        x1 = (fc1.expand(V, -1, -1) if shared1 else fc1).contiguous()
        ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(V, L0, L1), dtype=torch.uint8, device=dev)
        for li, name in enumerate(self.loftr_coarse.layer_names):
            w = Wb["coarse"][li]
            if name == "self":
                b0, b1 = torch.empty_like(x0), torch.empty_like(x1)
                call("ophip_encoder_layer_x3w8", P(x0), P(x1), P(b0), P(b1), V, L0, L1, P(w, None), None, 0, 0, 0, P(ws, None), S)
                x0, x1 = b0, b1
            else:
                n0 = torch.empty_like(x0)              # image 0 against image 1
                call("ophip_encoder_layer_x3w8_streams", P(x0), P(x1), P(n0), None, V, L0, L1, P(w, None), 1, 1, P(ws, None), S)
                n1 = torch.empty_like(x1)              # image 1 against the UPDATED image 0
                call("ophip_encoder_layer_x3w8_streams", P(n0), P(x1), None, P(n1), V, L0, L1, P(w, None), 1, 2, P(ws, None), S)
                x0, x1 = n0, n1

        # ---- coarse matching between the two grids -----------------------------------------------------------------------------
        mc = self.config["match_coarse"]
        scale = img0.shape[2] / hw0_c[0]
        ii = torch.arange(L0, device=dev)
        pts0 = torch.stack([(ii % hw0_c[1]).float() * scale, (ii // hw0_c[1]).float() * scale, torch.zeros(L0, device=dev)], 1)[None].contiguous()
        cap = V * L0
        conf = torch.empty(V, L0, L1, device=dev)
        cws = torch.empty(hip.load().ophip_coarse_workspace_floats(V, L0, L1), device=dev)
        ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(4)]
        mconf, mk0, mk1c = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
        gt_mask = torch.empty(cap, dtype=torch.bool, device=dev)
        count = torch.zeros(4, dtype=torch.int32, device=dev)
        call("ophip_coarse_match_2d", P(x0), P(x1), P(pts0), 0, V, L0, L1, hw0_c[1], hw1_c[1], float(mc["dsmax_temperature"]), float(mc["thr"]),
             int(mc["border_rm"]), float(scale), P(conf), P(cws), P(ids[0], torch.int64), P(ids[1], torch.int64), P(ids[2], torch.int64),
             P(mconf), P(mk0), P(mk1c), P(ids[3], torch.int64), P(gt_mask, torch.bool), P(count, torch.int32), 3, S)
        K = int(count[0].item())                           # the detector reads the matches on the host right after: one sync here
        b_ids, i_ids, j_ids = ids[0][:K], ids[1][:K], ids[2][:K]
        mk0c, mk1c = mk0[:K, :2].contiguous(), mk1c[:K].contiguous()
        data.update({"conf_matrix": conf, "b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "m_bids": ids[3][:K], "gt_mask": gt_mask[:K],
                     "mconf": mconf[:K], "mkpts0_c": mk0c, "mkpts1_c": mk1c})
        if not self.enable_fine_matching:
            data.update({"mkpts0_f": mk0c, "mkpts1_f": mk1c})
            return
        Wf = int(self.config["fine_window_size"])
        WW = Wf * Wf
        if K == 0:
            data.update({"expec_f": torch.empty(0, 3, device=dev), "mkpts0_f": mk0c, "mkpts1_f": mk1c})
            return
        # ---- fine stage: windows on both images, two-stream fine transformer, correlation + soft-argmax ------------------------
        stride = hw0_f[0] // hw0_c[0]
        f0, f1 = torch.empty(K, WW, 128, device=dev), torch.empty(K, WW, 128, device=dev)
        ff0c, ff1c = ff0.contiguous(), ff1.contiguous()          # [V or 1][hf * wf][128] channels-last
        call("ophip_fine2_gather_b", P(ff0c), ff0c.stride(0) if ff0c.shape[0] > 1 else 0, P(b_ids, torch.int64), hw0_f[0], hw0_f[1],
             P(i_ids, torch.int64), K, hw0_c[1], stride, Wf, P(f0), S)
        call("ophip_fine2_gather_b", P(ff1c), ff1c.stride(0) if ff1c.shape[0] > 1 else 0, P(b_ids, torch.int64), hw1_f[0], hw1_f[1],
             P(j_ids, torch.int64), K, hw1_c[1], hw1_f[0] // hw1_c[0], Wf, P(f1), S)
        T = K * WW

        def lin(xa, w, nout, xb=None, relu=False):
            y = torch.empty(T, nout, device=dev)
            call("ophip_rows_linear_x3", P(xa), xa.shape[-1], P(xb), xb.shape[-1] if xb is not None else 0, T, P(w, None), nout, 1 if relu else 0, P(y), S)
            return y

        def fine_layer(x, src, w):
            q, k, v = lin(x, w["q"], 128), lin(src, w["k"], 128), lin(src, w["v"], 128)
            msg = torch.empty(T, 128, device=dev)
            call("ophip_fine2_attention", P(q), P(k), P(v), K, WW, WW, P(msg), S)
            m = lin(msg, w["m"], 128)
            call("ophip_rows_layernorm128", P(m), P(w["norm1_weight"]), P(w["norm1_bias"]), None, T, P(m), S)
            h = lin(x.view(T, 128), w["w0"], 256, xb=m, relu=True)
            o = lin(h, w["w2"], 128)
            y = torch.empty(K, WW, 128, device=dev)
            call("ophip_rows_layernorm128", P(o), P(w["norm2_weight"]), P(w["norm2_bias"]), P(x), T, P(y), S)
            return y
        for li, name in enumerate(self.loftr_fine.layer_names):
            w = Wb["fine"][li]
            if name == "self":
                f0, f1 = fine_layer(f0, f0, w), fine_layer(f1, f1, w)
            else:
                f0 = fine_layer(f0, f1, w)
                f1 = fine_layer(f1, f0, w)
        expec, mk1f = torch.empty(K, 3, device=dev), torch.empty(K, 2, device=dev)
        call("ophip_fine2_match", P(f0), P(f1), P(mk1c), K, Wf, float((Wf // 2) * (img0.shape[2] / hw0_f[0])), P(expec), P(mk1f), S)
        data.update({"expec_f": expec, "mkpts0_f": mk0c, "mkpts1_f": mk1f})
        if kwargs.get("_debug"):
            data["_fine_f0"], data["_fine_f1"], data["_feat_c0"], data["_feat_c1"] = f0, f1, x0, x1


def build_2D_match_model(args: dict) -> LoFTR_for_OnePose_Plus:
    """``local_feature_2D_detector.py:24-37``: LoFTR with the default config, checkpoint loaded strictly (``weights_only=True``:
    nothing from the file is executed), ``eval()``.  The reference also seeds the global generators (``pl.seed_everything``): inference
    draws no random numbers, so there is nothing to seed here."""
    if args["method"] != "LoFTR":
        raise NotImplementedError
    matcher = LoFTR_for_OnePose_Plus(config=copy.deepcopy(default_cfg))
    state_dict = torch.load(args["weight_path"], map_location="cpu", weights_only=True)["state_dict"]
    for k in list(state_dict.keys()):
        state_dict[k.replace("matcher.", "")] = state_dict.pop(k)
    matcher.load_state_dict(state_dict, strict=True)
    matcher.eval()
    return matcher
