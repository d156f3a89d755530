"""Frame-loop harness and input formats around the matcher.

What the reference does per sequence (``inference.py:100-190``) with the pieces either side of ``OnePosePlus_model``:

* the per-object 3D block comes from ``anno_3d_average.npz`` + ``anno_3d_average_coarse.npz`` (keys ``keypoints3d [N,3]``,
  ``descriptors3d [dim,N]``, ``scores3d [N,1]``; reader ``OnePosePlus_inference_dataset.py:112-169``), truncated to
  ``shape3d`` points by a random index draw when larger, and is kept on the device for the whole sequence;
* frame 0 and every frame after a failed pose (< 20 PnP inliers) get their object box from a detector, every other frame
  from the projection of the 3D box with the previous pose (``local_feature_2D_detector.py:249-266``);
* the box is cropped and resized to 512 x 512 with the intrinsics updated accordingly
  (``local_feature_2D_detector.py:164-190``, ``data_utils.py:249-290``), matched, and solved by RANSAC-PnP
  (``metric_utils.py:121-209`` with reprojection error 7, scale 1000).

Here the crop runs on the GPU from the uploaded uint8 frame (``ophip_crop_resize_gray``), the matcher is the HIP path and
PnP the C++ solver.  ``SequenceRunner`` takes the object detector as a callable ``detector(frame, t) -> box``: the LoFTR 2D-2D
detector of row f-3 is :class:`onepose_st_amd.detector.LocalFeatureObjectDetector` (its ``__call__``).  ``.npz`` files are read
with ``allow_pickle=False``.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hip
from .pnp import ransac_PnP

MIN_INLIERS = 20          # inference.py:152


def load_object_block(avg_anno3d_file: str, device, shape3d: int | None = None, generator: torch.Generator | None = None) -> dict:
    """``read_anno3d`` (``OnePosePlus_inference_dataset.py:112-169``): returns the batch-1 device tensors the model reads
    (``keypoints3d [1,N,3]``, ``descriptors3d_db [1,dim,N]``, ``descriptors3d_coarse_db [1,dim_c,N]``) plus ``scores3d``,
    ``scores3d_coarse`` (host) and ``num_3d_orig``.  With ``shape3d`` smaller than the stored point count the points are
    drawn like ``pad_keypoints3d_random`` / ``pad_features3d_random`` (``data_utils.py:222-246``): ``torch.randint`` with
    replacement, the same index for every array."""
    root, ext = os.path.splitext(avg_anno3d_file)
    fine, coarse = np.load(avg_anno3d_file, allow_pickle=False), np.load(root + "_coarse" + ext, allow_pickle=False)
    for name, z in (("fine", fine), ("coarse", coarse)):
        for k in ("keypoints3d", "descriptors3d", "scores3d") if name == "fine" else ("descriptors3d", "scores3d"):
            if k not in z.files:
                raise KeyError(f"{name} annotation file lacks '{k}'")
    kp = torch.tensor(fine["keypoints3d"], dtype=torch.float32)                 # [N, 3]
    d_f = torch.tensor(fine["descriptors3d"], dtype=torch.float32)              # [dim, N]
    s_f = torch.tensor(fine["scores3d"], dtype=torch.float32)                   # [N, 1]
    d_c = torch.tensor(coarse["descriptors3d"], dtype=torch.float32)
    s_c = torch.tensor(coarse["scores3d"], dtype=torch.float32)
    n = kp.shape[0]
    if kp.dim() != 2 or kp.shape[1] != 3 or d_f.shape[1] != n or d_c.shape[1] != n:
        raise ValueError("annotation arrays disagree on the number of 3D points")
    if shape3d is not None and n > shape3d:
        idx = torch.randint(n, (shape3d,), generator=generator)
        kp, d_f, s_f, d_c, s_c = kp[idx], d_f[:, idx], s_f[idx, :], d_c[:, idx], s_c[idx, :]
    return {"keypoints3d": kp[None].to(device), "descriptors3d_db": d_f[None].to(device), "descriptors3d_coarse_db": d_c[None].to(device),
            "scores3d": s_f, "scores3d_coarse": s_c, "num_3d_orig": n}


def save_object_block(avg_anno3d_file: str, keypoints3d, descriptors3d, scores3d, descriptors3d_coarse, scores3d_coarse) -> None:
    """Writer of the same format (what ``feature_process.py:316-319,646-649`` stores), for tests and synthetic objects."""
    root, ext = os.path.splitext(avg_anno3d_file)
    np.savez(avg_anno3d_file, keypoints3d=np.asarray(keypoints3d, np.float32), descriptors3d=np.asarray(descriptors3d, np.float32),
             scores3d=np.asarray(scores3d, np.float32))
    np.savez(root + "_coarse" + ext, keypoints3d=np.asarray(keypoints3d, np.float32), descriptors3d=np.asarray(descriptors3d_coarse, np.float32),
             scores3d=np.asarray(scores3d_coarse, np.float32))


def crop_geometry(bbox, K, crop_size: int = 512):
    """``crop_img_by_bbox`` without the pixels: ``(K_crop [3,3], trans [3,3])`` for box ``[x0, y0, x1, y1]``.  The reference
    composes two ``get_affine_transform`` warps (``data_utils.py:32-62``): a shift of the box to the origin, then an
    isotropic scale ``crop_size / box_width`` that keeps the crop's vertical centre in the middle of the output."""
    x0, y0, x1, y1 = [float(v) for v in bbox]
    wb, hb = x1 - x0, y1 - y0
    if wb <= 0 or hb <= 0:
        raise ValueError("empty bounding box")
    s = crop_size / wb
    trans = np.array([[s, 0.0, -s * x0], [0.0, s, 0.5 * crop_size - s * (y0 + 0.5 * hb)], [0.0, 0.0, 1.0]])
    K = np.asarray(K, dtype=np.float64)
    return trans @ K[:3, :3], trans


def project_bbox(K, pose, bbox3d) -> np.ndarray:
    """``previous_pose_detect`` (``local_feature_2D_detector.py:263-266``): int32 ``[x0, y0, x1, y1]`` of the projected 3D box."""
    K, pose, pts = np.asarray(K, np.float64)[:3, :3], np.asarray(pose, np.float64)[:3, :4], np.asarray(bbox3d, np.float64).reshape(-1, 3)
    cam = pose[:, :3] @ pts.T + pose[:, 3:4]
    uv = K @ cam
    uv = (uv[:2] / uv[2:]).T
    return np.concatenate([uv.min(axis=0), uv.max(axis=0)]).astype(np.int32)


def crop_query(frame_u8: torch.Tensor, bbox, crop_size: int = 512) -> torch.Tensor:
    """uint8 ``[H, W]`` frame on the HIP device -> ``[1, 1, S, S]`` float query image in [0, 1] (``ophip_crop_resize_gray``)."""
    if frame_u8.dtype != torch.uint8 or frame_u8.dim() != 2 or not frame_u8.is_cuda:
        raise hip.HipLibraryError("crop_query needs a uint8 [H, W] frame on the HIP device (no CPU fallback)")
    frame_u8 = frame_u8.contiguous()
    out = torch.empty(1, 1, crop_size, crop_size, dtype=torch.float32, device=frame_u8.device)
    x0, y0, x1, y1 = [int(v) for v in bbox]
    hip.call("ophip_crop_resize_gray", hip.ptr(frame_u8, torch.uint8), frame_u8.shape[0], frame_u8.shape[1], x0, y0, x1, y1, crop_size,
             hip.ptr(out), hip.stream_handle())
    return out


class SequenceRunner:
    """The per-sequence loop of ``inference.py:136-190``.

    ``model``: an ``OnePosePlus_model`` on the device; ``object_block``: :func:`load_object_block`; ``K``: full-frame
    intrinsics; ``bbox3d [8, 3]``; ``detector(frame_u8_host, index) -> [x0, y0, x1, y1]``: the object detector used on
    frame 0 and after a failed pose.  ``run(frames)`` takes host uint8 ``[H, W]`` arrays and returns one record per frame:
    ``pose [3,4]``, ``inliers``, ``bbox``, ``K_crop``, ``trans``, ``num_matches``, ``redetected``.

    Within a sequence frame t + 1's crop depends on frame t's pose, so frames are processed strictly in order; the next
    frame's upload is queued on a side stream while the current one is matched.
    """

    def __init__(self, model, object_block: dict, K, bbox3d, detector, crop_size: int = 512, pnp_reprojection_error: float = 7,
                 pnp_scale: float = 1000, min_inliers: int = MIN_INLIERS, crop_fn=crop_query):
        self.model, self.block, self.crop_fn = model, object_block, crop_fn
        self.K, self.bbox3d, self.detector = np.asarray(K, np.float64), np.asarray(bbox3d, np.float64), detector
        self.crop_size, self.reproj, self.scale, self.min_inliers = crop_size, pnp_reprojection_error, pnp_scale, min_inliers
        self.device = object_block["keypoints3d"].device
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def _upload(self, frame):
        host = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))
        if self._copy_stream is None:
            return host.to(self.device), None
        with torch.cuda.stream(self._copy_stream):
            dev = host.pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return dev, ev

    def run(self, frames):
        frames = list(frames)
        records = []
        nxt = self._upload(frames[0]) if frames else None
        prev = None                                   # (pose, inliers) of the previous frame
        for t, frame in enumerate(frames):
            dev_frame, ev = nxt
            nxt = self._upload(frames[t + 1]) if t + 1 < len(frames) else None
            redetect = prev is None or len(prev[1]) < self.min_inliers
            bbox = np.asarray(self.detector(frame, t) if redetect else project_bbox(self.K, prev[0], self.bbox3d)).astype(np.int32)
            if bbox[2] <= bbox[0] or bbox[3] <= bbox[1]:      # degenerate projection: treat as a lost track
                redetect = True
                bbox = np.asarray(self.detector(frame, t)).astype(np.int32)
            K_crop, trans = crop_geometry(bbox, self.K, self.crop_size)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            data = {"query_image": self.crop_fn(dev_frame, bbox, self.crop_size), "keypoints3d": self.block["keypoints3d"],
                    "descriptors3d_db": self.block["descriptors3d_db"], "descriptors3d_coarse_db": self.block["descriptors3d_coarse_db"]}
            with torch.no_grad():
                self.model(data)
            mk3d, mk2d = data["mkpts_3d_db"].cpu().numpy(), data["mkpts_query_f"].cpu().numpy()
            pose, _, inliers = ransac_PnP(K_crop, mk2d, mk3d, scale=self.scale, pnp_reprojection_error=self.reproj,
                                          img_hw=[self.crop_size, self.crop_size], use_pycolmap_ransac=True)
            prev = (pose, inliers)
            records.append({"pose": pose, "inliers": inliers, "bbox": bbox, "K_crop": K_crop, "trans": trans,
                            "num_matches": int(mk2d.shape[0]), "redetected": bool(redetect)})
        return records
