"""``LocalFeatureObjectDetector`` -- the 2D object detector in front of the matcher.

Mirrors ``src/local_feature_object_detector/local_feature_2D_detector.py:40-280``: the query frame is matched against ~15
reference views of the object with LoFTR (:class:`onepose_st_amd.loftr.LoFTR_for_OnePose_Plus` on the HIP kernels), each view votes
a box -- the view's four corners through the RANSAC affinity of its matches, or a fixed 1000 x 1000 box around the image centre
when it has fewer than 6 matches -- and the view with the most inliers wins; the box is cropped to 512 x 512 with the intrinsics
updated (``crop_img_by_bbox``: :func:`onepose_st_amd.frameloop.crop_geometry` / ``ophip_crop_resize_gray``).

What differs from the reference, by necessity: the reference views are handed over as arrays (its constructor reads a COLMAP model and
decodes images with ``cv2`` / ``natsort``, neither of which exists here); ``cv2.estimateAffine2D`` is the build's own RANSAC
(``oppnp_estimate_affine2d``, parity unpinned).  The control flow, thresholds, integer truncations and the tie rule (first view
among equals) are the reference's.  ``detect`` plugs into :class:`onepose_st_amd.frameloop.SequenceRunner` as its ``detector``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import frameloop
from .pnp import estimate_affine2d


def sample_reference_views(n_images: int, n_ref_view: int = 15) -> list:
    """``load_ref_view_images`` (``local_feature_2D_detector.py:66-76``): every ``len // n_ref_view``-th image starting at index 1 of
    the naturally sorted image list."""
    gap = n_images // n_ref_view
    if gap < 1:
        raise ValueError(f"{n_images} reference images for n_ref_view = {n_ref_view}: the reference's sample gap would be 0")
    return list(range(1, n_images, gap))


class LocalFeatureObjectDetector:
    def __init__(self, matcher, db_imgs, device=None, min_matches: int = 6, ransac_reproj_threshold: float = 6.0):
        """``matcher``: a ``LoFTR_for_OnePose_Plus`` on the device; ``db_imgs``: the reference views, grayscale ``[H, W]`` uint8 arrays
        (or float tensors in [0, 1]) -- already sampled (:func:`sample_reference_views`)."""
        self.matcher = matcher
        self.device = torch.device(device) if device is not None else next(matcher.parameters()).device
        self.min_matches, self.thr = int(min_matches), float(ransac_reproj_threshold)
        self.db_imgs, self.db_corners_homo = [], []
        for im in db_imgs:
            t = torch.as_tensor(np.asarray(im)) if not torch.is_tensor(im) else im
            t = (t.float() / 255.0) if t.dtype == torch.uint8 else t.float()
            if t.dim() != 2:
                raise ValueError("reference views must be [H, W] grayscale images")
            self.db_imgs.append(t[None, None].contiguous().to(self.device))          # torch.from_numpy(img)[None][None] / 255.0
            H, W = t.shape
            self.db_corners_homo.append(np.array([[0, 0, 1], [W, 0, 1], [0, H, 1], [W, H, 1]], dtype=np.float64).T)      # 3 x 4

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def match_worker(self, query: torch.Tensor) -> dict:
        """``match_worker`` (:89-144): one LoFTR call per reference view; ``{view index: {"inliers", "bbox"}}``"""
        results = {}
        H, W = query.shape[-2:]
        for idx, db_img in enumerate(self.db_imgs):
            match_data = {"image0": db_img, "image1": query}
            self.matcher(match_data)
            mkpts0 = match_data["mkpts0_f"].cpu().numpy()
            mkpts1 = match_data["mkpts1_f"].cpu().numpy()
            if mkpts0.shape[0] < self.min_matches:                # failed view: a fixed box around the image centre
                cx, cy = W // 2, H // 2
                results[idx] = {"inliers": np.empty((0)), "bbox": np.array([cx - 500, cy - 500, cx + 500, cy + 500])}
                continue
            affine, inliers = estimate_affine2d(mkpts0, mkpts1, ransac_reproj_threshold=self.thr)
            if affine is None:                                    # (cv2 returns None when RANSAC finds no model: the reference would raise)
                cx, cy = W // 2, H // 2
                results[idx] = {"inliers": np.empty((0)), "bbox": np.array([cx - 500, cy - 500, cx + 500, cy + 500])}
                continue
            bbox = (affine @ self.db_corners_homo[idx]).T.astype(np.int32)          # 4 x 2, truncated like the reference
            left_top, right_bottom = np.min(bbox, axis=0), np.max(bbox, axis=0)
            w, h = right_bottom - left_top
            off = 0.0
            results[idx] = {"inliers": inliers,
                            "bbox": np.array([left_top[0] - int(w * off), left_top[1] - int(h * off), right_bottom[0] + int(w * off), right_bottom[1] + int(h * off)])}
        return results

    def detect_by_matching(self, query: torch.Tensor) -> np.ndarray:
        """(:146-162): the box of the view with the most inliers (stable sort: the first view among equals)"""
        res = self.match_worker(query)
        order = [k for k, _ in sorted(res.items(), reverse=True, key=lambda item: item[1]["inliers"].sum())]
        return res[order[0]]["bbox"]

    def detect(self, query_img, K, crop_size: int = 512):
        """(:208-247) ``query_img``: the full frame, uint8 ``[H, W]`` (host array or device tensor).  Returns ``(bbox, crop [1, 1, S, S]
        float on the device, K_crop, transformation)`` like the reference."""
        frame = torch.as_tensor(np.ascontiguousarray(query_img)) if not torch.is_tensor(query_img) else query_img
        if frame.dtype != torch.uint8 or frame.dim() != 2:
            raise ValueError("detect: a uint8 [H, W] grayscale frame")
        frame = frame.to(self.device)
        query = (frame.float() / 255.0)[None, None].contiguous()
        bbox = np.asarray(self.detect_by_matching(query)).astype(np.int32)
        K_crop, trans = frameloop.crop_geometry(bbox, K, crop_size)
        crop = frameloop.crop_query(frame, bbox, crop_size)
        return bbox, crop, K_crop, trans

    def __call__(self, frame, index=None):
        """the ``detector(frame, t) -> [x0, y0, x1, y1]`` hook of :class:`onepose_st_amd.frameloop.SequenceRunner`"""
        frame_t = torch.as_tensor(np.ascontiguousarray(frame)).to(self.device)
        query = (frame_t.float() / 255.0)[None, None].contiguous()
        return np.asarray(self.detect_by_matching(query)).astype(np.int32)

    def previous_pose_detect(self, K, pre_pose, bbox3D_corner):
        """(:249-266) the box of the projected 3D bounding box under the previous pose"""
        return frameloop.project_bbox(K, pre_pose, bbox3D_corner)
