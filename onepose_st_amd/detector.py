"""``LocalFeatureObjectDetector`` -- the 2D object detector in front of the matcher.

Mirrors ``src/local_feature_object_detector/local_feature_2D_detector.py:40-280``: the query frame is matched against ~15
reference views of the object with LoFTR (:class:`onepose_st_amd.loftr.LoFTR_for_OnePose_Plus` on the HIP kernels), each view votes
a box -- the view's four corners through the RANSAC affinity of its matches, or a fixed 1000 x 1000 box around the image centre
when it has fewer than 6 matches -- and the view with the most inliers wins; the box is cropped to 512 x 512 with the intrinsics
updated (``crop_img_by_bbox``: :func:`onepose_st_amd.frameloop.crop_geometry` / ``ophip_crop_resize_gray``).

What differs from the reference by design: the views are matched as ONE batch (one matcher call, one read-back, the per-view RANSACs in
parallel on the host) instead of one call and one synchronisation per view.  What differs by necessity: the reference views are handed over as arrays (its constructor reads a COLMAP model and
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
    def _vote(self, idx: int, mkpts0: np.ndarray, mkpts1: np.ndarray, hw) -> dict:
        """One view's vote (:104-144): the view's corners through the RANSAC affinity of its matches; a fixed 1000 x 1000 box around the
        image centre when the view has fewer than ``min_matches`` matches (or RANSAC finds no model)."""
        H, W = hw
        centre = {"inliers": np.empty((0)), "bbox": np.array([W // 2 - 500, H // 2 - 500, W // 2 + 500, H // 2 + 500])}
        if mkpts0.shape[0] < self.min_matches:
            return centre
        affine, inliers = estimate_affine2d(mkpts0, mkpts1, ransac_reproj_threshold=self.thr)
        if affine is None:                                        # (cv2 returns None when RANSAC finds no model: the reference would raise)
            return centre
        corners = (affine @ self.db_corners_homo[idx]).T.astype(np.int32)            # 4 x 2, truncated like the reference
        lo, hi = corners.min(axis=0), corners.max(axis=0)
        return {"inliers": inliers, "bbox": np.array([lo[0], lo[1], hi[0], hi[1]])}

    @torch.no_grad()
    def match_worker(self, query: torch.Tensor, batched: bool = True) -> dict:
        """``{view index: {"inliers", "bbox"}}`` for every reference view (:89-144).  The reference loops over the views -- one LoFTR call
        and one device -> host copy each; here ALL views go through the matcher as one batch (``image0 [V, 1, H, W]`` against the one
        query: its backbone features once, every kernel launched once over V pairs) and the matches come back in ONE copy; the
        per-view RANSACs then run side by side on host threads (``oppnp_estimate_affine2d`` releases the GIL).  ``batched=False`` keeps
        the view-by-view form (views of different sizes take it as well); both give the same boxes."""
        hw = tuple(query.shape[-2:])
        same = all(tuple(v.shape[-2:]) == hw for v in self.db_imgs)
        if not (batched and same and len(self.db_imgs) > 1):
            out = {}
            for idx, view in enumerate(self.db_imgs):
                pair = {"image0": view, "image1": query}
                self.matcher(pair)
                both = torch.cat([pair["mkpts0_f"], pair["mkpts1_f"]], 1).cpu().numpy()
                out[idx] = self._vote(idx, both[:, :2], both[:, 2:], hw)
            return out
        pair = {"image0": torch.cat(self.db_imgs, 0), "image1": query}
        self.matcher(pair)
        packed = torch.cat([pair["b_ids"].to(torch.float32)[:, None], pair["mkpts0_f"], pair["mkpts1_f"]], 1).cpu().numpy()      # the one read-back
        view_of = packed[:, 0].astype(np.int64)
        # matches arrive in ascending (view, cell) order: a view's matches are one slice
        bounds = np.searchsorted(view_of, np.arange(len(self.db_imgs) + 1))
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(self.db_imgs))) as ex:
            votes = list(ex.map(lambda i: self._vote(i, packed[bounds[i]:bounds[i + 1], 1:3], packed[bounds[i]:bounds[i + 1], 3:5], hw),
                                range(len(self.db_imgs))))
        return dict(enumerate(votes))

    def detect_by_matching(self, query: torch.Tensor) -> np.ndarray:
        """(:146-162): the box of the view with the most inliers; among equals the first view (the reference's stable descending sort)"""
        votes = self.match_worker(query)
        best = max(votes, key=lambda i: (votes[i]["inliers"].sum(), -i))
        return votes[best]["bbox"]

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
