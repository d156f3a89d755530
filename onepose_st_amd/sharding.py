"""Multi-GPU layer of the hot path: frames shard, nothing else communicates.

* :func:`frame_chunk` -- the contiguous chunk of frame indices a rank owns; same partition as the
  reference's Ray evaluation path (``chunk_index(len(dataset), ceil(len / n_workers))``,
  ``src/utils/ray_utils.py:99-106``, ``src/inference/inference_OnePosePlus.py:81-83``).
* :func:`broadcast_object_block` -- the job's single collective: rank 0's weights and the shared
  3D object block (keypoints3d, fine + coarse descriptors) are flattened into one buffer and
  broadcast once (``backend="nccl"`` = RCCL over xGMI on the GPU box; ``gloo`` in the CPU tests).
  Every rank then derives its frame-invariant state locally; per-frame results stay on the rank
  that produced them (K x 5 floats per frame) and are concatenated by the caller in rank order.
"""
from __future__ import annotations

import math

import torch

OBJECT_KEYS = ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")


def frame_chunk(n_frames: int, rank: int, world: int) -> range:
    """Indices of the rank's chunk; the union over ranks is ``range(n_frames)`` in order."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    sub = math.ceil(n_frames / world) if n_frames else 0
    lo = min(n_frames, rank * sub)
    return range(lo, min(n_frames, lo + sub))


def broadcast_object_block(state_dict: dict, obj: dict, device, src: int = 0):
    """Returns ``(state_dict, obj, n_bytes)`` holding rank ``src``'s values on every rank.

    All ranks must call it with tensors of the same shapes (every rank can build the shapes from
    the config; only ``src``'s *values* matter)."""
    import torch.distributed as dist

    names = sorted(state_dict)
    tensors = [state_dict[k] for k in names] + [obj[k] for k in OBJECT_KEYS]
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors]).to(device)
    if dist.get_rank() != src:
        flat.zero_()
    dist.broadcast(flat, src=src)
    out, off = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[off:off + n].view(t.shape).to(t.dtype))
        off += n
    sd = {k: v.cpu() for k, v in zip(names, out[:len(names)])}
    ob = {k: v for k, v in zip(OBJECT_KEYS, out[len(names):])}
    return sd, ob, flat.numel() * 4
