"""N > 1 path on CPU: world_size 2, gloo.  Rank 1 starts from different weights / object block and must
end with rank 0's; frame chunks partition the sequence like the reference's chunk_index."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from onepose_st_amd.config import default_config
from onepose_st_amd.sharding import OBJECT_KEYS, broadcast_object_block, frame_chunk
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict


def test_frame_chunk_matches_reference_partition():
    for n, w in [(100, 8), (7, 2), (5, 8), (0, 2), (16, 4), (101, 3)]:
        sub = int(np.ceil(n / w)) if n else 1
        ref = [list(range(i, min(n, i + sub))) for i in range(0, n, sub)]          # chunk_index(n, ceil(n / w))
        got = [list(frame_chunk(n, r, w)) for r in range(w)]
        assert [c for c in got if c] == ref
        assert sum(got, []) == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = default_config()
    sd = make_synthetic_state_dict(seed=rank, config=cfg, backbone=False)          # rank 1 has the WRONG weights
    obj = make_synthetic_inputs(sd, n_points=64, image_hw=(32, 40), n_plant=10, seed=1 + rank, config=cfg)
    sd2, ob2, nbytes = broadcast_object_block(sd, obj, torch.device("cpu"))
    cs = [float(sd2[k].double().sum()) for k in sorted(sd2)] + [float(ob2[k].double().sum()) for k in OBJECT_KEYS]
    frames = list(frame_chunk(9, rank, world))
    q.put((rank, cs, nbytes, frames, {k: tuple(v.shape) for k, v in ob2.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, cs0, nb0, f0, sh0), (_, cs1, nb1, f1, sh1) = res
    assert cs0 == cs1 and nb0 == nb1 and sh0 == sh1             # bit-identical payload on both ranks
    cfg = default_config()
    want = make_synthetic_state_dict(seed=0, config=cfg, backbone=False)
    assert cs0[:len(want)] == [float(want[k].double().sum()) for k in sorted(want)]
    assert f0 == [0, 1, 2, 3, 4] and f1 == [5, 6, 7, 8]
