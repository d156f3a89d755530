"""One process per GPU, started by the driver process itself.

The reference fans its frame chunks out from the driver process (Ray tasks,
``src/inference/inference_OnePosePlus.py:81-98``).  Here the driver starts one *child process* per rank
with the ``torch.distributed`` rendezvous variables set (``RANK``, ``LOCAL_RANK``, ``WORLD_SIZE``,
``MASTER_ADDR`` = 127.0.0.1, ``MASTER_PORT``), relays rank 0's stdout and returns the worst exit code.

The parent never touches the GPU and never replaces itself (no ``exec``): a process that has initialised
HIP must not be re-exec'ed on this pool, and a parent holding a HIP context would be one more process on
the card.  Pure host logic, covered on CPU by ``tests/test_launch.py``.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def visible_gpu_count(environ=None, kfd_root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process could use, WITHOUT bringing up the HIP runtime (``torch.cuda.device_count()`` falls back to
    ``hipGetDeviceCount`` when its amdsmi path fails, and a parent that has initialised HIP must not fan out children on
    this pool): the ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` list when one is set, else the KFD topology nodes that
    have SIMDs.  ``None`` when neither source is readable (the caller then lets a rank fail with its own message)."""
    env = os.environ if environ is None else environ
    for key in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if env.get(key, "").strip():
            return len([t for t in env[key].split(",") if t.strip() != ""])
    try:
        n = 0
        for node in os.listdir(kfd_root):
            props = open(os.path.join(kfd_root, node, "properties")).read().split()
            kv = dict(zip(props[0::2], props[1::2]))
            if int(kv.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_torchrun(environ=None) -> bool:
    """True inside a rank process (``torch.distributed.run`` or :func:`spawn_ranks` set the variables)."""
    env = os.environ if environ is None else environ
    return "RANK" in env and "WORLD_SIZE" in env


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this pool
    return env


def spawn_ranks(cmd, nproc: int, timeout: float | None = None, stdout=None, stderr=None, env=None, keep=None):
    """Run ``cmd`` (argv list) ``nproc`` times as rank processes.  Rank 0's stdout is relayed to ``stdout``
    (default ``sys.stdout``) line by line -- only the lines ``keep(line)`` accepts when a predicate is given, the
    rest go to this process's stderr (libraries print banners on stdout) --; the other ranks' stdout is dropped;
    every rank's stderr goes to ``stderr`` (default: inherited).  Returns the first non-zero exit code, else 0.  When a rank fails or the
    timeout expires the remaining ranks are terminated (exactly the PIDs started here)."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    out = sys.stdout if stdout is None else stdout
    port = free_port()
    procs = []
    for r in range(nproc):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, nproc, port, env), stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            dst = out if (keep is None or keep(line)) else sys.stderr
            dst.write(line)
            dst.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    live = set(range(nproc))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124
            for r in sorted(live):
                procs[r].terminate()
            for r in sorted(live):
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        else:
            time.sleep(0.05)
    t.join(5)
    return rc
