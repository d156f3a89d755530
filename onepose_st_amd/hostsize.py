"""Host-side sizing of a rank: which CPUs it may use and how many threads its PnP pool gets.

The reference fans frame chunks out to workers that each own ``num_cpus=1`` plus a GPU share
(``src/inference/inference_OnePosePlus.py:81-98``, worker ``:70``); here a rank owns one GPU and a *slice of the
host*: the CPUs this job may run on (affinity mask, narrowed by the cgroup CPU quota) are cut into ``world`` equal
contiguous ranges, rank ``r`` pins itself to range ``r`` and sizes its RANSAC pool from that range -- never from a
machine-wide constant, so that eight ranks on one node do not collapse to one PnP thread each (round-2 defect) and
8 x 14 background threads do not float over the same cores.

Pure host logic (no GPU, no torch): covered by ``tests/test_hostsize.py``.
"""
from __future__ import annotations

import os

# what one rank may take at most: a one-GPU box exposes every logical CPU of the machine in the affinity mask while the
# job's share is 16 cores per GPU; more RANSAC threads than that do not shorten a frame's pose either (a frame's 10 000 trials
# are split into 40 chunks of 256)
MAX_CORES_PER_RANK = 16
FEEDER_CORES = 2          # the Python thread that enqueues frames + the HIP runtime's helper thread


def _affinity():
    if hasattr(os, "sched_getaffinity"):
        return sorted(os.sched_getaffinity(0))
    return list(range(os.cpu_count() or 1))


def _cgroup_quota_cores():
    """CPU quota of this cgroup in cores (cgroup v2 ``cpu.max`` or v1 ``cfs_quota_us``); None when unlimited."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            return max(1, int(int(txt[0]) / int(txt[1])))
        return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def _siblings_primary(cpu: int) -> int:
    """Lowest logical CPU of the physical core ``cpu`` belongs to (itself when the topology files are absent)."""
    try:
        txt = open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read().strip()
        first = txt.replace("-", ",").split(",")[0]
        return int(first)
    except (OSError, ValueError, IndexError):
        return cpu


def order_by_core(cpus, primary_of=_siblings_primary):
    """One hardware thread per physical core first, the sibling threads after them: consecutive slices of the result are
    distinct physical cores for as long as there are any."""
    cpus = sorted(cpus)
    first = [c for c in cpus if primary_of(c) == c or primary_of(c) not in cpus]
    rest = [c for c in cpus if c not in set(first)]
    return first + rest


def job_cpus(affinity=None, quota=None):
    """The CPUs this job may really use: the affinity mask (physical cores first), cut to the cgroup quota when that is
    smaller (the first ``quota`` CPUs; a quota does not name CPUs, the cut only keeps the ranks' ranges disjoint)."""
    cpus = order_by_core(_affinity()) if affinity is None else list(affinity)
    q = _cgroup_quota_cores() if quota is None else quota
    if q is not None and q > 0:
        cpus = cpus[:max(1, min(len(cpus), int(q)))]
    return cpus


def rank_cpus(rank: int, world: int, affinity=None, quota=None, max_per_rank=None):
    """CPU ids of rank ``rank`` of ``world``: an equal contiguous slice of :func:`job_cpus`, at most ``max_per_rank``
    (``OPHIP_CPU_THREADS`` overrides the per-RANK cap -- it used to cap the whole job -- default 16) and at least one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} of world {world}")
    cpus = job_cpus(affinity, quota)
    cap = int(os.environ.get("OPHIP_CPU_THREADS", MAX_CORES_PER_RANK)) if max_per_rank is None else int(max_per_rank)
    share = min(max(1, len(cpus) // world), max(1, cap))
    if share * world <= len(cpus):          # ranges packed from the front of the list: physical cores before sibling threads
        return cpus[rank * share:(rank + 1) * share]
    return [cpus[rank % len(cpus)]]        # fewer CPUs than ranks: ranks share them round-robin


def pnp_threads(n_cores: int) -> int:
    """RANSAC pool threads for a rank that owns ``n_cores``: all of them but the feeder cores when the share allows it
    (>= 4 cores), never fewer than one."""
    if n_cores >= 2 * FEEDER_CORES:
        return n_cores - FEEDER_CORES
    return max(1, n_cores - 1)


def pin_rank(rank: int, world: int, affinity=None, quota=None):
    """Pin the calling process (and every thread it starts afterwards: the PnP pool, torch's intra-op pool) to this rank's
    CPU range; returns the CPU list.  A one-rank job keeps its mask (nothing to keep apart)."""
    mine = rank_cpus(rank, world, affinity, quota)
    if world > 1 and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            pass                            # a restricted container: sizing still holds, pinning is best effort
    return mine


def host_budget(world: int, fps_per_rank: float, ransac_cpu_ms_per_frame: float) -> dict:
    """CPU the "+PnP" leg needs: threads busy per rank = frames/s x CPU-seconds per frame."""
    busy = fps_per_rank * ransac_cpu_ms_per_frame * 1e-3
    return {"threads_busy_per_rank": busy, "threads_busy_total": busy * world}
