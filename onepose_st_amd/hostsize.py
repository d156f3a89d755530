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


def _cpu_busy_sample(seconds: float = 0.05) -> dict:
    """{cpu: busy fraction} over a short window from /proc/stat ({} when it cannot be read)."""
    def read():
        out = {}
        try:
            for ln in open("/proc/stat"):
                if ln.startswith("cpu") and ln[3:4].isdigit():
                    f = ln.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])          # total, idle + iowait
        except (OSError, ValueError):
            return {}
        return out
    import time
    a = read()
    if not a:
        return {}
    time.sleep(seconds)
    b = read()
    busy = {}
    for c, (tot1, idle1) in b.items():
        tot0, idle0 = a.get(c, (tot1, idle1))
        dt = tot1 - tot0
        busy[c] = 0.0 if dt <= 0 else 1.0 - (idle1 - idle0) / dt
    return busy


def job_cpus(affinity=None, quota=None, prefer_idle: bool = False):
    """The CPUs this job may really use: the affinity mask (physical cores first), cut to the cgroup quota when that is
    smaller (a quota does not name CPUs: the cut keeps the ranks' ranges disjoint and the job's threads together).  ``prefer_idle``
    (one-rank jobs on a shared host): of the physical cores in the mask, the ones that were idle over a 50 ms sample come first --
    the first ``quota`` CPUs by index are where every other tenant that reasons the same way sits (measured on the GPU box: 3 of 10
    runs pinned to CPUs 0-15 lost 30-60 % to neighbours)."""
    cpus = order_by_core(_affinity()) if affinity is None else list(affinity)
    q = _cgroup_quota_cores() if quota is None else quota
    if q is not None and q > 0:
        keep = max(1, min(len(cpus), int(q)))
        if prefer_idle and affinity is None and keep < len(cpus):
            busy = _cpu_busy_sample()
            if busy:
                n_phys = sum(1 for c in cpus if _siblings_primary(c) == c) or len(cpus)
                phys, rest = cpus[:n_phys], cpus[n_phys:]

                def core_busy(c):          # a core is as busy as the busier of its hardware threads
                    sib = [d for d in busy if _siblings_primary(d) == _siblings_primary(c)] or [c]
                    return max(busy.get(d, 0.0) for d in sib)
                idle_first = sorted(phys, key=lambda c: (round(core_busy(c), 2) > 0.05, core_busy(c) if core_busy(c) > 0.05 else 0.0, c))
                cpus = idle_first + rest
        cpus = cpus[:keep]
    return cpus


def rank_cpus(rank: int, world: int, affinity=None, quota=None, max_per_rank=None):
    """CPU ids of rank ``rank`` of ``world``: an equal contiguous slice of :func:`job_cpus`, at most ``max_per_rank``
    (``OPHIP_CPU_THREADS`` overrides the per-RANK cap -- it used to cap the whole job -- default 16) and at least one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} of world {world}")
    cpus = job_cpus(affinity, quota, prefer_idle=(world == 1))
    cap = int(os.environ.get("OPHIP_CPU_THREADS", MAX_CORES_PER_RANK)) if max_per_rank is None else int(max_per_rank)
    share = min(max(1, len(cpus) // world), max(1, cap))
    if share * world <= len(cpus):          # ranges packed from the front of the list: physical cores before sibling threads
        return cpus[rank * share:(rank + 1) * share]
    return [cpus[rank % len(cpus)]]        # fewer CPUs than ranks: ranks share them round-robin


QUOTA_MARGIN = 2          # cores of a cgroup CPU QUOTA a job leaves unused beyond the feeder cores: see quota_limited() and pin_rank()


def quota_limited(affinity=None, quota=None) -> bool:
    """True when a cgroup CPU quota (CPU TIME per period), not the affinity mask (CPUs), is what bounds this job.  A quota is enforced per
    100 ms period: a job whose runnable threads add up to the quota for one period is stopped -- every thread of it, the one that feeds
    the GPU included -- until the next (measured on the GPU box, 16-core quota on a 256-CPU mask: a 14-thread RANSAC pool + the feeder
    thread + the HIP runtime's polling thread were stopped for 20-90 ms in 5 of 12 short bench runs; 12 threads with the process confined
    to 48 CPUs, see pin_rank(): none in 10, and every run within 1.5 % of the others)."""
    q = _cgroup_quota_cores() if quota is None else quota
    n = len(_affinity()) if affinity is None else len(affinity)
    return q is not None and 0 < q < n


def pnp_threads(n_cores: int, under_quota: bool = False, world: int = 1) -> int:
    """RANSAC pool threads for a rank that owns ``n_cores``: all of them but the feeder cores when the share allows it
    (>= 4 cores), never fewer than one.  ``under_quota``: the share is CPU time under a cgroup quota -- the job also keeps
    ``QUOTA_MARGIN`` cores' worth unused (split over its ``world`` ranks), as long as at least half the share stays."""
    if under_quota:
        margin = -(-QUOTA_MARGIN // max(1, world))
        if n_cores - FEEDER_CORES - margin >= max(1, n_cores // 2):
            return n_cores - FEEDER_CORES - margin
    if n_cores >= 2 * FEEDER_CORES:
        return n_cores - FEEDER_CORES
    return max(1, n_cores - 1)


confined_order = []       # pin_rank(): the CPUs a one-rank job under a quota was confined to, idlest physical cores first (empty: not confined)
QUOTA_ROOM = 3            # a one-rank job under a CPU quota keeps its threads on QUOTA_ROOM x quota CPUs of its mask


def pin_rank(rank: int, world: int, affinity=None, quota=None):
    """Pin the calling process (and every thread it starts afterwards: the PnP pool, torch's intra-op pool, the HIP runtime's helpers) to
    this rank's CPU range; returns the rank's CPU list (its SHARE: what pool sizes are derived from).
    world > 1: the process is pinned to exactly its share (disjoint ranges per rank).
    world == 1 under a cgroup CPU quota smaller than the mask (the GPU box: 16 cores' worth of CPU time on a 256-CPU mask): CFS hands the
    quota out in 5 ms slices PER CPU a thread wakes up on, and ~210 threads (most of them the HIP runtime's and torch's, waking briefly)
    wandering over 256 CPUs strand enough of it that the job is stopped for 10-90 ms at 4-10 cores of real use (throttled periods in 5 of
    12 short bench runs with a 14-thread pool, still 3 of 15 with 10: `cgroup_cpu_throttled_in_timed_region`).  Pinned to EXACTLY the
    quota's CPUs it is never throttled but cannot leave a CPU a neighbour takes (3 of 10 runs lost 30-60 % on a shared host).  So the
    process is confined to ``QUOTA_ROOM`` x quota CPUs -- idle physical cores first -- which bounds the stranded slices to a tenth of
    the quota and leaves the scheduler room to move."""
    mine = rank_cpus(rank, world, affinity, quota)
    if not hasattr(os, "sched_setaffinity"):
        return mine
    try:
        if world > 1:
            os.sched_setaffinity(0, mine)
        elif affinity is None and quota_limited(None, quota) and os.environ.get("OPHIP_PIN_SINGLE_RANK", "1") != "0":
            q = _cgroup_quota_cores() if quota is None else quota
            wide = job_cpus(None, QUOTA_ROOM * int(q), prefer_idle=True)
            if len(wide) < len(_affinity()):
                os.sched_setaffinity(0, wide)
                confined_order[:] = wide
    except OSError:
        pass                                # a restricted container: sizing still holds, pinning is best effort
    return mine


def host_budget(world: int, fps_per_rank: float, ransac_cpu_ms_per_frame: float) -> dict:
    """CPU the "+PnP" leg needs: threads busy per rank = frames/s x CPU-seconds per frame."""
    busy = fps_per_rank * ransac_cpu_ms_per_frame * 1e-3
    return {"threads_busy_per_rank": busy, "threads_busy_total": busy * world}


class worker_cpus:
    """``with worker_cpus(my_cpus): pool = PnPPool(...)`` -- threads created inside inherit an affinity WITHOUT the rank's feeder cores (its
    first ``FEEDER_CORES`` CPUs), which stay with the thread that enqueues frames and the HIP runtime's helpers: on a pinned share the 14
    RANSAC workers would otherwise sit on every CPU the feeder can wake up on, and it waits for a worker's time slice (milliseconds) now and
    then.  No-op when the share is too small to split or the platform has no ``sched_setaffinity``."""

    def __init__(self, cpus, feeder_cores=FEEDER_CORES):
        self.cpus = list(cpus)
        self.saved = None
        self.feeder_cores = int(feeder_cores)

    def __enter__(self):
        if len(self.cpus) >= 2 * self.feeder_cores and hasattr(os, "sched_setaffinity"):
            try:
                self.saved = os.sched_getaffinity(0)
                os.sched_setaffinity(0, self.cpus[self.feeder_cores:])
            except OSError:
                self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            try:
                os.sched_setaffinity(0, self.saved)
            except OSError:
                pass
        return False

