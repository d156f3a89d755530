"""Per-rank host sizing (onepose_st_amd/hostsize.py): the round-2 defect was a machine-wide 16-thread cap divided by the
world size, i.e. ONE RANSAC thread per rank at 8 GPUs.  These run without a GPU."""
import json
import os
import subprocess
import sys

import pytest

from onepose_st_amd import hostsize

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("n_cpus", [128, 256])
def test_world8_keeps_at_least_10_pnp_threads_per_rank(n_cpus, monkeypatch):
    monkeypatch.delenv("OPHIP_CPU_THREADS", raising=False)
    aff = list(range(n_cpus))
    seen = set()
    for r in range(8):
        mine = hostsize.rank_cpus(r, 8, affinity=aff, quota=0)
        assert hostsize.pnp_threads(len(mine)) >= 10, (r, mine)
        assert not (seen & set(mine)), "rank CPU ranges must be disjoint"
        seen |= set(mine)
    assert len(seen) == 8 * 16


@pytest.mark.parametrize("world,want_cores,want_threads", [(1, 16, 14), (2, 16, 14), (4, 16, 14), (8, 16, 14)])
def test_sizing_for_each_world_on_a_256_cpu_host(world, want_cores, want_threads, monkeypatch):
    monkeypatch.delenv("OPHIP_CPU_THREADS", raising=False)
    for r in range(world):
        mine = hostsize.rank_cpus(r, world, affinity=list(range(256)), quota=0)
        assert len(mine) == want_cores and hostsize.pnp_threads(len(mine)) == want_threads


def test_quota_and_small_hosts(monkeypatch):
    monkeypatch.delenv("OPHIP_CPU_THREADS", raising=False)
    # a 16-core cgroup quota on a one-GPU box whose mask shows the whole machine
    assert len(hostsize.rank_cpus(0, 1, affinity=list(range(256)), quota=16)) == 16
    # 8 CPUs for 2 ranks: 4 each, 2 RANSAC threads + 2 feeder cores
    a, b = (hostsize.rank_cpus(r, 2, affinity=list(range(8)), quota=0) for r in (0, 1))
    assert a == [0, 1, 2, 3] and b == [4, 5, 6, 7] and hostsize.pnp_threads(4) == 2
    # fewer CPUs than ranks: still one thread each, nothing raises
    assert [len(hostsize.rank_cpus(r, 8, affinity=[0, 1, 2], quota=0)) for r in range(8)] == [1] * 8
    assert hostsize.pnp_threads(1) == 1 and hostsize.pnp_threads(2) == 1 and hostsize.pnp_threads(3) == 2
    # the per-rank cap is an override, not a machine-wide limit any more
    monkeypatch.setenv("OPHIP_CPU_THREADS", "8")
    assert [len(hostsize.rank_cpus(r, 8, affinity=list(range(256)), quota=0)) for r in range(8)] == [8] * 8


def test_physical_cores_come_before_sibling_threads():
    # 8 logical CPUs, siblings (i, i + 4): the first four entries are four distinct cores
    prim = lambda c: c % 4
    assert hostsize.order_by_core(range(8), primary_of=prim) == [0, 1, 2, 3, 4, 5, 6, 7]
    # siblings enumerated next to each other (0,1), (2,3), ...: one thread per core first
    prim2 = lambda c: c - (c % 2)
    assert hostsize.order_by_core(range(8), primary_of=prim2) == [0, 2, 4, 6, 1, 3, 5, 7]
    # a mask that holds only the sibling thread of a core keeps it among the primaries
    assert hostsize.order_by_core([1, 2, 3], primary_of=prim2) == [1, 2, 3]


def test_host_budget_of_the_scaling_target():
    # 8 ranks x 1 200 frames/s x 8 ms of RANSAC per frame = 9.6 busy threads per rank
    b = hostsize.host_budget(8, 1200.0, 8.0)
    assert abs(b["threads_busy_per_rank"] - 9.6) < 1e-9 and abs(b["threads_busy_total"] - 76.8) < 1e-9
    assert hostsize.pnp_threads(16) > b["threads_busy_per_rank"]


def test_bench_launcher_passes_the_sizing_to_both_gloo_ranks():
    """python bench.py --gpus 2 (self-launch, GPU part stubbed): every rank reports its own slice and pool size."""
    env = dict(os.environ, OPHIP_BENCH_LAUNCH_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "OPHIP_CPU_THREADS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.splitlines()[-1])
    n = len(hostsize.job_cpus())
    share = min(16, max(1, n // 2))
    assert rec["host_cores"] == [share, share]
    assert rec["pnp_threads"] == [hostsize.pnp_threads(share)] * 2
    if n >= 2:
        assert rec["first_cpu"][0] != rec["first_cpu"][1], "the two ranks must sit on different CPU ranges"


def test_pool_stays_below_a_cgroup_cpu_quota():
    """a CPU-time quota smaller than the mask: the pool leaves the feeder cores AND a margin unused (16 -> 12 threads, what ran without a
    single throttled period on the GPU box); a mask-limited share keeps the old rule"""
    assert hostsize.quota_limited(affinity=list(range(256)), quota=16)
    assert not hostsize.quota_limited(affinity=list(range(16)), quota=16) and not hostsize.quota_limited(affinity=list(range(64)), quota=None)
    assert hostsize.pnp_threads(16, under_quota=True) == 12 and hostsize.pnp_threads(16) == 14
    assert hostsize.pnp_threads(8, under_quota=True, world=2) == 5          # two ranks share the margin
    assert hostsize.pnp_threads(4, under_quota=True) == 2 and hostsize.pnp_threads(2, under_quota=True) == 1

