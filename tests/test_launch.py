"""The driver-side fan-out of ``bench.py --gpus N`` (onepose_st_amd/launch.py) on CPU: child processes with the
torch.distributed rendezvous variables, rank 0's stdout relayed, failures propagated, and bench.py's own launcher
branch rehearsed over gloo with the GPU part stubbed out (OPHIP_BENCH_LAUNCH_PROBE)."""
import io
import json
import os
import subprocess
import sys
import time

from onepose_st_amd.launch import launched_by_torchrun, rank_env, spawn_ranks

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_GLOO = r"""
import json, os, torch, torch.distributed as dist
torch.set_num_threads(1)
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
print("noise from rank", dist.get_rank())
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "sum": t.item(), "addr": os.environ["MASTER_ADDR"]}))
dist.destroy_process_group()
"""


def test_rank_env_and_detection():
    env = rank_env(1, 4, 29511, base={})
    assert env["RANK"] == "1" and env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launched_by_torchrun(env) and not launched_by_torchrun({})


def test_spawn_ranks_gloo_world2_relays_rank0_only():
    out = io.StringIO()
    rc = spawn_ranks([sys.executable, "-c", _GLOO], 2, timeout=240, stdout=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().strip().splitlines() if not ln.startswith("[Gloo]")]
    assert lines[0] == "noise from rank 0" and len(lines) == 2          # rank 1's stdout is not relayed
    rec = json.loads(lines[1])
    assert rec == {"world": 2, "sum": 3.0, "addr": "127.0.0.1"}


def test_spawn_ranks_propagates_failure_and_stops_the_others():
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(120)\n"
    t0 = time.monotonic()
    rc = spawn_ranks([sys.executable, "-c", code], 2, timeout=100, stdout=io.StringIO())
    assert rc == 7 and time.monotonic() - t0 < 60


def test_bench_self_launch_probe_world2():
    """python bench.py --gpus 2 (no torchrun): the parent fans out, never touches the GPU, relays ONE JSON line."""
    env = dict(os.environ, OPHIP_BENCH_LAUNCH_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout                                     # banners of rank 0 went to stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 3.0 and rec["probe"] is True


def test_bench_self_launch_probe_world8_reports_every_rank():
    """The 8-GPU run is the driver's; what can be checked here is that `python bench.py --gpus 8` fans out eight ranks, every one of them
    reports (frames done, own seconds, own PnP ceiling), and rank 0's ONE line says so: `ranks.reporting == ranks.expected == 8`,
    `all_frames_done`, per-rank records in rank order with distinct host slices."""
    env = dict(os.environ, OPHIP_BENCH_LAUNCH_PROBE="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--share-device", "--dist-backend", "gloo",
                        "--steps", "3", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rank_sum"] == 36.0
    rk = rec["ranks"]
    assert rk["expected"] == 8 and rk["reporting"] == 8 and rk["all_frames_done"] is True
    assert [d["rank"] for d in rk["per_rank"]] == list(range(8)) and [d["device"] for d in rk["per_rank"]] == list(range(8))
    assert all(d["frames"] == 3 and d["value"] > 0 and d["pnp_ceiling_fps"] > 0 for d in rk["per_rank"])
    assert abs(rk["value_sum_of_ranks"] - sum(3 / (1.0 + 0.01 * r) for r in range(8))) < 1e-9
    assert len(rec["host_cores"]) == 8 and len(set(rec["first_cpu"])) >= 1


def test_rank_report_flags_a_missing_or_short_rank():
    sys.path.insert(0, REPO)
    import bench
    full = [[r, 20, 0.017, 1900.0, r] for r in range(4)]
    ok = bench.rank_report(full, 4, 20)
    assert ok["all_frames_done"] and ok["reporting"] == 4 and abs(ok["value_sum_of_ranks"] - 4 * 20 / 0.017) < 1e-6
    assert not bench.rank_report(full[:3], 4, 20)["all_frames_done"]                      # a rank did not report
    short = [list(r) for r in full]
    short[2][1] = 19
    assert not bench.rank_report(short, 4, 20)["all_frames_done"]                         # a rank dropped a frame
    dup = [list(r) for r in full]
    dup[3][0] = 2
    assert not bench.rank_report(dup, 4, 20)["all_frames_done"]                           # two records from one rank


def test_rank_report_counts_poses_when_host_pnp_is_on():
    sys.path.insert(0, REPO)
    import bench
    full = [[r, 20, 0.017, 1900.0, r, 20] for r in range(2)]
    assert bench.rank_report(full, 2, 20, 20)["all_frames_done"]
    lost = [list(r) for r in full]
    lost[1][5] = 19                                                                       # a pose that was never joined
    assert not bench.rank_report(lost, 2, 20, 20)["all_frames_done"]
    assert bench.rank_report(lost, 2, 20)["all_frames_done"]                              # matcher only: poses are not expected


def test_a_short_rank_makes_every_rank_exit_nonzero():
    """The per-rank record carries what the rank COUNTED (frames whose results reached the host), every rank evaluates the same gathered rows,
    and a shortfall on one rank ends all of them with an error -- not only rank 0 (launch probe over gloo, world 2, rank 1 one frame short)."""
    env = dict(os.environ, OPHIP_BENCH_LAUNCH_PROBE="1", OPHIP_BENCH_PROBE_SHORT_RANK="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    procs = []
    port = 29500 + (os.getpid() % 2000)
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 TORCHELASTIC_RUN_ID="probe")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo",
                                       "--steps", "3", "--warmup", "0"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode != 0 for p in procs] == [True, True], [o[1][-500:] for o in outs]
    assert "not every rank finished its frames" in outs[0][1] and "not every rank finished its frames" in outs[1][1]
