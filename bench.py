"""Benchmark of the 2D-3D matching hot path (BASELINE.json metric: query frames/sec @ 7k x 4800, d256).

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload c2|c1|c4] [--no-cpu-baseline]

One process per GPU over RCCL: for N > 1 either ``torch.distributed.run`` starts the ranks, or -- when ``python bench.py
--gpus N`` is run directly -- this process starts them itself as children (``onepose_st_amd/launch.py``; the parent imports
nothing that touches the GPU) and relays rank 0's JSON line.  A *step* is one pass of the hot path over one batch of
``--batch`` frames (default 1) whose inputs are already resident in HBM: backbone-output feature maps of a frame
(``feat_c [B,256,60,80]``, ``feat_f [B,128,240,320]``) + the shared 3D object block -> match indices, confidences and
sub-pixel keypoints (rows a1-a11 of SURVEY.md section 8a) AND the host PnP/RANSAC of every frame with the reference's trial
policy (C++ pool, overlapped with the following frames, every pose joined before the clock stops).  The ResNet backbone is
outside the timed region unless ``--with-backbone``.  Frames shard across ranks with no data-path collective: rank 0 builds
weights + the 3D object block and broadcasts them once (RCCL) before the timed region; every rank then matches its own
frames ("scaling": "weak").  Each rank pins itself to its own slice of the host's CPUs and sizes its PnP pool from that
slice (``onepose_st_amd/hostsize.py``).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  "roofline"      dominant kernel (attn_apply: fused Q-proj + linear attention + merge + MLP + 2 LayerNorms + the next layer's
                  K/V reduce), algorithmic FLOPs per launch / average launch duration from HIP events (dispatch timestamps of
                  every 11th launch of that kernel inside the timed region), against the dense bf16 MFMA peak (2.5 PFLOP/s);
                  "traffic" / "mfma_busy..." come from the committed counter pass and are printed only when that pass was
                  taken on the library build that is running (source hash recorded in the pmc file)
  "cpu_baseline"  the oracle (CPU restatement of the reference, torch fp32) timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N = 1 only)
  "host"          this rank's CPU slice, PnP threads, the pool's measured ceiling (frames/s of RANSAC alone on recorded matches)
                  and "host_bound": true when that ceiling is below the matcher-only rate.
"""
from __future__ import annotations

import argparse
import contextlib
import gc
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from onepose_st_amd import hostsize  # noqa: E402        (pure host logic; torch and the HIP binding are imported inside main(),
from onepose_st_amd.launch import launched_by_torchrun, spawn_ranks, visible_gpu_count  # noqa: E402   after the launcher branch)

WORKLOADS = ("c1", "c2", "c4", "c1_hard", "c2_hard")          # *_hard: low-margin / outlier variants (synthetic.HARD_PROFILE), not BASELINE configs

# MI355X_MICROARCH.md dense matrix peaks: f32 (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s; bf16 (v_mfma_f32_32x32x16_bf16) 2.5 PFLOP/s.
# Split-bf16 issues 3 bf16 MFMAs per algorithmic product and is priced against the bf16 peak with 1x algorithmic FLOPs.
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}


def host_cores(rank: int = 0, world: int = 1) -> int:
    """CPU threads rank ``rank`` of ``world`` may really use: its slice of (affinity mask, cgroup quota), at most 16 per
    rank (``onepose_st_amd/hostsize.py``).  A GPU box exposes 256 logical CPUs; a one-GPU job owns a 16-core share."""
    return len(hostsize.rank_cpus(rank, world))


def attn_apply_flops(n_tokens: int, fused_kv: bool, n_layers: int = 6, C: int = 256, D: int = 32) -> float:
    """Algorithmic FLOPs of one attn_apply launch, averaged over the layers of a frame: per token q_proj 2C^2 + merge 2C^2 +
    mlp0 2(2C)(2C) + mlp2 2(2C)C = 16 C^2, plus phi(Q) KV: 2 C D.  In the bf16 modes the
    launches of layers 0..n-2 also carry the next layer's K/V projection and phi(K)^T V (4 C^2 + 2 C D per token)."""
    per_tok = 16.0 * C * C + 2.0 * C * D
    if fused_kv:
        per_tok += (n_layers - 1) / n_layers * (4.0 * C * C + 2.0 * C * D)
    return float(n_tokens) * per_tok


def rank_report(rows, world: int, frames_expected: int, poses_expected: int = -1) -> dict:
    """What rank 0 prints about EVERY rank, so that a --gpus N line checks itself: `rows` = the all-gathered per-rank records
    [rank, frames FINISHED inside the timed region (counted where a frame's results reach the host), own seconds of the timed region, own
    PnP ceiling in frames/s, local rank / device index, poses JOINED inside the timed region].  `reporting` must equal `expected` and every
    rank must have finished its frames (and joined as many poses when host PnP is on: poses_expected >= 0); `value_sum_of_ranks` (each
    rank's own rate, summed) against the line's `value` (all frames / slowest rank's time) shows how uneven the ranks were.  Every rank
    evaluates the same gathered rows and exits non-zero on a shortfall (bench.py main).  Reference pattern: the driver-side fan-out and
    gather of inference_OnePosePlus.py:81-98."""
    per = []
    for r in rows:
        rk, frames, secs, ceil_, dev_ix = (float(x) for x in r[:5])
        poses = int(r[5]) if len(r) > 5 else -1
        per.append({"rank": int(rk), "device": int(dev_ix), "frames": int(frames), "poses": poses, "seconds": secs,
                    "value": (frames / secs) if secs > 0 else None, "pnp_ceiling_fps": ceil_ if ceil_ > 0 else None})
    per.sort(key=lambda d: d["rank"])
    ok = (len(per) == world and [d["rank"] for d in per] == list(range(world)) and all(d["frames"] == frames_expected for d in per)
          and (poses_expected < 0 or all(d["poses"] == poses_expected for d in per)))
    vals = [d["value"] for d in per if d["value"]]
    return {"expected": world, "reporting": len(per), "all_frames_done": bool(ok),
            "value_sum_of_ranks": sum(vals) if vals else None,
            "slowest_over_fastest": (min(vals) / max(vals)) if vals else None, "per_rank": per}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="frames per step")
    ap.add_argument("--workload", default="c2", choices=WORKLOADS)
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic frames per rank (cycled)")
    ap.add_argument("--precision", default=os.environ.get("OPHIP_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"],
                    help="matrix arithmetic of the encoder kernels (see DESIGN.md section 4)")
    ap.add_argument("--main-region-only", action="store_true",
                    help="skip the side measurements (other PnP policy, matcher only, object cache): the process then runs the contract's region "
                         "alone -- what `tools/box.sh stats` traces, so that the rocprofv3 averages describe the same conditions as the line")
    ap.add_argument("--inputs-behind", action="store_true",
                    help="queue the input kernels (PE, transposes, keypoint encoding) behind the previous frame on the compute stream instead of "
                         "on a side stream (the feature maps and the object block are resident and complete here, so the side stream is legitimate)")
    ap.add_argument("--streams", type=int, default=1,
                    help="compute streams the frames alternate over.  Default 1: the model already runs the fine stage and the result "
                         "read-back on side streams (frame t's refinement under frame t + 1's input kernels) while attn_apply never shares "
                         "the chip, so its HIP-event time is its own; 2 measured no gain on top of that")
    ap.add_argument("--depth", type=int, default=0,
                    help="frames in flight between enqueue and finish (0: streams + 2 -- frame t's fine stage is launched behind frame t + 1's "
                         "encoder, so frame t + 2 must be queued before the host waits for frame t; streams + 1 with OPHIP_FRAME_DEFER_FINE=0)")
    ap.add_argument("--conf-matrix", default="eager", choices=["eager", "lazy"],
                    help="the headline keeps the reference's behaviour (conf_matrix stored every frame); lazy: for experiments with the lazy form alone")
    ap.add_argument("--with-backbone", action="store_true",
                    help="also run the ResNet-FPN backbone (SURVEY 8f-1, HIP convolution kernels) on a synthetic image in every step; "
                         "its maps are then replaced by the planted feature maps (a random image has no matches)")
    ap.add_argument("--no-pnp", action="store_true", help="time the matcher only (no host PnP)")
    ap.add_argument("--pnp-policy", default="reference", choices=["reference", "adaptive"],
                    help="RANSAC trial policy of the timed region: 'reference' = what the reference's inference loop runs (pycolmap branch of "
                         "ransac_PnP: at least 10 000 trials per frame, metric_utils.py:155-165); 'adaptive' = its OpenCV branch (stops at the "
                         "confidence).  The other policy and the matcher alone are timed too and reported beside `value`")
    ap.add_argument("--pnp-threads", type=int, default=0, help="host threads of the PnP pool (0: this rank's CPU slice minus 2 feeder cores, hostsize.pnp_threads)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0 (one-GPU box, with --dist-backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-kernel", default="attn_apply", choices=["attn_apply", "conf"],
                    help="which kernel carries the start / stop events inside the timed region and fills `roofline`: attn_apply (MFMA-bound, the "
                         "headline's) or conf (HBM-bound: the dual-softmax product pass over the N x M matrix; what the c4 side leg reports)")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="skip the BASELINE config 3 (32 frames per step) and config 4 (15k x 19 200) side legs that a default c2 run appends to its line")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="budget of the CPU baseline leg (all-core, full-forward and single-thread samples)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: this process becomes the driver and starts one child per GPU (it never touches the GPU
    #      itself and never exec's; the reference fans out from its driver too, inference_OnePosePlus.py:81-98) --------------
    if args.gpus > 1 and not launched_by_torchrun():
        n_vis = visible_gpu_count()                      # from the environment / KFD topology: the parent never brings up HIP
        if not args.share_device and n_vis is not None and n_vis < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {n_vis} GPU(s) are visible "
                             "(one-GPU rehearsal: --share-device --dist-backend gloo)")
        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                     keep=lambda ln: ln.lstrip().startswith("{")))      # stdout carries the ONE JSON line only
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # this rank's slice of the host: pinned before any thread pool exists (PnP workers and torch's intra-op threads inherit it)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    n_mask = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0
    under_quota = hostsize.quota_limited()               # asked BEFORE this rank narrows its own mask
    my_cpus = hostsize.pin_rank(local % max(local_world, 1), max(local_world, 1))
    n_now = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0
    pinned = world > 1 and n_now == len(my_cpus)          # exactly this rank's share (the pool's threads then keep off the feeder cores)
    confined = world == 1 and 0 < n_now < n_mask             # one rank under a CPU quota: kept on a few times the quota's CPUs (hostsize.pin_rank)
    pnp_threads = args.pnp_threads if args.pnp_threads > 0 else hostsize.pnp_threads(len(my_cpus), under_quota, max(local_world, 1))
    import torch
    if os.environ.get("OPHIP_BENCH_LAUNCH_PROBE"):
        # launcher rehearsal without a GPU (tests/test_launch.py): rendezvous over gloo, one all-reduce, one JSON line from rank 0
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        seen = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(seen)
        sizes = torch.tensor([float(len(my_cpus)), float(pnp_threads), float(my_cpus[0])])
        # the per-rank report of the real line (rank_report below), filled with stand-in numbers: frames done, own seconds, own PnP ceiling
        # (OPHIP_BENCH_PROBE_SHORT_RANK=r: rank r reports one frame fewer -- tests/test_launch.py checks that EVERY rank then exits non-zero)
        short = 1.0 if os.environ.get("OPHIP_BENCH_PROBE_SHORT_RANK") == str(rank) else 0.0
        mine = torch.tensor([float(rank), float(args.steps * args.batch) - short, 1.0 + 0.01 * rank, 1000.0 + rank, float(local),
                             float(args.steps * args.batch)], dtype=torch.float64)
        if world > 1:
            gathered = [torch.zeros(3) for _ in range(world)]
            dist.all_gather(gathered, sizes)
            reports = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(reports, mine)
        else:
            gathered, reports = [sizes], [mine]
        rep = rank_report([r.tolist() for r in reports], world, args.steps * args.batch, args.steps * args.batch)
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "rank_sum": float(seen.item()), "local_rank": local,
                              "host_cores": [int(g[0]) for g in gathered], "pnp_threads": [int(g[1]) for g in gathered],
                              "first_cpu": [int(g[2]) for g in gathered],
                              "ranks": rep}))
        if world > 1:
            dist.destroy_process_group()
        if not rep["all_frames_done"]:          # every rank holds the same gathered rows: all of them leave with an error
            raise SystemExit(f"rank {rank}: not every rank finished its frames: {rep}")
        return
    from onepose_st_amd import hip
    from onepose_st_amd.config import default_config
    from onepose_st_amd.model import OnePosePlus_model
    from onepose_st_amd.pnp import PnPPool
    from onepose_st_amd.sharding import OBJECT_KEYS as OBJ_KEYS, broadcast_object_block, frame_chunk
    from onepose_st_amd.synthetic import CONFIG_SIZES, make_synthetic_inputs, make_synthetic_state_dict, workload_kwargs
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # OPHIP_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box, launched through torchrun with one rank): the N > 1 code path -- RCCL process
    # group, broadcast of the object block, barriers, max / min over ranks -- with world size 1
    dist_on = world > 1 or bool(os.environ.get("OPHIP_BENCH_FORCE_DIST"))
    hip.load()
    if dist_on:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    cfg = default_config()
    cfg["hip_precision"] = args.precision
    if args.conf_matrix == "lazy":
        cfg["hip_conf_matrix"] = "lazy"
    n_points, image_hw, n_plant = CONFIG_SIZES[args.workload]
    H, W = image_hw
    M = (H // 8) * (W // 8)
    B = args.batch

    # ---- weights + the shared 3D object block: built on rank 0, broadcast once ----------------------
    sd = make_synthetic_state_dict(0, cfg)
    wkw = workload_kwargs(args.workload)
    first = make_synthetic_inputs(sd, n_points, image_hw, n_plant, seed=1, config=cfg, frame=0, **wkw)
    bcast_bytes, bcast_ms = 0, None
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        sd, obj, bcast_bytes = broadcast_object_block(sd, first, dev)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.perf_counter() - t_b) * 1e3          # rank 0's view, barrier to barrier: the one collective of the job (first use: includes RCCL's ring set-up)
    else:
        obj = {k: first[k].to(dev) for k in OBJ_KEYS}
    model = OnePosePlus_model(cfg).eval()
    model.load_state_dict(sd, strict=True)
    model.to(dev)

    # ---- this rank's frames (frame ids are disjoint across ranks) --------------------------------
    frames = []
    for fid in frame_chunk(world * args.frames, rank, world):
        inp = first if fid == 0 else make_synthetic_inputs(sd, n_points, image_hw, n_plant, seed=1, config=cfg, frame=fid, **wkw)
        frames.append((inp["feat_c"].to(dev), inp["feat_f"].to(dev)))
    batches = []
    for s in range(args.frames):
        idx = [(s + k) % args.frames for k in range(B)]
        batches.append((torch.cat([frames[i][0] for i in idx]), torch.cat([frames[i][1] for i in idx])))
    obj_b = {k: v.expand(B, *v.shape[1:]) for k, v in obj.items()}

    # host PnP (metric: "2D-3D match + PnP"): frame t's pose is solved on host threads (C++, GIL released) while the GPU
    # matches frame t + 1; every pose is joined before the clock stops
    # on a pinned share the pool's threads keep off the feeder cores.  (A one-rank job under a CPU quota is NOT pinned: on a shared host its
    # threads must be free to leave CPUs a neighbour takes -- pinned to fixed CPUs, 3 of 10 runs lost 30-60 % -- and it keeps below the
    # quota by running fewer pool threads instead, hostsize.pnp_threads(under_quota=True).)
    pin_workers = pinned
    # experiment (OPHIP_SPLIT_FEEDER=n): a confined one-rank job keeps its n idlest CPUs for the thread that enqueues frames; the RANSAC
    # workers roam over the others
    split_n = int(os.environ.get("OPHIP_SPLIT_FEEDER", "0"))
    split = confined and split_n > 0 and len(hostsize.confined_order) >= 4 * split_n
    worker_ctx = hostsize.worker_cpus(my_cpus) if pin_workers else (hostsize.worker_cpus(hostsize.confined_order, split_n) if split else contextlib.nullcontext())
    # experiment (OPHIP_PIN_WORKERS=k): each RANSAC worker of a confined one-rank job on its own k CPUs of the confined set (idlest first, after
    # the first two, which stay free for the feeder)
    pin_k = int(os.environ.get("OPHIP_PIN_WORKERS", "0"))
    if confined and pin_k > 0 and len(hostsize.confined_order) >= 2 + pin_k * pnp_threads:
        os.environ["OPPNP_WORKER_CPUS"] = ",".join(str(c) for c in hostsize.confined_order[2:2 + pin_k * pnp_threads])
        os.environ["OPPNP_WORKER_CPUS_PER"] = str(pin_k)
    with worker_ctx:
        pools = {} if args.no_pnp else {pol: PnPPool(first["K"].numpy(), threads=pnp_threads, pnp_reprojection_error=7, policy=pol)
                                        for pol in ("reference", "adaptive")}
    if pinned and pools and len(my_cpus) >= 2 * hostsize.FEEDER_CORES:
        try:
            os.sched_setaffinity(0, my_cpus[:hostsize.FEEDER_CORES])          # this thread (it enqueues the frames) stays on the feeder cores
        except OSError:
            pass
    if split and pools:
        try:
            os.sched_setaffinity(0, hostsize.confined_order[:split_n])
        except OSError:
            pass
    pool = pools.get(args.pnp_policy)
    pending = []
    last_host = [None]      # the most recent frame's matches on the host: what the PnP-ceiling measurement replays

    inflight = []
    done = {"frames": 0, "poses": 0}       # counted where it happens: frames whose results reached the host (complete), poses joined (join_poses)

    host_t = {"enqueue": 0.0, "wait": 0.0, "finish": 0.0, "submit": 0.0}       # host-side seconds, printed with OPHIP_BENCH_TRACE=1

    def complete(pend):
        """finish one frame (waits on ITS event only) and hand its matches to the host PnP pool"""
        t = time.perf_counter()
        pend.wait()
        host_t["wait"] += time.perf_counter() - t
        t = time.perf_counter()

        def submit(hst):          # called by finish() as soon as the host-side matches exist: the pose solve starts before the data dict is filled
            last_host[0] = hst
            if pool is not None:
                if B == 1:
                    pending.append(pool.submit(hst["mkpts_2d"], hst["mkpts_3d_db"]))
                else:
                    for bb in range(B):
                        sel = hst["b_ids"] == bb
                        pending.append(pool.submit(hst["mkpts_2d"][sel], hst["mkpts_3d_db"][sel]))
        data = pend.finish(on_host=submit)
        host_t["finish"] += time.perf_counter() - t          # (includes the submit since round 4)
        done["frames"] += B
        return data

    # frames alternate over `--streams` HIP streams: consecutive frames are independent, so the single-workgroup
    # kernels and launch tails of one frame (select, kpt_stats, kv_sum ...) overlap the wide kernels of the next
    streams = [torch.cuda.Stream(device=dev, priority=int(os.environ.get("OPHIP_MAIN_PRIO", "0"))) for _ in range(max(1, args.streams))]
    depth = args.depth if args.depth > 0 else len(streams) + (1 if os.environ.get("OPHIP_FRAME_DEFER_FINE", "1") == "0" else 2)

    image = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(7)).to(dev) if args.with_backbone else None

    def step(i):
        """enqueue batch i, then finish batch i - len(streams): the GPU always has queued work"""
        fc, ff = batches[i % len(batches)]
        t = time.perf_counter()
        with torch.cuda.stream(streams[i % len(streams)]):
            if image is not None:
                model.backbone_features(image)
            inflight.append(model.enqueue_features(dict(obj_b), fc, ff, image_hw, host_copy=pool is not None,
                                                   inputs_ready=image is None and not args.inputs_behind))
        host_t["enqueue"] += time.perf_counter() - t
        if len(inflight) >= depth:
            return complete(inflight.pop(0))
        return None

    def drain():
        last_data = None
        for i, st in enumerate(streams):                  # no further frame on these streams: the last frame's kept-back fine stage goes out now
            with torch.cuda.stream(st):
                model.flush()
        while inflight:
            last_data = complete(inflight.pop(0))
        return last_data

    def join_poses():
        if pool is None:
            return []
        pool.wait_all()
        out = [pool.result(tk) for tk in pending]
        done["poses"] += len(out)
        pending.clear()
        return out

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # setup, not measurement: untimed blocks of steps before the W warm-up steps, until the block time settles, so that first-use costs (caching
    # allocator growth for the frames in flight, pinned buffers, side streams, PnP worker start-up, GPU clock ramp) never land
    # in a short timed region; the W warm-up steps and the K timed steps follow as the contract says
    blk_prev = None
    setup_steps = 0
    t_setup = time.perf_counter()
    for rep in range(24):                                 # at most 24 x 16 steps (~0.4 s at c2), usually 2-3 blocks
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(16):
            step(i)
        drain()
        join_poses()
        torch.cuda.synchronize()
        setup_steps += 16
        blk = time.perf_counter() - tb
        if blk_prev is not None and abs(blk - blk_prev) <= 0.05 * blk_prev:
            break                                         # two consecutive blocks within 5 %: clocks and caches are settled
        blk_prev = blk
        if time.perf_counter() - t_setup > 3.0:
            break
    for i in range(args.warmup):
        step(i)
    last = drain()
    n_matches = int(last["i_ids"].numel()) if last is not None else -1
    poses = join_poses()
    n_inliers = int(poses[-1][1]) if poses else -1

    own_dt = [0.0]          # this rank's own seconds of the last timed region (before the max over ranks)
    own_done = [0, 0]       # frames finished / poses joined by this rank INSIDE the last timed region (measured, not args.steps * B)
    HOT_STEPS = int(os.environ.get("OPHIP_BENCH_HOT_STEPS", "32"))          # untimed matcher-only frames right before every timed region (see timed_region)

    def timed_region(active_pool, hot_steps=None):
        """W warm-up + exactly K timed steps with `active_pool` solving the poses (None: matcher only); returns seconds (max over ranks).
        hot_steps: untimed matcher-only frames right in front of the clock (default HOT_STEPS; 0 = the region starts on a GPU that has just
        been idle for the warm-up's pose join: the post-idle clock transient is then INSIDE the timed steps, `value_no_hot_steps`)"""
        nonlocal pool
        n_hot = HOT_STEPS if hot_steps is None else hot_steps
        pool = active_pool
        for i in range(args.warmup):
            step(i)
        drain()
        join_poses()
        gc.collect()
        gc.disable()                                     # no collector pause of the feeder thread inside the timed steps
        # Joining the warm-up's poses and the collector pause leave the GPU idle for 1-50 ms; a chip that comes out of idle first boosts, then
        # overshoots its power limit and runs ~10 % slow for 5-8 ms before it settles (kernel trace of a 20-step region: attn_apply 47 us
        # in frames 0-2, 51-54 us in frames 3-10, 47 us from frame 11 on) -- a third of a 17 ms region.  A few more untimed frames without
        # host PnP (nothing to join afterwards) bring it back to the state it holds in a running pipeline; then the barrier + synchronize the
        # contract asks for, and the clock starts on a GPU that has been idle for microseconds.  (Counted in setup_steps_untimed.)
        pool = None
        for i in range(n_hot):
            step(i)
        drain()
        pool = active_pool
        for st in streams:
            st.synchronize()
        sync_all()
        for k in host_t:
            host_t[k] = 0.0
        f0, p0 = done["frames"], done["poses"]
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        drain()
        t_gpu = time.perf_counter()
        join_poses()
        sync_all()
        dt_ = time.perf_counter() - t0
        gc.enable()
        host_t["tail"] = dt_ - (t_gpu - t0)          # after the last frame left the GPU: its pose (and the final barrier)
        own_dt[0] = dt_
        own_done[0], own_done[1] = done["frames"] - f0, done["poses"] - p0
        if dist_on:
            import torch.distributed as dist
            tt = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_

    def dependent_region(active_pool, n):
        """The reference's own loop (inference.py:148-190): ONE frame in flight -- frame t + 1 is enqueued only after frame t's pose has been
        joined (its crop would come from that pose: pred_poses[id - 1] -> previous_pose_detect -> model -> ransac_PnP).  Same workload, same PnP
        policy; returns (seconds for n frames, per-frame latencies enqueue -> pose in seconds).  A lone frame's kept-back fine stage goes
        out with its wait (ophip_frame_wait), so there is no flush() round trip; the pose solve starts from the finish() callback."""
        nonlocal pool
        pool = active_pool
        for i in range(4):                                # untimed: first-use costs of the depth-1 pattern
            complete(model.enqueue_features(dict(obj_b), *batches[i % len(batches)], image_hw, host_copy=pool is not None, inputs_ready=image is None))
            join_poses()
        sync_all()
        lat = []
        t0 = time.perf_counter()
        for i in range(n):
            fc, ff = batches[i % len(batches)]
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                pend = model.enqueue_features(dict(obj_b), fc, ff, image_hw, host_copy=pool is not None, inputs_ready=image is None and not args.inputs_behind)
            complete(pend)
            join_poses()
            lat.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, lat

    # side measurements first (the other PnP policy, the matcher alone), the contract's region last with the kernel timing on
    other = {"reference": "adaptive", "adaptive": "reference"}[args.pnp_policy]
    dt_other = dt_matcher = dt_cached = dt_lazy = lazy_reruns = dt_nohot = dt_dep = lat_dep = dt_dep_cached = None
    if not args.main_region_only and args.precision != "f32":
        # the mode in which conf_matrix is not requested, reported beside the headline (which stays eager: the reference
        # writes the matrix every frame).  config["hip_conf_matrix"] = "lazy": nothing N x M is stored, match lists bit-identical
        cfg_lazy = dict(cfg)
        cfg_lazy["hip_conf_matrix"] = "lazy"
        model_lazy = OnePosePlus_model(cfg_lazy).eval()
        model_lazy.load_state_dict(sd, strict=True)
        model_lazy.to(dev)
        model_eager, model = model, model_lazy
        for i in range(16):                               # setup of the second model (weight packing, its frame plan and block sizes): untimed
            step(i)
        drain()
        join_poses()
        dt_lazy = timed_region(pools.get(args.pnp_policy))
        lazy_reruns = model_lazy.lazy_reruns
        model = model_eager
        del model_lazy
    if not args.main_region_only:
        dt_other = timed_region(pools[other]) if pools else None
        dt_matcher = timed_region(None)
        model.cache_object = True                       # per-object cache (config["hip_cache_object"]): keypoint encoding, first layer's 3D rows, layer 1's
        dt_cached = timed_region(None)                  # 3D-source block -- a sequence's frames share one resident object block; matcher only, to compare with dt_matcher
        if world == 1 and B == 1:
            dt_dep_cached, _ = dependent_region(pools.get(args.pnp_policy), max(args.steps, 20))
        model.cache_object, model._obj_cache = False, None
        if world == 1 and B == 1:
            # the dependent sequence (one frame in flight): the number a user of the reference's inference.py loop sees
            dt_dep, lat_dep = dependent_region(pools.get(args.pnp_policy), max(args.steps, 20))
        # the same K steps WITHOUT the untimed hot frames in front of the clock: the post-idle clock transient inside the timed steps
        dt_nohot = timed_region(pools.get(args.pnp_policy), hot_steps=0)
    # the timed launches carry a start / stop event pair filled with the dispatch's own timestamps (hipExtLaunchKernelGGL); such a launch
    # costs the stream a few us more than a plain one: timing all 6 launches of every frame took 9 % off `value`, every 11th launch
    # (coprime with the 6 layers, so every layer is sampled equally) ~2 %, with the same average
    time_every = int(os.environ.get("OPHIP_BENCH_TIME_EVERY", "11" if args.roofline_kernel == "attn_apply" else "1"))
    hip.timing_select(args.roofline_kernel, every=time_every)

    def throttle_stat():
        """(nr_throttled, throttled_usec) of this cgroup's CPU quota (cgroup v2 cpu.stat / v1 cpu.stat in ns): a job that asks for more CPU
        time per period than its quota is stopped -- every thread, the GPU feeder included -- until the next period (tens of ms)"""
        for path, scale in (("/sys/fs/cgroup/cpu.stat", 1.0), ("/sys/fs/cgroup/cpu/cpu.stat", 1e-3)):
            try:
                kv = dict(ln.split()[:2] for ln in open(path).read().splitlines() if len(ln.split()) >= 2)
                return int(kv.get("nr_throttled", 0)), float(kv.get("throttled_usec", kv.get("throttled_time", 0))) * scale
            except (OSError, ValueError):
                continue
        return None
    def thread_cpu():
        """{tid: (comm, cpu seconds)} of this process's threads (OPHIP_BENCH_TRACE: who uses the cgroup's CPU quota)"""
        out = {}
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{tid}/stat").read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[tid] = (comm, (int(rest[11]) + int(rest[12])) / tck)
            except (OSError, ValueError, IndexError):
                pass
        return out
    tc0 = thread_cpu() if os.environ.get("OPHIP_BENCH_TRACE") else None
    t_wall0 = time.perf_counter()
    thr0 = throttle_stat()
    dt = timed_region(pools.get(args.pnp_policy))
    dt_own = own_dt[0]
    main_done = list(own_done)
    thr1 = throttle_stat()
    if tc0 is not None and rank == 0:
        tc1, wall = thread_cpu(), time.perf_counter() - t_wall0
        used = sorted(((tc1[t][1] - tc0.get(t, (None, 0.0))[1], tc1[t][0]) for t in tc1), reverse=True)
        tot = sum(u for u, _ in used)
        print(f"cpu over warm-up + timed region ({wall * 1e3:.0f} ms wall, {len(tc1)} threads): {tot / wall:.2f} cores in all; busiest threads "
              + " ".join(f"{u / wall:.2f}" for u, _ in used[:24]) + f"; threads above 2 %: {sum(1 for u, _ in used if u / wall > 0.02)}", file=sys.stderr)
    if os.environ.get("OPHIP_BENCH_TRACE") and rank == 0:
        print("host us/step: " + ", ".join(f"{k} {1e6 * v / args.steps:.0f}" for k, v in host_t.items() if k != "tail")
              + f"; wall {1e6 * dt / args.steps:.0f}; after the last frame left the GPU {1e6 * host_t.get('tail', 0.0):.0f} us in all", file=sys.stderr)
    launches, kern_ms = hip.timing_read()
    hip.timing_select("")
    stage_kernels = {}
    if args.roofline_kernel == "conf":
        # the N x M STAGE (similarity tiles + dual softmax): further regions of the same K steps, each with one of the stage's
        # kernels bracketed, so that the line can divide the stage's algorithmic bytes by the time of ALL its N x M kernels (`stage_frac`).
        # One-pass form: sim_stats (tiles, S stored) + conf (in-place conversion); two-pass form (ophip_coarse_two_pass: large N x M):
        # sim_stats (statistics only) + sim_conf (tiles again, every confidence written once) and no conf launch.
        if launches:
            stage_kernels["conf"] = kern_ms / launches
        for nm in ("sim_stats", "sim_conf", "stat_combine"):
            hip.timing_select(nm, every=1)
            timed_region(pools.get(args.pnp_policy))
            n_, ms_ = hip.timing_read()
            hip.timing_select("")
            if n_:
                stage_kernels[nm] = ms_ / n_
    # The same kernel with the chip to itself: eight frames one at a time (enqueue, flush the kept-back fine stage, finish, synchronize), every
    # launch bracketed.  Since the end of round 4 the pipeline lets the encoder's first layer start beside the previous frame's last fine
    # workgroups and the next frame's input kernels run beside the last layers (both raise frames/s and lengthen the launches they
    # touch): `roofline.achieved` stays what the contract asks for -- the launch time inside the timed region -- and `roofline.alone` says
    # what the kernel does when nothing shares the chip.
    alone_launches, alone_ms = 0, 0.0
    if args.roofline_kernel == "attn_apply":
        torch.cuda.synchronize()
        hip.timing_select(args.roofline_kernel, every=1)
        for i in range(8):
            fc_a, ff_a = batches[i % len(batches)]
            with torch.cuda.stream(streams[0]):
                pa = model.enqueue_features(dict(obj_b), fc_a, ff_a, image_hw, host_copy=False, inputs_ready=image is None and not args.inputs_behind)
                model.flush()
            pa.finish()
            torch.cuda.synchronize()
        alone_launches, alone_ms = hip.timing_read()
        hip.timing_select("")

    # what the host side can sustain by itself: this rank's pool alone on recorded matches, every rank at once (they share the host);
    # when that ceiling is below the matcher's rate, `value` is a host number and the line says so ("host_bound")
    pnp_ceiling = None
    own_rate = 0.0
    if pools and last_host[0] is not None and last_host[0]["K"] > 0:
        pl, hst = pools[args.pnp_policy], last_host[0]
        sel = slice(None) if B == 1 else (hst["b_ids"] == 0)
        p2, p3 = hst["mkpts_2d"][sel], hst["mkpts_3d_db"][sel]
        sync_all()
        n_rep = 64
        t0 = time.perf_counter()
        tickets = [pl.submit(p2, p3) for _ in range(n_rep)]
        pl.wait_all()
        for tk in tickets:
            pl.result(tk)
        rate = n_rep / (time.perf_counter() - t0)
        own_rate = rate
        if dist_on:
            import torch.distributed as dist
            tt = torch.tensor([rate], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            rate = float(tt.item())
        pnp_ceiling = rate * world

    # HBM bytes and matrix-pipe busy cycles per launch of the roofline kernel: from the committed counter passes (separate
    # rocprofv3 --pmc runs, tools/box.sh pmc), not live -- and quoted ONLY when those passes ran on the build that is
    # running now (the pmc file records ophip_build_stamp): a kernel change without a re-profile prints null, never a stale number
    traffic, mfma_busy, pmc_note, coarse_stage = None, None, None, None
    stem = {"bf16x3": "enc_x3w8_kernel<false,", "bf16": "attn_apply_bf16_kernel<1", "f32": "attn_apply_kernel"}[args.precision]
    try:
        pmc_path = sorted(p for p in os.listdir(os.path.join(REPO, "profiles")) if p.endswith("_pmc.json"))[-1]
        pmc_all = json.load(open(os.path.join(REPO, "profiles", pmc_path)))
        if pmc_all.get("library_build_stamp") != hip.build_stamp():
            pmc_note = f"profiles/{pmc_path} was taken on build {pmc_all.get('library_build_stamp')}, this is {hip.build_stamp()}: not quoted"
        else:
            hits = [v for k, v in pmc_all["kernels"].items() if k.startswith(stem)]
            if hits and B == 1 and args.workload == "c2":
                traffic = hits[0]["hbm_bytes_per_launch"]
                mfma_busy = hits[0].get("SQ_VALU_MFMA_BUSY_CYCLES_median")
                pmc_note = f"profiles/{pmc_path} (build {hip.build_stamp()})"
                # the HBM-bound stage of the path (dual softmax + mutual-NN): counter bytes per frame of its kernels against the
                # algorithmic figure (inputs once + ONE f32 write of conf_matrix), and the similarity tile kernel's matrix-pipe share
                kern = pmc_all["kernels"]
                pick = lambda part: [v for k, v in kern.items() if part in k]
                parts_ = {n: pick(n) for n in ("sim_frag_kernel<3, 0>", "conf_kernel", "stat_combine_kernel", "select_decide_kernel", "select_place_kernel")}
                if all(parts_.values()):
                    coarse_stage = {
                        "hbm_bytes_per_frame": sum(v[0]["hbm_bytes_per_launch"] for v in parts_.values()),
                        "algorithmic_bytes_per_frame": float((n_points + M) * 256 * 2 + 4 * n_points * M),
                        "per_kernel_bytes": {n.split("<")[0]: v[0]["hbm_bytes_per_launch"] for n, v in parts_.items()},
                        "sim_frag_mfma_busy_frac_at_peak_clock": parts_["sim_frag_kernel<3, 0>"][0].get("mfma_busy_frac"),
                        "counters_from": pmc_note,
                    }
    except (OSError, KeyError, ValueError, IndexError):
        pass

    # every rank's own record travels to rank 0: the line then says how many ranks reported, what each did, and what the broadcast cost
    # (frames / poses: what this rank COUNTED inside the timed region -- a frame whose results reached the host, a pose that was joined)
    mine = torch.tensor([float(rank), float(main_done[0]), float(dt_own), float(own_rate), float(local), float(main_done[1])], dtype=torch.float64, device=dev)
    if dist_on:
        import torch.distributed as dist
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        rows = [r.cpu().tolist() for r in rows]
    else:
        rows = [mine.cpu().tolist()]
    ranks = rank_report(rows, world, args.steps * B, -1 if pools.get(args.pnp_policy) is None else args.steps * B)
    if not ranks["all_frames_done"]:          # the verdict is the same on every rank (same gathered rows): all of them exit non-zero
        if dist_on:
            import torch.distributed as dist
            dist.destroy_process_group()
        raise SystemExit(f"--gpus {world} (rank {rank}): {ranks['reporting']} of {ranks['expected']} ranks reported / not every rank finished its frames: {ranks}")

    frames_total = world * args.steps * B
    value = frames_total / dt
    # encoder FLOPs per frame with and without the frame-invariant share (first layer on the 3D stream: 16 C^2 + 2 C D per 3D
    # point, its own K / V half 4 C^2 + 2 C D, and the 3D source's K / V of layer 1: 4 C^2 + 2 C D); utilisation stays on the UNCACHED figure
    _C, _D = 256.0, 32.0
    enc_flops_uncached = (n_points + M) * (6 * (16 * _C * _C + 2 * _C * _D) + 6 * (4 * _C * _C + 2 * _C * _D))
    enc_flops_cached_share = n_points * ((16 * _C * _C + 2 * _C * _D) + 2 * (4 * _C * _C + 2 * _C * _D))
    avg_ms = kern_ms / max(launches, 1)
    flops = attn_apply_flops(B * (n_points + M), fused_kv=args.precision != "f32")
    achieved = flops / (avg_ms * 1e-3) / 1e12 if launches else 0.0

    result = {
        "metric": "query frames/sec (2D-3D match+PnP) @ 7k x 4800 d256; 1/2/4/8 GPU",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "setup_steps_untimed": setup_steps + (1 if args.main_region_only else (5 if args.precision != "f32" else 4)) * (args.warmup + HOT_STEPS),      # settle blocks before the W warm-up steps + the warm-ups (and hot steps) of the side regions
        "hot_steps_untimed_before_each_timed_region": HOT_STEPS,      # matcher-only frames that bring the GPU out of its post-idle clock transient (see timed_region)
        "ms_per_step": dt / args.steps * 1e3,
        "value_matcher_only": (frames_total / dt_matcher) if dt_matcher else None,
        "lazy_conf_frames_rerun_eagerly": lazy_reruns,
        "value_lazy_conf": (frames_total / dt_lazy) if dt_lazy else None,          # conf_matrix not materialised (hip_conf_matrix = "lazy"), same PnP policy as `value`
        "value_matcher_only_object_cached": (frames_total / dt_cached) if dt_cached else None,
        "value_no_hot_steps": (frames_total / dt_nohot) if dt_nohot else None,          # same build, same K steps, OPHIP_BENCH_HOT_STEPS=0 for this region
        # ONE frame in flight, frame t + 1 enqueued after frame t's pose is joined (the reference's loop, inference.py:148-190); latency = enqueue -> pose
        "value_dependent_sequence": (len(lat_dep) / dt_dep) if dt_dep else None,
        "latency_ms": ({"mean": 1e3 * sum(lat_dep) / len(lat_dep), "median": 1e3 * sorted(lat_dep)[len(lat_dep) // 2], "max": 1e3 * max(lat_dep),
                        "frames": len(lat_dep)} if lat_dep else None),
        "value_dependent_sequence_object_cached": (max(args.steps, 20) / dt_dep_cached) if dt_dep_cached else None,
        ("value_pnp_" + other): (frames_total / dt_other) if dt_other else None,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA, f32 accumulate)", "bf16": "bf16 (f32 accumulate)"}[args.precision]
                 + ("; fine stage: plain bf16 (OPHIP_FINE_PRECISION)" if args.precision == "bf16x3" and os.environ.get("OPHIP_FINE_PRECISION") == "bf16" else ""),
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {n_points} 3D points x {M} 2D cells ({H}x{W} image), d256 coarse / d128 fine, "
                        f"{n_plant} planted matches per frame" + (f" (low-margin / outlier profile {wkw})" if wkw else "")
                        + f", {B} frame(s) per step, feature-boundary inputs resident in HBM",
            "frames_per_step": B,
            "matches_per_frame": n_matches // max(B, 1),
            "timed_region": "rows a1-a11 (PE, keypoint encoding, 6-layer coarse encoder, dual-softmax + mutual-NN incl. the "
                            "N x M conf_matrix write, fine refinement)" + ("" if args.no_pnp else " + host PnP/RANSAC of every frame (own C++ "
                            f"estimator, trial policy '{args.pnp_policy}': " + ("at least 10 000 RANSAC trials per frame like the reference's pycolmap call"
                            if args.pnp_policy == "reference" else "adaptive stop at confidence 0.99 like the reference's OpenCV branch")
                            + f"; {pnp_threads} host threads, overlapped with the following frames, all joined before the clock stops)")
                            + ("; backbone (HIP convolutions) inside, on a synthetic image" if args.with_backbone else "; backbone outside"),
            "pnp_inliers_per_frame": n_inliers,
            "streams": len(streams), "frames_in_flight": depth,
            "parallelism": f"frames sharded over {world} rank(s), one RCCL broadcast of weights + 3D block ({bcast_bytes} B)",
        },
        "encoder_flops_per_frame": {"algorithmic_uncached": enc_flops_uncached, "frame_invariant_share_cached_by_hip_cache_object": enc_flops_cached_share,
                                    "with_object_cache": enc_flops_uncached - enc_flops_cached_share,
                                    "note": "roofline.* is computed on the uncached figure (SURVEY 8d); the headline `value` runs WITHOUT the cache"},
        "ranks": ranks,
        "broadcast": {"bytes": bcast_bytes, "ms": bcast_ms, "backend": (args.dist_backend if dist_on else None),
                      "gb_per_s": (bcast_bytes / bcast_ms / 1e6) if bcast_ms else None},
        "host": {
            "cgroup_cpu_throttled_in_timed_region": ({"periods": thr1[0] - thr0[0], "usec": round(thr1[1] - thr0[1])} if thr0 and thr1 else None),
            "host_cores_per_rank": len(my_cpus), "cpus_of_rank0": [my_cpus[0], my_cpus[-1]], "pinned": pinned, "under_cgroup_cpu_quota": under_quota, "confined_to_cpus": (n_now if confined else None),
            "pnp_threads_per_rank": pnp_threads, "pnp_policy": args.pnp_policy,
            "pnp_ceiling_fps": pnp_ceiling,          # RANSAC pools alone on recorded matches, all ranks at once (min over ranks x N)
            "host_bound": (pnp_ceiling < (frames_total / dt_matcher if dt_matcher else value)) if pnp_ceiling else None,
        },
        "roofline": {
            "kernel": {"bf16x3": "enc_x3w8_kernel<false, false> (attn_apply)", "bf16": "attn_apply_bf16_kernel<1, 1, false>", "f32": "attn_apply_kernel"}[args.precision]
                      + " (fused Q-proj + linear attention + merge + MLP + 2 LayerNorms" + (" + next layer's K/V reduce)" if args.precision != "f32" else ")"),
            "bound": "mfma",
            "achieved": achieved,
            "peak": MFMA_PEAK_TFLOPS[args.precision],
            "unit": "TFLOP/s",
            "frac": achieved / MFMA_PEAK_TFLOPS[args.precision],
            "mfma_issue_frac": achieved * (3.0 if args.precision == "bf16x3" else 1.0) / MFMA_PEAK_TFLOPS[args.precision],
            "traffic": traffic,
            "counters_from": pmc_note,
            "library_build_stamp": hip.build_stamp(),
            # SQ_VALU_MFMA_BUSY_CYCLES of the committed counter pass over (1024 SIMDs x this run's launch time x the 2.4 GHz the
            # 2.5 PFLOP/s peak is quoted at): the matrix pipe's busy share at PEAK clock.  The clock the chip really holds in this
            # kernel is the in-kernel s_memtime / s_memrealtime ratio in profiles/r05_stamps_enc_x3w8.txt
            "mfma_busy_frac_at_peak_clock": (mfma_busy / (1024.0 * avg_ms * 1e-3 * 2.4e9)) if (mfma_busy and launches) else None,
            "launches": launches,
            "launches_sampled_every": time_every,
            "avg_launch_ms": avg_ms,
            "flops_per_launch": flops,
            # nothing else on the chip (frames one at a time, every launch timed): the kernel's own number; the fields above are its launches
            # inside the timed region, where the first layer shares the chip with the previous frame's last fine workgroups and the last
            # layers with the next frame's input kernels
            "alone": ({"avg_launch_ms": alone_ms / alone_launches, "launches": alone_launches,
                       "achieved": flops / (alone_ms / alone_launches * 1e-3) / 1e12,
                       "frac": flops / (alone_ms / alone_launches * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.precision],
                       "mfma_issue_frac": flops / (alone_ms / alone_launches * 1e-3) / 1e12 * (3.0 if args.precision == "bf16x3" else 1.0) / MFMA_PEAK_TFLOPS[args.precision]}
                      if alone_launches else None),
        },
    }
    if coarse_stage is not None:
        result["coarse_stage"] = coarse_stage
    if args.roofline_kernel == "conf":
        # the HBM-bound kernel of the path (dual softmax + mutual-NN): conf_kernel reads the stored similarity once and writes
        # conf_matrix once -- 2 x 4 B per (i, j) pair.  The STAGE's algorithmic traffic (inputs once + ONE f32 write of the matrix) is
        # printed beside it: the similarity kernel's store of S is the pass the fused design would not need.
        kbytes = 2.0 * 4.0 * B * n_points * M
        gbs = kbytes / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        kname = "conf_kernel<true, true> (dual-softmax product in log form + candidate tracking: reads S, writes conf_matrix)"
        if not launches and "sim_conf" in stage_kernels:
            # two-pass form: no conf_kernel; the kernel that writes the matrix is the second tile pass (operand fragments in, ONE f32 write out)
            kbytes = float(B) * ((n_points + M) * 256 * 2 * 2 + 4.0 * n_points * M)
            avg_ms = stage_kernels["sim_conf"]
            gbs = kbytes / (avg_ms * 1e-3) / 1e9
            kname = "sim_frag_kernel<3, 3> (second tile pass of the two-pass form: recomputes S on the matrix pipe, writes every confidence once)"
        result["roofline"] = {
            "kernel": kname,
            "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
            "frac_of_measured_copy_rate": gbs / 6290.0,          # MI355X_MICROARCH.md: 6.29 TB/s float4 copy
            "traffic": None, "bytes_per_launch": kbytes,
            "stage_algorithmic_bytes": float(B) * ((n_points + M) * 256 * 2 + 4.0 * n_points * M),
            # the honest figure for the stage: SURVEY 8d's algorithmic bytes (inputs once + ONE f32 write of the matrix) over the time of
            # BOTH N x M kernels (similarity tiles + confidence pass), against 8 TB/s; `frac` above divides the confidence kernel's OWN
            # bytes (S read + conf write = 2 x the algorithmic write) by its own time
            "stage_kernels_avg_launch_ms": stage_kernels,
            "stage_form": ("two passes over the tiles (statistics, then every confidence written once; no S store)" if "sim_conf" in stage_kernels
                           else "one tile pass that stores S + in-place conversion pass"),
            "stage_ms": sum(stage_kernels.values()) if stage_kernels else None,
            "stage_frac": ((float(B) * ((n_points + M) * 256 * 2 + 4.0 * n_points * M)) / (sum(stage_kernels.values()) * 1e-3) / 1e9 / 8000.0)
                          if stage_kernels else None,
            "launches": launches, "launches_sampled_every": time_every, "avg_launch_ms": avg_ms, "library_build_stamp": hip.build_stamp(),
        }

    # ---- side legs: BASELINE configs 3 and 4 on this GPU, each in a child process of its own (same script, its own sizes), untimed by the
    #      headline: `value` above is config 2's and is final before they start ----------------------------------------------------------
    if rank == 0 and world == 1 and not args.main_region_only and not args.no_side_legs and args.workload == "c2" and B == 1:
        import subprocess

        def side_leg(extra, timeout=240, env=None):
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--main-region-only", "--no-side-legs",
                   "--precision", args.precision] + extra
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=(dict(os.environ) | env) if env else None)
                line = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
                if p.returncode != 0 or not line:
                    return {"error": (p.stderr or p.stdout)[-400:]}
                return json.loads(line[-1])
            except (subprocess.TimeoutExpired, OSError, ValueError) as e:
                return {"error": repr(e)[:400]}
        torch.cuda.empty_cache()
        c3 = side_leg(["--workload", "c2", "--batch", "32", "--steps", "6", "--warmup", "2"])
        c4 = side_leg(["--workload", "c4", "--steps", "16", "--warmup", "4", "--roofline-kernel", "conf"])
        # the low-margin / outlier variant of c2 (synthetic.HARD_PROFILE: confidences all over (0, 1), ~40 % of the matches wrong): what
        # the matcher and -- above all -- the reference-policy RANSAC cost when frames are not clean (pose parity on it: tests/test_gpu_parity.py)
        ch = side_leg(["--workload", "c2_hard", "--steps", "40", "--warmup", "5"])
        # the same c2 region (same steps, same PnP policy) with the FINE stage on plain bf16 operands (config["hip_fine_precision"] = "bf16" /
        # OPHIP_FINE_PRECISION=bf16): indices bit-exact, pose within 1e-5 of the default's, keypoints within 0.05 px (test_fine_stage_in_plain_bf16...).
        # A side value: the headline keeps the arithmetic whose keypoints stay within 1e-4 relative of the reference's
        fb = side_leg(["--workload", "c2", "--steps", str(args.steps), "--warmup", str(args.warmup)], env={"OPHIP_FINE_PRECISION": "bf16"}) if args.precision == "bf16x3" else {}
        result["value_fine_bf16"] = fb.get("value")
        result["value_c3_b32"] = c3.get("value")
        result["value_c4"] = c4.get("value")
        result["value_c2_hard"] = ch.get("value")
        result["side_legs"] = {
            "fine_bf16": {k: fb.get(k) for k in ("value", "ms_per_step", "steps", "dtype", "error") if k in fb}
                         | {"what": "c2 with the fine stage alone on plain bf16 operands (one matrix instruction per product); coarse stage unchanged: match "
                                    "indices bit-exact, pose |dR|, |dt|/|t| <= 7e-6 against the reference's / the oracle's matches on c1, c1_hard, c2, c2_hard, "
                                    "keypoints within 0.05 px (tests/test_gpu_parity.py::test_fine_stage_in_plain_bf16_keeps_indices_and_pose)"},
            "c2_hard": {k: ch.get(k) for k in ("value", "ms_per_step", "steps", "error") if k in ch}
                       | {"workload": (ch.get("config") or {}).get("workload"), "matches_per_frame": (ch.get("config") or {}).get("matches_per_frame"),
                          "pnp_inliers_per_frame": (ch.get("config") or {}).get("pnp_inliers_per_frame"),
                          "pnp_ceiling_fps": (ch.get("host") or {}).get("pnp_ceiling_fps"), "host_bound": (ch.get("host") or {}).get("host_bound"),
                          "pnp_threads": (ch.get("host") or {}).get("pnp_threads_per_rank")},
            "c3": {k: c3.get(k) for k in ("value", "ms_per_step", "steps", "error") if k in c3} | {"workload": (c3.get("config") or {}).get("workload")},
            "c4": {k: c4.get(k) for k in ("value", "ms_per_step", "steps", "error") if k in c4} | {"workload": (c4.get("config") or {}).get("workload"),
                                                                                                     "conf_kernel": c4.get("roofline")},
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import onepose_oracle as orc          # cpu_baseline leg only
        import statistics

        cores = host_cores()
        budget = args.cpu_seconds

        def time_frames(fn, n_max, seconds):
            """1 warm-up, then up to n_max frames or `seconds`; -> (median frames/s, n)"""
            fn()
            ts, t_all = [], time.perf_counter()
            while len(ts) < n_max and (time.perf_counter() - t_all) < seconds:
                t1 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t1)
            return 1.0 / statistics.median(ts), len(ts)

        def feature_frame():
            orc.forward_from_features(sd, cfg, first, first["feat_c"], first["feat_f"], image_hw)

        img = torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(7))

        def full_frame():          # backbone on the synthetic image + the path on the planted maps (a random image has no matches)
            orc.backbone_8_2(sd, img)
            feature_frame()

        with torch.no_grad():
            torch.set_num_threads(cores)
            v_all, n_all = time_frames(feature_frame, 7, 0.45 * budget)
            v_full, n_full = time_frames(full_frame, 3, 0.25 * budget)
            torch.set_num_threads(1)
            v_one, n_one = time_frames(feature_frame, 2, 0.3 * budget)
            torch.set_num_threads(cores)
        result["cpu_baseline"] = {
            "value": v_all, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"median of {n_all} frame(s) of the same {args.workload} workload (feature boundary, B=1) after 1 warm-up, torch fp32 "
                      f"CPU oracle, {cores} threads; matcher only (no PnP)",
            "single_thread": {"value": v_one, "frames": n_one},
            "full_forward": {"value": v_full, "frames": n_full, "cores": cores, "note": "oracle backbone (ResNet-FPN) + matcher"},
        }
    if rank == 0:
        print(json.dumps(result))
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
