"""Diagnostic: where the host time of one frame's enqueue goes at c2 -- the C call (ophip_frame_enqueue_padded: ~35 kernel launches, events,
stream waits) against the Python around it (plan lookup, block allocation, pinned buffer, argument marshalling); depth 3 as in bench.py."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict, CONFIG_SIZES
dev = torch.device("cuda:0")
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg)
n, hw, pl = CONFIG_SIZES["c2"]
inp = make_synthetic_inputs(sd, n, hw, pl, seed=1, config=cfg)
model = OnePosePlus_model(cfg).eval(); model.load_state_dict(sd); model.to(dev)
obj = {k: inp[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
fc, ff = inp["feat_c"].to(dev), inp["feat_f"].to(dev)
acc = {}
real_call = hip.call
def timed_call(name, *a):
    t = time.perf_counter()
    try:
        return real_call(name, *a)
    finally:
        e = acc.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t
hip.call = timed_call
import onepose_st_amd.ops as ops, onepose_st_amd.model as mdl
for _ in range(20):
    model.enqueue_features(dict(obj), fc, ff, hw, host_copy=True, inputs_ready=True).finish()
torch.cuda.synchronize()
acc.clear()
import gc
from onepose_st_amd import hostsize
hostsize.pin_rank(0, 1)
gc.collect(); gc.disable()
PROFILE = len(sys.argv) > 1 and sys.argv[1] == "profile"
if PROFILE:
    import cProfile, pstats
    prof = cProfile.Profile()
N, depth = 2000, 3
fl = []
t_enq = t_fin = 0.0
tw0 = time.perf_counter()
for i in range(N):
    t = time.perf_counter()
    if PROFILE: prof.enable()
    fl.append(model.enqueue_features(dict(obj), fc, ff, hw, host_copy=True, inputs_ready=True))
    if PROFILE: prof.disable()
    t_enq += time.perf_counter() - t
    if len(fl) >= depth:
        p = fl.pop(0); p.wait()
        t = time.perf_counter(); p.finish(); t_fin += time.perf_counter() - t
model.flush()
for p in fl: p.finish()
torch.cuda.synchronize()
wall = time.perf_counter() - tw0
print(f"per frame: wall {1e6 * wall / N:.0f} us; enqueue_features {1e6 * t_enq / N:.0f} us; finish (after the wait) {1e6 * t_fin / N:.0f} us")
for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {c / N:5.2f} calls/frame  {1e6 * s / N:7.1f} us/frame")
if PROFILE:
    st = pstats.Stats(prof); st.sort_stats("tottime").print_stats(28)
