"""Diagnostic: host-side (Python + ctypes) time of enqueue_features / finish per frame at c2."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
HC = not (len(sys.argv) > 2 and sys.argv[2] == "nocopy")
_p = model.enqueue_features(dict(obj), fc, ff, hw, host_copy=True); _p.finish(); stale = _p.host
for _ in range(5): model.enqueue_features(dict(obj), fc, ff, hw, host_copy=HC).finish()
torch.cuda.synchronize()
from onepose_st_amd.pnp import PnPPool
pool = PnPPool(inp["K"].numpy(), threads=3, pnp_reprojection_error=7) if len(sys.argv) > 1 and sys.argv[1] == "pnp" else None
stale = stale
N = 200
t_enq = t_fin = t_wait = 0.0
prev = None
tw0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); cur = model.enqueue_features(dict(obj), fc, ff, hw, host_copy=HC); t_enq += time.perf_counter() - t
    if prev is not None:
        t = time.perf_counter(); prev.wait(); t_wait += time.perf_counter() - t
        t = time.perf_counter(); prev.finish()
        if pool is not None:
            if prev.host is not None:
                stale = prev.host
            pool.submit(stale["mkpts_2d"], stale["mkpts_3d_db"])
        t_fin += time.perf_counter() - t
    prev = cur
prev.finish(); torch.cuda.synchronize()
wall = time.perf_counter() - tw0
print(f"per frame: wall {1e6 * wall / N:.0f} us, host enqueue {1e6 * t_enq / N:.0f} us, host finish {1e6 * t_fin / N:.0f} us, event wait {1e6 * t_wait / N:.0f} us")
