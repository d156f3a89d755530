"""Diagnostic: per-phase cycle stamps of fine_refine_bf16 at c2 (prints medians over workgroups). Not part of the product."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from onepose_st_amd import hip
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict
cfg = default_config(); cfg["hip_precision"] = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
sd = make_synthetic_state_dict(0, cfg)
if __import__("os").environ.get("STAMPS_FINE_LAYERS"):          # experiment: another order of the two fine layers (same weights)
    cfg["loftr_fine"]["layer_names"] = __import__("os").environ["STAMPS_FINE_LAYERS"].split(",")
    print("fine layers", cfg["loftr_fine"]["layer_names"])
dev = torch.device("cuda:0")
m = OnePosePlus_model(cfg).eval(); m.load_state_dict(sd); m.to(dev)
inp = make_synthetic_inputs(sd, 7000, (480, 640), 3000, seed=1, config=cfg)
d = {k: inp[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
fc, ff = inp["feat_c"].to(dev), inp["feat_f"].to(dev)
for _ in range(3):
    m.forward_features(dict(d), fc, ff, inp["image_hw"])
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
import ctypes
hip.call("ophip_debug_stamps", ctypes.c_void_p(buf.data_ptr()))
data = dict(d); m.forward_features(data, fc, ff, inp["image_hw"])
torch.cuda.synchronize()
hip.call("ophip_debug_stamps", None)
K = data["i_ids"].numel()
PAIR = __import__("os").environ.get("OPHIP_FINE_PAIR", "1") != "0"
nwg = (K + 1) // 2 if PAIR else K
s = buf.view(-1, 32)[:nwg].cpu().numpy().astype(np.int64)
names = {0: "start", 1: "gather", 31: "end"}
for l in range(2):
    for i, n in enumerate(["q gemm (both matches)" if PAIR else "qkv gemm", "2 x (kv gemm + attention) + sync" if PAIR else "attention+sync", "merge gemm", "LN1+sync", "mlp0 gemm",
                           "sync + H store + sync" if PAIR else "H store+sync", "mlp2 gemm", "LN2+X+sync"]):
        names[2 + 8 * l + i] = f"L{l} {n}"
order = sorted(names)
prev = s[:, 0]
print("library build", hip.build_stamp(), "| two matches per workgroup" if PAIR else "| one match per workgroup")
print("workgroups", nwg, "total cycles median", np.median(s[:, 31] - s[:, 0]), " (s_memtime ticks = shader cycles)")
for k in order[1:]:
    print(f"{names[k]:34s} {np.median(s[:, k] - prev):10.0f}")
    prev = s[:, k]
first, last = s[:, 0].min(), s[:, 31].max()
print("kernel span (cycles)", last - first, " mean WG duration", (s[:, 31] - s[:, 0]).mean())
