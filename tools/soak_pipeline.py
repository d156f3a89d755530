"""Soak test of the frame pipeline on the GPU (by hand: `python tools/soak_pipeline.py [frames] [seed]`): a long random sequence of
enqueue / finish (oldest, newest or a random frame in flight) / drop-unfinished / flush on one compute stream, with the input kernels
on their side stream, up to five frames in flight; every finished frame's outputs are compared bit for bit with that frame's
stand-alone run.  Exercises the kept-back fine stage (csrc/frame.hip) across the ring of 16 event sets many times."""
import hashlib, os, random, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import make_synthetic_inputs, make_synthetic_state_dict

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg); dev = torch.device("cuda:0")
m = OnePosePlus_model(cfg).eval(); m.load_state_dict(sd, strict=True); m.to(dev)
frames = [make_synthetic_inputs(sd, n_points=1500, image_hw=(160, 224), n_plant=500, seed=21, config=cfg, frame=f) for f in range(5)]
obj = {k: frames[0][k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
feats = [(f["feat_c"].to(dev), f["feat_f"].to(dev)) for f in frames]
keys = ("i_ids", "j_ids", "mconf", "mkpts_query_c", "mkpts_query_f", "expec_f", "mkpts_3d_db")


def digest(d, p):
    h = hashlib.sha256()
    for k in keys:
        h.update(d[k].cpu().numpy().tobytes())
    h.update(p.host["mkpts_2d"].tobytes())
    return h.hexdigest()


ref = []
for fc, ff in feats:
    d = dict(obj)
    p = m.enqueue_features(d, fc, ff, frames[0]["image_hw"], host_copy=True)
    p.finish(); torch.cuda.synchronize()
    ref.append(digest(d, p))
assert len(set(ref)) == len(ref)
st = torch.cuda.Stream(device=dev)
bad = done = dropped = 0
with torch.cuda.stream(st):
    inflight = []
    for i in range(n_frames):
        f = rng.randrange(len(feats))
        d = dict(obj)
        inflight.append((f, d, m.enqueue_features(d, *feats[f], frames[0]["image_hw"], host_copy=True, inputs_ready=rng.random() < 0.8)))
        r = rng.random()
        if r < 0.05:
            inflight.pop(rng.randrange(len(inflight))); dropped += 1          # dropped unfinished
        elif r < 0.10:
            m.flush()
        while len(inflight) > rng.choice([0, 1, 2, 2, 3, 3, 4]):
            k = rng.choice([0, 0, 0, len(inflight) - 1, rng.randrange(len(inflight))])
            f0, d0, p0 = inflight.pop(k)
            p0.finish(); done += 1
            bad += digest(d0, p0) != ref[f0]
    while inflight:
        f0, d0, p0 = inflight.pop(0); p0.finish(); done += 1
        bad += digest(d0, p0) != ref[f0]
torch.cuda.synchronize()
print(f"soak: {n_frames} frames enqueued, {done} finished, {dropped} dropped unfinished, mismatches {bad}")
sys.exit(1 if bad else 0)
