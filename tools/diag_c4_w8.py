import os, sys, torch
sys.path.insert(0, "/root/repo")
from onepose_st_amd.config import default_config
from onepose_st_amd.model import OnePosePlus_model
from onepose_st_amd.synthetic import CONFIG_SIZES, make_synthetic_inputs, make_synthetic_state_dict
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg); dev = torch.device("cuda:0")
n_points, image_hw, n_plant = CONFIG_SIZES["c4"]
inp = make_synthetic_inputs(sd, n_points, image_hw, n_plant, seed=2, config=cfg)
res = {}
for w8 in (0, 1):
    os.environ["OPHIP_ENC_W8"] = "1" if w8 else "0"
    m = OnePosePlus_model(cfg).eval(); m.load_state_dict(sd, strict=True); m.to(dev)
    d = {k: inp[k].to(dev) for k in ("keypoints3d", "descriptors3d_db", "descriptors3d_coarse_db")}
    m.forward_features(d, inp["feat_c"].to(dev), inp["feat_f"].to(dev), inp["image_hw"])
    res[w8] = (d["i_ids"].cpu(), d["j_ids"].cpu(), d["mconf"].cpu(), d["conf_matrix"])
i0, j0, c0, cm0 = res[0]; i1, j1, c1, cm1 = res[1]
print(len(i0), len(i1))
s0, s1 = set(i0.tolist()), set(i1.tolist())
for i in sorted(s0 ^ s1):
    k = (i0 == i).nonzero()
    if len(k):
        k = int(k[0]); j = int(j0[k])
        print("only in 4-wave: i", i, "j", j, "mconf", float(c0[k]), "w8 conf at (i,j)", float(cm1[0, i, j]), "w8 row max", float(cm1[0, i].max()), "w8 col max", float(cm1[0, :, j].max()))
print("max |conf diff|", float((cm0 - cm1).abs().max()), "max rel diff of mconf on common", float(((c0[:100]-c1[:100]).abs()/c0[:100]).max()))
