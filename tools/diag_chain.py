"""Diagnostic: the three-layer chain of tests/test_gpu_parity.py::test_encoder_x3_chain_with_fused_kv_tail, layer by layer: fused tail against the
stand-alone K / V half against the oracle.  Not part of the product."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import onepose_oracle as orc
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0"); hip.load()
B, L3, L2 = 1, 100, 75
g = torch.Generator().manual_seed(5)
x3, x2 = torch.randn(B, L3, 256, generator=g), torch.randn(B, L2, 256, generator=g)
names = ["self", "cross", "self"]
ws_ = [packing.pack_coarse_layer_x3w8(sd, f"loftr_coarse.layers.{li}.").to(dev) for li in range(3)]
ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
def chain(fused, reps=1):
    outs = []
    a3, a2 = x3.to(dev), x2.to(dev)
    b3, b2 = torch.full_like(a3, float("nan")), torch.full_like(a2, float("nan"))
    for li, nm in enumerate(names):
        nxt = ws_[li + 1] if (fused and li + 1 < 3) else None
        hip.call("ophip_encoder_layer_x3w8", hip.ptr(a3), hip.ptr(a2), hip.ptr(b3), hip.ptr(b2), B, L3, L2, hip.ptr(ws_[li], None),
                 hip.ptr(nxt, None), 1 if nm == "cross" else 0, 1 if (fused and li > 0) else 0, (li & 1) if fused else 0, hip.ptr(ws, None), hip.stream_handle())
        torch.cuda.synchronize()
        t3, t2 = (L3 + 47) // 48, (L2 + 47) // 48
        part_floats = B * (t3 + t2) * (8 * 1024 + 8 * 32)
        kvoff = (2 * part_floats * 4 + 255 + ws.data_ptr() % 256) // 256 * 256 - ws.data_ptr() % 256
        kvblk = ws[2 * part_floats * 4: 2 * part_floats * 4 + 256 + 2 * B * (32768 + 1024)].clone().cpu()
        slabs = ws[:2 * part_floats * 4].view(torch.float32).clone().cpu()
        outs.append((b3.clone().cpu(), b2.clone().cpu(), kvblk, slabs))
        a3, b3, a2, b2 = b3, a3, b2, a2
    return outs
r3, r2 = x3, x2
refs = []
for li, nm in enumerate(names):
    p = f"loftr_coarse.layers.{li}."
    if nm == "cross":
        r2, r3 = orc.encoder_layer(sd, p, r2, r3, 8), orc.encoder_layer(sd, p, r3, r2, 8)
    else:
        r2, r3 = orc.encoder_layer(sd, p, r2, r2, 8), orc.encoder_layer(sd, p, r3, r3, 8)
    refs.append((r3, r2))
f, s_, s2 = chain(True), chain(False), chain(False)
for li in range(3):
    d = lambda a, b: f"{(a - b).abs().max().item():.3e}"
    kvd = (f[li][2] != s_[li][2]).sum().item()
    pf = f[li][3].view(2, -1); ps = s_[li][3].view(2, -1)
    # fused layer li reads slabs of slot li & 1 (written by layer li - 1's tail); stand-alone always slot 0 (its own kv_reduce)
    sl_f, sl_s = pf[li & 1], ps[0]
    print(f"layer {li}: K^T V block bytes differing {kvd}; slabs read by this layer: max abs diff {(sl_f - sl_s).abs().max().item():.3e}")
    print(f"layer {li}: fused vs standalone 3D {d(f[li][0], s_[li][0])} 2D {d(f[li][1], s_[li][1])} | standalone twice {d(s_[li][0], s2[li][0])} | "
          f"fused vs oracle 3D {d(f[li][0], refs[li][0])} 2D {d(f[li][1], refs[li][1])} | standalone vs oracle 3D {d(s_[li][0], refs[li][0])} 2D {d(s_[li][1], refs[li][1])}")
