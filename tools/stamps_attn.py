"""Diagnostic: per-phase cycle stamps of attn_apply_bf16 for one coarse layer at c2. Not part of the product."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
nsplit = 3 if (len(sys.argv) < 2 or sys.argv[1] == "bf16x3") else 1
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0")
L3, L2 = 7000, 4800
g = torch.Generator().manual_seed(0)
x3, x2 = torch.randn(1, L3, 256, generator=g).to(dev), torch.randn(1, L2, 256, generator=g).to(dev)
y3, y2 = torch.empty_like(x3), torch.empty_like(x2)
w = packing.pack_coarse_layer_bf16(sd, "loftr_coarse.layers.0.").to(dev)
ws = torch.empty(hip.load().ophip_encoder_bf16_workspace_bytes(1, L3, L2), dtype=torch.uint8, device=dev)
def run():
    hip.call("ophip_encoder_layer_bf16", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3), hip.ptr(y2), 1, L3, L2, hip.ptr(w, None), (hip.ptr(w, None) if __import__('os').environ.get('FUSE') else None), nsplit, 0, 0, 0, hip.ptr(ws, None), hip.stream_handle())
for _ in range(3): run()
buf = torch.zeros(1024 * 32, dtype=torch.int64, device=dev)
hip.call("ophip_debug_stamps", ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); hip.call("ophip_debug_stamps", None)
TOK = 32 * int(__import__('os').environ.get('OPHIP_ENC_TT', '1'))
nwg = (L3 + TOK - 1) // TOK + (L2 + TOK - 1) // TOK
s = buf.view(-1, 32)[:nwg].cpu().numpy().astype(np.int64)
names = {0: "start", 1: "X load+sync", 2: "Q gemm", 3: "attention+store+sync", 4: "merge gemm", 5: "LN1+store+sync", 22: "LN2", 30: "stage+residual store", 31: "fused next-layer kv slabs"}
for c in range(4):
    names[6 + 4 * c] = f"c{c} mlp0 gemm"; names[7 + 4 * c] = f"c{c} relu+H store+sync"; names[8 + 4 * c] = f"c{c} mlp2 gemm"; names[9 + 4 * c] = f"c{c} sync"
prev = s[:, 0]
d = s[:, 31] - s[:, 0]
print("workgroups", nwg, "WG cycles p10/p50/p90", np.percentile(d, 10), np.median(d), np.percentile(d, 90), "span", s[:, 31].max() - s[:, 0].min())
for k in sorted(names)[1:]:
    print(f"{names[k]:24s} {np.median(s[:, k] - prev):9.0f}")
    prev = s[:, k]
