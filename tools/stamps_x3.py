"""Diagnostic: per-phase cycle stamps and launch time of the eight-wave x3 encoder layer kernel (csrc/encoder_x3w8.hip, the shipped default) at c2 (or --tokens N).  Not part of the product."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0"); hip.load()
B = int(os.environ.get("B", "1"))
L3, L2 = 7000, 4800
g = torch.Generator().manual_seed(0)
x3, x2 = torch.randn(B, L3, 256, generator=g).to(dev), torch.randn(B, L2, 256, generator=g).to(dev)
y3, y2 = torch.empty_like(x3), torch.empty_like(x2)
ENTRY = "ophip_encoder_layer_x3w8"
w = packing.pack_coarse_layer_x3w8(sd, "loftr_coarse.layers.0.").to(dev)
ws = torch.empty(hip.load().ophip_encoder_x3w8_workspace_bytes(B, L3, L2), dtype=torch.uint8, device=dev)
fuse = not os.environ.get("NOFUSE")
def run():
    hip.call(ENTRY, hip.ptr(x3), hip.ptr(x2), hip.ptr(y3), hip.ptr(y2), B, L3, L2, hip.ptr(w, None),
             hip.ptr(w, None) if fuse else None, 0, 0, 0, hip.ptr(ws, None), hip.stream_handle())
for _ in range(3): run()
torch.cuda.synchronize()
import time
t_end = time.time() + 2.5                       # the chip settles on the clock it holds under this load (>= 2 s of back-to-back launches)
while time.time() < t_end:
    for _ in range(200): run()
    torch.cuda.synchronize()
for name in ("attn_apply", "kv_reduce", "kv_sum"):
    hip.timing_select(name)
    for _ in range(30): run()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    print(f"{name:12s} {ms / n * 1e3:7.1f} us per launch (B={B}, fused tail {fuse})")
nwg = B * ((L3 + 47) // 48 + (L2 + 47) // 48)
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
for _ in range(50): run()
hip.call("ophip_debug_stamps", ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); hip.call("ophip_debug_stamps", None)
s = buf.view(-1, 32)[:nwg].cpu().numpy().astype(np.int64)
names = ["start", "rows in + ring fill + stage + sync", "Q gemm", "attention + msg store + sync", "merge gemm", "LN1 + store + sync",
         "W0c0 + hidden store + sync", "W2c0", "W0c1 + sync + hidden store + sync", "W2c1", "LN2 + residual + store", "K|V tail"]
last = len(names) - 1
d = s[:, last] - s[:, 0]
print("library build", hip.build_stamp())
print("workgroups", nwg, "WG cycles p10/p50/p90", np.percentile(d, 10), np.median(d), np.percentile(d, 90))
real = (s[:, 31] - s[:, 30]).astype(np.float64)                 # 100 MHz ticks between the first and the last stamp
ok = real > 0
clk = d[ok] / real[ok] * 0.1
print(f"in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz), median over workgroups: {np.median(clk):.3f} GHz "
      f"(p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}); workgroup time {np.median(real[ok]) * 0.01:.1f} us")
mf = 1482 * 2 * 16                                              # MFMAs per wave x 2 waves per SIMD x 16 cycles
print(f"matrix pipe: {mf} issue cycles per SIMD of {np.median(d):.0f} = {mf / np.median(d):.3f} busy at the clock the kernel holds")
for k in range(1, last + 1):
    print(f"{names[k]:48s} {np.median(s[:, k] - s[:, k - 1]):9.0f}")
