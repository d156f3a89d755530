"""Diagnostic: cycle stamps of the similarity tile kernel at c2 (sim_frag_kernel, or sim_frag3_kernel with OPHIP_SIM_TILE=3).  Not part of the product.
Stamps per workgroup: 0 start | 1 end of the k-loop | 2 tile maximum known | 3 end of the store + statistics pass | 4 end."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip
dev = torch.device("cuda:0"); hip.load()
N, M, wc = 7000, 4800, 80
g = torch.Generator().manual_seed(0)
f3, f2 = torch.randn(1, N, 256, generator=g).to(dev), torch.randn(1, M, 256, generator=g).to(dev)
kp = torch.zeros(1, N, 3, device=dev)
conf = torch.empty(1, N, M, device=dev); ws = torch.empty(hip.load().ophip_coarse_workspace_floats(1, N, M), device=dev)
ids = [torch.empty(N, dtype=torch.int64, device=dev) for _ in range(3)]
mconf, mk3, mkc = torch.empty(N, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 2, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
def run():
    hip.call("ophip_coarse_match", hip.ptr(f3), hip.ptr(f2), hip.ptr(kp), kp.stride(0), 1, N, M, wc, 0.08, 0.1, 2, 8.0, hip.ptr(conf), hip.ptr(ws),
             *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(cnt, torch.int32), 3, hip.stream_handle())
for _ in range(3): run()
nwg = 8 * 7 * 38
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
hip.call("ophip_debug_stamps", ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); hip.call("ophip_debug_stamps", None)
s = buf.view(-1, 32).cpu().numpy().astype(np.int64)
s = s[s[:, 0] > 0]
print("library build", hip.build_stamp(), "| tile kernel", os.environ.get("OPHIP_SIM_TILE", "2 (default)"), "| workgroups", len(s))
for k, name in ((1, "k-loop (16 k-steps)"), (2, "scale + tile maximum + sync"), (3, "stage + store + exponentials (+ syncs)"), (4, "sums across waves, partial records / exact pass")):
    print(f"{name:52s} {np.median(s[:, k] - s[:, k - 1]):9.0f}")
d = s[:, 4] - s[:, 0]
print("workgroup cycles p10 / p50 / p90 / max", np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max())
t0 = s[:, 0].min()
print("kernel span", s[:, 4].max() - t0, "cycles; last workgroup START at", s[:, 0].max() - t0, "; slowest decile of workgroups (cycles):", np.sort(d)[-len(d) // 10:][::20])
