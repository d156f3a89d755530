"""Diagnostic: cycle stamps of sim_stats_bf16 at c2. Not part of the product."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
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
print("WGs", len(s))
for n in range(7):
    t = s[:, 4 * n: 4 * n + 4]
    nxt = s[:, 4 * n + 4] if n < 6 else None
    print(f"tile {n}: k-steps {np.median(t[:,1]-t[:,0]):7.0f}  E+F {np.median(t[:,2]-t[:,1]):6.0f}  out/stage+G {np.median(t[:,3]-t[:,2]):6.0f}" + (f"  to next {np.median(nxt - t[:,3]):5.0f}" if nxt is not None else ""))
print("step 5 of tile 1: issue", np.median(s[:,29]-s[:,28]), "mfma+reads issue", np.median(s[:,30]-s[:,29]), "to end of step 6", np.median(s[:,31]-s[:,30]))
