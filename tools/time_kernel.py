"""Diagnostic: HIP-event time of one kernel family of the coarse-matching stage at c2 (ophip_timing_select)."""
import sys, os, torch
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
for name in sys.argv[1:] or ["sim_stats", "stat_combine", "conf", "select"]:
    torch.cuda.synchronize(); hip.timing_select(name)
    for _ in range(20): run()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    print(f"{name:14s} {ms / max(n, 1) * 1e3:8.1f} us  ({n} launches)")
