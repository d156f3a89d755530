"""Diagnostic: per-kernel times of the coarse-matching stage at c2 (planted encoder-boundary features).  Not part of the product."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip
hip.load(); dev = torch.device("cuda:0")
B, N, M, wc = 1, 7000, 4800, 80
g = torch.Generator().manual_seed(0)
f3 = torch.randn(B, N, 256, generator=g) * 2.5
f2 = torch.randn(B, M, 256, generator=g) * 2.5
perm = torch.randperm(M, generator=g)[:3000]
f2[0, perm] = f3[0, :3000] + 0.1 * torch.randn(3000, 256, generator=g)
f3, f2 = f3.to(dev), f2.to(dev)
kp = torch.randn(B, N, 3).to(dev)
cap = B * N
conf = torch.empty(B, N, M, device=dev)
ws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev)
ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(4)]
mconf, mk3, mkc = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
gt = torch.empty(cap, dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
def run():
    hip.call("ophip_coarse_match", hip.ptr(f3), hip.ptr(f2), hip.ptr(kp), 0, B, N, M, wc, 0.08, 0.1, 2, 8.0, hip.ptr(conf), hip.ptr(ws),
             hip.ptr(ids[0], torch.int64), hip.ptr(ids[1], torch.int64), hip.ptr(ids[2], torch.int64), hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc),
             hip.ptr(ids[3], torch.int64), hip.ptr(gt, torch.uint8), hip.ptr(cnt, torch.int32), 3, hip.stream_handle())
for _ in range(3): run()
torch.cuda.synchronize()
print("matches", int(cnt.item()))
tot = 0.0
for name in ("frag_planes", "sim_stats", "stat_combine", "conf", "select"):
    hip.timing_select(name)
    for _ in range(20): run()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    if n:
        print(f"{name:14s} {ms / 20 * 1e3:7.1f} us per frame ({n // 20} launch(es))"); tot += ms / 20 * 1e3
print(f"sum            {tot:7.1f} us")
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print(f"stage wall     {s.elapsed_time(e) / 50 * 1e3:7.1f} us per frame")
