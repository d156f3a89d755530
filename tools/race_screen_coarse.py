"""Diagnostic: run the coarse stage many times on the same inputs (several sizes) and require bit-identical outputs every time --
a screen for ordering bugs in the LDS-DMA ring of sim_frag (counted vmcnt + barrier).  Not part of the product."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip
hip.load(); dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for (B, N, hc, wc) in ((1, 7000, 60, 80), (2, 3000, 40, 52), (1, 1000, 30, 40), (3, 517, 13, 21), (1, 15000, 120, 160)):
    M = hc * wc
    g = torch.Generator().manual_seed(N)
    f3 = torch.randn(B, N, 256, generator=g) * 1.5
    f2 = torch.randn(B, M, 256, generator=g)
    P = min(N, M) // 2
    for b in range(B):
        cells = torch.randperm(M, generator=g)[:P]
        f2[b, cells] = f3[b, :P] * 1.5 + 0.1 * torch.randn(P, 256, generator=g)
    f3, f2, kp = f3.to(dev), f2.to(dev), torch.randn(B, N, 3, generator=g).to(dev)
    cap = B * N
    conf = torch.empty(B, N, M, device=dev)
    ws = torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev)
    ids = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(3)]
    mconf, mk3, mkc = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    first = None
    n = reps if N < 10000 else max(10, reps // 10)
    for r in range(n):
        conf.fill_(float("nan"))
        hip.call("ophip_coarse_match", hip.ptr(f3), hip.ptr(f2), hip.ptr(kp), kp.stride(0), B, N, M, wc, 0.08, 0.1, 2, 8.0, hip.ptr(conf), hip.ptr(ws),
                 *[hip.ptr(t, torch.int64) for t in ids], hip.ptr(mconf), hip.ptr(mk3), hip.ptr(mkc), None, None, hip.ptr(cnt, torch.int32), 3, hip.stream_handle())
        K = int(cnt.item())
        sig = (K, float(conf.double().sum().item()), int(ids[1][:K].sum().item()), int(ids[2][:K].sum().item()), float(mconf[:K].double().sum().item()))
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
            print("MISMATCH", (B, N, M), r, sig, first)
    print(f"B={B} N={N} M={M}: {n} runs, K={first[0]}, {'identical' if bad == 0 else 'DIFFERENT'}", flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
