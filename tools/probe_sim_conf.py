"""Diagnostic (by hand): the confidence pass of one frame BESIDE the similarity tiles of the next -- what a pipeline that delays conf(t) to
run beside sim(t + 1) would see.  Two workspaces / conf buffers; stream A runs the similarity tiles of frame 1 while stream B runs the
statistics merge + confidence pass of frame 0.  Prints each alone and both together (wall per iteration)."""
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


def frame():
    return dict(conf=torch.empty(B, N, M, device=dev), ws=torch.empty(hip.load().ophip_coarse_workspace_floats(B, N, M), device=dev),
                ids=[torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(4)], mconf=torch.empty(cap, device=dev),
                mk3=torch.empty(cap, 3, device=dev), mkc=torch.empty(cap, 2, device=dev), gt=torch.empty(cap, dtype=torch.uint8, device=dev),
                cnt=torch.zeros(2, dtype=torch.int32, device=dev))


def call(fr, parts, stream):
    P = hip.ptr
    hip.call("ophip_coarse_match_masked", P(f3), P(f2), P(kp), 0, B, N, M, wc, 0.08, 0.1, 2, 8.0, P(fr["conf"]), P(fr["ws"]),
             P(fr["ids"][0], torch.int64), P(fr["ids"][1], torch.int64), P(fr["ids"][2], torch.int64), P(fr["mconf"]), P(fr["mk3"]), P(fr["mkc"]),
             P(fr["ids"][3], torch.int64), P(fr["gt"], torch.uint8), P(fr["cnt"], torch.int32), 3, parts, None, None, __import__("ctypes").c_void_p(stream.cuda_stream))


A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
f0, f1 = frame(), frame()
for fr in (f0, f1):
    call(fr, 3, A)
torch.cuda.synchronize()


def timed(fn, n=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(A)
    for _ in range(n):
        fn()
    Bs.synchronize()
    e.record(A); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def sim_only():
    call(f1, 4, A)


def conf_only():
    call(f0, 8, A)            # (re-running the pass on an already converted buffer has the same traffic)
    call(f0, 4, A) if False else None


def both():
    ev = torch.cuda.Event(); ev.record(A); Bs.wait_event(ev)
    call(f1, 4, A)
    call(f0, 8, Bs)
    ev2 = torch.cuda.Event(); ev2.record(Bs); A.wait_event(ev2)


for fn, label in ((sim_only, "alone"), (both, "side by side")):
    for name in ("sim_stats", "conf"):
        if fn is sim_only and name == "conf":
            continue
        hip.timing_select(name)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        n, ms = hip.timing_read(); hip.timing_select("")
        print(f"{name:10s} {label:13s} {ms / max(n, 1) * 1e3:7.1f} us per launch ({n} launches)")
hip.timing_select("conf")
for _ in range(30):
    conf_only()
torch.cuda.synchronize()
n, ms = hip.timing_read(); hip.timing_select("")
print(f"conf       alone         {ms / max(n, 1) * 1e3:7.1f} us per launch ({n} launches)")
