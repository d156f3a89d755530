"""Diagnostic: per-phase cycle stamps of one backbone convolution (csrc/conv.hip).  Not part of the product."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
cin, cout, H, W, ks, stride = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (128, 128, 240, 320, 3, 1))]
use_res = len(sys.argv) > 7 and sys.argv[7] == "res"
dev = torch.device("cuda:0"); hip.load()
cip, cop = packing.pad32(cin), packing.pad32(cout)
g = torch.Generator().manual_seed(0)
xh = torch.randn(1, H, W, cip, generator=g).to(dev).to(torch.bfloat16); xl = (0.003 * torch.randn(1, H, W, cip, generator=g)).to(dev).to(torch.bfloat16)
w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
wp = packing.pack_conv_bf16(w, torch.zeros(cout)).to(dev)
Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
oh = torch.empty(1, Ho, Wo, cop, dtype=torch.bfloat16, device=dev); ol = torch.empty_like(oh)
rh = torch.randn(1, Ho, Wo, cop, generator=g).to(dev).to(torch.bfloat16) if use_res else None
rl = torch.zeros_like(rh) if use_res else None
P = hip.ptr
def run():
    hip.call("ophip_conv2d_bf16", P(xh, None), P(xl, None), 1, H, W, cip, P(wp, None), cop, ks, stride, 1, P(rh, None), P(rl, None), None, 0, 0, None,
             P(oh, None), P(ol, None), None, 0, 3, hip.stream_handle())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
print(f"{cin}->{cout} k{ks} s{stride} @{Ho}x{Wo}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch")
nwg = 65536
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
hip.call("ophip_debug_stamps", ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); hip.call("ophip_debug_stamps", None)
s = buf.view(-1, 32).cpu().numpy().astype(np.int64)
s = s[s[:, 0] > 0]
ncc = cip // 32
d = s[:, 31] - s[:, 0]
print("workgroups", len(s), "WG cycles p10/p50/p90", np.percentile(d, 10), np.median(d), np.percentile(d, 90), "kernel span", s[:, 31].max() - s[:, 0].min())
st = np.zeros(len(s)); cp = np.zeros(len(s)); wt = np.zeros(len(s))
for cc in range(min(ncc, 8)):
    prev = s[:, 0] if cc == 0 else s[:, 3 * cc]
    wt += s[:, 1 + 3 * cc] - prev; st += s[:, 2 + 3 * cc] - s[:, 1 + 3 * cc]; cp += s[:, 3 + 3 * cc] - s[:, 2 + 3 * cc]
print(f"barrier wait {np.median(wt):.0f}  staging {np.median(st):.0f}  compute {np.median(cp):.0f}  epilogue {np.median(s[:, 31] - s[:, 30]):.0f}   (cycles, summed over {min(ncc, 8)} chunks)")
