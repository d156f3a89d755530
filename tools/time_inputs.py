"""Diagnostic: HIP-event time of the frame's input kernels and of the selection at c2, stand-alone (ophip_timing_select).  Not part of the product."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
cfg = default_config(); sd = make_synthetic_state_dict(0, cfg); dev = torch.device("cuda:0"); hip.load()
B, N, hc, wc = 1, 7000, 60, 80
M, hf, wf = hc * wc, 4 * hc, 4 * wc
g = torch.Generator().manual_seed(0)
feat_c = torch.randn(B, 256, hc, wc, generator=g).to(dev); pe = torch.randn(M, 256, generator=g).to(dev); x2d = torch.empty(B, M, 256, device=dev)
feat_f = torch.randn(B, 128, hf, wf, generator=g).to(dev); ffcl = torch.empty(B, hf * wf, 128, device=dev)
kp = torch.randn(B, N, 3, generator=g).to(dev); desc = torch.randn(B, 256, N, generator=g).to(dev); x3d = torch.empty(B, N, 256, device=dev)
wk = packing.pack_keypoint_encoder({k: v for k, v in sd.items()}).to(dev); stats = torch.empty(4 * B + 4, device=dev)
S = hip.stream_handle
runs = {
    "pe_add_transpose": lambda: hip.call("ophip_pe_add_transpose", hip.ptr(feat_c), hip.ptr(pe), hip.ptr(x2d), B, 256, M, S()),
    "transpose_cl": lambda: hip.call("ophip_transpose_cl", hip.ptr(feat_f), hip.ptr(ffcl), B, 128, hf * wf, S()),
    "kpt_stats": lambda: hip.call("ophip_kpt_encode", hip.ptr(kp), kp.stride(0), hip.ptr(desc), desc.stride(0), hip.ptr(wk), hip.ptr(stats), hip.ptr(x3d), B, N, S()),
    "kpt_encode": lambda: hip.call("ophip_kpt_encode", hip.ptr(kp), kp.stride(0), hip.ptr(desc), desc.stride(0), hip.ptr(wk), hip.ptr(stats), hip.ptr(x3d), B, N, S()),
}
for name, fn in runs.items():
    for _ in range(3): fn()
    torch.cuda.synchronize(); hip.timing_select(name)
    for _ in range(20): fn()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    print(f"{name:18s} {ms / max(n, 1) * 1e3:8.1f} us  ({n} launches)")
