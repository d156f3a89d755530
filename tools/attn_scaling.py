"""Diagnostic: attn_apply launch time against the token count (workgroup quantisation on 256 CUs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0"); hip.load()
w = packing.pack_coarse_layer_bf16(sd, "loftr_coarse.layers.0.").to(dev)
for tot in (4096, 8192, 10240, 11800, 12288, 16384, 24576, 32768):
    L3 = tot * 7000 // 11800 // 32 * 32; L2 = tot - L3
    g = torch.Generator().manual_seed(0)
    x3, x2 = torch.randn(1, L3, 256, generator=g).to(dev), torch.randn(1, L2, 256, generator=g).to(dev)
    y3, y2 = torch.empty_like(x3), torch.empty_like(x2)
    ws = torch.empty(hip.load().ophip_encoder_bf16_workspace_bytes(1, L3, L2), dtype=torch.uint8, device=dev)
    def run():
        hip.call("ophip_encoder_layer_bf16", hip.ptr(x3), hip.ptr(x2), hip.ptr(y3), hip.ptr(y2), 1, L3, L2, hip.ptr(w, None), hip.ptr(w, None), 3, 0, 0, 0, hip.ptr(ws, None), hip.stream_handle())
    for _ in range(3): run()
    torch.cuda.synchronize(); hip.timing_select("attn_apply")
    for _ in range(20): run()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    print(f"tokens {tot:6d} ({(L3 + 31) // 32 + (L2 + 31) // 32:4d} workgroups): attn_apply {ms / n * 1e3:7.1f} us  -> {ms / n * 1e6 / tot:6.2f} ns/token")
