"""Diagnostic: time of the fine stage at c2-like sizes (K matches of a 240 x 320 fine map).  Not part of the product."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from onepose_st_amd import hip, packing
from onepose_st_amd.config import default_config
from onepose_st_amd.synthetic import make_synthetic_state_dict
sd = make_synthetic_state_dict(0, default_config()); dev = torch.device("cuda:0"); hip.load()
B, N, hc, wc, K = 1, 7000, 60, 80, int(os.environ.get("K", "2975"))
hf, wf = 4 * hc, 4 * wc
g = torch.Generator().manual_seed(0)
ff = torch.randn(B, hf * wf, 128, generator=g).to(dev)             # channels-last
desc = torch.randn(B, 128, N, generator=g).to(dev)
cap = 7000
b_ids = torch.zeros(cap, dtype=torch.int64, device=dev)
i_ids = torch.randint(0, N, (cap,), generator=g).to(dev)
j_ids = torch.randint(0, hc * wc, (cap,), generator=g).to(dev)
mkc = torch.zeros(cap, 2, device=dev)
cnt = torch.tensor([K], dtype=torch.int32, device=dev)
expec, mkf = torch.empty(cap, 3, device=dev), torch.empty(cap, 2, device=dev)
head = (hip.ptr(ff), hf * wf * 128, 1, wf * 128, 128, hf, wf, hip.ptr(desc), desc.stride(0), desc.stride(1),
        hip.ptr(b_ids, torch.int64), hip.ptr(i_ids, torch.int64), hip.ptr(j_ids, torch.int64), hip.ptr(cnt, torch.int32), cap, hip.ptr(mkc))
tail = (wc, 4, 4.0, hip.ptr(expec), hip.ptr(mkf), None, None, hip.stream_handle())
w1 = packing.pack_fine_layers_bf16(sd, "loftr_fine.layers.", 2).to(dev)
def v1(): hip.call("ophip_fine_refine_bf16", *head, hip.ptr(w1, None), 2, ctypes.c_uint(2), 1, 3, *tail)
for name, fn in (("fine_refine_bf16 (fine_pair_kernel: two matches per workgroup; OPHIP_FINE_PAIR=0: one)", v1),):
    for _ in range(3): fn()
    torch.cuda.synchronize(); hip.timing_select("fine_refine")
    for _ in range(20): fn()
    torch.cuda.synchronize(); n, ms = hip.timing_read(); hip.timing_select("")
    print(f"{name:42s} K={K}: {ms / n * 1e3:7.1f} us")
