"""Host-side packing of the reference ``state_dict`` into the HIP kernels' weight blocks.

Fragment order (csrc/tile.h): a ``Linear`` weight ``W[out][in]`` (torch layout,
``y = x W^T``) becomes ``[out/32][in/8][lane = 32*h + out%32][j] = W[out][8*kb + 4*h + j]``
so that one wave's load of a 32x8 weight fragment is 1 KiB contiguous and each lane
holds the B operand of four consecutive ``v_mfma_f32_32x32x2_f32``.

Blocks (all float32, concatenated flat):

* coarse layer (d=256), ``ophip_encoder_layer``:
  ``Wq | Wkv | Wm | W0 | W2 | norm1.w norm1.b norm2.w norm2.b`` where ``Wkv`` stacks, per
  wave w = 0..3, ``Wk[64w:64w+64]`` then ``Wv[64w:64w+64]`` (a wave owns heads 2w, 2w+1).
* fine layer (d=128), ``ophip_fine_refine``: ``Wqkv | Wm | W0 | W2 | ln`` where ``Wqkv``
  stacks per wave ``Wq[32w:32w+32] | Wk[32w:32w+32] | Wv[32w:32w+32]``.
* keypoint encoder, ``ophip_kpt_encode``: ``W1 (K padded 3->8) | W2 | W3 | W4 | b1 b2 b3 b4``.

State-dict keys follow the reference (SURVEY.md section 8b;
``loftr_module/transformer.py:29-52``, ``utils/position_encoding.py:62-79``).
"""
from __future__ import annotations

import torch


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """``[out, in]`` -> flat fragment order; ``out % 32 == 0`` and ``in % 8 == 0``."""
    out_f, in_f = w.shape
    if out_f % 32 or in_f % 8:
        raise ValueError(f"pack_linear: shape {tuple(w.shape)} is not a multiple of (32, 8)")
    w = w.detach().to(torch.float32).cpu().contiguous()
    return w.view(out_f // 32, 32, in_f // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def _ln(sd, p):
    return [sd[p + k].detach().float().cpu().reshape(-1) for k in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]


def pack_coarse_layer(sd: dict, prefix: str) -> torch.Tensor:
    """``prefix`` like ``"loftr_coarse.layers.0."``; d_model 256, 8 heads."""
    wk, wv = sd[prefix + "k_proj.weight"], sd[prefix + "v_proj.weight"]
    if tuple(wk.shape) != (256, 256):
        raise ValueError("coarse encoder kernels are specialised for d_model = 256")
    wkv = torch.cat([torch.cat([wk[64 * w:64 * w + 64], wv[64 * w:64 * w + 64]], 0) for w in range(4)], 0)
    parts = [pack_linear(sd[prefix + "q_proj.weight"]), pack_linear(wkv), pack_linear(sd[prefix + "merge.weight"]),
             pack_linear(sd[prefix + "mlp.0.weight"]), pack_linear(sd[prefix + "mlp.2.weight"])] + _ln(sd, prefix)
    out = torch.cat(parts)
    assert out.numel() == 10 * 256 * 256 + 4 * 256
    return out


def pack_fine_layer(sd: dict, prefix: str) -> torch.Tensor:
    """``prefix`` like ``"loftr_fine.layers.0."``; d_model 128, 8 heads."""
    wq, wk, wv = (sd[prefix + n + ".weight"] for n in ("q_proj", "k_proj", "v_proj"))
    if tuple(wq.shape) != (128, 128):
        raise ValueError("fine encoder kernel is specialised for d_model = 128")
    wqkv = torch.cat([torch.cat([m[32 * w:32 * w + 32] for m in (wq, wk, wv)], 0) for w in range(4)], 0)
    parts = [pack_linear(wqkv), pack_linear(sd[prefix + "merge.weight"]), pack_linear(sd[prefix + "mlp.0.weight"]),
             pack_linear(sd[prefix + "mlp.2.weight"])] + _ln(sd, prefix)
    out = torch.cat(parts)
    assert out.numel() == 10 * 128 * 128 + 4 * 128
    return out


def pack_keypoint_encoder(sd: dict, prefix: str = "kpt_3d_pos_encoding.encoder.") -> torch.Tensor:
    ids = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    ws = [sd[f"{prefix}{i}.weight"].detach().float().cpu() for i in ids]
    bs = [sd[f"{prefix}{i}.bias"].detach().float().cpu().reshape(-1) for i in ids]
    if [tuple(w.shape) for w in ws] != [(32, 3), (64, 32), (128, 64), (256, 128)]:
        raise NotImplementedError("keypoint encoder kernel is specialised for 3->32->64->128->256")
    w1 = torch.zeros(32, 8)
    w1[:, :3] = ws[0]
    return torch.cat([pack_linear(w1), pack_linear(ws[1]), pack_linear(ws[2]), pack_linear(ws[3])] + bs)


# ----------------------------------------------------------------------------------------------
# bf16 / split-bf16 blocks (csrc/tile_bf16.h): fragments of 8 bf16 per lane,
#   [out/32][in/16][lane = 32*h + out%32][j] = W[out][16*kb + 8*h + j],  planes hi = bf16(W), lo = bf16(W - hi)
# ----------------------------------------------------------------------------------------------

def pack_linear_frag16(w: torch.Tensor) -> torch.Tensor:
    """``[out, in]`` f32 -> f32 tensor in 16-k fragment order (``out % 32 == 0``, ``in % 16 == 0``)."""
    out_f, in_f = w.shape
    if out_f % 32 or in_f % 16:
        raise ValueError(f"pack_linear_frag16: shape {tuple(w.shape)} is not a multiple of (32, 16)")
    w = w.detach().to(torch.float32).cpu().contiguous()
    return w.view(out_f // 32, 32, in_f // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def split_planes(flat: torch.Tensor):
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.float()).to(torch.bfloat16)
    return hi, lo


def pack_linear_x3(w: torch.Tensor) -> torch.Tensor:
    """uint8 block ``[hi fragments][lo fragments]`` of one ``Linear`` weight ``[out, in]`` for ``ophip_rows_linear_x3``
    (16-k fragment order of :func:`pack_linear_frag16`, split-bf16 planes)."""
    hi, lo = split_planes(pack_linear_frag16(w))
    return _bytes(hi, lo)


def _bytes(*tensors) -> torch.Tensor:
    return torch.cat([t.contiguous().view(torch.uint8).reshape(-1) for t in tensors])


def pack_coarse_layer_bf16(sd: dict, prefix: str) -> torch.Tensor:
    """uint8 block ``[hi: Wq | Wkv | Wm | W0 | W2][lo: same][norm1.w norm1.b norm2.w norm2.b f32]`` for
    ``ophip_encoder_layer_bf16`` (Wkv in the per-wave row order of :func:`pack_coarse_layer`)."""
    wk, wv = sd[prefix + "k_proj.weight"], sd[prefix + "v_proj.weight"]
    if tuple(wk.shape) != (256, 256):
        raise ValueError("coarse encoder kernels are specialised for d_model = 256")
    wkv = torch.cat([torch.cat([wk[64 * w:64 * w + 64], wv[64 * w:64 * w + 64]], 0) for w in range(4)], 0)
    flat = torch.cat([pack_linear_frag16(m) for m in (sd[prefix + "q_proj.weight"], wkv, sd[prefix + "merge.weight"],
                                                     sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.2.weight"])])
    hi, lo = split_planes(flat)
    out = _bytes(hi, lo, torch.cat(_ln(sd, prefix)))
    assert out.numel() == 2 * 2 * 10 * 256 * 256 + 16 * 256
    return out


# ----------------------------------------------------------------------------------------------
# split-bf16 coarse layer as per-wave STREAMS (csrc/encoder_x3w8.hip): v_mfma_f32_16x16x32_bf16 fragments
#   frag(W, row0, k0)[lane = 16 q + c16][j] = W[row0 + c16][k0 + 8 q + j]      (1 KiB per plane)
# in exactly the order wave fw consumes them, hi then lo per fragment.
# ----------------------------------------------------------------------------------------------
def x3_frag(w: torch.Tensor, row0: int, k0: int) -> torch.Tensor:
    """``[64 lanes][8]`` A/B-operand fragment of ``W[row0:row0+16, k0:k0+32]``."""
    return w[row0:row0 + 16, k0:k0 + 32].reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8)


def pack_fine_layers_bf16(sd: dict, prefix: str, n_layers: int) -> torch.Tensor:
    """uint8 block for ``ophip_fine_refine_bf16``: per layer ``[hi: Wq | Wkv | Wm | W0 | W2][lo: same][ln f32]`` where
    ``Wkv`` stacks per wave w = 0..3 ``Wk[32w:32w+32]`` then ``Wv[32w:32w+32]`` (``prefix`` like ``"loftr_fine.layers."``)."""
    blocks = []
    for i in range(n_layers):
        p = f"{prefix}{i}."
        wq, wk, wv = (sd[p + n + ".weight"] for n in ("q_proj", "k_proj", "v_proj"))
        if tuple(wq.shape) != (128, 128):
            raise ValueError("fine encoder kernel is specialised for d_model = 128")
        wkv = torch.cat([torch.cat([wk[32 * w:32 * w + 32], wv[32 * w:32 * w + 32]], 0) for w in range(4)], 0)
        flat = torch.cat([pack_linear_frag16(m) for m in (wq, wkv, sd[p + "merge.weight"], sd[p + "mlp.0.weight"], sd[p + "mlp.2.weight"])])
        hi, lo = split_planes(flat)
        blocks.append(_bytes(hi, lo, torch.cat(_ln(sd, p))))
    out = torch.cat(blocks)
    assert out.numel() == n_layers * (2 * 2 * 10 * 128 * 128 + 16 * 128)
    return out


def x3w8_program(fw: int):
    """(main, kv) entries for wave ``fw`` (= head fw) of the 8-wave kernel csrc/encoder_x3w8.hip: each wave owns 32 output features
    (two tiles) of every stage; the MLP's hidden layer goes in two 256-wide chunks: Q | merge | W0c0 | W2c0 | W0c1 | W2c1."""
    def gemm(mat, rows, k0, ksteps):
        return [(mat, r0, k0 + 32 * ks) for ks in range(ksteps) for r0 in rows]
    rows = [32 * fw + 16 * ft for ft in range(2)]
    main = gemm("q", rows, 0, 8) + gemm("m", rows, 0, 8)
    for c in range(2):
        main += gemm("w0", [256 * c + r0 for r0 in rows], 0, 16) + gemm("w2", rows, 256 * c, 8)
    kv = [(("k" if ft < 2 else "v"), 32 * fw + 16 * (ft & 1), 32 * ks) for ks in range(8) for ft in range(4)]
    assert 2 * len(main) == 256 and 2 * len(kv) == 64
    return main, kv


def pack_coarse_layer_x3w8(sd: dict, prefix: str) -> torch.Tensor:
    """uint8 block ``[main streams: 8 waves x 256 KiB][K|V streams: 8 x 64 KiB][norm1.w norm1.b norm2.w norm2.b f32]`` for
    ``ophip_encoder_layer_x3w8``."""
    mats = {"q": sd[prefix + "q_proj.weight"], "k": sd[prefix + "k_proj.weight"], "v": sd[prefix + "v_proj.weight"],
            "m": sd[prefix + "merge.weight"], "w0": sd[prefix + "mlp.0.weight"], "w2": sd[prefix + "mlp.2.weight"]}
    mats = {k: v.detach().to(torch.float32).cpu().contiguous() for k, v in mats.items()}
    if tuple(mats["q"].shape) != (256, 256) or tuple(mats["w0"].shape) != (512, 512) or tuple(mats["w2"].shape) != (256, 512):
        raise ValueError("coarse encoder kernels are specialised for d_model = 256")
    progs = [x3w8_program(fw) for fw in range(8)]
    frags = [x3_frag(mats[m], r0, k0) for sel in (0, 1) for fw in range(8) for (m, r0, k0) in progs[fw][sel]]
    flat = torch.stack(frags)
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.float()).to(torch.bfloat16)
    out = _bytes(torch.stack([hi, lo], 1), torch.cat(_ln(sd, prefix)))
    assert out.numel() == 8 * (256 + 64) * 1024 + 16 * 256
    return out


# ---------------------------------------------------------------------------------------------------
# backbone convolutions (csrc/conv.hip): BatchNorm folded in, weights as A-operand fragments in the order the kernel
# walks K: (32-channel chunk, tap, 16-channel k-block)
# ---------------------------------------------------------------------------------------------------
def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def fold_bn(w: torch.Tensor, sd: dict, bn_prefix: str | None, eps: float = 1e-5):
    """Eval-mode ``BatchNorm2d`` after a bias-free convolution: ``w * s``, ``beta - mean * s`` with ``s = gamma / sqrt(var + eps)``."""
    w = w.detach().to(torch.float32).cpu()
    if bn_prefix is None:
        return w, torch.zeros(w.shape[0])
    g, b = sd[bn_prefix + "weight"].float().cpu(), sd[bn_prefix + "bias"].float().cpu()
    m, v = sd[bn_prefix + "running_mean"].float().cpu(), sd[bn_prefix + "running_var"].float().cpu()
    s = g / torch.sqrt(v + eps)
    return w * s.view(-1, 1, 1, 1), b - m * s


def pack_conv_bf16(w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``[cout, cin, k, k]`` f32 (+ bias) -> bytes: hi fragments | lo fragments | bias f32 (``ophip_conv_wpack_bytes``)."""
    cout, cin, k, _ = w.shape
    cop, cip, T = pad32(cout), pad32(cin), k * k
    wp = torch.zeros(cop, cip, T)
    wp[:cout, :cin] = w.reshape(cout, cin, T)
    wm = wp.view(cop, cip // 32, 2, 16, T).permute(0, 1, 4, 2, 3).reshape(cop, cip * T)      # K order: chunk, tap, k-block, 16
    hi, lo = split_planes(pack_linear_frag16(wm))
    bp = torch.zeros(cop)
    bp[:cout] = bias
    return _bytes(hi, lo, bp)


def pack_stem(w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``[128, 1, 7, 7]`` folded stem weights -> f32 ``[49][128]`` followed by the bias."""
    return torch.cat([w.reshape(w.shape[0], 49).t().contiguous().reshape(-1), bias.reshape(-1)]).to(torch.float32)
