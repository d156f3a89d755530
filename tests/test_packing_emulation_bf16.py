"""bf16-path index arithmetic (csrc/tile_bf16.h, csrc/encoder_bf16.hip) on a lane-level emulation of
v_mfma_f32_32x32x16_bf16, in exact arithmetic: swizzled planes, fragment packing, operand-role swap, the
accumulator-as-operand k permutation through kv_reduce -> kv_sum -> attn_apply."""
import numpy as np
import torch

from onepose_st_amd import packing
from tests import mfma_emul as E


def test_gemm_both_orientations_and_epilogue_store():
    g = torch.Generator().manual_seed(0)
    W = torch.randn(64, 48, generator=g).double()
    x = torch.randn(64, 48, generator=g).double().numpy()
    wp = packing.pack_linear_frag16(W.float()).double().numpy()
    # the emulation needs exact values: re-pack from the double matrix with the same permutation
    wp = W.view(2, 32, 3, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1).numpy()
    pl = E.Plane(64, 48 * 2 + 160)            # any pitch that is a multiple of 16 with >= 16 chunks per row
    pl = E.Plane(64, 256)                     # 128 elements per row: 16 chunks (48 used)
    pl.load_rows(np.pad(x, ((0, 0), (0, 80))))
    ref = x @ W.numpy().T                      # [tok][feature]
    # W_IS_A = true: D[feature][token]
    acc = E.gemm_bf16(pl, wp, 0, 2, 2, 3 * 64, 3, 0, True)
    out = np.zeros((64, 64))
    for t in range(2):
        for tt in range(2):
            tmp = np.zeros((32, 32))
            E.acc_to_lds(acc[t][tt], tmp, 0)          # tmp[feature][token]
            out[32 * tt:32 * tt + 32, 32 * t:32 * t + 32] = tmp.T
    np.testing.assert_allclose(out, ref, atol=1e-10)
    # store_featrow_acc -> plane -> read back as activation rows
    dst = E.Plane(64, 128)                     # 64 features: needs >= 16 chunks? 128 B = 8 chunks -> use 256
    dst = E.Plane(64, 256)
    for t in range(2):
        for tt in range(2):
            E.store_featrow_acc(acc[t][tt], dst, 32 * t, 32 * tt)
    back = np.stack([np.concatenate([dst.read16(E.plane_off(row, ch, 256)) for ch in range(8)]) for row in range(64)])
    np.testing.assert_allclose(back, ref, atol=1e-10)
    # W_IS_A = false: D[token][feature]
    acc2 = E.gemm_bf16(pl, wp, 0, 2, 2, 3 * 64, 3, 0, False)
    out2 = np.zeros((64, 64))
    for t in range(2):
        for tt in range(2):
            tmp = np.zeros((32, 32))
            E.acc_to_lds(acc2[t][tt], tmp, 0)         # tmp[token][feature]
            out2[32 * tt:32 * tt + 32, 32 * t:32 * t + 32] = tmp
    np.testing.assert_allclose(out2, ref, atol=1e-10)


def test_linear_attention_chain_through_registers():
    """kv_reduce_bf16 (KV, Ksum from accumulators) -> kv_sum layout -> attn_apply_bf16 (phi(Q) accumulator as B operand)."""
    rng = np.random.default_rng(0)
    Kt, Vt = rng.random((32, 32)), rng.normal(size=(32, 32))       # [tok][d], [tok][v]  (one head, one token tile)
    Qt = rng.random((32, 32))                                      # [tok][d]
    kacc, vacc = Kt[E.ROWS, E.R[:, None]], Vt[E.ROWS, E.R[:, None]]          # D[token][feature] accumulators
    kv, ks = np.zeros((64, 16)), np.zeros((64, 16))
    ones = np.ones((64, 8))
    for st in range(2):
        kv = E.mfma_32x32x16(E.acc_frag(kacc, st), E.acc_frag(vacc, st), kv)
        ks = E.mfma_32x32x16(E.acc_frag(kacc, st), ones, ks)
    KV = Kt.T @ Vt
    np.testing.assert_allclose(kv, KV[E.ROWS, E.R[:, None]], atol=1e-12)     # D[d][v]
    # partial slab [s][lane][8] + Ksum [h][16]; kv_sum emits the same element order as bf16 A fragments
    slab = np.stack([kv[:, 8 * st:8 * st + 8] for st in range(2)])          # [s][lane][8]
    ksum = np.stack([ks[0], ks[32]])                                         # [h][16]  (lanes with r == 0)
    # attn_apply: Q accumulator D[feature d][token]  ->  B operand
    qacc = Qt.T[E.ROWS, E.R[:, None]]
    num, den = np.zeros((64, 16)), np.zeros((64, 16))
    for st in range(2):
        ksfrag = np.stack([ksum[E.H[l], 8 * st:8 * st + 8] for l in range(64)])
        num = E.mfma_32x32x16(slab[st], E.acc_frag(qacc, st), num)
        den = E.mfma_32x32x16(ksfrag, E.acc_frag(qacc, st), den)
    outT, dT = np.zeros((32, 32)), np.zeros((32, 32))
    E.acc_to_lds(num, outT, 0), E.acc_to_lds(den, dT, 0)                     # [v][tok]
    np.testing.assert_allclose(outT.T, Qt @ KV, atol=1e-10)
    np.testing.assert_allclose(dT.T, np.repeat((Qt @ Kt.sum(0))[:, None], 32, 1), atol=1e-10)


def test_coarse_bf16_block_layout(sd):
    blk = packing.pack_coarse_layer_bf16(sd, "loftr_coarse.layers.0.")
    C = 256
    assert blk.dtype == torch.uint8 and blk.numel() == 2 * 2 * 10 * C * C + 16 * C
    hi = blk[:2 * 10 * C * C].view(torch.bfloat16).float()
    lo = blk[2 * 10 * C * C:4 * 10 * C * C].view(torch.bfloat16).float()
    ln = blk[4 * 10 * C * C:].view(torch.float32)
    p = "loftr_coarse.layers.0."
    wq = packing.pack_linear_frag16(sd[p + "q_proj.weight"])
    np.testing.assert_allclose((hi + lo)[:C * C].numpy(), wq.numpy(), rtol=2 ** -15, atol=1e-9)      # hi + lo ~ 16 significant bits
    assert torch.equal(hi[:C * C], wq.to(torch.bfloat16).float())
    w2 = packing.pack_linear_frag16(sd[p + "mlp.2.weight"])
    assert torch.equal(hi[8 * C * C:10 * C * C], w2.to(torch.bfloat16).float())
    assert torch.equal(ln[:C], sd[p + "norm1.weight"]) and torch.equal(ln[3 * C:], sd[p + "norm2.bias"])
    # W0 tile (4c + w), second K half starts 16 k-blocks in: element check through the emulated GEMM
    g = torch.Generator().manual_seed(3)
    x = torch.randn(32, 512, generator=g).double()
    W0 = sd[p + "mlp.0.weight"].double()
    w0p = W0.view(16, 32, 32, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1).numpy()
    px, py = E.Plane(32, 512), E.Plane(32, 512)
    px.load_rows(x[:, :256].numpy()), py.load_rows(x[:, 256:].numpy())
    c, w = 2, 3
    a1 = E.gemm_bf16(px, w0p, 4 * c + w, 1, 1, 32 * 64, 16, 0, True)
    a2 = E.gemm_bf16(py, w0p, 4 * c + w, 1, 1, 32 * 64, 16, 0, True, kb0=16)
    tmp = np.zeros((32, 32))
    E.acc_to_lds(a1[0][0] + a2[0][0], tmp, 0)
    ref = (x @ W0.T).numpy()[:, 128 * c + 32 * w:128 * c + 32 * w + 32]
    np.testing.assert_allclose(tmp.T, ref, atol=1e-9)
