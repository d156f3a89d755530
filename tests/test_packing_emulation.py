"""Host packing + kernel index arithmetic, checked on CPU against a lane-level emulation of
v_mfma_f32_32x32x2_f32 (tests/mfma_emul.py).  These mirror, line for line, what
csrc/encoder.hip / fine.hip do with the packed blocks."""
import numpy as np
import torch

from onepose_st_amd import packing
from tests import mfma_emul as E


def test_pack_linear_gemm_roundtrip():
    g = torch.Generator().manual_seed(0)
    W = torch.randn(96, 40, generator=g)
    x = torch.randn(32, 40, generator=g)
    wp = packing.pack_linear(W).numpy()
    accs = E.gemm_lds_x_packed(x.numpy(), 40 // 8, wp, 0, 3, (40 // 8) * 64)
    out = np.zeros((32, 96))
    for t in range(3):
        E.acc_to_lds(accs[t], out, 32 * t)
    np.testing.assert_allclose(out, (x.double() @ W.double().T).numpy(), rtol=1e-12, atol=1e-12)


def test_coarse_layer_block_offsets_and_kv_row_order(sd):
    blk = packing.pack_coarse_layer(sd, "loftr_coarse.layers.1.").numpy()
    C = 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, C, generator=g)
    p = "loftr_coarse.layers.1."
    wkv = blk[C * C:3 * C * C]
    KB, TS = C // 8, (C // 8) * 64
    K_ref = (x.double() @ sd[p + "k_proj.weight"].double().T).numpy()
    V_ref = (x.double() @ sd[p + "v_proj.weight"].double().T).numpy()
    for wave in (0, 3):
        accs = E.gemm_lds_x_packed(x.numpy(), KB, wkv, 4 * wave, 4, TS)        # kv_reduce_kernel: tiles 4w..4w+3
        Kt, Vt = np.zeros((32, 64)), np.zeros((32, 64))
        E.acc_to_lds(accs[0], Kt, 0), E.acc_to_lds(accs[1], Kt, 32)
        E.acc_to_lds(accs[2], Vt, 0), E.acc_to_lds(accs[3], Vt, 32)
        np.testing.assert_allclose(Kt, K_ref[:, 64 * wave:64 * wave + 64], atol=1e-9)
        np.testing.assert_allclose(Vt, V_ref[:, 64 * wave:64 * wave + 64], atol=1e-9)
    # W0 (K = 512) with the concat split used by attn_apply_kernel: first 256 k from x, next 256 from msg
    w0 = blk[4 * C * C:8 * C * C]
    msg = torch.randn(32, C, generator=g)
    KB2, TS2 = 2 * C // 8, (2 * C // 8) * 64
    wave = 2
    a1 = E.gemm_lds_x_packed(x.numpy(), KB, w0, 4 * wave, 4, TS2, kb0=0)
    a2 = E.gemm_lds_x_packed(msg.numpy(), KB, w0, 4 * wave, 4, TS2, kb0=KB)
    Hh = np.zeros((32, 128))
    for t in range(4):
        E.acc_to_lds(a1[t] + a2[t], Hh, 32 * t)
    ref = (torch.cat([x, msg], 1).double() @ sd[p + "mlp.0.weight"].double().T).numpy()
    np.testing.assert_allclose(Hh, ref[:, 128 * wave:128 * wave + 128], atol=1e-9)
    # LayerNorm tail
    np.testing.assert_array_equal(blk[10 * C * C:10 * C * C + C], sd[p + "norm1.weight"].numpy())
    np.testing.assert_array_equal(blk[10 * C * C + 3 * C:], sd[p + "norm2.bias"].numpy())


def test_kv_accumulator_as_operand():
    """kv_reduce_kernel: KV = phi(K)^T V from the two accumulators, stored in fragment order, then used as
    the B operand of phi(Q) KV in attn_apply_kernel; Ksum via the ones-MFMA."""
    rng = np.random.default_rng(0)
    Kt, Vt, Q = rng.normal(size=(32, 32)), rng.normal(size=(32, 32)), rng.normal(size=(32, 32))
    kacc = Kt[E.ROWS, E.R[:, None]]           # accumulator layout: lane holds [row(reg,h)][col = lane&31]
    vacc = Vt[E.ROWS, E.R[:, None]]
    kv, ks = np.zeros((64, 16)), np.zeros((64, 16))
    for reg in range(16):
        kv = E.mfma_32x32x2(kacc[:, reg], vacc[:, reg], kv)
        ks = E.mfma_32x32x2(kacc[:, reg], np.ones(64), ks)
    KV = Kt.T @ Vt
    np.testing.assert_allclose(kv, KV[E.ROWS, E.R[:, None]], atol=1e-12)
    # slab layout: o[(kb*64 + lane)*4 + j] = kv[lane][4kb + j];  o[1024 + acc_row(reg,h)] = ks (lanes with r == 0)
    slab = np.zeros(1056)
    for kb in range(4):
        for l in range(64):
            slab[(kb * 64 + l) * 4:(kb * 64 + l) * 4 + 4] = kv[l, 4 * kb:4 * kb + 4]
    for l in (0, 32):
        for reg in range(16):
            slab[1024 + E.acc_row(reg, l >> 5)] = ks[l, reg]
    np.testing.assert_allclose(slab[1024:], Kt.sum(0), atol=1e-12)
    # consumer
    num, den = np.zeros((64, 16)), np.zeros((64, 16))
    for kb in range(4):
        aq = E.lds_a_frag(Q, kb)
        bk = slab[:1024].reshape(-1, 4)[kb * 64:(kb + 1) * 64]
        bs = np.stack([slab[1024 + 8 * kb + 4 * E.H[l]:1024 + 8 * kb + 4 * E.H[l] + 4] for l in range(64)])
        num, den = E.mfma4(aq, bk, num), E.mfma4(aq, bs, den)
    out, dn = np.zeros((32, 32)), np.zeros((32, 32))
    E.acc_to_lds(num, out, 0), E.acc_to_lds(den, dn, 0)
    np.testing.assert_allclose(out, Q @ KV, atol=1e-10)
    np.testing.assert_allclose(dn, np.repeat((Q @ Kt.sum(0))[:, None], 32, 1), atol=1e-10)


def test_fine_block_diagonal_two_heads():
    """fine_refine_kernel: D = 16, a 32-wide tile holds two heads; register KV masked to its block diagonal
    and consumed with frag_of()."""
    rng = np.random.default_rng(1)
    Kt, Vt, Q = rng.random((32, 32)), rng.normal(size=(32, 32)), rng.random((32, 32))
    rowmask = (np.arange(32) < 25).astype(float)[:, None]
    kacc = (Kt * rowmask)[E.ROWS, E.R[:, None]]
    vacc = Vt[E.ROWS, E.R[:, None]]
    kv, ks = np.zeros((64, 16)), np.zeros((64, 16))
    for reg in range(16):
        kv = E.mfma_32x32x2(kacc[:, reg], vacc[:, reg], kv)
        ks = E.mfma_32x32x2(kacc[:, reg], np.ones(64), ks)
    same = (E.ROWS >> 4) == (E.R[:, None] >> 4)
    kv, ks = kv * same, ks * same
    num, den = np.zeros((64, 16)), np.zeros((64, 16))
    for kb in range(4):
        aq = E.lds_a_frag(Q, kb)
        num, den = E.mfma4(aq, E.frag_of(kv, kb), num), E.mfma4(aq, E.frag_of(ks, kb), den)
    out, dn = np.zeros((32, 32)), np.zeros((32, 32))
    E.acc_to_lds(num, out, 0), E.acc_to_lds(den, dn, 0)
    for hd in range(2):
        s = slice(16 * hd, 16 * hd + 16)
        Kh, Vh, Qh = (Kt * rowmask)[:, s], Vt[:, s], Q[:, s]
        np.testing.assert_allclose(out[:, s], Qh @ (Kh.T @ Vh), atol=1e-10)
        np.testing.assert_allclose(dn[:, s], np.repeat((Qh @ Kh.sum(0))[:, None], 16, 1), atol=1e-10)


def test_fine_and_kpt_blocks(sd):
    blk = packing.pack_fine_layer(sd, "loftr_fine.layers.0.").numpy()
    C = 128
    g = torch.Generator().manual_seed(2)
    x = torch.randn(32, C, generator=g)
    p = "loftr_fine.layers.0."
    KB, TS = C // 8, (C // 8) * 64
    for wave in (1, 3):
        accs = E.gemm_lds_x_packed(x.numpy(), KB, blk[:3 * C * C], 3 * wave, 3, TS)     # Q | K | V tile of the wave
        for t, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            o = np.zeros((32, 32))
            E.acc_to_lds(accs[t], o, 0)
            ref = (x.double() @ sd[p + nm + ".weight"].double().T).numpy()[:, 32 * wave:32 * wave + 32]
            np.testing.assert_allclose(o, ref, atol=1e-9)
    kp = packing.pack_keypoint_encoder(sd).numpy()
    assert kp.size == 32 * 8 + 64 * 32 + 128 * 64 + 256 * 128 + 32 + 64 + 128 + 256
    pts = np.zeros((32, 8)); pts[:, :3] = np.random.default_rng(3).normal(size=(32, 3))
    acc = E.gemm_lds_x_packed(pts, 1, kp[:256], 0, 1, 64)[0]
    o = np.zeros((32, 32)); E.acc_to_lds(acc, o, 0)
    np.testing.assert_allclose(o, pts[:, :3] @ sd["kpt_3d_pos_encoding.encoder.0.weight"].double().numpy().T, atol=1e-9)
    np.testing.assert_array_equal(kp[-256:], sd["kpt_3d_pos_encoding.encoder.9.bias"].numpy())
