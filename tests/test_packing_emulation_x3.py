"""Index arithmetic of csrc/encoder_x3w8.hip on a lane-level emulation of v_mfma_f32_16x16x32_bf16, in exact arithmetic:
the per-wave weight streams of packing.x3w8_program (eight waves, wave fw = head fw), both operand orientations, the swizzled planes and quad stores, the
MLP chunk order, and the accumulator-as-operand chain  K|V tail -> slab -> kv_sum fragments -> attention."""
import numpy as np
import torch

from onepose_st_amd import packing
from tests import mfma_emul as E

L64 = np.arange(64)
C16, Q = L64 & 15, L64 >> 4
NTT = 3


def mfma16(a8, b8, acc):
    """a8, b8 [64][8] per-lane fragments, acc [64][4]:  A[row c16][k 8q+j], B[k 8q+j][col c16], D[row 4q+r][col c16]"""
    A, B = np.zeros((16, 32)), np.zeros((32, 16))
    for l in range(64):
        A[C16[l], 8 * Q[l]:8 * Q[l] + 8] = a8[l]
        B[8 * Q[l]:8 * Q[l] + 8, C16[l]] = b8[l]
    D = A @ B
    return acc + np.stack([[D[4 * Q[l] + r, C16[l]] for r in range(4)] for l in range(64)])


def frag(w, row0, k0):
    return packing.x3_frag(torch.from_numpy(w), row0, k0).numpy()


def read_x(plane, tt, chunk):
    return np.stack([plane.read16(E.plane_off(16 * tt + C16[l], chunk + Q[l], plane.rowb)) for l in range(64)])


def gemm_stage(prog, mats, plane, nf, nks, w_is_a, chunk0=0):
    """mirror of gemm_stage<NF, NKS, W_IS_A>: prog = this stage's slice of the stream program"""
    acc = [[np.zeros((64, 4)) for _ in range(NTT)] for _ in range(nf)]
    assert len(prog) == nf * nks
    for ks in range(nks):
        for ft in range(nf):
            m, r0, k0 = prog[ks * nf + ft]
            w = frag(mats[m], r0, k0)
            for tt in range(NTT):
                x = read_x(plane, tt, chunk0 + 4 * ks)
                acc[ft][tt] = mfma16(w, x, acc[ft][tt]) if w_is_a else mfma16(x, w, acc[ft][tt])
    return acc


def store_quad(plane, v4, tt, l, f0):
    row = 16 * tt + C16[l]
    plane.write_elems(E.plane_off(row, f0 >> 3, plane.rowb) + 2 * (f0 & 7), v4)


def mats_random(seed=0):
    g = np.random.default_rng(seed)
    return {"q": g.normal(size=(256, 256)), "k": g.normal(size=(256, 256)), "v": g.normal(size=(256, 256)),
            "m": g.normal(size=(256, 256)), "w0": g.normal(size=(512, 512)), "w2": g.normal(size=(256, 512))}


def test_stream_gemm_weights_as_a_and_quad_store_roundtrip():
    mats = mats_random()
    x = np.random.default_rng(1).normal(size=(48, 256))
    px = E.Plane(48, 512)
    px.load_rows(x)
    out = np.zeros((48, 256))
    dst = E.Plane(48, 512)
    for fw in range(8):
        main, _ = packing.x3w8_program(fw)
        acc = gemm_stage(main[:16], mats, px, 2, 8, True)                 # Q stage = first 16 entries (32 fragments hi + lo)
        for ft in range(2):
            for tt in range(NTT):
                for l in range(64):
                    f0 = 32 * fw + 16 * ft + 4 * Q[l]
                    out[16 * tt + C16[l], f0:f0 + 4] = acc[ft][tt][l]
                    store_quad(dst, acc[ft][tt][l], tt, l, f0)
    ref = x @ mats["q"].T
    np.testing.assert_allclose(out, ref, atol=1e-9)
    back = np.stack([np.concatenate([dst.read16(E.plane_off(row, ch, 512)) for ch in range(32)]) for row in range(48)])
    np.testing.assert_allclose(back, ref, atol=1e-9)
    # merge stage = entries 16..31 of the same stream
    acc = gemm_stage(packing.x3w8_program(5)[0][16:32], mats, px, 2, 8, True)
    np.testing.assert_allclose(acc[1][2][:, 0], (x @ mats["m"].T)[32 + C16, 32 * 5 + 16 + 4 * Q], atol=1e-9)


def test_mlp_chunk_order_of_the_stream():
    """W0c0 | W2c0 | W0c1 | W2c1: 256-wide hidden chunks through ONE hidden buffer, x-half then msg-half k-steps"""
    mats = mats_random(2)
    g = np.random.default_rng(3)
    x, msg = g.normal(size=(48, 256)), g.normal(size=(48, 256))
    px, py = E.Plane(48, 512), E.Plane(48, 512)
    px.load_rows(x), py.load_rows(msg)
    hidden_ref = np.maximum(np.concatenate([x, msg], 1) @ mats["w0"].T, 0.0)
    out_ref = hidden_ref @ mats["w2"].T
    streams = [packing.x3w8_program(fw)[0][32:] for fw in range(8)]
    o = [[[np.zeros((64, 4)) for _ in range(NTT)] for _ in range(2)] for _ in range(8)]
    for c in range(2):
        hbuf = E.Plane(48, 512)
        pos = 48 * c
        for fw in range(8):
            prog = streams[fw][pos:pos + 32]
            a = gemm_stage(prog[:16], mats, px, 2, 8, True)
            b = gemm_stage(prog[16:], mats, py, 2, 8, True)
            for ft in range(2):
                for tt in range(NTT):
                    for l in range(64):
                        store_quad(hbuf, np.maximum(a[ft][tt][l] + b[ft][tt][l], 0), tt, l, 32 * fw + 16 * ft + 4 * Q[l])
        for fw in range(8):
            a = gemm_stage(streams[fw][pos + 32:pos + 48], mats, hbuf, 2, 8, True)
            for ft in range(2):
                for tt in range(NTT):
                    o[fw][ft][tt] += a[ft][tt]
    assert all(len(st) == 96 for st in streams)
    out = np.zeros((48, 256))
    for fw in range(8):
        for ft in range(2):
            for tt in range(NTT):
                for l in range(64):
                    f0 = 32 * fw + 16 * ft + 4 * Q[l]
                    out[16 * tt + C16[l], f0:f0 + 4] = o[fw][ft][tt][l]
    np.testing.assert_allclose(out, out_ref, atol=1e-7)


def test_kv_tail_slab_kv_sum_attention_chain():
    mats = mats_random(4)
    g = np.random.default_rng(5)
    y = g.normal(size=(48, 256))                       # source tokens (one tile)
    xq = g.normal(size=(48, 256))                      # query-side tile
    py, pq = E.Plane(48, 512), E.Plane(48, 512)
    py.load_rows(y), pq.load_rows(xq)
    phi = lambda t: np.where(t > 0, t + 1.0, np.exp(t))
    Kf, Vf = phi(y @ mats["k"].T), y @ mats["v"].T     # [tok][256]
    blk = np.full((8, 2, 64, 8), np.nan)               # kv_sum fragments [head][vt][lane][j]  (hi plane only: exact)
    ksum = np.zeros((8, 32))
    for fw in range(8):                                # wave fw = head fw: ft 0, 1 = K tiles, ft 2, 3 = V tiles
        _, kvp = packing.x3w8_program(fw)
        kk = gemm_stage(kvp, mats, py, 4, 8, False)    # D[token 4q + r][feature c16]
        for ft in range(2):
            for tt in range(NTT):
                kk[ft][tt] = phi(kk[ft][tt])
        np.testing.assert_allclose(kk[3][1][:, 2], Vf[16 + 4 * Q + 2, 32 * fw + 16 + C16], atol=1e-9)
        z = np.zeros((64, 4))
        head = fw
        for dt in range(2):
            for vt in range(2):
                a0 = np.concatenate([kk[dt][0], kk[dt][1]], 1)
                a1 = np.concatenate([kk[dt][2], z], 1)
                b0 = np.concatenate([kk[2 + vt][0], kk[2 + vt][1]], 1)
                b1 = np.concatenate([kk[2 + vt][2], z], 1)
                kvt = mfma16(a1, b1, mfma16(a0, b0, np.zeros((64, 4))))
                # slab [head][dt][vt][lane][r]  ->  kv_sum: fragment element j = 4 dt + r of [head][vt][lane]
                blk[head, vt, :, 4 * dt:4 * dt + 4] = kvt
            s = sum(kk[dt][tt].sum(1) for tt in range(NTT))                   # per lane: its 12 tokens
            tot = np.array([s[(L64 & 15) == c].sum() for c in range(16)])     # + the permlane sums over q
            ksum[head, 16 * dt:16 * dt + 16] = tot
    KV = np.stack([Kf[:, 32 * h:32 * h + 32].T @ Vf[:, 32 * h:32 * h + 32] for h in range(8)])
    np.testing.assert_allclose(ksum, Kf.sum(0).reshape(8, 32), atol=1e-9)
    # attention on the query tile: phi(Q) accumulators (D[feature][token]) as the B operand
    Qf = phi(xq @ mats["q"].T)
    msg = np.zeros((48, 256))
    for fw in range(8):
        main, _ = packing.x3w8_program(fw)
        qa = gemm_stage(main[:16], mats, pq, 2, 8, True)
        head = fw
        for tt in range(NTT):
            p0, p1 = phi(qa[0][tt]), phi(qa[1][tt])
            den_l = np.array([p0[l] @ ksum[head, 4 * Q[l]:4 * Q[l] + 4] + p1[l] @ ksum[head, 16 + 4 * Q[l]:20 + 4 * Q[l]] for l in range(64)])
            den = np.array([den_l[(L64 & 15) == C16[l]].sum() for l in range(64)])
            qfrag = np.concatenate([p0, p1], 1)
            for vt in range(2):
                num = mfma16(blk[head, vt], qfrag, np.zeros((64, 4)))
                for l in range(64):
                    f0 = 32 * fw + 16 * vt + 4 * Q[l]
                    msg[16 * tt + C16[l], f0:f0 + 4] = num[l] / den[l]
    ref = np.concatenate([(Qf[:, 32 * h:32 * h + 32] @ KV[h]) / (Qf[:, 32 * h:32 * h + 32] @ Kf.sum(0)[32 * h:32 * h + 32])[:, None] for h in range(8)], 1)
    np.testing.assert_allclose(msg, ref, rtol=1e-9, atol=1e-9)


def test_x3w8_block_layout(sd):
    p = "loftr_coarse.layers.1."
    blk = packing.pack_coarse_layer_x3w8(sd, p)
    assert blk.dtype == torch.uint8 and blk.numel() == 8 * 320 * 1024 + 16 * 256
    fr = blk[:8 * 320 * 1024].view(torch.bfloat16).float().view(-1, 2, 64, 8)          # [entry][plane][lane][j]
    main, kv = packing.x3w8_program(3)
    e = 3 * 128 + 70                                   # wave 3, entry 70 of its main stream (128 entries = 256 fragments per wave)
    m, r0, k0 = main[70]
    w = {"q": "q_proj", "m": "merge", "w0": "mlp.0", "w2": "mlp.2"}[m]
    want = packing.x3_frag(sd[p + w + ".weight"].float(), r0, k0)
    assert torch.equal(fr[e, 0], want.to(torch.bfloat16).float())
    np.testing.assert_allclose((fr[e, 0] + fr[e, 1]).numpy(), want.numpy(), rtol=2 ** -15, atol=1e-9)
    e = 8 * 128 + 1 * 32 + 9                           # K|V streams follow the eight main streams (32 entries per wave)
    m, r0, k0 = packing.x3w8_program(1)[1][9]
    want = packing.x3_frag(sd[p + ("k_proj" if m == "k" else "v_proj") + ".weight"].float(), r0, k0)
    assert torch.equal(fr[e, 0], want.to(torch.bfloat16).float())
    ln = blk[8 * 320 * 1024:].view(torch.float32)
    assert torch.equal(ln[:256], sd[p + "norm1.weight"]) and torch.equal(ln[768:], sd[p + "norm2.bias"])
