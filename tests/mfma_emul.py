"""numpy emulation of the gfx950 lane maps the kernels rely on (csrc/tile.h), used by CPU tests to
check the host packing and the kernels' index arithmetic without a GPU.

v_mfma_f32_32x32x2_f32 (MI355X guide, "gfx950 intrinsic list" / "Fragment layout"):
  A[i = lane & 31][k = lane >> 5],  B[k = lane >> 5][j = lane & 31],
  D[row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)][col = lane & 31].
"""
import numpy as np

LANES = np.arange(64)
R, H = LANES & 31, LANES >> 5


def acc_row(reg, h):
    return (reg & 3) + 8 * (reg >> 2) + 4 * h


ROWS = np.array([[acc_row(reg, h) for reg in range(16)] for h in H])      # [64][16]


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64][16]; returns the new accumulator."""
    A = np.stack([a[:32], a[32:]], axis=1).astype(np.float64)           # [i][k]
    Bm = np.stack([b[:32], b[32:]], axis=0).astype(np.float64)          # [k][j]
    D = A @ Bm
    return acc + D[ROWS, R[:, None]]


def mfma4(a4, b4, acc):
    for j in range(4):
        acc = mfma_32x32x2(a4[:, j], b4[:, j], acc)
    return acc


def lds_a_frag(img, kb):
    """the ds_read_b128 of gemm_lds_x_packed: lane reads img[lane & 31][8 kb + 4 h : +4]"""
    return np.stack([img[R[l], 8 * kb + 4 * H[l]: 8 * kb + 4 * H[l] + 4] for l in range(64)])


def gemm_lds_x_packed(img, kblocks, wflat, tile0, ntiles, tstride_frags, kb0=0):
    """returns [ntiles] accumulators [64][16]; wflat is the flat packed weight (floats)."""
    w = wflat.reshape(-1, 4)
    accs = [np.zeros((64, 16)) for _ in range(ntiles)]
    for kb in range(kblocks):
        a = lds_a_frag(img, kb)
        for t in range(ntiles):
            base = (tile0 + t) * tstride_frags + (kb0 + kb) * 64
            accs[t] = mfma4(a, w[base:base + 64], accs[t])
    return accs


def acc_to_lds(acc, img, col0):
    for l in range(64):
        for reg in range(16):
            img[ROWS[l, reg], col0 + R[l]] = acc[l, reg]


def frag_of(acc, kb):
    return acc[:, 4 * kb:4 * kb + 4]
