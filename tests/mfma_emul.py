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


# ------------------------------------------------------------------------------------------------
# v_mfma_f32_32x32x16_bf16:  A[i = lane & 31][k = 8 (lane >> 5) + j],  B[k = 8 (lane >> 5) + j][col = lane & 31]
# (values kept in float64 here: these emulations check index arithmetic, not rounding)
# ------------------------------------------------------------------------------------------------

def mfma_32x32x16(a8, b8, acc):
    """a8, b8: [64][8] per-lane fragments; acc [64][16]"""
    A = np.zeros((32, 16))
    Bm = np.zeros((16, 32))
    for l in range(64):
        A[R[l], 8 * H[l]:8 * H[l] + 8] = a8[l]
        Bm[8 * H[l]:8 * H[l] + 8, R[l]] = b8[l]
    D = A @ Bm
    return acc + D[ROWS, R[:, None]]


def plane_off(row, chunk, rowb):
    return row * rowb + ((chunk ^ (row & 15)) << 4)


class Plane:
    """a swizzled bf16 LDS plane addressed in bytes (2 bytes per element)"""

    def __init__(self, rows, rowb):
        self.rowb = rowb
        self.mem = np.full(rows * rowb // 2, np.nan)

    def write_elems(self, byte_off, vals):
        assert byte_off % 2 == 0
        self.mem[byte_off // 2:byte_off // 2 + len(vals)] = vals

    def read16(self, byte_off):
        assert byte_off % 16 == 0
        return self.mem[byte_off // 2:byte_off // 2 + 8].copy()

    def load_rows(self, x):
        """load_rows_to_planes"""
        T, Cc = x.shape
        for row in range(T):
            for ch in range(Cc // 8):
                self.write_elems(plane_off(row, ch, self.rowb), x[row, 8 * ch:8 * ch + 8])


def gemm_bf16(plane, wflat16, tile0, nt, ttn, tstride, kblocks, chunk0, w_is_a, kb0=0):
    """tile_bf16.h gemm_bf16 with NS = 1 semantics on exact values; returns acc[t][tt]"""
    w = wflat16.reshape(-1, 8)
    acc = [[np.zeros((64, 16)) for _ in range(ttn)] for _ in range(nt)]
    for kb in range(kblocks):
        for tt in range(ttn):
            x = np.stack([plane.read16(plane_off(32 * tt + R[l], chunk0 + 2 * kb + H[l], plane.rowb)) for l in range(64)])
            for t in range(nt):
                base = (tile0 + t) * tstride + (kb0 + kb) * 64
                wf = w[base:base + 64]
                acc[t][tt] = mfma_32x32x16(wf, x, acc[t][tt]) if w_is_a else mfma_32x32x16(x, wf, acc[t][tt])
    return acc


def acc_frag(acc, s):
    return acc[:, 8 * s:8 * s + 8]


def store_featrow_acc(acc, plane, feat0, tok0):
    for l in range(64):
        row = tok0 + R[l]
        for g in range(4):
            off = plane_off(row, (feat0 >> 3) + g, plane.rowb) + 8 * H[l]
            plane.write_elems(off, acc[l, 4 * g:4 * g + 4])
