"""Lane-level numpy model of the fp32-MFMA MLP chain in csrc/tn_field.hip (CPU only).

The model executes `v_mfma_f32_32x32x2_f32` exactly as the gfx950 lane maps define it
(A: lane l holds A[l&31][l>>5]; B: lane l holds B[l>>5][l&31]; D: register r of lane l is
D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]) and replays the kernel's packing / chaining / weight-gradient index
arithmetic against plain matrix algebra.  It pins the index formulas the HIP kernels use, so that a layout
mistake is caught here on the CPU instead of on the GPU box.
"""
import numpy as np

LANES = np.arange(64)
J, H = LANES & 31, LANES >> 5


def rrow(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a, b, acc):
    """acc: [64 lanes, 16 regs]; a, b: [64]."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    A[J, H] = a
    B[H, J] = b
    D = A @ B
    out = acc.copy()
    for r in range(16):
        out[:, r] += D[rrow(r, H), J]
    return out


def slot_to_col(s):
    return s if s < 16 else (-1 if s == 16 else s - 1)


class Net:
    def __init__(self, rng, C=4):
        u = lambda *s: rng.uniform(-1, 1, s)  # noqa: E731
        self.C = C
        self.w = [u(64, 32), u(16, 64), u(64, 63), u(64, 64), u(C, 64)]
        self.b = [u(64), u(16), u(64), u(64), u(C)]
        self.mo = [2, 1, 2, 2, 1]
        self.ti = [1, 2, 2, 2, 2]

    def weight_at(self, layer, o, i):
        w = self.w[layer]
        if layer == 2:
            c = slot_to_col(i)
            return w[o, c] if (o < 64 and i < 64 and c >= 0) else 0.0
        return w[o, i] if (o < w.shape[0] and i < w.shape[1]) else 0.0

    def af(self, layer, m, t, r):
        return np.array([self.weight_at(layer, 32 * m + (l & 31), 32 * t + rrow(r, l >> 5)) for l in LANES])

    def ab(self, layer, t, m, r):
        return np.array([self.weight_at(layer, 32 * m + rrow(r, l >> 5), 32 * t + (l & 31)) for l in LANES])

    def bias_tile(self, layer, m):
        b = self.b[layer]
        out = np.zeros((64, 16))
        for r in range(16):
            idx = 32 * m + rrow(r, H)
            out[:, r] = np.where(idx < len(b), b[np.minimum(idx, len(b) - 1)], 0.0)
        return out


def to_tile(x, m):
    """row-major [32 samples, F] -> D-layout registers [64,16] of feature tile m."""
    out = np.zeros((64, 16))
    for r in range(16):
        out[:, r] = x[J, 32 * m + rrow(r, H)]
    return out


def from_tile(t, F, m, x=None):
    x = np.zeros((32, F)) if x is None else x
    for r in range(16):
        for l in LANES:
            f = 32 * m + rrow(r, l >> 5)
            if f < F:
                x[l & 31, f] = t[l, r]
    return x


def test_forward_chain_matches_dense_algebra():
    rng = np.random.default_rng(0)
    net = Net(rng)
    enc = rng.uniform(-1, 1, (32, 32))
    sh = rng.uniform(-1, 1, (32, 16))
    emb = rng.uniform(-1, 1, (32, 32))
    relu = lambda v: np.maximum(v, 0)  # noqa: E731
    # dense reference
    h1 = relu(enc @ net.w[0].T + net.b[0])
    bo = h1 @ net.w[1].T + net.b[1]
    hin = np.concatenate([sh, bo[:, 1:], emb], axis=1)  # 63 inputs
    hh1 = relu(hin @ net.w[2].T + net.b[2])
    hh2 = relu(hh1 @ net.w[3].T + net.b[3])
    out = hh2 @ net.w[4].T + net.b[4]
    # kernel replay
    in0 = to_tile(enc, 0)
    a = [net.bias_tile(0, 0), net.bias_tile(0, 1)]
    for r in range(16):
        for m in range(2):
            a[m] = mfma(net.af(0, m, 0, r), in0[:, r], a[m])
    a = [relu(x) for x in a]
    assert np.allclose(from_tile(a[1], 64, 1, from_tile(a[0], 64, 0)), h1)
    b = net.bias_tile(1, 0)
    for t in range(2):
        for r in range(16):
            b = mfma(net.af(1, 0, t, r), a[t][:, r], b)
    assert np.allclose(from_tile(b, 16, 0), bo)
    assert np.allclose(b[H == 0, 0], bo[:, 0])  # density logit: half 0, register 0
    hi0 = np.zeros((64, 16))
    for r in range(8):
        hi0[:, r] = sh[J, rrow(r, H)]
        hi0[:, 8 + r] = b[:, r]
    hi1 = np.zeros((64, 16))
    for g in range(4):
        for q in range(4):
            hi1[:, 4 * g + q] = emb[J, 8 * g + 4 * H + q]
    c = [net.bias_tile(2, 0), net.bias_tile(2, 1)]
    for t, hi in enumerate((hi0, hi1)):
        for r in range(16):
            for m in range(2):
                c[m] = mfma(net.af(2, m, t, r), hi[:, r], c[m])
    c = [relu(x) for x in c]
    assert np.allclose(from_tile(c[1], 64, 1, from_tile(c[0], 64, 0)), hh1)
    d = [net.bias_tile(3, 0), net.bias_tile(3, 1)]
    for t in range(2):
        for r in range(16):
            for m in range(2):
                d[m] = mfma(net.af(3, m, t, r), c[t][:, r], d[m])
    d = [relu(x) for x in d]
    e = net.bias_tile(4, 0)
    for t in range(2):
        for r in range(16):
            e = mfma(net.af(4, 0, t, r), d[t][:, r], e)
    for ch in range(net.C):
        assert np.allclose(e[H == 0, ch], out[:, ch])  # rows 0..C-1: half 0, registers 0..C-1


def test_backward_chain_matches_dense_algebra():
    rng = np.random.default_rng(1)
    net = Net(rng)
    g3 = rng.uniform(-1, 1, (32, 4))
    m_hh2 = rng.uniform(0, 1, (32, 64)) > 0.4
    m_hh1 = rng.uniform(0, 1, (32, 64)) > 0.4
    m_h1 = rng.uniform(0, 1, (32, 64)) > 0.4
    g_pre = rng.uniform(-1, 1, 32)  # trunc_exp gradient for the density logit
    # dense reference
    d_hh2 = (g3 @ net.w[4]) * m_hh2
    d_hh1 = (d_hh2 @ net.w[3]) * m_hh1
    d_hin = d_hh1 @ net.w[2]  # [32, 63]
    d_bo = np.concatenate([g_pre[:, None], d_hin[:, 16:31]], axis=1)
    d_h1 = (d_bo @ net.w[1]) * m_h1
    d_enc = d_h1 @ net.w[0]
    # kernel replay
    zero = np.zeros((64, 16))
    g3r = np.zeros((64, 4))
    g3r[H == 0] = g3
    dd = [zero, zero]
    for r in range(4):
        for t in range(2):
            dd[t] = mfma(net.ab(4, t, 0, r), g3r[:, r], dd[t])
    dd = [dd[t] * to_tile(m_hh2.astype(float), t) for t in range(2)]
    assert np.allclose(from_tile(dd[1], 64, 1, from_tile(dd[0], 64, 0)), d_hh2)
    dc = [zero, zero]
    for m in range(2):
        for r in range(16):
            for t in range(2):
                dc[t] = mfma(net.ab(3, t, m, r), dd[m][:, r], dc[t])
    dc = [dc[t] * to_tile(m_hh1.astype(float), t) for t in range(2)]
    di = [zero, zero]
    for m in range(2):
        for r in range(16):
            for t in range(2):
                di[t] = mfma(net.ab(2, t, m, r), dc[m][:, r], di[t])
    slots = from_tile(di[1], 64, 1, from_tile(di[0], 64, 0))
    assert np.allclose(slots[:, :16], d_hin[:, :16]) and np.allclose(slots[:, 17:], d_hin[:, 16:])
    assert np.allclose(slots[:, 16], 0.0)
    dbo = np.zeros((64, 8))
    for r in range(8):
        dbo[:, r] = di[0][:, 8 + r]
    dbo[H == 0, 0] = g_pre
    assert np.allclose(np.stack([dbo[l, r] for l in LANES for r in range(8)]).reshape(64, 8)[H == 0][:, :4], d_bo[:, 0:4])
    dh = [zero, zero]
    for r in range(8):
        for t in range(2):
            dh[t] = mfma(net.ab(1, t, 0, r), dbo[:, r], dh[t])
    dh = [dh[t] * to_tile(m_h1.astype(float), t) for t in range(2)]
    assert np.allclose(from_tile(dh[1], 64, 1, from_tile(dh[0], 64, 0)), d_h1)
    de = zero
    for m in range(2):
        for r in range(16):
            de = mfma(net.ab(0, 0, m, r), dh[m][:, r], de)
    assert np.allclose(from_tile(de, 32, 0), d_enc)


def test_wgrad_tile_and_epilogue_indices():
    rng = np.random.default_rng(2)
    P, out_dim, in_dim = 10, 16, 64
    dY = rng.uniform(-1, 1, (P, out_dim))
    X = rng.uniform(-1, 1, (P, in_dim))
    ref = dY.T @ X
    dW = np.zeros((out_dim, in_dim))
    db = np.zeros(out_dim)
    for b in range(2):  # input tiles
        acc = np.zeros((64, 16))
        bsum = np.zeros(64)
        for p0 in range(0, P, 2):
            p = p0 + H
            ok = p < P
            pc = np.minimum(p, P - 1)
            av = np.where(ok & (J < out_dim), dY[pc, np.minimum(J, out_dim - 1)], 0.0)
            bv = np.where(ok, X[pc, 32 * b + J], 0.0)
            if b == 0:
                bsum += av
            acc = mfma(av, bv, acc)
        for r in range(16):
            for l in LANES:
                o = rrow(r, l >> 5)
                if o < out_dim:
                    dW[o, 32 * b + (l & 31)] += acc[l, r]
        if b == 0:
            for l in LANES:
                if (l >> 5) == 0 and (l & 31) < out_dim:
                    db[l & 31] = bsum[l] + bsum[l + 32]
    assert np.allclose(dW, ref)
    assert np.allclose(db, dY.sum(0))


def test_store_tile_chunks_are_contiguous_float4():
    # registers 4g..4g+3 of tile m <-> features 32m + 8g + 4h + {0..3}: what store_tile/load_tile rely on
    for h in (0, 1):
        for g in range(4):
            assert [rrow(4 * g + q, h) for q in range(4)] == [8 * g + 4 * h + q for q in range(4)]
