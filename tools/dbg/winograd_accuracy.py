"""Accuracy of Winograd F(2,5) / F(4,5) fp32 k5 convs in the TemporalUnet (the kernel uses F(4,5)), against
fp64 truth, next to the direct fp32 conv.  CPU emulation: every stride-1 k5 conv of the oracle UNet is replaced."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, torch.nn.functional as F
from fractions import Fraction as Fr
from oracle import mmd_oracle as O
from mmd_amd import synth
from cases import rel_l2


def cook_toom(points, m, r):
    """AT [m, a], G [a, r], BT [a, a] for F(m, r) with finite `points` + infinity, as float64 arrays (exact rationals)."""
    a = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == a - 1
    # polynomial M(x) = prod (x - p_i); BT rows from Lagrange basis; standard construction
    def polymul(p, q):
        out = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                out[i + j] += x * y
        return out
    AT = [[pts[j] ** i for j in range(a - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    G = []
    for i in range(a - 1):
        N = Fr(1)
        for j in range(a - 1):
            if j != i:
                N *= (pts[i] - pts[j])
        G.append([pts[i] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT = []
    for i in range(a - 1):
        poly = [Fr(1)]
        for j in range(a - 1):
            if j != i:
                poly = polymul(poly, [-pts[j], Fr(1)])
        BT.append(poly + [Fr(0)])          # degree a-2 -> a-1 coeffs, pad
    Mx = [Fr(1)]
    for j in range(a - 1):
        Mx = polymul(Mx, [-pts[j], Fr(1)])
    BT.append(Mx)                            # degree a-1: a coeffs
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def check(points, m=2):
    AT, G, BT = cook_toom(points, m, 5)
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(m + 4), rng.standard_normal(5)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([np.dot(d[i:i + 5], g) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-12), (y, ref)
    return AT, G, BT


def make_conv(orig, AT, G, BT, dtype):
    AT_t, G_t, BT_t = (torch.tensor(M, dtype=dtype) for M in (AT, G, BT))
    m = AT.shape[0]                                             # outputs per tile
    def conv(x, w, b=None, stride=1, padding=0, **kw):
        if w.shape[-1] != 5 or stride != 1:
            return orig(x, w, b, stride=stride, padding=padding, **kw)
        n, cin, L = x.shape
        U = torch.einsum('pk,oik->poi', torch.tensor(G, dtype=torch.float64), w.double()).to(dtype)   # host-side, fp64
        xp = F.pad(x, (2, 2))
        d = xp.unfold(2, m + 4, m)                              # [n, cin, L/m, m+4]
        V = torch.einsum('pj,nitj->pnit', BT_t, d)              # [6, n, cin, tiles]
        M = torch.einsum('poi,pnit->pnot', U, V)
        Y = torch.einsum('mp,pnot->notm', AT_t, M).reshape(n, w.shape[0], L)
        return Y if b is None else Y + b.view(1, -1, 1)
    return conv


sd = O.state_dict_to_torch(synth.synth_unet_state_dict(0))
x = torch.from_numpy(synth.synth_noise(5, (16, 64, 4)))
c1 = F.conv1d
def conv64(x, w, b=None, **kw):      # "truth": every conv accumulated in fp64, rounded once
    return c1(x.double(), w.double(), None if b is None else b.double(), **kw).float()
for t in (0, 37, 99):
    tt = torch.full((16,), t)
    F.conv1d = conv64
    truth = O.unet_forward(sd, x, tt)
    F.conv1d = c1
    direct = O.unet_forward(sd, x, tt)
    print(f"t={t}: direct fp32 vs fp64 {rel_l2(direct.double(), truth):.2e}")
    for name, m, pts in (("F(2,5) 0,+-1,+-2", 2, [0, 1, -1, 2, -2]), ("F(2,5) 0,+-1,+-1/2", 2, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
                         ("F(4,5) 0,+-1,+-2,+-1/2 (the kernel's)", 4, [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)])):
        AT, G, BT = check(pts, m)
        F.conv1d = make_conv(c1, AT, G, BT, torch.float32)
        try:
            out = O.unet_forward(sd, x, tt)
        finally:
            F.conv1d = c1
        print(f"   winograd {name} fp32 vs fp64 {rel_l2(out.double(), truth):.2e}   vs direct fp32 {rel_l2(out, direct):.2e}")
