#!/usr/bin/env python
"""Numerics study BEFORE building a Winograd F(4x4,3x3) kernel (round-2 verdict, item 3): would the L3C forward stay inside
north_star's 1e-5 if every 3x3 stride-1 64-channel convolution ran as F(4x4,3x3) in fp32 instead of F(2x2,3x3)?

The whole forward (oracle/net.py's functional restatement, CPU) is run with the convolution swapped for an fp32 Winograd emulation:
  V = B^T d B (fp32), U = G g G^T (computed in fp64, rounded once to fp32: the packed weights), M = sum_c U (.) V (fp32 accumulate,
  the order of a GEMM over channels), Y = A^T M A (fp32), tiles of m x m outputs from (m+2) x (m+2) inputs;
for m = 2 (the shipped kernel's algorithm) and m = 4 with the standard points {0, +-1, +-2, inf} and with {0, +-1/2, +-1, inf}
(smaller transform constants), against (a) the same forward in fp64 direct form (truth) and (b) the fp32 direct oracle (what the
parity tests compare with).  Output: max |P - P_ref| / max |P_ref| per scale and parameter group, per checkpoint.

    python tests/numerics/wino_f43_numerics.py [H W] > profiles/r03_wino_f43_numerics.log
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402
from oracle import net as onet  # noqa: E402


def cook_toom(points, m, r=3):
    """-> (AT (m x a), G (a x r), BT (a x a)) for F(m, r) with a = m + r - 1 and the given a - 1 finite points (+ infinity), in
    float64, by solving the defining identities (Lavin & Gray 2015; Vandermonde construction)."""
    a = m + r - 1
    pts = list(points)
    assert len(pts) == a - 1
    # evaluation matrices: rows = points (last row = infinity)
    def vander(n):
        V = np.zeros((a, n))
        for i, p in enumerate(pts):
            V[i] = [p ** j for j in range(n)]
        V[a - 1, n - 1] = 1.0
        return V
    AT = vander(m).T                       # (m, a)
    Gm = vander(r)                         # (a, r)
    # B^T: d -> interpolation; solve so that  AT [(G g) . (BT d)] = conv(d, g)  for all g, d: use the polynomial identity
    # y = A^T [ (G g) (.) (B^T d) ]: B^T = inverse-transpose of the Vandermonde of size a scaled per row by N_i
    Va = vander(a)                         # (a, a)
    BT = np.linalg.inv(Va).T               # (a, a)
    # move the per-point normalisation into G so that BT has "nice" entries: scale row i of BT by s_i and row i of G by 1/s_i
    s = np.array([1.0 / np.abs(BT[i]).min() if False else 1.0 for i in range(a)])
    return AT, Gm / s[:, None], BT * s[:, None]


def check_transform(AT, G, BT, m, r=3):
    rng = np.random.RandomState(0)
    d = rng.randn(m + r - 1)
    g = rng.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-10), (y, ref)


def integer_bt(AT, G, BT):
    """Move each row's normalisation from B^T into G so that B^T has small integer entries (what a kernel would use: the input
    transform is then exact multiplications by 2, 3, 4, 5 and additions; G g G^T is computed offline in double precision)."""
    from fractions import Fraction
    BT2, G2 = BT.copy(), G.copy()
    for i in range(len(BT)):
        fr = [Fraction(float(v)).limit_denominator(1000) for v in BT[i]]
        den = 1
        for f in fr:
            den = den * f.denominator // np.gcd(den, f.denominator)
        ints = [int(f * den) for f in fr]
        g = 0
        for v in ints:
            g = np.gcd(g, abs(v))
        scale = Fraction(den, max(int(g), 1))
        BT2[i] = [float(f * scale) for f in fr]
        G2[i] = G[i] / float(scale)
    return AT, G2, BT2


class WinoConv(object):
    def __init__(self, m, points, integer=False):
        self.m = m
        AT, G, BT = cook_toom(points, m)
        if integer:
            AT, G, BT = integer_bt(AT, G, BT)
        check_transform(AT, G, BT, m)
        self.AT64, self.G64, self.BT64 = AT, G, BT
        self.AT = torch.from_numpy(AT).float()
        self.BT = torch.from_numpy(BT).float()
        self.cache = {}

    def weights(self, w):
        key = w.data_ptr()
        if key not in self.cache:
            G = torch.from_numpy(self.G64)
            U = torch.einsum('ai,koij,bj->koab', G, w.double(), G)          # (K, C, a, a) in fp64
            self.cache[key] = U.float()
        return self.cache[key]

    def conv(self, x, w, b):
        """3x3 stride 1 pad 1, fp32."""
        m, a = self.m, self.m + 2
        N, C, H, W = x.shape
        Hp, Wp = -(-H // m) * m, -(-W // m) * m
        xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
        d = F.unfold(xp, kernel_size=a, stride=m)                            # (N, C*a*a, T)
        T = d.shape[-1]
        d = d.reshape(N, C, a, a, T)
        V = torch.einsum('ai,ncijt,bj->nabct', self.BT, d, self.BT)           # fp32
        U = self.weights(w)                                                   # (K, C, a, a)
        M = torch.einsum('kcab,nabct->nabkt', U, V)                           # fp32 GEMM over c per position
        Y = torch.einsum('ia,nabkt,jb->nkijt', self.AT, M, self.AT)           # (N, K, m, m, T)
        K = w.shape[0]
        y = F.fold(Y.reshape(N, K * m * m, T), output_size=(Hp, Wp), kernel_size=m, stride=m)
        return y[:, :, :H, :W] + b.view(1, -1, 1, 1)


def make_conv(kind, wino=None, dtype=torch.float32):
    def conv(x, sd, key, stride=1, rate=1):
        w, b = sd[key + '.weight'], sd[key + '.bias']
        k = w.shape[-1]
        if wino is not None and k == 3 and stride == 1 and w.shape[1] == 64:
            if rate == 1:
                return wino.conv(x, w, b)
            out = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], dtype=x.dtype)
            for i in range(rate):
                for j in range(rate):
                    out[:, :, i::rate, j::rate] = wino.conv(x[:, :, i::rate, j::rate].contiguous(), w, b)
            return out
        return F.conv2d(x.to(dtype), w.to(dtype), b.to(dtype), stride=stride, dilation=rate, padding=(k // 2 if rate == 1 else rate))
    return conv


def forward_with(conv, img, sd, dtype=torch.float32):
    saved = onet.conv
    onet.conv = conv
    try:
        with torch.no_grad():
            return onet.forward(img.to(dtype), {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()})
    finally:
        onet.conv = saved


def group_errs(P, ref, num_params):
    n = P.shape[1] // num_params
    out = []
    for p in range(num_params):
        a, b = P[:, p * n:(p + 1) * n].double(), ref[:, p * n:(p + 1) * n].double()
        out.append(float((a - b).abs().max() / max(float(b.abs().max()), 1.0)))
    return out


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 384)
    torch.set_num_threads(8)
    cfg = config_parser.parse_builtin('ms', 'cr')
    variants = [('direct fp32 (oracle)', None),
                ('F(2x2,3x3) {0,1,-1}', WinoConv(2, [0, 1, -1])),
                ('F(4x4,3x3) {0,1,-1,2,-2}', WinoConv(4, [0, 1, -1, 2, -2])),
                ('F(4x4,3x3) {0,1,-1,1/2,-1/2}', WinoConv(4, [0, 1, -1, 0.5, -0.5])),
                ('F(4x4,3x3) {0,1,-1,1/2,-2}', WinoConv(4, [0, 1, -1, 0.5, -2])),
                ('F(4x4,3x3) {0,1,-1,2,-2} integer B^T (the kernel)', WinoConv(4, [0, 1, -1, 2, -2], integer=True)),
                ('F(4x4,3x3) {0,1,-1,1/2,-2} integer B^T', WinoConv(4, [0, 1, -1, 0.5, -2], integer=True)),
                ('F(3x3,3x3) {0,1,-1,2}', WinoConv(3, [0, 1, -1, 2])),
                ('F(3x3,3x3) {0,1,-1,1/2}', WinoConv(3, [0, 1, -1, 0.5]))]
    img = synthetic.make_image(H, W, 0, 'natural').unsqueeze(0).float()
    print('image {}x{} natural seed 0; errors are max |x - ref| / max(max |ref|, 1) per tensor / parameter group'.format(W, H))
    for ck in ('default', 'calibrated'):
        sd = synthetic.make_state_dict(cfg, 0, calibrated=(ck == 'calibrated'))
        truth = forward_with(make_conv('direct', None, torch.float64), img, sd, torch.float64)
        oracle = forward_with(make_conv('direct'), img, sd)
        print('== checkpoint', ck)
        for name, wino in variants:
            out = forward_with(make_conv('x', wino), img, sd) if wino is not None else oracle
            # decoder side must be compared on identical bottlenecks: recompute P from the ORACLE's bn_q with this conv
            saved = onet.conv
            onet.conv = make_conv('x', wino)
            try:
                with torch.no_grad():
                    f_prev, Ps, Fd = None, [None] * 3, [None] * 3
                    for s in (2, 1, 0):
                        Ps[s], f_prev = onet.get_P(s, oracle.bn[s + 1], f_prev, sd)
                        Fd[s] = f_prev
            finally:
                onet.conv = saved
            line = []
            for s in range(3):
                np_ = 4 if s == 0 else 3
                fe = float((out.F_enc[s].double() - truth.F_enc[s]).abs().max() / truth.F_enc[s].abs().max())
                vs_truth = group_errs(Ps[s], truth.P[s], np_) if (oracle.S[s + 1] == truth.S[s + 1]).all() else None
                vs_oracle = group_errs(Ps[s], oracle.P[s], np_)
                line.append('  scale {}: F_enc vs fp64 {:.2e} | P vs fp32 oracle (pi, mu, log sigma, lambda) {} | P vs fp64 {}'.format(
                    s, fe, ' '.join('%.2e' % v for v in vs_oracle),
                    ' '.join('%.2e' % v for v in vs_truth) if vs_truth else '(bottleneck symbols differ from fp64: n/a)'))
            flips = sum(int((out.S[s + 1] != oracle.S[s + 1]).sum()) for s in range(3))
            print('-- {}: symbol flips vs oracle {}'.format(name, flips))
            print('\n'.join(line))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
