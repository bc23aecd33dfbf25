"""Fit the CALIBRATED synthetic checkpoints: l3c-pytorch_amd/helpers/calibrated/<config>_seed<k>.npz.

Why: a default-init checkpoint predicts mixtures around 0 with sigma ~ 1 while the pixels are 0..255, so two of the three
RGB streams are coded entirely at the probability floor (16 bits per symbol, CDF tables saturated): parity, losslessness and
head / decoder timings are then not exercised in the regime the codec works in (round-2 verdict, row h).  The released weights
cannot be fetched (no network), so the last layer of every probability classifier (`prob_clfs.*.atrous.lin`, a 1x1 conv
192 -> Kp) is FITTED here in closed form on synthetic natural-like images, everything else keeps its seeded default init:

  * `nets.s.enc.to_q.0` is rescaled per channel so the bottleneck has zero mean and a spread that uses all L = 25 levels;
  * mu rows: ridge regression (float64 normal equations) of the coded value -- for G and B minus the lambda-coupled share of the
    channels already coded, logistic_mixture.py:262-272 -- on the 192 classifier features, one regression per (channel,
    mixture), the K mixtures offset around the prediction by multiples of the residual's spread;
  * log sigma rows: the seeded rows scaled to a per-pixel spread, biased per mixture from below the clamp at -7
    (logistic_mixture.py:57) up to a broad component (log sigma ~ 4 for RGB) that catches outliers;
  * logit pi rows: the seeded rows scaled up, biased towards the central mixtures;  lambda rows: scaled + biased so the
    coupling coefficients sigmoid(lambda) spread over ~0.5 .. 0.9.

The result is a proper (if weak) probability model: ~4-6 bpsp on the bench's images instead of 16.2, no channel at the floor.
This is a TEST-INPUT generator (it uses the oracle's CPU forward to get the features); the product only reads the npz through
helpers/synthetic.make_state_dict(..., calibrated=True).  Run in the build container:  python tests/golden/make_calibrated.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]

import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402
from oracle import dmll, net as onet  # noqa: E402

OUT_DIR = os.path.join(ROOT, 'l3c-pytorch_amd', 'helpers', 'calibrated')
CAL_SEED0 = 100000          # calibration images: seeds the bench / tests never use
PCA_TO_Q = True
TARGET_STD = 0.42           # bottleneck spread in units of the levels range (-1, 1)

# per-mixture design (K = 10): offset of the mean in residual spreads, log of sigma relative to the residual spread, logit bias
MU_OFF = np.array([0., -0.5, 0.5, -1.2, 1.2, -2.5, 2.5, 0., 0., 0.])
LS_REL = np.array([-1.1, -0.9, -0.9, -0.6, -0.6, -0.2, -0.2, 0.6, None, None], dtype=object)
PI_BIAS = np.array([1.5, 1.0, 1.0, 0.5, 0.5, 0., 0., -0.5, -2.0, -1.0])


def _features(feat, sd, s):
    p = 'prob_clfs.{}.atrous'.format(s)
    return torch.cat([onet.conv(feat, sd, '{}.atrous.{}'.format(p, i), rate=r) for i, r in enumerate((1, 2, 4))], dim=1)


def _ridge(X, T, lam=1e-3):
    """X (n,192) float64, T (n,m) -> W (m,192), b (m): least squares with a small ridge on the centred features."""
    mx, mt = X.mean(0), T.mean(0)
    Xc, Tc = X - mx, T - mt
    A = Xc.T @ Xc / len(X)
    A += lam * np.trace(A) / len(A) * np.eye(len(A))
    W = np.linalg.solve(A, Xc.T @ Tc / len(X)).T
    return W, mt - W @ mx


def _fit_head(sd, key, X, x_coded, rgb, abs_ls_broad, abs_ls_spike=-7.6):
    """Rewrite sd[key + '.weight' / '.bias'] (Kp,192,1,1).  X: (n,192) features; x_coded: (n,C) the values being coded."""
    w0 = sd[key + '.weight'].double().numpy()[:, :, 0, 0]
    b0 = sd[key + '.bias'].double().numpy()
    C = x_coded.shape[1]
    P = 4 if rgb else 3
    K = w0.shape[0] // (P * C)
    assert K == 10
    w, b = w0.copy(), b0.copy()
    raw_std = (X @ w0.T + b0).std(0)                         # per-pixel spread the seeded rows produce

    def row(p, c, k):
        return p * C * K + c * K + k

    sig = lambda v: 1. / (1. + np.exp(-v))
    if rgb:                                                  # lambda rows: p = 3, [g<-r, b<-r, b<-g]
        for j, centre in enumerate((3.0, -0.1, -0.1)):
            for k in range(K):
                r = row(3, j, k)
                g = 0.4 / max(raw_std[r], 1e-9)
                w[r], b[r] = w0[r] * g, -(X @ w0[r]).mean() * g + centre + 0.15 * (k - 4.5) / 4.5
    lam = sig(X @ w.T + b)[:, 3 * C * K:] if rgb else None
    report = []
    for c in range(C):
        T = np.repeat(x_coded[:, c:c + 1], K, axis=1)
        if rgb and c == 1:
            T = T - lam[:, 0 * K:1 * K] * x_coded[:, 0:1]
        elif rgb and c == 2:
            T = T - lam[:, 1 * K:2 * K] * x_coded[:, 0:1] - lam[:, 2 * K:3 * K] * x_coded[:, 1:2]
        Wc, bc = _ridge(X, T)
        res = (T - (X @ Wc.T + bc)).std(0)
        spread = float(res.mean())
        report.append(spread)
        for k in range(K):
            r = row(1, c, k)
            w[r], b[r] = Wc[k], bc[k] + MU_OFF[k] * spread
            r = row(2, c, k)                                 # log sigma
            g = 0.35 / max(raw_std[r], 1e-9)
            centre = (np.log(spread) + LS_REL[k]) if LS_REL[k] is not None else (abs_ls_spike if k == 8 else abs_ls_broad)
            w[r], b[r] = w0[r] * g, -(X @ w0[r]).mean() * g + centre
            r = row(0, c, k)                                 # logit pi
            g = 0.8 / max(raw_std[r], 1e-9)
            w[r], b[r] = w0[r] * g, -(X @ w0[r]).mean() * g + PI_BIAS[k]
    sd[key + '.weight'] = torch.from_numpy(w.astype(np.float32)).reshape(-1, w.shape[1], 1, 1).contiguous()
    sd[key + '.bias'] = torch.from_numpy(b.astype(np.float32)).contiguous()
    return report


def _flat(t):          # (N,C,H,W) -> (N*H*W, C) float64
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).double().numpy()


def calibrate_l3c(seed=0, n_img=3, H=512, W=768, verbose=True):
    cfg = config_parser.parse_builtin('ms', 'cr')
    hp = onet.L3C_HYPER
    sd = synthetic.make_state_dict(cfg, seed)
    changed = []
    img = torch.stack([synthetic.make_image(H, W, CAL_SEED0 + i) for i in range(n_img)]).float()
    with torch.no_grad():
        x = onet.conv(img, sd, 'sub_rgb_mean')
        encs = []
        for s in range(hp.num_scales):
            h = onet.head(x, sd, s)
            e = onet.encoder(h, sd, s, hp)
            key = 'nets.{}.enc.to_q.0'.format(s)
            if PCA_TO_Q:                      # rows 0-2: the feature directions that best predict the 2x2-pooled R, G, B of
                Ff = _flat(e.F)               # this scale's input; rows 3-4: the strongest principal directions of what is left
                mu_f = Ff.mean(0)
                Fc = Ff - mu_f
                pooled = _flat(torch.nn.functional.avg_pool2d(img, 2 ** (s + 1)))
                Wr, _ = _ridge(Ff, pooled, lam=1e-4)                             # (3, Cf)
                Q, _ = np.linalg.qr(Wr.T)                                        # orthonormal basis of the 3 directions
                rest = Fc - (Fc @ Q) @ Q.T
                evals, evecs = np.linalg.eigh(np.cov(rest.T))
                top = np.concatenate([Wr, evecs[:, ::-1][:, :e.bn.shape[1] - 3].T])   # (C, Cf)
                top = top * np.sign(top[np.arange(len(top)), np.abs(top).argmax(1)])[:, None]   # fixed sign convention
                g = TARGET_STD / (Fc @ top.T).std(0)
                sd[key + '.weight'] = torch.from_numpy((top * g[:, None]).astype(np.float32)).reshape(-1, top.shape[1], 1, 1).contiguous()
                sd[key + '.bias'] = torch.from_numpy((-(top @ mu_f) * g).astype(np.float32)).contiguous()
            else:
                m, sdev = e.bn.mean((0, 2, 3)), e.bn.std((0, 2, 3))
                g = TARGET_STD / sdev
                sd[key + '.weight'] = (sd[key + '.weight'] * g.view(-1, 1, 1, 1)).contiguous()
                sd[key + '.bias'] = ((sd[key + '.bias'] - m) * g).contiguous()
            changed += [key + '.weight', key + '.bias']
            e = onet.encoder(h, sd, s, hp)
            if verbose:
                used = [int(torch.unique(e.S[:, c]).numel()) for c in range(e.S.shape[1])]
                print('scale', s + 1, 'bottleneck levels used per channel', used)
            encs.append(e)
            x = e.F
        fuse = None
        for s in reversed(range(hp.num_scales)):
            fuse = onet.decoder(encs[s].bn_q, fuse, sd, s, hp)
            X = _flat(_features(fuse, sd, s))
            coded = _flat(img if s == 0 else encs[s - 1].bn_q)
            key = 'prob_clfs.{}.atrous.lin'.format(s)
            rep = _fit_head(sd, key, X, coded, rgb=(s == 0), abs_ls_broad=(4.0 if s == 0 else 0.3))
            changed += [key + '.weight', key + '.bias']
            if verbose:
                print('scale', s, 'residual spread per channel', ['%.3f' % r for r in rep])
    return cfg, sd, changed


def calibrate_rgb_shared(seed=0, n_img=2, H=512, W=768, verbose=True):
    """cr_rgb_shared (one decoder + one RGB head shared by every pyramid level, auto_recurse 3): fit on levels 0..3 together."""
    cfg = config_parser.parse_builtin('ms', 'cr_rgb_shared')
    hp = onet.RGB_SHARED_HYPER
    sd = synthetic.make_state_dict(cfg, seed)
    img = torch.stack([synthetic.make_image(H, W, CAL_SEED0 + i) for i in range(n_img)]).float()
    Xs, Ts = [], []
    with torch.no_grad():
        x = onet.conv(img, sd, 'sub_rgb_mean')
        levels, inp = [img], x
        bns = []
        for _ in range(4):
            b, s = onet.bicubic_encoder(inp)
            bns.append(b)
            levels.append(s.float())
            inp = b
        for i in range(4):
            f = onet.decoder(bns[i], None, sd, 0, hp)
            Xs.append(_flat(_features(f, sd, 0)))
            Ts.append(_flat(levels[i]))
    key = 'prob_clfs.0.atrous.lin'
    rep = _fit_head(sd, key, np.concatenate(Xs), np.concatenate(Ts), rgb=True, abs_ls_broad=4.0)
    if verbose:
        print('rgb_shared residual spread per channel', ['%.3f' % r for r in rep])
    return cfg, sd, [key + '.weight', key + '.bias']


def evaluate_l3c(sd, seeds=(0, 1), H=512, W=768):
    from oracle import bitcoding as obc
    for sdd in seeds:
        img = synthetic.make_image(H, W, sdd).unsqueeze(0).float()
        with torch.no_grad():
            out = onet.forward(img, sd)
            bpsp = obc.losses_bpsp(out)
        P0 = out.P[0].reshape(1, 4, 3, 10, H, W)
        print('image', sdd, 'bpsp per scale', ['%.4f' % v for v in bpsp], 'total %.4f' % sum(bpsp),
              '| mu range', float(P0[:, 1].min()), float(P0[:, 1].max()),
              '| log sigma range', float(P0[:, 2].min()), float(P0[:, 2].max()),
              '| clamp fires on %.3f of the entries' % float((P0[:, 2] < -7).float().mean()),
              '| z levels used', [int(torch.unique(s).numel()) for s in out.S[1:]])


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    torch.set_num_threads(8)
    cfg, sd, changed = calibrate_l3c()
    evaluate_l3c(sd)
    np.savez(os.path.join(OUT_DIR, 'cr_seed0.npz'), **{k: sd[k].numpy() for k in changed})
    cfg, sd, changed = calibrate_rgb_shared()
    np.savez(os.path.join(OUT_DIR, 'cr_rgb_shared_seed0.npz'), **{k: sd[k].numpy() for k in changed})


if __name__ == '__main__':
    main()
