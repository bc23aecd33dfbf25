"""Network oracle: torch-CPU fp32 restatement of the L3C forward (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

A *functional* restatement -- no nn.Module tree; it walks a reference state-dict (key schema: SURVEY.md section 5 /
Appendix A) with F.conv2d.  Restates, for the L3C configuration (configs/ms/cr.cf: EDSRLikeEnc / EDSRDec, enc.feed_F,
dec.skip, non-recursive):
  conv                <- pytorch_ext.default_conv :57-61 (padding = k//2, or = dilation when dilated)
  encoder             <- modules/net.py EDSRLikeEnc.forward :136-148, edsr.ResBlock :83-86
  quantise            <- modules/quantizer.py Quantizer.forward :62-90 (eval: hard symbols, first-min of torch.min)
  decoder             <- modules/net.py EDSRDec.forward :173-184, edsr.Upsampler :92-119 (conv 64->256 + PixelShuffle(2))
  prob_clf            <- modules/prob_clf.py StackedAtrousConvs.forward :71-74
  forward / get_P     <- modules/multiscale_network.py :226-246, :260-306, :308-322 (eval mode: decoders are fed bn_q)
  sample_forward      <- modules/multiscale_network.py :328-406 (the uniform draws are delegated to a `draw` callable)
The RGB baselines (BicubicSubsampling) are not restated here ("next").
"""
from collections import namedtuple

import torch
import torch.nn.functional as F

Hyper = namedtuple('Hyper', ['num_scales', 'Cf', 'C', 'L', 'K', 'enc_blocks', 'dec_blocks', 'levels_range'])
L3C_HYPER = Hyper(num_scales=3, Cf=64, C=5, L=25, K=10, enc_blocks=8, dec_blocks=8, levels_range=(-1, 1))

EncOut = namedtuple('EncOut', ['bn', 'bn_q', 'S', 'F'])
Out = namedtuple('Out', ['S', 'L', 'bn', 'P', 'F_enc', 'F_dec'])


def conv(x, sd, key, stride=1, rate=1):
    w, b = sd[key + '.weight'], sd[key + '.bias']
    k = w.shape[-1]
    return F.conv2d(x, w, b, stride=stride, dilation=rate, padding=(k // 2 if rate == 1 else rate))


def _body(x, sd, prefix, num_blocks):
    for i in range(num_blocks):
        r = conv(x, sd, '{}.{}.body.0'.format(prefix, i))
        r = F.relu(r)
        r = conv(r, sd, '{}.{}.body.2'.format(prefix, i))
        x = r + x
    return conv(x, sd, '{}.{}'.format(prefix, num_blocks))


def quantise(x, levels):
    N, C, H, W = x.shape
    d = torch.pow(x.reshape(N, C, H * W, 1) - levels, 2)
    _, sym = torch.min(d, dim=-1)
    sym = sym.view(N, C, H, W)
    return levels[sym], sym


def encoder(x, sd, s, hp):
    p = 'nets.{}.enc'.format(s)
    x = conv(x, sd, p + '.down', stride=2)
    x = _body(x, sd, p + '.body', hp.enc_blocks) + x
    feat = x
    bn = conv(x, sd, p + '.to_q.0')
    bn_q, sym = quantise(bn, sd[p + '.levels'])
    return EncOut(bn, bn_q, sym, feat)


def decoder(bn_q, fuse, sd, s, hp):
    p = 'nets.{}.dec'.format(s)
    x = conv(bn_q, sd, p + '.head')
    if fuse is not None:
        x = x + fuse
    x = _body(x, sd, p + '.body', hp.dec_blocks) + x
    x = conv(x, sd, p + '.tail.0')
    return F.pixel_shuffle(x, 2)


def prob_clf(feat, sd, s):
    p = 'prob_clfs.{}.atrous'.format(s)
    x = torch.cat([conv(feat, sd, '{}.atrous.{}'.format(p, i), rate=r) for i, r in enumerate((1, 2, 4))], dim=1)
    return conv(x, sd, p + '.lin')


def head(x, sd, s):
    if s == 0:
        x = conv(x, sd, 'heads.0.head.0')
        return conv(x, sd, 'heads.0.head.1.head')
    return conv(x, sd, 'heads.{}.head'.format(s))


def forward(img, sd, hp=L3C_HYPER):
    """img: (N,3,H,W) float 0..255 -> Out (lists fine->coarse as in multiscale_network.Out)."""
    S, Ls, bn = [img.round().long()], [256], [None]
    x = conv(img, sd, 'sub_rgb_mean')
    encs = []
    for s in range(hp.num_scales):
        e = encoder(head(x, sd, s), sd, s, hp)
        encs.append(e)
        x = e.F
    decs = [None] * hp.num_scales
    for s in reversed(range(hp.num_scales)):
        fuse = None if s == hp.num_scales - 1 else decs[s + 1]
        decs[s] = decoder(encs[s].bn_q, fuse, sd, s, hp)
    P = [prob_clf(decs[s], sd, s) for s in range(hp.num_scales)]
    for e in encs:
        S.append(e.S)
        Ls.append(hp.L)
        bn.append(e.bn_q)
    return Out(S, Ls, bn, P, [e.F for e in encs], decs)


def get_P(scale, bn_q, dec_F_prev, sd, hp=L3C_HYPER):
    f = decoder(bn_q, dec_F_prev, sd, scale, hp)
    return prob_clf(f, sd, scale), f


def default_draw(shape, lo, hi):
    """The reference's draw: torch.zeros_like(t).uniform_(lo, hi) on the CPU generator (consumes it in the same order)."""
    return torch.zeros(shape).uniform_(lo, hi)


def sample_forward(img, sd, sample_scales, hp=L3C_HYPER, draw=default_draw, partial_final=None):
    """multiscale_network.py:328-406: encode, then decode coarse -> fine feeding either the encoder's bn_q or the values
    sampled from the coarser scale's prediction; returns the sampled RGB image (N,3,H,W) float in [0, 255], not rounded."""
    from oracle import dmll
    x = conv(img, sd, 'sub_rgb_mean')
    encs, Cs = [], [3]
    for s in range(hp.num_scales):
        e = encoder(head(x, sd, s), sd, s, hp)
        encs.append(e)
        Cs.append(e.bn.shape[1])
        x = e.F
    prev_x, fuse = None, None
    for s in reversed(range(hp.num_scales)):
        spec = dmll.RGB if s == 0 else dmll.z_spec(hp.levels_range, hp.L)
        if s in sample_scales:
            if prev_x is None:
                fake = draw(tuple(encs[-1].bn_q.shape), -1, 1)
                prev_x, _ = quantise(fake, sd['nets.{}.enc.levels'.format(hp.num_scales - 1)])
                if partial_final:
                    for c in partial_final:
                        prev_x[:, c] = encs[s].bn_q[:, c]
            dec_in = prev_x
        else:
            dec_in = encs[s].bn_q
        fuse = decoder(dec_in, fuse, sd, s, hp)
        P = prob_clf(fuse, sd, s)
        if s == 0 or s - 1 in sample_scales:
            C = Cs[s]
            N, Kp, H, W = P.shape
            K = Kp // (spec.num_params * C)
            u_mix = draw((N, C, K, H, W), 1e-5, 1. - 1e-5)
            u_log = draw((N, C, H, W), 1e-5, 1. - 1e-5)
            prev_x = dmll.sample(spec, P, C, u_mix, u_log)
    return prev_x


# ---- RGB baselines (BicubicSubsampling encoders; configs/ms/cr_rgb_shared.cf, cr_rgb.cf) ---------------------------------------

RGB_SHARED_HYPER = Hyper(num_scales=1, Cf=64, C=3, L=256, K=10, enc_blocks=0, dec_blocks=8, levels_range=None)
RGB_HYPER = Hyper(num_scales=3, Cf=64, C=3, L=256, K=10, enc_blocks=0, dec_blocks=8, levels_range=None)
_RGB_MEAN = torch.tensor([0.4488, 0.4371, 0.4040], dtype=torch.float32).reshape(3, 1, 1).mul(255.)

RgbOut = namedtuple('RgbOut', ['S', 'L', 'bn', 'P', 'auto_recursive_from'])


def bicubic_encoder(x):
    """BicubicDownsamplingEnc.forward (modules/net.py:72-80) with resize_bicubic (dataloaders/images_loader.py:277-288):
    back to 0..255, round to uint8, PIL BICUBIC resize to (int(W*0.5), int(H*0.5)), symbols = pixel values, bn = value - mean."""
    from PIL import Image
    import numpy as np
    u8 = (x + _RGB_MEAN).clamp(0, 255.).round().to(torch.uint8)
    outs = []
    for n in range(u8.shape[0]):
        img = Image.fromarray(u8[n].permute(1, 2, 0).numpy())
        w, h = img.size
        img = img.resize((int(w * 0.5), int(h * 0.5)), Image.BICUBIC)
        outs.append(torch.from_numpy(np.array(img)).permute(2, 0, 1))
    down = torch.stack(outs, dim=0)
    return down.float() - _RGB_MEAN, down.long()


def forward_rgb(img, sd, hp, dec_skip, auto_recurse=0):
    """RGB-baseline forward (multiscale_network.py:226-306 with rgb_bicubic_baseline: identity heads, enc.feed_F False)."""
    scales = list(range(hp.num_scales)) + [-1] * auto_recurse
    x = conv(img, sd, 'sub_rgb_mean')
    S, bn = [img.round().long()], [None]
    inp = x
    encs = []
    for _ in scales:
        b, s = bicubic_encoder(inp)
        encs.append(b)
        S.append(s)
        bn.append(b)
        inp = b
    decs = []
    for i, s in reversed(list(enumerate(scales))):
        fuse = None if (not dec_skip or s == -1 or s == max(scales)) else decs[0]
        net = s if s >= 0 else hp.num_scales - 1
        decs.insert(0, decoder(encs[i], fuse, sd, net, hp))
    P = [prob_clf(decs[i], sd, s if s >= 0 else hp.num_scales - 1) for i, s in enumerate(scales)]
    return RgbOut(S, [256] * len(S), bn, P, hp.num_scales if auto_recurse > 0 else None)
