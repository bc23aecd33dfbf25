#!/usr/bin/env python
"""Numerics study for the decoder item of DESIGN.md section 8 (windowed table rows): how often would a symbol of the RGB scale fall
outside a row of W entries placed around the mixture -- i.e. how often would a decoder that gets W instead of 257 entries per symbol
have to evaluate the rest itself?  CPU only: the oracle's forward pass (tests infrastructure) on one bench image, both synthetic
checkpoints.

    python tests/numerics/window_miss_rate.py [--H 512 --W 768] > profiles/r04_window_miss_rate.log

Per channel the window is [l0, l0 + W) with l0 = clamp(round(m) - W / 2, 0, 256 - W), m = the mixture MEAN sum_k pi_k mu_k (the means of G
and B with their lambda coupling to the symbols already decoded, as the decoder has them).  Also reported: the mass of the mixture
outside the window (the decoder could use it to fall back to a full row for a pixel), as a distribution."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402
from oracle import dmll, net  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--H', type=int, default=512)
ap.add_argument('--W', type=int, default=768)
ap.add_argument('--image', type=int, default=0)
a = ap.parse_args()
torch.set_num_threads(8)
cfg = config_parser.parse_builtin('ms', 'cr')
img = synthetic.make_image(a.H, a.W, a.image, 'natural')[None].float()
print('image {} ({}x{}, natural-like synthetic), RGB scale of L3C cr; K = 10 components per channel'.format(a.image, a.W, a.H))
for name, calibrated in (('calibrated checkpoint (6.2 bpsp)', True), ('default-init checkpoint (16.2 bpsp)', False)):
    sd = {k: torch.as_tensor(v) for k, v in synthetic.make_state_dict(cfg, 0, calibrated=calibrated).items()}
    with torch.no_grad():
        out = net.forward(img, sd)
    P = out.P[0]
    x = out.S[0].float()
    print('\n' + name)
    for c, ch in enumerate('RGB'):
        pi, mu, ls = dmll.params_for_channel(dmll.RGB, P, c, 3, x)
        mean = (pi * mu).sum(1)[0]                       # (H, W)
        sig = torch.exp(ls)
        sym = x[0, c]
        line = '  {}: sigma of the heaviest component: median {:.2f}, 90 % {:.2f}'.format(
            ch, float(sig.gather(1, pi.argmax(1, keepdim=True)).median()), float(sig.gather(1, pi.argmax(1, keepdim=True)).quantile(0.9)))
        print(line)
        for Wn in (32, 64, 96, 128):
            l0 = (mean.round() - Wn // 2).clamp(0, 256 - Wn)
            inside = (sym >= l0) & (sym < l0 + Wn)
            # mass of the mixture inside the window: cdf(l0 + Wn - 0.5) - cdf(l0 - 0.5) with the two ends of the alphabet open
            def cdf(t):
                return (pi * torch.sigmoid((t[None, None] - mu) / sig)).sum(1)[0]
            hi = torch.where(l0 + Wn >= 256, torch.ones_like(l0), cdf(l0 + Wn - 0.5))
            lo = torch.where(l0 <= 0, torch.zeros_like(l0), cdf(l0 - 0.5))
            outside = 1.0 - (hi - lo)
            print('     window of {:3d} entries: symbol outside for {:8.4f} % of the pixels; mixture mass outside: median {:.1e}, 99 % {:.1e}; '
                  'pixels with more than 2^-10 outside: {:6.2f} %'.format(
                      Wn, 100.0 * float((~inside).float().mean()), float(outside.median()), float(outside.quantile(0.99)),
                      100.0 * float((outside > 2.0 ** -10).float().mean())))
