#!/usr/bin/env python
"""Development probe: end-to-end encode (incl. D2H + file assembly) and decode timing for a batch of 768x512 images."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding  # noqa: E402
from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint  # noqa: E402
from l3c_pytorch_amd.helpers import config_parser, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=8)
ap.add_argument('--H', type=int, default=512)
ap.add_argument('--W', type=int, default=768)
a = ap.parse_args()
cfg = config_parser.parse_builtin('ms', 'cr')
bp = MultiscaleBlueprint(cfg)
bp.net.load_state_dict(synthetic.make_state_dict(cfg, 0), strict=True)
bp.set_eval()
bc = Bitcoding(bp)
imgs = torch.stack([synthetic.make_image(a.H, a.W, i, 'natural') for i in range(a.B)]).long()
mp = a.B * a.H * a.W / 1e6
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    enc = bc.encode_batch(imgs.cuda())
    torch.cuda.synchronize(); t1 = time.time()
    files = enc.to_bytes()
    t2 = time.time()
    dec, _ = bc.decode_batch(files)
    torch.cuda.synchronize(); t3 = time.time()
    ok = torch.equal(dec.cpu(), imgs)
    print('rep {}: encode(GPU) {:.1f} ms = {:.1f} MPix/s | to_bytes {:.1f} ms | decode {:.1f} ms = {:.2f} MPix/s | lossless {} | {:.0f} KB/img'.format(
        rep, (t1 - t0) * 1e3, mp / (t1 - t0), (t2 - t1) * 1e3, (t3 - t2) * 1e3, mp / (t3 - t2), ok, sum(map(len, files)) / a.B / 1e3))
