#!/usr/bin/env python
"""Development: F(4x4,3x3) vs F(2x2,3x3) on SMALL launches (batch 1..8 at the resolutions of a 768x512 image): the F(4x4) block covers
16x16 pixels, the F(2x2) block 4x32 -- half as many blocks per launch, which matters when a launch cannot fill 512 block slots."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
w = torch.randn(64, 64, 3, 3, generator=g) / 24
b = torch.randn(64, generator=g)
layer = ops.PackedConv(w, b)


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (H, W) in [(512, 768), (256, 384), (128, 192), (64, 96)]:
    for B in (1, 2, 4, 8, 16):
        x = torch.randn(B, H, W, 64, generator=g).cuda()
        r = torch.randn(B, H, W, 64, generator=g).cuda()
        t2 = min(timed(lambda: ops.conv(x, layer, residual=r, impl='wino2')) for _ in range(2))
        t4 = min(timed(lambda: ops.conv(x, layer, residual=r, impl='wino4')) for _ in range(2))
        blocks4 = B * ((H + 15) // 16) * ((W + 15) // 16)
        print('{:4d}x{:<4d} B{:<3d} F(4x4) blocks {:6d} | F(2x2) {:8.1f} us | F(4x4) {:8.1f} us | ratio {:.2f}'.format(H, W, B, blocks4, t2, t4, t2 / t4))
        sys.stdout.flush()
