#!/usr/bin/env python
"""Development probe: Winograd F(4x4,3x3) (csrc/conv_wino4.hip) vs F(2x2,3x3) (csrc/conv_wino.hip) -- accuracy against fp64 and time
on the layers of the headline configuration."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--B', type=int, default=32)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# accuracy on a mid-sized layer
w = torch.randn(64, 64, 3, 3, generator=g) / 24
b = torch.randn(64, generator=g)
layer = ops.PackedConv(w, b)
x = torch.randn(2, 50, 70, 64, generator=g).cuda()
ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
for impl in ('wino2', 'wino4'):
    got = ops.conv(x, layer, impl=impl)
    print(impl, 'max |err| vs fp64: %.3g' % (got.double().cpu() - ref64).abs().max().item())

for (H, W, cout, dil, relu, res, shuffle, label) in [
        (256, 384, 64, 1, True, False, False, 'body conv + ReLU @R1'),
        (256, 384, 64, 1, False, True, False, 'body conv + residual @R1'),
        (256, 384, 256, 1, False, False, True, 'tail 64->256 + PixelShuffle @R1'),
        (512, 768, 64, 1, False, False, False, 'atrous d1 @R0'),
        (512, 768, 64, 2, False, False, False, 'atrous d2 @R0'),
        (512, 768, 64, 4, False, False, False, 'atrous d4 @R0'),
        (128, 192, 64, 1, True, False, False, 'body conv + ReLU @R2'),
        (64, 96, 64, 1, True, False, False, 'body conv + ReLU @R3')]:
    B = a.B if H < 512 else max(1, a.B // 4)
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24
    b = torch.randn(cout, generator=g)
    layer = ops.PackedConv(w, b, dilation=dil)
    x = torch.randn(B, H, W, 64, generator=g).cuda()
    r = torch.randn(B, H, W, cout, generator=g).cuda() if res else None
    flops = 2.0 * B * H * W * cout * 64 * 9
    line = '{:34s} B{:3d}'.format(label, B)
    outs, best = {}, {}
    for rep in range(2):          # alternate, keep the better of two: the clock the chip grants drifts by several per cent
        for impl in ('wino2', 'wino4'):
            fn = lambda: ops.conv(x, layer, relu=relu, residual=r, pixel_shuffle=shuffle, impl=impl)   # noqa: E731
            best[impl] = min(best.get(impl, 1e9), timed(fn, a.iters))
            outs[impl] = fn()
    for impl in ('wino2', 'wino4'):
        line += ' | {} {:7.3f} ms {:6.1f} alg. TFLOP/s'.format('F(2x2)' if impl == 'wino2' else 'F(4x4)', best[impl], flops / best[impl] / 1e9)
    line += ' | max diff %.2e' % (outs['wino2'] - outs['wino4']).abs().max().item()
    print(line)
    sys.stdout.flush()
