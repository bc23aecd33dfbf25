#!/usr/bin/env python
"""Development probe (round-5 verdict, next 1): the F(4x4,3x3) probe kernel on v_mfma_f32_32x32x2_f32 (csrc/conv_wino4w.hip, test-only
library) against the product's conv_wino4_kernel -- correctness (fp64, the implicit GEMM, the product kernel) and time on the verdict's
shape: 64 -> 64, 256 x 384, batch 32, ReLU.

usage: python tools/wino4w_probe.py [B]      (L3C_W4W_TPB = tiles per block of the probe kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)


def best_ms(fn, reps=3, iters=10):
    best = 1e9
    for _ in range(reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


# ---- correctness: odd sizes (ragged tiles on both axes), against fp64, the implicit GEMM and the product kernel
for (b, H, W, cout, relu) in [(2, 50, 70, 64, True), (1, 16, 32, 64, False), (3, 37, 129, 64, True), (1, 256, 384, 64, True), (2, 40, 200, 40, False)]:
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24
    bias = torch.randn(cout, generator=g)
    layer = ops.PackedConv(w, bias)
    x = torch.randn(b, H, W, 64, generator=g).cuda()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.clamp_min(0)
    line = '{}x{}x{} cout {} relu {}:'.format(b, H, W, cout, int(relu))
    outs = {}
    for impl in ('wino4w', 'wino4', 'gemm'):
        for tpb in (('1', '2', '0') if impl == 'wino4w' else ('0',)):
            os.environ['L3C_W4W_TPB'] = tpb
            out = torch.full((b, H, W, cout), float('nan'), device='cuda')
            ops.conv(x, layer, out=out, relu=relu, impl=impl)
            err = (out.double().cpu() - ref).abs().max().item()
            outs[impl + tpb] = out
            line += '  {}{} {:.2e}'.format(impl, '/tpb' + tpb if impl == 'wino4w' else '', err)
    same = all(torch.equal(outs['wino4w0'], outs['wino4w' + t]) for t in ('1', '2'))
    print(line, ' probe vs product max |diff| {:.2e}'.format((outs['wino4w0'] - outs['wino40']).abs().max().item()), ' tpb-invariant:', same, flush=True)

# ---- time on the verdict's shape
H, W = 256, 384
w = torch.randn(64, 64, 3, 3, generator=g) / 24
bias = torch.randn(64, generator=g)
layer = ops.PackedConv(w, bias)
x = torch.randn(B, H, W, 64, generator=g).cuda()
out = torch.empty(B, H, W, 64, device='cuda')
flops = 2.0 * B * H * W * 64 * 64 * 9
t_prod = best_ms(lambda: ops.conv(x, layer, out=out, relu=True, impl='wino4'))
print('product conv_wino4_kernel            B={}: {:.3f} ms  ({:.0f} algorithmic TFLOP/s)'.format(B, t_prod, flops / t_prod / 1e9))
for tpb in ('1', '2', '3', '4', '6', '12'):
    os.environ['L3C_W4W_TPB'] = tpb
    t = best_ms(lambda: ops.conv(x, layer, out=out, relu=True, impl='wino4w'))
    print('probe conv_wino4w_kernel tpb {:>2s}      B={}: {:.3f} ms  ({:.0f} algorithmic TFLOP/s)  {:+.1f} % vs product'.format(
        tpb, B, t, flops / t / 1e9, (t / t_prod - 1) * 100), flush=True)
