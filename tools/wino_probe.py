#!/usr/bin/env python
"""Development probe: Winograd F(2x2,3x3) conv (csrc/conv_wino.hip) vs the implicit-GEMM conv -- accuracy and time."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops, _lib  # noqa: E402
from l3c_pytorch_amd._lib import call, ptr, stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--res', action='store_true', help='time with the residual epilogue')
ap.add_argument('--no-check', action='store_true')
for k, v in dict(cin=64, cout=64, B=32, H=256, W=384, iters=10).items():
    ap.add_argument('--' + k, type=int, default=v)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)


def wino(x, w_packed, bias, Cout, relu=False, residual=None, shuffle=False):
    B, H, W, C = x.shape
    out = (torch.empty(B, 2 * H, 2 * W, Cout // 4, device='cuda') if shuffle else torch.empty(B, H, W, Cout, device='cuda'))
    d = _lib.ConvDesc()
    d.inp, d.in_cstride, d.in_coff = ptr(x), C, 0
    d.packed_w, d.bias = ptr(w_packed), ptr(bias)
    d.residual = ptr(residual) if residual is not None else None
    d.res_cstride = residual.shape[-1] if residual is not None else 0
    d.out, d.out_cstride, d.out_coff = ptr(out), out.shape[-1], 0
    d.B, d.Hin, d.Win, d.Cin, d.Cout = B, H, W, C, Cout
    d.KS, d.stride, d.dilation = 3, 1, 1
    d.epilogue = (_lib.EPI_RELU if relu else 0) | (_lib.EPI_RESIDUAL if residual is not None else 0) | (_lib.EPI_PIXEL_SHUFFLE if shuffle else 0)
    call('l3c_conv_wino', d, stream())
    return out


def pack(w):
    Cout, Cin = w.shape[:2]
    p = torch.empty(_lib.load().l3c_conv_wino_packed_words(Cout, Cin), device='cuda')
    call('l3c_conv_wino_pack_weights', ptr(w.cuda().contiguous()), Cout, Cin, ptr(p), stream())
    return p


for (B, H, W, Cout, relu, res, shuffle) in [] if a.no_check else [(2, 16, 32, 64, False, False, False), (1, 9, 33, 64, True, False, False),
                                            (3, 50, 70, 64, False, True, False), (2, 24, 40, 256, False, False, True),
                                            (1, 64, 96, 120, True, False, False)]:
    w = torch.randn(Cout, 64, 3, 3, generator=g) / 24
    b = torch.randn(Cout, generator=g)
    layer = ops.PackedConv(w, b)
    x = torch.randn(B, H, W, 64, generator=g).cuda()
    r = torch.randn(B, H, W, Cout, generator=g).cuda() if res else None
    ref = ops.conv(x, layer, relu=relu, residual=r, pixel_shuffle=shuffle)
    got = wino(x, pack(w), b.cuda(), Cout, relu, r, shuffle)
    ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double(), b.double(), padding=1)
    if relu:
        ref64 = ref64.clamp(min=0)
    if res:
        ref64 = ref64 + r.permute(0, 3, 1, 2).double().cpu()
    if shuffle:
        ref64 = torch.nn.functional.pixel_shuffle(ref64, 2)
    ref64 = ref64.permute(0, 2, 3, 1)
    e_w = (got.double().cpu() - ref64).abs().max().item()
    e_d = (ref.double().cpu() - ref64).abs().max().item()
    print((B, H, W, Cout, relu, res, shuffle), 'max |wino - mfma| %.2e' % (got - ref).abs().max().item(),
          ' vs fp64: wino %.2e  direct %.2e' % (e_w, e_d))

w = torch.randn(a.cout, a.cin, 3, 3, generator=g) / (a.cin * 9) ** 0.5
b = torch.randn(a.cout, generator=g)
layer = ops.PackedConv(w, b)
wp = pack(w)
x = torch.randn(a.B, a.H, a.W, a.cin, generator=g).cuda()
bc = b.cuda()
rt = torch.randn(a.B, a.H, a.W, a.cout, generator=g).cuda() if a.res else None
for name, fn in [('mfma', lambda: ops.conv(x, layer, relu=True)), ('wino', lambda: wino(x, wp, bc, a.cout, relu=True, residual=rt))]:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    flops = 2.0 * a.B * a.H * a.W * a.cout * a.cin * 9
    print('{} 3x3 {}->{} B{} {}x{}: {:.3f} ms  {:.1f} algorithmic TFLOP/s'.format(name, a.cin, a.cout, a.B, a.H, a.W, ms, flops / ms / 1e9))
