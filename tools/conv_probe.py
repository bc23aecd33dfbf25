#!/usr/bin/env python
"""Development probe: time one conv configuration of the L3C stack (default: 3x3 64->64 at 256x384, batch 16).

    python tools/conv_probe.py [--ks 3 --dil 1 --stride 1 --cin 64 --cout 64 --B 16 --H 256 --W 384 --iters 20]

--ablate N (probe bits 0x100..0x800 of `epilogue`: kernels with parts removed / the v1 kernel) needs the development library:
    python l3c-pytorch_amd/csrc/build.py --dev-probes && L3C_LIB=l3c-pytorch_amd/csrc/libl3c_hip_devprobes.so python tools/conv_probe.py --ablate 256
The product library rejects those bits.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
for k, v in dict(ks=3, dil=1, stride=1, cin=64, cout=64, B=16, H=256, W=384, iters=20).items():
    ap.add_argument('--' + k, type=int, default=v)
ap.add_argument('--relu', action='store_true')
ap.add_argument('--ablate', type=int, default=0)
ap.add_argument('--res', action='store_true')
a = ap.parse_args()
if a.ablate:
    _orig = _lib.ConvDesc.__setattr__
    import ctypes
    def _patched_call(name, *args, _c=_lib.call):
        if name == 'l3c_conv_mfma':
            args[0].epilogue |= a.ablate
        return _c(name, *args)
    ops.call = _patched_call
g = torch.Generator().manual_seed(0)
w = torch.randn(a.cout, a.cin, a.ks, a.ks, generator=g) / (a.cin * a.ks * a.ks) ** 0.5
b = torch.randn(a.cout, generator=g)
layer = ops.PackedConv(w, b, stride=a.stride, dilation=a.dil)
x = torch.randn(a.B, a.H, a.W, a.cin, generator=g).cuda()
res = torch.randn(a.B, a.H, a.W, a.cout, generator=g).cuda() if a.res else None
Ho, Wo = layer.out_hw(a.H, a.W)
out = torch.empty(a.B, Ho, Wo, a.cout, device='cuda')
for _ in range(3):
    ops.conv(x, layer, out=out, relu=a.relu, residual=res)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.conv(x, layer, out=out, relu=a.relu, residual=res)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
flops = 2.0 * a.B * Ho * Wo * a.cout * a.cin * a.ks * a.ks
print('conv k{} s{} d{} {}->{} B{} {}x{}: {:.3f} ms  {:.1f} TFLOP/s ({:.1f}% of 157.3)'.format(
    a.ks, a.stride, a.dil, a.cin, a.cout, a.B, a.H, a.W, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573))
