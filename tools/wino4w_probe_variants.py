#!/usr/bin/env python
"""Development: time the F(4x4,3x3) PROBE kernel (csrc/conv_wino4w.hip) with parts removed (csrc/build.py --variant w4wpN "-DL3C_W4W_PROBE=N"
--xcheck, loaded through L3C_XCHECK_LIB).  Wrong results by construction; only the time means something."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
B, H, W = int(os.environ.get('W4_PROBE_B', '32')), 256, 384
w = torch.randn(64, 64, 3, 3, generator=g) / 24
b = torch.randn(64, generator=g)
layer = ops.PackedConv(w, b)
x = torch.randn(B, H, W, 64, generator=g).cuda()
out = torch.empty(B, H, W, 64, device='cuda')
res = []
for tpb in ('3', '6', '12'):
    os.environ['L3C_W4W_TPB'] = tpb
    fn = lambda: ops.conv(x, layer, out=out, relu=True, impl='wino4w')   # noqa: E731
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    res.append('tpb {} {:.3f} ms'.format(tpb, best))
print('{:28s} B={}  {}'.format(os.path.basename(os.environ.get('L3C_XCHECK_LIB', 'product build of the probe')), B, ' | '.join(res)), flush=True)
