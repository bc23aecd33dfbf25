#!/usr/bin/env python
"""Development probe: per-wavefront s_memtime stamps of conv_wino_kernel -- needs csrc/conv_wino.hip compiled with
-DL3C_WINO_TIMELINE (HIPCC_EXTRA=-DL3C_WINO_TIMELINE python l3c-pytorch_amd/csrc/build.py --force); prints how long the
prologue, the MFMA loop and the epilogue of a wavefront take and how the two wavefronts of a SIMD overlap."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops, _lib  # noqa: E402

B, H, W, C = 32, 256, 384, 64
g = torch.Generator().manual_seed(0)
w = torch.randn(C, C, 3, 3, generator=g) / 24
layer = ops.PackedConv(w, torch.randn(C, generator=g))
x = torch.randn(B, H, W, C, generator=g).cuda()
res = torch.randn(B, H, W, C, generator=g).cuda() if '--res' in sys.argv else None
for _ in range(3):
    ops.conv(x, layer, relu=res is None, residual=res)
torch.cuda.synchronize()
n_blocks = (H // 4) * (W // 32) * B
dbg = torch.zeros(n_blocks * 4 * 12, dtype=torch.int64, device='cuda')
lib = _lib.load()
lib.l3c_conv_wino_set_debug.argtypes = [ctypes.c_void_p]
lib.l3c_conv_wino_set_debug(ctypes.c_void_p(dbg.data_ptr()))
ops.conv(x, layer, relu=res is None, residual=res)
torch.cuda.synchronize()
lib.l3c_conv_wino_set_debug(None)
d = dbg.cpu().numpy().reshape(-1, 12)
d = d[d[:, 5] != 0]
hw, blk, t0, t1, t2, t3, p1, p2, p3, e1, e2, _ = d.T
print('waves with stamps', len(d), 'of', n_blocks * 4)
for name, v in (('P: decode+issue loads', p1 - t0), ('P: wait+store patches', p2 - p1), ('P: barrier 1', p3 - p2), ('P: transform+barrier 2', t1 - p3), ('E: partial sums + X write', e1 - t2), ('E: barrier', e2 - e1), ('E: X read, final, stores', t3 - e2), ('prologue', t1 - t0), ('loop', t2 - t1), ('epilogue', t3 - t2), ('total', t3 - t0)):
    print('{:26s} mean {:9.0f}  p10 {:9.0f}  p50 {:9.0f}  p90 {:9.0f} ticks'.format(name, v.mean(), *np.percentile(v, [10, 50, 90])))
key = hw >> 4                      # (xcc, se, sh, cu, pipe, simd): everything but the wave slot
slot = hw & 0xf
print('wave slots used:', np.unique(slot), ' SIMDs:', len(np.unique(key)))
tot = dict(span=0, loop1=0, loop2=0, idle_slot=0, gaps=[])
for k in np.unique(key):
    m = key == k
    sp0, sp1 = t0[m].min(), t3[m].max()
    ev = []
    for a, b_ in zip(t1[m], t2[m]):
        ev.append((a, 1)); ev.append((b_, -1))
    ev.sort()
    depth, last = 0, ev[0][0]
    for t, dlt in ev:
        if depth >= 1: tot['loop1'] += t - last
        if depth >= 2: tot['loop2'] += t - last
        depth += dlt; last = t
    tot['span'] += sp1 - sp0
    for sl in (0, 1):
        ms = m & (slot == sl)
        o = np.argsort(t0[ms])
        a, b_ = t0[ms][o], t3[ms][o]
        tot['gaps'] += list(a[1:] - b_[:-1])
g = np.array(tot['gaps'])
print('per SIMD: some wave in its MFMA loop {:.1%} of the time, two at once {:.1%}'.format(tot['loop1'] / tot['span'], tot['loop2'] / tot['span']))
print('slot turnover (end of a wave -> start of the next in the same slot): mean {:.0f}  p10 {:.0f}  p50 {:.0f}  p90 {:.0f} ticks'.format(
    g.mean(), *np.percentile(g, [10, 50, 90])))
