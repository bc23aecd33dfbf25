#!/usr/bin/env python
"""Development: per-wavefront time line of conv_wino4_kernel from s_memtime stamps (csrc/build.py --variant w4tl "-DL3C_W4_TIMELINE",
loaded through L3C_LIB): cycles a wavefront spends in the prologue, per tile in the chunk loop (8 chunks x 72 MFMAs = 18 432 cycles of
matrix-pipe time at one wavefront per SIMD) and in the output transform.

    L3C_LIB=l3c-pytorch_amd/csrc/libl3c_hip_w4tl.so python tools/wino4_timeline.py [--res] [--B 32]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=32)
ap.add_argument('--H', type=int, default=256)
ap.add_argument('--W', type=int, default=384)
ap.add_argument('--res', action='store_true')
a = ap.parse_args()
lib = _lib.load()
g = torch.Generator().manual_seed(0)
layer = ops.PackedConv(torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g))
x = torch.randn(a.B, a.H, a.W, 64, generator=g).cuda()
r = torch.randn(a.B, a.H, a.W, 64, generator=g).cuda() if a.res else None
kw = dict(residual=r) if a.res else dict(relu=True)
for _ in range(3):
    ops.conv(x, layer, impl='wino4', **kw)
torch.cuda.synchronize()
dbg = torch.zeros(8192 * 4 * 32, dtype=torch.int64, device='cuda')
lib.l3c_conv_wino4_set_debug.argtypes = [ctypes.c_void_p]
lib.l3c_conv_wino4_set_debug(ctypes.c_void_p(dbg.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv(x, layer, impl='wino4', **kw)
e1.record()
torch.cuda.synchronize()
lib.l3c_conv_wino4_set_debug(None)
d = dbg.cpu().numpy().reshape(8192, 4, 32)
d = d[d[:, 0, 15] > 0]
nt = d[:, :, 15]
pro = (d[:, :, 1] - d[:, :, 0]).astype(np.float64)
loops, epis = [], []
for t in range(6):
    m = nt > t
    start = d[:, :, 1] if t == 0 else d[:, :, 3 + 2 * (t - 1)]
    loops.append((d[:, :, 2 + 2 * t] - start)[m].astype(np.float64))
    epis.append((d[:, :, 3 + 2 * t] - d[:, :, 2 + 2 * t])[m].astype(np.float64))
loop, epi = np.concatenate(loops), np.concatenate(epis)
life = (np.take_along_axis(d, (1 + 2 * nt)[:, :, None].astype(np.int64), 2)[:, :, 0] - d[:, :, 0]).astype(np.float64)
print('kernel {:.3f} ms, {} blocks recorded, tiles per block {:.2f}'.format(e0.elapsed_time(e1), len(d), nt.mean()))
print('cycles per wavefront: prologue {:.0f} | per tile: loop {:.0f} (p10 {:.0f}, p90 {:.0f}) + output transform {:.0f} (p10 {:.0f}, p90 {:.0f})'.format(
    pro.mean(), loop.mean(), np.percentile(loop, 10), np.percentile(loop, 90), epi.mean(), np.percentile(epi, 10), np.percentile(epi, 90)))
print('first tile of a block: loop {:.0f}, later tiles {:.0f}'.format(loops[0].mean(), np.concatenate(loops[1:]).mean() if len(loops) > 1 and len(loops[1]) else float('nan')))
print('matrix-pipe time per tile and wavefront 18432 cycles: share of the wavefront\'s life {:.3f}'.format((18432.0 * nt).sum() / life.sum()))
# inside the output transform of tile 0 (stamps 16..23): per tile column r, cycles from the previous stamp to "window written" (both
# transform stages of the column: ~120 VALU instructions + 16 LDS writes) and on to "previous round stored, this round's reads issued"
prev = d[:, :, 2]
parts = []
for r in range(4):
    a, b = d[:, :, 16 + 2 * r], d[:, :, 17 + 2 * r]
    parts.append('r{}: transform {:.0f} + stores/reads {:.0f}'.format(r, (a - prev).astype(np.float64).mean(), (b - a).astype(np.float64).mean()))
    prev = b
parts.append('last round out {:.0f}'.format((d[:, :, 3] - prev).astype(np.float64).mean()))
print('output transform of tile 0: ' + ' | '.join(parts))
# inside the chunk loops of tile 0 (first of the block, right behind the prologue) and tile 1: cycles for chunk 0, chunks 1-2, 3-4, 5-6, 7
for t in (0, 1):
    m = nt[:, 0] > t
    start = (d[:, :, 1] if t == 0 else d[:, :, 3])[m]
    marks = [d[:, :, 24 + 4 * t + k][m] for k in range(4)] + [d[:, :, 2 + 2 * t][m]]
    seg, prev = [], start
    for x in marks:
        seg.append((x - prev).astype(np.float64).mean())
        prev = x
    print('tile {} loop: chunk 0 {:.0f} | chunks 1-2 {:.0f} | 3-4 {:.0f} | 5-6 {:.0f} | chunk 7 {:.0f}'.format(t, *seg))
