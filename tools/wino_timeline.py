#!/usr/bin/env python
"""Development probe: per-wavefront s_memtime stamps of conv_wino_kernel.  Needs the timeline build of the library:

    python l3c-pytorch_amd/csrc/build.py --timeline
    L3C_LIB=l3c-pytorch_amd/csrc/libl3c_hip_timeline.so python tools/wino_timeline.py [--res] [--tpb N] [--B 32]

Prints how long the prologue of a block and the MFMA loop / output transform of its LAST tile take (shader cycles), and how
busy the matrix pipe would be if every loop ran 32 x 8 MFMAs of 64 cycles."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--res', action='store_true')
ap.add_argument('--tpb', type=int, default=0)
ap.add_argument('--B', type=int, default=32)
a = ap.parse_args()
B, H, W, C = a.B, 256, 384, 64
g = torch.Generator().manual_seed(0)
w = torch.randn(C, C, 3, 3, generator=g) / 24
layer = ops.PackedConv(w, torch.randn(C, generator=g))
x = torch.randn(B, H, W, C, generator=g).cuda()
res = torch.randn(B, H, W, C, generator=g).cuda() if a.res else None
lib = _lib.load()
lib.l3c_conv_wino_set_tiles_per_block(a.tpb)
for _ in range(3):
    ops.conv(x, layer, relu=res is None, residual=res)
torch.cuda.synchronize()
n_blocks = (H // 4) * (W // 32) * B                    # upper bound (tpb = 1)
dbg = torch.zeros(n_blocks * 4 * 16, dtype=torch.int64, device='cuda')
lib.l3c_conv_wino_set_debug.argtypes = [ctypes.c_void_p]
lib.l3c_conv_wino_set_debug(ctypes.c_void_p(dbg.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv(x, layer, relu=res is None, residual=res)
e1.record()
torch.cuda.synchronize()
lib.l3c_conv_wino_set_debug(None)
ms = e0.elapsed_time(e1)
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 10] != 0]
t = d[:, :11].astype(np.float64)
n_t = d[:, 14]
print('waves with stamps {}  tiles per block: {}  launch {:.3f} ms = {:.1f} algorithmic TFLOP/s (stamped build)'.format(
    len(d), np.unique(n_t), ms, 2.0 * B * H * W * C * C * 9 / ms / 1e9))
rows = [('P: decode + issue loads', t[:, 1] - t[:, 0]), ('P: wait + store patches', t[:, 2] - t[:, 1]),
        ('P: barrier', t[:, 3] - t[:, 2]), ('P: transform + barrier', t[:, 4] - t[:, 3]), ('prologue', t[:, 4] - t[:, 0]),
        ('last tile: MFMA loop', t[:, 6] - t[:, 5]), ('E: hand over + residual fetch', t[:, 7] - t[:, 6]),
        ('E: barrier A', t[:, 8] - t[:, 7]), ('E: sums in place + read back', t[:, 9] - t[:, 8]),
        ('E: (barrier B) bias, stores', t[:, 10] - t[:, 9]), ('epilogue', t[:, 10] - t[:, 6]),
        ('block total', t[:, 10] - t[:, 0])]
for name, v in rows:
    print('{:32s} mean {:9.0f}  p10 {:9.0f}  p50 {:9.0f}  p90 {:9.0f}'.format(name, v.mean(), *np.percentile(v, [10, 50, 90])))
tile = ((t[:, 10] - t[:, 4]) / n_t).mean()
print('per tile (loop + epilogue, averaged over the block): {:.0f} cycles for 16384 cycles of MFMA -> two wavefronts per SIMD '
      'keep the matrix pipe {:.1%} busy at most'.format(tile, min(1.0, 2 * 16384 / tile)))
