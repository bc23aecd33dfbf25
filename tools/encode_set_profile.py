#!/usr/bin/env python
"""Development: where the time of dataset_codec.encode_set goes on config 4 (N images of different sizes) -- wall time per pass and, under
`rocprofv3 --kernel-trace`, the kernels of the LAST pass (tools/trace_busy.py, tools/decode_timeline.py analyse the last burst of a trace)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l3c_pytorch_amd  # noqa: E402

l3c_pytorch_amd.configure_hip_queues()
import torch  # noqa: E402
import bench  # noqa: E402
from l3c_pytorch_amd.helpers import dataset_codec  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
sizes = dataset_codec.draw_sizes(N)
with bench.single_thread():
    imgs = {i: synthetic.make_image(sizes[i][0], sizes[i][1], i, 'natural') for i in range(N)}
order = list(range(N))
pix = sum(h * w for h, w in sizes) / 1e6
for rep in range(4):
    torch.cuda.synchronize()
    if rep == 3:
        time.sleep(0.4)      # a gap in the kernel trace before the pass that is analysed
    t0 = time.perf_counter()
    files, n_shapes, n_fwd = dataset_codec.encode_set(bc, imgs, order, max_batch=16)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('encode_set of {} images ({:.1f} MPix, {} shapes, {} forward passes): {:.3f} s = {:.1f} MPix/s'.format(N, pix, n_shapes, n_fwd, t1 - t0, pix / (t1 - t0)), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == 'cprofile':      # the host side of one more pass
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    dataset_codec.encode_set(bc, imgs, order, max_batch=16)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)
