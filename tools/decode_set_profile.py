#!/usr/bin/env python
"""Development: where the time of dataset_codec.decode_set goes on config 4's 200-image leg -- host seconds until everything is enqueued against
the wall time, and the host side under cProfile."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l3c_pytorch_amd  # noqa: E402

l3c_pytorch_amd.configure_hip_queues()
import torch  # noqa: E402
import bench  # noqa: E402
from l3c_pytorch_amd.helpers import dataset_codec  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
sizes = dataset_codec.draw_sizes(N)
imgs = {i: synthetic.make_image(sizes[i][0], sizes[i][1], i, 'natural') for i in range(N)}
order = list(range(N))
files, _, _ = dataset_codec.encode_set(bc, imgs, order, max_batch=16)
pix = sum(h * w for h, w in sizes) / 1e6
for rep in range(3):
    torch.cuda.synchronize()
    marks = {}
    t0 = time.perf_counter()
    back = dataset_codec.decode_set(bc, files, order, max_batch=16, marks=marks)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('decode_set of {} images: returned after {:.3f} s ({:.1f} MPix/s); plan {:.3f} s'.format(N, t1 - t0, pix / (t1 - t0), marks['plan (host)'] - t0), flush=True)
torch.cuda.synchronize()
time.sleep(0.4)      # a gap in the kernel trace: tools/trace_busy.py analyses the LAST burst
pr = cProfile.Profile()
pr.enable()
back = dataset_codec.decode_set(bc, files, order, max_batch=16)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
