#!/usr/bin/env python
"""Development probe: how does the decode of SMALL batches scale with the lanes of Bitcoding.decode_many?  (a) the same one-image file 32
times (one shape: no new allocations after the first pass), (b) 32 images of different sizes, one file per batch -- with the allocator's
device-malloc count and the host time beside the wall time.

usage: GPU_MAX_HW_QUEUES=8 python tools/decode_concurrency_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l3c_pytorch_amd  # noqa: E402

l3c_pytorch_amd.configure_hip_queues()
import torch  # noqa: E402
import bench  # noqa: E402
from l3c_pytorch_amd.helpers import dataset_codec, pad  # noqa: E402

cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
N = 32
sizes = [s for s in dataset_codec.draw_sizes(400) if s not in ((512, 768), (768, 512), (576, 768), (512, 683))][:N]
imgs = [synthetic.make_image(h, w, i, 'natural') for i, (h, w) in enumerate(sizes)]
files_het = []
for im in imgs:
    x, pt = pad.pad(im.unsqueeze(0), 8, mode='constant')
    files_het.append(bc.encode_batch(x.cuda()).to_bytes([pt if isinstance(pt, tuple) else (0, 0, 0, 0)])[0])
one = files_het[0]
torch.cuda.synchronize()
pix_het = sum(h * w for h, w in sizes) / 1e6
pix_one = sizes[0][0] * sizes[0][1] * N / 1e6


def run(label, batches, mpix, lanes):
    out = []
    for rep in range(3):
        st0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bc.decode_many(batches, on_batch=lambda i, p, pd: None, lanes=lanes, out_dtype=torch.uint8)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append('{:.3f}/{:.3f} s ({} mallocs)'.format(t1 - t0, t2 - t0, torch.cuda.memory_stats().get('num_device_alloc', 0) - st0))
    print('{:34s} lanes {:2d}: host/done {}  -> {:.1f} MPix/s'.format(label, lanes, '  '.join(out), mpix / (t2 - t0)), flush=True)


print('hardware queues:', os.environ.get('GPU_MAX_HW_QUEUES'))
for lanes in (1, 2, 4, 8, 16):
    run('one image, 32 times', [[one]] * N, pix_one, lanes)
for lanes in (1, 2, 4, 8, 16):
    run('32 images of different sizes', [[f] for f in files_het], pix_het, lanes)
