#!/usr/bin/env python
"""Development probe: does Bitcoding.decode_batch block the host when the GPU is still busy with the previous call?  Eight one-image decodes back
to back without synchronising, host seconds per call; then the same under cProfile."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
img = synthetic.make_image(512, 768, 0, 'natural').unsqueeze(0).cuda()
f = bc.encode_batch(img.float()).to_bytes()
for rep in range(2):
    torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for k in range(8):
        t0 = time.perf_counter()
        bc.decode_batch(f, out_dtype=torch.uint8)
        ts.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('host seconds per call:', ' '.join('{:.4f}'.format(t) for t in ts), ' all enqueued after {:.3f} s, done after {:.3f} s'.format(t1 - t00, time.perf_counter() - t00), flush=True)
pr = cProfile.Profile()
pr.enable()
for k in range(8):
    bc.decode_batch(f, out_dtype=torch.uint8)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
