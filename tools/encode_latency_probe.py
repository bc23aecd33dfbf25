#!/usr/bin/env python
"""Development probe: ONE 768x512 image, image in HBM -> `.l3c` bytes on the host (bench.py's `latency` leg, encode side only),
median of N repetitions; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (phase 1 of the coder =
ac_state_groups_kernel sets it)."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=9)
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--warm-batch', type=int, default=0, help='first code a batch of this size once (the streams a long-running process has created by then)')
a = ap.parse_args()
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
img = torch.stack([synthetic.make_image(bench.H, bench.W, i, 'natural') for i in range(a.batch)]).cuda().float().contiguous()
if a.warm_batch:
    big = torch.stack([synthetic.make_image(bench.H, bench.W, i, 'natural') for i in range(a.warm_batch)]).cuda().float().contiguous()
    bc.decode_batch(bc.encode_batch(big).to_bytes())
    del big
    torch.cuda.synchronize()
ts = []
for k in range(a.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = bp.net(img)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    enc = bc.encode_batch(img, out=out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    data = enc.to_bytes()
    t3 = time.perf_counter()
    if k:
        ts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
tn = []
for k in range(a.reps + 1):   # the same without the synchronisations between the stages (bench.py's latency leg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    data = bc.encode_batch(img, out=bp.net(img)).to_bytes()
    torch.cuda.synchronize()
    if k:
        tn.append(time.perf_counter() - t0)
med = [statistics.median(x) * 1e3 for x in zip(*ts)]
print('batch {}: forward {:.2f} ms | coder (intervals + phase 1 + phase 2) {:.2f} ms | to_bytes {:.2f} ms | total {:.2f} ms (serialised by the '
      'synchronisations between the stages; {:.2f} ms without them) | {} bytes'.format(a.batch, med[0], med[1], med[2], med[3],
                                                                                        statistics.median(tn) * 1e3, sum(map(len, data))))
td = []
for k in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec, _ = bc.decode_batch(data)
    torch.cuda.synchronize()
    td.append(time.perf_counter() - t0)
print('decode {:.2f} ms (median of 3 after a warm-up) | lossless {}'.format(statistics.median(td[1:]) * 1e3,
                                                                            bool(torch.equal(dec.to(torch.uint8), img.to(torch.uint8)))))
