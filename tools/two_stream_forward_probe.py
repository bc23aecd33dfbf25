#!/usr/bin/env python
"""Development: the forward pass of a batch as ONE pass against TWO half-batches on two streams (kernel tails / prologues of one half
under the other half's launches).  python tools/two_stream_forward_probe.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
with bench.single_thread():
    imgs = torch.stack([synthetic.make_image(512, 768, i % 8, 'natural') for i in range(B)]).cuda().float()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
halves = [imgs[:B // 2].contiguous(), imgs[B // 2:].contiguous()]


def one():
    return bp.net(imgs)


def two():
    cur = torch.cuda.current_stream()
    outs = []
    for s, x in zip((s1, s2), halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(bp.net(x))
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return outs


for name, fn in (('one pass of {}'.format(B), one), ('two passes of {} on two streams'.format(B // 2), two), ('one pass again', one)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print('{}: {:.1f} ms per forward = {:.1f} MPix/s (forward only)'.format(name, dt * 1e3, B * 512 * 768 / 1e6 / dt), flush=True)
