#!/usr/bin/env python
"""Development: how long does Event.synchronize() take for an event whose work finished long ago, while ANOTHER stream is busy?
(the set decoder's host waited ~1 s for the upload ring's events -- recorded after H2D copies that had completed within milliseconds --
whenever a long phase had been enqueued on other streams in between)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

a = torch.randn(8192, 8192, device='cuda')
pinned = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
busy, copy = torch.cuda.Stream(), torch.cuda.Stream()


def long_work(n=60):
    with torch.cuda.stream(busy):
        for _ in range(n):
            a @ a


for order in ('copy, event, THEN long work on another stream', 'long work on another stream, then copy + event'):
    for how in ('synchronize', 'query loop'):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if order.startswith('long'):
            long_work()
        with torch.cuda.stream(copy):
            dev = pinned.cuda(non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        if order.startswith('copy'):
            time.sleep(0.05)
            long_work()
        t1 = time.perf_counter()
        if how == 'synchronize':
            ev.synchronize()
        else:
            while not ev.query():
                time.sleep(0.0005)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print('{:52s} {:12s}: enqueued in {:.3f} s, event seen complete after {:.3f} s more, everything done after {:.3f} s'.format(
            order, how, t1 - t0, t2 - t1, t3 - t0), flush=True)
