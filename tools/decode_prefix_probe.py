#!/usr/bin/env python
"""Development: the front of a batch-128 decode_batch on the host's clock and on the GPU's -- when does the first kernel start, when are the
streams of the coarse scales on the device, when does the host return, when is the GPU done."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from l3c_pytorch_amd.bitcoding import bitcoding as bcm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
imgs = torch.stack([synthetic.make_image(512, 768, i, 'natural') for i in range(B)]).cuda()
files = bc.encode_batch(imgs.float()).to_bytes()
del imgs
torch.cuda.synchronize()
marks = {}
orig_upload, orig_parse = bcm._upload_streams, bcm.parse_containers


def upload(files_, parsed):
    marks['parsed (host)'] = time.perf_counter()
    r = orig_upload(files_, parsed)
    marks['part one staged + enqueued (host)'] = time.perf_counter()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks['ev part one'] = ev
    fin = r.pending

    def finish():
        marks['part two asked for (host)'] = time.perf_counter()
        fin()
        marks['part two staged + enqueued (host)'] = time.perf_counter()
    if fin is not None:
        r.pending = finish
    return r


bcm._upload_streams = upload
orig_get_P = bp.net.get_P


def get_P(scale, *a, **k):
    marks['get_P scale {} called (host)'.format(scale)] = time.perf_counter()
    return orig_get_P(scale, *a, **k)


bp.net.get_P = get_P
for rep in range(10):
    torch.cuda.synchronize()
    n_alloc = torch.cuda.memory_stats().get('num_device_alloc', 0)
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    t0 = time.perf_counter()
    dec, _ = bc.decode_batch(files)
    t1 = time.perf_counter()
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('decode_batch B={}: '.format(B) + ', '.join('{} {:.1f} ms'.format(k, (v - t0) * 1e3) for k, v in marks.items() if not k.startswith('ev')) +
          '; host returned {:.1f} ms, GPU done {:.1f} ms; GPU clock: coarse streams on the device {:.1f} ms after the call, end {:.1f} ms'.format(
              (t1 - t0) * 1e3, (t2 - t0) * 1e3, start.elapsed_time(marks['ev part one']), start.elapsed_time(end)) +
          '; device allocations in this call: {}'.format(torch.cuda.memory_stats().get('num_device_alloc', 0) - n_alloc), flush=True)
