#!/usr/bin/env python
"""Development probe (round-5 verdict, next 2): steady-state decode of several batches of 128 images -- one lane (batches back to back)
against two lanes (batch i + 1's get_P convolutions beside batch i's RGB chains), with and without the chains confined to compute units
of their own.

usage: python tools/decode_lanes_probe.py [B] [n_batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l3c_pytorch_amd  # noqa: E402

l3c_pytorch_amd.configure_hip_queues()
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, True)
H, W = 512, 768
imgs = torch.stack([synthetic.make_image(H, W, i, 'natural') for i in range(B)]).cuda()
files = bc.encode_batch(imgs.float()).to_bytes()
torch.cuda.synchronize()
want = imgs.to(torch.uint8)
del imgs
torch.cuda.empty_cache()      # the encode's 88 GB of cached blocks belong to the default stream's pool: the lanes could not use them


def run(label, **kw):
    ts, ok = [], True
    for rep in range(3):
        got = {}

        def on_batch(i, pixels, padding):
            got[i] = pixels        # (kept on the device for the check; uint8)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bc.decode_many([files] * N, on_batch=on_batch, out_dtype=torch.uint8, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
        ok = ok and all(bool(torch.equal(got[i], want)) for i in got) and len(got) == N
        del got
    best = min(t for _, t in ts)
    peak = torch.cuda.max_memory_allocated() / 1e9
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    print('{:58s} host {:.3f} s, done {}  -> {:.1f} MPix/s steady state  lossless={}  peak {:.0f} GB'.format(
        label, ts[-1][0], ' '.join('{:.3f}'.format(t) for _, t in ts), N * B * H * W / 1e6 / best, ok, peak), flush=True)


print('hardware queues:', os.environ.get('GPU_MAX_HW_QUEUES'), ' batches of', B, 'x', N)
run('one lane (decode_batch after decode_batch)', lanes=1)
run('two lanes', lanes=2)
run('three lanes', lanes=3)
for n in (64, 128):
    run('two lanes, chains on {} CUs of their own'.format(n), lanes=2, chain_cus=n)
run('one lane again', lanes=1)
