#!/usr/bin/env python
"""Development probe (round-5 verdict, next 2, in the form it names): steady-state decode of several batches of 128 images with batch i + 1's
get_P convolutions under batch i's RGB chains -- the PHASED form of Bitcoding.decode_many (every batch a group of its own: lanes run the
convolutions, a stream pair per group parity runs the tables and chains), against one decode_batch after the other.

NOT in the product (measured: 143.9 against 142.4 MPix/s, profiles/r06_decode_phase_overlap_probe.log): the probe needs the two-slot form of
decode_many that profiles/r06_decode_phase_overlap.patch adds (`git apply profiles/r06_decode_phase_overlap.patch`).

usage: python tools/decode_phase_overlap_probe.py [B] [n_batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l3c_pytorch_amd  # noqa: E402

l3c_pytorch_amd.configure_hip_queues()
import torch  # noqa: E402
import bench  # noqa: E402

from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding  # noqa: E402

assert hasattr(Bitcoding, 'RAGGED_OVERLAP'), 'apply profiles/r06_decode_phase_overlap.patch first'

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


def run(label, overlap=None, per_group=1, **kw):
    if overlap is not None:
        bc.RAGGED_OVERLAP = overlap
    bc.RAGGED_GROUP_PIXELS = per_group * B * H * W
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
    print('{:66s} host {:.3f} s, done {}  -> {:.1f} MPix/s steady state  lossless={}  peak {:.0f} GB'.format(
        label, ts[-1][0], ' '.join('{:.3f}'.format(t) for _, t in ts), N * B * H * W / 1e6 / best, ok, peak), flush=True)


print('hardware queues:', os.environ.get('GPU_MAX_HW_QUEUES'), ' batches of', B, 'x', N)
run('one decode_batch after the other', lanes=1)
run('phased, one lane, one group after the other', overlap=False, ragged=True, lanes=1)
run('phased, one lane, group g+1 convolutions beside group g chains', overlap=True, ragged=True, lanes=1)
run('  the same, 2 lanes', overlap=True, ragged=True, lanes=2)
for n in (64, 128):
    run('  one lane, chains on {} CUs of their own'.format(n), overlap=True, ragged=True, lanes=1, chain_cus=n)
run('phased + overlapped, one lane, two batches per group', overlap=True, per_group=2, ragged=True, lanes=1)
