#!/usr/bin/env python
"""Development: where the time of Bitcoding.decode_batch goes at batch B -- host side (cProfile) and wall clock per phase."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CALIBRATED = not (len(sys.argv) > 2 and sys.argv[2] == 'default')       # second argument `default`: the default-init checkpoint
cfg, sd, bp, bc, synthetic = bench.build_path('cr', 0, CALIBRATED)
if len(sys.argv) > 3:                                                    # third argument: rgb_window mode (auto / always / never)
    bc.rgb_window = sys.argv[3]
if len(sys.argv) > 4:                                                    # fourth argument: chunks per RGB channel
    bc.RGB_CHUNKS = int(sys.argv[4])
imgs = torch.stack([synthetic.make_image(512, 768, i, 'natural') for i in range(B)]).cuda()
enc = bc.encode_batch(imgs.float())
files = enc.to_bytes()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    dec, _ = bc.decode_batch(files)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('decode_batch B={}: host returned after {:.3f} s, GPU done after {:.3f} s'.format(B, t1 - t0, t2 - t0))
torch.cuda.synchronize()
time.sleep(0.3)            # a gap in the kernel trace: tools/decode_timeline.py cuts the LAST decode out at gaps > 100 ms
pr = cProfile.Profile()
pr.enable()
dec, _ = bc.decode_batch(files)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
