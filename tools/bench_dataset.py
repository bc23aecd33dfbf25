#!/usr/bin/env python
"""Config-4 style run with a per-stage host time line (development view of `bench.py --config dataset`, which is the
measured form): a heterogeneous image set, every image encoded END TO END: host uint8 image -> H2D -> pad -> forward ->
fused heads -> range coder -> D2H -> `.l3c` bytes in host memory (l3c-pytorch_amd/helpers/dataset_codec.py).

    python tools/bench_dataset.py [--images 64] [--max-batch 16]
"""
import argparse
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # see Bitcoding.encode_many; must precede HIP start-up

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding  # noqa: E402
from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint  # noqa: E402
from l3c_pytorch_amd.helpers import config_parser, dataset_codec, pad, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', dest='n', type=int, default=64)
    ap.add_argument('--max-batch', type=int, default=16)
    ap.add_argument('--repeat', type=int, default=2, help='timed passes over the set (the first one also warms the allocator)')
    ap.add_argument('--groups', type=int, default=0, help='equal coder groups per set (0: the halving groups of encode_set)')
    ap.add_argument('--no-probe', action='store_true')
    a = ap.parse_args()
    cfg = config_parser.parse_builtin('ms', 'cr')
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(synthetic.make_state_dict(cfg, 0, calibrated=True), strict=True)
    bp.set_eval()
    bc = Bitcoding(bp)
    sizes = dataset_codec.draw_sizes(a.n)
    order = list(range(a.n))
    imgs = {i: synthetic.make_image(sizes[i][0], sizes[i][1], i, 'natural') for i in order}     # host uint8
    bc.encode_batch(pad.pad(imgs[0].unsqueeze(0), 8, mode='constant')[0].cuda()).to_bytes()      # warm-up
    torch.cuda.synchronize()
    # host cost of ENQUEUEING one forward + heads vs the GPU's time for it, at the set's typical batch (2 images of 768x512)
    for nb in (() if a.no_probe else (1, 2, 4, 16)):
        x = torch.cat([pad.pad(imgs[0].unsqueeze(0), 8, mode='constant')[0]] * nb)[:, :, :512, :768].contiguous().cuda()
        bc.prepare_batch(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            bc.prepare_batch(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('forward + heads, batch {:2d} of {}x{}: host enqueue {:.2f} ms, until the GPU is done {:.2f} ms per pass ({:.1f} MPix/s)'.format(
            nb, x.shape[2], x.shape[3], (t1 - t0) * 100, (t2 - t0) * 100, nb * x.shape[2] * x.shape[3] * 10 / 1e6 / (t2 - t0)))
    for _ in range(a.repeat):
        marks = {}
        t0 = time.perf_counter()
        files, n_shapes, n_launches = dataset_codec.encode_set(bc, imgs, order, max_batch=a.max_batch, marks=marks, **({'n_groups': a.groups} if a.groups else {}))
        dt = time.perf_counter() - t0
        report(a, sizes, files, n_shapes, n_launches, dt, t0, marks)


def report(a, sizes, files, n_shapes, n_launches, dt, t0, marks):
    pixels = sum(h * w for h, w in sizes)
    print('{} images, {} distinct padded shapes, {} forward passes: {:.2f} MPix/s end to end, {:.3f} bpsp, {:.2f} s'.format(
        a.n, n_shapes, n_launches, pixels / 1e6 / dt, sum(len(f) for f in files.values()) * 8 / (3 * pixels), dt))
    prev = t0
    for k, v in marks.pop('host seconds', {}).items():
        print('    host: {:50s} {:7.1f} ms'.format(k, v * 1e3))
    for k, v in marks.items():
        print('    {:28s} {:7.1f} ms'.format(k, (v - prev) * 1e3))
        prev = v


if __name__ == '__main__':
    main()
