#!/usr/bin/env python
"""Config-4 style run: a heterogeneous image set (sizes drawn like the reference's Open Images preprocessing: random
downscale, short side >= 512; import_train_images.py:150-164), sharded round-robin over the ranks (replicas only), every
image encoded END TO END: host uint8 image -> H2D -> pad -> forward -> fused heads -> range coder -> D2H -> `.l3c` bytes in
host memory.  Images of equal padded shape share a batch; ONE grouped range-coder launch (l3c_ac_encode_groups) then codes every
stream of every batch concurrently.

    python tools/bench_dataset.py [--images 64] [--max-batch 16]
    python -m torch.distributed.run --nproc-per-node N tools/bench_dataset.py ...     (one process per GPU, RCCL for the stats)
"""
import argparse
import collections
import os
import sys
import time

# Small, differently shaped batches only fill the machine when several of them run side by side: give the HIP runtime 8
# hardware queues (default 4) so that Bitcoding.encode_many's forward streams and the coder's side stream do not alias.
# (bench.py's large batches are 2 % faster with the default, so this is a per-tool setting.)  Must precede HIP start-up.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import l3c_pytorch_amd  # noqa: E402,F401
from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding  # noqa: E402
from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint  # noqa: E402
from l3c_pytorch_amd.helpers import config_parser, pad, sharding, synthetic  # noqa: E402


def draw_sizes(n, seed=0):
    """(H, W) per image: a 'camera' size in landscape or portrait, downscaled by a random factor so the short side ends up
    in [512, 1024]; a few popular sizes repeat (as in real sets), the rest are unique."""
    rng = np.random.RandomState(seed)
    sizes = []
    for i in range(n):
        if rng.rand() < 0.5:
            h, w = [(512, 768), (768, 512), (576, 768), (512, 683)][rng.randint(4)]
        else:
            short = int(rng.randint(512, 1025))
            aspect = float(rng.choice([4 / 3, 3 / 2, 16 / 9, 1.0]))
            h, w = (short, int(round(short * aspect)))
            if rng.rand() < 0.3:
                h, w = w, h
        sizes.append((h, w))
    return sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', dest='n', type=int, default=64)
    ap.add_argument('--max-batch', type=int, default=16)
    ap.add_argument('--per-batch-coder', action='store_true', help='one coder launch per batch instead of one grouped launch')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', rank=rank, world_size=world)
    cfg = config_parser.parse_builtin('ms', 'cr')
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(synthetic.make_state_dict(cfg, 0), strict=True)
    bp.set_eval()
    bc = Bitcoding(bp)
    sizes = draw_sizes(a.n)
    mine = sharding.shard_indices(a.n, rank, world)
    imgs = {i: synthetic.make_image(sizes[i][0], sizes[i][1], i, 'natural') for i in mine}     # host uint8
    # warm-up (kernel load, allocator) on one image
    bc.encode_batch(pad.pad(imgs[mine[0]].unsqueeze(0), 8, mode='constant')[0].cuda()).to_bytes()
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    marks = {}
    groups = collections.defaultdict(list)
    padded, pads = {}, {}
    for i in mine:
        x, pt = pad.pad(imgs[i].unsqueeze(0), 8, mode='constant')
        padded[i], pads[i] = x, (pt if isinstance(pt, tuple) else (0, 0, 0, 0))
        groups[tuple(x.shape[-2:])].append(i)
    marks['pad + group (host)'] = time.perf_counter()
    chunks, batches = [], []
    for shape, idxs in groups.items():
        for k in range(0, len(idxs), a.max_batch):
            chunks.append(idxs[k:k + a.max_batch])
            batches.append(torch.cat([padded[i] for i in chunks[-1]]).cuda(non_blocking=True))
    marks['H2D enqueue'] = time.perf_counter()
    if a.per_batch_coder:
        pending = [(c, bc.encode_batch(b)) for c, b in zip(chunks, batches)]    # one coder launch pair per batch
    else:
        pending = list(zip(chunks, bc.encode_many(batches)))                    # ONE grouped coder launch for the set
    marks['forward + coder enqueue'] = time.perf_counter()
    torch.cuda.synchronize()
    marks['GPU drain'] = time.perf_counter()
    files = {}
    from l3c_pytorch_amd.bitcoding.bitcoding import EncodedBatch
    all_files = EncodedBatch.many_to_bytes([enc for _, enc in pending], [[pads[i] for i in chunk] for chunk, _ in pending])
    for (chunk, _), fs in zip(pending, all_files):                  # ... and is collected at the end: one sync, one D2H
        for i, f in zip(chunk, fs):
            files[i] = f
    torch.cuda.synchronize()
    marks['file assembly + D2H'] = time.perf_counter()
    dt = time.perf_counter() - t0

    pixels = sum(sizes[i][0] * sizes[i][1] for i in mine)
    bits = sum(len(files[i]) for i in mine) * 8
    stats = sharding.combine_stats(sharding.gather_stats({'pixels': pixels, 'subpixels': 3 * pixels, 'bits': float(bits),
                                                          'seconds': dt}))
    if rank == 0:
        print('{} images, {} distinct padded shapes on rank 0, {} launches'.format(a.n, len(groups), len(pending)))
        print('end-to-end encode (host image -> .l3c bytes on the host): {:.2f} MPix/s aggregate over {} rank(s), {:.3f} bpsp, '
              '{:.2f} s'.format(stats['mpix_per_s'], stats['ranks'], stats['bpsp'], stats['seconds']))
    if rank == 0:
        prev = t0
        for k, v in marks.items():
            print('    {:28s} {:7.1f} ms'.format(k, (v - prev) * 1e3))
            prev = v
    # spot-check: decode two files
    for i in mine[:2]:
        dec, padding = bc.decode_batch([files[i]])
        out = pad.undo_pad(dec, *padding[0]) if any(padding[0]) else dec
        assert torch.equal(out.cpu()[0], imgs[i].long()), 'round trip failed for image {}'.format(i)
    if rank == 0:
        print('round trip of 2 images: lossless')


if __name__ == '__main__':
    main()
