"""Coding a SET of differently sized images (BASELINE.json config 4: the reference's test.py over a folder of Open Images
crops, multiscale_tester.py:236-381, one image after the other).  Images of equal padded shape share a forward pass, every
stream of every batch goes through grouped range-coder launches (Bitcoding.encode_many), the files are assembled on the
device and cross PCIe in ONE copy (EncodedBatch.many_to_bytes)."""
import collections

import numpy as np
import torch

from . import pad


def draw_sizes(n, seed=0):
    """(H, W) per image, drawn like the reference's Open Images preprocessing (import_train_images.py:150-164: random
    downscale, short side >= 512): a 'camera' size in landscape or portrait whose short side ends up in [512, 1024]; a few
    popular sizes repeat (as in real sets), the rest are unique."""
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


def encode_set(bc, imgs, order, max_batch=16, fac=8, marks=None):
    """imgs: {index: uint8 (3,H,W) HOST tensor}; order: the indices to code.  -> ({index: `.l3c` bytes}, number of distinct
    padded shapes, number of forward passes).  `marks` (dict) receives host time stamps of the stages."""
    import time
    from ..bitcoding.bitcoding import EncodedBatch

    def mark(name):
        if marks is not None:
            marks[name] = time.perf_counter()

    groups = collections.defaultdict(list)
    padded, pads = {}, {}
    for i in order:
        x, pt = pad.pad(imgs[i].unsqueeze(0), fac, mode='constant')
        padded[i], pads[i] = x, (pt if isinstance(pt, tuple) else (0, 0, 0, 0))
        groups[tuple(x.shape[-2:])].append(i)
    mark('pad + group (host)')
    chunks, batches = [], []
    for shape, idxs in groups.items():
        for k in range(0, len(idxs), max_batch):
            chunks.append(idxs[k:k + max_batch])
            batches.append(torch.cat([padded[i] for i in chunks[-1]]).cuda(non_blocking=True))
    mark('H2D enqueue')
    encs = bc.encode_many(batches)                       # grouped coder launches for the whole set
    mark('forward + coder enqueue')
    all_files = EncodedBatch.many_to_bytes(encs, [[pads[i] for i in chunk] for chunk in chunks])   # one sync, one D2H
    mark('file assembly + D2H')
    files = {}
    for chunk, fs in zip(chunks, all_files):
        for i, f in zip(chunk, fs):
            files[i] = f
    return files, len(groups), len(batches)
