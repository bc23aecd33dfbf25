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


class _PinnedRing(object):
    """Page-locked staging buffers for the H2D copies, used round robin; a buffer is reused only after the copy that read it
    has completed (its event).  Kept on the Bitcoding object: page-locking costs far more than the copies."""

    def __init__(self, n=8):
        self.bufs, self.events, self.turn = [None] * n, [None] * n, 0

    def take(self, nbytes):
        k = self.turn = (self.turn + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
            self.events[k] = None
        if self.bufs[k] is None or self.bufs[k].numel() < nbytes:
            self.bufs[k] = torch.empty(max(nbytes, 16 << 20), dtype=torch.uint8, pin_memory=True)
        return k, self.bufs[k][:nbytes]

    def sent(self, k):
        self.events[k] = torch.cuda.Event()
        self.events[k].record(torch.cuda.current_stream())


def plan_set(shapes, order, max_batch, fac):
    """Host-side plan (no pixel is touched): images of equal PADDED shape share a forward pass of at most max_batch images.
    shapes: {index: (h, w)}.  -> (chunks: list of index lists, padded shape per chunk, {index: (left, right, top, bottom)},
    number of distinct padded shapes)."""
    groups = collections.defaultdict(list)
    pads = {}
    for i in order:
        h, w = shapes[i]
        pads[i] = pad.padding_for(h, w, fac)
        groups[(h + pads[i][2] + pads[i][3], w + pads[i][0] + pads[i][1])].append(i)
    chunks, padded = [], []
    for shape, idxs in groups.items():
        for k in range(0, len(idxs), max_batch):
            chunks.append(idxs[k:k + max_batch])
            padded.append(shape)
    return chunks, padded, pads, len(groups)


def encode_set(bc, imgs, order, max_batch=16, fac=8, marks=None, n_groups=None, n_pinned=8, shapes=None, loader=None, pool=None):
    """imgs: {index: uint8 (3,H,W) HOST tensor}; order: the indices to code.  LAZY form (files on disk, test.py --write_to_files): `shapes` =
    {index: (H, W)} known up front (an image file's header), `loader(index)` -> the tensor, run on the worker threads of `pool` in the
    order the passes will be enqueued, so that reading and decoding the image files overlaps the GPU's work on the earlier passes; `imgs`
    (may start empty) is filled with what was loaded.  -> ({index: `.l3c` bytes}, number of distinct
    padded shapes, number of forward passes).  `marks` (dict) receives host time stamps of the stages.  n_pinned: page-locked staging
    buffers of this process (helpers/sharding.host_budget: fewer per rank when several ranks share a host).

    The host never waits for the GPU between images and the GPU never for the host [measured on 500 images, profiles/
    r03_dataset_stages.log: padding on the host, pageable H2D copies and one D2H + slicing at the end were 48 % of the wall time, all of it
    with the GPU idle]:
      * the raw images of a forward pass are copied back to back into a page-locked staging buffer and cross PCIe as ONE
        asynchronous copy on that pass's forward stream; the zero padding (reference: helpers/pad.py, mode 'constant' as the tester
        uses it) is applied on the DEVICE while the batch tensor is filled;
      * this happens right before the pass is enqueued (Bitcoding.encode_many's `upload` hook), so staging pass k+1 overlaps
        the GPU's work on pass k;
      * the files of coder group g are assembled on the device, copied back and cut into bytes objects while the GPU runs
        the forward passes of group g+1 (`on_group` hook), on the side stream that coded them.  n_groups (default: one group per
        ~13 passes, 2..8): nothing overlaps the last coder launch and its collection, and the first group holds the longest chains
        [measured: 100 images (53 passes): 8 / 4 / 2 equal groups 104 / 120 / 108 MPix/s, halving groups (1/2, 1/4, ...) 99-102, a small
        first and last group 105-108; 500 images (224 passes): 8 equal groups 143-148, 16 / 32: 125-147 / 110-113, halving 139-143]."""
    import time
    from ..bitcoding.bitcoding import EncodedBatch

    def mark(name):
        if marks is not None:
            marks[name] = time.perf_counter()

    def check(i, t):   # the staging memcpy reinterprets the image's bytes: anything but host uint8 CHW would be silently cast / wrapped
        if t.dtype != torch.uint8 or t.is_cuda or t.dim() != 3 or t.shape[0] != 3 or not t.is_contiguous():
            raise ValueError('encode_set: image {} must be a contiguous host uint8 (3,H,W) tensor, got {} {} on {}'.format(
                i, t.dtype, tuple(t.shape), t.device))
        if shapes is not None and tuple(t.shape[-2:]) != tuple(shapes[i]):
            raise ValueError('encode_set: image {} is {} but was announced as {}'.format(i, tuple(t.shape[-2:]), tuple(shapes[i])))
        return t

    if loader is None:
        for i in order:
            check(i, imgs[i])
        shapes = {i: tuple(imgs[i].shape[-2:]) for i in order}
    futures = {}
    # (Round 4 also built passes that hold images of DIFFERENT padded shapes -- "canvas" batches: byte-identical files in 3-6x fewer passes, but
    # no faster [measured, profiles/r04_canvas_passes.log: 200 images 117.5 vs 116.7 MPix/s, 500 images 122.6 vs 146.9]: with three forward
    # streams the GPU is already > 90 % busy on the small passes.  Removed in round 6; the code is in the history, DESIGN_HISTORY.md.)
    chunks, padded, pads, n_shapes = plan_set(shapes, order, max_batch, fac)
    mark('plan (host)')
    ring = getattr(bc, '_h2d_ring', None)
    if ring is None:
        ring = bc._h2d_ring = _PinnedRing(n_pinned)
    files = {}
    spent = collections.defaultdict(float)       # host seconds per activity (marks['host seconds'])

    def upload(ci):
        chunk, (Hp, Wp) = chunks[ci], padded[ci]
        t0 = time.perf_counter()
        for i in chunk:
            if i in futures:
                imgs[i] = check(i, futures.pop(i).result())
        spent['wait for the image readers'] += time.perf_counter() - t0
        sizes = [imgs[i].numel() for i in chunk]
        t0 = time.perf_counter()
        k, stage = ring.take(sum(sizes))
        t1 = time.perf_counter()
        off, stage_np = 0, stage.numpy()
        for i, n in zip(chunk, sizes):
            # a plain memcpy: torch's copy_ fans a 2 MB copy out over every core it sees [measured: 0.5 GB/s on the 128-core box]
            stage_np[off:off + n] = imgs[i].numpy().reshape(-1)
            off += n
        t2 = time.perf_counter()
        spent['wait for a staging buffer'] += t1 - t0
        spent['copy into the staging buffer'] += t2 - t1
        try:
            return to_device(chunk, sizes, Hp, Wp, k, stage)
        finally:
            spent['enqueue H2D + device padding'] += time.perf_counter() - t2

    def to_device(chunk, sizes, Hp, Wp, k, stage):
        dev = stage.cuda(non_blocking=True)
        ring.sent(k)
        if not any(any(pads[i]) for i in chunk):
            return dev.view(len(chunk), 3, Hp, Wp)
        # zero padding of every image: written on the device
        x = torch.zeros((len(chunk), 3, Hp, Wp), dtype=torch.uint8, device='cuda')
        off = 0
        for b, (i, n) in enumerate(zip(chunk, sizes)):
            h, w = imgs[i].shape[-2:]
            left, _, top, _ = pads[i]
            x[b, :, top:top + h, left:left + w] = dev[off:off + n].view(3, h, w)
            off += n
        return x

    def on_group(group):
        t0 = time.perf_counter()
        encs, pad_lists, owners = [], [], []
        for ci, enc in group:
            encs.append(enc)
            pad_lists.append([pads[i] for i in chunks[ci]])
            owners.append(chunks[ci])
        for idxs, fs in zip(owners, EncodedBatch.many_to_bytes(encs, pad_lists)):
            for i, f in zip(idxs, fs):
                files[i] = f
        spent['collect files (sizes, assembly, D2H, slicing)'] += time.perf_counter() - t0

    weights = [len(c) * p[0] * p[1] for c, p in zip(chunks, padded)]
    if loader is not None:      # the readers work through the images in the order encode_many enqueues the passes (largest first)
        for ci in sorted(range(len(chunks)), key=lambda c: -weights[c]):
            for i in chunks[ci]:
                futures[i] = pool.submit(loader, i)
    if n_groups is None:
        n_groups = max(2, min(8, int(round(len(chunks) / 13.0))))
    bc.encode_many(list(range(len(chunks))), upload=upload, on_group=on_group, n_groups=n_groups, weights=weights)
    mark('staging + H2D + forward + coder + D2H + files (pipelined)')
    if marks is not None:
        marks['host seconds'] = dict(spent)
    return files, n_shapes, len(chunks)


def file_padded_shape(data):
    """(H, W) of the PADDED image a `.l3c` byte string holds, from its first (coarsest) scale record alone: u16 x4 padding, then
    u8 C, u16 H, u16 W of the coarsest scale (reference bitcoding.py:326-375) -- the image is 2**(records - 1) times that.  Cheap: the
    record count comes from walking the length fields, no payload is touched."""
    import struct
    from ..bitcoding.bitcoding import count_scale_records
    n = count_scale_records(data)
    _, H, W = struct.unpack_from('<BHH', data, 8)
    return (H << (n - 1), W << (n - 1))


def plan_decode_set(files, order, max_batch):
    """Host-side plan of a set decode: files of equal padded shape share a batch of at most max_batch (the mirror of plan_set, from
    the files' own headers).  -> (chunks: list of index lists, padded shape per chunk)."""
    groups = collections.defaultdict(list)
    for i in order:
        groups[file_padded_shape(files[i])].append(i)
    chunks, padded = [], []
    for shape, idxs in groups.items():
        for k in range(0, len(idxs), max_batch):
            chunks.append(idxs[k:k + max_batch])
            padded.append(shape)
    return chunks, padded


def decode_set(bc, files, order, max_batch=16, marks=None, lanes=None, chain_cus=0, n_pinned=None, ragged=None):
    """files: {index: `.l3c` bytes} (as `encode_set` returns them); order: the indices to decode.  -> {index: uint8 (3,H,W) HOST tensor},
    the padding undone.  The mirror of `encode_set` for the reference's folder evaluation, which decodes EVERY file it wrote and
    compares it with the input (multiscale_tester.py:353-381, assert_equal at :373): files of equal padded shape share a batch,
    largest batches first, the batches stream through `Bitcoding.decode_many` (batch i + 1's convolutions beside batch i's RGB chains);
    the decoded pixels leave the device as uint8 through `n_pinned` page-locked buffers while later batches decode."""
    import time
    from . import pad as _pad

    def mark(name):
        if marks is not None:
            marks[name] = time.perf_counter()

    n_lanes = bc.N_DECODE_LANES if lanes is None else int(lanes)
    if n_pinned is None:
        n_pinned = max(n_lanes, min(bc.RAGGED_GROUP, len(order))) + 2     # a finished batch must never wait for a free D2H buffer (a ragged group finishes all its batches at once)
    chunks, padded = plan_decode_set(files, order, max_batch)
    by_size = sorted(range(len(chunks)), key=lambda k: -len(chunks[k]) * padded[k][0] * padded[k][1])
    mark('plan (host)')
    out, pending = {}, []
    ring = getattr(bc, '_d2h_ring', None)
    if ring is None or len(ring) != n_pinned:
        ring = bc._d2h_ring = [None] * n_pinned
    turn = [0]

    def collect(block, keep=0):
        while len(pending) > keep and (block or pending[0][2].query()):
            ci, host, ev, paddings = pending.pop(0)
            ev.synchronize()
            for k, i in enumerate(chunks[ci]):
                img = host[k]
                if any(paddings[k]):
                    img = _pad.undo_pad(img.unsqueeze(0), *paddings[k])[0]
                out[i] = img.clone()

    def on_batch(n, pixels, paddings):
        ci = by_size[n]
        u8 = pixels                                       # uint8: 0..255 by construction (symbols of a 256-symbol alphabet)
        collect(True, keep=n_pinned - 1)                  # the buffer about to be reused must have been read out
        k = turn[0] = (turn[0] + 1) % n_pinned
        if ring[k] is None or ring[k].numel() < u8.numel():
            ring[k] = torch.empty(-(-u8.numel() // (1 << 20)) << 20, dtype=torch.uint8, pin_memory=True)
        host = ring[k][:u8.numel()].view(u8.shape)
        host.copy_(u8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        pending.append((ci, host, ev, paddings))
        collect(False)

    bc.decode_many([[files[i] for i in chunks[ci]] for ci in by_size], on_batch=on_batch, lanes=lanes, chain_cus=chain_cus,
                   out_dtype=torch.uint8, ragged=ragged)
    collect(True)
    mark('parse + H2D + decode + D2H (pipelined)')
    return out
