"""Bitcoding -- `.l3c` container encode / decode with the whole data path on the GPU.

Same surface as the reference's bitcoding/bitcoding.py (`Bitcoding(blueprint, times, compare_with_theory)`,
`encode(img, pout) -> bpsp` :50-123, `decode(pin) -> 1CHW long` :125-161) and the same byte format (:326-375):

    u16 x4  padding (left, right, top, bottom)
    for scale = coarsest .. 0:   u8 C, u16 H, u16 W;  for each channel: u32 nbytes, payload;  magic 46 E2 84 92

What changed is WHERE the work happens.  The reference loops over scales and channels in Python, builds one CDF table per
channel, and range-codes it on one CPU thread (coders.py:38-66 -> torchac.cpp).  Here a batch of B equally sized images
is pushed through
    encoder:  net forward  ->  per scale ONE fused kernel P,symbols -> coding intervals of all B*C streams
              ->  ONE grouped range-coder launch over all scales (and, with encode_many, over all batches of a
                  heterogeneous image set)                                         -> bytes + lengths in HBM
    decoder:  per scale get_P -> parameters -> uint16 tables -> ONE range-decoder launch (a stream per wavefront);
              only the RGB scale is serial in its channels (R -> G -> B, the lambda coupling of logistic_mixture.py:262-272)
and only finished byte strings cross PCIe.  `encode_batch` / `decode_batch` are the native entry points; `encode` /
`decode` are the reference's one-image file API on top of them (auto-crop parts included).
"""
import os
import struct

import numpy as np
import torch

from .. import auto_crop, ops
from ..helpers import pad
from . import part_suffix_helper

_MAGIC_VALUE_SEP = b'\x46\xE2\x84\x92'


class _NullTimes(object):
    """Stand-in for the reference's StackTimeLogger when no timing is requested."""

    class _Ctx(object):
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def run(self, *a, **kw):
        return self._Ctx()

    prefix_scope = combine = run


def uniform_cdf_row(L):
    """The coarsest scale's table: round(cumsum(1/L) * 2^16) with a leading 0, as int16 (bitcoding.py:297-323).
    Same fp32 torch ops as the reference so the row is bit-identical (L=25: 0, 2621, 5243, ...)."""
    histo = torch.ones(L, dtype=torch.float32) / L
    cdf = torch.cumsum(torch.ones(1, L) * histo, -1).mul_(2 ** 16).round()
    cdf = torch.cat((torch.zeros(1, 1), cdf), dim=-1)
    return cdf.to(torch.int16).reshape(-1)


_STAGING = [None]


def _staging(nbytes):
    """Host staging buffer for the D2H copy of finished files: ONE page-locked buffer, grown geometrically and reused (page-
    locking costs far more than the copy itself); the caller consumes the returned view before the next call."""
    buf = _STAGING[0]
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 2 * (buf.numel() if buf is not None else 0), 64 << 20), dtype=torch.uint8, pin_memory=True)
        _STAGING[0] = buf
    return buf[:nbytes]


def coder_group_cuts(n_passes, n_groups):
    """After which forward passes (1-based counts, ascending, the last one == n_passes) `encode_many` launches the range coder:
    n_groups equal groups (int), or cut points given as cumulative fractions of the passes (list / tuple, e.g. (0.5, 0.75, 1.0))."""
    if n_passes <= 0:
        return []
    if isinstance(n_groups, (list, tuple)):
        return sorted({min(n_passes, max(1, int(round(f * n_passes)))) for f in n_groups} | {n_passes})
    per_group = -(-n_passes // max(1, int(n_groups)))
    return list(range(per_group, n_passes, per_group)) + [n_passes]


def rgb_pipeline_schedule(n_chunks, C, D):
    """Steps of the chunk-pipelined RGB decode: step t handles chunk t - D c of channel c.  Returns a list (one entry per
    step) of lists of (channel, chunk).  Channel c's chunk j needs the symbols of the channels < c in chunk j: with D = 1
    they come from the previous step, with D = 2 from two steps back -- the tables of step t + 1 then do not depend on the
    decode launch of step t and the two can overlap."""
    return [[(c, t - D * c) for c in range(C) if 0 <= t - D * c < n_chunks] for t in range(n_chunks + D * (C - 1))]


class EncodedBatch(object):
    """Device-resident result of `Bitcoding.encode_batch`: per scale (coarse -> fine) the coder output of its B*C streams.
    Nothing has been synchronised or copied to the host until `payloads()` / `to_bytes()` is called."""

    def __init__(self, B, padded_shape):
        self.B = B
        self.padded_shape = padded_shape     # (H, W) of the padded images
        self.scales = []                     # (C, H, W, out uint8 (B*C, stride), nbytes int32 (B*C,))
        self.pending = []                    # (C, H, W, intervals) of prepare_batch, consumed by Bitcoding.code
        self.done = []                       # events on the coder streams; wait() orders the current stream after them
        self.coder_stream = None             # the side stream of the last `code` call (the natural place for the D2H of the files)

    def wait(self):
        """Make the current stream wait for the range-coder launches (they run on side streams)."""
        cur = torch.cuda.current_stream()
        for ev in self.done:
            cur.wait_event(ev)
        for _, _, _, out, nbytes in self.scales:     # allocated under the coder's side stream, consumed on this one
            out.record_stream(cur)
            nbytes.record_stream(cur)
        return self

    def total_payload_bytes(self):
        """int64 device tensor (B,): entropy-coded bytes per image (no host sync)."""
        self.wait()
        tot = None
        for C, H, W, out, nbytes in self.scales:
            t = nbytes.to(torch.int64)
            t = torch.where(t < 0, torch.full_like(t, -(1 << 40)), t).reshape(self.B, C).sum(dim=1)    # L3C_AC_OVERRUN stays visible in the sum
            tot = t if tot is None else tot + t
        return tot

    def file_sizes(self):
        """(B,) file size in bytes incl. the fixed framing: 8 + sum_scales (5 + 4*C + 4) + payload."""
        overhead = 8 + sum(5 + 4 * C + 4 for C, _, _, _, _ in self.scales)
        return self.total_payload_bytes() + overhead

    @staticmethod
    def _checked_nbytes(n):
        """Host-side stream lengths (numpy) -> the same, or L3CError when a stream reported L3C_AC_OVERRUN (int32 -1: its intervals
        violated c_high > c_low, i.e. table rows not strictly increasing).  The ONE place that turns the device-side marker into an
        error: `payloads`, `many_to_host_buffer` (through the file sizes) and therefore every file writer go through it."""
        if (np.asarray(n) < 0).any():
            from .. import _lib
            raise _lib.L3CError('range coder overrun: a stream asked for more than 16 bits per symbol (table rows not strictly increasing)')
        return n

    def payloads(self):
        """list over scales (coarse -> fine) of [B][C] bytes objects (one D2H copy per scale)."""
        self.wait()
        res = []
        for C, H, W, out, nbytes in self.scales:
            n = self._checked_nbytes(nbytes.cpu().numpy())
            host = out[:, :int(n.max())].cpu().numpy()
            res.append([[host[b * C + c, :n[b * C + c]].tobytes() for c in range(C)] for b in range(self.B)])
        return res

    def to_bytes(self, padding_tuples=None):
        """-> list of B `.l3c` byte strings.  The files are assembled ON THE DEVICE (l3c_container_write: headers, length
        fields, payloads re-aligned to their byte offsets, all files back to back) and cross PCIe as one copy of exactly
        their size; the only host work left is cutting the buffer into B bytes objects."""
        return EncodedBatch.many_to_bytes([self], [padding_tuples])[0]

    def to_host_buffer(self, padding_tuples=None):
        """-> (uint8 numpy array holding the B files back to back, file offsets, file sizes).  The array is a view of a
        reused staging buffer: consume it before the next call."""
        return EncodedBatch.many_to_host_buffer([self], [padding_tuples])

    def _write_container(self, dst, offsets, padding_tuples):
        """Enqueue the assembly of this batch's files into `dst` at the (device, int64) byte `offsets`."""
        from .. import _lib
        B = self.B
        pads = np.asarray(padding_tuples if padding_tuples else [(0, 0, 0, 0)] * B, dtype=np.uint16).reshape(B, 4)
        pads = torch.from_numpy(pads.view(np.int16)).cuda()
        arr = (_lib.ContainerScale * len(self.scales))()
        for k, (C, H, W, out, nbytes) in enumerate(self.scales):
            arr[k] = _lib.ContainerScale(ops.ptr(out, torch.uint8), ops.ptr(nbytes, torch.int32), out.shape[1], C, H, W)
        ops.call('l3c_container_write', arr, len(self.scales), B, ops.ptr(pads), ops.ptr(offsets, torch.int64), ops.ptr(dst),
                 ops.stream())

    @staticmethod
    def many_to_host_buffer(encs, padding_lists=None):
        """The files of several EncodedBatches in ONE device buffer, ONE host synchronisation (their sizes) and ONE D2H copy.
        -> (uint8 numpy array, file offsets, file sizes); files in the order of `encs`, batch order inside."""
        for e in encs:
            e.wait()
        sizes = torch.cat([e.file_sizes() for e in encs])
        offs = torch.cumsum(sizes, 0) - sizes
        # the synchronisation: how many bytes there are (an overrun stream makes its file's size hugely negative: total_payload_bytes)
        sizes_h = EncodedBatch._checked_nbytes(sizes.cpu().numpy().astype(np.int64))
        total = int(sizes_h.sum())
        dst = torch.empty(total, dtype=torch.uint8, device='cuda')
        first = 0
        for i, e in enumerate(encs):
            e._write_container(dst, offs[first:first + e.B].contiguous(), padding_lists[i] if padding_lists else None)
            first += e.B
        host = _staging(total)
        host.copy_(dst)
        torch.cuda.current_stream().synchronize()
        offs_h = np.concatenate([[0], np.cumsum(sizes_h)[:-1]]).astype(np.int64)
        return host.numpy(), offs_h, sizes_h

    @staticmethod
    def many_to_bytes(encs, padding_lists=None):
        """-> per EncodedBatch the list of its `.l3c` byte strings."""
        host, offs, sizes = EncodedBatch.many_to_host_buffer(encs, padding_lists)
        res, k = [], 0
        for e in encs:
            res.append([host[offs[k + b]:offs[k + b] + sizes[k + b]].tobytes() for b in range(e.B)])
            k += e.B
        return res

    def to_bytes_host_assembled(self, padding_tuples=None):
        """Reference implementation of `to_bytes` on the host (per-scale copies + Python joins); kept for the tests."""
        pl = self.payloads()
        files = []
        for b in range(self.B):
            pt = padding_tuples[b] if padding_tuples else (0, 0, 0, 0)
            chunks = [struct.pack('<4H', *pt)]
            for (C, H, W, _, _), scale_payloads in zip(self.scales, pl):
                chunks.append(struct.pack('<BHH', C, H, W))
                for p in scale_payloads[b]:
                    chunks += [struct.pack('<I', len(p)), p]
                chunks.append(_MAGIC_VALUE_SEP)
            files.append(b''.join(chunks))
        return files


class Bitcoding(object):
    def __init__(self, blueprint, times=None, compare_with_theory=False, coder_cus=0, auto_recurse=0, file_writer=None,
                 coder_streams=4, forward_streams=3, decode_overlap=None, rgb_window='auto'):
        """coder_streams: side streams the range-coder launches rotate over; forward_streams: streams `encode_many` spreads the forward
        passes of a heterogeneous set over (used when the HIP runtime runs with >= 8 hardware queues -- helpers/runtime.py; the package
        does NOT ask for them on import: applications that code image sets call l3c_pytorch_amd.configure_hip_queues() before their
        first HIP call -- else one, with a warning); decode_overlap: None = the chunk-pipelined RGB decode overlaps its table
        and decoder launches from 16 images on, True / False force either; rgb_window: 'auto' = the RGB decoder builds 64-entry table
        rows around the mixture mean where the previous chunks of the stream say that pays, 'always' / 'never' force either form
        (_decode_rgb_pipelined).  None of these changes a bit of a file or of a decoded image.
        coder_cus > 0: reserve that many compute units for the range coder.  `self.compute_stream` is then a stream
        confined to the remaining CUs -- run the network under `torch.cuda.stream(bc.compute_stream)` so the MFMA-bound
        conv kernels and the latency-bound coder wavefronts never share a SIMD (they slow the coder down 2.3x).
        auto_recurse: RGB Shared baseline only -- how many times the coarsest scale is applied again (the reference evaluates it
        with 3, multiscale_tester.py:50, and has no file coding for it, :187-188; here the `.l3c` layout simply carries one more
        scale record per recursion, and the decoder counts the records).
        file_writer: an AsyncFileWriter -- `encode` then hands the finished bytes to its worker threads instead of writing them
        itself, `decode` waits for a pending write of the path it is asked to read."""
        self.blueprint = blueprint
        self.auto_recurse = int(auto_recurse)
        if self.auto_recurse and not blueprint.net.config_ms.rgb_bicubic_baseline:
            raise ValueError('auto_recurse is only defined for the RGB baselines')
        if self.auto_recurse and blueprint.net.scales != 1:
            # the decoder accepts extra scale records only for the single-scale RGB Shared model (decode_batch), like the tester's
            # --recursive flag (multiscale_tester.py:123-132): a file written otherwise could never be read back
            raise ValueError('auto_recurse needs the single-scale RGB Shared model (num_scales == 1)')
        self.file_writer = file_writer
        self.compare_with_theory = compare_with_theory
        self.times = times if times is not None else _NullTimes()
        self._const = {}
        self._coder_streams = None
        self.coder_cus = coder_cus
        self.compute_stream = None
        self.N_SIDE_STREAMS = max(1, int(coder_streams))
        self.N_FORWARD_STREAMS = max(1, int(forward_streams))
        self.decode_overlap = decode_overlap
        if rgb_window not in ('auto', 'always', 'never'):
            raise ValueError('rgb_window must be auto, always or never')
        self.rgb_window = rgb_window
        if coder_cus:
            from .. import _lib
            _, n_cu, _ = _lib.device_info()
            self.compute_stream = _lib.cu_range_stream(0, n_cu - coder_cus)
            self._coder_range = (n_cu - coder_cus, coder_cus)

    # [measured, profiles/r02_coder_streams_small_batches.log] 8 or 12 side streams do not help with the runtime's default of four
    # hardware queues (they alias); with GPU_MAX_HW_QUEUES=8, four streams reach 27.8 / 75.7 MPix/s at 1 / 4 images per step
    N_SIDE_STREAMS = 4        # (instances: the constructor's coder_streams)

    def _side_stream(self):
        """Side stream for the range coder: its launches are a handful of long-running wavefronts (a lane per stream),
        so they are overlapped with whatever the main stream does next (the next batch's convolutions).  Consecutive
        calls rotate over N_SIDE_STREAMS streams so that the coder launches of successive batches may overlap too."""
        if self._coder_streams is None:
            if self.coder_cus:
                from .. import _lib
                make = lambda: _lib.cu_range_stream(*self._coder_range)   # noqa: E731
            else:
                make = torch.cuda.Stream
            self._coder_streams = [make() for _ in range(self.N_SIDE_STREAMS)]
            self._stream_turn = 0
        self._stream_turn = (self._stream_turn + 1) % self.N_SIDE_STREAMS
        return self._coder_streams[self._stream_turn]

    # ---- constants --------------------------------------------------------------------------------------------------

    def _targets(self, dmll):
        key = ('t', dmll.x_min, dmll.x_max, dmll.L)
        if key not in self._const:
            self._const[key] = dmll.coding_targets('cuda')
        return self._const[key]

    def _uniform_row(self, L):
        key = ('u', L)
        if key not in self._const:
            self._const[key] = uniform_cdf_row(L).cuda()
        return self._const[key]

    def n_predicted_scales(self):
        """Scales coded with the network's prediction (the coarsest one on top of them is coded with the uniform prior)."""
        return self.blueprint.net.scales + self.auto_recurse

    def padding_factor(self):
        return 2 ** self.n_predicted_scales()

    def iter_scale_dmll(self, n_predicted=None):
        """coarsest -> finest: (scale, dmll, uniform)   (reference :163-169; for the RGB baselines every scale is an RGB scale)"""
        losses = self.blueprint.losses
        n = self.n_predicted_scales() if n_predicted is None else n_predicted
        for scale in reversed(range(n + 1)):
            yield (scale, losses.loss_dmol_rgb if scale == 0 else losses.loss_dmol_n, scale == n)

    # ---- native batched API -----------------------------------------------------------------------------------------

    def prepare_batch(self, imgs, out=None):
        """First half of `encode_batch`: network forward (unless `out` is given) and, per scale, the fused head that
        turns P and the symbols into the coding intervals of all B*C streams.  Everything is enqueued on the current
        stream.  -> EncodedBatch whose `pending` list awaits `code()`."""
        net = self.blueprint.net
        fac = self.padding_factor()
        B, _, H, W = imgs.shape
        assert H % fac == 0 and W % fac == 0, 'pad first: {}x{} not divisible by {}'.format(H, W, fac)
        if out is None:
            out = net(imgs.to('cuda', torch.float32), self.auto_recurse)
        assert len(out.raw.P) == self.n_predicted_scales(), (len(out.raw.P), self.n_predicted_scales())
        raw = out.raw
        K = net.config_ms.prob.K
        enc = EncodedBatch(B, (H, W))
        for scale, dmll, uniform in self.iter_scale_dmll():
            sym = raw.sym[scale]
            _, C, Hs, Ws = sym.shape
            if uniform:
                iv = ops.intervals_from_table(self._uniform_row(dmll.L), sym.reshape(B * C, Hs * Ws), B * C, Hs * Ws,
                                              broadcast_row=True)
            else:
                iv = ops.dmll_encode_intervals(raw.P[scale], sym.contiguous(), self._targets(dmll), C, K, dmll.rgb_scale)
            enc.pending.append((C, Hs, Ws, iv))
        return enc

    def code(self, batches):
        """Second half: ONE grouped range-coder launch (l3c_ac_encode_groups) over every scale of every prepared batch
        -- all their streams are coded concurrently, whatever the image sizes -- on a side stream that the current
        stream does not wait for.  Returns `batches`."""
        groups, owners = [], []
        for enc in batches:
            for C, Hs, Ws, iv in enc.pending:
                groups.append((iv, enc.B * C, Hs * Ws))
                owners.append((enc, C, Hs, Ws))
        if not groups:
            return batches
        main, side = torch.cuda.current_stream(), self._side_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            for iv, _, _ in groups:
                iv.record_stream(side)
            results, ws = ops.ac_encode_groups(groups)
            done = torch.cuda.Event()
            done.record(side)
        for (enc, C, Hs, Ws), (stream_bytes, nbytes) in zip(owners, results):
            enc.scales.append((C, Hs, Ws, stream_bytes, nbytes))
        for enc in batches:
            enc.pending = []
            enc.done.append(done)
            enc.coder_stream = side
        return batches

    def encode_batch(self, imgs, out=None):
        """imgs: (B,3,H,W), H and W multiples of 2**num_scales, values 0..255 (any dtype / device).
        Enqueues the whole encode and returns an EncodedBatch (no host sync).
        `out`: a network output for `imgs` computed earlier (avoids a second forward)."""
        return self.code([self.prepare_batch(imgs, out)])[0]

    N_FORWARD_STREAMS = 3     # (instances: the constructor's forward_streams) used when the HIP runtime was given >= 8 hardware queues, see encode_many
    N_CODER_GROUPS = 4

    def encode_many(self, batches, upload=None, on_group=None, n_groups=None, weights=None):
        """batches: list of (B_i,3,H_i,W_i) tensors (shapes may differ between entries).  -> list, in the order given, of EncodedBatch.
        Small batches leave most of the machine idle (one 768x512 image is 192 tiles at the first scale, 12 at the
        coarsest, for 256 CUs) and their serial coder chains are as long as ever, so
          * the forward passes + heads of different batches run on N_FORWARD_STREAMS streams side by side, largest
            batches first -- only if the process runs with GPU_MAX_HW_QUEUES >= 8: with the runtime's default of 4
            hardware queues the extra streams alias the coder's queue and the long coder kernel stalls them
            [measured: 64 images of 36 shapes end to end: 42 MPix/s on one stream, 38-43 on three with 4 queues, 52 with 8];
          * the range coder is launched N_CODER_GROUPS times (one grouped launch per quarter of the set, on the side
            streams), so that the long chains of the early, large images overlap the later forward passes and only the
            short chains of the smallest images are left at the end.
        Host pipelining (helpers/dataset_codec.encode_set): `upload(entry)` turns an entry of `batches` into its device tensor
        right before that forward pass is enqueued, under the forward stream (staging + H2D of batch k+1 then overlap the GPU's
        work on batch k; `weights[i]` = the entry's pixel count, for the largest-first order); `on_group(list of (index,
        EncodedBatch))` is called for a coder group as soon as its launch has completed (polled after every enqueued pass; the
        rest at the end), under the side stream that coded it -- the D2H and the host's slicing of finished files overlap the
        later groups' forward passes."""
        if not batches:
            return []
        self.blueprint.net._prepare()                  # pack the weights before forking streams
        if weights is None:
            weights = [b.shape[0] * b.shape[2] * b.shape[3] for b in batches]
        order = sorted(range(len(batches)), key=lambda i: -weights[i])
        main = torch.cuda.current_stream()
        from ..helpers import runtime
        if runtime.forward_streams_allowed(self.N_FORWARD_STREAMS) > 1:
            if getattr(self, '_fwd_streams', None) is None:
                self._fwd_streams = [torch.cuda.Stream() for _ in range(self.N_FORWARD_STREAMS)]
            fwd = self._fwd_streams
        else:
            fwd = [main]
        start = torch.cuda.Event()
        start.record(main)
        for st in fwd:
            if st is not main:
                st.wait_event(start)
        cuts = set(coder_group_cuts(len(order), n_groups or self.N_CODER_GROUPS))
        result = [None] * len(batches)
        pending, coded = [], []

        def flat(k):
            return [result[k]]

        def collect_finished(block):
            # oldest group first; without `block` only groups whose coder launch has COMPLETED (the host must never sit waiting for a
            # latency-bound coder launch while the forward streams run dry)
            while coded and (block or flat(coded[0][0])[0].done[-1].query()):
                self._collect(coded.pop(0), result, on_group)

        for n, i in enumerate(order):
            st = fwd[n % len(fwd)]
            with torch.cuda.stream(st):
                x = batches[i] if upload is None else upload(batches[i])
                if x.is_cuda:
                    x.record_stream(st)
                result[i] = self.prepare_batch(x)
                ev = torch.cuda.Event()
                ev.record(st)
            pending.append((i, ev))
            if n + 1 in cuts:
                for _, ev in pending:
                    main.wait_event(ev)            # `code` orders its side stream after the current stream
                self.code([e for k, _ in pending for e in flat(k)])
                if on_group is not None:
                    coded.append([k for k, _ in pending])
                pending = []
            collect_finished(False)
        collect_finished(True)
        return result

    @staticmethod
    def _collect(indices, result, on_group):
        first = result[indices[0]]
        with torch.cuda.stream(first.coder_stream):
            on_group([(k, result[k]) for k in indices])

    def decode_batch(self, files, out_dtype=torch.int64):
        """files: list of B `.l3c` byte strings of equally sized (padded) images -> ((B,3,H,W) int64 on the GPU (the reference's
        dtype, bitcoding.py:125-161; out_dtype: torch.uint8 / int16 for callers that want the pixels without the 8-byte form),
        list of padding tuples).
        Round 6: the host only parses the FRAMING (a few length fields per file, `parse_containers`); the files cross PCIe as they
        are, in one copy from a page-locked buffer, and one kernel (l3c_container_read) cuts every stream out of them in the
        aligned, zero-padded form the range decoders read -- before, 14 `pack_streams` calls copied every payload on the host
        and uploaded it from pageable memory.
        (Tried and dropped [measured, batch 128]: cutting the batch into 2-4 parts that run the same chain on streams of their
        own, so that one part's latency-bound decoder launches would leave room for another part's convolutions and tables:
        0.70 s became 0.87 - 2.98 s with the runtime's four hardware queues (the parts' main and side streams alias and serialise
        each other's long decoder launches) and 0.76 / 1.00 s with 8 or 16 queues -- a decoder wavefront that shares its SIMD
        with MFMA wavefronts runs 2.3x slower, which costs more than the overlap saves; profiles/r02_decode_parts_tried.log.)"""
        net = self.blueprint.net
        rgb_net = bool(net.config_ms.rgb_bicubic_baseline)
        K = net.config_ms.prob.K
        B = len(files)
        parsed = parse_containers(files)
        n_pred = len(parsed.scales) - 1
        if n_pred < net.scales or (n_pred != net.scales and not (rgb_net and net.scales == 1)):
            raise ValueError('invalid file: {} scale records, the model codes {}'.format(n_pred + 1, net.scales + 1))
        streams = _upload_streams(files, parsed)
        bn_prev, F_prev, sym, prev_hw = None, None, None, None
        for k, (scale, dmll, uniform) in enumerate(self.iter_scale_dmll(n_pred)):
            C, H, W = parsed.scales[k]
            # the headers are untrusted input: a wrong C / H / W would make the table and decoder kernels index P and the symbol
            # buffers out of bounds (the reference fails with a shape error here, bitcoding.py:248-266)
            if uniform:
                buf, offs, lens = streams.scale(k)
                if C != net.config_ms.q.C or H < 1 or W < 1:
                    raise ValueError('invalid file: coarsest scale header (C={}, H={}, W={})'.format(C, H, W))
                if int(parsed.nbytes[k].max()) > 2 * H * W + 64:      # > 16 bits per symbol: not a stream of this coder
                    raise ValueError('invalid file: coarsest scale payload longer than {} symbols can be'.format(H * W))
                sym = ops.ac_decode(self._uniform_row(dmll.L), buf, offs, lens, B * C, H * W, True,
                                    broadcast_row=True).reshape(B, C, H, W)
            else:
                P, F_prev = net.get_P(scale, bn_prev, F_prev, n_scales_total=n_pred)
                P = ops.as_pixel_major(P)
                n_params = 4 if dmll.rgb_scale else 3
                expect = (P.shape[-1] // (n_params * K), 2 * prev_hw[0], 2 * prev_hw[1])
                if (C, H, W) != expect or tuple(P.shape[1:3]) != (H, W):
                    raise ValueError('invalid file: scale {} header (C, H, W) = {} but the network predicts {}'.format(
                        scale, (C, H, W), expect))
                targets = self._targets(dmll)
                buf, offs, lens = streams.scale(k)      # (the last record's streams are staged and uploaded HERE: behind the convolutions just enqueued)
                if dmll.rgb_scale:
                    sym = self._decode_rgb_pipelined(P, targets, (buf, offs, lens), B, C, K, H, W)
                else:
                    sym = self._decode_z_scale(P, targets, (buf, offs, lens), B, C, K, H, W)
            prev_hw = (H, W)
            if scale == 0:
                break                                   # the finest scale's symbols ARE the pixel values (to_bn of the RGB scale: x 1 + 0)
            bn_prev = ops.sym_to_bn(sym, dmll.bin_width, dmll.x_min)
            if rgb_net and scale > 0:                  # BicubicDownsamplingEnc: the decoder is fed value - mean (net.py:72-80)
                bn_prev = bn_prev - _rgb_mean_tensor(bn_prev.device)
        return sym.to(out_dtype), parsed.padding

    N_DECODE_LANES = 8       # decode_many: lanes when every batch is small (fewer than 64 images: latency-bound chains); large batches run one after the other

    def _lanes(self, n, chain_cus):
        """`n` pairs (main stream, side stream) for decode_many.  chain_cus > 0: every side stream -- the latency-bound range-decoder
        chains -- is confined to `chain_cus` compute units (the same number of every XCD, helpers/runtime.balanced_cu_sets) and every main
        stream -- convolutions, table kernels -- to the others, so that a lane's MFMA wavefronts never take the registers or the issue
        slots of another lane's chains."""
        key = (n, chain_cus)
        if getattr(self, '_lane_key', None) != key:
            from .. import _lib
            if chain_cus:
                from ..helpers import runtime
                _, n_cu, _ = _lib.device_info()
                chains, rest = runtime.balanced_cu_sets(n_cu, chain_cus)
                self._lane_streams = [(_lib.cu_mask_stream(rest), _lib.cu_mask_stream(chains)) for _ in range(n)]
            else:
                self._lane_streams = [(torch.cuda.Stream(), torch.cuda.Stream()) for _ in range(n)]
            self._lane_key = key
        return self._lane_streams

    RAGGED_GROUP = 512               # decode_many: at most this many images are decoded together as one ragged group ...
    RAGGED_GROUP_PIXELS = 128 << 20  # ... and at most this many pixels (P of the RGB scale is 480 bytes per pixel: 64 GB; its tables 8 GB)

    def decode_many(self, batches, on_batch=None, lanes=None, chain_cus=0, out_dtype=torch.int64, ragged=None):
        """batches: list of lists of `.l3c` byte strings; the files of ONE entry are equally sized (padded) images (a forward pass of
        `encode_many`), entries may differ in shape.  -> list, in the order given, of ((B_i,3,H_i,W_i) int64 on the GPU, padding tuples),
        or None per entry when `on_batch(index, pixels, padding)` consumes the results as they are enqueued (called under the stream
        that decodes the entry; then nothing is kept on the device).  The reference decodes a folder one file after the other
        (bitcoding.py:125-161 per file, multiscale_tester.py:353-381 over the folder).
        Small batches are LATENCY-bound (one 768x512 image: 93 ms whatever the machine does -- three 393 216-symbol chains, a wavefront
        each), so a set of differently sized images needs many images in flight (round 6):
          * LANES: entry i runs on lane i % lanes, a stream pair of its own -- up to the hardware queues' concurrency (8 lanes: 5.5x);
          * RAGGED RGB (ragged=True, the default for small batches): the lanes stop before the RGB scale -- two thirds of an image's
            latency --, and the RGB scales of up to RAGGED_GROUP images of DIFFERENT sizes run in lock step as ONE ragged batch
            (l3c_decode_rgb_ragged: one table launch and one decoder launch per pipeline step for all of them), on a stream pair of
            its own while the lanes work on the next group.
        Large batches (>= 64 images) run one after the other: [measured, profiles/r06_decode_lanes_probe.log] batches of 128 on two lanes
        62-146 MPix/s (erratic: the lanes' long decoder launches and convolutions alias on the hardware queues) against 142-144 for one
        lane; with the chains on 64 CUs of their own 125-147 -- the round-5 verdict's bar for keeping that overlap was 180."""
        n = self.N_DECODE_LANES if lanes is None else int(lanes)
        if lanes is None and max(len(f) for f in batches) >= 64:
            n = 1
        result = [None] * len(batches)
        if n <= 1 or len(batches) <= 1:
            for i, files in enumerate(batches):
                pixels, padding = self.decode_batch(files, out_dtype)
                if on_batch is not None:
                    on_batch(i, pixels, padding)
                else:
                    result[i] = (pixels, padding)
            return result
        self.blueprint.net._prepare()                  # pack the weights before forking streams
        outer = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(outer)
        lane_streams = self._lanes(n, chain_cus)
        for main, _ in lane_streams:
            main.wait_event(start)
        use_ragged = (ragged is None or ragged) and all(len(f) < 64 for f in batches)
        done = []

        def finish(i, pixels, padding, stream):
            if on_batch is not None:
                on_batch(i, pixels, padding)
            else:
                pixels.record_stream(outer)
                result[i] = (pixels, padding)

        if not use_ragged:
            for i, files in enumerate(batches):
                main, side = lane_streams[i % n]
                with torch.cuda.stream(main):
                    self._lane_side = side
                    try:
                        pixels, padding = self.decode_batch(files, out_dtype)
                    finally:
                        self._lane_side = None
                    finish(i, pixels, padding, main)
                    done.append(main.record_event())
        else:
            if getattr(self, '_rgb_streams', None) is None:
                self._rgb_streams = (torch.cuda.Stream(), torch.cuda.Stream())
            rgb_main, rgb_side = self._rgb_streams
            rgb_main.wait_event(start)
            # groups of (nearly) EQUAL pixel counts below the budget: [measured, profiles/r06_set_decode_group_budget.log, 500 images = 312 MPix]
            # groups of 48 / 96 / 160 MPix: 69 / 95 / 32 MPix/s (peak 55 / 96 / 157 GB: two consecutive groups' buffers then no longer fit the
            # allocator's caches and every phase pays hipFree + hipMalloc)
            pix = []
            for files in batches:
                _, H, W = parse_containers(files[:1]).scales[-1]
                pix.append(len(files) * H * W)
            n_groups = max(1, -(-sum(pix) // self.RAGGED_GROUP_PIXELS))
            target = sum(pix) / float(n_groups)
            group, n_img, n_pix = [], 0, 0
            for i, files in enumerate(batches):
                group.append((i, files))
                n_img += len(files)
                n_pix += pix[i]
                if n_img >= self.RAGGED_GROUP or n_pix >= target or i + 1 == len(batches):
                    self._decode_group_ragged(group, lane_streams, rgb_main, rgb_side, out_dtype, finish)
                    group, n_img, n_pix = [], 0, 0
            done.append(rgb_main.record_event())
        for ev in done:
            outer.wait_event(ev)
        return result

    def _decode_group_ragged(self, group, lane_streams, rgb_main, rgb_side, out_dtype, finish):
        """One group of decode_many's ragged form, in PHASES over all its entries (batches of different shapes):
            lanes:   upload, the coarsest scale (uniform prior), P of the next scale (get_P per shape)      -- small launches, a lane per entry
            ragged:  that scale's symbols of ALL images in lock step (bottleneck scale: one table launch + one decoder launch for every
                     image and channel; RGB scale: the chunk pipeline of l3c_decode_rgb_ragged), on (rgb_main, rgb_side)
            lanes:   P of the next finer scale ...                                                          -- and so on down to scale 0
        An image's serial chains -- 12 ms at scale 1, 60-70 ms at scale 0 for 768x512 -- are thereby paid once per GROUP instead of once
        per image; what is left per image are the decoder-side convolutions of its shape."""
        import numpy as np
        net = self.blueprint.net
        rgb_net = bool(net.config_ms.rgb_bicubic_baseline)
        K = net.config_ms.prob.K
        n = len(lane_streams)
        st = []
        for i, files in group:
            parsed = parse_containers(files)
            n_pred = len(parsed.scales) - 1
            if n_pred < net.scales or (n_pred != net.scales and not (rgb_net and net.scales == 1)):
                raise ValueError('invalid file: {} scale records, the model codes {}'.format(n_pred + 1, net.scales + 1))
            st.append({'i': i, 'files': files, 'parsed': parsed, 'B': len(files), 'lane': lane_streams[i % n], 'F': None, 'n_pred': n_pred})
        if len({e['n_pred'] for e in st}) != 1:
            raise ValueError('decode_many: the files of a set must come from one model (different numbers of scale records)')
        n_pred = st[0]['n_pred']
        plan = list(self.iter_scale_dmll(n_pred))            # record k -> (scale, dmll, uniform), coarse -> fine
        if getattr(self, '_alloc_stream', None) is None:
            self._alloc_stream = torch.cuda.Stream()
        mains = [m for m, _ in lane_streams]

        def ragged_buffers(n_floats, n_sym):
            # from a stream that runs nothing but the zero fill: allocated under rgb_main they would be ordered behind the previous group's whole
            # last phase (the allocator reuses a stream's blocks in stream order) and every lane of this group would wait for it
            with torch.cuda.stream(self._alloc_stream):
                P_rag = torch.empty(n_floats, dtype=torch.float32, device='cuda')
                sym_rag = torch.zeros(n_sym, dtype=torch.int16, device='cuda')
                ev = self._alloc_stream.record_event()
            for t in (P_rag, sym_rag):
                for s_ in mains + [rgb_main, rgb_side]:
                    t.record_stream(s_)
            return P_rag, sym_rag, ev

        # ---- the coarsest scale: uniform prior, one launch per entry on its lane
        scale, dmll, uniform = plan[0]
        assert uniform
        for e in st:
            C, H, W = e['parsed'].scales[0]
            if C != net.config_ms.q.C or H < 1 or W < 1:
                raise ValueError('invalid file: coarsest scale header (C={}, H={}, W={})'.format(C, H, W))
            if int(e['parsed'].nbytes[0].max()) > 2 * H * W + 64:
                raise ValueError('invalid file: coarsest scale payload longer than {} symbols can be'.format(H * W))
            main, _ = e['lane']
            with torch.cuda.stream(main):
                e['streams'] = _upload_streams(e['files'], e['parsed'])
                buf, offs, lens = e['streams'].scale(0)
                e['sym'] = ops.ac_decode(self._uniform_row(dmll.L), buf, offs, lens, e['B'] * C, H * W, True, broadcast_row=True).reshape(e['B'], C, H, W)
                e['hw'] = (H, W)
        prev = dmll
        # ---- every predicted scale, coarse -> fine
        keep = []
        for k in range(1, n_pred + 1):
            scale, dmll, _ = plan[k]
            n_params = 4 if dmll.rgb_scale else 3
            Cs = 3 if dmll.rgb_scale else net.config_ms.q.C
            Kp, Lp = n_params * Cs * K, dmll.L + 1
            hws, pixbase, p = [], [], 0
            for e in st:
                C, H, W = e['parsed'].scales[k]
                if (C, H, W) != (Cs, 2 * e['hw'][0], 2 * e['hw'][1]):
                    raise ValueError('invalid file: scale {} header (C, H, W) = {} but the network predicts {}'.format(
                        scale, (C, H, W), (Cs, 2 * e['hw'][0], 2 * e['hw'][1])))
                pixbase.append(p)
                p += e['B'] * H * W
                hws += [H * W] * e['B']
                e['hw'] = (H, W)
            total, Btot = p, len(hws)
            P_rag, sym_rag, alloc_ev = ragged_buffers(total * Kp, Cs * total)
            rgb_main.wait_event(alloc_ev)
            for g, e in enumerate(st):
                main, side = e['lane']
                main.wait_event(alloc_ev)
                with torch.cuda.stream(main):
                    bn = ops.sym_to_bn(e['sym'], prev.bin_width, prev.x_min)
                    if rgb_net:                                  # BicubicDownsamplingEnc: the decoder is fed value - mean (net.py:72-80)
                        bn = bn - _rgb_mean_tensor(bn.device)
                    self._lane_side = side
                    try:
                        P, e['F'] = net.get_P(scale, bn, e['F'], n_scales_total=n_pred)
                    finally:
                        self._lane_side = None
                    P = ops.as_pixel_major(P)
                    H, W = e['hw']
                    if tuple(P.shape) != (e['B'], H, W, Kp):
                        raise ValueError('invalid file: the network predicts {} at scale {}, the file says {}'.format(tuple(P.shape), scale, (e['B'], H, W, Kp)))
                    P_rag[pixbase[g] * Kp:(pixbase[g] + e['B'] * H * W) * Kp].view(e['B'], H, W, Kp).copy_(P)
                    rgb_main.wait_event(main.record_event())
            # one stream table for the whole group: CHANNEL-major over all images; every entry keeps its own stream buffer, addressed from the lowest one
            base_t = min((e['streams'].buf for e in st), key=lambda t: t.data_ptr())
            offs = np.zeros((Cs, Btot), dtype=np.int64)
            lens = np.zeros((Cs, Btot), dtype=np.int32)
            b0 = 0
            for e in st:
                o, l = e['streams'].scale_host(k)                             # (Cs * B,) channel-major within the entry
                offs[:, b0:b0 + e['B']] = o.reshape(Cs, e['B']) + (e['streams'].buf.data_ptr() - base_t.data_ptr())
                lens[:, b0:b0 + e['B']] = l.reshape(Cs, e['B'])
                b0 += e['B']
            targets = self._targets(dmll)
            with torch.cuda.stream(rgb_main):
                for e in st:
                    e['streams'].buf.record_stream(rgb_main)
                    e['streams'].buf.record_stream(rgb_side)
                    if k == n_pred:
                        e['streams'].finish()       # the bulk of the files (the last record) crosses PCIe here, behind the convolutions just enqueued
                offs_d = ops.upload_small(offs.reshape(-1))
                lens_d = ops.upload_small(lens.reshape(-1))
                if dmll.rgb_scale:
                    min_hw = min(hws)
                    mode = {'never': 0, 'auto': 1, 'always': 2}[self.rgb_window]
                    probe = self.RGB_PROBE if (self.rgb_window == 'auto' and min_hw >= 16 * self.RGB_PROBE) else 0
                    n_regular = max(1, min(self.RGB_CHUNKS, (min_hw - 2 * probe) // 4096))
                    pix0, npix = ops.ragged_rgb_plan(hws, n_regular, probe)
                    overlap = Btot >= 16 if self.decode_overlap is None else bool(self.decode_overlap)
                    keep.append(ops.decode_rgb_ragged(P_rag, targets, sym_rag, base_t, offs_d, lens_d, hws, pix0, npix, K,
                                                      2 if overlap else 1, mode, rgb_side if overlap else None))
                else:
                    keep.append(ops.decode_z_ragged(P_rag, targets, sym_rag, base_t, offs_d, lens_d, hws, Cs, K))
                done = rgb_main.record_event()
            for g, e in enumerate(st):
                H, W = e['hw']
                a = Cs * pixbase[g]
                e['sym'] = sym_rag[a:a + Cs * e['B'] * H * W].view(e['B'], Cs, H, W)
                e['lane'][0].wait_event(done)
            keep.append((sym_rag, offs_d, lens_d))
            del P_rag          # (its block goes back to the allocator as soon as the streams that touched it have passed this point: a group's three P buffers never pile up)
            prev = dmll
        with torch.cuda.stream(rgb_main):
            for e in st:
                finish(e['i'], e['sym'].to(out_dtype), e['parsed'].padding, rgb_main)
        del keep

    def _decode_z_scale(self, P, targets, streams, B, C, K, H, W):
        """A bottleneck scale: its C channels are independent given P, so ONE grouped table launch (fused, straight from P) and one
        grouped decoder launch handle them side by side; table validity is a device-side flag (no host synchronisation).
        streams: (buffer, offsets, lengths) with stream (c, b) at index c * B + b."""
        HW = H * W
        buf, offs, lens = streams
        sym = torch.empty(B, C, H, W, dtype=torch.int16, device='cuda')
        flag = torch.zeros(1, dtype=torch.int32, device='cuda')
        parts = []
        for k in range(0, C, 8):
            cs = list(range(k, min(C, k + 8)))
            tables = ops.dmll_cdf_table_parts(P, None, targets, C, K, False, [(c, 0, HW, flag, None) for c in cs])
            parts = [ops.ac_decode_part(t.reshape(B * HW, -1), buf, offs[c * B:(c + 1) * B], lens[c * B:(c + 1) * B], B, HW, flag,
                                        None, None, True, sym, C * HW, c * HW) for c, t in zip(cs, tables)]
            ops.ac_decode_chunks(parts)
        return sym

    RGB_PROBE = 1024         # symbols of the two probe chunks a channel starts with when window rows are in use (a multiple of 64)
    RGB_CHUNKS = 32          # chunks per RGB channel for batches of 16 images and more (round 5, with window rows: 32 instead of 16 --
    #                          the row form of a chunk follows the stream's misses two chunks earlier, and shorter chunks follow faster:
    #                          batch of 128 0.379 -> 0.365 s, the default-init checkpoint 0.585 -> 0.55 s)
    #                          64 / 96 chunks: another 1.3 / 1.7 % (0.3545 -> 0.350 / 0.3485 s) for two / three times the launches; not taken:
    #                          the host already needs 0.24 s of the 0.35 s to issue them [profiles/r05_decode_chunks_probe.log]
    RGB_CHUNKS_FEW = 32      # ... for a few images: the pipeline's fill (two extra chunk steps) weighs more than a step's launches

    def _decode_rgb_pipelined(self, P, targets, streams, B, C, K, H, W):
        """The RGB scale: channel c's means depend on the decoded values of the channels < c AT THE SAME PIXEL
        (logistic_mixture.py:262-272), so R, G and B are three serial chains of H*W symbols that only have to stay a
        chunk of pixels apart.  Pipeline step t: channel c handles chunk t - D c -- its table rows are built straight from P
        and the symbols decoded so far, then ONE grouped launch resumes the range decoders of all active channels side by side.
        Table validity is a device-side flag, nothing synchronises with the host.  Round 6: the whole schedule is ONE call into
        the library (l3c_decode_rgb, csrc/decode_pipeline.hip: a C loop over a workspace -- one grouped table launch and one decoder
        launch pair per step, a few microseconds of host time each); this method only chooses the chunks, the lag and the row form.

        D = 1 (small batches): everything on the current stream, (chunks + 2) steps of table + decode.
        D = 2 (16 images or more): the channels stay TWO chunks apart, so the tables of step t + 1 need only the symbols of
        step t - 1 and are built on the current stream WHILE a side stream decodes step t: (chunks + 4) steps of
        max(table, decode).  [round 4, profiles/r04_decode_isolation_experiments.log: by the kernel trace the tables (0.28-0.32 s per batch
        of 128) and the decoders (0.29 s) overlap 80 %; confining the decoders to compute units of their own (CU-masked streams, balanced
        over the XCDs) or launching them on a high-priority stream did not shorten the whole decode and is not in the product]

        WINDOW ROWS (round 5; include/l3c_hip.h, l3c_ac_decode_part).  A full row has 257 entries although the symbol almost always
        lies near the mixture's mean: chunk j of a channel gets 65-entry rows around the mean for every image whose decoder missed at most
        1/64 of the symbols of chunk j - 2 (the newest chunk that is certain to be complete when the tables of chunk j are built, with either
        schedule), full rows otherwise and for chunks 0 and 1 (short PROBES on full rows) -- a quarter of the table arithmetic and bytes,
        the symbols the same (a decoder evaluates a missed pixel's full row itself).  rgb_window = 'always' / 'never' (tests) force a form."""
        assert C == 3
        HW = H * W
        buf, offs, lens = streams
        # one image: 393 216-symbol chains at ~140 ns per symbol; with c chunks the three channels take (c + 2) / c chain times
        n_chunks = max(1, min(self.RGB_CHUNKS if B >= 16 else self.RGB_CHUNKS_FEW, HW // 4096))
        step = -(-HW // n_chunks)
        step = -(-step // 64) * 64                       # chunk boundaries on the 64-symbol store blocks
        bounds = [(p0, min(step, HW - p0)) for p0 in range(0, HW, step)]
        if self.rgb_window == 'auto' and HW >= 16 * self.RGB_PROBE:
            P2 = 2 * self.RGB_PROBE
            bounds = [(0, self.RGB_PROBE), (self.RGB_PROBE, self.RGB_PROBE)] + [(p0, min(step, HW - p0)) for p0 in range(P2, HW, step)]
        sym = torch.zeros(B, C, H, W, dtype=torch.int16, device='cuda')
        # the two extra steps cost more than the overlap saves while the tables are small (they grow with the batch, a decode
        # step does not): D = 2 from 16 images on [measured at 128: 0.726 s instead of 0.825 s]; the constructor's decode_overlap forces
        overlap = B >= 16 if self.decode_overlap is None else bool(self.decode_overlap)
        mode = {'never': 0, 'auto': 1, 'always': 2}[self.rgb_window]
        ws, stats = ops.decode_rgb(P, targets, sym, buf, offs, lens, bounds, K, 2 if overlap else 1, mode,
                                   (getattr(self, '_lane_side', None) or self._side_stream()) if overlap else None)
        self.last_rgb_window_stats = stats      # (development / tests: misses per channel, chunk + 2, image)
        return sym

    # ---- reference API: one image <-> one file -----------------------------------------------------------------------

    def encode(self, img, pout):
        """Encode image to disk at path `pout`.  img: int64 tensor CHW or 1CHW.  Returns the actual bpsp."""
        assert not os.path.isfile(pout)
        if img.dim() == 3:
            img = img.unsqueeze(0)
        assert img.dim() == 4 and img.shape[0] == 1 and img.shape[1] == 3, img.shape
        assert img.dtype == torch.int64, img.dtype

        if auto_crop.needs_crop(img):
            print('Need to encode individual crops!')
            return self._encode_crops(list(auto_crop.iter_crops(img)), pout)

        fac = self.padding_factor()
        _, _, H, W = img.shape
        if H % fac != 0 or W % fac != 0:
            print('*** INFO: image shape ({}X{}) not divisible by {}, will pad.'.format(H, W, fac))
            img, padding_tuple = pad.pad(img, fac=fac, mode=self.blueprint.get_padding_mode())
        else:
            padding_tuple = (0, 0, 0, 0)

        with self.times.run('[-] encode forwardpass'):
            out = self.blueprint.net(img.to('cuda', torch.float32), self.auto_recurse)
        loss_out = self.blueprint.get_loss(out) if self.compare_with_theory else None
        enc = self.encode_batch(img, out=out)
        data = enc.to_bytes([padding_tuple])[0]
        self._write_file(pout, data)

        num_subpixels = int(np.prod(img.shape))
        actual_bpsp = len(data) * 8 / num_subpixels
        if self.compare_with_theory:
            per_scale = [int(n.sum().item()) * 8 / num_subpixels for _, _, _, _, n in enc.scales]
            tostr = lambda l: ' | '.join(map('{:.3f}'.format, l)) + ' => {:.3f}'.format(sum(l))   # noqa: E731
            theory = [float(b) for b in (loss_out.recursive_bpsps if self.auto_recurse else loss_out.nonrecursive_bpsps)]
            overhead = (sum(per_scale) / sum(theory) - 1) * 100
            print('Bitrates:\ntheory:  {}\nassumed: {} [{:.2f}%]\nactual:                                => {:.3f} '
                  '[{} bytes]'.format(tostr(theory), tostr(list(reversed(per_scale))), overhead, actual_bpsp, len(data)))
        return actual_bpsp

    def _encode_crops(self, crops, pout):
        """The auto-crops of a large image (reference :63-71 codes them one after the other): crops of equal padded shape
        share a batch, all batches share ONE grouped coder launch (encode_many); part i goes to `pout`.part<i> as before."""
        fac = self.padding_factor()
        padded, pads, groups = [], [], {}
        for i, crop in enumerate(crops):
            assert not os.path.isfile(pout + part_suffix_helper.make_part_suffix(i))
            _, _, H, W = crop.shape
            if H % fac != 0 or W % fac != 0:
                print('*** INFO: image shape ({}X{}) not divisible by {}, will pad.'.format(H, W, fac))
                crop, pt = pad.pad(crop, fac=fac, mode=self.blueprint.get_padding_mode())
            else:
                pt = (0, 0, 0, 0)
            padded.append(crop)
            pads.append(pt)
            groups.setdefault(tuple(crop.shape[-2:]), []).append(i)
        order = list(groups.values())
        with self.times.run('[-] encode forwardpass + coder, {} crops'.format(len(crops))):
            encs = self.encode_many([torch.cat([padded[i] for i in idxs]).to('cuda', torch.float32) for idxs in order])
        comb = auto_crop.CropLossCombinator()
        sizes = {}
        files = EncodedBatch.many_to_bytes(encs, [[pads[i] for i in idxs] for idxs in order])
        for idxs, datas in zip(order, files):
            for i, data in zip(idxs, datas):
                self._write_file(pout + part_suffix_helper.make_part_suffix(i), data)
                sizes[i] = len(data)
        for i, crop in enumerate(crops):     # as the reference: bpsp of a part over its PADDED sub-pixels, weighted by its area
            comb.add(sizes[i] * 8 / int(np.prod(padded[i].shape)), int(np.prod(crop.shape[-2:])))
        return comb.get_bpsp()

    def _write_file(self, path, data):
        if self.file_writer is not None:
            self.file_writer.submit(path, data)
        else:
            with open(path, 'wb') as fout:
                fout.write(data)

    def _read_file(self, path):
        if self.file_writer is not None:
            self.file_writer.wait(path)
        with open(path, 'rb') as fin:
            return fin.read()

    def _decode_parts(self, paths):
        """Part files of one image: parts of equal (padded) shape are decoded as one batch."""
        datas = [self._read_file(p) for p in paths]
        groups = {}
        for i, d in enumerate(datas):
            groups.setdefault(d[8:13], []).append(i)      # the coarsest scale's header (C, H, W) identifies the padded shape
        parts = [None] * len(datas)
        for idxs in groups.values():
            out, padding = self.decode_batch([datas[i] for i in idxs])
            for k, i in enumerate(idxs):
                o = out[k:k + 1]
                parts[i] = pad.undo_pad(o, *padding[k]) if any(padding[k]) else o
        return parts

    def decode(self, pin, _recurse_part=True):
        """-> decoded image, 1CHW long (on the GPU)."""
        if self.file_writer is not None:
            self.file_writer.wait(pin)                 # (a part suffix is resolved below: every part is waited for when read)
        if _recurse_part and part_suffix_helper.contains_part_suffix(pin):
            parts = self._decode_parts(list(part_suffix_helper.iter_part_suffixes(pin)))
            print('Stitching {} parts...'.format(len(parts)))
            return auto_crop.stitch(parts)
        data = self._read_file(pin)
        out, padding = self.decode_batch([data])
        if any(padding[0]):
            out = pad.undo_pad(out, *padding[0])
        return out


_RGB_MEAN_T = {}


def _rgb_mean_tensor(device):
    """(1,3,1,1) fp32 tensor of (0.4488, 0.4371, 0.4040) * 255 -- the same fp32 values the encoder side subtracts (ops.rgb_mean)."""
    key = str(device)
    if key not in _RGB_MEAN_T:
        _RGB_MEAN_T[key] = torch.tensor([float(v) for v in ops.rgb_mean()], dtype=torch.float32, device=device).reshape(1, 3, 1, 1)
    return _RGB_MEAN_T[key]


class ParsedContainers(object):
    """Framing of B `.l3c` files of equally sized images (`parse_containers`): padding tuples, per scale record (coarsest first) its
    (C, H, W) and, as (B, C) arrays, where every channel's payload lies inside its file (`offset`) and how long it is (`nbytes`)."""

    def __init__(self, padding, scales, offset, nbytes):
        self.padding, self.scales, self.offset, self.nbytes = padding, scales, offset, nbytes


def parse_containers(files):
    """The byte format of bitcoding.py:326-375 -- u16 x4 padding | per scale (coarsest first): u8 C, u16 H, u16 W | per channel u32 n +
    payload | 46 E2 84 92 -- walked by its length fields only.  ValueError on broken framing or when the files disagree in shape."""
    B = len(files)
    padding, scales, offset, nbytes = [], None, None, None
    for b, f in enumerate(files):
        n_rec = count_scale_records(f)
        if scales is None:
            scales = [None] * n_rec
            offset, nbytes = [None] * n_rec, [None] * n_rec
        elif n_rec != len(scales):
            raise ValueError('decode_batch needs equally sized images: {} vs {} scale records'.format(n_rec, len(scales)))
        padding.append(struct.unpack_from('<4H', f, 0))
        p = 8
        for k in range(n_rec):
            shape = struct.unpack_from('<BHH', f, p)
            p += 5
            if scales[k] is None:
                scales[k] = shape
                offset[k] = np.zeros((B, shape[0]), dtype=np.int64)
                nbytes[k] = np.zeros((B, shape[0]), dtype=np.int64)
            elif shape != scales[k]:
                raise ValueError('decode_batch needs equally sized images, got shapes {}'.format(sorted({shape, scales[k]})))
            for c in range(shape[0]):
                n, = struct.unpack_from('<I', f, p)
                offset[k][b, c] = p + 4
                nbytes[k][b, c] = n
                p += 4 + n
            p += 4                                   # the separator (count_scale_records has checked it)
    return ParsedContainers(padding, scales, offset, nbytes)


class _H2DRing(object):
    """Page-locked staging buffers for uploads, round robin; a buffer is reused only after the copy that read it has completed."""

    def __init__(self, n=3):
        self.bufs, self.events, self.turn = [None] * n, [None] * n, 0

    def take(self, nbytes):
        k = self.turn = (self.turn + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
            self.events[k] = None
        if self.bufs[k] is None or self.bufs[k].numel() < nbytes:
            self.bufs[k] = torch.empty(max(nbytes, 2 * (self.bufs[k].numel() if self.bufs[k] is not None else 0), 8 << 20),
                                       dtype=torch.uint8, pin_memory=True)
        return k, self.bufs[k][:nbytes]

    def sent(self, k):
        self.events[k] = torch.cuda.Event()
        self.events[k].record(torch.cuda.current_stream())


_UPLOAD_RING = _H2DRing(6)
_UPLOAD_STREAM = [None]      # the files of a batch cross PCIe on a stream of their own: a lane's upload never queues behind that lane's previous batch


class _DeviceStreams(object):
    """The entropy-coded streams of a batch of files on the device, 4-byte aligned and zero padded, all scales in one buffer.
    The streams of the LAST scale record (the finest scale: ~95 % of a file's bytes) may still be on the host: `finish()` -- called by
    `scale(last)`, or by the set decoder before its last phase -- stages, uploads and cuts them out then, under the stream that is current."""

    def __init__(self, buf, offs, lens, first, count, offs_host=None, lens_host=None, pending=None):
        self.buf, self.offs, self.lens, self.first, self.count = buf, offs, lens, first, count
        self.offs_host, self.lens_host = offs_host, lens_host       # (numpy: the set decoder merges the tables of several batches on the host)
        self.pending = pending

    def finish(self):
        if self.pending is not None:
            pending, self.pending = self.pending, None
            pending()

    def scale_host(self, k):
        a, n = self.first[k], self.count[k]
        return self.offs_host[a:a + n], self.lens_host[a:a + n]

    def scale(self, k):
        """(buffer, offsets int64, lengths int32) of scale record k: the coarsest record in image-major order (stream b * C + c, what
        the uniform-prior decoder writes as (B, C, H, W)), the others channel-major (stream c * B + b: a channel's B streams adjacent)."""
        if k == len(self.first) - 1:
            self.finish()
        a, n = self.first[k], self.count[k]
        return self.buf, self.offs[a:a + n], self.lens[a:a + n]


def _upload_stream():
    if _UPLOAD_STREAM[0] is None:
        _UPLOAD_STREAM[0] = torch.cuda.Stream()
    return _UPLOAD_STREAM[0]


def _h2d(stage, k, dev_slice):
    """One asynchronous copy of a staging buffer of the upload ring on the upload stream; the current stream waits for it."""
    cur = torch.cuda.current_stream()
    with torch.cuda.stream(_upload_stream()):
        dev_slice.copy_(stage, non_blocking=True)
        _UPLOAD_RING.sent(k)
        copied = torch.cuda.Event()
        copied.record(_UPLOAD_STREAM[0])
    cur.wait_event(copied)


def _upload_streams(files, parsed):
    """Files -> _DeviceStreams on the current stream: page-locked staging, asynchronous H2D copies on the upload stream, l3c_container_read cuts
    the streams out on the device.  No payload byte is touched by Python.
    In TWO parts: everything before a file's last scale record (a twentieth of its bytes) and the stream table now, the last record's streams
    when they are asked for (`_DeviceStreams.finish`): the host's staging copy of the bulk (20 ms for a batch of 128) then runs while the
    GPU is busy with the coarse scales, not before its first kernel [measured, same box: 0.366 -> see DESIGN 7.2]."""
    B = len(files)
    sizes = np.asarray([len(f) for f in files], dtype=np.int64)
    last = len(parsed.scales) - 1
    cut = parsed.offset[last][:, 0] - 9                     # the last record's header: u8 C, u16 H, u16 W, then channel 0's u32 length
    al = lambda n: (n + 3) // 4 * 4                         # noqa: E731 -- file pieces 4-byte aligned in the buffer
    base_a = np.concatenate([[0], np.cumsum(al(cut))]).astype(np.int64)
    base_b = np.concatenate([[0], np.cumsum(al(sizes - cut))]).astype(np.int64)
    src, dst_len, first, count = [], [], [], []
    for k, (C, H, W) in enumerate(parsed.scales):
        o = parsed.offset[k] + (base_a[:B, None] if k < last else (base_b[:B] - cut)[:, None])     # (the last record: relative to part B's start, added below)
        n = parsed.nbytes[k]
        if k:                                        # channel-major
            o, n = o.T, n.T
        first.append(sum(count))
        count.append(B * C)
        src.append(o.reshape(-1))
        dst_len.append(n.reshape(-1))
    S_a = first[last]
    lens = np.concatenate(dst_len)
    padded = (lens + 3) // 4 * 4 + 4
    dst = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    S = lens.shape[0]
    a_bytes = int(base_a[-1])
    table_at = (a_bytes + 7) // 8 * 8
    b_at = table_at + (S * 20 + 7) // 8 * 8
    b_bytes = int(base_b[-1])
    src[last] = src[last] + b_at
    src = np.concatenate(src)
    k, stage = _UPLOAD_RING.take(b_at)
    st = stage.numpy()
    for b, f in enumerate(files):
        st[base_a[b]:base_a[b] + cut[b]] = np.frombuffer(f, dtype=np.uint8, count=int(cut[b]))
    st[table_at:table_at + 8 * S] = src.view(np.uint8)
    st[table_at + 8 * S:table_at + 16 * S] = dst.view(np.uint8)
    st[table_at + 16 * S:table_at + 20 * S] = lens.astype(np.int32).view(np.uint8)
    cur = torch.cuda.current_stream()
    with torch.cuda.stream(_upload_stream()):       # (the upload stream's pool: the copies that fill it are ordered behind that block's previous use)
        dev = torch.empty(b_at + b_bytes, dtype=torch.uint8, device='cuda')
    dev.record_stream(cur)
    _h2d(stage, k, dev[:b_at])
    src_d = dev[table_at:table_at + 8 * S].view(torch.int64)
    dst_d = dev[table_at + 8 * S:table_at + 16 * S].view(torch.int64)
    len_d = dev[table_at + 16 * S:table_at + 20 * S].view(torch.int32)
    out = torch.empty(int(padded.sum()), dtype=torch.uint8, device='cuda')
    ops.container_read(dev, src_d[:S_a], dst_d[:S_a], len_d[:S_a], int(lens[:S_a].max()), out)

    def finish():
        k2, stage2 = _UPLOAD_RING.take(b_bytes)
        st2 = stage2.numpy()
        for b, f in enumerate(files):
            st2[base_b[b]:base_b[b] + sizes[b] - cut[b]] = np.frombuffer(f, dtype=np.uint8, offset=int(cut[b]))
        now = torch.cuda.current_stream()
        if now != cur:
            dev.record_stream(now)
            out.record_stream(now)
        _h2d(stage2, k2, dev[b_at:])
        ops.container_read(dev, src_d[S_a:], dst_d[S_a:], len_d[S_a:], int(lens[S_a:].max()), out)

    return _DeviceStreams(out, dst_d, len_d, first, count, dst, lens, finish)


def count_scale_records(data):
    """Number of scale records of a `.l3c` byte string (u8 C, u16 H, u16 W, C x (u32 n + payload), magic); ValueError if the
    framing is broken.  An L3C file has num_scales + 1 of them; an RGB Shared file one more per recursion."""
    r = _Reader(data)
    r.take(8)
    n = 0
    while r.p < len(data):
        C, _, _ = r.unpack('<BHH')
        for _ in range(C):
            nb, = r.unpack('<I')
            r.take(nb)
        if r.take(4) != _MAGIC_VALUE_SEP:
            raise ValueError('invalid file: scale separator missing')
        n += 1
    if n < 2:
        raise ValueError('invalid file: {} scale record(s)'.format(n))
    return n


class AsyncFileWriter(object):
    """Writes finished `.l3c` byte strings on worker threads (file output is not part of the hot path: reference bitcoding.py:
    326-375 writes from the coding loop).  `wait(path)` blocks until a pending write of `path` is on disk; `close()` drains."""

    def __init__(self, n_threads=2):
        import concurrent.futures
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix='l3c-write')
        self._pending = {}

    @staticmethod
    def _write(path, data):
        with open(path, 'wb') as f:
            f.write(data)

    def submit(self, path, data):
        self.wait(path)
        self._pending[path] = self._pool.submit(self._write, path, data)

    def wait(self, path=None):
        for p in ([path] if path is not None else list(self._pending)):
            fut = self._pending.pop(p, None)
            if fut is not None:
                fut.result()

    def pending(self):
        return len(self._pending)

    def close(self):
        self.wait()
        self._pool.shutdown()


class _Reader(object):
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        b = self.d[self.p:self.p + n]
        if len(b) != n:
            raise ValueError('invalid file: truncated')
        self.p += n
        return b

    def unpack(self, fmt):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))
