"""MultiscaleTester -- the reference's test/multiscale_tester.py on the MI355X path: experiment loading (`__init__` :160-220),
bpsp evaluation with a result cache (`test_all` / `test` / `_test` :236-351, `TestOutputCache` :67-102), the
`--write_to_files` encode -> decode -> assert round trip with a time report (`_write_to_file` :353-381), and single-image
`encode` / `decode` (:383-408, `_read_img` / `_write_img` :410-434).

Evaluation is batched where the reference is image-by-image: crops of equal padded shape go through one forward; bpsp is
still per image (per-image sums of the NLL maps), combined over auto-crops by area like `CropLossCombinator`."""
import collections
import fcntl
import os
import pickle
import time

import numpy as np
import torch
from PIL import Image

from .. import auto_crop
from ..bitcoding import part_suffix_helper
from ..bitcoding.bitcoding import AsyncFileWriter, Bitcoding
from ..blueprints.multiscale_blueprint import MultiscaleBlueprint
from ..helpers import config_parser, paths

_FILE_EXT = '.l3c'
TestID = collections.namedtuple('TestID', ['dataset_id', 'restore_itr'])


class TestResult(object):
    """filename -> metric value of one test set (reference :104-121)."""

    def __init__(self, metric_name='bpsp'):
        self.metric_name = metric_name
        self.per_img = collections.OrderedDict()

    def __setitem__(self, filename, value):
        self.per_img[filename] = float(value)

    def __len__(self):
        return len(self.per_img)

    def mean(self):
        return float(np.mean(list(self.per_img.values())))


class TestOutputCache(object):
    """Pickle file of {TestID: TestResult}, guarded by an inter-process file lock (reference :67-102)."""

    def __init__(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.pickle_p = os.path.join(out_dir, 'cache.pkl')
        self.lock_p = os.path.join(out_dir, '.cache.lock')

    def _locked(self, fn):
        with open(self.lock_p, 'w') as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                return fn()
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)

    def _read(self):
        if not os.path.isfile(self.pickle_p):
            return {}
        with open(self.pickle_p, 'rb') as f:
            return pickle.load(f)

    def __contains__(self, test_id):
        return self._locked(lambda: test_id in self._read())

    def __getitem__(self, test_id):
        return self._locked(lambda: self._read()[test_id])

    def __setitem__(self, test_id, result):
        def write():
            cache = self._read()
            cache[test_id] = result
            with open(self.pickle_p, 'wb') as f:
                pickle.dump(cache, f)
        self._locked(write)

    def reset(self):
        self._locked(lambda: os.path.isfile(self.pickle_p) and os.remove(self.pickle_p))


class StackTimeLogger(object):
    """Named wall-clock timers around device work (synchronising before and after, like test/cuda_timer.py:29-35, :107-151)."""

    def __init__(self):
        self.times = collections.OrderedDict()
        self.last = collections.OrderedDict()
        self._skip = False

    class _Run(object):
        def __init__(self, logger, name):
            self.logger, self.name = logger, name

        def __enter__(self):
            torch.cuda.synchronize()
            self.t0 = time.time()

        def __exit__(self, *a):
            torch.cuda.synchronize()
            dt = time.time() - self.t0
            self.logger.last[self.name] = dt
            if not self.logger._skip:
                self.logger.times.setdefault(self.name, []).append(dt)
            return False

    def run(self, name):
        return self._Run(self, name)

    prefix_scope = combine = run

    def skip(self, flag):
        """first image = warm-up: measured but not averaged (multiscale_tester.py:297)"""
        logger = self

        class _Skip(object):
            def __enter__(self_):
                logger._skip = flag

            def __exit__(self_, *a):
                logger._skip = False
                return False
        return _Skip()

    def get_last_strs(self):
        return ['{}: {:.5f}'.format(k, v) for k, v in self.last.items()]

    def get_mean_strs(self):
        return ['{}: {:.5f} (n={})'.format(k, float(np.mean(v)), len(v)) for k, v in self.times.items()]


class EncodeError(Exception):
    pass


class DecodeError(Exception):
    pass


class ImageSaver(object):
    """test/image_saver.py:28-60: tensors with values in 0..255 (1CHW / CHW; truncated to uint8 like the reference's
    `.type(torch.uint8)`) -> PNG files in one directory."""

    def __init__(self, out_dir):
        self.out_dir = out_dir
        os.makedirs(self.out_dir, exist_ok=True)
        self.saved_fs = []

    def __str__(self):
        return 'ImageSaver({})'.format(self.out_dir)

    def save_img(self, img, filename):
        img = img.detach().to('cpu').type(torch.uint8)
        if img.dim() == 4:
            assert img.shape[0] == 1, img.shape
            img = img[0]
        out_p = self.get_save_p(filename)
        Image.fromarray(np.ascontiguousarray(img.permute(1, 2, 0).numpy())).save(out_p)
        return out_p

    def get_save_p(self, file_name):
        self.saved_fs.append(file_name)
        return os.path.join(self.out_dir, file_name)

    def file_starting_with_exists(self, prefix):
        import glob
        return len(glob.glob(os.path.join(self.out_dir, prefix) + '*')) > 0


class MultiscaleTester(object):
    def __init__(self, log_date, flags, restore_itr, l3c=False, configs_dir=None):
        """flags needs `.log_dir`; optional `.compare_theory`."""
        self.flags = flags
        self.log_date = log_date
        experiment_dir = paths.get_experiment_dir(flags.log_dir, log_date)
        configs_dir = configs_dir or config_parser.CONFIG_DIR
        (config_p_ms, _), postfix = paths.parse_log_dir(experiment_dir, configs_dir)
        self.config_ms, _ = config_parser.parse(config_p_ms)
        if postfix:   # "key=value" overrides carried by the directory name (global_config.py:73-97)
            for kv in postfix:
                if '=' in kv:
                    k, v = kv.split('=', 1)
                    self.config_ms.set_attr(k, config_parser._eval_value(v))
        self.blueprint = MultiscaleBlueprint(self.config_ms)
        self.blueprint.set_eval()
        self.restore_itr, ckpt_p = paths.get_ckpt_for_itr(paths.get_ckpts_dir(experiment_dir), restore_itr)
        paths.restore({'net': self.blueprint.net}, ckpt_p, strict=True)
        self.times = StackTimeLogger()
        self.recursive = self._parse_recursive_flag(getattr(flags, 'recursive', '0'), self.config_ms)
        # --write_to_files works for --recursive too (the reference raises NotImplementedError, :187-188): the `.l3c` layout carries
        # one more scale record per recursion.  Finished files are written by worker threads.
        self.file_writer = AsyncFileWriter() if getattr(flags, 'write_to_files', None) else None
        self.bc = Bitcoding(self.blueprint, times=self.times if getattr(flags, 'write_to_files', None) else None,
                            compare_with_theory=bool(getattr(flags, 'compare_theory', False)), auto_recurse=self.recursive,
                            file_writer=self.file_writer)
        self.max_batch = int(getattr(flags, 'batch', None) or 8)
        self.io_threads = int(getattr(flags, 'io_threads', None) or 4)
        exp_name = os.path.basename(experiment_dir)
        self.out_dir = os.path.join(flags.log_dir.rstrip(os.path.sep) + '_test', exp_name)
        self.test_output_cache = TestOutputCache(self.out_dir)
        if getattr(flags, 'reset_entire_cache', False):
            self.test_output_cache.reset()

    # ---- bpsp evaluation ------------------------------------------------------------------------------------------------

    @staticmethod
    def _parse_recursive_flag(flag, config_ms):
        """'auto' -> 3 for the RGB Shared baseline (num_scales == 1), 0 otherwise; a number is taken as is (reference :123-132)."""
        flag = str(flag or '0')
        if flag == 'auto':
            return 3 if (config_ms.rgb_bicubic_baseline and config_ms.num_scales == 1) else 0
        r = int(flag)
        if r and not (config_ms.rgb_bicubic_baseline and config_ms.num_scales == 1):
            raise ValueError('--recursive only makes sense for the RGB Shared baseline (num_scales == 1)')
        return r

    def _padding_fac(self):
        return 2 ** (self.config_ms.num_scales + self.recursive)

    def test_all(self, testsets):
        results = [self.test(testset) for testset in testsets]
        if getattr(self.flags, 'write_to_files', None):
            return [None]
        return [(testset, self.log_date, self.restore_itr, '{}={}'.format(r.metric_name, r.mean()))
                for testset, r in zip(testsets, results)]

    def test(self, testset):
        test_id = TestID(testset.id, self.restore_itr)
        write = getattr(self.flags, 'write_to_files', None)
        if not getattr(self.flags, 'overwrite_cache', False) and not write and test_id in self.test_output_cache:
            print('*** Found cached: {}'.format(test_id))
            return self.test_output_cache[test_id]
        print('Testing {}'.format(testset))
        with torch.no_grad():
            result = self._test_write(testset) if write else self._test(testset)
        if write:
            return None
        self.test_output_cache[test_id] = result
        return result

    def _iter_images(self, ps, ahead=None):
        """(index, path, uint8 CHW tensor) in order; the files are read and decoded by `io_threads` worker threads `ahead` images
        (default: two per thread) ahead of the consumer (PIL releases the GIL while it inflates a PNG: reference
        images_loader.py:91-129 decodes on the main thread, one image at a time)."""
        import concurrent.futures
        ahead = max(2 * self.io_threads, ahead or 0)
        with concurrent.futures.ThreadPoolExecutor(max_workers=self.io_threads, thread_name_prefix='l3c-read') as pool:
            futs = collections.deque()
            it = iter(enumerate(ps))
            for i, p in it:
                futs.append((i, p, pool.submit(self._load_uint8, p)))
                if len(futs) >= ahead:
                    break
            while futs:
                i, p, f = futs.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    futs.append((nxt[0], nxt[1], pool.submit(self._load_uint8, nxt[1])))
                yield i, p, f.result()

    def _load_uint8(self, img_p):
        img = Image.open(img_p)
        crop = getattr(self.flags, 'crop', None)
        if crop:
            w, h = img.size
            l, t = (w - crop) // 2, (h - crop) // 2
            img = img.crop((l, t, l + crop, t + crop))
        arr = np.array(img)
        if arr.ndim == 2:
            arr = np.stack([arr] * 3, axis=-1)
        return torch.from_numpy(np.ascontiguousarray(arr[..., :3].transpose(2, 0, 1)))

    def per_image_bpsp(self, out, num_subpixels_before_pad):
        """(B,) bpsp of a forward: per-image sums of the NLL maps + the uniform cost of the coarsest scale
        (multiscale_network.Losses.get :145-165, blueprint.get_loss :64-95, evaluated per batch item)."""
        nats = None
        for loss, target, P in out.iter_targets_and_predictions(self.blueprint.losses.loss_dmol_rgb,
                                                                self.blueprint.losses.loss_dmol_n):
            n = loss(target, P).sum(dim=(1, 2, 3))
            nats = n if nats is None else nats + n     # recursive evaluation sums every scale (reference :330-333)
        _, C, H, W = out.S[-1].shape
        nats = nats + C * H * W * float(np.log(out.L[-1]))
        conv = torch.tensor([np.log(2.) * n for n in num_subpixels_before_pad], dtype=torch.float64, device=nats.device)
        return (nats.double() / conv).cpu().numpy()

    def _test_sample(self, testset):
        """--sample OUT_DIR (reference :276-282, :327-328, :436-448): per image the ground truth and three sampled images --
        RGB only, RGB + z1, RGB + z1 + z2 -- named with the bpsp of the scales that were NOT sampled."""
        test_result = TestResult('bpsp')
        image_saver = ImageSaver(os.path.join(self.flags.sample, self.log_date))
        print('Will store samples in {}.'.format(image_saver.out_dir))
        for i, img_p in enumerate(testset.ps):
            filename = os.path.splitext(os.path.basename(img_p))[0]
            raw = self._load_uint8(img_p).unsqueeze(0)
            img_batch = MultiscaleBlueprint.pad(raw, self._padding_fac()).to('cuda', torch.float32)
            out = self.blueprint.forward(img_batch)
            loss_out = self.blueprint.get_loss(out, num_subpixels_before_pad=int(np.prod(raw.shape)))
            test_result[filename] = float(sum(loss_out.nonrecursive_bpsps))
            self._sample([float(b) for b in loss_out.nonrecursive_bpsps], img_batch, image_saver, '{}_{}'.format(i, filename))
            print('{}: {} ({: 10d}): mean {}={}'.format(self.log_date, filename, i, test_result.metric_name, test_result.mean()))
        return test_result

    def _sample(self, bpsps, img_batch, image_saver, save_prefix):
        if image_saver.file_starting_with_exists(save_prefix):
            raise FileExistsError('Previous sample outputs found in {}. Please remove.'.format(image_saver.out_dir))
        image_saver.save_img(img_batch, '{}_{:.3f}_gt.png'.format(save_prefix, sum(bpsps)))
        for style, sample_scales in (('rgb', []),               # sample the RGB scale only
                                     ('rgb+bn0', [0]),          # RGB + z^(1)
                                     ('rgb+bn0+bn1', [0, 1])):  # RGB + z^(1) + z^(2)
            sampled = self.blueprint.sample_forward(img_batch, sample_scales)
            bpsp_sample = sum(bpsps[len(sample_scales) + 1:])
            image_saver.save_img(sampled, '{}_{}_{:.3f}.png'.format(save_prefix, style, bpsp_sample))

    def _test(self, testset):
        if getattr(self.flags, 'sample', None):
            return self._test_sample(testset)
        test_result = TestResult('bpsp recursive' if self.recursive else 'bpsp')
        # Streaming evaluation: the auto-crops of the images read so far wait, grouped by padded shape, until a group fills a batch
        # (equal shapes share one forward) or until its oldest image falls `window` images behind the reader; an image's result is
        # combined (area-weighted over its crops, auto_crop.py:139-152) as soon as its last crop is through, and reported in input
        # order like the reference's one-image-at-a-time loop (multiscale_tester.py:322-343).  Memory and latency are bounded by the
        # window, not by the size of the set.
        window = 8 * self.max_batch
        pending = collections.defaultdict(list)        # padded shape -> [(image index, number of sub-pixels, padded crop)]
        per_img = {}                                   # image index -> [CropLossCombinator, crops still to come, file name]
        state = {'next': 0}

        def run(shape):
            chunk, pending[shape] = pending[shape][:self.max_batch], pending[shape][self.max_batch:]
            if not pending[shape]:
                del pending[shape]
            batch = torch.cat([c[2] for c in chunk], dim=0).cuda().float()      # bytes over PCIe, widened on the device
            out = self.blueprint.forward(batch, self.recursive)
            for (i, n_sub, _), b in zip(chunk, self.per_image_bpsp(out, [c[1] for c in chunk])):
                per_img[i][0].add(float(b), n_sub)
                per_img[i][1] -= 1

        def report():
            while state['next'] in per_img and per_img[state['next']][1] == 0:
                comb, _, filename = per_img.pop(state['next'])
                test_result[filename] = comb.get_bpsp()
                print('{}: {} ({: 10d}): mean {}={}'.format(self.log_date, filename, state['next'], test_result.metric_name,
                                                            test_result.mean()))
                state['next'] += 1

        for i, img_p, raw in self._iter_images(testset.ps):
            crops = list(auto_crop.iter_crops(raw.unsqueeze(0)))
            per_img[i] = [auto_crop.CropLossCombinator(), len(crops), os.path.splitext(os.path.basename(img_p))[0]]
            for crop in crops:
                padded = MultiscaleBlueprint.pad(crop, self._padding_fac())
                shape = tuple(padded.shape[-2:])
                pending[shape].append((i, int(np.prod(crop.shape)), padded))
                if len(pending[shape]) >= self.max_batch:
                    run(shape)
            while i - state['next'] >= window and per_img[state['next']][1] > 0:     # the oldest image has waited long enough
                run(next(sh for sh, items in pending.items() if any(c[0] == state['next'] for c in items)))
                report()
            report()
        while pending:
            run(next(iter(pending)))
        report()
        assert not per_img
        return test_result

    # ---- --write_to_files: real files, real round trip ---------------------------------------------------------------------

    def _test_write(self, testset):
        """--write_to_files: every image -> a real `.l3c` file -> read back -> decoded -> compared with the input (reference :353-381,
        one image at a time: encode, decode, assert_equal :373).  Here the images stream through in WINDOWS of `--write_window` images
        (default 32 x --batch): a window is coded as a set (dataset_codec.encode_set: images of equal padded shape share a forward pass,
        one grouped coder launch) -- planned from the image files' HEADERS, the pixels read and decoded by the reader threads in the order
        the passes are enqueued, i.e. while the GPU works on the earlier passes --, its files are written by worker threads, read back
        from disk and decoded as a set (decode_set).  Images that need auto-crops (`.partN` files) and a window of 1 take the
        reference's one-image path.  The time report (`--time_report`) names every stage, per window."""
        import concurrent.futures
        from ..helpers import dataset_codec
        test_result = TestResult('bpsp')
        out_dir = self.flags.write_to_files
        os.makedirs(out_dir, exist_ok=True)
        window = int(getattr(self.flags, 'write_window', None) or 32 * self.max_batch)
        # the first window is the warm-up (reference: the first image, :297) -- when there is more than one
        skip_first = bool(getattr(self.flags, 'skip_first_window', True)) and len(testset.ps) > window
        if window <= 1:
            for i, img_p, raw in self._iter_images(testset.ps):       # the next images are decoded while this one is on the GPU
                self._write_one(i, img_p, raw, out_dir, test_result)
        else:
            with concurrent.futures.ThreadPoolExecutor(max_workers=self.io_threads, thread_name_prefix='l3c-read') as pool:
                buf, n_windows = [], 0
                for i, img_p in enumerate(testset.ps):
                    shape = self._image_shape(img_p)
                    if auto_crop.needs_crop(torch.empty((1, 3) + shape, dtype=torch.uint8, device='meta')):
                        self._write_one(i, img_p, self._load_uint8(img_p), out_dir, test_result)
                        continue
                    buf.append((i, img_p, shape))
                    if len(buf) >= window or i + 1 == len(testset.ps):
                        with self.times.skip(skip_first and n_windows == 0):
                            self._write_window(buf, out_dir, test_result, dataset_codec, pool)
                        n_windows += 1
                        buf = []
                if buf:
                    with self.times.skip(skip_first and n_windows == 0):
                        self._write_window(buf, out_dir, test_result, dataset_codec, pool)
        if self.file_writer is not None:
            self.file_writer.wait()
        if getattr(self.flags, 'time_report', None):
            with open(self.flags.time_report, 'w') as f:
                f.write('Average times:\n')
                f.write('\n'.join(self.times.get_mean_strs()))
        return test_result

    def _image_shape(self, img_p):
        """(H, W) as `_load_uint8` will return it, from the file's header alone (PIL opens lazily: no pixel is decoded)."""
        with Image.open(img_p) as img:
            w, h = img.size
        crop = getattr(self.flags, 'crop', None)
        return (crop, crop) if crop else (h, w)

    def _write_one(self, i, img_p, raw, out_dir, test_result):
        filename = os.path.splitext(os.path.basename(img_p))[0]
        print('***', filename)
        img = raw.unsqueeze(0).long()
        with self.times.skip(i == 0):
            test_result[filename] = self._write_to_file(img, os.path.join(out_dir, filename + _FILE_EXT))
        print('{}: {} ({: 10d}): mean {}={}'.format(self.log_date, filename, i, test_result.metric_name, test_result.mean()))

    def _note(self, name, seconds):
        self.times.last[name] = seconds
        if not self.times._skip:
            self.times.times.setdefault(name, []).append(seconds)

    def _write_window(self, buf, out_dir, test_result, dataset_codec, pool):
        order = [i for i, _, _ in buf]
        shapes = {i: shape for i, _, shape in buf}
        src = {i: img_p for i, img_p, _ in buf}
        names = {i: os.path.splitext(os.path.basename(img_p))[0] for i, img_p, _ in buf}
        paths_ = {i: os.path.join(out_dir, names[i] + _FILE_EXT) for i in order}
        fac = self._padding_fac()
        for i in order:
            for stale in (paths_[i], paths_[i] + part_suffix_helper.make_part_suffix(0)):
                if os.path.isfile(stale):
                    os.remove(stale)
        imgs, marks = {}, {}
        with self.times.run('=== bc.encode, {} image files as a set (read + PNG decode on {} threads || H2D, forward, heads, coder, file assembly, D2H)'.format(
                len(order), self.io_threads)):
            files, _, _ = dataset_codec.encode_set(self.bc, imgs, order, max_batch=self.max_batch, fac=fac, shapes=shapes,
                                                   loader=lambda i: self._load_uint8(src[i]).contiguous(), pool=pool, marks=marks)
        self._note('    of which the host waited for the image readers', marks.get('host seconds', {}).get('wait for the image readers', 0.0))
        with self.times.run('=== file write ({} files, worker threads, waited for)'.format(len(order))):
            for i in order:
                self.bc._write_file(paths_[i], files[i])
            if self.file_writer is not None:
                self.file_writer.wait()
        if getattr(self.flags, 'round_trip', True) is False:      # (bench.py --config files: the encode half alone, image files -> .l3c files)
            for i in order:
                H, W = shapes[i]
                test_result[names[i]] = len(files[i]) * 8 / float(3 * (-(-H // fac) * fac) * (-(-W // fac) * fac))
            return
        with self.times.run('=== file read ({} files)'.format(len(order))):
            datas = {i: self.bc._read_file(paths_[i]) for i in order}
        with self.times.run('=== bc.decode, {} files as a set (parse, H2D, decode, D2H)'.format(len(order))):
            back = dataset_codec.decode_set(self.bc, datas, order, max_batch=self.max_batch)
        with self.times.run('=== compare with the inputs'):
            for i in order:
                if not torch.equal(back[i], imgs[i]):
                    raise AssertionError('decoded image differs from the input: {}'.format(paths_[i]))
        for i in order:
            H, W = shapes[i]
            Hp, Wp = -(-H // fac) * fac, -(-W // fac) * fac
            test_result[names[i]] = len(datas[i]) * 8 / float(3 * Hp * Wp)      # as the reference: over the PADDED sub-pixels (bitcoding.py:108-110)
            print('{}: {} ({: 10d}): mean {}={}'.format(self.log_date, names[i], i, test_result.metric_name, test_result.mean()))
        print('\n'.join(self.times.get_last_strs()))

    def _write_to_file(self, img, out_p):
        for stale in [out_p] + ([p for p in part_suffix_helper.iter_part_suffixes(out_p + '.part0')]
                                if os.path.isfile(out_p + '.part0') else []):
            if os.path.isfile(stale):
                os.remove(stale)
        with self.times.run('=== bc.encode'):
            bpsp = self.bc.encode(img, pout=out_p)
        if self.file_writer is not None:       # the round trip below reads the file back: here the write has to be waited for
            with self.times.run('=== file write (worker thread, waited for)'):
                self.file_writer.wait()
        out_p_part = out_p + part_suffix_helper.make_part_suffix(0)
        if not os.path.isfile(out_p) and os.path.isfile(out_p_part):
            out_p = out_p_part
        with self.times.run('=== bc.decode'):
            img_o = self.bc.decode(pin=out_p)
        if not torch.equal(img_o.cpu(), img.cpu()):
            raise AssertionError('decoded image differs from the input: {}'.format(out_p))
        print('\n'.join(self.times.get_last_strs()))
        if getattr(self.flags, 'time_report', None):
            with open(self.flags.time_report, 'w') as f:
                f.write('Average times:\n')
                f.write('\n'.join(self.times.get_mean_strs()))
        return bpsp

    def encode(self, img_p, pout, overwrite=False):
        pout_dir = os.path.dirname(os.path.abspath(pout))
        if not os.path.isdir(pout_dir):
            raise EncodeError('pout directory ({}) does not exists!'.format(pout_dir))
        if overwrite and os.path.isfile(pout):
            print('Removing {}...'.format(pout))
            os.remove(pout)
        if os.path.isfile(pout):
            raise EncodeError('{} exists. Consider --overwrite'.format(pout))
        img = self._read_img(img_p)
        bpsp = self.bc.encode(img, pout=pout)
        from .. import _lib
        # the container has no version field (reference bitcoding.py:326-375): say which decoder generation reads this file
        print('---\nSaved: {}  (bitstream generation {}, include/l3c_hip.h)'.format(pout, _lib.load().l3c_bitstream_generation()))
        return bpsp

    def decode(self, pin, png_out_p):
        pout_dir = os.path.dirname(os.path.abspath(png_out_p))
        if not os.path.isdir(pout_dir):
            raise DecodeError('png_out_p directory ({}) does not exists!'.format(pout_dir))
        if not png_out_p.endswith('.png'):
            raise DecodeError('png_out_p must end in .png, got {}'.format(png_out_p))
        decoded = self.bc.decode(pin)
        self._write_img(decoded, png_out_p)
        print('---\nDecoded: {}'.format(png_out_p))

    @staticmethod
    def _read_img(img_p):
        img = np.array(Image.open(img_p))
        if img.ndim != 3:
            raise EncodeError('Image has {} dimensions, expected HxWxC.'.format(img.ndim))
        img = img.transpose(2, 0, 1)
        C = img.shape[0]
        if C == 4:
            print('*** WARN: Will discard 4th (alpha) channel.')
            img = img[:3, ...]
        elif C != 3:
            raise EncodeError('Image has {} channels, expected 3 or 4.'.format(C))
        return torch.from_numpy(np.ascontiguousarray(img)).unsqueeze(0).long()

    @staticmethod
    def _write_img(decoded, png_out_p):
        assert decoded.shape[0] == 1 and decoded.shape[1] == 3, decoded.shape
        img = decoded.squeeze(0).cpu().numpy().transpose(1, 2, 0).astype(np.uint8)
        Image.fromarray(img).save(png_out_p)
