"""Host-side logic (no GPU): config parser, padding, auto-crop, part suffixes, experiment-dir conventions, the uniform table,
the schema, container framing."""
import os

import pytest
import torch

import l3c_pytorch_amd  # noqa: F401
from l3c_pytorch_amd import auto_crop
from l3c_pytorch_amd.bitcoding import part_suffix_helper as psh
from l3c_pytorch_amd.bitcoding.bitcoding import uniform_cdf_row
from l3c_pytorch_amd.helpers import config_parser, pad, paths, synthetic
from l3c_pytorch_amd.modules import quantizer, schema


def test_config_parser_inheritance_and_api(tmp_path):
    cfg = config_parser.parse_builtin('ms', 'cr_rgb')
    assert cfg.num_scales == 3 and cfg.q.C == 3 and cfg.q.L == 5 and cfg.dec.skip is True
    assert cfg.enc.cls == 'BicubicSubsampling' and cfg.prob.K == 10       # inherited through two `use` levels
    kv = dict(cfg.all_params_and_values())
    assert kv['q.levels_range'] == (-1, 1) and kv['lr.initial'] == 0.0001
    cfg.set_attr('q.L', 7)
    assert cfg.q.L == 7
    p = tmp_path / 'x.cf'
    p.write_text('a = 1\nb.c = "x"  # comment\n# full comment\nbad line\n')
    with pytest.raises(ValueError):
        config_parser.parse(str(p))
    with pytest.raises(FileNotFoundError):
        config_parser.parse(str(tmp_path / 'nope.cf'))


def test_pad_matches_reference_semantics():
    img = torch.arange(3 * 37 * 51).reshape(1, 3, 37, 51)
    out, t = pad.pad(img, 8, mode='constant')
    assert out.shape[-2:] == (40, 56) and t == (2, 3, 1, 2)               # left, right, top, bottom
    assert torch.equal(pad.undo_pad(out, *t), img)
    same, ident = pad.pad(out, 8)
    assert same is out and ident(5) == 5


def test_auto_crop_counts_and_stitch():
    for H, W, n in [(100, 60, 64), (49, 33, 16), (20, 20, 4), (10, 10, 1)]:
        img = torch.arange(3 * H * W).reshape(1, 3, H, W)
        crops = list(auto_crop.iter_crops(img, 210))
        assert len(crops) == n, (H, W, len(crops))
        if n > 1:
            assert torch.equal(auto_crop.stitch(crops), img)
    assert not auto_crop.needs_crop(torch.zeros(1, 3, 1500, 2000))        # strictly greater (auto_crop.py:47)
    assert auto_crop.needs_crop(torch.zeros(1, 3, 1500, 2001))
    c = auto_crop.CropLossCombinator()
    c.add(2.0, 100)
    c.add(4.0, 300)
    assert abs(c.get_bpsp() - 3.5) < 1e-12


def test_part_suffixes(tmp_path):
    assert psh.make_part_suffix(10) == '.part10' and psh.contains_part_suffix('a/b.part3')
    assert not psh.contains_part_suffix('a/b.part3/more') and psh.index_of_part_suffix('x.part12') == 12
    for i in range(12):
        (tmp_path / ('some.file' + psh.make_part_suffix(i))).write_text('x')
    got = psh.iter_part_suffixes(str(tmp_path / 'some.file.part3'))
    assert [os.path.basename(p) for p in got] == ['some.file.part{}'.format(i) for i in range(12)]


def test_uniform_row_and_quantiser_maps():
    assert uniform_cdf_row(25).numpy().view('uint16').tolist()[:6] == [0, 2621, 5243, 7864, 10486, 13107]
    assert uniform_cdf_row(25).numpy().view('uint16').tolist()[-2:] == [62915, 0]
    assert uniform_cdf_row(256).numpy().view('uint16').tolist()[:3] == [0, 256, 512]
    S = torch.arange(25)
    bn = quantizer.to_bn(S, -1, 1, 25)
    assert torch.equal(quantizer.to_sym(bn, -1, 1, 25), S)
    assert torch.equal(synthetic.quantiser_levels((-1, 1), 25), bn)


def test_experiment_dir_conventions(tmp_path):
    cfg = config_parser.parse_builtin('ms', 'cr')
    exp = tmp_path / '0306_0001 cr oi'
    (exp / 'ckpts').mkdir(parents=True)
    sd = synthetic.make_state_dict(cfg, 0)
    torch.save({'net': sd}, str(exp / 'ckpts' / 'ckpt_0000000010.pt'))
    torch.save({'net': sd}, str(exp / 'ckpts' / 'ckpt_0000000020.pt.tmp'))
    d = paths.get_experiment_dir(str(tmp_path), '0306_0001')
    assert d == str(exp)
    comps = paths.parse_log_dir(d, config_parser.CONFIG_DIR)
    assert comps.config_paths[0].endswith(os.path.join('ms', 'cr.cf')) and comps.postfix is None
    assert paths.get_ckpt_for_itr(paths.get_ckpts_dir(d), -1)[0] == 20
    assert paths.get_ckpt_for_itr(paths.get_ckpts_dir(d), 5)[0] == 10
    with pytest.raises(ValueError):
        paths.get_experiment_dir(str(tmp_path), '0101_0000')
    assert paths.parse_log_dir(str(tmp_path / '0306_0002 cr oi lr.initial=0.1'), config_parser.CONFIG_DIR).postfix == ('lr.initial=0.1',)


def test_schema_strict_checks(synthetic_l3c):
    cfg, sd = synthetic_l3c
    schema.check_state_dict(sd, cfg)
    bad = dict(sd)
    bad['extra.weight'] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        schema.check_state_dict(bad, cfg)
    assert schema.non_shared_get_Kp(10, 3) == 120 and schema.non_shared_get_Kp(10, 5) == 150


def test_rgb_baseline_models_construct_and_load():
    from l3c_pytorch_amd.modules.multiscale_network import MultiscaleNetwork
    from l3c_pytorch_amd.test.multiscale_tester import MultiscaleTester
    for name, n_keys in [('cr_rgb_shared', 48), ('cr_rgb', 140)]:
        cfg = config_parser.parse_builtin('ms', name)
        net = MultiscaleNetwork(cfg)
        sd = synthetic.make_state_dict(cfg, 0)
        assert len(sd) == n_keys
        net.load_state_dict(sd, strict=True)
    shared = config_parser.parse_builtin('ms', 'cr_rgb_shared')
    assert MultiscaleTester._parse_recursive_flag('auto', shared) == 3
    assert MultiscaleTester._parse_recursive_flag('auto', config_parser.parse_builtin('ms', 'cr')) == 0
    with pytest.raises(ValueError):
        MultiscaleTester._parse_recursive_flag('2', config_parser.parse_builtin('ms', 'cr_rgb'))


def test_rgb_decode_pipeline_schedule():
    """Chunk-pipelined RGB decode (bitcoding.rgb_pipeline_schedule): every (channel, chunk) exactly once, in chunk order per
    channel, and channel c's chunk j at least D steps after channel c - 1's -- D = 2 is what lets the tables of the next
    step be built while the current step decodes."""
    from l3c_pytorch_amd.bitcoding.bitcoding import rgb_pipeline_schedule
    for n_chunks in (1, 2, 5, 16):
        for C in (1, 3, 5):
            for D in (1, 2):
                steps = rgb_pipeline_schedule(n_chunks, C, D)
                assert len(steps) == n_chunks + D * (C - 1)
                when = {}
                for t, active in enumerate(steps):
                    assert len(active) <= 8          # parts per grouped decoder launch (l3c_ac_decode_chunks)
                    for c, j in active:
                        assert (c, j) not in when
                        when[(c, j)] = t
                assert sorted(when) == [(c, j) for c in range(C) for j in range(n_chunks)]
                for (c, j), t in when.items():
                    if j:
                        assert when[(c, j - 1)] == t - 1      # a stream's state is carried from one step to the next
                    if c:
                        assert when[(c - 1, j)] <= t - D


def test_count_scale_records_and_async_writer(tmp_path):
    """the `.l3c` framing walk that tells the decoder how many scales a file holds (L3C: 4; RGB Shared: one more per recursion),
    and the worker-thread file writer: wait(path) makes a pending write visible, overwriting waits for the previous write."""
    import struct
    from l3c_pytorch_amd.bitcoding.bitcoding import AsyncFileWriter, count_scale_records, _MAGIC_VALUE_SEP

    def record(C, H, W, payloads):
        return struct.pack('<BHH', C, H, W) + b''.join(struct.pack('<I', len(p)) + p for p in payloads) + _MAGIC_VALUE_SEP

    head = struct.pack('<4H', 0, 0, 0, 0)
    f4 = head + record(5, 1, 1, [b'a'] * 5) + record(5, 2, 2, [b'bb'] * 5) + record(5, 4, 4, [b''] * 5) + record(3, 8, 8, [b'xyz'] * 3)
    assert count_scale_records(f4) == 4
    f5 = head + b''.join(record(3, 1 << i, 1 << i, [bytes(i)] * 3) for i in range(5))
    assert count_scale_records(f5) == 5
    import pytest
    with pytest.raises(ValueError):
        count_scale_records(f4[:-2])                      # truncated
    with pytest.raises(ValueError):
        count_scale_records(f4[:-4] + b'\\0\\0\\0\\0')        # separator missing
    with pytest.raises(ValueError):
        count_scale_records(head + record(5, 1, 1, [b'a'] * 5))
    w = AsyncFileWriter(n_threads=2)
    paths = [str(tmp_path / 'f{}.bin'.format(i)) for i in range(20)]
    for i, p in enumerate(paths):
        w.submit(p, bytes([i]) * 100000)
    w.submit(paths[3], b'second')                         # same path again: ordered after the first write
    w.wait(paths[3])
    assert open(paths[3], 'rb').read() == b'second'
    w.close()
    assert w.pending() == 0
    for i, p in enumerate(paths):
        if i != 3:
            assert open(p, 'rb').read() == bytes([i]) * 100000


def test_tester_image_prefetch_keeps_order(tmp_path):
    """MultiscaleTester._iter_images: files decoded on worker threads ahead of the consumer, yielded strictly in order."""
    import numpy as np
    from PIL import Image
    from l3c_pytorch_amd.test.multiscale_tester import MultiscaleTester
    ps = []
    for i in range(11):
        p = str(tmp_path / '{:02d}.png'.format(i))
        Image.fromarray(np.full((5 + i, 7, 3), i, dtype=np.uint8)).save(p)
        ps.append(p)

    class Flags(object):
        crop = None
    t = MultiscaleTester.__new__(MultiscaleTester)           # only the loader is exercised: no checkpoint, no GPU
    t.flags, t.io_threads = Flags(), 3
    got = list(t._iter_images(ps))
    assert [i for i, _, _ in got] == list(range(11)) and [p for _, p, _ in got] == ps
    for i, _, img in got:
        assert tuple(img.shape) == (3, 5 + i, 7) and int(img.max()) == i


def test_saturated_sigmoid_shortcuts_are_exact_in_fp32():
    """csrc/dmll_kernels.hip sigmoid_sat: the two shortcuts must reproduce 1.0f / (1.0f + expf(-a)) bit for bit, with margin for an
    expf that is off by a few ulp (checked here with numpy's fp32 arithmetic; the GPU side is pinned by the table fixtures)."""
    import numpy as np
    hi = np.linspace(16.7, 120.0, 200001, dtype=np.float32)
    e = np.exp(-hi.astype(np.float64)).astype(np.float32)
    for scale in (np.float32(1.0), np.float32(1.0 + 8e-7), np.float32(1.0 - 8e-7)):        # expf wrong by ~7 ulp either way
        assert np.all(np.float32(1.0) / (np.float32(1.0) + e * scale) == np.float32(1.0))
    assert np.float32(np.exp(np.float64(-16.7))) < np.float32(2.0 ** -24)
    with np.errstate(over='ignore'):
        lo = np.linspace(-200.0, -89.0, 100001, dtype=np.float32)
        e = np.exp(-lo)                                   # fp32 overflow -> inf
        assert np.all(np.isinf(e)) and np.all(np.float32(1.0) / (np.float32(1.0) + e) == 0.0)
    assert np.exp(np.float64(89.0)) > float(np.finfo(np.float32).max) * 1.3                       # e^89 overflows fp32 with margin


def test_image_set_plan_matches_host_padding():
    """dataset_codec.plan_set (config 4's host plan: no pixel touched) against pad.pad: the padded shape and the padding tuple of every
    image are what the reference's centre padding gives (helpers/pad.py:23-59); chunks hold at most max_batch images of ONE padded shape
    and together cover the set exactly once."""
    from l3c_pytorch_amd.helpers import dataset_codec, pad
    sizes = dataset_codec.draw_sizes(120, seed=3) + [(510, 768), (512, 766), (509, 765), (512, 768)]
    order = list(range(len(sizes)))
    chunks, padded, pads, n_shapes = dataset_codec.plan_set(dict(enumerate(sizes)), order, 4, 8)
    assert sorted(i for c in chunks for i in c) == order
    assert n_shapes == len(set(padded)) and all(1 <= len(c) <= 4 for c in chunks)
    for c, shape in zip(chunks, padded):
        for i in c:
            h, w = sizes[i]
            x, pt = pad.pad(torch.zeros(1, 3, h, w, dtype=torch.uint8), 8, mode='constant')
            assert tuple(x.shape[-2:]) == shape
            assert (pt if isinstance(pt, tuple) else (0, 0, 0, 0)) == pads[i]
    # images of different raw shapes belong to one group when their padded shapes agree
    assert {shape for c, shape in zip(chunks, padded) for i in c if i >= 120} == {(512, 768)}


def test_coder_group_cuts():
    """Bitcoding.encode_many launches the coder after these pass counts: equal groups, or cuts at given fractions; always ascending,
    always ending with the last pass, never an empty group."""
    from l3c_pytorch_amd.bitcoding.bitcoding import coder_group_cuts
    assert coder_group_cuts(53, 8) == [7, 14, 21, 28, 35, 42, 49, 53]
    assert coder_group_cuts(224, 8) == [28, 56, 84, 112, 140, 168, 196, 224]
    assert coder_group_cuts(3, 4) == [1, 2, 3] and coder_group_cuts(1, 4) == [1] and coder_group_cuts(0, 4) == []
    assert coder_group_cuts(5, 1) == [5]
    assert coder_group_cuts(53, (0.5, 0.75, 0.875, 0.9375, 1.0)) == [26, 40, 46, 50, 53]
    assert coder_group_cuts(3, (0.5, 0.75, 1.0)) == [2, 3] and coder_group_cuts(1, (0.1, 0.2)) == [1]
    for n in range(1, 60):
        for g in (1, 2, 3, 4, 8, 16, (0.08, 0.31, 0.54, 0.77, 0.92, 1.0)):
            c = coder_group_cuts(n, g)
            assert c == sorted(set(c)) and c[-1] == n and c[0] >= 1


def test_balanced_shards_on_the_config_4_size_law():
    """helpers/sharding.shard_balanced (largest-first greedy on pixel counts): disjoint cover, deterministic, and at 8 ranks the
    heaviest rank stays within 3 % of the mean where round robin is 10 % over (config 4: 500 images, areas differing 4x)."""
    from l3c_pytorch_amd.helpers import dataset_codec, sharding
    sizes = dataset_codec.draw_sizes(500)
    costs = [h * w for h, w in sizes]
    for world in (1, 2, 3, 8):
        shards = [sharding.shard_balanced(costs, r, world) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(500))
        assert shards == [sharding.shard_balanced(costs, r, world) for r in range(world)]
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) <= 1.03 * sum(costs) / world, (world, max(loads) * world / sum(costs))
    rr = [sum(costs[i] for i in sharding.shard_indices(500, r, 8)) for r in range(8)]
    assert max(rr) > 1.08 * sum(costs) / 8                       # what the balanced plan removes
    assert sharding.shard_balanced([], 0, 4) == [] and sharding.shard_balanced([5], 3, 4) == [] and sharding.shard_balanced([5], 0, 4) == [0]
    b = [sharding.host_budget(w, 256) for w in (1, 2, 4, 8)]
    assert [x['pinned_buffers'] for x in b] == [8, 8, 4, 3] and all(x['torch_threads'] * w <= 256 for x, w in zip(b, (1, 2, 4, 8)))


def test_coding_targets_host_bits_equal_the_oracle():
    """the bin edges of the coder (coders_helpers.py:42-44) are computed on the host like the oracle's: identical bits"""
    from oracle import cdf as ocdf
    from l3c_pytorch_amd.criterion.logistic_mixture import DiscretizedMixLogisticLoss
    for x_min, x_max, L in ((0, 255, 256), (-1, 1, 25)):
        got = DiscretizedMixLogisticLoss(rgb_scale=(L == 256), x_min=x_min, x_max=x_max, L=L).coding_targets('cpu')
        assert got.numpy().tobytes() == ocdf.coding_targets(x_min, x_max, L).numpy().tobytes()


# ---- helpers/runtime.py: hardware queues and NUMA placement (round 5) -------------------------------------------------------------

def test_numa_plan_from_a_fake_sysfs(tmp_path):
    """plan_affinity reads the GPU's NUMA node and that node's CPUs from sysfs; ranks whose GPUs hang off different nodes get
    disjoint CPU sets; without NUMA information several ranks still get disjoint slices, a single rank stays unbound."""
    from l3c_pytorch_amd.helpers import runtime
    sysfs = tmp_path
    for card, node in ((0, 0), (1, 1)):
        d = sysfs / 'class' / 'drm' / 'card{}'.format(card) / 'device'
        d.mkdir(parents=True)
        (d / 'numa_node').write_text('{}\n'.format(node))
    for node, cpus in ((0, '0-3'), (1, '4-7')):
        d = sysfs / 'devices' / 'system' / 'node' / 'node{}'.format(node)
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cpus + '\n')
    allowed = set(range(8))
    p0 = runtime.plan_affinity(0, 2, 0, allowed=allowed, sysfs=str(sysfs))
    p1 = runtime.plan_affinity(1, 2, 1, allowed=allowed, sysfs=str(sysfs))
    assert p0['numa_node'] == 0 and p0['cpus'] == [0, 1, 2, 3] and p1['numa_node'] == 1 and p1['cpus'] == [4, 5, 6, 7]
    # two local ranks whose GPUs share a node split its CPUs
    d = sysfs / 'class' / 'drm' / 'card2' / 'device'
    d.mkdir(parents=True)
    (d / 'numa_node').write_text('1\n')
    s1 = runtime.plan_affinity(1, 3, 1, allowed=allowed, sysfs=str(sysfs))
    s2 = runtime.plan_affinity(2, 3, 2, allowed=allowed, sysfs=str(sysfs))
    assert s1['cpus'] == [4, 5] and s2['cpus'] == [6, 7] and runtime.plan_affinity(0, 3, 0, allowed=allowed, sysfs=str(sysfs))['cpus'] == [0, 1, 2, 3]
    # a device sysfs knows nothing about: an even slice per rank (disjoint), or nothing for a single rank
    q = [runtime.plan_affinity(r, 4, 5 + r, allowed=allowed, sysfs=str(sysfs)) for r in range(4)]
    assert [x['cpus'] for x in q] == [[0, 1], [2, 3], [4, 5], [6, 7]] and all(x['numa_node'] is None for x in q)
    assert runtime.plan_affinity(0, 1, 9, allowed=allowed, sysfs=str(sysfs))['cpus'] is None
    assert runtime._parse_cpulist('0-2,8,10-11') == {0, 1, 2, 8, 10, 11}


def test_forward_streams_follow_the_hardware_queues(monkeypatch):
    """encode_many's side-by-side forward streams need >= 8 hardware queues; with fewer it uses one stream and says so ONCE."""
    import warnings
    from l3c_pytorch_amd.helpers import runtime
    monkeypatch.setattr(runtime, '_SNAPSHOT', [None])
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')
    assert runtime.hw_queues() == 8 and runtime.forward_streams_allowed(3) == 3
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    monkeypatch.setattr(runtime, '_warned', [False])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert runtime.forward_streams_allowed(3) == 1 and runtime.forward_streams_allowed(3) == 1
    assert len([x for x in w if 'GPU_MAX_HW_QUEUES' in str(x.message)]) == 1
    assert runtime.forward_streams_allowed(1) == 1


def test_hw_queues_are_what_the_runtime_started_with(monkeypatch):
    """Once the package has touched HIP (`snapshot_hw_queues`: _lib.load / require_gpu), a later change of GPU_MAX_HW_QUEUES cannot reach the
    runtime: `hw_queues()` keeps answering with the value of THEN, `configure_hip_queues()` returns None, warns and leaves the environment alone
    -- so encode_many never runs three forward streams on four queues silently (advisor, round 5)."""
    import warnings
    from l3c_pytorch_amd.helpers import runtime
    monkeypatch.setattr(runtime, '_SNAPSHOT', [None])
    monkeypatch.setattr(runtime, '_warned', [False])
    monkeypatch.delenv('GPU_MAX_HW_QUEUES', raising=False)
    assert runtime.snapshot_hw_queues() == 4            # HIP "started" with the default
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert runtime.configure_hip_queues() is None and 'GPU_MAX_HW_QUEUES' not in os.environ
        monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')    # too late
        assert runtime.hw_queues() == 4 and runtime.forward_streams_allowed(3) == 1
    assert any('already running' in str(x.message) for x in w)


def test_configure_hip_queues_is_an_explicit_call():
    """`l3c_pytorch_amd.configure_hip_queues()` before HIP starts -> GPU_MAX_HW_QUEUES = 8 unless the caller chose a value; importing
    the package alone leaves the runtime's default (batches of equally sized images are faster with it) -- fresh interpreters."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import os, sys; sys.path.insert(0, {!r}); import l3c_pytorch_amd; a = os.environ.get("GPU_MAX_HW_QUEUES"); '
            'l3c_pytorch_amd.configure_hip_queues(); print(a, os.environ.get("GPU_MAX_HW_QUEUES"))').format(root)
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    assert subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE).stdout.decode().split() == ['None', '8']
    env['GPU_MAX_HW_QUEUES'] = '2'
    assert subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE).stdout.decode().split() == ['2', '2']


def test_pmc_table_is_accepted_per_kernel_source(tmp_path, monkeypatch):
    """bench.load_pmc_table: counters are reported only for kernels whose sources are byte-identical to those the rocprofv3 --pmc passes ran on.
    A table taken on exactly today's sources is current; one taken before a source WITHOUT roofline kernels changed stays current and says what
    changed; one taken before a roofline kernel's source changed is stale; a table without per-file stamps falls back to the combined stamp."""
    import json
    import bench
    now = bench.csrc_file_stamps()
    assert set(bench.ROOFLINE_KERNEL_SOURCES) <= set(now)
    path = tmp_path / 'pmc.json'
    monkeypatch.setattr(bench, 'PMC_TABLE', str(path))

    def state_of(table):
        path.write_text(json.dumps(table))
        t, state = bench.load_pmc_table()
        return state, (t or {}).get('_changed_since')

    assert state_of({'csrc_stamp': bench.csrc_stamp(), 'csrc_files': now}) == ('current', None)
    other = dict(now, **{'conv_small.hip': '0' * 16})
    assert state_of({'csrc_stamp': 'x', 'csrc_files': other}) == ('current', ['conv_small.hip'])
    for f in bench.ROOFLINE_KERNEL_SOURCES:
        assert state_of({'csrc_stamp': 'x', 'csrc_files': dict(now, **{f: '0' * 16})})[0] == 'stale', f
    assert state_of({'csrc_stamp': 'x'})[0] == 'stale'
    path.unlink()
    assert bench.load_pmc_table() == (None, 'absent')
    # the committed table of this round must be usable by the line the driver runs
    monkeypatch.undo()
    t, state = bench.load_pmc_table()
    assert state == 'current', 'profiles/{} was taken on other sources of the roofline kernels: regenerate it (tools/make_evidence.sh)'.format(bench.PMC_TABLE)


def test_committed_hip_bitstream_fixtures_are_intact():
    """tests/golden/hip_*.l3c were written by the HIP path of the generation hip_bitstream.json names (tests/golden/make_hip_bitstream.py on an
    MI355X); the GPU suite decodes them with today's build (tests/test_gpu_bitstream.py).  Here, without a GPU: the files are the ones the
    record describes, their framing parses, the inputs they must decode to are the committed fixture images, and the record is of the
    generation the header declares."""
    import hashlib
    import json
    import re
    import numpy as np
    from l3c_pytorch_amd.bitcoding.bitcoding import count_scale_records, parse_containers
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, 'tests', 'golden')
    rec = json.load(open(os.path.join(golden, 'hip_bitstream.json')))
    hdr = open(os.path.join(root, 'include', 'l3c_hip.h')).read()
    assert rec['bitstream_generation'] == int(re.search(r'#define L3C_BITSTREAM_GENERATION (\d+)', hdr).group(1))
    for name, records, img_file in (('hip_l3c_cal_64x96.l3c', 4, 'net_cal_64x96.npz'), ('hip_rgb_shared_32x48_r3.l3c', 5, 'net_rgb_32x48.npz')):
        data = open(os.path.join(golden, name), 'rb').read()
        meta = rec['files'][name]
        assert len(data) == meta['bytes'] and hashlib.sha256(data).hexdigest() == meta['sha256']
        assert count_scale_records(data) == records
        p = parse_containers([data])
        assert p.padding == [(0, 0, 0, 0)] and len(p.scales) == records
        H, W = meta['shape'][-2:]
        assert p.scales[-1][1:] == (H, W) and p.scales[0][1:] == (H >> (records - 1), W >> (records - 1))
        img = np.load(os.path.join(golden, img_file))['img']
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == meta['pixels_sha256']
    assert all(len(rec['forward_64x96'][k]['P']) == 3 and len(rec['forward_64x96'][k]['S']) == 4 for k in ('default', 'calibrated'))


def test_container_framing_parser_and_set_decode_plan():
    """`parse_containers` walks the `.l3c` framing by its length fields only (bitcoding.py:326-375 of the reference) and is what the decoder
    uploads by: offsets and lengths of every payload, (C, H, W) per scale record, padding; files that disagree in shape or are broken raise
    ValueError; `plan_decode_set` groups files by the padded shape their own headers announce."""
    import struct
    import pytest
    from l3c_pytorch_amd.bitcoding.bitcoding import _MAGIC_VALUE_SEP, parse_containers
    from l3c_pytorch_amd.helpers import dataset_codec

    def make(shapes, pay, padding=(0, 0, 0, 0)):
        out = [struct.pack('<4H', *padding)]
        for (C, H, W), lens in zip(shapes, pay):
            out.append(struct.pack('<BHH', C, H, W))
            for n in lens:
                out += [struct.pack('<I', n), bytes(range(256)) * (n // 256) + bytes(n % 256)]
            out.append(_MAGIC_VALUE_SEP)
        return b''.join(out)

    sh = [(5, 4, 6), (5, 8, 12), (5, 16, 24), (3, 32, 48)]
    a = make(sh, [[3, 0, 7, 1, 2], [10] * 5, [300, 2, 2, 2, 2], [1000, 999, 998]], (1, 2, 3, 4))
    b = make(sh, [[1] * 5, [2] * 5, [3] * 5, [4, 5, 6]])
    p = parse_containers([a, b])
    assert p.padding == [(1, 2, 3, 4), (0, 0, 0, 0)] and p.scales == sh
    assert p.nbytes[0].tolist() == [[3, 0, 7, 1, 2], [1] * 5] and p.nbytes[3].tolist() == [[1000, 999, 998], [4, 5, 6]]
    for k in range(4):
        for bi, f in enumerate((a, b)):
            for c in range(sh[k][0]):
                o, n = int(p.offset[k][bi, c]), int(p.nbytes[k][bi, c])
                assert struct.unpack_from('<I', f, o - 4)[0] == n
    assert f[int(p.offset[3][1, 2]) + 6:][:4] == _MAGIC_VALUE_SEP
    with pytest.raises(ValueError):
        parse_containers([a, make([(5, 4, 6), (5, 8, 12), (5, 16, 24), (3, 32, 40)], [[1] * 5, [2] * 5, [3] * 5, [4, 5, 6]])])
    with pytest.raises(ValueError):
        parse_containers([a[:-3]])
    with pytest.raises(ValueError):
        parse_containers([a, make(sh[1:], [[2] * 5, [3] * 5, [4, 5, 6]])])
    assert dataset_codec.file_padded_shape(a) == (32, 48)
    c = make([(5, 8, 6), (5, 16, 12), (5, 32, 24), (3, 64, 48)], [[1] * 5, [2] * 5, [3] * 5, [4, 5, 6]])
    chunks, padded = dataset_codec.plan_decode_set({0: a, 1: c, 2: b, 3: a, 4: a}, [0, 1, 2, 3, 4], 3)
    assert sorted(map(tuple, chunks)) == [(0, 2, 3), (1,), (4,)] and sorted(padded) == [(32, 48), (32, 48), (64, 48)]


def test_balanced_cu_sets_hold_every_xcd_equally():
    import collections
    from l3c_pytorch_amd.helpers import runtime
    first, rest = runtime.balanced_cu_sets(256, 64)
    assert len(first) == 64 and len(rest) == 192 and not set(first) & set(rest)
    for numbering in (lambda i: i % 8, lambda i: i // 32):
        assert sorted(collections.Counter(map(numbering, first)).values()) == [8] * 8
    import pytest
    with pytest.raises(ValueError):
        runtime.balanced_cu_sets(256, 40)


def test_ragged_rgb_plan_tiles_every_image_in_lock_step():
    """`ops.ragged_rgb_plan`: the chunk plan of the set decoder's ragged RGB pipeline (l3c_decode_rgb_ragged validates the same on the C side):
    every image the same NUMBER of chunks, its chunks tile [0, HW) in order, none empty, every boundary but the image's end on a multiple of 64."""
    import numpy as np
    from l3c_pytorch_amd import ops
    hws = [512 * 768, 768 * 512, 584 * 880, 1024 * 1368, 131072 + 64, 400 * 400]
    for n_regular, probe in ((32, 1024), (32, 0), (7, 1024), (1, 0)):
        pix0, npix = ops.ragged_rgb_plan(hws, n_regular, probe)
        assert pix0.shape == npix.shape == (n_regular + (2 if probe else 0), len(hws))
        for b, hw in enumerate(hws):
            assert (npix[:, b] > 0).all() and pix0[0, b] == 0
            assert (pix0[1:, b] == np.cumsum(npix[:-1, b])).all() and pix0[-1, b] + npix[-1, b] == hw
            assert (pix0[:, b] % 64 == 0).all()
            if probe:
                assert npix[0, b] == npix[1, b] == probe
    pix0, npix = ops.ragged_rgb_plan([6, 24, 96], 1, 0)          # the coarsest scales of an RGB Shared file: one chunk each
    assert npix.tolist() == [[6, 24, 96]]
