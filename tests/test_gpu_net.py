"""-m gpu: MultiscaleNetwork / MultiscaleBlueprint / Bitcoding on the MI355X path against the reference fixture (config[0])
and the oracle at other sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bitcoding as obc, net as onet  # noqa: E402
from tests.conftest import NET_FIXTURES  # noqa: E402
from tests.parity_gate import assert_P, assert_P_truth, truth_chain  # noqa: E402


def _make_blueprint(cfg, sd):
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    return bp


@pytest.fixture(scope='module')
def blueprint(synthetic_l3c):
    return _make_blueprint(*synthetic_l3c)


@pytest.fixture(scope='module')
def blueprint_cal(synthetic_l3c_cal):
    """the same network with the CALIBRATED checkpoint (live probability model; tests/golden/make_calibrated.py)."""
    return _make_blueprint(*synthetic_l3c_cal)


@pytest.fixture(scope='module')
def blueprints(blueprint, blueprint_cal):
    return lambda calibrated: blueprint_cal if calibrated else blueprint


def _near_tie(x_prequant, levels, tol=5e-5):
    d = np.sort(np.abs(x_prequant[..., None] - levels), axis=-1)
    return (d[..., 1] - d[..., 0]) < tol


# Tolerances of the network parity tests (measured on the MI355X at 768x512 -- tests/test_gpu_headline.py
# records them -- and no larger on the small images here): north_star's 1e-5, RELATIVE to the tensor's largest magnitude (floored
# at 1: the default-init checkpoint's tensors stay below 1.2, the calibrated one's means reach 280, where 1 ulp is 3e-5).
TOL_F = 1e-5


def _tol(ref, tol):
    """features only (|F| stays below ~3 on both checkpoints); P is gated per parameter group: tests/parity_gate.py"""
    return tol * max(float(np.abs(np.asarray(ref)).max()), 1.0)


def _symbols_equal_up_to_near_ties(S, S_ref, bn_ref, levels, tol=5e-5):
    """-> number of flipped symbols; every flip must sit where the REFERENCE's own pre-quantiser value is within 5e-5 of a
    decision boundary (to_q sums 64 features: a 1e-5 difference in F moves it by a few 1e-5; the calibrated checkpoint's to_q
    rows are ~10x larger -- they spread the bottleneck over all 25 levels -- hence 5e-4 there)."""
    bad = S != S_ref
    assert bad.sum() == 0 or _near_tie(bn_ref, levels, tol)[bad].all()
    return int(bad.sum())


@pytest.mark.parametrize('fixture,calibrated', NET_FIXTURES)
def test_forward_matches_reference_fixture(golden, blueprints, l3c_checkpoint, fixture, calibrated):
    """config[0] (32x32; and the calibrated checkpoint at 32x32 and 64x96) against the reference's own CPU forward
    (tests/golden/net_*.npz): encoder features within 1e-5, symbols identical except provable quantiser near-ties, P within 1e-5
    -- through get_P on the reference's bottlenecks, so that a flipped symbol cannot void the comparison -- and per-scale bpsp
    within 1e-4 relative (+ 16 bits per flipped symbol)."""
    cfg, sd = l3c_checkpoint(calibrated)
    blueprint = blueprints(calibrated)
    g = golden(fixture)
    st = int(g['p0_stride']) if 'p0_stride' in g.files else 1
    img = torch.from_numpy(g['img'].astype(np.float32)).cuda()
    out = blueprint.forward(img)
    levels = sd['nets.0.enc.levels'].numpy()
    assert torch.equal(out.S[0].cpu(), torch.from_numpy(g['S0'].astype(np.int64)))
    flips = 0
    for s in range(3):
        flips += _symbols_equal_up_to_near_ties(out.S[s + 1].cpu().numpy(), g['S%d' % (s + 1)], g['enc_bn%d' % s], levels,
                                                tol=(5e-4 if calibrated else 5e-5))
        if 'enc_F%d' % s not in g.files:
            continue
        Fe = out.raw.F_enc[s].cpu().permute(0, 3, 1, 2).numpy()
        err = np.abs(Fe - g['enc_F%d' % s]).max()
        print('scale {}: max |F_enc - reference| = {:.3g}'.format(s, err))
        assert err < _tol(g['enc_F%d' % s], TOL_F), (s, err)
    f_prev = None
    P64 = truth_chain([None] + [g['bn%d' % k] for k in (1, 2, 3)], sd)     # the same chain in double: the accuracy gate (parity_gate.py)
    for s in (2, 1, 0):
        P, f_prev = blueprint.net.get_P(s, torch.from_numpy(g['bn%d' % (s + 1)]).cuda(), f_prev)
        assert_P_truth(P.cpu(), P64[s], s, fixture)
        Pn = P.cpu().numpy()
        if s == 0:
            Pn = Pn[:, :, ::st, ::st]
        assert Pn.shape == g['P%d' % s].shape
        err = np.abs(Pn - g['P%d' % s]).max()
        print('scale {}: max |P - reference| = {:.3g} (largest |P| {:.3g})'.format(s, err, np.abs(g['P%d' % s]).max()))
        assert_P(Pn, g['P%d' % s], s, fixture)
        if flips == 0:
            assert torch.equal(P, out.P[s]), s            # and the forward pass computed exactly this
    loss = blueprint.get_loss(out)
    got = np.array([float(b) for b in loss.nonrecursive_bpsps])
    assert np.allclose(got, g['bpsp'], rtol=1e-4, atol=16.0 * flips / img.numel()), (got, g['bpsp'], flips)
    if calibrated:
        assert sum(got) < 10.5
    assert out.L == [256, 25, 25, 25] and out.bn[0] is None
    for s in range(1, 4):
        assert torch.equal(out.bn[s].cpu(), torch.from_numpy(levels)[out.S[s].cpu()])


@pytest.mark.parametrize('fixture,calibrated', NET_FIXTURES[:2])
def test_decoder_side_on_reference_bottlenecks(golden, blueprints, fixture, calibrated):
    """get_P fed with the REFERENCE's bn_q (so no quantiser flip can leak in): F and P within 1e-5 (relative)."""
    g = golden(fixture)
    blueprint = blueprints(calibrated)
    f_prev = None
    for s in (2, 1, 0):
        bn = torch.from_numpy(g['bn%d' % (s + 1)]).cuda()
        P, f_prev = blueprint.net.get_P(s, bn, f_prev)
        assert np.abs(f_prev.cpu().numpy() - g['dec_F%d' % s]).max() < _tol(g['dec_F%d' % s], TOL_F), s
        assert_P(P, g['P%d' % s], s, fixture)


@pytest.mark.parametrize('calibrated', [False, True])
@pytest.mark.parametrize('H,W', [(40, 56), (64, 96), (8, 8), (104, 200)])
def test_forward_vs_oracle_other_sizes(blueprints, l3c_checkpoint, H, W, calibrated):
    """Other sizes against the oracle: the encoder chain (which never sees the symbols: enc.feed_F) from the forward pass, the
    decoder chain and P through get_P on the ORACLE's bottlenecks; a flipped near-tie symbol is counted, never skipped."""
    from l3c_pytorch_amd.helpers import synthetic
    cfg, sd = l3c_checkpoint(calibrated)
    blueprint = blueprints(calibrated)
    img = synthetic.make_image(H, W, 5, 'natural').unsqueeze(0).float()
    with torch.no_grad():
        ref = onet.forward(img, sd)
        bn_ref = [onet.conv(ref.F_enc[s], sd, 'nets.{}.enc.to_q.0'.format(s)).numpy() for s in range(3)]
    out = blueprint.forward(img.cuda())
    levels = sd['nets.0.enc.levels'].numpy()
    flips = 0
    for s in range(3):
        flips += _symbols_equal_up_to_near_ties(out.S[s + 1].cpu().numpy(), ref.S[s + 1].numpy(), bn_ref[s], levels,
                                                tol=(5e-4 if calibrated else 5e-5))
        assert (out.raw.F_enc[s].cpu().permute(0, 3, 1, 2) - ref.F_enc[s]).abs().max() < _tol(ref.F_enc[s], TOL_F), s
    f_prev = None
    P64 = truth_chain(ref.bn, sd)
    for s in (2, 1, 0):
        P, f_prev = blueprint.net.get_P(s, ref.bn[s + 1].cuda(), f_prev)
        assert (f_prev.cpu() - ref.F_dec[s]).abs().max() < _tol(ref.F_dec[s], TOL_F), s
        assert_P(P, ref.P[s], s, (H, W, calibrated))
        assert_P_truth(P.cpu(), P64[s], s, (H, W, calibrated))       # the accuracy gate: flat 1e-5 against fp64
        if flips == 0:
            assert torch.equal(P, out.P[s]), s


@pytest.mark.parametrize('calibrated', [False, True])
def test_get_P_is_bit_identical_to_forward_and_batch_invariant(blueprints, calibrated):
    """The lossless contract (P4): the decoder recomputes P from bn_q with the same kernels and tile schedule."""
    from l3c_pytorch_amd.helpers import synthetic
    blueprint = blueprints(calibrated)
    imgs = torch.stack([synthetic.make_image(48, 72, i, 'natural') for i in range(3)]).float().cuda()
    out = blueprint.forward(imgs)
    f_prev = None
    for s in (2, 1, 0):
        P, f_prev = blueprint.net.get_P(s, out.bn[s + 1], f_prev)
        assert torch.equal(P, out.P[s]), s
    single = blueprint.forward(imgs[1:2])
    for s in range(3):
        assert torch.equal(single.P[s], out.P[s][1:2]), s
        assert torch.equal(single.S[s + 1], out.S[s + 1][1:2])


@pytest.mark.parametrize('calibrated', [False, True])
@pytest.mark.parametrize('H,W,B,kind', [(32, 32, 1, 'uniform'), (64, 96, 3, 'natural'), (8, 16, 2, 'smooth'),
                                        (128, 192, 2, 'natural')])
def test_encode_decode_lossless(blueprints, H, W, B, kind, calibrated):
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    blueprint = blueprints(calibrated)
    imgs = torch.stack([synthetic.make_image(H, W, 10 + i, kind) for i in range(B)]).long()
    bc = Bitcoding(blueprint)
    enc = bc.encode_batch(imgs)
    files = enc.to_bytes()
    sizes = enc.file_sizes().cpu().tolist()
    assert sizes == [len(f) for f in files]
    dec, pads = bc.decode_batch(files)
    assert torch.equal(dec.cpu(), imgs)
    # a batch codes every image exactly like a batch of one
    alone = bc.encode_batch(imgs[:1]).to_bytes()[0]
    assert alone == files[0]
    # fixed framing overhead: 116 bytes
    n_payload = int(enc.total_payload_bytes()[0])
    assert len(files[0]) - n_payload == 116


def test_device_side_file_assembly_equals_host_assembly(blueprint):
    """l3c_container_write (headers + length fields + payloads re-aligned to byte offsets, all files of a batch in one
    buffer) == the byte-by-byte host assembly of the same coder output, with and without padding headers."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    bc = Bitcoding(blueprint)
    for B, H, W in [(1, 8, 8), (3, 40, 24), (5, 64, 96)]:
        imgs = torch.stack([synthetic.make_image(H, W, 90 + i, ['natural', 'uniform', 'smooth'][i % 3]) for i in range(B)]).long()
        enc = bc.encode_batch(imgs)
        pads = [(i, 2 * i, 3, 65535 - i) for i in range(B)]
        assert enc.to_bytes() == enc.to_bytes_host_assembled()
        assert enc.to_bytes(pads) == enc.to_bytes_host_assembled(pads)
        host, offs, sizes = enc.to_host_buffer()
        assert int(sizes.sum()) == host.size and list(sizes) == enc.file_sizes().cpu().tolist()


def test_encode_many_heterogeneous_equals_per_batch(blueprint):
    """encode_many: batches of different shapes share ONE grouped coder launch; bytes equal encode_batch's."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    bc = Bitcoding(blueprint)
    shapes = [(2, 32, 48), (1, 64, 64), (3, 16, 8), (1, 128, 72)]
    batches = [torch.stack([synthetic.make_image(H, W, 50 + 7 * k + i, 'natural') for i in range(B)]).long()
               for k, (B, H, W) in enumerate(shapes)]
    many = [e.to_bytes() for e in bc.encode_many(batches)]
    for x, files in zip(batches, many):
        assert files == bc.encode_batch(x).to_bytes()
        dec, _ = bc.decode_batch(files)
        assert torch.equal(dec.cpu(), x)


@pytest.mark.parametrize('fixture,calibrated', NET_FIXTURES)
def test_file_api_with_padding_and_reference_file_size(golden, blueprints, tmp_path, fixture, calibrated):
    """reference API: encode(img, path) -> bpsp, decode(path) -> 1CHW long; odd sizes are centre padded.  The file is within
    16 bytes of the REFERENCE's own file for the same weights and image (7111 B default 32x32; 3976 / 18210 B calibrated)."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    bc = Bitcoding(blueprints(calibrated), compare_with_theory=True)
    g = golden(fixture)
    img = torch.from_numpy(g['img'].astype(np.int64))
    p = str(tmp_path / 'a.l3c')
    bpsp = bc.encode(img, p)
    size = os.path.getsize(p)
    assert abs(bpsp - size * 8 / img.numel()) < 1e-9
    # same weights, same image: the reference's file is 7111 bytes; ours differs only where a CDF rounding moved
    assert abs(size - len(g['l3c'])) <= 16, (size, len(g['l3c']))
    assert torch.equal(bc.decode(p).cpu(), img)
    odd = synthetic.make_image(37, 51, 4, 'natural').unsqueeze(0).long()
    p2 = str(tmp_path / 'b.l3c')
    bc.encode(odd, p2)
    back = bc.decode(p2)
    assert back.shape == odd.shape and torch.equal(back.cpu(), odd)
    with open(p2, 'rb') as f:
        import struct
        assert struct.unpack('<4H', f.read(8)) == (2, 3, 1, 2)      # left, right, top, bottom for 37x51 -> 40x56


def test_auto_crop_parts(blueprint, tmp_path, monkeypatch):
    from l3c_pytorch_amd import auto_crop
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.helpers import synthetic
    monkeypatch.setattr(auto_crop, '_NEEDS_CROP_DIM', 40 * 40)
    img = synthetic.make_image(64, 80, 8, 'natural').unsqueeze(0).long()
    bc = Bitcoding(blueprint)
    p = str(tmp_path / 'big.l3c')
    bpsp = bc.encode(img, p)
    parts = sorted(os.listdir(str(tmp_path)))
    assert parts == ['big.l3c.part0', 'big.l3c.part1', 'big.l3c.part2', 'big.l3c.part3']
    back = bc.decode(p + '.part2')
    assert torch.equal(back.cpu(), img) and bpsp > 0
    # crops that need padding (35x45 -> 40x48), 16 parts (recursive split): every part file equals the crop coded on its own
    monkeypatch.setattr(auto_crop, '_NEEDS_CROP_DIM', 40 * 50)
    img2 = synthetic.make_image(140, 180, 9, 'natural').unsqueeze(0).long()
    p2 = str(tmp_path / 'odd.l3c')
    bpsp2 = bc.encode(img2, p2)
    crops = list(auto_crop.iter_crops(img2))
    assert len(crops) == 16 and tuple(crops[0].shape[-2:]) == (35, 45)
    comb = auto_crop.CropLossCombinator()
    monkeypatch.setattr(auto_crop, '_NEEDS_CROP_DIM', 10 ** 9)
    for i, crop in enumerate(crops):
        single = str(tmp_path / 'single{}.l3c'.format(i))
        comb.add(bc.encode(crop.clone(), single), 35 * 45)
        assert open(single, 'rb').read() == open(p2 + '.part{}'.format(i), 'rb').read(), i
    assert abs(comb.get_bpsp() - bpsp2) < 1e-12
    assert torch.equal(bc.decode(p2 + '.part0').cpu(), img2)


def test_l3c_cli_enc_dec_roundtrip(synthetic_l3c, tmp_path):
    """`python l3c.py LOG_DIR LOG_DATE enc IMG OUT` / `dec OUT PNG` (reference l3c.py:74-125) on a synthetic experiment dir."""
    import importlib.util
    from PIL import Image
    from l3c_pytorch_amd.helpers import synthetic
    cfg, sd = synthetic_l3c
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = tmp_path / 'logs' / '0306_0001 cr oi' / 'ckpts'
    exp.mkdir(parents=True)
    torch.save({'net': sd}, str(exp / 'ckpt_0000000001.pt'))
    img = synthetic.make_image(45, 70, 21, 'natural')
    src = str(tmp_path / 'in.png')
    Image.fromarray(img.permute(1, 2, 0).numpy()).save(src)
    spec = importlib.util.spec_from_file_location('l3c_cli', os.path.join(root, 'l3c.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    out = str(tmp_path / 'out.l3c')
    assert cli.main([str(tmp_path / 'logs'), '0306_0001', 'enc', src, out]) == 0
    assert os.path.getsize(out) > 116
    assert cli.main([str(tmp_path / 'logs'), '0306_0001', 'enc', src, out]) == 1          # exists, no --overwrite
    assert cli.main([str(tmp_path / 'logs'), '0306_0001', 'enc', src, out, '-f']) == 0
    png = str(tmp_path / 'dec.png')
    assert cli.main([str(tmp_path / 'logs'), '0306_0001', 'dec', out, png]) == 0
    back = torch.from_numpy(np.array(Image.open(png))).permute(2, 0, 1)
    assert torch.equal(back, img)


def test_test_py_driver_bpsp_cache_and_write_to_files(synthetic_l3c, tmp_path, capsys):
    """`python test.py LOG_DIR LOG_DATE IMAGES [--write_to_files D --time_report P]` (reference test.py:44-137): per-image bpsp
    equals the oracle's get_loss within 1e-4 relative, the result cache answers the second run, --write_to_files round-trips."""
    import importlib.util
    from PIL import Image
    from l3c_pytorch_amd.helpers import synthetic
    cfg, sd = synthetic_l3c
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = tmp_path / 'logs' / '0306_0001 cr oi' / 'ckpts'
    exp.mkdir(parents=True)
    torch.save({'net': sd}, str(exp / 'ckpt_0000000001.pt'))
    imgs_dir = tmp_path / 'imgs'
    imgs_dir.mkdir()
    imgs = {}
    for name, (H, W) in {'a': (32, 48), 'b': (32, 48), 'c': (27, 41)}.items():
        imgs[name] = synthetic.make_image(H, W, ord(name), 'natural')
        Image.fromarray(imgs[name].permute(1, 2, 0).numpy()).save(str(imgs_dir / (name + '.png')))
    spec = importlib.util.spec_from_file_location('l3c_test_cli', os.path.join(root, 'test.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    res = cli.main([str(tmp_path / 'logs'), '0306_0001', str(imgs_dir)])
    (testset, log_date, itr, result), = res
    assert testset.id == 'imgs_3' and log_date == '0306_0001' and itr == 1 and result.startswith('bpsp=')
    # oracle per image (pad like the tester does, bpsp over the unpadded sub-pixel count)
    from l3c_pytorch_amd.helpers import pad as padmod
    expect = []
    for name in 'abc':
        x = imgs[name].unsqueeze(0)
        xp, _ = padmod.pad(x, 8, mode='constant')
        with torch.no_grad():
            o = onet.forward(xp.float(), sd)
        expect.append(sum(obc.losses_bpsp(o, num_subpixels=x.numel())))
    assert abs(float(result.split('=')[1]) - np.mean(expect)) < 1e-4 * np.mean(expect)
    capsys.readouterr()
    cli.main([str(tmp_path / 'logs'), '0306_0001', str(imgs_dir)])
    assert 'Found cached' in capsys.readouterr().out
    out_dir = tmp_path / 'written'
    report = tmp_path / 'times.txt'
    cli.main([str(tmp_path / 'logs'), '0306_0001', str(imgs_dir), '--write_to_files', str(out_dir), '--time_report', str(report)])
    assert sorted(os.listdir(str(out_dir))) == ['a.l3c', 'b.l3c', 'c.l3c']
    assert 'bc.encode' in report.read_text() and 'bc.decode' in report.read_text()
    # --sample (reference multiscale_tester.py:436-448): ground truth + 3 sampled images per input, refuses to overwrite
    samples = tmp_path / 'samples'
    cli.main([str(tmp_path / 'logs'), '0306_0001', str(imgs_dir), '--sample', str(samples), '--overwrite_cache'])
    files = sorted(os.listdir(str(samples / '0306_0001')))
    assert len(files) == 12 and sum(f.endswith('_gt.png') for f in files) == 3, files
    assert {f.split('_')[2] for f in files if not f.endswith('_gt.png')} == {'rgb', 'rgb+bn0', 'rgb+bn0+bn1'}
    gt = np.array(Image.open(str(samples / '0306_0001' / [f for f in files if f.startswith('2_c') and f.endswith('_gt.png')][0])))
    assert gt.shape == (32, 48, 3)       # the 27x41 image, padded to a multiple of 8 like the tester's forward
    with pytest.raises(FileExistsError):
        cli.main([str(tmp_path / 'logs'), '0306_0001', str(imgs_dir), '--sample', str(samples), '--overwrite_cache'])


def test_bicubic_encoder_is_pillow_exact():
    """the device-side pyramid step of the RGB baselines against PIL (what the reference calls, images_loader.py:277-288)."""
    from PIL import Image
    from l3c_pytorch_amd import ops
    rng = np.random.RandomState(4)
    mean = torch.tensor([0.4488, 0.4371, 0.4040]).mul(255.).reshape(1, 3, 1, 1)
    for (H, W) in [(32, 48), (37, 51), (6, 10), (250, 188)]:
        u8 = rng.randint(0, 256, size=(2, 3, H, W)).astype(np.uint8)
        x = torch.from_numpy(u8).float() - mean + torch.from_numpy(rng.uniform(-0.4, 0.4, size=(2, 3, H, W)).astype(np.float32))
        bn, sym = ops.bicubic_encoder(x.cuda().contiguous())
        ref_u8 = (x + mean).clamp(0, 255.).round().to(torch.uint8)
        for n in range(2):
            ref = np.array(Image.fromarray(ref_u8[n].permute(1, 2, 0).numpy()).resize((int(W * 0.5), int(H * 0.5)), Image.BICUBIC))
            assert np.array_equal(sym[n].cpu().numpy().transpose(1, 2, 0), ref), (H, W)
        assert torch.equal(bn.cpu(), sym.cpu().float() - mean)


@pytest.mark.parametrize('name,recurse', [('cr_rgb_shared', 3), ('cr_rgb', 0)])
def test_rgb_baselines_forward_and_loss_vs_reference_fixture(golden, name, recurse):
    """config 5 family: symbols of every pyramid level equal to the reference's (PIL bicubic), P within fp32 tolerance (2e-4
    abs), bpsp (recursive and non-recursive) within 2e-3 relative END TO END: with the synthetic random weights many
    log-scales sit at the -7 clamp (inverse std ~1100), which amplifies a 1e-4 difference in mu a hundredfold inside the
    sigmoid; the NLL kernel itself is pinned at 1e-5 on the reference's own P (tests/test_gpu_head.py)."""
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import config_parser, synthetic
    g = golden('net_rgb_32x48.npz')
    cfg = config_parser.parse_builtin('ms', name)
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(synthetic.make_state_dict(cfg, 0), strict=True)
    bp.set_eval()
    img = torch.from_numpy(g['img'].astype(np.float32)).cuda()
    out = bp.forward(img, recurse)
    n_scales = len(out.P)
    assert n_scales == (4 if recurse else 3) and out.L == [256] * (n_scales + 1)
    for i in range(n_scales + 1):
        assert (out.S[i].cpu().numpy() == g['{}/S{}'.format(name, i)]).all(), i
    for i in range(n_scales):
        P = out.P[i].cpu()
        P = P[:, :, ::4, ::4] if i == 0 else P
        assert np.abs(P.numpy() - g['{}/P{}'.format(name, i)]).max() < 2e-4, i
    loss = bp.get_loss(out)
    assert np.allclose([float(b) for b in loss.nonrecursive_bpsps], g[name + '/nonrecursive_bpsps'], rtol=2e-3)
    if recurse:
        assert np.allclose([float(b) for b in loss.recursive_bpsps], g[name + '/recursive_bpsps'], rtol=2e-3)
    else:
        assert loss.recursive_bpsps is None


def test_sample_forward_vs_oracle_with_the_same_draws(golden, blueprint, synthetic_l3c):
    """sample_forward with a seeded host generator as `draw` requests its uniforms in the reference's order, so the
    oracle (pinned on the reference's own sample_forward by tests/test_oracle.py) sees the same noise.  Sampling only the
    RGB scale keeps every decision local to a pixel: all but a few values must agree; deeper sampling cascades any flipped
    mixture choice through the decoders, so there only shape / range / determinism are checked."""
    from oracle import net as onet
    g = golden('sample_32.npz')
    _, sd = synthetic_l3c
    img = torch.from_numpy(g['img'].astype(np.float32))
    draw = lambda shape, lo, hi: torch.zeros(shape).uniform_(lo, hi)   # noqa: E731
    torch.manual_seed(5)
    got = blueprint.net.sample_forward(img, blueprint.losses, [], draw=draw).cpu()
    torch.manual_seed(5)
    with torch.no_grad():
        want = onet.sample_forward(img, sd, [])
    assert got.shape == want.shape == (1, 3, 32, 32)
    assert ((got - want).abs() > 1e-2).float().mean() < 0.01
    for scales in ([0], [0, 1], [0, 1, 2]):
        torch.manual_seed(6)
        a = blueprint.net.sample_forward(img, blueprint.losses, scales, draw=draw)
        torch.manual_seed(6)
        b = blueprint.net.sample_forward(img, blueprint.losses, scales, draw=draw)
        assert a.shape == (1, 3, 32, 32) and torch.equal(a, b)
        assert a.min() >= 0 and a.max() <= 255
    # device-side draws (the default) work and differ from call to call
    c, d = blueprint.sample_forward(img, [0]), blueprint.sample_forward(img, [0])
    assert c.is_cuda and not torch.equal(c, d)


def test_reference_per_channel_api_produces_the_batched_files_payloads(blueprint):
    """coders.ArithmeticCoder + coders_helpers.CodingCDFNonshared -- the reference's per-scale, per-channel loop
    (bitcoding.py:171-232) written against our mirrors -- yields byte-for-byte the payloads of the batched encoder, and
    range_decode inverts them."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.bitcoding.coders import ArithmeticCoder
    from l3c_pytorch_amd.bitcoding.coders_helpers import CodingCDFNonshared
    from l3c_pytorch_amd.helpers import synthetic
    img = synthetic.make_image(32, 48, 21, 'natural').unsqueeze(0).long()
    bc = Bitcoding(blueprint)
    out = blueprint.forward(img.float())
    payloads = bc.encode_batch(img, out=out).payloads()          # coarse -> fine, [scale][b][c]
    for k, (scale, dmll, uniform) in enumerate(bc.iter_scale_dmll()):
        if uniform:
            continue
        S, C = out.S[scale], out.S[scale].shape[1]
        x = img.float().cuda() if scale == 0 else out.bn[scale]
        helper = CodingCDFNonshared(out.P[scale], total_C=C, dmll=dmll)
        coder = ArithmeticCoder(dmll.L)
        for c in range(C):
            cdf = helper.get_next_C(x)
            data = coder.range_encode(S[:, c].to(torch.int16), cdf)
            assert data == payloads[k][0][c], (scale, c, len(data), len(payloads[k][0][c]))
            back = coder.range_decode(data, cdf)
            assert torch.equal(back.long(), S[:, c].cpu().long())
