"""-m gpu: BASELINE.json config 5 -- large images with the reference's REAL auto-crop threshold (auto_crop.py:31-47: more than
2000 x 1500 pixels -> 2 x 2 crops, recursively) -- on the RGB Shared baseline (forward + bpsp, padding to 16,
multiscale_tester.py:222-225) against the oracle, and through the L3C file API (`.l3c.part0..3`, bitcoding.py:63-71, :131-135)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bitcoding as obc, net as onet  # noqa: E402


def test_real_threshold_decisions():
    from l3c_pytorch_amd import auto_crop
    assert auto_crop._NEEDS_CROP_DIM == 2000 * 1500
    assert not auto_crop.needs_crop(torch.zeros(1, 3, 1500, 2000))          # exactly the threshold: not cropped (strictly greater)
    assert auto_crop.needs_crop(torch.zeros(1, 3, 2000, 3000))
    crops = list(auto_crop.iter_crops(torch.zeros(1, 3, 2000, 3000)))
    assert [tuple(c.shape[-2:]) for c in crops] == [(1000, 1500)] * 4
    assert len(list(auto_crop.iter_crops(torch.zeros(1, 3, 4000, 3100)))) == 16      # 12.4 MPix: two levels


def test_rgb_shared_on_a_3000x2000_image_vs_oracle():
    """RGB Shared (cr_rgb_shared.cf, auto_recurse 3) on the four 1500x1000 crops of a 3000x2000 image, each padded to a multiple
    of 16: symbols of all five pyramid levels equal the oracle's (PIL bicubic) for the crop the oracle is run on, P within 1e-5
    RELATIVE to its largest magnitude (this model family feeds the decoder un-normalised pixel values, +-128: its activations
    and P are ~100x those of L3C, the absolute error scales along: 4.4e-5 measured on values up to ~40) at 1504x1008 -- a size
    whose dilated sub-grids do not divide the tile size --, bpsp within 1e-4 relative of the oracle's, and the area-weighted
    combination (auto_crop.py:139-152) equals the mean over equal crops."""
    from l3c_pytorch_amd import auto_crop
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import config_parser, pad, synthetic
    cfg = config_parser.parse_builtin('ms', 'cr_rgb_shared')
    sd = synthetic.make_state_dict(cfg, 0)
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    img = synthetic.make_image(2000, 3000, 77, 'natural').unsqueeze(0)
    crops = list(auto_crop.iter_crops(img))
    assert len(crops) == 4
    comb = auto_crop.CropLossCombinator()
    per_crop = []
    for k, crop in enumerate(crops):
        x, pt = pad.pad(crop, 16, mode='constant')
        assert tuple(x.shape[-2:]) == (1008, 1504) and pt == (2, 2, 4, 4)
        out = bp.forward(x.float().cuda(), 3)
        loss = bp.get_loss(out, num_subpixels_before_pad=crop.numel())
        bpsp = float(sum(loss.recursive_bpsps))
        per_crop.append(bpsp)
        comb.add(bpsp, crop.shape[-2] * crop.shape[-1])
        if k == 3:                                                   # the oracle on the bottom-right crop (CPU: a few seconds)
            with torch.no_grad():
                ref = onet.forward_rgb(x.float(), sd, onet.RGB_SHARED_HYPER, dec_skip=False, auto_recurse=3)
            for i in range(5):
                assert torch.equal(out.S[i].cpu(), ref.S[i]), i
            for i in range(4):
                err = (out.P[i].cpu() - ref.P[i]).abs().max().item()
                mag = ref.P[i].abs().max().item()
                print('RGB Shared 1504x1008, scale {}: max |P - oracle| = {:.3g}, max |P| = {:.3g}, relative {:.3g}'.format(i, err, mag, err / mag))
                assert err < 1e-5 * mag and err < 2e-4, (i, err, mag)
            _, rec = obc.losses_bpsp_rgb(ref, num_subpixels=crop.numel())
            assert abs(bpsp - sum(rec)) < 1e-4 * sum(rec), (bpsp, sum(rec))
    assert abs(comb.get_bpsp() - np.mean(per_crop)) < 1e-9


def test_l3c_file_api_on_an_image_above_the_threshold(synthetic_l3c, tmp_path):
    """L3C encode -> four part files -> decode, threshold untouched: 2048x1536 = 3.15 MPix > 3.0 MPix.  Every part equals the crop
    coded on its own, the stitched decode is the image, bpsp is the area-weighted mean of the parts."""
    from l3c_pytorch_amd import auto_crop
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import synthetic
    cfg, sd = synthetic_l3c
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(sd, strict=True)
    bp.set_eval()
    bc = Bitcoding(bp)
    img = synthetic.make_image(1536, 2048, 78, 'natural').unsqueeze(0).long()
    assert auto_crop.needs_crop(img)
    p = str(tmp_path / 'big.l3c')
    bpsp = bc.encode(img, p)
    assert sorted(os.listdir(str(tmp_path))) == ['big.l3c.part{}'.format(i) for i in range(4)]
    sizes = [os.path.getsize(p + '.part{}'.format(i)) for i in range(4)]
    assert abs(bpsp - sum(sizes) * 8 / img.numel()) < 1e-9
    crop2 = list(auto_crop.iter_crops(img))[2]
    alone = bc.encode_batch(crop2).to_bytes()[0]
    assert open(p + '.part2', 'rb').read() == alone
    back = bc.decode(p + '.part1')
    assert torch.equal(back.cpu(), img)


@pytest.mark.parametrize('name,recurse,shape', [('cr_rgb_shared', 3, (96, 160)), ('cr_rgb_shared', 3, (75, 100)), ('cr_rgb', 0, (64, 80))])
def test_rgb_baselines_write_to_files_round_trip(name, recurse, shape, tmp_path):
    """file coding for the RGB baselines -- which the reference does not have for recursive models
    (multiscale_tester.py:187-188).  The `.l3c` layout carries one scale record per pyramid level (auto_recurse 3: five), the
    decoder counts them; encode -> decode is lossless (odd sizes: padded to 2**(num_scales + recurse)), the file is no larger
    than the theoretical recursive cost + 3 % + framing (it may be smaller: a symbol the random-weight model gives less than
    2^-16 costs 16 bits in the file and more in theory), and the coarsest record is the uniformly coded 1/16-resolution image."""
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding, count_scale_records
    from l3c_pytorch_amd.blueprints.multiscale_blueprint import MultiscaleBlueprint
    from l3c_pytorch_amd.helpers import config_parser, synthetic
    cfg = config_parser.parse_builtin('ms', name)
    bp = MultiscaleBlueprint(cfg)
    bp.net.load_state_dict(synthetic.make_state_dict(cfg, 0), strict=True)
    bp.set_eval()
    bc = Bitcoding(bp, auto_recurse=recurse)
    img = synthetic.make_image(shape[0], shape[1], 31, 'natural').unsqueeze(0).long()
    p = str(tmp_path / 'x.l3c')
    bpsp = bc.encode(img, p)
    data = open(p, 'rb').read()
    n_scales = cfg.num_scales + recurse
    assert count_scale_records(data) == n_scales + 1
    fac = 2 ** n_scales
    Hp, Wp = -(-shape[0] // fac) * fac, -(-shape[1] // fac) * fac
    assert data[8:13] == bytes([3]) + (Hp // fac).to_bytes(2, 'little') + (Wp // fac).to_bytes(2, 'little')
    back = bc.decode(p)
    assert torch.equal(back.cpu(), img)
    # against the theoretical cost of the same forward pass (recursive evaluation sums every scale + the uniform top)
    from l3c_pytorch_amd.helpers import pad
    x = pad.pad(img, fac, mode='constant')[0]
    out = bp.forward(x.float().cuda(), recurse)
    loss = bp.get_loss(out)
    theory = float(sum(loss.recursive_bpsps if recurse else loss.nonrecursive_bpsps))
    framing = 8 + (n_scales + 1) * (5 + 3 * 4 + 4)
    assert len(data) - framing <= 1.03 * theory * x.numel() / 8 + 3 * (n_scales + 1) * 4, (len(data), theory * x.numel() / 8)
    assert abs(bpsp - len(data) * 8 / x.numel()) < 1e-9


def test_test_py_write_to_files_recursive(tmp_path):
    """`python test.py LOG_DIR 0306_0002 IMAGES --recursive auto --write_to_files D`: the reference's command line, which the
    reference answers with NotImplementedError for recursive models."""
    import importlib.util
    from PIL import Image
    from l3c_pytorch_amd.helpers import config_parser, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config_parser.parse_builtin('ms', 'cr_rgb_shared')
    exp = tmp_path / 'logs' / '0306_0002 cr_rgb_shared oi' / 'ckpts'
    exp.mkdir(parents=True)
    torch.save({'net': synthetic.make_state_dict(cfg, 0)}, str(exp / 'ckpt_0000000001.pt'))
    imgs_dir = tmp_path / 'imgs'
    imgs_dir.mkdir()
    for name, (H, W) in {'a': (48, 64), 'b': (33, 47), 'c': (48, 64)}.items():
        Image.fromarray(synthetic.make_image(H, W, ord(name), 'natural').permute(1, 2, 0).numpy()).save(str(imgs_dir / (name + '.png')))
    spec = importlib.util.spec_from_file_location('l3c_test_cli2', os.path.join(root, 'test.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    out_dir = tmp_path / 'written'
    cli.main([str(tmp_path / 'logs'), '0306_0002', str(imgs_dir), '--recursive', 'auto', '--write_to_files', str(out_dir)])
    assert sorted(os.listdir(str(out_dir))) == ['a.l3c', 'b.l3c', 'c.l3c']
