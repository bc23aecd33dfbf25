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
    of 16: symbols of all five pyramid levels of EVERY crop equal the oracle's (PIL bicubic) for the crop the oracle is run on,
    P within 1e-5 (values at 1504x1008 incl. sub-grids that do not divide the tile size), bpsp within 1e-4 relative of the
    oracle's, and the area-weighted combination (auto_crop.py:139-152) equals the mean over equal crops."""
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
                print('RGB Shared 1504x1008, scale {}: max |P - oracle| = {:.3g}'.format(i, err))
                assert err < 1e-5, (i, err)
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
