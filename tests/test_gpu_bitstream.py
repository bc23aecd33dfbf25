"""-m gpu: the bitstream of THIS build is the committed one (round-5 verdict, next 4).

The `.l3c` container has no version field (reference src/bitcoding/bitcoding.py:326-375) and a file decodes only with kernels that
reproduce the encoder's P bit for bit (the reference's own check: src/test/multiscale_tester.py:373).  tests/golden/hip_*.l3c were WRITTEN by
the HIP path of bitstream generation `hip_bitstream.json: bitstream_generation` (tests/golden/make_hip_bitstream.py, run on an MI355X);
today's build must decode them to the committed pixels and reproduce the committed SHA-256 of P and of the symbols on both
checkpoints.  A kernel change that moves one bit of P on the decoder side fails here -- the moment to bump L3C_BITSTREAM_GENERATION
(include/l3c_hip.h) and regenerate the fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.conftest import GOLDEN  # noqa: E402
from tests.golden import make_hip_bitstream as gen  # noqa: E402


@pytest.fixture(scope='module')
def record():
    with open(os.path.join(GOLDEN, 'hip_bitstream.json')) as f:
        return json.load(f)


def test_fixtures_are_of_this_builds_generation(record):
    from l3c_pytorch_amd import _lib
    assert record['bitstream_generation'] == _lib.load().l3c_bitstream_generation(), \
        'L3C_BITSTREAM_GENERATION was bumped: regenerate tests/golden/hip_* with tests/golden/make_hip_bitstream.py on an MI355X'


@pytest.mark.parametrize('fname,cfg', [('hip_l3c_cal_64x96.l3c', 'cr'), ('hip_rgb_shared_32x48_r3.l3c', 'cr_rgb_shared')])
def test_committed_hip_files_decode_to_the_committed_pixels(record, fname, cfg):
    from l3c_pytorch_amd.bitcoding.bitcoding import Bitcoding
    meta = record['files'][fname]
    data = open(os.path.join(GOLDEN, fname), 'rb').read()
    assert len(data) == meta['bytes'] and hashlib.sha256(data).hexdigest() == meta['sha256']
    image = gen.l3c_case() if cfg == 'cr' else gen.rgb_case()
    assert gen.sha(image.numpy().astype(np.uint8)) == meta['pixels_sha256']
    bp = gen.blueprint(cfg, True)
    bc = Bitcoding(bp, auto_recurse=meta['auto_recurse'])
    dec, pads = bc.decode_batch([data])
    assert torch.equal(dec.cpu(), image), 'a file written by generation {} no longer decodes: a decoder-side kernel changed P'.format(
        record['bitstream_generation'])
    # and today's encoder writes the same bytes (encoder side pinned too: the quantiser's symbols and the interval kernel)
    again = bc.encode_batch(image).to_bytes()[0]
    assert again == data, 'the encoder no longer writes the committed file ({} vs {} bytes)'.format(len(again), len(data))


@pytest.mark.parametrize('calibrated', [False, True])
def test_P_and_symbols_hash_to_the_committed_values(record, calibrated):
    want = record['forward_64x96']['calibrated' if calibrated else 'default']
    got = gen.p_hashes(gen.blueprint('cr', calibrated), gen.l3c_case())
    assert got['S'] == want['S'], 'symbols changed'
    assert got['P'] == want['P'], 'P changed bits (scales {}): bump L3C_BITSTREAM_GENERATION and regenerate the fixtures'.format(
        [s for s, (a, b) in enumerate(zip(got['P'], want['P'])) if a != b])
