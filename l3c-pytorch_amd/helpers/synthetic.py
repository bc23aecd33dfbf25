"""Deterministic synthetic checkpoints and images (there is no network: the released weights cannot be fetched).

`make_state_dict(config_ms, seed)` fills the schema of modules/schema.py the way a freshly constructed reference
model would look (nn.Conv2d default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias; MeanShift layers as
edsr.py:52-60 builds them; quantiser levels evenly spaced) but from a generator that does NOT depend on module
construction order: every tensor is drawn from its own CPU `torch.Generator` seeded with crc32(key) ^ seed, so the
build container (which makes tests/golden/*) and the GPU box regenerate bit-identical weights.

`levels` is written as `to_bn(arange(L))` = arange(L) * bin + x_min in fp32 -- the value the decoder reconstructs
(quantizer.py:44-47).  A torch>=2 `linspace` differs from it by 1 ulp at 14/25 levels, which breaks the reference's own
losslessness for fresh checkpoints; checkpoints trained with torch 1.1 carry matching levels.

`make_state_dict(config_ms, seed, calibrated=True)` overlays the tensors of `calibrated/<config>_seed<k>.npz` (the last 1x1
layer of every probability classifier and the bottleneck projections `to_q`, fitted in closed form on synthetic images by
tests/golden/make_calibrated.py): a default-init model predicts mixtures near 0 for pixels in 0..255 and codes two of the three
RGB streams at the 16-bit probability floor; the calibrated one is a working probability model (~6 bpsp on the bench's
images, every bottleneck level in use, log sigma from below the clamp at -7 up to ~5.5), which is the regime a trained
checkpoint puts the head, the tables and both coders in.
"""
import os
import math
import zlib

import numpy as np
import torch

from ..modules import schema as _schema


def quantiser_levels(levels_range, L):
    lo, hi = levels_range
    return torch.arange(L, dtype=torch.float32).mul((hi - lo) / (L - 1)).add(lo)


def _gen(key, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


CALIBRATED_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'calibrated')


def calibrated_name(config_ms):
    """The built-in config a calibration overlay exists for."""
    if config_ms.rgb_bicubic_baseline:
        return 'cr_rgb_shared' if config_ms.num_scales == 1 else None
    return 'cr'


def make_state_dict(config_ms, seed=0, calibrated=False):
    sd = _default_init(config_ms, seed)
    if calibrated:
        name = calibrated_name(config_ms)
        path = os.path.join(CALIBRATED_DIR, '{}_seed{}.npz'.format(name, seed))
        if name is None or not os.path.isfile(path):
            raise FileNotFoundError('no calibrated overlay for this config / seed: {}'.format(path))
        with np.load(path) as z:
            for key in z.files:
                t = torch.from_numpy(z[key].copy())
                assert key in sd and tuple(sd[key].shape) == tuple(t.shape), key
                sd[key] = t
    return sd


def _default_init(config_ms, seed=0):
    sd = {}
    for key, shape in _schema.param_schema(config_ms).items():
        if key.endswith('levels'):
            sd[key] = quantiser_levels(config_ms.q.levels_range, config_ms.q.L)
        elif key.startswith('sub_rgb_mean'):
            sd[key] = (torch.eye(3).view(3, 3, 1, 1) if key.endswith('weight')
                       else -255. * torch.tensor([0.4488, 0.4371, 0.4040]))
        elif key.startswith('heads.0.head.0'):
            sd[key] = (torch.eye(3).view(3, 3, 1, 1) / 128. if key.endswith('weight') else torch.zeros(3))
        else:
            wshape = shape if key.endswith('weight') else _schema.param_schema(config_ms)[key[:-4] + 'weight']
            fan_in = wshape[1] * wshape[2] * wshape[3]
            bound = 1. / math.sqrt(fan_in)
            sd[key] = (torch.rand(shape, generator=_gen(key, seed), dtype=torch.float32) * 2 - 1) * bound
    return sd


def make_image(H, W, seed=0, kind='natural'):
    """uint8 (3,H,W) test image.

    natural: 5 octaves of bilinearly upsampled Gaussian noise per channel, 0.8 cross-channel correlation, + N(0,2)
    uniform: i.i.d. U{0..255} (worst case for the coder)
    smooth:  diagonal gradient + uniform noise in [-8, 8]
    """
    rng = np.random.RandomState(seed)
    if kind == 'uniform':
        return torch.from_numpy(rng.randint(0, 256, size=(3, H, W)).astype(np.uint8))
    if kind == 'smooth':
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        base = (xx / max(W - 1, 1) * 160 + yy / max(H - 1, 1) * 80)[None].repeat(3, 0)
        base = base + np.array([0., 8., -8.], dtype=np.float32)[:, None, None]
        img = base + rng.uniform(-8, 8, size=(3, H, W))
        return torch.from_numpy(np.clip(np.round(img), 0, 255).astype(np.uint8))
    assert kind == 'natural', kind
    shared = np.zeros((H, W), dtype=np.float32)
    chans = np.zeros((3, H, W), dtype=np.float32)
    for octave in range(5):
        h, w = max(2, H >> (octave + 2)), max(2, W >> (octave + 2))
        amp = 2.0 ** octave
        t = torch.from_numpy(rng.randn(4, 1, h, w).astype(np.float32))
        up = torch.nn.functional.interpolate(t, size=(H, W), mode='bilinear', align_corners=False).numpy()[:, 0]
        shared += amp * up[0]
        chans += amp * up[1:]
    img = 0.8 * shared[None] + 0.2 * chans
    img = (img - img.min()) / max(float(img.max() - img.min()), 1e-6) * 255.
    img = img + rng.randn(3, H, W) * 2.
    return torch.from_numpy(np.clip(np.round(img), 0, 255).astype(np.uint8))
