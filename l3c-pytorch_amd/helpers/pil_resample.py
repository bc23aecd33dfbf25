"""Coefficient tables of Pillow's antialiased BICUBIC resize (`ImagingResample`, libImaging/Resample.c), computed on the host
exactly as Pillow does (double arithmetic, then 22-bit fixed point) so that the HIP resampling kernel
(csrc/conv_small.hip: resample_u8_kernel), which only does the integer accumulate, is bit-exact with
`PIL.Image.resize(size, Image.BICUBIC)`.  The reference's RGB baselines call exactly that inside the model
(modules/net.py:72-80 -> dataloaders/images_loader.py:277-288).

    precompute_coeffs   Resample.c: support = 2 * max(scale, 1); ksize = ceil(support)*2 + 1; per output sample the window
                        [xmin, xmin + xmax) and the normalised bicubic (a = -0.5) weights
    normalize_coeffs_8bpc   PRECISION_BITS = 32 - 8 - 2 = 22; k -> (int)(k * 2^22 +- 0.5)
    the two passes      horizontal first, then vertical, each: ss = 2^21 + sum(pixel * k); out = clip8(ss >> 22)
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """-> bounds int32 (out_size, 2) [xmin, count], kk int32 (out_size, ksize) fixed-point weights, ksize"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def half_size(h, w):
    """output size of the reference's `resize_bicubic(t, 0.5)`: (int(H * 0.5), int(W * 0.5))"""
    return int(h * 0.5), int(w * 0.5)


def resample_u8_numpy(img, out_h, out_w):
    """numpy statement of the two integer passes (what the HIP kernel does); img uint8 (..., H, W) -> (..., out_h, out_w)."""
    img = np.asarray(img, dtype=np.uint8)
    H, W = img.shape[-2:]

    def one_pass(a, in_size, out_size):   # along the last axis
        bounds, kk, ksize = precompute_coeffs(in_size, out_size)
        out = np.empty(a.shape[:-1] + (out_size,), dtype=np.uint8)
        for xx in range(out_size):
            xmin, cnt = bounds[xx]
            acc = (a[..., xmin:xmin + cnt].astype(np.int64) * kk[xx, :cnt].astype(np.int64)).sum(-1) + (1 << (PRECISION_BITS - 1))
            out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        return out

    tmp = one_pass(img, W, out_w) if out_w != W else img
    if out_h != H:
        tmp = np.swapaxes(one_pass(np.swapaxes(tmp, -1, -2), H, out_h), -1, -2)
    return tmp
