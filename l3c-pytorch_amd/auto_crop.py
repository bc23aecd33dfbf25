"""Tiling of large images (reference auto_crop.py): images with H*W > AC_NEEDS_CROP_DIM (default 2000*1500, strictly
greater, :44-47) are split recursively into 2x2 crops (TL, TR, BL, BR, :65-75) which are coded independently -- on this
build they simply become more items of the batch -- and stitched back in image order (:109-136); bpsp is combined
weighted by area (:139-152)."""
import functools
import math
import operator
import os

import torch


def prod(it):
    return functools.reduce(operator.mul, it, 1)


_NEEDS_CROP_DIM_DEFAULT = '2000,1500'
_NEEDS_CROP_DIM = prod(map(int, os.environ.get('AC_NEEDS_CROP_DIM', _NEEDS_CROP_DIM_DEFAULT).split(',')))


def _assert_valid_image(i):
    if len(i.shape) != 4 or i.shape[1] != 3:
        raise ValueError('Expected BCHW image, got {}'.format(tuple(i.shape)))


def needs_crop(img, needs_crop_dim=None):
    _assert_valid_image(img)
    H, W = img.shape[-2:]
    return H * W > (_NEEDS_CROP_DIM if needs_crop_dim is None else needs_crop_dim)


def _crop4(img):
    H, W = img.shape[-2:]
    h, w = H // 2, W // 2
    return [img[..., :h, :w], img[..., :h, w:], img[..., h:, :w], img[..., h:, w:]]


def iter_crops(img, needs_crop_dim=None):
    _assert_valid_image(img)
    if not needs_crop(img, needs_crop_dim):
        yield img
        return
    for quarter in _crop4(img):
        for c in iter_crops(quarter, needs_crop_dim):
            yield c


def _z_order_to_raster(side):
    """position (raster index) in the side x side grid of the i-th crop produced by the recursive split."""
    if side == 1:
        return [0]
    sub = _z_order_to_raster(side // 2)
    half = side // 2
    order = []
    for qy, qx in ((0, 0), (0, 1), (1, 0), (1, 1)):
        for r in sub:
            y, x = r // half + qy * half, r % half + qx * half
            order.append(y * side + x)
    return order


def stitch(parts):
    side = int(math.sqrt(len(parts)))
    if side * side != len(parts) or (side & (side - 1)):
        raise ValueError('Invalid number of parts {}'.format(len(parts)))
    grid = [None] * len(parts)
    for part, pos in zip(parts, _z_order_to_raster(side)):
        grid[pos] = part
    rows = [torch.cat(grid[r * side:(r + 1) * side], dim=3) for r in range(side)]
    return torch.cat(rows, dim=2)


class CropLossCombinator(object):
    def __init__(self):
        self._num_bits_total = 0.
        self._num_subpixels_total = 0

    def add(self, bpsp, num_subpixels_crop):
        self._num_bits_total += bpsp * num_subpixels_crop
        self._num_subpixels_total += num_subpixels_crop

    def get_bpsp(self):
        assert self._num_subpixels_total > 0
        return self._num_bits_total / self._num_subpixels_total
