"""Centre padding to a multiple of the total downscaling factor (reference helpers/pad.py:23-59).

Returns `(img, identity)` untouched when nothing has to be padded and `(img, (left, right, top, bottom))` otherwise,
exactly like the reference (whose callers unpack both forms)."""
from torch.nn import functional as F


def _identity(x):
    return x


def _split(total):
    first = total // 2
    return first, total - first


def padding_for(h, w, fac):
    """(left, right, top, bottom) that `pad` would apply to an h x w image, (0, 0, 0, 0) if none."""
    top, bottom = _split((-h) % fac)
    left, right = _split((-w) % fac)
    return (left, right, top, bottom)


def pad(img, fac, mode='replicate'):
    _, _, h, w = img.shape
    left, right, top, bottom = padding_for(h, w, fac)
    if not (left or right or top or bottom):
        return img, _identity
    assert (h + top + bottom) % fac == 0 and (w + left + right) % fac == 0
    padding_tuple = (left, right, top, bottom)
    return F.pad(img, padding_tuple, mode), padding_tuple


def undo_pad(img, padLeft, padRight, padTop, padBottom, target_shape=None):
    H, W = img.shape[-2:]
    out = img[..., padTop:H - padBottom, padLeft:W - padRight]
    if target_shape:
        assert tuple(out.shape[-2:]) == tuple(target_shape), (out.shape[-2:], target_shape)
    return out
