"""ArithmeticCoder -- the per-channel coder object of the reference API (bitcoding/coders.py:33-90), on the HIP range coder.

    ArithmeticCoder(L).range_encode(data (N,H,W) int16, cdf, time_logger) -> bytes
    ArithmeticCoder(L).range_decode(bytes, cdf, time_logger)              -> (N,H,W) int16, CPU

`cdf` is a CDFOut (mixture parameters of one channel: the uint16 table is built on the device) or a finished (N,H,W,Lp) int16
table.  One stream per call -- the entry point for code written against the reference; `Bitcoding.encode_batch` /
`decode_batch` code every stream of a batch in one grouped launch and never come through here
(tests/test_gpu_net.py::test_reference_per_channel_api_produces_the_batched_files_payloads pins the two to the same bytes).
"""
import torch

from .. import torchac
from ..criterion.logistic_mixture import CDFOut
from .bitcoding import _NullTimes


def _table_shape(cdf, L):
    """(N, H, W) of what `cdf` describes; a finished table must have L + 1 entries per row."""
    if isinstance(cdf, CDFOut):
        N, _, H, W = cdf.means_c.shape
        return N, H, W
    N, H, W, Lp = cdf.shape
    if Lp != L + 1:
        raise ValueError('table rows have {} entries, this coder was built for L = {} (L + 1 expected)'.format(Lp, L))
    return N, H, W


class ArithmeticCoder(object):
    def __init__(self, L):
        self.L = L

    def range_encode(self, data, cdf, time_logger=None):
        if data.dim() != 3 or data.dtype != torch.int16:
            raise ValueError('symbols must be (N,H,W) int16, got {} {}'.format(tuple(data.shape), data.dtype))
        _table_shape(cdf, self.L)
        flat = data.reshape(-1).contiguous()
        with (time_logger or _NullTimes()).run('ac.encode'):
            if isinstance(cdf, CDFOut):
                return torchac.encode_logistic_mixture(cdf.targets, cdf.means_c, cdf.log_scales_c, cdf.logit_probs_c_sm, flat)
            return torchac.encode_cdf(cdf, flat)

    def range_decode(self, encoded_bytes, cdf, time_logger=None):
        shape = _table_shape(cdf, self.L)
        # (the reference times BOTH directions under the key 'ac.encode', coders.py:76; consumers of its time-logger keys find the same key here)
        with (time_logger or _NullTimes()).run('ac.encode'):
            if isinstance(cdf, CDFOut):
                flat = torchac.decode_logistic_mixture(cdf.targets, cdf.means_c, cdf.log_scales_c, cdf.logit_probs_c_sm, encoded_bytes)
            else:
                flat = torchac.decode_cdf(cdf, encoded_bytes)
        return flat.reshape(shape)
