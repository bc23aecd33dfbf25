"""ArithmeticCoder -- the reference's per-channel coder object (bitcoding/coders.py:33-90) on the HIP range coder.

    ArithmeticCoder(L).range_encode(data NHW int16, cdf, time_logger) -> bytes
    ArithmeticCoder(L).range_decode(bytes, cdf, time_logger)          -> NHW int16

`cdf` is either a CDFOut (mixture parameters: the table is built on the device, torchac.{en,de}code_logistic_mixture) or a
finished N,H,W,Lp int16 table (torchac.{en,de}code_cdf).  This is the one-stream-at-a-time API kept for callers of the
reference; `Bitcoding.encode_batch` / `decode_batch` code all streams of a batch in one launch and do not go through it.
"""
import torch

from .. import torchac
from ..criterion.logistic_mixture import CDFOut
from .bitcoding import _NullTimes


class ArithmeticCoder(object):
    def __init__(self, L):
        self.L = L
        self._cached_cdf = None

    def range_encode(self, data, cdf, time_logger=None):
        """data: (N,H,W) int16 symbols (any device) -> bytes."""
        time_logger = time_logger if time_logger is not None else _NullTimes()
        assert len(data.shape) == 3, data.shape
        assert data.dtype == torch.int16, 'Wrong dtype: {}'.format(data.dtype)
        data = data.reshape(-1).contiguous()
        if isinstance(cdf, CDFOut):
            logit_probs_c_sm, means_c, log_scales_c, K, targets = cdf
            with time_logger.run('ac.encode'):
                return torchac.encode_logistic_mixture(targets, means_c, log_scales_c, logit_probs_c_sm, data)
        N, H, W, Lp = cdf.shape
        assert Lp == self.L + 1, (Lp, self.L)
        with time_logger.run('ac.encode'):
            return torchac.encode_cdf(cdf, data)

    def range_decode(self, encoded_bytes, cdf, time_logger=None):
        """-> decoded symbols, (N,H,W) int16 on the CPU (as the reference returns them)."""
        time_logger = time_logger if time_logger is not None else _NullTimes()
        if isinstance(cdf, CDFOut):
            logit_probs_c_sm, means_c, log_scales_c, K, targets = cdf
            N, _, H, W = means_c.shape
            with time_logger.run('ac.encode'):
                decoded = torchac.decode_logistic_mixture(targets, means_c, log_scales_c, logit_probs_c_sm, encoded_bytes)
        else:
            N, H, W, Lp = cdf.shape
            assert Lp == self.L + 1, (Lp, self.L)
            with time_logger.run('ac.encode'):
                decoded = torchac.decode_cdf(cdf, encoded_bytes)
        return decoded.reshape(N, H, W)
