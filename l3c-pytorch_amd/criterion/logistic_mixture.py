"""DiscretizedMixLogisticLoss on HIP kernels -- same surface as the reference's criterion/logistic_mixture.py
(non-shared mixtures, lambda-coupled RGB means): `forward` (NLL map, :146-207), `cdf_step_non_shared` (:134-141),
`to_sym` / `to_bn` (:120-124), `CDFOut` (:61-65), `non_shared_get_Kp/K` (:68-81).

`l` is the network output, logical shape (N,Kp,H,W).  The kernels read it pixel-major; when `l` is the permuted view that
MultiscaleNetwork hands out this costs nothing, any other layout is re-laid out once.

Difference from the reference, on purpose: `cdf_step_non_shared` does NOT write the coupled means back into `l`
(the reference's `means_c += ...` mutates the network output in place through a view, :267/:272); the returned
`means_c` is identical, `l` stays pristine.  Sampling (`sample`, :277-323) is outside the hot path.
"""
from collections import namedtuple

import torch
from torch import nn

from .. import ops
from ..modules import quantizer
from ..modules.schema import non_shared_get_Kp  # noqa: F401  (re-export, reference name)

_NUM_PARAMS_RGB = 4    # mu, sigma, pi, lambda
_NUM_PARAMS_OTHER = 3  # mu, sigma, pi
_LOG_SCALES_MIN = -7.

CDFOut = namedtuple('CDFOut', ['logit_probs_c_sm', 'means_c', 'log_scales_c', 'K', 'targets'])


def non_shared_get_K(Kp, C):
    return Kp // ((_NUM_PARAMS_RGB if C == 3 else _NUM_PARAMS_OTHER) * C)


class DiscretizedMixLogisticLoss(nn.Module):
    def __init__(self, rgb_scale, x_min=0, x_max=255, L=256):
        super(DiscretizedMixLogisticLoss, self).__init__()
        self.rgb_scale = rgb_scale
        self.x_min, self.x_max, self.L = x_min, x_max, L
        self.use_coeffs = rgb_scale
        self._num_params = _NUM_PARAMS_RGB if rgb_scale else _NUM_PARAMS_OTHER
        self.bin_width = (x_max - x_min) / (L - 1)
        self.x_lower_bound = x_min + 0.001
        self.x_upper_bound = x_max - 0.001
        self._extra_repr = 'DMLL: x={}, L={}, coeffs={}, P={}, bin_width={}'.format(
            (self.x_min, self.x_max), self.L, self.use_coeffs, self._num_params, self.bin_width)

    def extra_repr(self):
        return self._extra_repr

    def to_sym(self, x):
        return quantizer.to_sym(x, self.x_min, self.x_max, self.L)

    def to_bn(self, S):
        return quantizer.to_bn(S, self.x_min, self.x_max, self.L)

    def _K(self, l, C):
        return l.shape[1] // (self._num_params * C)

    def coding_targets(self, device='cuda'):
        """Bin edges the coder evaluates the CDF at (bitcoding/coders_helpers.py:42-44); torch.linspace on purpose -- its rounding is
        part of the bitstream contract, and it is evaluated on the HOST: the device kernel of linspace may round `start + i * step`
        differently, and the oracle (oracle/cdf.py:coding_targets) is the CPU form (tests/test_gpu_head.py pins the bits)."""
        return torch.linspace(self.x_min - self.bin_width / 2, self.x_max + self.bin_width / 2, self.L + 1,
                              dtype=torch.float32).to(device)

    def cdf_step_non_shared(self, l, targets, c_cur, C, x_c=None):
        """-> CDFOut(pi softmaxed, mu (lambda-coupled), log_sigma clamped, K, targets), each parameter (N,K,H,W)."""
        assert c_cur < C
        K = self._K(l, C)
        sym = None
        if self.use_coeffs and c_cur != 0:
            assert x_c is not None
            # the coupling uses the actual values of the previous channels; on the RGB scale values == symbols
            sym = x_c.round().to(device='cuda', dtype=torch.int16).contiguous()
        pi, mu, ls = ops.dmll_channel_params(ops.as_pixel_major(l), sym, C, K, self.rgb_scale, c_cur)
        return CDFOut(pi, mu, ls, K, targets.to(l.device))

    def forward(self, x, l, scale=0):
        """x: targets (N,C,H,W) float; l: (N,Kp,H,W) -> negative log-likelihood (N,C,H,W), nats."""
        N, C, H, W = x.shape
        K = self._K(l, C)
        x = x.to('cuda', torch.float32).contiguous()
        return ops.dmll_nll(ops.as_pixel_major(l), x, C, K, self.rgb_scale, self.x_min, self.x_max, self.L)

    def sample(self, l, C, noise=None):
        """Sample from the model (reference _non_shared_sample :277-323): l (N,Kp,H,W) -> x (N,C,H,W) fp32, not rounded.
        noise: (u_mix (N,C,K,H,W), u_logistic (N,C,H,W)) uniforms; default: drawn on the device like the reference does
        (uniform_(1e-5, 1 - 1e-5), :286, :300)."""
        N, Kp, H, W = l.shape
        K = self._K(l, C)
        if noise is None:
            noise = (torch.empty(N, C, K, H, W, device='cuda').uniform_(1e-5, 1. - 1e-5),
                     torch.empty(N, C, H, W, device='cuda').uniform_(1e-5, 1. - 1e-5))
        u_mix, u_log = [u.to('cuda', torch.float32).contiguous() for u in noise]
        return ops.dmll_sample(ops.as_pixel_major(l), u_mix, u_log, C, K, self.rgb_scale)
