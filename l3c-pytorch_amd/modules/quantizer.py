"""Symbol <-> bottleneck-value maps of the reference quantiser (modules/quantizer.py:38-47).

The quantiser itself (Quantizer.forward :62-90, eval path) is fused into the `to_q` 1x1 conv kernel
(csrc/conv_small.hip: to_q_quantize_kernel); these two helpers work on any device tensor with torch ops because callers
of the reference use them on small host-side tensors too (bitcoding.py:147, :258)."""


def to_sym(x, x_min, x_max, L):
    bin_size = (x_max - x_min) / (L - 1)
    return x.clamp(x_min, x_max).sub(x_min).div(bin_size).round().long()


def to_bn(S, x_min, x_max, L):
    bin_size = (x_max - x_min) / (L - 1)
    return S.float().mul(bin_size).add(x_min)
