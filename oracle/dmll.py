"""Discretised mixture-of-logistics oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates criterion/logistic_mixture.py of the reference for the non-shared head used by L3C:
  params_for_channel  <- _extract_non_shared_c :248-275 + cdf_step_non_shared :134-141 (WITHOUT the reference's
                         in-place write-through into `l`; callers here always pass the pristine P)
  nll                 <- forward :146-207 with _extract_non_shared :209-246, log_softmax :334-337, log_sum_exp :340-345
  sample              <- _non_shared_sample :277-323, with the two uniform draws (:286, :300) as explicit inputs
  to_sym / to_bn      <- modules/quantizer.py:38-47
Channel layout of P: index = p*(C*K) + c*K + k, p in {0: logit pi, 1: mu, 2: log sigma, 3: lambda (RGB only)}.
"""
import torch
import torch.nn.functional as F

LOG_SCALES_MIN = -7.


class Spec(object):
    """x_min/x_max/L of one DMLL head (logistic_mixture.py:88-118)."""

    def __init__(self, rgb, x_min, x_max, L):
        self.rgb, self.x_min, self.x_max, self.L = rgb, x_min, x_max, L
        self.num_params = 4 if rgb else 3
        self.bin_width = (x_max - x_min) / (L - 1)

    def to_sym(self, x):
        return x.clamp(self.x_min, self.x_max).sub(self.x_min).div(self.bin_width).round().long()

    def to_bn(self, S):
        return S.float().mul(self.bin_width).add(self.x_min)


RGB = Spec(True, 0, 255, 256)


def z_spec(levels_range=(-1, 1), L=25):
    return Spec(False, levels_range[0], levels_range[1], L)


def params_for_channel(spec, l, c, C, x=None):
    """-> (pi softmaxed, mu, log_sigma), each (N,K,H,W), for channel c given already-coded channels x (N,C,H,W)."""
    N, Kp, H, W = l.shape
    K = Kp // (spec.num_params * C)
    l = l.reshape(N, spec.num_params, C, K, H, W)
    logit = l[:, 0, c]
    mu = l[:, 1, c].clone()
    log_sigma = torch.clamp(l[:, 2, c], min=LOG_SCALES_MIN)
    if spec.rgb and c != 0:
        lam = l[:, 3]
        if c == 1:
            mu += torch.sigmoid(lam[:, 0]) * x[:, 0:1]
        else:
            mu += torch.sigmoid(lam[:, 1]) * x[:, 0:1] + torch.sigmoid(lam[:, 2]) * x[:, 1:2]
    return F.softmax(logit, dim=1), mu, log_sigma


def _log_softmax(x, dim):
    m, _ = torch.max(x, dim=dim, keepdim=True)
    return x - m - torch.log(torch.sum(torch.exp(x - m), dim=dim, keepdim=True))


def _log_sum_exp(x, dim):
    m, _ = torch.max(x, dim=dim)
    mk, _ = torch.max(x, dim=dim, keepdim=True)
    return (x - mk).exp().sum(dim=dim).log().add(m)


def nll(spec, x, l):
    """x: (N,C,H,W) float targets, l: (N,Kp,H,W) -> (N,C,H,W) nats."""
    N, C, H, W = x.shape
    K = l.shape[1] // (spec.num_params * C)
    l = l.reshape(N, spec.num_params, C, K, H, W)
    logit = l[:, 0]
    mu = l[:, 1]
    log_sigma = torch.clamp(l[:, 2], min=LOG_SCALES_MIN)
    x = x.reshape(N, C, 1, H, W)
    if spec.rgb:
        lam = torch.sigmoid(l[:, 3])
        mu = torch.stack((mu[:, 0],
                          mu[:, 1] + lam[:, 0] * x[:, 0],
                          mu[:, 2] + lam[:, 1] * x[:, 0] + lam[:, 2] * x[:, 1]), dim=1)
    centered = x - mu
    inv = torch.exp(-log_sigma)
    plus_in = inv * (centered + spec.bin_width / 2)
    min_in = inv * (centered - spec.bin_width / 2)
    cdf_plus, cdf_min = torch.sigmoid(plus_in), torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    out_a = torch.log(torch.clamp(cdf_plus - cdf_min, min=1e-12))
    cond_b = (x > spec.x_max - 0.001).float()
    out_b = cond_b * log_one_minus_cdf_min + (1. - cond_b) * out_a
    cond_c = (x < spec.x_min + 0.001).float()
    log_probs = cond_c * log_cdf_plus + (1. - cond_c) * out_b
    return -_log_sum_exp(log_probs + _log_softmax(logit, dim=2), dim=2)


def sample(spec, l, C, u_mix, u_log):
    """l (N,Kp,H,W); u_mix (N,C,K,H,W), u_log (N,C,H,W): the uniforms the reference draws with uniform_(1e-5, 1 - 1e-5).
    -> x (N,C,H,W) float, not rounded (the RGB scale is clamped to [0, 255])."""
    N, Kp, H, W = l.shape
    K = Kp // (spec.num_params * C)
    l = l.reshape(N, spec.num_params, C, K, H, W)
    sel = torch.argmax(l[:, 0] - torch.log(-torch.log(u_mix)), dim=2).unsqueeze(2)          # Gumbel-max over K
    means = torch.gather(l[:, 1], 2, sel).squeeze(2)
    log_scales = torch.clamp(torch.gather(l[:, 2], 2, sel).squeeze(2), min=LOG_SCALES_MIN)
    x = means + torch.exp(log_scales) * (torch.log(u_log) - torch.log(1. - u_log))           # inverse logistic CDF
    if spec.rgb:
        assert C == 3
        coeffs = torch.sigmoid(l[:, 3])                                                      # (N,3,K,H,W)
        sel_g, sel_b = sel[:, 1], sel[:, 2]
        g_r = torch.gather(coeffs[:, 0], 1, sel_g).squeeze(1)
        b_r = torch.gather(coeffs[:, 1], 1, sel_b).squeeze(1)
        b_g = torch.gather(coeffs[:, 2], 1, sel_b).squeeze(1)
        x0 = torch.clamp(x[:, 0], 0, 255.)
        x1 = torch.clamp(x[:, 1] + g_r * x0, 0, 255.)
        x2 = torch.clamp(x[:, 2] + b_r * x0 + b_g * x1, 0, 255.)
        x = torch.stack((x0, x1, x2), dim=1)
    return x
