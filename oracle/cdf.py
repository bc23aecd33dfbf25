"""CDF-table oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates, with the same torch CPU ops and in the same order, the reference's pure-PyTorch statement of its CUDA
kernel `calculate_cdf_kernel` (torchac_kernel.cu:26-76): `_get_uint16_cdf`, `_get_C_cur_weighted`, `_get_C_cur`,
`_renorm_cast_cdf_` (torchac/torchac.py:174-213), and the uniform-prior table of the coarsest scale
(`_get_uniform_pr` + `_get_cdf_from_pr`, bitcoding/bitcoding.py:297-323).
"""
import torch


def mixture_cdf_float(pi, targets, mu, log_sigma):
    """pi, mu, log_sigma: (N,K,H,W) f32; targets: (Lp,) f32 -> (N,H,W,Lp) f32.   torchac.py:181-200"""
    inv_stdv = torch.exp(-log_sigma).unsqueeze(-1)
    centered = targets - mu.unsqueeze(-1)
    c = centered.mul(inv_stdv).sigmoid()                      # N K H W Lp
    return c.mul(pi.unsqueeze(-1)).sum(1)                     # N H W Lp


def renorm_cast(cdf_float, precision=16):
    """round(cdf * (2^16 - (Lp-1))) -> int16 (wraps), + arange(Lp) (wraps).   torchac.py:203-213"""
    Lp = cdf_float.shape[-1]
    f = torch.tensor(2, dtype=torch.float32).pow_(precision)
    cdf = cdf_float.mul(f - (Lp - 1)).round().to(dtype=torch.int16)
    cdf.add_(torch.arange(Lp, dtype=torch.int16))
    return cdf


def mixture_cdf_table(pi, targets, mu, log_sigma, max_elems=1 << 27):
    """-> (N,H,W,Lp) int16, the table the reference feeds its coder.   torchac.py:174-178
    Built in row blocks so that the N*K*H*W*Lp intermediate stays below `max_elems` floats (the reference allocates it
    whole: 4 GB for a 768x512 RGB channel); every op is elementwise in H, so the result is identical."""
    N, K, H, W = pi.shape
    rows = max(1, min(H, max_elems // max(1, N * K * W * targets.shape[0])))
    parts = [renorm_cast(mixture_cdf_float(pi[:, :, r:r + rows], targets, mu[:, :, r:r + rows], log_sigma[:, :, r:r + rows]))
             for r in range(0, H, rows)]
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


def uniform_cdf_table(H, W, L):
    """-> (1,H,W,L+1) int16.   bitcoding.py:297-323"""
    histo = torch.ones(L, dtype=torch.float32) / L
    pr = torch.ones(1, H, W, L) * histo
    cdf = torch.cumsum(pr, -1).mul_(2 ** 16).round()
    cdf = torch.cat((torch.zeros((1, H, W, 1), dtype=cdf.dtype), cdf), dim=-1)
    return cdf.to(dtype=torch.int16)


def coding_targets(x_min, x_max, L):
    """The Lp bin edges the coder evaluates the CDF at.   coders_helpers.py:42-44"""
    bin_width = (x_max - x_min) / (L - 1)
    return torch.linspace(x_min - bin_width / 2, x_max + bin_width / 2, L + 1, dtype=torch.float32)
