"""`torchac` facade -- same four functions and two flags as the reference's src/torchac/torchac.py:41-48, :87-166, backed by
the HIP range coder (csrc/ac_kernels.hip) and the HIP CDF kernel (csrc/dmll_kernels.hip) instead of the CPU coder of
torchac_backend_{cpu,gpu}.

  encode_cdf(cdf, sym) -> bytes                      cdf (1,H,W,Lp) int16, sym (N,) int16
  decode_cdf(cdf, input_string) -> (N,) int16 CPU
  encode_logistic_mixture(targets, means, log_scales, logit_probs_softmax, sym) -> bytes
  decode_logistic_mixture(targets, means, log_scales, logit_probs_softmax, input_string) -> (N,) int16 CPU

Tensors may live on the host (they are uploaded) or on the GPU; results that the reference returns on the CPU are
returned on the CPU.  The whole coder runs on the device; only the finished byte string crosses PCIe.
There is no CPU back end: CPU_SUPPORTED is False and a missing HIP library / GPU raises.
"""
import torch

from . import _lib, ops

CUDA_SUPPORTED = True    # "GPU back end available" in the reference's vocabulary (l3c.py:47-63)
CPU_SUPPORTED = False


def _dev(t, dtype=None):
    _lib.require_gpu()
    t = t.to('cuda')
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _check_cdf(cdf):
    if cdf.dim() != 4 or cdf.shape[0] != 1:
        raise RuntimeError('Invalid size for cdf! Expected 1HWLp')       # torchac.cpp:134-135
    if cdf.dtype != torch.int16:
        raise RuntimeError('cdf must be int16')
    return cdf.shape[1] * cdf.shape[2], cdf.shape[3]


def _fetch_stream(out, nbytes, s=0):
    n = int(nbytes[s].item())
    return out[s, :n].cpu().numpy().tobytes()


def _check_symbols(sym, Lp):
    """The coder indexes table rows with the symbols: out-of-range ones would read past a row (the reference does, silently)."""
    if sym.numel() and (int(sym.min()) < 0 or int(sym.max()) > Lp - 2):
        raise ValueError('symbols must lie in [0, {}], got [{}, {}]'.format(Lp - 2, int(sym.min()), int(sym.max())))


def _check_intervals(table, sym):
    """Encoder precondition on a USER table (torchac.cpp:174-207 divides the range by c_high - c_low): every coded symbol
    needs c_high > c_low.  An empty interval makes the reference emit an undecodable stream; here it could additionally emit
    more than the 16 bits per symbol the output rows are sized for -- refuse it.  Checked on the table itself (the packed interval
    word stores c_high - 1, where an empty interval at 0 is indistinguishable from c_high = 65536): c_low = cdf[sym], c_high =
    cdf[sym + 1], read as uint16, with 0 standing for 65536 only at the top symbol (torchac.cpp:181)."""
    N, Lp = table.shape
    s = sym.reshape(-1, 1).long()
    # gather on the int16 table first: two N-vectors, not two N x Lp int32 copies of the table (0.8 GB for one 768x512 channel)
    lo = table.gather(1, s)[:, 0].to(torch.int32) & 0xFFFF
    hi = table.gather(1, s + 1)[:, 0].to(torch.int32) & 0xFFFF
    hi = torch.where(s[:, 0] == Lp - 2, torch.full_like(hi, 65536), hi)
    if bool((hi <= lo).any()):
        raise ValueError('cdf is not increasing at a coded symbol (c_high <= c_low): the stream would not be decodable')


def encode_cdf(cdf, sym):
    N, Lp = _check_cdf(cdf)
    sym = _dev(sym.reshape(-1), torch.int16)
    if sym.numel() != N:
        raise RuntimeError('cdf has {} rows but {} symbols were given'.format(N, sym.numel()))
    _check_symbols(sym, Lp)
    table = _dev(cdf).reshape(N, Lp)
    _check_intervals(table, sym)
    iv = ops.intervals_from_table(table, sym.reshape(1, N), 1, N)
    out, nbytes = ops.ac_encode(iv, 1, N)
    return _fetch_stream(out, nbytes)


def decode_cdf(cdf, input_string):
    N, Lp = _check_cdf(cdf)
    table = _dev(cdf).reshape(N, Lp)
    buf, offs, lens = ops.pack_streams([bytes(input_string)])
    monotone = ops.table_is_monotone(table) if Lp <= 257 else False
    return ops.ac_decode(table, buf, offs, lens, 1, N, monotone).reshape(-1).cpu()


def _check_mixture(targets, means, log_scales, pi):
    if not (targets.is_cuda == means.is_cuda == log_scales.is_cuda == pi.is_cuda):
        raise ValueError('targets, means, log_scales, logit_probs_softmax must all be on the same device! Got '
                         '{}, {}, {}, {}.'.format(targets.device, means.device, log_scales.device, pi.device))
    if means.dim() != 4 or means.shape[0] != 1:
        raise RuntimeError('Invalid size for means! Expected 1KHW')      # torchac.cpp:242
    if means.shape != log_scales.shape or means.shape != pi.shape:
        raise RuntimeError('Invalid size for log_scales / logit_probs_softmax! Expected 1KHW')
    return means.shape[2] * means.shape[3]


def _mixture_table(targets, means, log_scales, pi):
    return ops.cdf_table_mixture(_dev(targets, torch.float32), _dev(pi, torch.float32), _dev(means, torch.float32),
                                 _dev(log_scales, torch.float32))


def encode_logistic_mixture(targets, means, log_scales, logit_probs_softmax, sym):
    N = _check_mixture(targets, means, log_scales, logit_probs_softmax)
    table, _ = _mixture_table(targets, means, log_scales, logit_probs_softmax)
    sym = _dev(sym.reshape(-1), torch.int16)
    _check_symbols(sym, table.shape[-1])
    _check_intervals(table.reshape(N, -1), sym)
    iv = ops.intervals_from_table(table.reshape(N, -1), sym.reshape(1, N), 1, N)
    out, nbytes = ops.ac_encode(iv, 1, N)
    return _fetch_stream(out, nbytes)


def decode_logistic_mixture(targets, means, log_scales, logit_probs_softmax, input_string):
    N = _check_mixture(targets, means, log_scales, logit_probs_softmax)
    table, flag = _mixture_table(targets, means, log_scales, logit_probs_softmax)
    buf, offs, lens = ops.pack_streams([bytes(input_string)])
    monotone = int(flag.item()) == 0
    return ops.ac_decode(table.reshape(N, -1), buf, offs, lens, 1, N, monotone).reshape(-1).cpu()
