"""Helpers shared by the -m gpu tests."""
import numpy as np
import torch

from l3c_pytorch_amd import ops


def hip_encode_streams(tabs, syms):
    """tabs: (S, N, Lp) uint16 numpy, syms: (S, N) int16 -> list of S byte strings, coded by the HIP coder in ONE launch."""
    S, N, Lp = tabs.shape
    t = torch.from_numpy(np.ascontiguousarray(tabs).view(np.int16)).cuda().reshape(S * N, Lp)
    s = torch.from_numpy(np.ascontiguousarray(syms)).cuda()
    iv = ops.intervals_from_table(t, s, S, N)
    out, n = ops.ac_encode(iv, S, N)
    n = n.cpu().numpy()
    out = out.cpu().numpy()
    return [out[i, :n[i]].tobytes() for i in range(S)]


def hip_decode_streams(tabs, payloads, monotone):
    S, N, Lp = tabs.shape
    t = torch.from_numpy(np.ascontiguousarray(tabs).view(np.int16)).cuda().reshape(S * N, Lp)
    buf, offs, lens = ops.pack_streams(payloads)
    return ops.ac_decode(t, buf, offs, lens, S, N, monotone).cpu().numpy()


def random_tables(rng, S, N, Lp, shape=0.3):
    w = rng.gamma(shape, size=(S, N, Lp - 1)) + 1e-4
    c = np.cumsum(w, -1)
    c /= c[..., -1:]
    first = rng.randint(0, 400, size=(S, N, 1))
    tab = np.concatenate([first, np.round(c * (65535 - (Lp - 1) - first)) + np.arange(1, Lp) + first], -1)
    return tab.astype(np.int64).astype(np.uint16)


def sample_symbols(rng, tab):
    Lp = tab.shape[-1]
    hi = np.concatenate([tab[..., 1:Lp - 1].astype(np.int64), np.full(tab.shape[:-1] + (1,), 65536)], -1)
    u = rng.randint(0, 65536, size=tab.shape[:-1] + (1,))
    return np.clip((u >= hi).sum(-1), 0, Lp - 2).astype(np.int16)
