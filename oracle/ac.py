"""ctypes front end of oracle/ac_oracle.c (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'ac_oracle.c')
_SO = os.path.join(_HERE, '_build', 'libac_oracle.so')
_lib = None


def build(force=False):
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', _SO, _SRC])
    return _SO


def _get():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        lib.l3c_oracle_ac_encode.restype = ctypes.c_longlong
        lib.l3c_oracle_ac_encode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_longlong,
                                             ctypes.c_void_p, ctypes.c_longlong]
        lib.l3c_oracle_ac_decode.restype = None
        lib.l3c_oracle_ac_decode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_longlong,
                                             ctypes.c_void_p, ctypes.c_longlong]
        _lib = lib
    return _lib


def _table(cdf):
    """Accepts (N, Lp) / (1, H, W, Lp) int16|uint16, or a single row (Lp,) which is broadcast."""
    cdf = np.ascontiguousarray(np.asarray(cdf)).view(np.uint16)
    if cdf.ndim == 1:
        return cdf, 0, cdf.shape[0]
    cdf = cdf.reshape(-1, cdf.shape[-1])
    return cdf, cdf.shape[1], cdf.shape[1]


def encode(cdf, sym):
    """cdf: see _table; sym: N int16 -> bytes  (reference: torchac.encode_cdf, torchac.py:87-96)."""
    tab, stride, Lp = _table(cdf)
    sym = np.ascontiguousarray(np.asarray(sym).reshape(-1), dtype=np.int16)
    N = sym.shape[0]
    if stride:
        assert tab.shape[0] == N, (tab.shape, N)
    assert sym.min() >= 0 and sym.max() <= Lp - 2
    cap = 2 * N + 64
    out = np.empty(cap, dtype=np.uint8)
    n = _get().l3c_oracle_ac_encode(tab.ctypes.data, stride, Lp, sym.ctypes.data, N, out.ctypes.data, cap)
    assert n <= cap
    return out[:n].tobytes()


def decode(cdf, data, N=None):
    """-> N int16 symbols  (reference: torchac.decode_cdf, torchac.py:99-108)."""
    tab, stride, Lp = _table(cdf)
    if stride:
        N = tab.shape[0]
    assert N is not None
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(N, dtype=np.int16)
    _get().l3c_oracle_ac_decode(tab.ctypes.data, stride, Lp, buf.ctypes.data if len(buf) else None, len(buf),
                                out.ctypes.data, N)
    return out
