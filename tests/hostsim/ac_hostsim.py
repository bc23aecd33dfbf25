"""ctypes driver of tests/hostsim/ac_hostsim.cpp (TEST INFRASTRUCTURE: the product coder core run on the host)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SRC = os.path.join(_HERE, 'ac_hostsim.cpp')
_CORE = os.path.join(_ROOT, 'l3c-pytorch_amd', 'csrc', 'ac_core.h')
_SO = os.path.join(_HERE, '_build', 'libac_hostsim.so')
_lib = None


def _get():
    global _lib
    if _lib is None:
        newest = max(os.path.getmtime(_SRC), os.path.getmtime(_CORE))
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < newest:
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-I', os.path.dirname(_CORE), '-o', _SO, _SRC])
        lib = ctypes.CDLL(_SO)
        lib.hostsim_encode.restype = ctypes.c_longlong
        lib.hostsim_encode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
        lib.hostsim_decode.restype = None
        lib.hostsim_decode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
        lib.hostsim_step_forms_agree.restype = ctypes.c_longlong
        lib.hostsim_step_forms_agree.argtypes = [ctypes.c_longlong, ctypes.c_ulonglong]
        _lib = lib
    return _lib


def encode(tab, sym, fast=True):
    tab = np.ascontiguousarray(tab).view(np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.int16)
    out = np.zeros(2 * len(sym) + 64, np.uint8)
    n = _get().hostsim_encode(tab.ctypes.data, tab.shape[1], tab.shape[1], sym.ctypes.data, len(sym),
                              out.ctypes.data, len(out), int(fast))
    assert n <= len(out)
    return out[:n].tobytes()


def decode(tab, data, N, monotone):
    tab = np.ascontiguousarray(tab).view(np.uint16)
    buf = np.frombuffer(data, np.uint8)
    out = np.zeros(N, np.int16)
    _get().hostsim_decode(tab.ctypes.data, tab.shape[1], tab.shape[1], buf.ctypes.data if len(buf) else None,
                          len(buf), int(monotone), out.ctypes.data, N)
    return out


def step_forms_disagreements(n, seed):
    return int(_get().hostsim_step_forms_agree(n, seed))
