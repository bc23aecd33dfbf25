"""ctypes driver of tests/hostsim/window_hostsim.hip (TEST INFRASTRUCTURE: the product's window-row rules run on the host)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SRC = os.path.join(_HERE, 'window_hostsim.hip')
_CORE = os.path.join(_ROOT, 'l3c-pytorch_amd', 'csrc', 'dmll_core.h')
_SO = os.path.join(_HERE, '_build', 'libwindow_hostsim.so')
_lib = None


def get():
    global _lib
    if _lib is None:
        newest = max(os.path.getmtime(_SRC), os.path.getmtime(_CORE))
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < newest:
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            hipcc = os.environ.get('HIPCC') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc')
            subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-I', os.path.dirname(_CORE), '-o', _SO, _SRC])
        lib = ctypes.CDLL(_SO)
        for name in ('hostsim_win_lp', 'hostsim_win_top', 'hostsim_win_max_offset', 'hostsim_win_bad'):
            getattr(lib, name).restype = ctypes.c_int
            getattr(lib, name).argtypes = []
        lib.hostsim_use_window.argtypes = [ctypes.c_int]
        lib.hostsim_window_stat.argtypes = [ctypes.c_uint] * 3
        lib.hostsim_window_miss.argtypes = [ctypes.c_uint] * 2
        lib.hostsim_window_would_miss.argtypes = [ctypes.c_uint] * 2
        lib.hostsim_window_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        for name in ('hostsim_use_window', 'hostsim_window_stat', 'hostsim_window_miss', 'hostsim_window_would_miss', 'hostsim_window_decode'):
            getattr(lib, name).restype = ctypes.c_int
        _lib = lib
    return _lib
