"""The C-ABI library loads and exports every symbol include/l3c_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

import l3c_pytorch_amd  # noqa: F401
from l3c_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'l3c_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(l3c_[a-z0-9_]+)\s*\(', src)))


def test_library_is_built_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.l3c_abi_version() == 1


def test_pure_host_entry_points():
    lib = _lib.load()
    assert lib.l3c_interval_words(3, 100) == 2 * 3 * 64
    assert lib.l3c_interval_words(48, 393216) == 393216 * 48
    assert lib.l3c_ac_max_bytes(393216) >= 2 * 393216 + 8 and lib.l3c_ac_max_bytes(393216) % 4 == 0
    assert lib.l3c_conv_packed_words(64, 64, 3) == 64 * 64 * 9
    assert lib.l3c_conv_packed_words(120, 192, 1) == 128 * 192


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.l3c_ac_encode(None, 1, 1, None, 0, None, None, None)
    assert rc == -1 and b'null pointer' in lib.l3c_last_error()
    with pytest.raises(_lib.L3CError):
        _lib.check(rc)


def test_no_cpu_fallback():
    import torch
    with pytest.raises(_lib.L3CError):
        _lib.ptr(torch.zeros(4))
    if not torch.cuda.is_available():
        with pytest.raises(_lib.L3CError):
            _lib.require_gpu()


def test_ring_decoder_prefetch_registers_untouched():
    """The ring decoder's LDS row prefetch is inline asm the compiler cannot see into; tools/check_asm_prefetch.py rebuilds
    the CFG of the compiled kernels and proves no compiler-generated instruction names a register with a read in flight."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_asm_prefetch.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
