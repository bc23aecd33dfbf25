"""The C-ABI library loads and exports every symbol include/l3c_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

import l3c_pytorch_amd  # noqa: F401
from l3c_pytorch_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header='l3c_hip.h'):
    src = open(os.path.join(ROOT, 'include', header)).read()
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
    assert lib.l3c_abi_version() == _lib.ABI_VERSION == 4
    hdr = open(os.path.join(ROOT, 'include', 'l3c_hip.h')).read()
    assert lib.l3c_bitstream_generation() == int(re.search(r'#define L3C_BITSTREAM_GENERATION (\d+)', hdr).group(1)) >= 3
    # the product library holds ONE generation of convolution kernels: the F(2x2,3x3) cross-check kernel lives in the test-only
    # library (include/l3c_xcheck.h), which exports exactly what that header declares
    assert not any(n.startswith('l3c_conv_wino_') or n == 'l3c_conv_wino' for n in names)
    assert not hasattr(lib, 'l3c_conv_wino')
    x = _lib.load_xcheck()
    xnames = _declared('l3c_xcheck.h')
    assert sorted(_lib.XCHECK_PROTOTYPES) == xnames and len(xnames) == 8
    for n in xnames:
        assert hasattr(x, n), n


def test_pure_host_entry_points():
    lib = _lib.load()
    assert lib.l3c_interval_words(3, 100) == 2 * 3 * 64 * 2      # two role words per symbol
    assert lib.l3c_interval_words(48, 393216) == 393216 * 48 * 2
    assert lib.l3c_ac_max_bytes(393216) >= 2 * 393216 + 8 and lib.l3c_ac_max_bytes(393216) % 4 == 0
    assert lib.l3c_conv_packed_words(64, 64, 3) == 64 * 64 * 9
    assert lib.l3c_conv_packed_words(120, 192, 1) == 128 * 192


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.l3c_ac_encode(None, 1, 1, None, 0, None, None, None)
    assert rc == -1 and b'null pointer' in lib.l3c_last_error()
    with pytest.raises(_lib.L3CError):
        _lib.check(rc)


def test_argument_validation_of_every_family_precedes_any_launch():
    """Status codes + messages instead of crashes, checked WITHOUT a GPU: validation happens before any HIP call.  Fake but
    well-aligned non-null pointers stand in for device memory (they are never dereferenced on these paths)."""
    import ctypes
    lib = _lib.load()
    xlib = _lib.load_xcheck()
    fake = 0x1000

    def err():
        return lib.l3c_last_error().decode()

    def xerr():
        return xlib.l3c_last_error().decode()

    # range coder
    assert lib.l3c_ac_decode(fake, 300, 300, fake, fake, fake, 1, 1, 1, fake, None) == -1 and 'Lp out of range' in err()
    assert lib.l3c_ac_decode(fake, 7, 26, fake, fake, fake, 1, 1, 1, fake, None) == -1 and 'row_stride' in err()
    assert lib.l3c_ac_decode(fake, 26, 26, fake + 2, fake, fake, 1, 1, 1, fake, None) == -1 and '4-byte aligned' in err()
    assert lib.l3c_ac_encode(fake, 0, 5, fake, 64, fake, fake, None) == -1
    assert lib.l3c_ac_encode_groups(None, 1, fake, None) == -1 and 'bad arguments' in err()
    grp = (_lib.AcGroup * 1)(_lib.AcGroup(fake, fake, fake, 4, 100, 8))        # stride smaller than l3c_ac_max_bytes(100)
    assert lib.l3c_ac_encode_groups(grp, 1, fake, None) == -1 and 'stride too small' in err()
    part = _lib.AcDecodePart(fake, 26, fake, fake, fake, 2, 10, None, fake, fake, 0, fake, 10, 0)
    assert lib.l3c_ac_decode_chunks((_lib.AcDecodePart * 1)(part), 1, None) == -1 and 'must differ' in err()
    assert lib.l3c_ac_decode_chunks((_lib.AcDecodePart * 1)(part), 9, None) == -1 and 'parts per call' in err()
    # window rows (ABI 3): only for the 256-symbol alphabet, and only with everything a missed row needs
    assert lib.l3c_dmll_cdf_table(fake, fake, fake, 1, 10, 5, 10, 0, 1, 0, 10, 26, fake, None, fake, None) == -1 and 'window rows' in err()
    wpart = _lib.AcDecodePart(fake, 257, fake, fake, fake, 2, 10, fake, None, fake, 1, fake, 10, 0)
    wpart.window_stats_in, wpart.window_stats_out = fake, fake
    wpart.C, wpart.K, wpart.c, wpart.HW, wpart.pix0 = 3, 10, 0, 100, 0
    assert lib.l3c_ac_decode_chunks((_lib.AcDecodePart * 1)(wpart), 1, None) == -1 and 'window rows' in err()      # P / targets missing
    assert lib.l3c_ac_decode_state_bytes() == 32
    # mixture head
    assert lib.l3c_dmll_nll(fake, fake, 1, 10, 3, 17, 1, 0.0, 255.0, 256, fake, None) == -1 and 'K out of range' in err()
    assert lib.l3c_dmll_nll(fake, fake, 1, 10, 5, 10, 1, 0.0, 255.0, 256, fake, None) == -1 and 'C == 3' in err()
    assert lib.l3c_dmll_cdf_table(fake, None, fake, 1, 10, 3, 10, 1, 1, 0, 10, 257, fake, None, None, None) == -1 and 'decoded so far' in err()
    assert lib.l3c_dmll_cdf_table(fake, fake, fake, 1, 10, 3, 10, 1, 1, 5, 10, 257, fake, None, None, None) == -1 and 'outside the image' in err()
    assert lib.l3c_dmll_sample(fake, fake, None, 1, 10, 3, 10, 1, fake, None) == -1 and 'null pointer' in err()
    # convolution
    assert lib.l3c_conv_pack_weights(fake, 64, 60, 3, fake, None) == -1 and 'multiple of 8' in err()
    assert lib.l3c_conv_pack_weights(fake, 64, 64, 7, fake, None) == -1 and 'KS must be' in err()
    d = _lib.ConvDesc()
    d.inp = d.packed_w = d.bias = d.out = fake
    d.B, d.Hin, d.Win, d.Cin, d.Cout, d.KS, d.stride, d.dilation = 1, 8, 8, 64, 64, 5, 1, 1
    d.in_cstride = 64
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and '3x3, stride 1 only' in xerr()
    d.KS, d.dilation = 3, 3
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and 'dilation must be' in xerr()
    d.dilation, d.Cin = 1, 60
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and 'multiple of 16' in xerr()
    d.Cin, d.out_cstride = 64, 62
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and '16-byte stores' in xerr()
    d.out_cstride, d.out = 64, fake + 4
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and '16-byte aligned' in xerr()
    d.out, d.Cout = fake, 62
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and 'multiple of 4' in xerr()
    d.Cout, d.epilogue = 64, 0x100
    assert xlib.l3c_conv_wino(ctypes.byref(d), None) == -1 and 'unknown epilogue bits' in xerr()
    assert lib.l3c_conv_mfma(ctypes.byref(d), None) == -1 and 'unknown epilogue bits' in err()      # no probe kernels in the product
    d.epilogue = 0
    assert xlib.l3c_conv_wino_set_tiles_per_block(2) == 0 and xlib.l3c_conv_wino_set_tiles_per_block(0) == 2
    assert xlib.l3c_conv_wino_packed_words(64, 64) == 16 * 64 * 64 and xlib.l3c_conv_wino_packed_words(120, 64) == 16 * 128 * 64
    # container
    sc = (_lib.ContainerScale * 1)(_lib.ContainerScale(fake, fake, 6, 5, 8, 8))
    assert lib.l3c_container_write(sc, 1, 2, fake, fake, fake, None) == -1 and '4-byte aligned' in err()
    sc = (_lib.ContainerScale * 1)(_lib.ContainerScale(fake, fake, 8, 5, 70000, 8))
    assert lib.l3c_container_write(sc, 1, 2, fake, fake, fake, None) == -1 and 'u16' in err()
    assert lib.l3c_container_write(sc, 9, 2, fake, fake, fake, None) == -1 and 'scales' in err()
    del ctypes


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


def test_no_store_data_hazard_in_isa():
    """gfx950: a VALU write right behind a 16-byte buffer store with a register soffset corrupts the stored data and hipcc
    inserts no wait state there (found with conv_wino.hip); tools/check_store_hazard.py scans every kernel's ISA."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_store_hazard.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_mfma_kernels_stay_inside_their_register_budget():
    """conv_wino4_kernel holds 144 accumulators per lane at two blocks per CU: every variant the network launches must fit 256 VGPRs
    without scratch spills (the allocator once spilled 7 registers of the residual variant's epilogue); tools/check_registers.py reads the
    code-object metadata of the ISA compiled with the product flags."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_registers.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
