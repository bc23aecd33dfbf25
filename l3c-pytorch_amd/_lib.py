"""ctypes binding of the C ABI in include/l3c_hip.h (libl3c_hip.so, built by csrc/build.py).

This is the reference-side binding a maintainer would add in place of `import torchac_backend_gpu`
(src/torchac/torchac.py:36-48): raw device pointers (`tensor.data_ptr()`), sizes, and the current HIP stream.
There is no CPU implementation behind it: a missing library or a CPU tensor raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# L3C_LIB: a development build of the same library (e.g. the -DL3C_WINO_TIMELINE one of csrc/build.py --timeline)
LIB_PATH = os.environ.get('L3C_LIB') or os.path.join(_HERE, 'csrc', 'libl3c_hip.so')

c_i64, c_int, c_vp, c_f32 = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_float


class L3CError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """l3c_conv_desc (include/l3c_hip.h)."""
    _fields_ = [('inp', c_vp), ('in_cstride', c_int), ('in_coff', c_int),
                ('packed_w', c_vp), ('bias', c_vp),
                ('residual', c_vp), ('res_cstride', c_int), ('res_coff', c_int),
                ('out', c_vp), ('out_cstride', c_int), ('out_coff', c_int),
                ('B', c_int), ('Hin', c_int), ('Win', c_int), ('Cin', c_int), ('Cout', c_int),
                ('KS', c_int), ('stride', c_int), ('dilation', c_int), ('epilogue', c_int)]


class AcGroup(ctypes.Structure):
    """l3c_ac_group (include/l3c_hip.h)."""
    _fields_ = [('intervals', c_vp), ('out', c_vp), ('out_nbytes', c_vp),
                ('n_streams', c_i64), ('n_sym', c_i64), ('out_stride_bytes', c_i64)]


class ContainerScale(ctypes.Structure):
    """l3c_container_scale (include/l3c_hip.h)."""
    _fields_ = [('out', c_vp), ('nbytes', c_vp), ('stride', c_i64), ('C', c_int), ('H', c_int), ('W', c_int)]


class AcDecodePart(ctypes.Structure):
    """l3c_ac_decode_part (include/l3c_hip.h)."""
    _fields_ = [('cdf', c_vp), ('Lp', c_int), ('in_', c_vp), ('in_offsets', c_vp), ('in_nbytes', c_vp),
                ('n_streams', c_i64), ('n_sym', c_i64), ('not_monotone_flag', c_vp), ('state_in', c_vp),
                ('state_out', c_vp), ('final_chunk', c_int), ('sym_out', c_vp), ('sym_stride', c_i64), ('sym_offset', c_i64),
                ('window_stats_in', c_vp), ('window_stats_out', c_vp), ('P', c_vp), ('sym_all', c_vp), ('targets', c_vp),
                ('HW', c_i64), ('pix0', c_i64), ('C', c_int), ('K', c_int), ('c', c_int),
                ('r_npix', c_vp), ('r_table_off', c_vp), ('r_pixbase', c_vp), ('r_hw', c_vp), ('r_pix0', c_vp),
                ('r_C', c_int), ('r_c', c_int), ('r_table_bytes', c_i64)]


class TablePart(ctypes.Structure):
    """l3c_table_part (include/l3c_hip.h)."""
    _fields_ = [('c', c_int), ('pix0', c_i64), ('npix', c_i64), ('cdf', c_vp), ('not_monotone', c_vp), ('window_stats', c_vp)]


class RgbDecodeDesc(ctypes.Structure):
    """l3c_rgb_decode_desc (include/l3c_hip.h)."""
    _fields_ = [('P', c_vp), ('targets', c_vp), ('sym', c_vp), ('B', c_i64), ('HW', c_i64), ('K', c_int),
                ('in_', c_vp), ('in_offsets', c_vp), ('in_nbytes', c_vp), ('n_chunks', c_int),
                ('chunk_pix0_host', ctypes.POINTER(c_i64)), ('chunk_npix_host', ctypes.POINTER(c_i64)),
                ('lag', c_int), ('window_mode', c_int), ('workspace', c_vp), ('workspace_bytes', c_i64)]


class RaggedBatch(ctypes.Structure):
    """l3c_ragged_batch (include/l3c_hip.h)."""
    _fields_ = [('B', c_i64), ('max_hw', c_i64), ('pixbase', c_vp), ('hw', c_vp)]


class RaggedPart(ctypes.Structure):
    """l3c_ragged_part (include/l3c_hip.h)."""
    _fields_ = [('pix0', c_vp), ('npix', c_vp), ('table_off', c_vp)]


class RgbRaggedDesc(ctypes.Structure):
    """l3c_rgb_ragged_desc (include/l3c_hip.h)."""
    _fields_ = [('P', c_vp), ('targets', c_vp), ('sym', c_vp), ('B', c_i64), ('hw_host', ctypes.POINTER(c_i64)), ('K', c_int),
                ('in_', c_vp), ('in_offsets', c_vp), ('in_nbytes', c_vp), ('n_chunks', c_int),
                ('chunk_pix0_host', ctypes.POINTER(c_i64)), ('chunk_npix_host', ctypes.POINTER(c_i64)), ('tables_dev', c_vp),
                ('lag', c_int), ('window_mode', c_int), ('workspace', c_vp), ('workspace_bytes', c_i64)]


EPI_RELU, EPI_RESIDUAL, EPI_PIXEL_SHUFFLE = 1, 2, 4
ABI_VERSION = 4      # include/l3c_hip.h: L3C_ABI_VERSION (4: grouped tables, l3c_decode_rgb, l3c_container_read; no canvas batches)

# name -> (restype, argtypes); must list every symbol include/l3c_hip.h declares (tests/test_abi.py checks)
PROTOTYPES = {
    'l3c_abi_version': (c_int, []),
    'l3c_bitstream_generation': (c_int, []),
    'l3c_last_error': (ctypes.c_char_p, []),
    'l3c_device_info': (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_int), ctypes.c_char_p, c_int]),
    'l3c_stream_create_cu_range': (c_int, [c_int, c_int, ctypes.POINTER(c_vp)]),
    'l3c_stream_create_cu_mask': (c_int, [ctypes.POINTER(ctypes.c_uint32), c_int, ctypes.POINTER(c_vp)]),
    'l3c_stream_destroy': (c_int, [c_vp]),
    'l3c_interval_words': (c_i64, [c_i64, c_i64]),
    'l3c_ac_intervals_from_table': (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'l3c_ac_max_bytes': (c_i64, [c_i64]),
    'l3c_ac_encode_workspace_bytes': (c_i64, [c_i64]),
    'l3c_ac_encode': (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'l3c_ac_encode_groups_workspace_bytes': (c_i64, [c_int, c_i64]),
    'l3c_ac_encode_groups': (c_int, [ctypes.POINTER(AcGroup), c_int, c_vp, c_vp]),
    'l3c_container_write': (c_int, [ctypes.POINTER(ContainerScale), c_int, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'l3c_container_read': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_uint32, c_vp, c_vp]),
    'l3c_dmll_cdf_table_parts': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, ctypes.POINTER(TablePart), c_int, c_vp]),
    'l3c_decode_rgb_workspace_bytes': (c_i64, [c_i64, c_i64, c_int, c_int]),
    'l3c_decode_rgb_stats_offset': (c_i64, [c_i64, c_i64, c_int, c_int]),
    'l3c_decode_rgb': (c_int, [ctypes.POINTER(RgbDecodeDesc), c_vp, c_vp]),
    'l3c_decode_rgb_ragged_workspace_bytes': (c_i64, [c_i64, c_i64, c_int, c_int]),
    'l3c_decode_rgb_ragged': (c_int, [ctypes.POINTER(RgbRaggedDesc), c_vp, c_vp]),
    'l3c_dmll_cdf_table_ragged': (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(RaggedBatch), c_int, c_int, c_int, c_int, ctypes.POINTER(TablePart),
                                          ctypes.POINTER(RaggedPart), c_int, c_vp]),
    'l3c_ac_decode_state_bytes': (c_i64, []),
    'l3c_ac_decode_chunks': (c_int, [ctypes.POINTER(AcDecodePart), c_int, c_vp]),
    'l3c_ac_decode': (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp]),
    'l3c_cdf_check_monotone': (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    'l3c_dmll_channel_params': (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'l3c_cdf_table_mixture': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'l3c_dmll_encode_intervals': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'l3c_dmll_cdf_table': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    'l3c_dmll_sample': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    'l3c_dmll_nll': (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_int, c_vp, c_vp]),
    'l3c_conv_packed_words': (c_i64, [c_int, c_int, c_int]),
    'l3c_conv_pack_weights': (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'l3c_conv_mfma': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_conv_wino4_packed_words': (c_i64, [c_int, c_int]),
    'l3c_conv_wino4_pack_weights': (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    'l3c_conv_wino4': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_conv_wino4_phase': (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int, c_vp]),
    'l3c_conv_wino4_stride2': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_conv_wino4_set_tiles_per_block': (c_int, [c_int]),
    'l3c_conv_pw_packed_words': (c_i64, [c_int, c_int]),
    'l3c_conv_pw_pack_weights': (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    'l3c_conv_pw': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_conv_direct': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_rgb_head': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    'l3c_to_q_quantize': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'l3c_dec_head': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp]),
    'l3c_meanshift_planar': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'l3c_rgb_to_u8': (c_int, [c_vp, ctypes.POINTER(c_f32), c_i64, c_i64, c_vp, c_vp]),
    'l3c_resample_u8': (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    'l3c_u8_to_sym_bn': (c_int, [c_vp, ctypes.POINTER(c_f32), c_i64, c_i64, c_vp, c_vp, c_vp]),
    'l3c_sym_to_bn': (c_int, [c_vp, c_i64, c_f32, c_f32, c_vp, c_vp]),
}

# include/l3c_xcheck.h: the TEST-ONLY cross-check library (round-1/2 Winograd F(2x2,3x3) kernel); see load_xcheck()
XCHECK_PROTOTYPES = {
    'l3c_conv_wino_packed_words': (c_i64, [c_int, c_int]),
    'l3c_conv_wino_pack_weights': (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    'l3c_conv_wino': (c_int, [ctypes.POINTER(ConvDesc), c_vp]),
    'l3c_conv_wino_set_tiles_per_block': (c_int, [c_int]),
    'l3c_xcheck_sigmoid_exhaustive': (c_int, [c_vp, c_vp, c_vp]),
    'l3c_conv_wino4w_packed_words': (c_i64, [c_int, c_int]),
    'l3c_conv_wino4w_pack_weights': (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    'l3c_conv_wino4w': (c_int, [ctypes.POINTER(ConvDesc), c_int, c_vp]),
}
XCHECK_LIB_PATH = os.environ.get('L3C_XCHECK_LIB') or os.path.join(_HERE, 'csrc', 'libl3c_hip_xcheck.so')   # (L3C_XCHECK_LIB: a development variant)

_lib = None
_xcheck = None


def load():
    """The shared library, with prototypes installed.  Raises L3CError when it has not been built."""
    global _lib
    if _lib is None:
        _snapshot_queues()
        if not os.path.isfile(LIB_PATH):
            raise L3CError('libl3c_hip.so not found at {} -- build it with `python l3c-pytorch_amd/csrc/build.py` '
                           '(or __graft_entry__.build()); there is no CPU fallback.'.format(LIB_PATH))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.l3c_abi_version() != ABI_VERSION:
            raise L3CError('ABI version mismatch: library {} != binding {}'.format(lib.l3c_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def load_xcheck():
    """libl3c_hip_xcheck.so -- for tests and development probes ONLY: an independent second implementation of the 3x3 convolution
    to compare the product's kernels with.  Nothing under l3c-pytorch_amd/ calls this on the product path."""
    global _xcheck
    if _xcheck is None:
        if not os.path.isfile(XCHECK_LIB_PATH):
            raise L3CError('libl3c_hip_xcheck.so not found at {} -- build it with `python l3c-pytorch_amd/csrc/build.py`'.format(XCHECK_LIB_PATH))
        lib = ctypes.CDLL(XCHECK_LIB_PATH)
        lib.l3c_last_error.restype = ctypes.c_char_p
        for name, (res, args) in XCHECK_PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _xcheck = lib
    return _xcheck


def call_xcheck(name, *args):
    rc = getattr(load_xcheck(), name)(*args)
    if rc != 0:
        raise L3CError('libl3c_hip_xcheck: {} (status {})'.format(load_xcheck().l3c_last_error().decode(), rc))


def check(rc):
    if rc != 0:
        raise L3CError('libl3c_hip: {} (status {})'.format(load().l3c_last_error().decode(), rc))


def call(name, *args):
    check(getattr(load(), name)(*args))


def _snapshot_queues():
    # the package's first HIP touch: remember what GPU_MAX_HW_QUEUES said when the runtime (may have) started -- helpers/runtime.py
    from .helpers import runtime
    runtime.snapshot_hw_queues()


def require_gpu():
    _snapshot_queues()
    if not torch.cuda.is_available():
        raise L3CError('no HIP device visible: the l3c-pytorch_amd compute path runs on the GPU only (no CPU fallback)')


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise L3CError('expected a GPU tensor, got {} -- there is no CPU fallback'.format(t.device))
    if not t.is_contiguous():
        raise L3CError('expected a contiguous tensor')
    if dtype is not None and t.dtype != dtype:
        raise L3CError('expected dtype {}, got {}'.format(dtype, t.dtype))
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def cu_range_stream(first_cu, n_cu):
    """torch stream object whose kernels are confined to CUs [first_cu, first_cu + n_cu)."""
    require_gpu()
    h = c_vp()
    call('l3c_stream_create_cu_range', first_cu, n_cu, ctypes.byref(h))
    return torch.cuda.ExternalStream(h.value)


def cu_mask_stream(cus):
    """torch stream object whose kernels are confined to the compute units listed in `cus` (hipExtStreamCreateWithCUMask)."""
    require_gpu()
    _, n_cu, _ = device_info()
    words = (ctypes.c_uint32 * ((n_cu + 31) // 32))()
    for cu in cus:
        if not 0 <= cu < n_cu:
            raise L3CError('compute unit {} outside the device ({} CUs)'.format(cu, n_cu))
        words[cu >> 5] |= 1 << (cu & 31)
    h = c_vp()
    call('l3c_stream_create_cu_mask', words, len(words), ctypes.byref(h))
    return torch.cuda.ExternalStream(h.value)


def device_info():
    name = ctypes.create_string_buffer(256)
    arch = ctypes.create_string_buffer(256)
    ncu = c_int(0)
    call('l3c_device_info', name, 256, ctypes.byref(ncu), arch, 256)
    return name.value.decode(), ncu.value, arch.value.decode()
