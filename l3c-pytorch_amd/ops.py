"""Tensor-level wrappers of the HIP entry points (include/l3c_hip.h).  torch is only the owner of device memory and of
the current stream here; every function enqueues kernels from libl3c_hip.so and nothing else."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr, stream

# Optional per-launch timing of the MFMA conv kernels (bench.py's roofline leg): when PROFILE is a list, every conv launch
# appends (kernel key, algorithmic FLOPs, algorithmic HBM bytes, start event, end event), the events being recorded on the launch
# stream.
PROFILE = None
PROFILE_DETAIL = bool(os.environ.get('L3C_PROFILE_DETAIL'))   # split the keys by layer shape (development)


# ---- convolution stack ------------------------------------------------------------------------------------------------
#
# ONE dispatch (no environment switches: encoder and decoder of a file must compute bit-identical P, so the kernel a layer runs on
# is a function of the layer and the tensors alone):
#   3x3 stride 1 (dilation 1, 2, 4), Cin % 16 == 0   -> Winograd F(4x4,3x3), csrc/conv_wino4.hip
#   5x5 stride 2, even input size                     -> four 3x3 polyphase convolutions in one launch of the same kernel
#   1x1, Cin % 64 == 0, Cout <= 160, bias only        -> pointwise GEMM, csrc/conv_pw.hip
#   anything else with Cin % 16 == 0                  -> implicit GEMM, csrc/conv_mfma.hip (odd sizes, unaligned channel slices)
# `impl=` forces one kernel for the tests and probes: 'wino4', 'poly5', 'poly5x4' (four accumulating phase launches), 'gemm',
# 'direct' (plain-VALU cross-check in the product library), 'wino2' (the F(2x2,3x3) kernel of the TEST-ONLY libl3c_hip_xcheck.so).


class PackedConv(object):
    """One conv layer resident on the device: OIHW weights, bias, and the packed copies of the kernels that can run it."""

    def __init__(self, weight, bias, stride=1, dilation=1):
        _lib.require_gpu()
        lib = _lib.load()
        self.Cout, self.Cin, self.KS, _ = weight.shape
        self.stride, self.dilation = stride, dilation
        self.weight = weight.detach().to('cuda', torch.float32).contiguous()
        self.bias = bias.detach().to('cuda', torch.float32).contiguous()
        self.packed = None            # implicit GEMM (generic form)
        if self.Cin % 16 == 0:
            self.packed = torch.empty(lib.l3c_conv_packed_words(self.Cout, self.Cin, self.KS), dtype=torch.float32, device='cuda')
            call('l3c_conv_pack_weights', ptr(self.weight), self.Cout, self.Cin, self.KS, ptr(self.packed), stream())
        self.packed_wino4 = None      # G g G^T of F(4x4,3x3)
        if self.KS == 3 and stride == 1 and dilation in (1, 2, 4) and self.Cin % 16 == 0:
            self.packed_wino4 = self._pack_wino4(self.weight)
        # 5x5 stride 2 padding 2 = four 3x3 stride-1 convolutions of the input's 2x2 phases (include/l3c_hip.h, l3c_conv_wino4_stride2):
        # phase kernel w_ab[u][v] = w[2u+a][2v+b], zero where the index exceeds 4; all four in ONE launch: the phase kernels
        # concatenated along the input-channel axis
        self.packed_poly_fused = None
        if self.KS == 5 and stride == 2 and dilation == 1 and self.Cin % 16 == 0 and self.Cout % 4 == 0:
            w_cat = torch.zeros(self.Cout, 4 * self.Cin, 3, 3, dtype=torch.float32, device='cuda')
            for k, sub in enumerate(self._phase_kernels()):
                w_cat[:, k * self.Cin:(k + 1) * self.Cin, :sub.shape[2], :sub.shape[3]] = sub
            self.packed_poly_fused = self._pack_wino4(w_cat)
        self.packed_pw = None
        if self.KS == 1 and stride == 1 and self.Cin % 64 == 0 and self.Cout <= 160:
            self.packed_pw = torch.empty(lib.l3c_conv_pw_packed_words(self.Cout, self.Cin), dtype=torch.float32, device='cuda')
            call('l3c_conv_pw_pack_weights', ptr(self.weight), self.Cout, self.Cin, ptr(self.packed_pw), stream())
        self._packed_poly = self._packed_wino2 = self._zero_bias = self._packed_wino4_shuffle = None

    def _pack_wino4(self, w):
        packed = torch.empty(_lib.load().l3c_conv_wino4_packed_words(w.shape[0], w.shape[1]), dtype=torch.float32, device='cuda')
        call('l3c_conv_wino4_pack_weights', ptr(w.contiguous()), w.shape[0], w.shape[1], ptr(packed), stream())
        return packed

    def packed_wino4_shuffle(self):
        """the 64 -> 256 PixelShuffle tail on the Winograd kernel (include/l3c_hip.h, L3C_EPI_PIXEL_SHUFFLE): G g G^T of the weights in
        sub-pixel-major order -- row 64 s + oc = the layer's row 4 oc + s -- packed on first use"""
        if self._packed_wino4_shuffle is None:
            assert self.Cout == 256
            w = self.weight.reshape(64, 4, self.Cin, 3, 3).permute(1, 0, 2, 3, 4).reshape(256, self.Cin, 3, 3)
            self._packed_wino4_shuffle = self._pack_wino4(w)
        return self._packed_wino4_shuffle

    def _phase_kernels(self):
        return [self.weight[:, :, a::2, b::2] for a in (0, 1) for b in (0, 1)]

    def packed_poly(self):
        """tests only ('poly5x4'): the four phase kernels packed one by one, for l3c_conv_wino4_phase"""
        if self._packed_poly is None:
            self._packed_poly = []
            for sub in self._phase_kernels():
                wp = torch.zeros(self.Cout, self.Cin, 3, 3, dtype=torch.float32, device='cuda')
                wp[:, :, :sub.shape[2], :sub.shape[3]] = sub
                self._packed_poly.append(self._pack_wino4(wp))
            self._zero_bias = torch.zeros(self.Cout, dtype=torch.float32, device='cuda')
        return self._packed_poly

    def packed_wino2(self):
        """tests only ('wino2'): G g G^T of the F(2x2,3x3) kernel in the cross-check library"""
        if self._packed_wino2 is None:
            assert self.KS == 3 and self.stride == 1 and self.Cin % 16 == 0 and self.Cout % 4 == 0
            n = _lib.load_xcheck().l3c_conv_wino_packed_words(self.Cout, self.Cin)
            self._packed_wino2 = torch.empty(n, dtype=torch.float32, device='cuda')
            _lib.call_xcheck('l3c_conv_wino_pack_weights', ptr(self.weight), self.Cout, self.Cin, ptr(self._packed_wino2), stream())
        return self._packed_wino2

    def packed_wino4w(self):
        """tests / probes only ('wino4w'): the F(4x4,3x3) probe kernel on the 32x32x2 MFMA in the cross-check library (csrc/conv_wino4w.hip)"""
        if getattr(self, '_packed_wino4w', None) is None:
            assert self.KS == 3 and self.stride == 1 and self.dilation == 1 and self.Cin % 16 == 0 and self.Cout <= 64
            n = _lib.load_xcheck().l3c_conv_wino4w_packed_words(self.Cout, self.Cin)
            self._packed_wino4w = torch.empty(n, dtype=torch.float32, device='cuda')
            _lib.call_xcheck('l3c_conv_wino4w_pack_weights', ptr(self.weight), self.Cout, self.Cin, ptr(self._packed_wino4w), stream())
        return self._packed_wino4w

    def out_hw(self, H, W):
        pad = self.KS // 2 if self.dilation == 1 else self.dilation
        ext = (self.KS - 1) * self.dilation + 1
        return (H + 2 * pad - ext) // self.stride + 1, (W + 2 * pad - ext) // self.stride + 1


def conv(x, layer, out=None, in_coff=0, out_coff=0, residual=None, res_coff=0, relu=False, pixel_shuffle=False, impl=None):
    """x: (B,H,W,cstride) pixel-major fp32.  Returns `out` ((B,Ho,Wo,Cout) freshly allocated when None).  impl: None = the
    product's dispatch (see above); a name forces one kernel (tests, probes)."""
    B, H, W, cstride = x.shape
    Ho, Wo = layer.out_hw(H, W)
    if out is None:
        out = (torch.empty(B, 2 * Ho, 2 * Wo, layer.Cout // 4, dtype=torch.float32, device=x.device) if pixel_shuffle
               else torch.empty(B, Ho, Wo, layer.Cout, dtype=torch.float32, device=x.device))
    assert impl in (None, 'wino4', 'wino2', 'wino4w', 'gemm', 'poly5', 'poly5x4', 'direct'), impl

    # The Winograd kernel stores / loads 16 bytes per lane: channel strides and offsets of the output (and residual) slices must be
    # multiples of 4, the pointers 16-byte aligned, Cout a multiple of 4 (pixel shuffle: 16); it addresses one input image with
    # 32-bit offsets (< 2 GB) and has no pixel shuffle combined with dilation, ReLU or a residual -- the preconditions
    # l3c_conv_wino4 checks.  Anything else goes to the implicit-GEMM kernel (64-bit addressing, 4-byte stores).
    def _aligned(t, coff):
        return t is None or (t.shape[-1] % 4 == 0 and coff % 4 == 0 and t.data_ptr() % 16 == 0)
    in_ok = x.data_ptr() % 16 == 0 and cstride % 4 == 0 and in_coff % 4 == 0 and H * W * cstride * 4 < 0x7ffffff0
    wino_ok = (layer.KS == 3 and layer.stride == 1 and in_ok and _aligned(out, out_coff) and _aligned(residual, res_coff) and
               layer.Cout % 4 == 0 and (not pixel_shuffle or layer.Cout == 256) and
               not (pixel_shuffle and (relu or residual is not None or layer.dilation != 1)))
    poly_ok = (layer.packed_poly_fused is not None and H % 2 == 0 and W % 2 == 0 and in_ok and _aligned(out, out_coff) and
               residual is None and not relu and not pixel_shuffle)
    if impl in ('poly5', 'poly5x4') or (impl is None and poly_ok):
        assert poly_ok, 'this layer / tensor has no polyphase form'
        return _conv_poly5(x, layer, out, in_coff, out_coff, fused=impl != 'poly5x4')
    if impl is None:
        kernel = ('wino4' if layer.packed_wino4 is not None and wino_ok else
                  'pw' if layer.packed_pw is not None and not (relu or pixel_shuffle or residual is not None) else 'gemm')
    else:
        kernel = impl
    if kernel == 'wino4w':      # probe kernel of the test-only library: 3x3 / stride 1 / dilation 1, bias (+ ReLU)
        assert layer.KS == 3 and layer.stride == 1 and layer.dilation == 1 and residual is None and not pixel_shuffle
        d = ConvDesc()
        d.inp, d.in_cstride, d.in_coff = ptr(x, torch.float32), cstride, in_coff
        d.packed_w, d.bias = ptr(layer.packed_wino4w()), ptr(layer.bias)
        d.out, d.out_cstride, d.out_coff = ptr(out, torch.float32), out.shape[-1], out_coff
        d.B, d.Hin, d.Win, d.Cin, d.Cout = B, H, W, layer.Cin, layer.Cout
        d.KS, d.stride, d.dilation = 3, 1, 1
        d.epilogue = _lib.EPI_RELU if relu else 0
        _lib.call_xcheck('l3c_conv_wino4w', d, int(os.environ.get('L3C_W4W_TPB', '0')), stream())
        return out
    if kernel in ('wino4', 'wino2'):
        assert wino_ok and (kernel == 'wino2' or layer.packed_wino4 is not None), 'this layer / epilogue has no Winograd form'
    if kernel == 'gemm' and layer.packed is None:
        raise _lib.L3CError('convolution outside every MFMA kernel\'s preconditions (Cin % 16 != 0)')
    d = ConvDesc()
    d.inp, d.in_cstride, d.in_coff = ptr(x, torch.float32), cstride, in_coff
    d.packed_w = ptr({'wino4': layer.packed_wino4_shuffle() if (kernel == 'wino4' and pixel_shuffle) else layer.packed_wino4,
                      'pw': layer.packed_pw, 'gemm': layer.packed, 'direct': layer.weight}[kernel]
                     if kernel != 'wino2' else layer.packed_wino2())
    d.bias = ptr(layer.bias)
    d.residual = ptr(residual, torch.float32) if residual is not None else None
    d.res_cstride = residual.shape[-1] if residual is not None else 0
    d.res_coff = res_coff
    d.out, d.out_cstride, d.out_coff = ptr(out, torch.float32), out.shape[-1], out_coff
    d.B, d.Hin, d.Win, d.Cin, d.Cout = B, H, W, layer.Cin, layer.Cout
    d.KS, d.stride, d.dilation = layer.KS, layer.stride, layer.dilation
    d.epilogue = ((_lib.EPI_RELU if relu else 0) | (_lib.EPI_RESIDUAL if residual is not None else 0) |
                  (_lib.EPI_PIXEL_SHUFFLE if pixel_shuffle else 0))
    if kernel == 'wino2':
        _lib.call_xcheck('l3c_conv_wino', d, stream())
        return out
    entry = {'wino4': 'l3c_conv_wino4', 'pw': 'l3c_conv_pw', 'gemm': 'l3c_conv_mfma', 'direct': 'l3c_conv_direct'}[kernel]
    if PROFILE is None or kernel == 'direct':
        call(entry, d, stream())
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(entry, d, stream())
    e1.record()
    key = ('conv_wino4_kernel' if kernel == 'wino4' else 'conv_lds_kernel<3,{}>'.format(layer.dilation) if layer.KS == 3 else
           'conv k{} s{} (mfma)'.format(layer.KS, layer.stride))   # 3x3: the kernel name rocprofv3 reports
    if PROFILE_DETAIL:
        key += ' {}->{} {}x{}{}{}'.format(layer.Cin, layer.Cout, Ho, Wo, ' +res' if residual is not None else '', ' shuffle' if pixel_shuffle else '')
    nbytes = 4.0 * B * (H * W * layer.Cin + Ho * Wo * layer.Cout * (2 if residual is not None else 1))   # in + out (+ residual), once
    # (sixth field: the kernel's template arguments as rocprofv3 prints them -- bench.py joins the PMC table's per-variant bytes with these)
    variant = '{},{},{},false'.format(*('true' if f else 'false' for f in (relu, residual is not None, pixel_shuffle))) if kernel == 'wino4' else ''
    PROFILE.append((key, 2.0 * B * Ho * Wo * layer.Cout * layer.Cin * layer.KS * layer.KS, nbytes, e0, e1, variant))
    return out


def _conv_poly5(x, layer, out, in_coff, out_coff, fused=True):
    """5x5 stride 2 in polyphase form on the F(4x4,3x3) kernel: all four phases in one launch, or (fused=False) four phase launches
    accumulating into `out` (the first carries the bias)."""
    B, H, W, cstride = x.shape
    e0 = e1 = None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if fused:
        d = ConvDesc()
        d.inp, d.in_cstride, d.in_coff = ptr(x, torch.float32), cstride, in_coff
        d.packed_w, d.bias, d.residual = ptr(layer.packed_poly_fused), ptr(layer.bias), None
        d.out, d.out_cstride, d.out_coff = ptr(out, torch.float32), out.shape[-1], out_coff
        d.B, d.Hin, d.Win, d.Cin, d.Cout = B, H, W, layer.Cin, layer.Cout
        d.KS, d.stride, d.dilation, d.epilogue = 5, 2, 1, 0
        call('l3c_conv_wino4_stride2', d, stream())
    for k, (a, b) in enumerate(() if fused else ((0, 0), (0, 1), (1, 0), (1, 1))):
        d = ConvDesc()
        d.inp, d.in_cstride, d.in_coff = ptr(x, torch.float32), cstride, in_coff
        d.packed_w = ptr(layer.packed_poly()[k])
        d.bias = ptr(layer.bias if k == 0 else layer._zero_bias)
        d.residual = ptr(out, torch.float32) if k else None
        d.res_cstride, d.res_coff = (out.shape[-1], out_coff) if k else (0, 0)
        d.out, d.out_cstride, d.out_coff = ptr(out, torch.float32), out.shape[-1], out_coff
        d.B, d.Hin, d.Win, d.Cin, d.Cout = B, H, W, layer.Cin, layer.Cout
        d.KS, d.stride, d.dilation = 3, 2, 1
        d.epilogue = _lib.EPI_RESIDUAL if k else 0
        call('l3c_conv_wino4_phase', d, a, b, stream())
    if PROFILE is not None:
        e1.record()
        nbytes = 4.0 * B * (H * W * layer.Cin + (H // 2) * (W // 2) * layer.Cout)
        PROFILE.append(('conv k5 s2 (4 F(4x4,3x3) phases)', 2.0 * B * (H // 2) * (W // 2) * layer.Cout * layer.Cin * 25, nbytes, e0, e1))
    return out


def rgb_head(img, ms1_w, ms1_b, ms2_w, ms2_b, conv_w, conv_b, want_shifted=False):
    """img (B,3,H,W) planar 0..255 -> (B,H,W,Cf) pixel-major features [, (B,3,H,W) mean-shifted image]."""
    B, _, H, W = img.shape
    Cf = conv_w.shape[0]
    out = torch.empty(B, H, W, Cf, dtype=torch.float32, device=img.device)
    shifted = torch.empty_like(img) if want_shifted else None
    call('l3c_rgb_head', ptr(img, torch.float32), ptr(ms1_w), ptr(ms1_b), ptr(ms2_w), ptr(ms2_b), ptr(conv_w),
         ptr(conv_b), B, H, W, Cf, ptr(out), ptr(shifted), stream())
    return (out, shifted) if want_shifted else out


def to_q_quantize(feat, w, b, levels, want_bn=False):
    """feat (B,H,W,Cf) -> sym int16 (B,C,H,W), bn_q fp32 (B,C,H,W) [, bn pre-quantisation]."""
    B, H, W, Cf = feat.shape
    C, L = w.shape[0], levels.shape[0]
    sym = torch.empty(B, C, H, W, dtype=torch.int16, device=feat.device)
    bn_q = torch.empty(B, C, H, W, dtype=torch.float32, device=feat.device)
    bn = torch.empty_like(bn_q) if want_bn else None
    call('l3c_to_q_quantize', ptr(feat, torch.float32), ptr(w), ptr(b), ptr(levels), B, H * W, Cf, C, L, ptr(sym),
         ptr(bn_q), ptr(bn), stream())
    return (sym, bn_q, bn) if want_bn else (sym, bn_q)


def dec_head(bn_q, w, b, fuse=None):
    """bn_q (B,C,H,W) planar -> (B,H,W,Cf) pixel-major, + fuse."""
    B, C, H, W = bn_q.shape
    Cf = w.shape[0]
    out = torch.empty(B, H, W, Cf, dtype=torch.float32, device=bn_q.device)
    call('l3c_dec_head', ptr(bn_q, torch.float32), ptr(w), ptr(b), ptr(fuse, torch.float32) if fuse is not None else None,
         B, H * W, C, Cf, ptr(out), stream())
    return out


def sym_to_bn(sym, bin_width, x_min):
    bn = torch.empty(sym.shape, dtype=torch.float32, device=sym.device)
    call('l3c_sym_to_bn', ptr(sym, torch.int16), sym.numel(), float(bin_width), float(x_min), ptr(bn), stream())
    return bn


# ---- RGB baselines: bicubic pyramid encoder ------------------------------------------------------------------------------

_RGB_MEAN = None
_COEFFS = {}


def rgb_mean():
    """(0.4488, 0.4371, 0.4040) * 255 in fp32, as BicubicDownsamplingEnc builds it (net.py:69-70)."""
    global _RGB_MEAN
    if _RGB_MEAN is None:
        m = torch.tensor([0.4488, 0.4371, 0.4040], dtype=torch.float32).mul(255.)
        _RGB_MEAN = (ctypes.c_float * 3)(*[float(v) for v in m])
    return _RGB_MEAN


def _resample_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _COEFFS:
        from .helpers import pil_resample
        bounds, kk, ksize = pil_resample.precompute_coeffs(in_size, out_size)
        _COEFFS[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    return _COEFFS[key]


def meanshift_planar(img, w, b):
    B, _, H, W = img.shape
    out = torch.empty_like(img)
    call('l3c_meanshift_planar', ptr(img, torch.float32), ptr(w), ptr(b), B, H * W, ptr(out), stream())
    return out


def bicubic_encoder(x):
    """x (B,3,H,W) fp32 planar, mean-shifted -> (bn (B,3,H/2,W/2) fp32, sym int16): round to uint8, Pillow-exact BICUBIC
    half-size resize (horizontal pass, then vertical), symbols = pixel values, bn = value - mean."""
    B, _, H, W = x.shape
    oh, ow = int(H * 0.5), int(W * 0.5)
    if oh < 1 or ow < 1:
        raise ValueError('image too small for another bicubic scale: {}x{}'.format(H, W))
    u8 = torch.empty(B, 3, H, W, dtype=torch.uint8, device=x.device)
    call('l3c_rgb_to_u8', ptr(x, torch.float32), rgb_mean(), B, H * W, ptr(u8), stream())
    bw, kw, ksw = _resample_tables(W, ow, x.device)
    tmp = torch.empty(B, 3, H, ow, dtype=torch.uint8, device=x.device)
    call('l3c_resample_u8', ptr(u8), B * 3, H, W, 0, ow, ptr(bw), ptr(kw), ksw, ptr(tmp), stream())
    bh, kh, ksh = _resample_tables(H, oh, x.device)
    down = torch.empty(B, 3, oh, ow, dtype=torch.uint8, device=x.device)
    call('l3c_resample_u8', ptr(tmp), B * 3, H, ow, 1, oh, ptr(bh), ptr(kh), ksh, ptr(down), stream())
    sym = torch.empty(B, 3, oh, ow, dtype=torch.int16, device=x.device)
    bn = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=x.device)
    call('l3c_u8_to_sym_bn', ptr(down), rgb_mean(), B, oh * ow, ptr(sym), ptr(bn), stream())
    return bn, sym


# ---- logistic-mixture head --------------------------------------------------------------------------------------------


def as_pixel_major(l):
    """(N,Kp,H,W) logical tensor -> (N,H,W,Kp) contiguous storage (free when `l` already is a permuted NHWC view)."""
    return l.permute(0, 2, 3, 1).contiguous()


def dmll_channel_params(P_nhwc, sym, C, K, rgb, c):
    """-> pi, mu, log_sigma, each (B,K,H,W)."""
    B, H, W, _ = P_nhwc.shape
    outs = [torch.empty(B, K, H, W, dtype=torch.float32, device=P_nhwc.device) for _ in range(3)]
    call('l3c_dmll_channel_params', ptr(P_nhwc, torch.float32), ptr(sym, torch.int16) if sym is not None else None,
         B, H * W, C, K, int(rgb), c, ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), stream())
    return outs


def cdf_table_mixture(targets, pi, mu, log_sigma, check_monotone=True):
    """params (B,K,H,W) -> uint16 table viewed as int16 (B,H,W,Lp); also returns the not-monotone flag tensor."""
    B, K, H, W = pi.shape
    Lp = targets.shape[0]
    cdf = torch.empty(B, H, W, Lp, dtype=torch.int16, device=pi.device)
    flag = torch.zeros(1, dtype=torch.int32, device=pi.device) if check_monotone else None
    call('l3c_cdf_table_mixture', ptr(targets, torch.float32), ptr(pi, torch.float32), ptr(mu, torch.float32),
         ptr(log_sigma, torch.float32), B, H * W, K, Lp, ptr(cdf), ptr(flag), stream())
    return cdf, flag


def dmll_cdf_table(P_nhwc, sym, targets, C, K, rgb, c, pix0, npix, flag=None, window_stats=None):
    """Fused decoder head: uint16 table rows (viewed as int16, (B, npix, Lp)) of channel c for pixels [pix0, pix0 + npix) of
    every image; `flag` (int32[1], device) is set if a row is not strictly increasing (never cleared).
    window_stats (int32 (B,), device; RGB scale only): per image the miss count its decoder reported two chunks earlier -- an image
    whose count allows it gets 65-entry WINDOW rows packed from the start of its (B, npix, Lp) slot, the others full rows
    (include/l3c_hip.h, l3c_ac_decode_part); pass the same tensor to `ac_decode_part(window=...)`."""
    B, H, W, _ = P_nhwc.shape
    Lp = targets.shape[0]
    cdf = torch.empty(B, npix, Lp, dtype=torch.int16, device=P_nhwc.device)
    call('l3c_dmll_cdf_table', ptr(P_nhwc, torch.float32), ptr(sym, torch.int16) if sym is not None else None,
         ptr(targets, torch.float32), B, H * W, C, K, int(rgb), c, pix0, npix, Lp, ptr(cdf),
         ptr(flag, torch.int32) if flag is not None else None,
         ptr(window_stats, torch.int32) if window_stats is not None else None, stream())
    return cdf


def dmll_cdf_table_parts(P_nhwc, sym, targets, C, K, rgb, parts):
    """Grouped form of `dmll_cdf_table` (l3c_dmll_cdf_table_parts): parts = [(c, pix0, npix, flag or None, window_stats or None)], up to 8,
    ONE launch.  -> list of tables (int16 (B, npix, Lp)) in the order of `parts`."""
    B, H, W, _ = P_nhwc.shape
    Lp = targets.shape[0]
    arr = (_lib.TablePart * len(parts))()
    tables = []
    for i, (c, pix0, npix, flag, wstats) in enumerate(parts):
        t = torch.empty(B, npix, Lp, dtype=torch.int16, device=P_nhwc.device)
        tables.append(t)
        arr[i] = _lib.TablePart(c, pix0, npix, ptr(t), ptr(flag, torch.int32) if flag is not None else None,
                                ptr(wstats, torch.int32) if wstats is not None else None)
    call('l3c_dmll_cdf_table_parts', ptr(P_nhwc, torch.float32), ptr(sym, torch.int16) if sym is not None else None,
         ptr(targets, torch.float32), B, H * W, C, K, int(rgb), Lp, arr, len(parts), stream())
    return tables


def decode_rgb(P_nhwc, targets, sym, buf, offs, lens, bounds, K, lag, window_mode, side_stream=None):
    """The whole RGB scale in one host call (l3c_decode_rgb): P (B,H,W,120), sym int16 (B,3,H,W) ZEROED (receives the symbols), the streams
    CHANNEL-major in (buf, offs int64 (3B,), lens int32 (3B,)), bounds = [(pix0, npix)] tiling H*W.  lag 2 decodes on `side_stream` (a
    torch stream) while the current stream builds the next step's tables.  -> (workspace tensor, window statistics view (3, chunks + 2, B)
    or None); the current stream is ordered after the last symbols."""
    import ctypes
    lib = _lib.load()
    B, H, W, _ = P_nhwc.shape
    n = len(bounds)
    max_npix = max(b[1] for b in bounds)
    nbytes = lib.l3c_decode_rgb_workspace_bytes(B, max_npix, n, lag)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=P_nhwc.device)
    p0 = (_lib.c_i64 * n)(*[b[0] for b in bounds])
    np_ = (_lib.c_i64 * n)(*[b[1] for b in bounds])
    desc = _lib.RgbDecodeDesc(ptr(P_nhwc, torch.float32), ptr(targets, torch.float32), ptr(sym, torch.int16), B, H * W, K,
                              ptr(buf, torch.uint8), ptr(offs, torch.int64), ptr(lens, torch.int32), n,
                              ctypes.cast(p0, ctypes.POINTER(_lib.c_i64)), ctypes.cast(np_, ctypes.POINTER(_lib.c_i64)),
                              lag, window_mode, ptr(ws), nbytes)
    main = torch.cuda.current_stream()
    if lag == 2:
        for t in (P_nhwc, targets, sym, buf, offs, lens, ws):
            t.record_stream(side_stream)
    call('l3c_decode_rgb', ctypes.byref(desc), main.cuda_stream, side_stream.cuda_stream if lag == 2 else None)
    stats = None
    if window_mode:
        o = lib.l3c_decode_rgb_stats_offset(B, max_npix, n, lag)
        stats = ws[o:o + 3 * (n + 2) * B * 4].view(torch.int32).view(3, n + 2, B)
    return ws, stats


_PINNED_SCRATCH = {'bufs': [None] * 16, 'events': [None] * 16, 'turn': 0}


def upload_small(array):
    """numpy array -> device tensor of the same dtype through a small ring of page-locked buffers (an H2D copy from pageable memory makes the
    host wait for the stream): descriptor tables of a few KB."""
    import numpy as np
    a = np.ascontiguousarray(array)
    r = _PINNED_SCRATCH
    k = r['turn'] = (r['turn'] + 1) % len(r['bufs'])
    if r['events'][k] is not None:
        r['events'][k].synchronize()
    n = a.nbytes
    if r['bufs'][k] is None or r['bufs'][k].numel() < n:
        r['bufs'][k] = torch.empty(max(n, 1 << 18), dtype=torch.uint8, pin_memory=True)
    r['bufs'][k].numpy()[:n] = a.view(np.uint8).reshape(-1)
    dev = r['bufs'][k][:n].cuda(non_blocking=True)
    r['events'][k] = torch.cuda.Event()
    r['events'][k].record(torch.cuda.current_stream())
    return dev.view(torch.from_numpy(a[:0]).dtype).reshape(a.shape)


def ragged_rgb_plan(hws, n_regular, probe):
    """Chunk plan of a RAGGED RGB decode (l3c_decode_rgb_ragged): every image gets the same NUMBER of chunks -- two probe chunks of `probe`
    symbols (0: none) and n_regular regular ones, the image's own chunk length rounded DOWN to a multiple of 64 with the last chunk taking
    the rest -- so that all images step through the pipeline together.  hws: pixels per image.  -> (pix0, npix) int64 arrays (n_chunks, B)."""
    import numpy as np
    B = len(hws)
    n_chunks = n_regular + (2 if probe else 0)
    pix0 = np.zeros((n_chunks, B), dtype=np.int64)
    npix = np.zeros((n_chunks, B), dtype=np.int64)
    for b, hw in enumerate(hws):
        start, j = 0, 0
        if probe:
            pix0[0, b], npix[0, b], pix0[1, b], npix[1, b] = 0, probe, probe, probe
            start, j = 2 * probe, 2
        step = (hw - start) // n_regular // 64 * 64
        assert n_regular == 1 or step >= 64, 'image too small for {} chunks'.format(n_regular)
        for k in range(n_regular):
            pix0[j + k, b] = start + k * step
            npix[j + k, b] = step if k + 1 < n_regular else hw - start - k * step
    return pix0, npix


def decode_rgb_ragged(P_ragged, targets, sym_ragged, buf, offs, lens, hws, pix0, npix, K, lag, window_mode, side_stream=None):
    """The RGB scale of B images of DIFFERENT sizes in lock step (l3c_decode_rgb_ragged): P_ragged (sum HW, 120) fp32, sym_ragged int16
    (3 * sum HW,) ZEROED (image b: its three planes from element 3 * pixbase[b]), streams CHANNEL-major in (buf, offs (3B,), lens (3B,)),
    hws = pixels per image, (pix0, npix) = ragged_rgb_plan(...).  -> workspace tensor (kept alive by the caller until the stream is done)."""
    import ctypes
    import numpy as np
    lib = _lib.load()
    B, n = len(hws), pix0.shape[0]
    hw = np.asarray(hws, dtype=np.int64)
    pixbase = np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)
    table_off = (np.cumsum(npix, axis=1) - npix) * (257 * 2)
    tables = upload_small(np.concatenate([pixbase, hw, pix0.reshape(-1), npix.reshape(-1), table_off.reshape(-1)]).astype(np.int64))
    max_total = int(npix.sum(axis=1).max())
    nbytes = lib.l3c_decode_rgb_ragged_workspace_bytes(B, max_total, n, lag)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=P_ragged.device)
    hw_c = (_lib.c_i64 * B)(*[int(v) for v in hw])
    p0_c = (_lib.c_i64 * (n * B))(*[int(v) for v in pix0.reshape(-1)])
    np_c = (_lib.c_i64 * (n * B))(*[int(v) for v in npix.reshape(-1)])
    desc = _lib.RgbRaggedDesc(ptr(P_ragged, torch.float32), ptr(targets, torch.float32), ptr(sym_ragged, torch.int16), B,
                              ctypes.cast(hw_c, ctypes.POINTER(_lib.c_i64)), K, ptr(buf, torch.uint8), ptr(offs, torch.int64),
                              ptr(lens, torch.int32), n, ctypes.cast(p0_c, ctypes.POINTER(_lib.c_i64)),
                              ctypes.cast(np_c, ctypes.POINTER(_lib.c_i64)), ptr(tables, torch.int64), lag, window_mode, ptr(ws), nbytes)
    main = torch.cuda.current_stream()
    if lag == 2:
        for t in (P_ragged, targets, sym_ragged, buf, offs, lens, ws, tables):
            t.record_stream(side_stream)
    call('l3c_decode_rgb_ragged', ctypes.byref(desc), main.cuda_stream, side_stream.cuda_stream if lag == 2 else None)
    return ws, tables


def decode_z_ragged(P_ragged, targets, sym_ragged, buf, offs, lens, hws, C, K):
    """A bottleneck scale of B images of DIFFERENT sizes: its C channels are independent given P, so ONE ragged table launch
    (l3c_dmll_cdf_table_ragged, C parts) and ONE ragged decoder launch (l3c_ac_decode_chunks with the r_* fields) decode every image and
    channel side by side.  P_ragged (sum HW, 3 C K) fp32; sym_ragged int16 (C * sum HW,): image b's C planes from element C * pixbase[b];
    streams CHANNEL-major in (buf, offs (C B,), lens (C B,)).  -> tensors to keep alive until the stream is done."""
    import ctypes
    import numpy as np
    B = len(hws)
    assert C <= 8
    hw = np.asarray(hws, dtype=np.int64)
    pixbase = np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)
    Lp = targets.shape[0]
    table_off = pixbase * (Lp * 2)                    # every image's whole plane is one range: npix = hw, pix0 = 0
    tables = upload_small(np.concatenate([pixbase, hw, np.zeros(B, dtype=np.int64), hw, table_off]).astype(np.int64))
    base = tables.data_ptr()
    total = int(hw.sum())
    flag = torch.zeros(1, dtype=torch.int32, device=P_ragged.device)
    tabs = [torch.empty(total * Lp, dtype=torch.int16, device=P_ragged.device) for _ in range(C)]
    batch = _lib.RaggedBatch(B, int(hw.max()), base, base + 8 * B)
    tparts = (_lib.TablePart * C)()
    rparts = (_lib.RaggedPart * C)()
    dparts = (_lib.AcDecodePart * C)()
    for c in range(C):
        tparts[c] = _lib.TablePart(c, 0, int(hw.max()), ptr(tabs[c]), ptr(flag, torch.int32), None)
        rparts[c] = _lib.RaggedPart(base + 16 * B, base + 24 * B, base + 32 * B)
        d = _lib.AcDecodePart(ptr(tabs[c]), Lp, ptr(buf, torch.uint8), ptr(offs[c * B:(c + 1) * B], torch.int64), ptr(lens[c * B:(c + 1) * B], torch.int32),
                              B, int(hw.max()), ptr(flag, torch.int32), None, None, 1, ptr(sym_ragged, torch.int16), 0, 0)
        d.r_npix, d.r_table_off, d.r_pixbase, d.r_hw, d.r_pix0 = base + 24 * B, base + 32 * B, base, base + 8 * B, base + 16 * B
        d.r_C, d.r_c, d.r_table_bytes = C, c, total * Lp * 2
        dparts[c] = d
    call('l3c_dmll_cdf_table_ragged', ptr(P_ragged, torch.float32), None, ptr(targets, torch.float32), ctypes.byref(batch), C, K, 0, Lp,
         tparts, rparts, C, stream())
    call('l3c_ac_decode_chunks', dparts, C, stream())
    return tabs, tables, flag


def container_read(files_dev, src_off, dst_off, nbytes, max_nbytes, dst):
    """l3c_container_read: the streams of many raw `.l3c` files (one device buffer) -> 4-byte aligned, zero padded streams in `dst`."""
    call('l3c_container_read', ptr(files_dev, torch.uint8), ptr(src_off, torch.int64), ptr(dst_off, torch.int64), ptr(nbytes, torch.int32),
         nbytes.numel(), int(max_nbytes), ptr(dst, torch.uint8), stream())


def dmll_encode_intervals(P_nhwc, sym, targets, C, K, rgb):
    """P (B,H,W,Kp), sym int16 (B,C,H,W) -> packed interval words for the B*C streams of this scale."""
    B, H, W, _ = P_nhwc.shape
    Lp = targets.shape[0]
    n = _lib.load().l3c_interval_words(B * C, H * W)
    iv = torch.empty(n, dtype=torch.int32, device=P_nhwc.device)
    call('l3c_dmll_encode_intervals', ptr(P_nhwc, torch.float32), ptr(sym, torch.int16), ptr(targets, torch.float32),
         B, H * W, C, K, int(rgb), Lp, ptr(iv), stream())
    return iv


def dmll_nll(P_nhwc, x, C, K, rgb, x_min, x_max, L):
    """x (B,C,H,W) fp32 targets -> nll (B,C,H,W) in nats."""
    B, H, W, _ = P_nhwc.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=P_nhwc.device)
    call('l3c_dmll_nll', ptr(P_nhwc, torch.float32), ptr(x, torch.float32), B, H * W, C, K, int(rgb), float(x_min),
         float(x_max), L, ptr(out), stream())
    return out


def dmll_sample(P_nhwc, u_mix, u_log, C, K, rgb):
    """P (B,H,W,Kp); u_mix (B,C,K,H,W), u_log (B,C,H,W) uniforms in (0,1) -> x fp32 (B,C,H,W), not rounded."""
    B, H, W, _ = P_nhwc.shape
    assert tuple(u_mix.shape) == (B, C, K, H, W) and tuple(u_log.shape) == (B, C, H, W), (u_mix.shape, u_log.shape)
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=P_nhwc.device)
    call('l3c_dmll_sample', ptr(P_nhwc, torch.float32), ptr(u_mix, torch.float32), ptr(u_log, torch.float32), B, H * W, C, K,
         int(rgb), ptr(out), stream())
    return out


# ---- arithmetic coder -------------------------------------------------------------------------------------------------


def intervals_from_table(cdf, sym, n_streams, n_sym, broadcast_row=False):
    """cdf int16/uint16: (n_streams*n_sym, Lp) rows, or one row (Lp,) with broadcast_row; sym int16 (n_streams, n_sym)."""
    Lp = cdf.shape[-1]
    n = _lib.load().l3c_interval_words(n_streams, n_sym)
    iv = torch.empty(n, dtype=torch.int32, device=sym.device)
    call('l3c_ac_intervals_from_table', ptr(cdf), 0 if broadcast_row else Lp, Lp, ptr(sym, torch.int16), n_streams,
         n_sym, ptr(iv), stream())
    return iv


def ac_encode(iv, n_streams, n_sym):
    """-> (out uint8 (n_streams, stride), nbytes int32 (n_streams,)) on the device.  `iv` is clobbered."""
    stride = _lib.load().l3c_ac_max_bytes(n_sym)
    out = torch.empty(n_streams, stride, dtype=torch.uint8, device=iv.device)
    nbytes = torch.empty(n_streams, dtype=torch.int32, device=iv.device)
    ws = torch.empty(_lib.load().l3c_ac_encode_workspace_bytes(n_streams), dtype=torch.uint8, device=iv.device)
    call('l3c_ac_encode', ptr(iv), n_streams, n_sym, ptr(out), stride, ptr(nbytes), ptr(ws), stream())
    return out, nbytes


def ac_encode_groups(groups):
    """groups: list of (iv, n_streams, n_sym).  ONE state launch + ONE pack launch code every stream of every group
    concurrently.  -> list of (out uint8 (n_streams, stride), nbytes int32 (n_streams,)); the iv buffers are clobbered."""
    lib = _lib.load()
    arr = (_lib.AcGroup * len(groups))()
    res = []
    total = 0
    for g, (iv, n_streams, n_sym) in enumerate(groups):
        stride = lib.l3c_ac_max_bytes(n_sym)
        out = torch.empty(n_streams, stride, dtype=torch.uint8, device=iv.device)
        nbytes = torch.empty(n_streams, dtype=torch.int32, device=iv.device)
        arr[g] = _lib.AcGroup(ptr(iv), ptr(out), ptr(nbytes), n_streams, n_sym, stride)
        res.append((out, nbytes))
        total += n_streams
    ws = torch.empty(lib.l3c_ac_encode_groups_workspace_bytes(len(groups), total), dtype=torch.uint8, device=groups[0][0].device)
    call('l3c_ac_encode_groups', arr, len(groups), ptr(ws), stream())
    return res, ws


def pack_streams(payloads, device='cuda'):
    """list of bytes -> (uint8 buffer with every stream 4-byte aligned and zero padded, offsets int64, nbytes int32)."""
    import numpy as np
    offs, pos = [], 0
    for p in payloads:
        offs.append(pos)
        pos += (len(p) + 3) // 4 * 4 + 4
    buf = np.zeros(max(pos, 4), dtype=np.uint8)
    for o, p in zip(offs, payloads):
        buf[o:o + len(p)] = np.frombuffer(p, dtype=np.uint8)
    return (torch.from_numpy(buf).to(device), torch.tensor(offs, dtype=torch.int64, device=device),
            torch.tensor([len(p) for p in payloads], dtype=torch.int32, device=device))


def ac_decode(cdf, payload_buf, offsets, nbytes, n_streams, n_sym, monotone, broadcast_row=False):
    """-> sym int16 (n_streams, n_sym) on the device."""
    Lp = cdf.shape[-1]
    sym = torch.empty(n_streams, n_sym, dtype=torch.int16, device=payload_buf.device)
    call('l3c_ac_decode', ptr(cdf), 0 if broadcast_row else Lp, Lp, ptr(payload_buf, torch.uint8),
         ptr(offsets, torch.int64), ptr(nbytes, torch.int32), n_streams, n_sym, int(bool(monotone)), ptr(sym), stream())
    return sym


def ac_decode_state(n_streams, device='cuda'):
    """Opaque per-stream coder state carried between the chunks of l3c_ac_decode_chunk."""
    return torch.empty(n_streams * _lib.load().l3c_ac_decode_state_bytes(), dtype=torch.uint8, device=device)


def ac_decode_part(cdf, payload_buf, offsets, nbytes, n_streams, n_sym, flag, state_in, state_out, final, sym_out,
                   sym_stride, sym_offset, window=None):
    """One part of `ac_decode_chunks`: decode symbols [sym_offset, sym_offset + n_sym) of every stream from the table rows of
    that range (cdf: (n_streams * n_sym, Lp)), resuming from `state_in` (None: start of the streams) and saving into
    `state_out`; writes into `sym_out` (int16, row stride `sym_stride`).  `flag`: device int32 'table not validated'
    (None: treat as not validated).  The tuple keeps the tensors alive until the launch.
    window: None, or (stats_in, stats_out, P_nhwc, sym_all, targets, pix0, C, K, c) for a table built with `dmll_cdf_table(...,
    window_stats=stats_in)`: streams whose rows are window rows are decoded from those, a symbol outside its window from the pixel's
    full row evaluated by the decoder itself (needs P, the decoded channels `sym_all` (B, C, H, W) and the bin edges); stats_out
    (int32 (B,)) receives every stream's miss count of this chunk."""
    part = _lib.AcDecodePart(ptr(cdf), cdf.shape[-1], ptr(payload_buf, torch.uint8), ptr(offsets, torch.int64),
                             ptr(nbytes, torch.int32), n_streams, n_sym,
                             ptr(flag, torch.int32) if flag is not None else None,
                             ptr(state_in) if state_in is not None else None,
                             ptr(state_out) if state_out is not None else None, int(bool(final)),
                             ptr(sym_out, torch.int16), sym_stride, sym_offset)
    keep = (cdf, payload_buf, offsets, nbytes, flag, state_in, state_out, sym_out)
    if window is not None:
        stats_in, stats_out, P_nhwc, sym_all, targets, pix0, C, K, c = window
        part.window_stats_in, part.window_stats_out = ptr(stats_in, torch.int32), ptr(stats_out, torch.int32)
        part.P, part.sym_all, part.targets = ptr(P_nhwc, torch.float32), ptr(sym_all, torch.int16), ptr(targets, torch.float32)
        part.HW, part.pix0 = P_nhwc.shape[1] * P_nhwc.shape[2], pix0
        part.C, part.K, part.c = C, K, c
        keep += (stats_in, stats_out, P_nhwc, sym_all, targets)
    return part, keep


def ac_decode_chunks(parts):
    """1..8 independent parts (see ac_decode_part) decoded side by side in one launch pair."""
    arr = (_lib.AcDecodePart * len(parts))(*[p for p, _ in parts])
    call('l3c_ac_decode_chunks', arr, len(parts), stream())


def table_is_monotone(cdf):
    """Host-synchronising check (used by the generic torchac.decode_cdf path on user tables)."""
    Lp = cdf.shape[-1]
    flag = torch.zeros(1, dtype=torch.int32, device=cdf.device)
    call('l3c_cdf_check_monotone', ptr(cdf), cdf.numel() // Lp, Lp, ptr(flag), stream())
    return int(flag.item()) == 0
