"""MultiscaleNetwork on hand-written HIP kernels -- same surface as the reference's
modules/multiscale_network.py (`forward` :226-246, `get_P` :308-322, `Out` :54-130, `Losses` :133-165).

The module owns an nn.Parameter tree whose state-dict keys are exactly the reference's (modules/schema.py), so released
checkpoints load with `load_state_dict(strict=True)`; there are no nn.Conv2d children -- the compute is the conv stack of
csrc/ (MFMA implicit GEMM + the small fused kernels), driven layer by layer in the order of SURVEY.md Appendix A.

Inference only (the reference's eval mode): decoders are fed the quantised bottleneck (`bn_q`), `Out.bn[i]` is `bn_q`.
Internally every wide activation is pixel-major (NHWC); tensors handed out through the reference API keep the reference's
logical NCHW shape (P as a zero-copy permuted view, symbols / bottlenecks planar).
Both model families of the reference are covered: L3C (EDSRLikeEnc/EDSRDec, configs/ms/cr.cf) and the RGB / RGB Shared
baselines (BicubicSubsampling, cr_rgb.cf / cr_rgb_shared.cf, forward + loss like the reference: it has no file coding for them).
"""
import itertools

import numpy as np
import torch
from torch import nn

from .. import ops
from .. import _lib
from ..criterion.logistic_mixture import DiscretizedMixLogisticLoss
from . import schema
from .net import EncOut


class Out(object):
    """Outputs of the network, fine to coarse (reference multiscale_network.py:54-130).

    S[0] = image symbols, S[i] = symbols of z^(i) (NCHW long); L[i] = number of levels; bn[0] = None, bn[i] = z^(i);
    P[i] = parameters of p(S[i] | f^(i+1)) as logical NCHW.  `raw` keeps the device-native buffers for the coder:
    raw.sym[i] int16 planar, raw.P[i] pixel-major.
    """

    def __init__(self, targets_style='S', auto_recursive_from=None):
        assert targets_style in ('S', 'bn'), targets_style
        self.S, self.L, self.bn, self.P = [], [], [], []
        self.auto_recursive_from = auto_recursive_from
        self.targets_style = targets_style
        self.raw = _Raw()

    def append(self, enc_out, P, is_training=False):
        self.S.append(enc_out.S)
        self.L.append(enc_out.L)
        self.P.append(P)
        self.bn.append(enc_out.bn if is_training else enc_out.bn_q)
        assert len(self.S) == len(self.L) == len(self.bn) == len(self.P) + 1

    def append_input_image(self, x):
        self.S.append(x.round().long())
        self.L.append(256)
        self.bn.append(None)

    def get_uniform_P(self):
        N, C, H, W = self.S[-1].shape
        return torch.ones(N, self.L[-1], C, H, W, dtype=torch.float32, device=self.S[-1].device)

    def iter_all_scales(self):
        return zip(self.S[1:], self.bn[1:], self.P[1:] + [self.get_uniform_P()], self.L[1:])

    def iter_targets_and_predictions(self, loss_rgb, loss_others):
        yield (loss_rgb, self.S[0].float(), self.P[0])
        other_targets = (S.float() for S in self.S[1:]) if self.targets_style == 'S' else self.bn[1:]
        for t in zip(itertools.repeat(loss_others), other_targets, self.P[1:]):
            yield t

    def get_nat_count(self, i):
        N, C, H, W = self.S[i].shape
        return N * C * H * W * np.log(self.L[i])


class _Raw(object):
    def __init__(self):
        self.sym = []      # int16 (B,C,H,W) per scale, fine -> coarse (index 0 = image)
        self.P = []        # fp32 (B,H,W,Kp) per predicted scale
        self.bn_q = []     # fp32 (B,C,H,W); index 0 = None
        self.F_enc = []    # fp32 (B,H,W,Cf) encoder features (pre-quantiser), per scale
        self.F_dec = []    # fp32 (B,H,W,Cf) decoder features, per scale


class Losses(nn.Module):
    """reference multiscale_network.py:133-165."""

    def __init__(self, config_ms):
        super(Losses, self).__init__()
        self.loss_dmol_rgb = DiscretizedMixLogisticLoss(rgb_scale=True, x_min=0, x_max=255, L=256)
        if config_ms.rgb_bicubic_baseline:
            self.loss_dmol_n = self.loss_dmol_rgb
        else:
            x_min, x_max = config_ms.q.levels_range
            self.loss_dmol_n = DiscretizedMixLogisticLoss(rgb_scale=False, x_min=x_min, x_max=x_max, L=config_ms.q.L)

    def get(self, out):
        costs = [loss(target_i, P_i, scale).sum()
                 for scale, (loss, target_i, P_i) in enumerate(out.iter_targets_and_predictions(
                     loss_rgb=self.loss_dmol_rgb, loss_others=self.loss_dmol_n))]
        final_non_recursive_idx = -1 if out.auto_recursive_from is None else out.auto_recursive_from
        final_cost_uniform = out.get_nat_count(final_non_recursive_idx)
        num_subpixels = int(np.prod(out.S[0].shape))
        return costs, final_cost_uniform, num_subpixels


class _ParamNode(nn.Module):
    """Container that only exists to give parameters the reference's dotted names."""

    def forward(self, *a, **kw):
        raise NotImplementedError('parameter container')


def _build_param_tree(root, shapes):
    for key, shape in shapes.items():
        parts = key.split('.')
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _ParamNode())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


class MultiscaleNetwork(nn.Module):
    def __init__(self, config_ms):
        super(MultiscaleNetwork, self).__init__()
        self._rgb = bool(config_ms.rgb_bicubic_baseline)
        if config_ms.dec.cls != 'EDSRDec':
            raise NotImplementedError('decoder class {}'.format(config_ms.dec.cls))
        if self._rgb:
            # RGB / RGB Shared baselines: bicubic pyramid instead of learned encoders, every scale predicts RGB symbols
            if config_ms.enc.cls != 'BicubicSubsampling' or config_ms.enc.feed_F or config_ms.q.C != 3:
                raise NotImplementedError('RGB baselines need enc.cls=BicubicSubsampling, enc.feed_F=False, q.C=3')
        elif config_ms.enc.cls != 'EDSRLikeEnc' or not (config_ms.enc.feed_F and config_ms.dec.skip):
            raise NotImplementedError('L3C needs enc.cls=EDSRLikeEnc, enc.feed_F and dec.skip (configs/ms/cr.cf)')
        if config_ms.Cf % 32 != 0 or config_ms.kernel_size != 3:
            raise NotImplementedError('Cf must be a multiple of 32 and kernel_size 3')
        self.config_ms = config_ms
        self.scales = config_ms.num_scales
        self._fuse_feat = bool(config_ms.dec.skip)
        self._schema = schema.param_schema(config_ms)
        _build_param_tree(self, self._schema)
        self._packed = None
        self.eval()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    # -- parameters -----------------------------------------------------------------------------------------------------

    def _invalidate(self):
        self._packed = None

    def _apply(self, fn, *a, **kw):   # .to() / .cuda() move the parameters: re-pack lazily
        self._invalidate()
        return super(MultiscaleNetwork, self)._apply(fn, *a, **kw)

    def get_losses(self):
        return Losses(self.config_ms)

    def _prepare(self):
        """Upload + pre-pack every layer once (weights are read-only at inference)."""
        if self._packed is not None:
            return self._packed
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        cfg = self.config_ms
        lv = True if self._rgb else schema_levels_check(sd, cfg)
        dev = lambda t: t.to('cuda', torch.float32).contiguous()   # noqa: E731
        pk = {'levels_ok': lv}

        def pc(key, stride=1, dilation=1):
            return ops.PackedConv(sd[key + '.weight'], sd[key + '.bias'], stride=stride, dilation=dilation)

        def body(prefix, n):
            blocks = [(pc('{}.{}.body.0'.format(prefix, i)), pc('{}.{}.body.2'.format(prefix, i))) for i in range(n)]
            return blocks, pc('{}.{}'.format(prefix, n))

        pk['ms1'] = (dev(sd['sub_rgb_mean.weight'].reshape(3, 3)), dev(sd['sub_rgb_mean.bias']))
        if not self._rgb:
            pk['ms2'] = (dev(sd['heads.0.head.0.weight'].reshape(3, 3)), dev(sd['heads.0.head.0.bias']))
            pk['head0'] = (dev(sd['heads.0.head.1.head.weight']), dev(sd['heads.0.head.1.head.bias']))
            pk['heads'] = [None] + [pc('heads.{}.head'.format(s)) for s in range(1, self.scales)]
        pk['enc'], pk['dec'], pk['prob'] = [], [], []
        for s in range(self.scales):
            e, d = 'nets.{}.enc'.format(s), 'nets.{}.dec'.format(s)
            if not self._rgb:
                eb, et = body(e + '.body', cfg.enc.num_blocks)
                pk['enc'].append({'down': pc(e + '.down', stride=2), 'blocks': eb, 'tail': et,
                                  'to_q': (dev(sd[e + '.to_q.0.weight'].reshape(cfg.q.C, cfg.Cf)), dev(sd[e + '.to_q.0.bias'])),
                                  'levels': dev(sd[e + '.levels'])})
            db, dt = body(d + '.body', cfg.dec.num_blocks)
            pk['dec'].append({'head': (dev(sd[d + '.head.weight'].reshape(cfg.Cf, cfg.q.C)), dev(sd[d + '.head.bias'])),
                              'blocks': db, 'tail': dt, 'up': pc(d + '.tail.0')})
            p = 'prob_clfs.{}.atrous'.format(s)
            pk['prob'].append({'atrous': [pc('{}.atrous.{}'.format(p, i), dilation=r) for i, r in enumerate((1, 2, 4))],
                               'lin': pc(p + '.lin')})
        self._packed = pk
        return pk

    # -- building blocks ------------------------------------------------------------------------------------------------

    @staticmethod
    def _body(x, blocks, tail):
        """8 x ResBlock(conv-ReLU-conv, += input) + conv, + global skip (edsr.py:83-86, net.py:142 / :181)."""
        skip = x
        for c1, c2 in blocks:
            t = ops.conv(x, c1, relu=True)
            x = ops.conv(t, c2, residual=x)
        return ops.conv(x, tail, residual=skip)

    def _encoder(self, inp, s, pk, want_bn=False):
        e = pk['enc'][s]
        x = ops.conv(inp, e['down'])
        F = self._body(x, e['blocks'], e['tail'])
        q = ops.to_q_quantize(F, e['to_q'][0], e['to_q'][1], e['levels'], want_bn=want_bn)
        return (F,) + tuple(q)

    def _decoder(self, bn_q, fuse, s, pk):
        d = pk['dec'][s]
        x = ops.dec_head(bn_q, d['head'][0], d['head'][1], fuse)
        x = self._body(x, d['blocks'], d['tail'])
        return ops.conv(x, d['up'], pixel_shuffle=True)

    def _prob(self, F, s, pk):
        pr = pk['prob'][s]
        B, H, W, Cf = F.shape
        cat = torch.empty(B, H, W, 3 * Cf, dtype=torch.float32, device=F.device)
        for i, a in enumerate(pr['atrous']):
            ops.conv(F, a, out=cat, out_coff=i * Cf)
        return ops.conv(cat, pr['lin'])

    # -- reference API --------------------------------------------------------------------------------------------------

    def forward(self, x, auto_recurse=0):
        """x: image NCHW in [0, 255] (float) -> Out.  auto_recurse: how many times the coarsest scale is applied again (the RGB
        Shared baseline is evaluated with auto_recurse=3, multiscale_tester.py:50, :123-132)."""
        _lib.require_gpu()
        x = self._as_device_image(x)
        pk = self._prepare()
        if self._rgb:
            return self._forward_rgb(x, auto_recurse, pk)
        if auto_recurse:
            raise NotImplementedError('auto_recurse is only used by the RGB Shared baseline')
        out = Out(targets_style='bn', auto_recursive_from=None)
        out.append_input_image(x)
        raw = out.raw
        raw.sym.append(out.S[0].to(torch.int16))
        raw.bn_q.append(None)

        inp = ops.rgb_head(x, pk['ms1'][0], pk['ms1'][1], pk['ms2'][0], pk['ms2'][1], pk['head0'][0], pk['head0'][1])
        enc = []
        for s in range(self.scales):
            if s:
                inp = ops.conv(enc[-1][0], pk['heads'][s])
            enc.append(self._encoder(inp, s, pk))
        dec = [None] * self.scales
        for s in reversed(range(self.scales)):
            fuse = None if s == self.scales - 1 else dec[s + 1]
            dec[s] = self._decoder(enc[s][2], fuse, s, pk)
        for s in range(self.scales):
            F, sym, bn_q = enc[s]
            P = self._prob(dec[s], s, pk)
            raw.sym.append(sym)
            raw.bn_q.append(bn_q)
            raw.P.append(P)
            raw.F_enc.append(F)
            raw.F_dec.append(dec[s])
            out.append(EncOut(bn_q, bn_q, sym.long(), self.config_ms.q.L, F.permute(0, 3, 1, 2)), P.permute(0, 3, 1, 2))
        return out

    def _forward_rgb(self, x, auto_recurse, pk):
        """RGB baselines (reference :226-306 with rgb_bicubic_baseline): identity heads, bicubic pyramid encoders
        (Pillow-exact, on the device), the scale's decoder + classifier predict the RGB symbols of the finer scale."""
        forward_scales = list(range(self.scales)) + [-1] * auto_recurse
        out = Out(targets_style='S', auto_recursive_from=self.scales if auto_recurse > 0 else None)
        out.append_input_image(x)
        raw = out.raw
        raw.sym.append(out.S[0].to(torch.int16))
        raw.bn_q.append(None)
        inp = ops.meanshift_planar(x, pk['ms1'][0], pk['ms1'][1])
        encs = []
        for _ in forward_scales:
            bn, sym = ops.bicubic_encoder(inp)
            encs.append((bn, sym))
            inp = bn                                     # enc.feed_F is False: the next scale sees the subsampled image
        decs = []
        for i, s in reversed(list(enumerate(forward_scales))):
            fuse = None if (not self._fuse_feat or s == -1 or s == max(forward_scales)) else decs[0]
            decs.insert(0, self._decoder(encs[i][0], fuse, s if s >= 0 else self.scales - 1, pk))
        for i, s in enumerate(forward_scales):
            bn, sym = encs[i]
            P = self._prob(decs[i], s if s >= 0 else self.scales - 1, pk)
            raw.sym.append(sym)
            raw.bn_q.append(bn)
            raw.P.append(P)
            raw.F_dec.append(decs[i])
            out.append(EncOut(bn, bn, sym.long(), 256, None), P.permute(0, 3, 1, 2))
        return out

    def get_P(self, scale, bn_q, dec_F_prev=None, n_scales_total=None):
        """Decoder-side step (reference :308-322): (P_scale, F_scale) from the quantised bottleneck of scale+1 and the
        coarser decoder's features.  Tensors use the reference's logical NCHW shapes (F / P are permuted views).
        n_scales_total (RGB baselines with auto_recurse only): number of predicted scales of the forward pass this step belongs
        to -- scales >= num_scales are the recursive applications of the last network, which take no fused features."""
        n_total = self.config_ms.num_scales if n_scales_total is None else n_scales_total
        assert 0 <= scale < n_total, 'Out of range: {}'.format(scale)
        assert n_total == self.config_ms.num_scales or self._rgb, 'recursion is only defined for the RGB baselines'
        _lib.require_gpu()
        pk = self._prepare()
        bn_q = bn_q.to('cuda', torch.float32).contiguous()
        net = min(scale, self.scales - 1)
        if self._rgb:      # same rule as _forward_rgb: no fusion for recursive scales, nor for the coarsest proper one
            forward_scales = list(range(self.scales)) + [-1] * (n_total - self.scales)
            s = forward_scales[scale]
            if not self._fuse_feat or s == -1 or s == max(forward_scales):
                dec_F_prev = None
        fuse = None
        if dec_F_prev is not None:
            fuse = dec_F_prev.permute(0, 2, 3, 1).contiguous()
        F = self._decoder(bn_q, fuse, net, pk)
        P = self._prob(F, net, pk)
        return P.permute(0, 3, 1, 2), F.permute(0, 3, 1, 2)

    @staticmethod
    def _as_device_image(x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('Expected BCHW image, got {}'.format(tuple(x.shape)))
        return x.to('cuda', torch.float32).contiguous()

    # -- sampling (reference :328-406) ------------------------------------------------------------------------------------

    def sample_forward(self, x, losses, sample_scales, partial_final=None, auto_recurse=0, draw=None):
        """x: image NCHW in [0, 255] -> sampled image (N,3,H,W) fp32 in [0, 255], not rounded.
        The image is encoded; the decoders then run coarse -> fine and are fed the encoder's z^(s+1) or -- for the scales in
        `sample_scales` -- values SAMPLED from the coarser scale's predicted mixture (continuous, as in the reference);
        the finest prediction is always sampled.  If the coarsest scale is in `sample_scales` its input is drawn from the
        uniform prior and quantised.
        draw(shape, lo, hi) -> uniforms (any device); default: torch.empty(shape, device='cuda').uniform_(lo, hi).  The
        draws are requested in the reference's order, so a seeded host generator reproduces its noise."""
        if auto_recurse != 0:
            raise NotImplementedError('Currently not supported for sampling: autorecurse={}'.format(auto_recurse))
        if self._rgb:
            raise NotImplementedError('sampling is implemented for the L3C configuration')
        _lib.require_gpu()
        if draw is None:
            draw = lambda shape, lo, hi: torch.empty(shape, device='cuda').uniform_(lo, hi)   # noqa: E731
        print('-' * 40)
        print('- Sampling {}'.format(sample_scales))
        print('-' * 40)
        x = self._as_device_image(x)
        pk = self._prepare()
        inp = ops.rgb_head(x, pk['ms1'][0], pk['ms1'][1], pk['ms2'][0], pk['ms2'][1], pk['head0'][0], pk['head0'][1])
        enc = []
        for s in range(self.scales):
            if s:
                inp = ops.conv(enc[-1][0], pk['heads'][s])
            enc.append(self._encoder(inp, s, pk))
        Cs = [3] + [e[2].shape[1] for e in enc]
        prev_x, fuse = None, None
        for s in reversed(range(self.scales)):
            loss_dmm = losses.loss_dmol_rgb if s == 0 else losses.loss_dmol_n
            if s in sample_scales:
                if prev_x is None:
                    print('Sampling uniformly!')
                    fake = draw(tuple(enc[-1][2].shape), -1, 1).to('cuda', torch.float32)
                    prev_x = self.quantize_x(fake, self.scales - 1)
                    if partial_final:
                        print('partial sampling')
                        for c in partial_final:
                            prev_x[:, c, ...] = enc[s][2][:, c, ...]
                print('{}: Feeding sampled to decoder'.format(s))
                dec_in = prev_x
            else:
                print('{}: Feeding encoder output to decoder'.format(s))
                dec_in = enc[s][2]
            fuse = self._decoder(dec_in.contiguous(), fuse, s, pk)
            P = self._prob(fuse, s, pk)
            if s == 0 or s - 1 in sample_scales:
                C = Cs[s]
                print('{}: sampling N{}HW for next scale'.format(s, C))
                B, H, W, Kp = P.shape
                K = Kp // ((4 if s == 0 else 3) * C)
                noise = (draw((B, C, K, H, W), 1e-5, 1. - 1e-5), draw((B, C, H, W), 1e-5, 1. - 1e-5))
                prev_x = loss_dmm.sample(P.permute(0, 3, 1, 2), C, noise=noise)
        return prev_x

    def quantize_x(self, x, scale):
        """EDSRLikeEnc.quantize_x (net.py:132-134): x (N,C,H,W) -> levels[argmin_l (x - levels_l)^2] of that scale's quantiser."""
        levels = self._prepare()['enc'][scale]['levels']
        d = (x.to('cuda', torch.float32).unsqueeze(-1) - levels) ** 2
        return levels[torch.min(d, dim=-1)[1]]


def schema_levels_check(sd, cfg):
    """The decoder rebuilds z from symbols with to_bn (quantizer.py:44-47) while the encoder feeds levels[sym]
    (net.py:125, quantizer.py:85): a checkpoint is only losslessly decodable when the two agree bit for bit."""
    lo, hi = cfg.q.levels_range
    expect = torch.arange(cfg.q.L, dtype=torch.float32).mul((hi - lo) / (cfg.q.L - 1)).add(lo)
    ok = True
    for s in range(cfg.num_scales):
        for name in ('nets.{}.enc.levels', 'nets.{}.enc.q.levels'):
            if not torch.equal(sd[name.format(s)].cpu().float(), expect):
                ok = False
    if not ok:
        import warnings
        warnings.warn('checkpoint quantiser levels differ from to_bn(arange(L)): encode->decode will not be lossless '
                      '(the reference has the same property; see SURVEY.md section 8c)')
    return ok
