"""`.l3c` container oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates bitcoding/bitcoding.py of the reference on top of the other oracle pieces:
  encode  <- Bitcoding.encode :50-123, encode_uniform :171-186, encode_scale :212-246, code_with_cdf :268-294
  decode  <- Bitcoding.decode :125-161, decode_uniform :188-205, decode_scale :248-266
  header  <- write_padding_tuple :355-359 (4 x u16), write_shape :326-338 (u8 C, u16 H, u16 W),
             write_num_bytes_encoded :345-348 (u32), magic separator :36
Works on bytes in memory (the reference writes a file); padding/auto-crop are done by the caller.
"""
import struct

import numpy as np
import torch

from . import ac, cdf, dmll, net

MAGIC = b'\x46\xE2\x84\x92'


def losses_bpsp(out, hp=net.L3C_HYPER, num_subpixels=None):
    """per-scale bpsp incl. the uniform cost of the coarsest scale  <- multiscale_network.Losses.get :145-165,
    multiscale_blueprint.get_loss :64-95."""
    zs = dmll.z_spec(hp.levels_range, hp.L)
    costs = [dmll.nll(dmll.RGB, out.S[0].float(), out.P[0]).sum()]
    for s in range(1, hp.num_scales):
        costs.append(dmll.nll(zs, out.bn[s], out.P[s]).sum())
    N, C, H, W = out.S[-1].shape
    final = N * C * H * W * np.log(hp.L)
    n = num_subpixels or int(np.prod(out.S[0].shape))
    conv = np.log(2.) * n
    return [float(c) / conv for c in costs] + [final / conv]


def encode(img, sd, hp=net.L3C_HYPER, padding_tuple=(0, 0, 0, 0), coder=None):
    """img: (1,3,H,W) int64 (already padded to a multiple of 2**num_scales) -> bytes of the .l3c file.
    coder: None = the C restatement (oracle/ac.py), or a callable(table (N, Lp) int16 array, symbols (N,) int16 array) -> bytes, e.g.
    the REFERENCE's compiled coder (oracle/cpu_baseline.py times the port's forward and tables with the reference's own torchac.cpp)."""
    enc = coder if coder is not None else ac.encode
    out = net.forward(img.float(), sd, hp)
    zs = dmll.z_spec(hp.levels_range, hp.L)
    chunks = [struct.pack('<4H', *padding_tuple)]
    for scale in reversed(range(hp.num_scales + 1)):
        S = out.S[scale]
        _, C, H, W = S.shape
        chunks.append(struct.pack('<BHH', C, H, W))
        if scale == hp.num_scales:
            table = cdf.uniform_cdf_table(H, W, hp.L)
            for c in range(C):
                b = enc(table.numpy(), S[0, c].to(torch.int16).numpy())
                chunks += [struct.pack('<I', len(b)), b]
        else:
            spec = dmll.RGB if scale == 0 else zs
            targets = cdf.coding_targets(spec.x_min, spec.x_max, spec.L)
            x = img.float() if scale == 0 else out.bn[scale]
            for c in range(C):
                pi, mu, ls = dmll.params_for_channel(spec, out.P[scale], c, C, x)
                table = cdf.mixture_cdf_table(pi, targets, mu, ls)
                b = enc(table.numpy(), S[0, c].to(torch.int16).numpy())
                chunks += [struct.pack('<I', len(b)), b]
        chunks.append(MAGIC)
    return b''.join(chunks)


class _Reader(object):
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        b = self.d[self.p:self.p + n]
        assert len(b) == n
        self.p += n
        return b

    def unpack(self, fmt):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))


def decode(data, sd, hp=net.L3C_HYPER):
    """bytes -> ((1,3,H,W) int64 padded image, padding_tuple)."""
    r = _Reader(data)
    padding_tuple = r.unpack('<4H')
    zs = dmll.z_spec(hp.levels_range, hp.L)
    bn_prev, f_prev = None, None
    for scale in reversed(range(hp.num_scales + 1)):
        if scale == hp.num_scales:
            C, H, W = r.unpack('<BHH')
            table = cdf.uniform_cdf_table(H, W, hp.L)
            S = []
            for c in range(C):
                n, = r.unpack('<I')
                S.append(torch.from_numpy(ac.decode(table.numpy(), r.take(n))).reshape(1, H, W))
            bn_prev = zs.to_bn(torch.stack(S, dim=1))
        else:
            P, f_prev = net.get_P(scale, bn_prev, f_prev, sd, hp)
            C, H, W = r.unpack('<BHH')
            spec = dmll.RGB if scale == 0 else zs
            targets = cdf.coding_targets(spec.x_min, spec.x_max, spec.L)
            decoded = torch.zeros(1, C, H, W)
            for c in range(C):
                pi, mu, ls = dmll.params_for_channel(spec, P, c, C, decoded)
                table = cdf.mixture_cdf_table(pi, targets, mu, ls)
                n, = r.unpack('<I')
                sym = torch.from_numpy(ac.decode(table.numpy(), r.take(n))).reshape(1, H, W)
                decoded[:, c] = spec.to_bn(sym)
            bn_prev = decoded
        assert r.take(4) == MAGIC
    return bn_prev.round().long(), padding_tuple


def losses_bpsp_rgb(out, num_subpixels=None):
    """RGB baselines: every scale is an RGB DMLL on symbols (targets_style 'S', losses shared: multiscale_network.py:136-139).
    -> (nonrecursive_bpsps, recursive_bpsps or None) like MultiscaleBlueprint.get_loss :64-95."""
    costs = [float(dmll.nll(dmll.RGB, out.S[i].float(), out.P[i]).sum()) for i in range(len(out.P))]
    n = num_subpixels or int(np.prod(out.S[0].shape))
    conv = np.log(2.) * n
    costs_bpsp = [c / conv for c in costs]

    def nat_count(i):
        return int(np.prod(out.S[i].shape)) * np.log(256)
    final_idx = -1 if out.auto_recursive_from is None else out.auto_recursive_from
    nonrec = costs_bpsp[:out.auto_recursive_from] + [nat_count(final_idx) / conv]
    rec = costs_bpsp + [nat_count(-1) / conv] if out.auto_recursive_from is not None else None
    return nonrec, rec
