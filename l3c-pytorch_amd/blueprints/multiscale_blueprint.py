"""MultiscaleBlueprint -- same surface as the reference's blueprints/multiscale_blueprint.py:42-150 (`net`, `losses`,
`set_eval`, `forward`, `get_loss`, `unpack_batch_pad`, `pad`, `get_padding_mode`, `unpack`), inference only."""
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from ..helpers.pad import pad
from ..modules.multiscale_network import MultiscaleNetwork, Out  # noqa: F401

MultiscaleLoss = namedtuple('MultiscaleLoss', ['loss_pc',             # loss to minimize
                                               'nonrecursive_bpsps',  # bpsp of the non-recursive scales + uniform cost
                                               'recursive_bpsps'])    # None if not recursive


class MultiscaleBlueprint(nn.Module):
    def __init__(self, config_ms):
        super(MultiscaleBlueprint, self).__init__()
        self.net = MultiscaleNetwork(config_ms)
        self.losses = self.net.get_losses()

    def set_eval(self):
        self.net.eval()
        self.losses.loss_dmol_rgb.eval()
        self.losses.loss_dmol_n.eval()

    def forward(self, in_batch, auto_recurse=0):
        """in_batch: NCHW 0..255 float -> Out"""
        return self.net(in_batch, auto_recurse)

    def get_loss(self, out, num_subpixels_before_pad=None):
        """nats -> bits per sub-pixel per scale (reference :64-95)."""
        costs, final_cost_uniform, num_subpixels = self.losses.get(out)
        if num_subpixels_before_pad:
            assert num_subpixels_before_pad <= num_subpixels, num_subpixels_before_pad
            num_subpixels = num_subpixels_before_pad
        conversion = np.log(2.) * num_subpixels
        costs_bpsp = [cost / conversion for cost in costs]
        nonrecursive_bpsps = costs_bpsp[:out.auto_recursive_from] + [final_cost_uniform / conversion]
        if out.auto_recursive_from is not None:
            recursive_bpsps = costs_bpsp + [out.get_nat_count(-1) / conversion]
        else:
            recursive_bpsps = None
        return MultiscaleLoss(sum(costs_bpsp), nonrecursive_bpsps, recursive_bpsps)

    def sample_forward(self, in_batch, sample_scales, partial_final=None):
        return self.net.sample_forward(in_batch, self.losses, sample_scales, partial_final)

    @staticmethod
    def unpack_batch_pad(raw, fac):
        """raw: uint8 image (CHW or NCHW) -> (float batch on the GPU, long symbols), padded to a multiple of `fac`."""
        if raw.dim() == 3:
            raw = raw.unsqueeze(0)
        assert raw.dim() == 4
        raw = MultiscaleBlueprint.pad(raw, fac)
        raw = raw.to('cuda')
        return raw.float(), raw.long()

    @staticmethod
    def pad(raw, fac):
        raw, _ = pad(raw, fac, mode=MultiscaleBlueprint.get_padding_mode())
        return raw

    @staticmethod
    def get_padding_mode():
        return 'constant'

    @staticmethod
    def unpack(img_batch):
        idxs = img_batch['idx'].squeeze().tolist()
        raw = img_batch['raw'].to('cuda')
        return idxs, raw.float(), raw.long()
