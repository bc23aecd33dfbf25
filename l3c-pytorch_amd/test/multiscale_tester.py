"""MultiscaleTester, the enc / dec side of the reference's test/multiscale_tester.py (`__init__` :160-220, `encode` :383-395,
`decode` :397-408, `_read_img` / `_write_img` :410-434) on the MI355X path.  The bpsp evaluation driver with its result
cache (`test_all` :236-351) is listed under "next" (SURVEY.md section 8f, item 1)."""
import os

import numpy as np
import torch
from PIL import Image

from ..bitcoding.bitcoding import Bitcoding
from ..blueprints.multiscale_blueprint import MultiscaleBlueprint
from ..helpers import config_parser, paths


class EncodeError(Exception):
    pass


class DecodeError(Exception):
    pass


class MultiscaleTester(object):
    def __init__(self, log_date, flags, restore_itr, l3c=False, configs_dir=None):
        """flags needs `.log_dir`; optional `.compare_theory`."""
        self.flags = flags
        self.log_date = log_date
        experiment_dir = paths.get_experiment_dir(flags.log_dir, log_date)
        configs_dir = configs_dir or config_parser.CONFIG_DIR
        (config_p_ms, _), postfix = paths.parse_log_dir(experiment_dir, configs_dir)
        self.config_ms, _ = config_parser.parse(config_p_ms)
        if postfix:   # "key=value" overrides carried by the directory name (global_config.py:73-97)
            for kv in postfix:
                if '=' in kv:
                    k, v = kv.split('=', 1)
                    self.config_ms.set_attr(k, config_parser._eval_value(v))
        self.blueprint = MultiscaleBlueprint(self.config_ms)
        self.blueprint.set_eval()
        self.restore_itr, ckpt_p = paths.get_ckpt_for_itr(paths.get_ckpts_dir(experiment_dir), restore_itr)
        paths.restore({'net': self.blueprint.net}, ckpt_p, strict=True)
        self.bc = Bitcoding(self.blueprint, compare_with_theory=bool(getattr(flags, 'compare_theory', False)))

    def encode(self, img_p, pout, overwrite=False):
        pout_dir = os.path.dirname(os.path.abspath(pout))
        if not os.path.isdir(pout_dir):
            raise EncodeError('pout directory ({}) does not exists!'.format(pout_dir))
        if overwrite and os.path.isfile(pout):
            print('Removing {}...'.format(pout))
            os.remove(pout)
        if os.path.isfile(pout):
            raise EncodeError('{} exists. Consider --overwrite'.format(pout))
        img = self._read_img(img_p)
        bpsp = self.bc.encode(img, pout=pout)
        print('---\nSaved:', pout)
        return bpsp

    def decode(self, pin, png_out_p):
        pout_dir = os.path.dirname(os.path.abspath(png_out_p))
        if not os.path.isdir(pout_dir):
            raise DecodeError('png_out_p directory ({}) does not exists!'.format(pout_dir))
        if not png_out_p.endswith('.png'):
            raise DecodeError('png_out_p must end in .png, got {}'.format(png_out_p))
        decoded = self.bc.decode(pin)
        self._write_img(decoded, png_out_p)
        print('---\nDecoded: {}'.format(png_out_p))

    @staticmethod
    def _read_img(img_p):
        img = np.array(Image.open(img_p))
        if img.ndim != 3:
            raise EncodeError('Image has {} dimensions, expected HxWxC.'.format(img.ndim))
        img = img.transpose(2, 0, 1)
        C = img.shape[0]
        if C == 4:
            print('*** WARN: Will discard 4th (alpha) channel.')
            img = img[:3, ...]
        elif C != 3:
            raise EncodeError('Image has {} channels, expected 3 or 4.'.format(C))
        return torch.from_numpy(np.ascontiguousarray(img)).unsqueeze(0).long()

    @staticmethod
    def _write_img(decoded, png_out_p):
        assert decoded.shape[0] == 1 and decoded.shape[1] == 3, decoded.shape
        img = decoded.squeeze(0).cpu().numpy().transpose(1, 2, 0).astype(np.uint8)
        Image.fromarray(img).save(png_out_p)
