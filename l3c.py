#!/usr/bin/env python
"""Encoder / decoder CLI -- same arguments as the reference's src/l3c.py:74-125:

    python l3c.py LOG_DIR LOG_DATE [--device auto|gpu|cpu] [-i ITR] enc IMG_P OUT_P [--overwrite]
    python l3c.py LOG_DIR LOG_DATE [--device auto|gpu|cpu] [-i ITR] dec IMG_P OUT_P_PNG

Everything runs on the MI355X path; `--device cpu` is refused (there is no CPU back end in this build).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import l3c_pytorch_amd  # noqa: E402,F401

l3c_pytorch_amd.configure_hip_queues()      # image sets / auto-crops run several forward passes side by side (before the first HIP call)
from l3c_pytorch_amd import torchac  # noqa: E402
from l3c_pytorch_amd.test.multiscale_tester import DecodeError, EncodeError, MultiscaleTester  # noqa: E402


def parse_device_flag(flag):
    import torch
    gpu = torch.cuda.is_available()
    print('Status: torchac-backend-gpu (HIP) available: {} // torchac-backend-cpu available: {} // GPU available: {}'.format(
        torchac.CUDA_SUPPORTED, torchac.CPU_SUPPORTED, gpu))
    if flag == 'auto':
        flag = 'gpu'
    if flag == 'cpu':
        raise ValueError('torchac-backend-cpu is not available: this build only has the MI355X (HIP) path.')
    if not gpu:
        raise ValueError('Selected GPU backend but no GPU is visible!')
    print('*** Using the HIP back end on', torch.cuda.get_device_name(0))


def main(argv=None):
    p = argparse.ArgumentParser(description='Encoder/Decoder for L3C')
    p.add_argument('log_dir', help='Directory of experiments.')
    p.add_argument('log_date', help='A log_date, such as 0104_1345.')
    p.add_argument('--device', type=str, choices=['auto', 'gpu', 'cpu'], default='auto')
    p.add_argument('--restore_itr', '-i', default=-1, type=int,
                   help='Which iteration to restore. -1 means latest iteration. Default: -1')
    p.add_argument('--compare_theory', action='store_true', help='print cross-entropy bpsp next to the on-disk bpsp')
    mode = p.add_subparsers(title='mode', dest='mode')
    enc = mode.add_parser('enc', help='Encode image: enc IMG_P OUT_P [--overwrite | -f]')
    dec = mode.add_parser('dec', help='Decode image: dec IMG_P OUT_P_PNG')
    enc.add_argument('img_p')
    enc.add_argument('out_p')
    enc.add_argument('--overwrite', '-f', action='store_true')
    dec.add_argument('img_p')
    dec.add_argument('out_p_png')
    flags = p.parse_args(argv)
    if flags.mode is None:
        p.error('mode (enc | dec) required')
    parse_device_flag(flags.device)
    print('Testing {} at {} ---'.format(flags.log_date, flags.restore_itr))
    tester = MultiscaleTester(flags.log_date, flags, flags.restore_itr, l3c=True)
    if flags.mode == 'enc':
        try:
            tester.encode(flags.img_p, flags.out_p, flags.overwrite)
        except EncodeError as e:
            print('*** EncodeError:', e)
            return 1
    else:
        try:
            tester.decode(flags.img_p, flags.out_p_png)
        except DecodeError as e:
            print('*** DecodeError:', e)
            return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
