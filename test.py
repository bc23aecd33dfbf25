#!/usr/bin/env python
"""Evaluation driver -- same arguments as the reference's src/test.py:44-137:

    python test.py LOG_DIR LOG_DATES IMAGES [--match_filenames F ..] [-m MAX] [--crop N] [--names N,..] [--overwrite_cache]
                   [--reset_entire_cache] [-i ITR,..] [--write_to_files DIR [--compare_theory] [--time_report PATH]]
                   [--sort_output testset|exp|itr|res] [--batch B]

Prints the mean bpsp of every (test set, experiment, iteration); results are cached in LOG_DIR_test/<experiment>/cache.pkl.
`--write_to_files` encodes every image to DIR/<name>.l3c with the HIP coder, decodes it again and asserts equality.
`--recursive auto|N` evaluates the RGB Shared baseline recursively; `--sample OUT_DIR` stores sampled images (RGB, RGB+z1,
RGB+z1+z2) next to the ground truth.
"""
import argparse
import os
import sys
from operator import itemgetter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import l3c_pytorch_amd  # noqa: E402,F401

l3c_pytorch_amd.configure_hip_queues()      # image sets / auto-crops run several forward passes side by side (before the first HIP call)
from l3c_pytorch_amd.helpers.testset import Testset  # noqa: E402
from l3c_pytorch_amd.test.multiscale_tester import MultiscaleTester  # noqa: E402


def print_aligned(rows):
    widths = [max(len(str(r[c])) for r in rows) for c in range(len(rows[0]))]
    for r in rows:
        print('  '.join(str(v).ljust(w) for v, w in zip(r, widths)))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('log_dir')
    p.add_argument('log_dates', help='comma-separated log dates, e.g. 0306_0001')
    p.add_argument('images', help='comma-separated directories of images or single image paths')
    p.add_argument('--match_filenames', '-fns', nargs='+', metavar='FILTER')
    p.add_argument('--max_imgs_per_folder', '-m', type=int, metavar='MAX')
    p.add_argument('--crop', type=int)
    p.add_argument('--names', '-n', type=str)
    p.add_argument('--overwrite_cache', '-f', action='store_true')
    p.add_argument('--reset_entire_cache', action='store_true')
    p.add_argument('--restore_itr', '-i', default='-1')
    p.add_argument('--recursive', default='0')
    p.add_argument('--sample', type=str, metavar='SAMPLE_OUT_DIR')
    p.add_argument('--write_to_files', type=str, metavar='WRITE_OUT_DIR')
    p.add_argument('--compare_theory', action='store_true')
    p.add_argument('--time_report', type=str, metavar='TIME_REPORT_PATH')
    p.add_argument('--sort_output', '-s', choices=['testset', 'exp', 'itr', 'res'], default='testset')
    p.add_argument('--batch', type=int, default=8, help='crops of equal padded shape evaluated per forward')
    p.add_argument('--io_threads', type=int, default=4, help='worker threads that read and decode the image files ahead of the GPU')
    p.add_argument('--write_window', type=int, default=None,
                   help='--write_to_files: images coded / decoded as one set (default 32 x --batch; 1 = one image at a time like the reference)')
    flags = p.parse_args(argv)

    if flags.compare_theory and not flags.write_to_files:
        raise ValueError('Cannot have --compare_theory without --write_to_files.')
    if flags.write_to_files and flags.sample:
        raise ValueError('Cannot have --write_to_files and --sample.')
    if flags.time_report and not flags.write_to_files:
        raise ValueError('--time_report only valid with --write_to_files.')

    testsets = [Testset(s.rstrip('/'), flags.max_imgs_per_folder, append_id='_crop{}'.format(flags.crop) if flags.crop else None)
                for s in flags.images.split(',')]
    if flags.match_filenames:
        for ts in testsets:
            ts.filter_filenames(flags.match_filenames)

    splitter = ',' if ',' in flags.log_dates else '|'
    log_dates = flags.log_dates.split(splitter)
    results = []
    for log_date in log_dates:
        for restore_itr in map(int, flags.restore_itr.split(',')):
            print('Testing {} at {} ---'.format(log_date, restore_itr))
            tester = MultiscaleTester(log_date, flags, restore_itr)
            results += tester.test_all(testsets)

    names = flags.names.split(splitter) if flags.names else log_dates
    label = {d: ('{} ({})'.format(n, d) if flags.names else d) for d, n in zip(log_dates, names)}
    if not flags.write_to_files:
        print('*** Summary:')
        sortby = {'testset': 0, 'exp': 1, 'itr': 2, 'res': 3}[flags.sort_output]
        rows = [('Testset', 'Experiment', 'Itr', 'Result')]
        for testset, log_date, restore_itr, result in sorted(results, key=itemgetter(sortby)):
            rows.append((testset.id, label[log_date], str(restore_itr), result))
        print_aligned(rows)
    return results


if __name__ == '__main__':
    main()
