#!/usr/bin/env python
"""Reduce rocprofv3 --pmc passes over `bench.py` to the per-kernel table bench.py reads (profiles/r05_pmc_bench.json).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_bench/fetch -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode ...
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_bench/write ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
              -d gpurun_out/pmc_bench/sq ...
    python tools/pmc_bench.py gpurun_out/pmc_bench BATCH [--steps 2] [--decode-root gpurun_out/pmc_decode] > profiles/r05_pmc_bench.json

ONLY the launches of bench.py's timed region are reduced (round-3 verdict: the round-3 table averaged warm-up, timed steps AND the
decode leg's batch-128 / batch-1 launches, so its per-launch bytes were not comparable with the algorithmic bytes of the timed
region): the passes run with --no-decode --no-parity --no-extra-legs --no-cpu-baseline, so that every encode-side kernel is launched
by (warm-up + steps) identical steps and nothing else; of each kernel's launches, sorted by start time, the LAST steps x
launches_per_step are kept (launches_per_step = count / (warm-up + steps), which must divide).  `launches_profiled` therefore
equals steps x launches_per_step, and `avg_launch_us_profiled` is comparable with the HIP-event average of an unprofiled run
(bench.py checks both).  The decode-side kernels come from a SEPARATE set of passes over a decode-only command (--decode-root).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM (separate passes; FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is DOUBLED; WRITE_SIZE was calibrated in round 1 against the known output
bytes of this kernel: 1.000x): hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.
MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / 8 /
kernel duration."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

KEYS = [('conv_wino4_kernel', 'conv_wino4_kernel'), ('conv_wino4x_kernel', 'conv_wino4_kernel'), ('conv_wino_kernel', 'conv_wino_kernel'),
        ('cdf_table_from_P_kernel', 'cdf_table_from_P_kernel'),
        ('ac_decode_lean_kernel', 'ac_decode_lean_kernel'), ('ac_decode_ring_kernel', 'ac_decode_ring_kernel'), ('channel_params_kernel', 'channel_params_kernel'), ('conv_pw_kernel', 'conv k1 s1 (mfma)'), ('conv_lds_kernel<1', 'conv k1 s1 (mfma)'), ('conv_mfma_kernel<1', 'conv k1 s1 (mfma)'),
        ('conv_mfma_kernel<5', 'conv k5 s2 (mfma)'), ('encode_intervals_kernel', 'encode_intervals_kernel'),
        ('rgb_head_kernel', 'rgb_head_kernel'), ('ac_state_groups_kernel', 'ac_state_groups_kernel'),
        ('ac_pack_groups_kernel', 'ac_pack_groups_kernel'), ('to_q_quantize_kernel', 'to_q_quantize_kernel'),
        ('dec_head_kernel', 'dec_head_kernel')]
DECODE_KEYS = ('cdf_table_from_P_kernel', 'ac_decode_lean_kernel', 'ac_decode_ring_kernel', 'channel_params_kernel')


def key_of(name):
    # the polyphase instantiations of the F(4x4,3x3) kernel (4th template argument) are the 5x5 stride-2 layers: bench.py's key for them
    if 'conv_wino4' in name and variant_of(name).endswith(',true') and variant_of(name).count(',') == 3:
        return 'conv k5 s2 (4 F(4x4,3x3) phases)'
    for sub, key in KEYS:
        if sub in name:
            return key
    return None


def variant_of(name):
    """the template arguments of a kernel name, e.g. conv_wino4_kernel<true, false, false> -> 'true,false,false'"""
    m = re.search(r'<([^>]*)>', name)
    return m.group(1).replace(' ', '') if m else ''


def read_rows(root):
    """-> {key: {counter: [(start, end, value, variant), ...]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = key_of(row['Kernel_Name'])
            if k is None:
                continue
            acc[k][row['Counter_Name']].append((int(row['Start_Timestamp']), int(row['End_Timestamp']), float(row['Counter_Value']),
                                                variant_of(row['Kernel_Name'])))
    return acc


def window(rows, steps, total_steps):
    """The launches of the last `steps` of `total_steps` identical steps (None when the count does not divide: not a per-step kernel)."""
    rows = sorted(rows)
    if total_steps <= 0 or len(rows) % total_steps:
        return None
    per_step = len(rows) // total_steps
    return rows[len(rows) - steps * per_step:]


def reduce_kernel(c, steps, total_steps, windowed):
    e = {}
    sel = {}
    for n, rows in c.items():
        w = window(rows, steps, total_steps) if windowed else sorted(rows)
        if w is None:
            e['window'] = 'launch count {} does not divide by {} steps: all launches averaged'.format(len(rows), total_steps)
            w = sorted(rows)
        sel[n] = w
    mean = {n: sum(r[2] for r in w) / len(w) for n, w in sel.items()}
    e['launches_profiled'] = max(len(w) for w in sel.values())
    if windowed:
        e['launches_per_step'] = e['launches_profiled'] // steps
    e['counters_per_launch'] = {n: round(m, 1) for n, m in mean.items()}
    if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
        e['hbm_fetch_bytes_per_launch'] = mean['FETCH_SIZE'] * 1024 * 2
        e['hbm_write_bytes_per_launch'] = mean['WRITE_SIZE'] * 1024
        e['hbm_bytes_per_launch'] = e['hbm_fetch_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
        by_var = collections.defaultdict(lambda: [0.0, 0.0, 0])
        for n, scale in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):
            for r in sel[n]:
                by_var[r[3]][0 if n == 'FETCH_SIZE' else 1] += r[2] * scale
                if n == 'FETCH_SIZE':
                    by_var[r[3]][2] += 1
        if len(by_var) > 1:
            e['per_variant'] = {v: {'launches': x[2], 'hbm_bytes_per_launch': (x[0] + x[1]) / max(x[2], 1)} for v, x in sorted(by_var.items())}
    ns = None
    if 'GRBM_GUI_ACTIVE' in sel:
        d = [r[1] - r[0] for r in sel['GRBM_GUI_ACTIVE']]
        ns = sum(d) / len(d)
        e['avg_launch_us_profiled'] = round(ns / 1e3, 2)
        e['effective_clock_ghz'] = round(mean['GRBM_GUI_ACTIVE'] / 8 / ns, 3)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean:
            e['mfma_busy_frac'] = round(mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
    return e, ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('batch', type=int)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--decode-root', default=None, help='PMC passes over a decode-only command (tools/codec_probe.py): the decode_kernels table')
    a = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {'batch': a.batch, 'steps': a.steps, 'warmup': a.warmup,
           'source': 'rocprofv3 --pmc passes over bench.py --no-decode (tools/pmc_bench.py): the launches of the timed region only',
           'csrc_stamp': bench.csrc_stamp(), 'csrc_files': bench.csrc_file_stamps(), 'kernels': {}, 'decode_kernels': {}}
    for k, c in read_rows(a.root).items():
        if k in DECODE_KEYS:
            continue
        out['kernels'][k], _ = reduce_kernel(c, a.steps, a.steps + a.warmup, True)
    if a.decode_root:
        for k, c in read_rows(a.decode_root).items():
            if k not in DECODE_KEYS:
                continue
            e, ns = reduce_kernel(c, 0, 0, False)
            # decode side: GB/s against the HBM peak and, for the table kernel, sigmoid evaluations per second
            if 'hbm_bytes_per_launch' in e and ns:
                e['hbm_gb_per_s'] = round(e['hbm_bytes_per_launch'] / ns, 1)
                e['frac_of_hbm_peak_8tb'] = round(e['hbm_bytes_per_launch'] / ns / 8000.0, 4)
            if k == 'cdf_table_from_P_kernel' and 'hbm_write_bytes_per_launch' in e and ns:
                # every uint16 entry written is the sum of K = 10 sigmoids (BASELINE configs: prob.K = 10): SURVEY.md section 8d's exp/s figure
                e['table_entries_per_launch'] = int(e['hbm_write_bytes_per_launch'] / 2)
                e['sigmoid_evals_per_s'] = round(e['hbm_write_bytes_per_launch'] / 2 * 10 / (ns * 1e-9), -9)
            out['decode_kernels'][k] = e
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
