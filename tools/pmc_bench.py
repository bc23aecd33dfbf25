#!/usr/bin/env python
"""Reduce rocprofv3 --pmc passes over `bench.py` to the per-kernel table bench.py reads (profiles/r03_pmc_bench.json).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_bench/fetch -o run --output-format csv -- python bench.py --steps 2 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_bench/write ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
              -d gpurun_out/pmc_bench/sq ...
    python tools/pmc_bench.py gpurun_out/pmc_bench BATCH > profiles/r03_pmc_bench.json

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM (separate passes; FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is DOUBLED; WRITE_SIZE was calibrated in round 1 against the known output
bytes of this kernel: 1.000x): hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.
MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / 8 /
kernel duration."""
import collections
import csv
import glob
import json
import os
import sys

KEYS = [('conv_wino4_kernel', 'conv_wino4_kernel'), ('conv_wino_kernel', 'conv_wino_kernel'), ('cdf_table_from_P_kernel', 'cdf_table_from_P_kernel'),
        ('ac_decode_ring_kernel', 'ac_decode_ring_kernel'), ('channel_params_kernel', 'channel_params_kernel'), ('conv_pw_kernel', 'conv k1 s1 (mfma)'), ('conv_lds_kernel<1', 'conv k1 s1 (mfma)'), ('conv_mfma_kernel<1', 'conv k1 s1 (mfma)'),
        ('conv_mfma_kernel<5', 'conv k5 s2 (mfma)'), ('encode_intervals_kernel', 'encode_intervals_kernel'),
        ('rgb_head_kernel', 'rgb_head_kernel'), ('ac_state_groups_kernel', 'ac_state_groups_kernel'),
        ('ac_pack_groups_kernel', 'ac_pack_groups_kernel'), ('to_q_quantize_kernel', 'to_q_quantize_kernel'),
        ('dec_head_kernel', 'dec_head_kernel')]


def key_of(name):
    for sub, key in KEYS:
        if sub in name:
            return key
    return None


def main():
    root, batch = sys.argv[1], int(sys.argv[2])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = key_of(row['Kernel_Name'])
            if k is None:
                continue
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
            if row['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                dur[k].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {'batch': batch, 'source': 'rocprofv3 --pmc passes over bench.py (tools/pmc_bench.py)', 'csrc_stamp': bench.csrc_stamp(),
           'kernels': {}, 'decode_kernels': {}}
    for k, c in acc.items():
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        e = {'launches_profiled': max(len(v) for v in c.values()), 'counters_per_launch': {n: round(m, 1) for n, m in mean.items()}}
        if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
            e['hbm_fetch_bytes_per_launch'] = mean['FETCH_SIZE'] * 1024 * 2
            e['hbm_write_bytes_per_launch'] = mean['WRITE_SIZE'] * 1024
            e['hbm_bytes_per_launch'] = e['hbm_fetch_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
        if 'GRBM_GUI_ACTIVE' in mean and dur[k]:
            ns = sum(dur[k]) / len(dur[k])
            e['avg_launch_us_profiled'] = round(ns / 1e3, 2)
            e['effective_clock_ghz'] = round(mean['GRBM_GUI_ACTIVE'] / 8 / ns, 3)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean:
                e['mfma_busy_frac'] = round(mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        if k in ('cdf_table_from_P_kernel', 'ac_decode_ring_kernel', 'channel_params_kernel'):
            # decode side: GB/s against the HBM peak and, for the table kernel, sigmoid evaluations per second
            if 'hbm_bytes_per_launch' in e and dur[k]:
                ns = sum(dur[k]) / len(dur[k])
                e['hbm_gb_per_s'] = round(e['hbm_bytes_per_launch'] / ns, 1)
                e['frac_of_hbm_peak_8tb'] = round(e['hbm_bytes_per_launch'] / ns / 8000.0, 4)
            if k == 'cdf_table_from_P_kernel' and 'hbm_write_bytes_per_launch' in e and dur[k]:
                # every uint16 entry written is the sum of K = 10 sigmoids (BASELINE configs: prob.K = 10): SURVEY.md section 8d's exp/s figure
                ns = sum(dur[k]) / len(dur[k])
                e['table_entries_per_launch'] = int(e['hbm_write_bytes_per_launch'] / 2)
                e['sigmoid_evals_per_s'] = round(e['hbm_write_bytes_per_launch'] / 2 * 10 / (ns * 1e-9), -9)
            out['decode_kernels'][k] = e
        else:
            out['kernels'][k] = e
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
