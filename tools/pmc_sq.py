#!/usr/bin/env python
"""Per-launch averages of rocprofv3 PMC counters for one kernel: python tools/pmc_sq.py OUT_DIR [kernel substring]."""
import collections
import csv
import glob
import os
import sys

out_dir = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else 'conv_wino_kernel'
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        if kernel in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('{:32s} {:16.0f}  (n={})'.format(k, sum(v) / len(v), len(v)))
