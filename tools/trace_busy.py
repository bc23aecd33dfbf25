#!/usr/bin/env python
"""Development: how busy the GPU is over a traced run (rocprofv3 --kernel-trace --output-format csv): union of all kernel intervals
against the span from the first to the last kernel of the LAST burst of work, the kernels' summed durations by name, and the average
number of kernels in flight.

    python tools/trace_busy.py DIR [gap_ms]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 200e6
    rows = []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # last burst: everything after the last idle gap longer than `gap`
    cut, end = rows[0][0], rows[0][1]
    for a, b, _ in rows:
        if a - end > gap:
            cut = a
        end = max(end, b)
    rows = [r for r in rows if r[0] >= cut]
    span = max(b for _, b, _ in rows) - rows[0][0]
    busy, cur_a, cur_b = 0, rows[0][0], rows[0][1]
    for a, b, _ in rows:
        if a > cur_b:
            busy += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    tot = sum(b - a for a, b, _ in rows)
    by = collections.Counter()
    cnt = collections.Counter()
    for a, b, n in rows:
        k = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0][:60]
        by[k] += b - a
        cnt[k] += 1
    print('last burst: {} launches, span {:.1f} ms, some kernel running {:.1f} ms ({:.1f} %), summed kernel time {:.1f} ms (avg {:.2f} in flight)'.format(
        len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / float(busy)))
    for k, v in by.most_common(12):
        print('  {:9.1f} ms  n={:6d}  {}'.format(v / 1e6, cnt[k], k))


if __name__ == '__main__':
    main()
