#!/usr/bin/env python
"""Development: how much of the RGB decode's table time and decoder time overlap, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace -d gpurun_out/dtrace -o run --output-format csv -- python tools/decode_profile.py 128
    python tools/decode_overlap.py gpurun_out/dtrace

Intervals of cdf_table_from_P_kernel (T) and ac_decode_lean_kernel / ac_decode_ring_kernel (D) launches: union length of each, length of their
intersection, and the average launch durations -- `T + D - both` is the wall time the pipelined chunk steps need."""
import csv
import glob
import os
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


def intersect(u, v):
    i = j = 0
    tot = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if b > a:
            tot += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    root = sys.argv[1]
    T, D, other = [], [], []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            n = r['Kernel_Name']
            (T if 'cdf_table_from_P_kernel' in n else D if ('ac_decode_ring_kernel' in n or 'ac_decode_lean_kernel' in n) else other).append((a, b))
    if not T or not D:
        print('no table / decoder launches found under', root)
        return
    # the LAST decode of the process: everything after the last gap > 50 ms in the decoder launches
    D.sort()
    cut = D[0][0]
    for (a0, b0), (a1, b1) in zip(D, D[1:]):
        if a1 - b0 > 100e6:
            cut = a1
    T = [x for x in T if x[0] >= cut - 50e6]
    D = [x for x in D if x[0] >= cut]
    uT, uD = union(T), union(D)
    both = intersect(uT, uD)
    span = max(max(b for _, b in T), max(b for _, b in D)) - min(min(a for a, _ in T), min(a for a, _ in D))
    ms = 1e-6
    print('last decode: {} table launches (avg {:.2f} ms, union {:.1f} ms), {} decoder launches (avg {:.2f} ms, union {:.1f} ms)'.format(
        len(T), sum(b - a for a, b in T) / len(T) * ms, length(uT) * ms, len(D), sum(b - a for a, b in D) / len(D) * ms, length(uD) * ms))
    print('both running {:.1f} ms; table or decoder running {:.1f} ms; first launch to last end {:.1f} ms'.format(
        both * ms, (length(uT) + length(uD) - both) * ms, span * ms))


if __name__ == '__main__':
    main()
