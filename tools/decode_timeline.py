#!/usr/bin/env python
"""Development: time line of the LAST decode_batch of a rocprofv3 kernel trace (tools/decode_profile.py) -- per kernel: launches,
busy time (union of its launches), first start / last end relative to the decode's first kernel; and the GPU-idle gaps.

    rocprofv3 --kernel-trace -d D -o run --output-format csv -- python tools/decode_profile.py 128 ; python tools/decode_timeline.py D"""
import csv
import glob
import os
import re
import sys
from decode_overlap import union, length


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    return n[:70]


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    cut = rows[0][0]
    end = rows[0][1]
    for a, b, n in rows:
        if a - end > 100e6:
            cut = a
        end = max(end, b)
    rows = [r for r in rows if r[0] >= cut]
    t0 = rows[0][0]
    ms = 1e-6
    names = {}
    for a, b, n in rows:
        names.setdefault(n, []).append((a, b))
    print('last decode: {} launches, first start to last end {:.1f} ms, some kernel running {:.1f} ms'.format(
        len(rows), (max(b for _, b, _ in rows) - t0) * ms, length(union([(a, b) for a, b, _ in rows])) * ms))
    print('{:72s} {:>5s} {:>9s} {:>9s} {:>9s} {:>9s}'.format('kernel', 'n', 'busy ms', 'sum ms', 'first', 'last end'))
    for n, iv in sorted(names.items(), key=lambda kv: min(a for a, _ in kv[1])):
        print('{:72s} {:5d} {:9.1f} {:9.1f} {:9.1f} {:9.1f}'.format(n, len(iv), length(union(iv)) * ms, sum(b - a for a, b in iv) * ms,
                                                                     (min(a for a, _ in iv) - t0) * ms, (max(b for _, b in iv) - t0) * ms))
    u = union([(a, b) for a, b, _ in rows])
    gaps = sorted(((c - b, b - t0) for (a, b), (c, d) in zip(u, u[1:])), reverse=True)[:8]
    print('largest idle gaps (ms, at ms):', [(round(g * ms, 2), round(at * ms, 1)) for g, at in gaps])
    if len(sys.argv) > 2 and sys.argv[2] == 'clusters':    # when each kernel ran: its launches clustered (gaps > 30 ms split), as [start, end] x n in ms
        for n, iv in sorted(names.items(), key=lambda kv: min(a for a, _ in kv[1])):
            iv = sorted(iv)
            cl, cur = [], [iv[0][0], iv[0][1], 1]
            for a, b in iv[1:]:
                if a - cur[1] > 30e6:
                    cl.append(cur)
                    cur = [a, b, 1]
                else:
                    cur[1] = max(cur[1], b)
                    cur[2] += 1
            cl.append(cur)
            print('{:60s} {}'.format(n[:60], '  '.join('[{:.0f}, {:.0f}] x{}'.format((a - t0) * ms, (b - t0) * ms, k) for a, b, k in cl)))
    if len(sys.argv) > 2 and sys.argv[2] == 'around':      # what ran right before and right after the four largest gaps
        for g, at in gaps[:4]:
            before = sorted((r for r in rows if r[1] - t0 <= at), key=lambda r: r[1])[-4:]
            after = sorted((r for r in rows if r[0] - t0 >= at + g), key=lambda r: r[0])[:6]
            print('gap of {:.2f} ms at {:.1f} ms: before: {} | after: {}'.format(
                g * ms, at * ms, ', '.join('{} (ended {:.2f} ms before)'.format(n[:40], (at - (b - t0)) * ms) for a, b, n in before),
                ', '.join('{} (+{:.2f} ms)'.format(n[:40], (a - t0 - at - g) * ms) for a, b, n in after)))


if __name__ == '__main__':
    main()
