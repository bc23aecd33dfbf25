#!/usr/bin/env python
"""L2 hit rate per kernel VARIANT from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass over bench.py (round-5 verdict, next 1: "explain the
residual 1.35x L2-turnover traffic with TCC_HIT / MISS per variant").  Joined, when a FETCH_SIZE / WRITE_SIZE root is given, with the bytes
that crossed the L2's memory side per launch (FETCH doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_tcc/tcc -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode ...
    python tools/pmc_tcc.py gpurun_out/pmc_tcc [gpurun_out/evidence/pmc]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_bench import variant_of  # noqa: E402

NAMES = {'false,false,false,false': 'plain (heads, dilated classifier branches)', 'true,false,false,false': 'ReLU', 'false,true,false,false': 'residual',
         'false,false,true,false': 'PixelShuffle tail (64 -> 256)', 'false,false,false,true': 'polyphase 5x5 stride 2 (four phases in one launch)'}


def rows(root, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] in counters:
                name = r['Kernel_Name']
                short = name.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
                short = short.split('<')[0] + ('<' + variant_of(name) + '>' if '<' in name else '')
                acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def main():
    tcc = rows(sys.argv[1], ('TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum'))
    mem = rows(sys.argv[2], ('FETCH_SIZE', 'WRITE_SIZE')) if len(sys.argv) > 2 else {}
    out = {}
    for k, c in sorted(tcc.items()):
        hit, miss = sum(c.get('TCC_HIT_sum', [0])), sum(c.get('TCC_MISS_sum', [0]))
        n = len(c.get('TCC_HIT_sum', [])) or 1
        if hit + miss < 1e6 * n:
            continue
        e = {'launches': n, 'l2_hits_per_launch': round(hit / n), 'l2_misses_per_launch': round(miss / n), 'l2_hit_rate': round(hit / (hit + miss), 4)}
        for extra in ('TCC_REQ_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum'):
            if extra in c:
                e[extra + '_per_launch'] = round(sum(c[extra]) / len(c[extra]))
        v = variant_of(k)
        if 'conv_wino4_kernel' in k and v in NAMES:
            e['variant'] = NAMES[v]
        if k in mem and 'FETCH_SIZE' in mem[k]:
            f = sum(mem[k]['FETCH_SIZE']) / len(mem[k]['FETCH_SIZE']) * 2048.0
            w = sum(mem[k].get('WRITE_SIZE', [0])) / max(1, len(mem[k].get('WRITE_SIZE', [0]))) * 1024.0
            e['memory_side_gb_per_launch'] = {'fetched (x2)': round(f / 1e9, 3), 'written': round(w / 1e9, 3)}
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
