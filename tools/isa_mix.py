#!/usr/bin/env python
"""Instruction mix of a kernel's MFMA loop, from the compiled ISA (round-5 verdict, next 1: "report from the compiled ISA: instructions per
MFMA-equivalent").  Splits the kernel's listing into basic blocks (label to label), prints for every block that holds MFMAs its counts by
class, and the totals of the steady-state loop (the blocks of the innermost backward branch that hold the most MFMAs).

    python tools/isa_mix.py conv_wino4w.hip conv_wino4w_kernel [mfma cycles per instruction]
    python tools/isa_mix.py conv_wino4.hip conv_wino4_kernelILb1ELb0ELb0ELb0 32"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'l3c-pytorch_amd', 'csrc')


def isa_of(src):
    out = os.path.join(tempfile.gettempdir(), 'isa_mix_' + os.path.basename(src) + '.s')
    if not os.path.isfile(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(CSRC, src)):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
                               '--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', out], stderr=subprocess.DEVNULL)
    return open(out).read().split('\n')


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op in ('s_waitcnt',):
        return 'waitcnt'
    if op in ('s_nop',):
        return 'nop'
    if op in ('s_barrier',):
        return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    mfma_cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    lines = isa_of(src)
    start = next(i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_]\S*' + re.escape(kernel) + r'\S*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    blocks, cur = [], {'label': 'entry', 'n': {}}
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r'^(\.LBB\S+):', t)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1), 'n': {}, 'branches': []}
            continue
        if not t or t.startswith((';', '.', '//')):
            continue
        op = t.split()[0]
        c = classify(op)
        cur['n'][c] = cur['n'].get(c, 0) + 1
        if c == 'branch':
            cur.setdefault('branches', []).append(t.split()[-1])
    blocks.append(cur)
    keys = ['mfma', 'valu', 'salu', 'lds', 'vmem', 'waitcnt', 'nop', 'barrier', 'branch', 'other']
    print('{:14s}'.format('block') + ''.join('{:>8s}'.format(k) for k in keys) + '   total  besides/mfma')
    tot_all = {}
    for b in blocks:
        for k, v in b['n'].items():
            tot_all[k] = tot_all.get(k, 0) + v
        if b['n'].get('mfma', 0):
            n = sum(b['n'].values())
            print('{:14s}'.format(b['label'][:14]) + ''.join('{:8d}'.format(b['n'].get(k, 0)) for k in keys) +
                  '  {:6d}  {:6.2f}'.format(n, (n - b['n']['mfma']) / b['n']['mfma']))
    n = sum(tot_all.values())
    print('{:14s}'.format('whole kernel') + ''.join('{:8d}'.format(tot_all.get(k, 0)) for k in keys) + '  {:6d}'.format(n))
    # the steady-state loop = the block with the most MFMAs
    loop = max(blocks, key=lambda b: b['n'].get('mfma', 0))
    m = loop['n']['mfma']
    others = sum(loop['n'].values()) - m
    print('steady-state block {}: {} MFMAs of {} cycles, {} other instructions = {:.2f} per MFMA = {:.2f} per 32 matrix-pipe cycles'.format(
        loop['label'], m, mfma_cycles, others, others / m, others / m * 32 / mfma_cycles))
    for l in lines[end:end + 400]:
        if re.search(r'\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|agpr_count)', l):
            pass
    meta = [l.strip() for l in lines if re.match(r'\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|agpr_count):', l)]
    for i, l in enumerate(meta):
        if l.startswith('.name:') and kernel in l:
            print('  '.join(meta[max(0, i - 6):i + 6]))
            break


if __name__ == '__main__':
    main()
