#!/usr/bin/env python
"""Build-time check for the decoders' software-pipelined LDS reads (csrc/ac_kernels.hip: lds_row_issue / lds_row_take of the
generic ring decoder, row_hi_issue / row_hi_wait of the lean one).

The rows are fetched by inline-asm ds_read_u16 / ds_read_u16_d16_hi whose completion the compiler knows nothing about, so the
registers they target must not be touched by compiler-generated code between the issue and the wait (s_waitcnt lgkmcnt(0) in an
asm block).  This script compiles the kernels to ISA, rebuilds the control-flow graph of every ac_decode_ring_kernel and
ac_decode_lean_kernel instantiation, propagates the state "reads in flight" (set by an asm block with such a read, cleared by
an asm block with s_waitcnt lgkmcnt(0) -- the hand-written block loop of the lean decoder is one asm block that ends waited)
and verifies that no instruction outside inline asm names one of the target registers while reads may be in flight.

    python tools/check_asm_prefetch.py        (exit status 0 = ok)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABEL = re.compile(r'^(\.LBB\d+_\d+):')
BRANCH = re.compile(r'^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)')


def compile_isa():
    src = os.path.join(ROOT, 'l3c-pytorch_amd', 'csrc', 'ac_kernels.hip')
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'ac.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                               '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', out, src],
                              cwd=tmp, stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def regs_named(code):
    found = set(re.findall(r'\bv(\d+)\b', code))
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', code):
        found.update(str(r) for r in range(int(a), int(b) + 1))
    return {int(r) for r in found}


def check_function(name, lines):
    # ---- split into items: ('label', name) | ('asm', [lines]) | ('ins', line)
    items, asm = [], None
    for l in lines:
        code = l.split(';')[0].rstrip()
        if '#ASMSTART' in l:
            asm = []
            continue
        if '#ASMEND' in l:
            items.append(('asm', asm))
            asm = None
            continue
        if asm is not None:
            asm.append(l.strip())
            continue
        m = LABEL.match(l)
        if m:
            items.append(('label', m.group(1)))
        elif code.strip() and not code.strip().startswith('.'):
            items.append(('ins', code.strip()))
    # ---- basic blocks
    blocks, cur = [], {'label': None, 'items': []}
    for it in items:
        if it[0] == 'label':
            blocks.append(cur)
            cur = {'label': it[1], 'items': []}
            continue
        cur['items'].append(it)
        if it[0] == 'ins' and (BRANCH.match(it[1]) or it[1].startswith('s_endpgm')):
            blocks.append(cur)
            cur = {'label': None, 'items': []}
    blocks.append(cur)
    blocks = [b for b in blocks if b['items'] or b['label']]
    index = {b['label']: i for i, b in enumerate(blocks) if b['label']}
    succ = []
    for i, b in enumerate(blocks):
        last = b['items'][-1] if b['items'] else None
        s = []
        if last and last[0] == 'ins':
            m = BRANCH.match(last[1])
            if m:
                s.append(index[m.group(2)])
                if m.group(1) != 's_branch' and i + 1 < len(blocks):
                    s.append(i + 1)
            elif last[1].startswith('s_endpgm'):
                pass
            elif i + 1 < len(blocks):
                s.append(i + 1)
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    # ---- forward dataflow: which registers may have an LDS read in flight at block entry?
    entry = [frozenset()] * len(blocks)
    n_issue = 0

    def transfer(b, state, report=None):
        for kind, body in b['items']:
            if kind == 'asm':
                dests = {int(m.group(1)) for l in body for m in [re.match(r'ds_read_u16(?:_d16_hi)? v(\d+),', l)] if m}
                if dests:
                    state = state | dests
                if any('lgkmcnt(0)' in l for l in body):
                    state = frozenset()
            elif state and report is not None and regs_named(body) & state:
                report.append(body)
        return frozenset(state)

    for b in blocks:
        n_issue += sum(1 for kind, body in b['items'] if kind == 'asm' and any(l.startswith('ds_read_u16') for l in body))
    if not n_issue:
        print(name, ': no ds_read_u16 found')
        return 1
    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            out = transfer(b, entry[i])
            for j in succ[i]:
                if not out <= entry[j]:
                    entry[j] = entry[j] | out
                    changed = True
    hits = []
    for i, b in enumerate(blocks):
        transfer(b, entry[i], hits)
    print('{}: {} prefetch sites -- {} compiler-generated uses of a register with a read in flight'.format(
        name[:72], n_issue, len(hits)))
    for h in hits[:10]:
        print('    ', h)
    return len(hits)


def main():
    text = compile_isa()
    funcs = [i for i, l in enumerate(text) if re.match(r'^_ZN.*ac_decode_(ring|lean)_kernel.*:\s', l)]
    if len(funcs) != 12:
        print('expected 5 generic + 7 lean decoder instantiations (two of each also hold the window-row body), found', len(funcs))
        return 1
    bad = 0
    for start in funcs:
        end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
        bad += check_function(text[start].split(':')[0], text[start + 1:end])
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
