#!/usr/bin/env python
"""Build-time check of the register budget of the kernels whose speed depends on it (exit status 0 = ok).

`conv_wino4_kernel` holds 144 accumulators per lane and runs two blocks per CU: every variant the network launches must fit 256
VGPRs WITHOUT scratch spills (a spill inside the chunk loop is a scratch round trip per MFMA group; the round-3 residual variant spilled 7
registers until its residual ring was deepened -- the allocator's choice, not visible in the source).  The decoder ring kernel keeps its
table row in named registers: no spills either.  Compiles with csrc/build.py's flags and reads the code-object
metadata hipcc prints into the ISA listing.

    python tools/check_registers.py
"""
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_store_hazard import CSRC, compile_to_isa  # noqa: E402

# source -> [(substring of the mangled kernel name, max VGPRs, max spilled VGPRs)]
BUDGET = {
    'conv_wino4.hip': [('conv_wino4_kernelILb0ELb0ELb0', 256, 0),      # plain
                       ('conv_wino4_kernelILb1ELb0ELb0', 256, 0),      # ReLU
                       ('conv_wino4_kernelILb0ELb1ELb0', 256, 0),      # residual
                       ('conv_wino4_kernelILb0ELb0ELb1', 256, 0)],     # PixelShuffle tail
    'conv_wino.hip': [('conv_wino_kernel', 256, 0)],
    'conv_pw.hip': [('conv_pw_kernelILi4ELb0', 168, 0),                # three blocks of four waves per CU
                    ('conv_pw_kernelILi4ELb1', 256, 0)],               # the split-fifth-group form (128 < Cout <= 160): two blocks per CU
    'ac_kernels.hip': [('ac_decode_ring_kernel', 256, 0), ('ac_decode_lean_kernel', 256, 0)],
}


def kernels(isa):
    """{kernel name: {'vgpr': n, 'spill': n}} from the .amdhsa metadata at the end of the listing"""
    out, name = {}, None
    for line in isa:
        m = re.match(r'\s*\.name:\s+(\S+)', line)
        if m:
            name = m.group(1)
            out.setdefault(name, {})
        m = re.match(r'\s*\.vgpr_count:\s+(\d+)', line)
        if m and name:
            out[name]['vgpr'] = int(m.group(1))
        m = re.match(r'\s*\.vgpr_spill_count:\s+(\d+)', line)
        if m and name:
            out[name]['spill'] = int(m.group(1))
    return out


def main():
    bad = 0
    for src, rules in BUDGET.items():
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'k.s')
            compile_to_isa(os.path.join(CSRC, src), out)
            ks = kernels(open(out).read().split('\n'))
        for sub, max_vgpr, max_spill in rules:
            hit = {n: v for n, v in ks.items() if sub in n and 'vgpr' in v}
            if not hit:
                print('{}: no kernel matching {!r}'.format(src, sub))
                bad += 1
            for n, v in sorted(hit.items()):
                ok = v['vgpr'] <= max_vgpr and v.get('spill', 0) <= max_spill
                print('{}: {:<70s} {:3d} VGPRs, {} spilled{}'.format(src, n[:70], v['vgpr'], v.get('spill', 0), '' if ok else '   <-- over budget'))
                bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
