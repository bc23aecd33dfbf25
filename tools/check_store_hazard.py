#!/usr/bin/env python
"""Build-time check for a gfx950 store-data hazard hipcc does not model (found with conv_wino.hip).

A `buffer_store_dwordx3/x4` whose soffset is a REGISTER gets no wait states from the compiler before a following write to its
data VGPRs -- and on the MI355X such a VALU write issued right behind the store overtakes the store's read of its last
dwords (the last lanes of every 16-lane group stored the next value).  This script compiles every kernel source to ISA and
flags a >8-byte buffer store with a register soffset that is followed, within WINDOW instructions and with no intervening
wait, by an instruction writing one of its data registers.

    python tools/check_store_hazard.py        (exit status 0 = ok)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'l3c-pytorch_amd', 'csrc')
WINDOW = 2
STORE = re.compile(r'^(buffer_store_dwordx[34])\s+(v\[\d+:\d+\]),\s*[^,]+,\s*s\[\d+:\d+\],\s*(\S+)')
# VALU writes only: a load into the store's data registers returns its data tens of cycles after the store has read them (conv_wino.hip,
# compiled with the product flags, has `buffer_store_dwordx4 v[0:3] ... ; ds_read_b128 v[0:3]` pairs and is bit-exact in tests/test_gpu_conv.py)
WRITERS = ('v_',)


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def product_flags(src):
    """The flags csrc/build.py compiles `src` with (the scan has to look at the ISA that ships: -fno-slp-vectorize changes the schedule)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('l3c_csrc_build', os.path.join(CSRC, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return [f for f in mod.FLAGS if f not in ('-fPIC', '-Wall', '-Wno-unused-function')] + mod.SOURCE_FLAGS.get(os.path.basename(src), [])


def compile_to_isa(src, out):
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + product_flags(src) + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                                                                          '-o', out, src], cwd=os.path.dirname(out), stderr=subprocess.DEVNULL)


def scan(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        compile_to_isa(src, out)
        lines = [l.split(';')[0].strip() for l in open(out)]
    ins = [l for l in lines if l and not l.startswith('.') and not l.endswith(':')]
    hits = []
    for i, l in enumerate(ins):
        m = STORE.match(l)
        if not m or not m.group(3).startswith('s'):     # immediate soffset: the compiler inserts the wait states itself
            continue
        data = regs(m.group(2))
        for n in ins[i + 1:i + 1 + WINDOW]:
            if n.startswith(('s_waitcnt', 's_nop', 's_barrier')):
                break
            if n.startswith(WRITERS) and not n.startswith('v_cmp') and regs(n.split()[1].rstrip(',')) & data:
                hits.append((l, n))
                break
    return hits


def main():
    bad = 0
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.hip'):
            hits = scan(os.path.join(CSRC, f))
            print('{}: {} store(s) with a write to their data registers right behind'.format(f, len(hits)))
            for l, n in hits[:5]:
                print('   ', l, ' -> ', n)
            bad += len(hits)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
