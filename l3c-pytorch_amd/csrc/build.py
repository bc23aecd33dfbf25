"""Build libl3c_hip.so for gfx950 with hipcc (cross-compiles without a GPU), in-tree next to the sources.

    python l3c-pytorch_amd/csrc/build.py [--force] [--verbose]

-ffp-contract=off: every fp32 op of the mixture head is individually rounded so that the encoder's fused interval
kernel and the decoder's table kernel produce identical entries (dmll_kernels.hip); kernels that want an FMA say fmaf().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['l3c_api.hip', 'ac_kernels.hip', 'dmll_kernels.hip', 'conv_mfma.hip', 'conv_small.hip', 'container.hip', 'conv_pw.hip', 'conv_wino4.hip', 'decode_pipeline.hip']
# test-only second library (include/l3c_xcheck.h): the round-1/2 Winograd F(2x2,3x3) kernel, an independent implementation the tests
# compare the product's kernels with.  The product library does not contain it and the package never loads it outside the tests.
XCHECK_SOURCES = ['l3c_api.hip', 'conv_wino.hip', 'xcheck_dmll.hip', 'conv_wino4w.hip']
HEADERS = ['ac_core.h', 'dmll_core.h', 'l3c_common.h', os.path.join('..', '..', 'include', 'l3c_hip.h'), os.path.join('..', '..', 'include', 'l3c_xcheck.h')]
LIB = os.path.join(HERE, 'libl3c_hip.so')
XCHECK_LIB = os.path.join(HERE, 'libl3c_hip_xcheck.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
# per-source flags.  The MFMA kernels: no SLP vectorisation -- hipcc otherwise packs adjacent scalar fp32 adds of the input
# transform into v_pk_add_f32 / v_pk_fma_f32, which beside MFMAs cost more issue time than the plain instructions they replace.
SOURCE_FLAGS = {'conv_wino.hip': ['-fno-slp-vectorize'], 'conv_pw.hip': ['-fno-slp-vectorize'], 'conv_wino4.hip': ['-fno-slp-vectorize'],
                'conv_wino4w.hip': ['-fno-slp-vectorize']}


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), lib=LIB, objdir='_obj', xcheck=True):
    """extra / lib / objdir: development variants next to the product build (csrc/build.py --variant NAME FLAGS, loaded through the
    L3C_LIB environment variable).  xcheck: also (re)build the test-only cross-check library next to the product one."""
    if xcheck and lib == LIB:
        _build(XCHECK_SOURCES, force, verbose, extra, XCHECK_LIB, objdir)
    return _build(SOURCES, force, verbose, extra, lib, objdir)


def _build(sources, force, verbose, extra, lib, objdir):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
    procs = []
    for src in sources:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(src, []) + list(extra) + os.environ.get('HIPCC_EXTRA', '').split() + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on {}:\n{}'.format(src, out.decode()))
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    kw = dict(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv)
    if '--dev-probes' in sys.argv:       # development: l3c_conv_mfma honours the probe bits of `epilogue` (tools/conv_probe.py)
        kw.update(extra=['-DL3C_DEV_PROBES'], lib=os.path.join(HERE, 'libl3c_hip_devprobes.so'), objdir='_obj_devprobes')
    if '--variant' in sys.argv:          # development: an A/B build with extra compiler flags: --variant NAME "-DX -fno-slp-vectorize"
        n = sys.argv[sys.argv.index('--variant') + 1]
        flags = sys.argv[sys.argv.index('--variant') + 2].split()
        kw.update(extra=flags, lib=os.path.join(HERE, 'libl3c_hip_{}.so'.format(n)), objdir='_obj_l3c_' + n)
        if '--xcheck' in sys.argv:       # ... of the TEST-ONLY library instead (loaded through L3C_XCHECK_LIB): the probe kernels live there
            print(_build(XCHECK_SOURCES, kw['force'], kw['verbose'], flags, os.path.join(HERE, 'libl3c_hip_xcheck_{}.so'.format(n)), '_obj_l3c_x' + n))
            sys.exit(0)
    print(build(**kw))
