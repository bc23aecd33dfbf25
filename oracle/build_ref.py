"""Build the REAL reference range coder (test infrastructure only).

Compiles `/root/reference/src/torchac/torchac_backend/torchac.cpp` *where it lies* (no source is copied
into this repo) as the pybind11 torch extension `torchac_backend_cpu` (module name from the reference's
setup.py:49,84) into `oracle/_ref/`.  `-DAT_CHECK=TORCH_CHECK` is the only adaptation needed for
torch 2.x.  The resulting .so is git-ignored but travels to the GPU box with the gpurun snapshot.

Used by: tests (validates oracle/ac_oracle.c and pins the HIP coder bit-exactly), bench.py's
`cpu_baseline` leg.  Never imported by the product package.
"""
import glob
import os
import sys

REF_CPP = '/root/reference/src/torchac/torchac_backend/torchac.cpp'
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')


def built_path():
    c = glob.glob(os.path.join(OUT_DIR, 'torchac_backend_cpu*.so'))
    return c[0] if c else None


def build(force=False, verbose=False):
    """Returns the path of the built module, or None when the reference sources are absent."""
    if built_path() and not force:
        return built_path()
    if not os.path.isfile(REF_CPP):
        return None
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    cpp_extension.load(name='torchac_backend_cpu', sources=[REF_CPP],
                       extra_cflags=['-DAT_CHECK=TORCH_CHECK', '-O2'],
                       build_directory=OUT_DIR, verbose=verbose, is_python_module=False)
    return built_path()


def load():
    """Import and return the reference module `torchac_backend_cpu` (None if it is not built)."""
    p = built_path()
    if p is None:
        return None
    import torch  # noqa: F401  (the extension links against libtorch)
    if OUT_DIR not in sys.path:
        sys.path.insert(0, OUT_DIR)
    import importlib
    return importlib.import_module('torchac_backend_cpu')


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
