"""Import the UNMODIFIED reference (fab-jul/L3C-PyTorch) from /root/reference/src.

TEST INFRASTRUCTURE ONLY (build container only — /root/reference does not exist on the GPU box).
Used by `tests/golden/make_golden.py` to generate the committed fixtures and by the `-m "not gpu"`
tests that validate the oracle restatements against the real reference when it is present.

Recipe: stubs for fjcommon/torchvision/fasteners on sys.path, the reference's own
`torchac.cpp` built as `torchac_backend_cpu` (oracle/build_ref.py), cwd = /root/reference/src because the
reference resolves `configs/...` relative to the cwd (config_checker.py:22, multiscale_tester.py:230).
"""
import contextlib
import os
import sys

REF_SRC = '/root/reference/src'
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(REF_SRC)


@contextlib.contextmanager
def reference_modules():
    """Context manager: inside it `import blueprints...`, `import bitcoding...`, `from torchac import torchac`
    resolve to the reference.  Restores sys.path / cwd / removes the reference's top-level modules afterwards so
    they cannot shadow anything else (the reference has top-level packages called `test`, `helpers`, ...)."""
    import build_ref  # noqa  (same dir)
    assert available(), 'reference sources not present'
    build_ref.build()
    saved_path = list(sys.path)
    saved_cwd = os.getcwd()
    saved_mods = set(sys.modules)
    shadowed = {k: sys.modules.pop(k) for k in list(sys.modules)
                if k == 'test' or k.startswith('test.')}
    sys.path[:0] = [os.path.join(_HERE, 'shims'), build_ref.OUT_DIR, REF_SRC]
    os.chdir(REF_SRC)
    try:
        yield
    finally:
        os.chdir(saved_cwd)
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_mods:
                m = sys.modules[k]
                f = getattr(m, '__file__', None) or ''
                if f.startswith(REF_SRC) or f.startswith(os.path.join(_HERE, 'shims')):
                    del sys.modules[k]
        sys.modules.update(shadowed)


sys.path.insert(0, _HERE) if _HERE not in sys.path else None
