"""Stand-in for fjcommon.no_op: an object that swallows every call / attribute / context."""


class _NoOp(object):
    def __getattr__(self, name):
        return self

    def __call__(self, *a, **kw):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return iter(())


NoOp = _NoOp()
